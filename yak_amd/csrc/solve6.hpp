// solve6.hpp -- the two pieces of dense linear algebra ProjectiveICP asks OpenCV for, restated without
// it: cv::determinant of the 6 x 6 normal matrix and cv::solve(A, b, x, DECOMP_SVD)
// (reference yak/src/kfusion/projective_icp.cpp:215, 225).  Shared by the C++ host layer
// (ProjectiveICP::estimateTransform) and by the slow path of yakhip_icp_estimate; not installed.
//
// OpenCV is un-vendored and unpinned in the reference (CMakeLists.txt:16), so this is "parity
// unpinned" by construction: same mathematics (LU determinant; minimum-norm least squares through the
// singular value decomposition with OpenCV's relative cut-off for small singular values), computed
// in double, not the same rounding.
#pragma once
#include <cfloat>
#include <cmath>

namespace yak
{
namespace host
{
// Determinant by LU with partial pivoting.
inline double det6(const double A_in[36])
{
  double A[36];
  for (int i = 0; i < 36; ++i)
    A[i] = A_in[i];
  double det = 1.0;
  for (int k = 0; k < 6; ++k)
  {
    int piv = k;
    for (int r = k + 1; r < 6; ++r)
      if (std::fabs(A[r * 6 + k]) > std::fabs(A[piv * 6 + k]))
        piv = r;
    if (A[piv * 6 + k] == 0.0)
      return 0.0;
    if (piv != k)
    {
      for (int c = 0; c < 6; ++c)
      {
        const double tmp = A[k * 6 + c];
        A[k * 6 + c] = A[piv * 6 + c];
        A[piv * 6 + c] = tmp;
      }
      det = -det;
    }
    det *= A[k * 6 + k];
    for (int r = k + 1; r < 6; ++r)
    {
      const double f = A[r * 6 + k] / A[k * 6 + k];
      for (int c = k; c < 6; ++c)
        A[r * 6 + c] -= f * A[k * 6 + c];
    }
  }
  return det;
}

// x = pinv(A) b for a SYMMETRIC 6 x 6 A (the normal matrix is).  For a symmetric matrix the singular
// values are |eigenvalues| and the singular vectors the eigenvectors, so the SVD back-substitution is
// x = sum_i (q_i . b / lambda_i) q_i over |lambda_i| > threshold, threshold = 2 eps_float * sum |lambda|
// (the cut-off OpenCV's SVD::backSubst applies to a float system).  Cyclic Jacobi in double.
inline void solve6_svd_symmetric(const double A_in[36], const double b[6], double x[6])
{
  double A[36], V[36];
  for (int i = 0; i < 36; ++i)
  {
    A[i] = A_in[i];
    V[i] = (i % 7 == 0) ? 1.0 : 0.0;
  }
  for (int sweep = 0; sweep < 64; ++sweep)
  {
    double off = 0.0;
    for (int p = 0; p < 6; ++p)
      for (int q = p + 1; q < 6; ++q)
        off += A[p * 6 + q] * A[p * 6 + q];
    if (off < 1e-300)
      break;
    for (int p = 0; p < 6; ++p)
      for (int q = p + 1; q < 6; ++q)
      {
        const double apq = A[p * 6 + q];
        if (apq == 0.0)
          continue;
        const double theta = (A[q * 6 + q] - A[p * 6 + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 6; ++k)
        {  // A <- A J
          const double akp = A[k * 6 + p], akq = A[k * 6 + q];
          A[k * 6 + p] = c * akp - s * akq;
          A[k * 6 + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 6; ++k)
        {  // A <- J^T A
          const double apk = A[p * 6 + k], aqk = A[q * 6 + k];
          A[p * 6 + k] = c * apk - s * aqk;
          A[q * 6 + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 6; ++k)
        {  // V <- V J
          const double vkp = V[k * 6 + p], vkq = V[k * 6 + q];
          V[k * 6 + p] = c * vkp - s * vkq;
          V[k * 6 + q] = s * vkp + c * vkq;
        }
      }
  }
  double sum = 0.0;
  for (int i = 0; i < 6; ++i)
    sum += std::fabs(A[i * 6 + i]);
  const double threshold = sum * 2.0 * (double)FLT_EPSILON;
  for (int i = 0; i < 6; ++i)
    x[i] = 0.0;
  for (int i = 0; i < 6; ++i)
  {
    const double lambda = A[i * 6 + i];
    if (std::fabs(lambda) <= threshold)
      continue;
    double qb = 0.0;
    for (int k = 0; k < 6; ++k)
      qb += V[k * 6 + i] * b[k];
    const double f = qb / lambda;
    for (int k = 0; k < 6; ++k)
      x[k] += f * V[k * 6 + i];
  }
}
}  // namespace host
}  // namespace yak
