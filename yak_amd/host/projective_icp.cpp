// kfusion::cuda::ProjectiveICP (reference yak/src/kfusion/projective_icp.cpp:66-104, 175-232).
#include <yak/kfusion/cuda/projective_icp.hpp>

#include <yakhip.h>

#include <cmath>
#include <iostream>
#include <stdexcept>

#include "solve6.hpp"

namespace kfusion
{
namespace cuda
{
// projective_icp.cpp:67-75
ProjectiveICP::ProjectiveICP() : angle_thres_(deg2rad(20.f)), dist_thres_(0.1f)
{
  const int iters[] = { 10, 5, 4, 0 };
  setIterationsNum(std::vector<int>(iters, iters + 4));
}
ProjectiveICP::~ProjectiveICP() {}
float ProjectiveICP::getDistThreshold() const { return dist_thres_; }
void ProjectiveICP::setDistThreshold(float distance) { dist_thres_ = distance; }
float ProjectiveICP::getAngleThreshold() const { return angle_thres_; }
void ProjectiveICP::setAngleThreshold(float angle) { angle_thres_ = angle; }

// :87-96 -- longer lists are cut to MAX_PYRAMID_LEVELS, shorter ones padded with 0
void ProjectiveICP::setIterationsNum(const std::vector<int>& iters)
{
  if (iters.size() >= (size_t)MAX_PYRAMID_LEVELS)
    iters_.assign(iters.begin(), iters.begin() + MAX_PYRAMID_LEVELS);
  else
  {
    iters_ = std::vector<int>(MAX_PYRAMID_LEVELS, 0);
    std::copy(iters.begin(), iters.end(), iters_.begin());
  }
}

// :98-104 -- index of the last level with a non-zero count, plus one
int ProjectiveICP::getUsedLevelsNum() const
{
  int i = MAX_PYRAMID_LEVELS - 1;
  for (; i >= 0 && !iters_[i]; --i)
    ;
  return i + 1;
}

// :175-232.  Coarse to fine; per iteration one device pass gives the 27 sums of the normal
// equations (blocking), the 6 x 6 system is solved on the host and the increment is composed on the
// left of the running estimate.
bool ProjectiveICP::estimateTransform(const Affine3f& affine, Affine3f& correctedAffine, const Intr& intr,
                                      const PointsPyr& vcurr, const NormalsPyr ncurr, const PointsPyr vprev,
                                      const NormalsPyr nprev)
{
  const int LEVELS = getUsedLevelsNum();
  if ((int)vcurr.size() < LEVELS || (int)ncurr.size() < LEVELS || (int)vprev.size() < LEVELS || (int)nprev.size() < LEVELS)
    throw std::invalid_argument("ProjectiveICP::estimateTransform: pyramids shorter than the used levels");
  correctedAffine = affine;
  const float in[4] = { intr.fx, intr.fy, intr.cx, intr.cy };

  for (int level_index = LEVELS - 1; level_index >= 0; --level_index)
  {
    const Normals& n = nprev[level_index];
    const Cloud& v = vprev[level_index];
    const Cloud& vc = vcurr[level_index];
    const Normals& nc = ncurr[level_index];

    for (int iter = 0; iter < iters_[level_index]; ++iter)
    {
      const Mat3f R = correctedAffine.rotation();
      const Vec3f t = correctedAffine.translation();
      const float tf[3] = { t[0], t[1], t[2] };
      float sums[27];
      if (yakhip_icp_accumulate(context(), R.val, tf, in, level_index, dist_thres_, angle_thres_, vc.ptr(), vc.step(),
                                nc.ptr(), nc.step(), v.ptr(), v.step(), n.ptr(), n.step(), n.rows(), n.cols(), sums,
                                nullptr) != 0)
        error(yakhip_last_error(), __FILE__, __LINE__, "yakhip_icp_accumulate");

      // StreamHelper::get, :42-61 (A is float in the reference; the values are widened unchanged)
      double A[36], b[6];
      int shift = 0;
      for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j)
        {
          const double value = sums[shift++];
          if (j == 6)
            b[i] = value;
          else
            A[j * 6 + i] = A[i * 6 + j] = value;
        }

      // :214-222 checking nullspace
      const double det = yak::host::det6(A);
      if (std::fabs(det) < 1e-15 || std::isnan(det))
      {
        if (std::isnan(det))
          std::cout << "qnan" << std::endl;
        return false;
      }

      double r[6];
      yak::host::solve6_svd_symmetric(A, b, r);  // :224-225
      const Affine3f Tinc(Vec3f((float)r[0], (float)r[1], (float)r[2]), Vec3f((float)r[3], (float)r[4], (float)r[5]));
      correctedAffine = Tinc * correctedAffine;  // :227-228
    }
  }
  return true;
}
}  // namespace cuda
}  // namespace kfusion
