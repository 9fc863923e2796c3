// yak/kfusion/compat/linalg.hpp -- the handful of OpenCV / Eigen value types that appear in Yak's
// public API (cv::Vec3f, cv::Vec3i, cv::Matx33f, cv::Affine3f, cv::Mat as a view, Eigen::Affine3f),
// restated with the same member names for builds where those libraries are absent (this image has
// neither).  When <opencv2/core/affine.hpp> / <Eigen/Geometry> exist, types.hpp and yak_server.h use
// the real ones instead and this header is not included.
//
// Arithmetic here is plain fp32 written for this project; it is NOT claimed to round like OpenCV's
// or Eigen's (the reference pins neither version: CMakeLists.txt:16,29).  Poses cross the device
// boundary as the 12 floats these types hold, so kernel parity is unaffected.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstring>

namespace yak
{
namespace compat
{
template <typename T, int N>
struct Vec
{
  T val[N];
  Vec()
  {
    for (int i = 0; i < N; ++i)
      val[i] = T(0);
  }
  Vec(T a, T b, T c)
  {
    static_assert(N == 3, "three-component constructor");
    val[0] = a;
    val[1] = b;
    val[2] = c;
  }
  static Vec all(T v)
  {
    Vec r;
    for (int i = 0; i < N; ++i)
      r.val[i] = v;
    return r;
  }
  T& operator[](int i) { return val[i]; }
  const T& operator[](int i) const { return val[i]; }
  T& operator()(int i) { return val[i]; }
  const T& operator()(int i) const { return val[i]; }
};
typedef Vec<float, 3> Vec3f;
typedef Vec<int, 3> Vec3i;

inline Vec3f operator+(const Vec3f& a, const Vec3f& b) { return Vec3f(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline Vec3f operator-(const Vec3f& a, const Vec3f& b) { return Vec3f(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline Vec3f operator*(const Vec3f& a, float s) { return Vec3f(a[0] * s, a[1] * s, a[2] * s); }
inline double norm(const Vec3f& v) { return std::sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]); }

// Row-major 3x3, as cv::Matx33f::val (the layout device_cast relies on, reference precomp.hpp:15-30).
struct Matx33f
{
  float val[9];
  Matx33f() { std::memset(val, 0, sizeof(val)); }
  static Matx33f eye()
  {
    Matx33f m;
    m.val[0] = m.val[4] = m.val[8] = 1.f;
    return m;
  }
  float& operator()(int r, int c) { return val[3 * r + c]; }
  float operator()(int r, int c) const { return val[3 * r + c]; }
  Matx33f t() const
  {
    Matx33f m;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        m(r, c) = (*this)(c, r);
    return m;
  }
  // adjugate / determinant inverse in double, rounded once (the reference asks for DECOMP_SVD)
  Matx33f inv(int /*method*/ = 0) const
  {
    const double a = val[0], b = val[1], c = val[2], d = val[3], e = val[4], f = val[5], g = val[6], h = val[7],
                 i = val[8];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    Matx33f m;
    if (det == 0.0)
      return m;
    const double s = 1.0 / det;
    m.val[0] = (float)((e * i - f * h) * s);
    m.val[1] = (float)((c * h - b * i) * s);
    m.val[2] = (float)((b * f - c * e) * s);
    m.val[3] = (float)((f * g - d * i) * s);
    m.val[4] = (float)((a * i - c * g) * s);
    m.val[5] = (float)((c * d - a * f) * s);
    m.val[6] = (float)((d * h - e * g) * s);
    m.val[7] = (float)((b * g - a * h) * s);
    m.val[8] = (float)((a * e - b * d) * s);
    return m;
  }
};

inline Matx33f operator*(const Matx33f& a, const Matx33f& b)
{
  Matx33f m;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      m(r, c) = a(r, 0) * b(0, c) + a(r, 1) * b(1, c) + a(r, 2) * b(2, c);
  return m;
}
inline Vec3f operator*(const Matx33f& a, const Vec3f& v)
{
  return Vec3f(a(0, 0) * v[0] + a(0, 1) * v[1] + a(0, 2) * v[2], a(1, 0) * v[0] + a(1, 1) * v[1] + a(1, 2) * v[2],
               a(2, 0) * v[0] + a(2, 1) * v[1] + a(2, 2) * v[2]);
}

struct Matx44f
{
  float val[16];
  Matx44f()
  {
    std::memset(val, 0, sizeof(val));
    val[0] = val[5] = val[10] = val[15] = 1.f;
  }
  float& operator()(int r, int c) { return val[4 * r + c]; }
  float operator()(int r, int c) const { return val[4 * r + c]; }
};

// cv::Affine3f look-alike: 4x4 row-major `matrix`, rotation() / translation() views, inv(), products.
struct Affine3f
{
  Matx44f matrix;
  Affine3f() {}
  Affine3f(const Matx44f& m) : matrix(m) {}
  Affine3f(const Matx33f& R, const Vec3f& t)
  {
    rotation(R);
    translation(t);
  }
  // Rodrigues vector + translation (cv::Affine3f(rvec, t), used by ICP: projective_icp.cpp:168,227)
  Affine3f(const Vec3f& rvec, const Vec3f& t)
  {
    const double th = norm(rvec);
    Matx33f R = Matx33f::eye();
    if (th > 2.220446049250313e-16)
    {
      const double kx = rvec[0] / th, ky = rvec[1] / th, kz = rvec[2] / th, c = std::cos(th), s = std::sin(th),
                   v = 1.0 - c;
      R(0, 0) = (float)(c + kx * kx * v);
      R(0, 1) = (float)(kx * ky * v - kz * s);
      R(0, 2) = (float)(kx * kz * v + ky * s);
      R(1, 0) = (float)(ky * kx * v + kz * s);
      R(1, 1) = (float)(c + ky * ky * v);
      R(1, 2) = (float)(ky * kz * v - kx * s);
      R(2, 0) = (float)(kz * kx * v - ky * s);
      R(2, 1) = (float)(kz * ky * v + kx * s);
      R(2, 2) = (float)(c + kz * kz * v);
    }
    rotation(R);
    translation(t);
  }
  static Affine3f Identity() { return Affine3f(); }

  Matx33f rotation() const
  {
    Matx33f R;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        R(r, c) = matrix(r, c);
    return R;
  }
  Vec3f translation() const { return Vec3f(matrix(0, 3), matrix(1, 3), matrix(2, 3)); }
  void rotation(const Matx33f& R)
  {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        matrix(r, c) = R(r, c);
  }
  void translation(const Vec3f& t)
  {
    matrix(0, 3) = t[0];
    matrix(1, 3) = t[1];
    matrix(2, 3) = t[2];
  }
  Affine3f translate(const Vec3f& t) const
  {
    Affine3f r = *this;
    r.matrix(0, 3) += t[0];
    r.matrix(1, 3) += t[1];
    r.matrix(2, 3) += t[2];
    return r;
  }
  Affine3f inv(int /*method*/ = 0) const
  {
    const Matx33f Ri = rotation().inv();
    const Vec3f t = translation();
    const Vec3f ti = Ri * t;
    return Affine3f(Ri, Vec3f(-ti[0], -ti[1], -ti[2]));
  }
  // rotation vector (cv::Affine3f::rvec), used for the camera-movement gate kinfu.cpp:279
  Vec3f rvec() const
  {
    const Matx33f R = rotation();
    double rx = R(2, 1) - R(1, 2), ry = R(0, 2) - R(2, 0), rz = R(1, 0) - R(0, 1);
    const double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R(0, 0) + R(1, 1) + R(2, 2) - 1.0) * 0.5;
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    const double theta = std::acos(c);
    if (s < 1e-5)
    {
      if (c > 0)
        return Vec3f(0, 0, 0);
      double tx = std::sqrt(std::fmax((R(0, 0) + 1) * 0.5, 0.0));
      double ty = std::sqrt(std::fmax((R(1, 1) + 1) * 0.5, 0.0)) * (R(0, 1) < 0 ? -1.0 : 1.0);
      double tz = std::sqrt(std::fmax((R(2, 2) + 1) * 0.5, 0.0)) * (R(0, 2) < 0 ? -1.0 : 1.0);
      if (std::fabs(tx) < std::fabs(ty) && std::fabs(tx) < std::fabs(tz) && (R(1, 2) > 0) != (ty * tz > 0))
        tz = -tz;
      const double k = theta / std::sqrt(tx * tx + ty * ty + tz * tz);
      return Vec3f((float)(tx * k), (float)(ty * k), (float)(tz * k));
    }
    const double k = theta / (2.0 * s);
    return Vec3f((float)(rx * k), (float)(ry * k), (float)(rz * k));
  }
};

inline Affine3f operator*(const Affine3f& a, const Affine3f& b)
{
  const Matx33f R = a.rotation() * b.rotation();
  const Vec3f t = a.rotation() * b.translation() + a.translation();
  return Affine3f(R, t);
}
inline Vec3f operator*(const Affine3f& a, const Vec3f& v) { return a.rotation() * v + a.translation(); }

// cv::Mat as FusionServer::fuse reads it: data pointer, row step in bytes, rows, cols.
struct Mat
{
  unsigned char* data;
  size_t step;
  int rows, cols;
  Mat() : data(nullptr), step(0), rows(0), cols(0) {}
  Mat(int rows_, int cols_, void* data_, size_t step_bytes)
    : data((unsigned char*)data_), step(step_bytes), rows(rows_), cols(cols_)
  {
  }
};

// Eigen::Affine3f look-alike for yak_server.h: inverse(), operator*, Identity(), matrix access.
struct EigenAffine3f
{
  Affine3f a;
  EigenAffine3f() {}
  explicit EigenAffine3f(const Affine3f& x) : a(x) {}
  static EigenAffine3f Identity() { return EigenAffine3f(); }
  EigenAffine3f inverse() const { return EigenAffine3f(a.inv()); }
  EigenAffine3f operator*(const EigenAffine3f& o) const { return EigenAffine3f(a * o.a); }
  float& operator()(int r, int c) { return a.matrix(r, c); }
  float operator()(int r, int c) const { return a.matrix(r, c); }
  const Matx44f& matrix() const { return a.matrix; }
};
}  // namespace compat
}  // namespace yak
