// yak/kfusion/types.hpp -- value types of Yak's API (reference yak/include/yak/kfusion/types.hpp:13-77).
// Math types come from OpenCV when it is installed, otherwise from compat/linalg.hpp.
#pragma once
#include <iosfwd>
#include <vector>

#include <yak/kfusion/cuda/device_array.hpp>
#include <yak/kfusion/exports.hpp>

#if defined(YAK_USE_OPENCV) && __has_include(<opencv2/core/affine.hpp>)
#include <opencv2/core/affine.hpp>
#include <opencv2/core/core.hpp>
#define YAK_HAVE_OPENCV 1
#else
#include <yak/kfusion/compat/linalg.hpp>
namespace cv
{
// source-compatibility aliases for callers written against OpenCV names
using yak::compat::Affine3f;
using yak::compat::Mat;
using yak::compat::Matx33f;
using yak::compat::Vec3f;
using yak::compat::Vec3i;
enum
{
  DECOMP_SVD = 1
};
}  // namespace cv
#endif

namespace kfusion
{
typedef cv::Matx33f Mat3f;
typedef cv::Vec3f Vec3f;
typedef cv::Vec3i Vec3i;
typedef cv::Affine3f Affine3f;

/** Pinhole intrinsics; operator()(level) gives the intrinsics of pyramid level `level`
 *  (reference types.hpp:18-25, precomp.cpp:10-14). */
struct KF_EXPORTS Intr
{
  float fx, fy, cx, cy;
  Intr();
  Intr(float fx, float fy, float cx, float cy);
  Intr operator()(int level_index) const;
};
KF_EXPORTS std::ostream& operator<<(std::ostream& os, const Intr& intr);

struct Point
{
  union
  {
    float data[4];
    struct
    {
      float x, y, z;
    };
  };
};
typedef Point Normal;

struct RGB
{
  union
  {
    struct
    {
      unsigned char b, g, r;
    };
    int bgra;
  };
};

namespace cuda
{
typedef DeviceMemory CudaData;
typedef DeviceArray2D<unsigned short> Depth;  // millimetres
typedef DeviceArray2D<unsigned short> Dists;  // IEEE half bits, metres along the ray
typedef DeviceArray2D<RGB> Image;
typedef DeviceArray2D<Normal> Normals;
typedef DeviceArray2D<Point> Cloud;

struct Frame
{
  bool use_points;
  std::vector<Depth> depth_pyr;
  std::vector<Cloud> points_pyr;
  std::vector<Normals> normals_pyr;
};
}  // namespace cuda

inline float deg2rad(float alpha) { return alpha * 0.017453293f; }
}  // namespace kfusion
