// yak/kfusion/cuda/imgproc.hpp -- depth pre-processing entry points
// (reference yak/include/yak/kfusion/cuda/imgproc.hpp; host wrappers yak/src/kfusion/imgproc.cpp).
#pragma once
#include <yak/kfusion/types.hpp>

namespace kfusion
{
namespace cuda
{
/** depth (mm) -> ray length (m, half): reference imgproc.cpp:45-49, imgproc.cu:259-283. */
KF_EXPORTS void computeDists(const Depth& depth, Dists& dists, const Intr& intr);
/** Edge-preserving smoothing of the depth frame (reference imgproc.cpp:3-11, imgproc.cu:10-56). */
KF_EXPORTS void depthBilateralFilter(const Depth& in, Depth& out, int ksz, float sigma_spatial, float sigma_depth);
/** depth > threshold (metres) -> 0, in place (reference imgproc.cpp:13, imgproc.cu:66-85). */
KF_EXPORTS void depthTruncation(Depth& depth, float threshold);
/** Half-resolution level of the depth pyramid (reference imgproc.cpp:15-19, imgproc.cu:94-136). */
KF_EXPORTS void depthBuildPyramid(const Depth& depth, Depth& pyramid, float sigma_depth);
/** Back-projection + triangle normals (reference imgproc.cpp:33-43, imgproc.cu:210-250). */
KF_EXPORTS void computePointNormals(const Intr& intr, const Depth& depth, Cloud& points, Normals& normals);
/** 2 x 2 box average of point / normal maps (reference imgproc.cpp:65-80, imgproc.cu:348-394). */
KF_EXPORTS void resizePointsNormals(const Cloud& points, const Normals& normals, Cloud& points_out, Normals& normals_out);
/** Two levels in one launch (no counterpart in the reference, which calls resizePointsNormals level by level:
 *  kinfu.cpp:294-306): the same maps, bit for bit. */
KF_EXPORTS void resizePointsNormals2(const Cloud& points, const Normals& normals, Cloud& points_half, Normals& normals_half,
                                     Cloud& points_quarter, Normals& normals_quarter);
/** Debug views, 4 bytes per pixel {b, g, r, 0}; both block like the reference's (imgproc.cpp:82-126,
 *  imgproc.cu:400-561).  renderImage: Phong shading of camera-frame maps lit from light_pose; the Depth
 *  overload re-projects with (fx, fy, cx, fy) exactly as imgproc.cpp:92 builds its Reprojector. */
KF_EXPORTS void renderTangentColors(const Normals& normals, Image& image);
KF_EXPORTS void renderImage(const Depth& depth, const Normals& normals, const Intr& intr, const Vec3f& light_pose,
                            Image& image);
KF_EXPORTS void renderImage(const Cloud& points, const Normals& normals, const Intr& intr, const Vec3f& light_pose,
                            Image& image);
/** Blocks until everything queued so far is done (reference imgproc.cpp:21: cudaDeviceSynchronize). */
KF_EXPORTS void waitAllDefaultStream();
}  // namespace cuda
}  // namespace kfusion
