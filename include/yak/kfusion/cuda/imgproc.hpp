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
/** Blocks until everything queued so far is done (reference imgproc.cpp:21: cudaDeviceSynchronize). */
KF_EXPORTS void waitAllDefaultStream();
}  // namespace cuda
}  // namespace kfusion
