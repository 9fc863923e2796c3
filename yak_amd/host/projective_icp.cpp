// kfusion::cuda::ProjectiveICP (reference yak/src/kfusion/projective_icp.cpp:66-104, 175-232).
#include <yak/kfusion/cuda/projective_icp.hpp>

#include <yakhip.h>

#include <cmath>
#include <iostream>
#include <stdexcept>


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

// :175-232.  Coarse to fine.  The reference comes back to the host after every linearisation (stream
// sync, cv::determinant, cv::solve, compose); yakhip_icp_estimate chains accumulation, reduction, solve
// and pose update on the device and is waited for once (include/yakhip.h).
bool ProjectiveICP::estimateTransform(const Affine3f& affine, Affine3f& correctedAffine, const Intr& intr,
                                      const PointsPyr& vcurr, const NormalsPyr ncurr, const PointsPyr vprev,
                                      const NormalsPyr nprev)
{
  const int LEVELS = getUsedLevelsNum();
  if ((int)vcurr.size() < LEVELS || (int)ncurr.size() < LEVELS || (int)vprev.size() < LEVELS || (int)nprev.size() < LEVELS)
    throw std::invalid_argument("ProjectiveICP::estimateTransform: pyramids shorter than the used levels");
  correctedAffine = affine;
  if (LEVELS == 0)
    return true;

  std::vector<yakhip_icp_level> levels((size_t)LEVELS);
  for (int l = 0; l < LEVELS; ++l)
  {
    yakhip_icp_level& L = levels[(size_t)l];
    L.vcurr = vcurr[l].ptr(), L.vcurr_pitch = vcurr[l].step();
    L.ncurr = ncurr[l].ptr(), L.ncurr_pitch = ncurr[l].step();
    L.vprev = vprev[l].ptr(), L.vprev_pitch = vprev[l].step();
    L.nprev = nprev[l].ptr(), L.nprev_pitch = nprev[l].step();
    L.rows = nprev[l].rows(), L.cols = nprev[l].cols();  // :200-201
    L.iterations = iters_[l];
  }
  const Mat3f R = affine.rotation();
  const Vec3f t = affine.translation();
  const float tf[3] = { t[0], t[1], t[2] }, in[4] = { intr.fx, intr.fy, intr.cx, intr.cy };
  float Ro[9], to[3];
  int ok = 0;
  const Shard& sh = currentShard();
  if (sh.world > 1)
  {
    // every rank its band of rows, the 33 sums of each linearisation all-reduced on the stream, the same solve
    // from the same bits on every rank (include/yakhip.h)
    if (yakhip_icp_estimate_sharded(context(), sh.comm, R.val, tf, in, dist_thres_, angle_thres_, levels.data(), LEVELS, Ro,
                                    to, &ok) != 0)
      error(yakhip_last_error(), __FILE__, __LINE__, "yakhip_icp_estimate_sharded");
  }
  else if (yakhip_icp_estimate(context(), R.val, tf, in, dist_thres_, angle_thres_, levels.data(), LEVELS, Ro, to, &ok) != 0)
    error(yakhip_last_error(), __FILE__, __LINE__, "yakhip_icp_estimate");
  Mat3f Rm;
  for (int i = 0; i < 9; ++i)
    Rm.val[i] = Ro[i];
  correctedAffine = Affine3f(Rm, Vec3f(to[0], to[1], to[2]));
  if (!ok)
    std::cout << "ICP: singular normal equations" << std::endl;  // :217-222
  return ok != 0;
}
}  // namespace cuda
}  // namespace kfusion
