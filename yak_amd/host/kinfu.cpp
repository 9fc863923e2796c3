// kfusion::KinFu -- the per-frame loop (reference yak/src/kfusion/kinfu.cpp:10-309) and
// ProjectiveICP's parameter front (reference yak/src/kfusion/projective_icp.cpp:66-104).
#include <yak/kfusion/cuda/imgproc.hpp>
#include <yak/kfusion/kinfu.hpp>

#include <iostream>
#include <stdexcept>

namespace kfusion
{
// kinfu.cpp:10-48
KinFuParams KinFuParams::default_params()
{
  const int iters[] = { 10, 5, 4, 0 };
  KinFuParams p;
  p.cols = 640;
  p.rows = 480;
  p.intr = Intr(580.f, 580.f, p.cols / 2 - 0.5f, p.rows / 2 - 0.5f);
  p.volume_dims = Vec3i::all(512);
  p.volume_resolution = 0.005859375f;
  p.volume_pose = Affine3f::Identity().translate(Vec3f(0, 0, 0));
  p.bilateral_sigma_depth = 0.04f;
  p.bilateral_sigma_spatial = 4.5f;
  p.bilateral_kernel_size = 7;
  p.icp_truncate_depth_dist = 0.f;
  p.icp_dist_thres = 0.1f;
  p.icp_angle_thres = deg2rad(30.f);
  p.icp_iter_num.assign(iters, iters + 4);
  p.tsdf_min_camera_movement = 0.f;
  p.tsdf_trunc_dist = 0.04f;
  p.tsdf_max_weight = 64;
  p.raycast_step_factor = 0.75f;
  p.gradient_delta_factor = 0.5f;
  p.light_pose = Vec3f::all(0.f);
  p.use_pose_hints = false;
  p.use_icp = true;
  p.update_via_sensor_motion = true;
  return p;
}

namespace cuda
{
// projective_icp.cpp:66-104
ProjectiveICP::ProjectiveICP() : angle_thres_(deg2rad(20.f)), dist_thres_(0.1f)
{
  const int iters[] = { 10, 5, 4, 0 };
  iters_.assign(iters, iters + 4);
}
ProjectiveICP::~ProjectiveICP() {}
float ProjectiveICP::getDistThreshold() const { return dist_thres_; }
void ProjectiveICP::setDistThreshold(float distance) { dist_thres_ = distance; }
float ProjectiveICP::getAngleThreshold() const { return angle_thres_; }
void ProjectiveICP::setAngleThreshold(float angle) { angle_thres_ = angle; }
void ProjectiveICP::setIterationsNum(const std::vector<int>& iters)
{
  if (iters.size() > MAX_PYRAMID_LEVELS)
    throw std::invalid_argument("ProjectiveICP: at most MAX_PYRAMID_LEVELS iteration counts");
  iters_ = iters;
}
int ProjectiveICP::getUsedLevelsNum() const  // number of leading levels with a non-zero count
{
  int i = MAX_PYRAMID_LEVELS - 1;
  for (; i >= 0; --i)
    if (i < (int)iters_.size() && iters_[i] > 0)
      break;
  return i + 1;
}
bool ProjectiveICP::estimateTransform(Affine3f&, Affine3f&, const Intr&, const PointsPyr&, const NormalsPyr,
                                      const PointsPyr, const NormalsPyr)
{
  throw std::logic_error("ProjectiveICP::estimateTransform: ICP kernels are not built yet (SURVEY section 8f rank 2); "
                         "run with KinFuParams::use_icp = false (pose hints)");
}
}  // namespace cuda

// kinfu.cpp:50-80
KinFu::KinFu(const KinFuParams& params) : frame_counter_(0), params_(params)
{
  volume_ = std::make_shared<cuda::TsdfVolume>(params_.volume_dims);
  volume_->setTruncDist(params_.tsdf_trunc_dist);
  volume_->setMaxWeight(params_.tsdf_max_weight);
  const Vec3f volumeSize(params_.volume_dims[0] * params_.volume_resolution,
                         params_.volume_dims[1] * params_.volume_resolution,
                         params_.volume_dims[2] * params_.volume_resolution);
  volume_->setSize(volumeSize);
  // the reference leaves the volume pose at identity (kinfu.cpp:62 is commented out)
  volume_->setRaycastStepFactor(params_.raycast_step_factor);
  volume_->setGradientDeltaFactor(params_.gradient_delta_factor);

  icp_ = std::make_shared<cuda::ProjectiveICP>();
  icp_->setDistThreshold(params_.icp_dist_thres);
  icp_->setAngleThreshold(params_.icp_angle_thres);
  icp_->setIterationsNum(params_.icp_iter_num);

  allocate_buffers();
  resetVolume();
  poses_.reserve(30000);
  poses_.push_back(params_.volume_pose);
}

const KinFuParams& KinFu::params() const { return params_; }
KinFuParams& KinFu::params() { return params_; }
const cuda::TsdfVolume& KinFu::tsdf() const { return *volume_; }
cuda::TsdfVolume& KinFu::tsdf() { return *volume_; }
const cuda::ProjectiveICP& KinFu::icp() const { return *icp_; }
cuda::ProjectiveICP& KinFu::icp() { return *icp_; }

void KinFu::allocate_buffers() { dists_.create(params_.rows, params_.cols); }  // kinfu.cpp:94-129 (fusion part)

void KinFu::resetPose()  // kinfu.cpp:131-146
{
  poses_.clear();
  poses_.reserve(30000);
  poses_.push_back(params_.volume_pose);
}
void KinFu::resetVolume()  // kinfu.cpp:148-153
{
  frame_counter_ = 0;
  volume_->clear();
}
Affine3f KinFu::getCameraPose(int time) const  // kinfu.cpp:155-160
{
  if (time > (int)poses_.size() || time < 0)
    time = (int)poses_.size() - 1;
  return poses_[time];
}

// kinfu.cpp:162-309
bool KinFu::operator()(const Affine3f& inputCameraMotion, const Affine3f& currentCameraPose,
                       const Affine3f& previousCameraPose, const cuda::Depth& depth, const cuda::Image& /*image*/)
{
  const KinFuParams& p = params_;
  cuda::computeDists(depth, dists_, p.intr);  // :171
  // :172-186 (bilateral filter, pyramid, point normals) only feed ICP; they come with it
  cuda::waitAllDefaultStream();  // :188

  if (frame_counter_ == 0)  // :191-201 -- integrates at the stored pose, reports false
  {
    volume_->integrate(dists_, poses_.back(), p.intr);
    return ++frame_counter_, false;
  }

  Affine3f cameraMotion = Affine3f::Identity();  // :214-220
  Affine3f cameraMotionCorrected;
  Affine3f cameraPoseCorrected = currentCameraPose;
  if (p.use_pose_hints)
    cameraMotion = inputCameraMotion;

  bool ok = true;  // :228-253
  if (p.use_icp)
  {
    ok = icp_->estimateTransform(cameraMotion, cameraMotionCorrected, p.intr, curr_.points_pyr, curr_.normals_pyr,
                                 prev_.points_pyr, prev_.normals_pyr);
    cameraPoseCorrected = previousCameraPose * cameraMotionCorrected;
  }
  else
  {
    cameraPoseCorrected = currentCameraPose;
    cameraMotionCorrected = cameraMotion;
  }
  if (!ok)
  {
    resetVolume();
    resetPose();
    return false;
  }

  if (p.update_via_sensor_motion)  // :256-268
    poses_.push_back(poses_.back() * cameraMotionCorrected);
  else
    poses_.push_back(cameraPoseCorrected);

  // :279-290 -- the gate uses the hint motion, not the corrected one
  const float rnorm = (float)yak::compat::norm(cameraMotion.rvec());
  const float tnorm = (float)yak::compat::norm(cameraMotion.translation());
  if ((rnorm + tnorm) / 2 >= p.tsdf_min_camera_movement)
    volume_->integrate(dists_, poses_.back(), p.intr);

  // :294-306 (raycast of the model for the next frame's ICP) comes with the raycast kernel
  return ++frame_counter_, true;
}
}  // namespace kfusion
