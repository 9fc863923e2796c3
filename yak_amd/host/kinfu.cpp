// kfusion::KinFu -- the per-frame loop (reference yak/src/kfusion/kinfu.cpp:10-309).
#include <yak/kfusion/cuda/imgproc.hpp>
#include <yak/kfusion/kinfu.hpp>

#include <yakhip.h>

#include <cstdlib>
#include <iostream>
#include <stdexcept>

#define YAK_SET_OPTION(key, value)                                                       \
  do                                                                                     \
  {                                                                                      \
    if (yakhip_ctx_set_option(cuda::context(), key, value) != 0)                         \
      cuda::error(yakhip_last_error(), __FILE__, __LINE__, "yakhip_ctx_set_option");     \
  } while (0)

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

  // Frame f + 1 is prepared on the context's second stream while frame f is still being integrated and ray cast,
  // and nothing waits at the end of a frame (the reference synchronises the device three times per frame:
  // kinfu.cpp:188, tsdf_volume.cu:139, kinfu.cpp:307); every later call is ordered behind the work in the stream
  // anyway.  YAK_HIP_OVERLAP=0 restores the waits.
  const char* env = std::getenv("YAK_HIP_OVERLAP");
  overlap_ = !(env && env[0] == '0');
  volume_->setBlockingIntegrate(!overlap_);

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

void KinFu::allocate_buffers()  // kinfu.cpp:94-129 (the USE_DEPTH-only depth pyramid of prev_ is not kept)
{
  const int LEVELS = cuda::ProjectiveICP::MAX_PYRAMID_LEVELS;
  int cols = params_.cols, rows = params_.rows;
  dists_.create(rows, cols);
  dists_alt_.create(rows, cols);
  curr_.depth_pyr.resize(LEVELS);
  curr_.normals_pyr.resize(LEVELS);
  curr_.points_pyr.resize(LEVELS);
  prev_.normals_pyr.resize(LEVELS);
  prev_.points_pyr.resize(LEVELS);
  for (int i = 0; i < LEVELS; ++i)
  {
    curr_.depth_pyr[i].create(rows, cols);
    curr_.normals_pyr[i].create(rows, cols);
    curr_.points_pyr[i].create(rows, cols);
    prev_.normals_pyr[i].create(rows, cols);
    prev_.points_pyr[i].create(rows, cols);
    cols /= 2;
    rows /= 2;
  }
  depths_.create(params_.rows, params_.cols);
  normals_.create(params_.rows, params_.cols);
  points_.create(params_.rows, params_.cols);
}

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

namespace
{
// kinfu.cpp:322-333 / :353-364
void render_views(const cuda::Cloud& points, const cuda::Normals& normals, const KinFuParams& p, cuda::Image& image, int flag)
{
  if (flag < 1 || flag > 3)
    cuda::renderImage(points, normals, p.intr, p.light_pose, image);
  else if (flag == 2)
    cuda::renderTangentColors(normals, image);
  else
  {
    // two views of the caller's image side by side; create() on a view of the right size keeps its memory
    cuda::DeviceArray2D<RGB> i1(p.rows, p.cols, image.ptr(), image.step());
    cuda::DeviceArray2D<RGB> i2(p.rows, p.cols, image.ptr() + p.cols, image.step());
    cuda::renderImage(points, normals, p.intr, p.light_pose, i1);
    cuda::renderTangentColors(normals, i2);
  }
}
}  // namespace

// kinfu.cpp:311-335
void KinFu::renderImage(cuda::Image& image, int flag)
{
  const KinFuParams& p = params_;
  image.create(p.rows, flag != 3 ? p.cols : p.cols * 2);
  render_views(prev_.points_pyr[0], prev_.normals_pyr[0], p, image, flag);
}

// kinfu.cpp:337-366
void KinFu::renderImage(cuda::Image& image, const Affine3f& pose, int flag)
{
  const KinFuParams& p = params_;
  image.create(p.rows, flag != 3 ? p.cols : p.cols * 2);
  depths_.create(p.rows, p.cols);
  normals_.create(p.rows, p.cols);
  points_.create(p.rows, p.cols);
  volume_->raycast(pose, p.intr, points_, normals_);
  render_views(points_, normals_, p, image, flag);
}

// kinfu.cpp:368-377
std::vector<Point> KinFu::downloadCloud() const
{
  cuda::DeviceArray<Point> cloud_buffer;
  cuda::DeviceArray<Point> extracted = volume_->fetchCloud(cloud_buffer);
  std::vector<Point> result(extracted.size());
  if (!result.empty())
    extracted.download(result.data());
  return result;
}

void KinFu::downloadModelMaps(void* points_out, void* normals_out) const
{
  const cuda::Cloud& p = prev_.points_pyr[0];
  const cuda::Normals& n = prev_.normals_pyr[0];
  p.download(points_out, (size_t)p.cols() * sizeof(Point));
  n.download(normals_out, (size_t)n.cols() * sizeof(Normal));
}

// kinfu.cpp:162-309
void KinFu::beginFrame()
{
  if (overlap_ && yakhip_ctx_side_begin(cuda::context()) != 0)
    cuda::error(yakhip_last_error(), __FILE__, __LINE__, "yakhip_ctx_side_begin");
}

bool KinFu::operator()(const Affine3f& inputCameraMotion, const Affine3f& currentCameraPose,
                       const Affine3f& previousCameraPose, const cuda::Depth& depth, const cuda::Image& /*image*/)
{
  const KinFuParams& p = params_;
  const int LEVELS = icp_->getUsedLevelsNum();

  cuda::Dists& dists = (frame_counter_ & 1) ? dists_alt_ : dists_;
  beginFrame();  // (goes on with the section if the caller opened it for the upload)
  cuda::computeDists(depth, dists, p.intr);  // :171
  if (overlap_)
    volume_->prepareIntegrate(dists);  // (beside the previous frame's work; integrate() below then starts at the walk)
  // :172-186 feed ICP only; the reference runs them on every frame regardless of use_icp, and so
  // does this loop (prev_ must hold frame 0's maps when tracking starts on frame 1)
  cuda::depthBilateralFilter(depth, curr_.depth_pyr[0], p.bilateral_kernel_size, p.bilateral_sigma_spatial,
                             p.bilateral_sigma_depth);
  if (p.icp_truncate_depth_dist > 0)
    cuda::depthTruncation(curr_.depth_pyr[0], p.icp_truncate_depth_dist);
  for (int i = 1; i < LEVELS; ++i)
    cuda::depthBuildPyramid(curr_.depth_pyr[i - 1], curr_.depth_pyr[i], p.bilateral_sigma_depth);
  for (int i = 0; i < LEVELS; ++i)
    cuda::computePointNormals(p.intr(i), curr_.depth_pyr[i], curr_.points_pyr[i], curr_.normals_pyr[i]);
  if (overlap_)
  {
    if (yakhip_ctx_side_end(cuda::context()) != 0)
      cuda::error(yakhip_last_error(), __FILE__, __LINE__, "yakhip_ctx_side_end");
  }
  else
    cuda::waitAllDefaultStream();  // :188

  if (frame_counter_ == 0)  // :191-201 -- integrates at the stored pose, reports false
  {
    volume_->integrate(dists, poses_.back(), p.intr);
    curr_.points_pyr.swap(prev_.points_pyr);
    curr_.normals_pyr.swap(prev_.normals_pyr);
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
    volume_->integrate(dists, poses_.back(), p.intr);

  // :294-306 -- the model as seen from the new pose becomes the next frame's ICP target
  volume_->raycastPyramid(poses_.back(), p.intr, prev_.points_pyr, prev_.normals_pyr, LEVELS);
  if (!overlap_ || cuda::currentShard().world > 1)
    cuda::waitAllDefaultStream();
  if (volume_->raycastOverflowed())  // sharded volume only: more pixels to settle than the composite's list holds
  {
    YAK_SET_OPTION("composite_capacity", p.rows * p.cols);
    volume_->raycast(poses_.back(), p.intr, prev_.points_pyr[0], prev_.normals_pyr[0]);
    for (int i = 1; i < LEVELS; ++i)
      cuda::resizePointsNormals(prev_.points_pyr[i - 1], prev_.normals_pyr[i - 1], prev_.points_pyr[i],
                                prev_.normals_pyr[i]);
    cuda::waitAllDefaultStream();
    YAK_SET_OPTION("composite_capacity", 0);
  }
  return ++frame_counter_, true;
}
}  // namespace kfusion
