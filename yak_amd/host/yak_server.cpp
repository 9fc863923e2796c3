// yak::FusionServer and yak::TSDFContainer (reference yak/src/yak_server.cpp:4-110,
// yak/src/kfusion/tsdf_container.cpp).
#include <yak/yak_server.h>

#include <cstdlib>
#include <stdexcept>
#if defined(__linux__)
#include <sys/mman.h>
#endif

namespace yak
{
namespace detail
{
void* lazy_allocate(std::size_t bytes)
{
  const std::size_t huge = (std::size_t)2 << 20;
  void* p = nullptr;
  if (bytes >= 16 * huge)
  {
    if (posix_memalign(&p, huge, (bytes + huge - 1) / huge * huge) != 0)
      p = nullptr;
#if defined(__linux__) && defined(MADV_HUGEPAGE)
    if (p)
      madvise(p, bytes, MADV_HUGEPAGE);  // advice only: a refusal changes nothing but speed
#endif
  }
  else
    p = std::malloc(bytes ? bytes : 1);
  if (!p)
    throw std::bad_alloc();
  return p;
}
}  // namespace detail

typedef std::vector<std::uint32_t, detail::LazyAllocator<std::uint32_t>> VoxelVector;

TSDFContainer::TSDFContainer(int x, int y, int z)
  : data_(std::make_shared<VoxelVector>((std::size_t)x * (std::size_t)y * (std::size_t)z, 0u))
{
  dims_.v[0] = x;
  dims_.v[1] = y;
  dims_.v[2] = z;
}
TSDFContainer::TSDFContainer(int x, int y, int z, Uninitialized)
  : data_(std::make_shared<VoxelVector>((std::size_t)x * (std::size_t)y * (std::size_t)z))
{
  dims_.v[0] = x;
  dims_.v[1] = y;
  dims_.v[2] = z;
}
float TSDFContainer::distance(std::size_t index) const
{
  std::uint16_t d, w;
  read(index, d, w);
  return kfusion::cuda::TsdfVolume::Entry::half2float(d);
}

namespace
{
kfusion::Affine3f to_kf(const Eigen::Affine3f& e)
{
  kfusion::Affine3f a;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c)
      a.matrix(r, c) = e(r, c);
  return a;
}
}  // namespace

// yak_server.cpp:4-13 (no window)
void FusionServer::create(const kfusion::KinFuParams& params)
{
  const std::vector<int> devices = kfusion::ShardedKinFu::devicesFromEnvironment();
  kinfu_.reset();
  sharded_.reset();
  if (devices.size() > 1)
    sharded_.reset(new kfusion::ShardedKinFu(params, devices));
  else
  {
    if (devices.size() == 1)
      kfusion::cuda::setDevice(devices[0]);
    kinfu_.reset(new kfusion::KinFu(params));
  }
}

FusionServer::FusionServer(const kfusion::KinFuParams& params, const Eigen::Affine3f& world_to_volume)
  : volume_to_world_(world_to_volume.inverse())
  , last_camera_pose_(Eigen::Affine3f::Identity())
  , display_every_frame_(false)
{
  create(params);
  const char* env = std::getenv("YAK_HIP_DISPLAY");
  display_every_frame_ = env && std::atoi(env) != 0;
}

// yak_server.cpp:15-39
bool FusionServer::fuse(const cv::Mat& depth_data, const Eigen::Affine3f& world_to_camera)
{
  const Eigen::Affine3f current_camera_in_volume = volume_to_world_ * world_to_camera;
  Eigen::Affine3f motion = current_camera_in_volume * last_camera_pose_.inverse();
  motion = motion.inverse();
  bool result;
  if (sharded_)  // every rank uploads the frame to its own GPU
    result = sharded_->operator()(to_kf(motion), to_kf(current_camera_in_volume), to_kf(last_camera_pose_), depth_data.data,
                                  depth_data.step, depth_data.rows, depth_data.cols);
  else
  {
    // the upload opens the side-stream section that KinFu::operator() continues (pre-processing) and ends: the copy
    // does not queue up behind the integrate and ray cast of the frame before
    kinfu_->beginFrame();
    depthDevice_.upload(depth_data.data, depth_data.step, depth_data.rows, depth_data.cols);
    result = kinfu_->operator()(to_kf(motion), to_kf(current_camera_in_volume), to_kf(last_camera_pose_), depthDevice_);
  }
  last_camera_pose_ = current_camera_in_volume;
  if (display_every_frame_)  // yak_server.cpp:36 does this unconditionally (render + download per frame)
    display();
  return result;
}

bool FusionServer::reset()
{
  if (sharded_)
    sharded_->resetVolume();
  else
    kinfu_->resetVolume();
  return true;
}
bool FusionServer::resetWithNewParams(const kfusion::KinFuParams& params)
{
  create(params);
  return true;
}

// yak_server.cpp:67-76
yak::PolygonMesh FusionServer::extractMesh(const yak::MarchingCubesParameters& params)
{
  if (sharded_)
    return yak::marchingCubesCPU(downloadTSDF(), params);  // (runs on the device: marching_cubes.cpp)
  const kfusion::cuda::TsdfVolume& vol = kinfu_->tsdf();
  const kfusion::Vec3i dims = vol.getDims();
  return yak::marchingCubesDevice(vol.data().ptr<void>(), dims[0], dims[1], dims[2], params);
}

yak::TSDFContainer FusionServer::downloadTSDF()
{
  if (sharded_)  // the slabs are contiguous z ranges: every rank copies its own into place
  {
    const kfusion::Vec3i d = sharded_->params().volume_dims;
    yak::TSDFContainer result(d[0], d[1], d[2], yak::TSDFContainer::Uninitialized());
    sharded_->downloadVolume(result.data());
    return result;
  }
  const kfusion::cuda::TsdfVolume& vol = kinfu_->tsdf();
  const kfusion::Vec3i dims = vol.getDims();
  yak::TSDFContainer result(dims[0], dims[1], dims[2], yak::TSDFContainer::Uninitialized());  // overwritten below
  vol.data().download(result.data());
  return result;
}

bool FusionServer::uploadTSDF(const yak::TSDFContainer& volume)
{
  if (sharded_)
  {
    const kfusion::Vec3i d = sharded_->params().volume_dims;
    const auto have = volume.dims();
    if (have.v[0] != d[0] || have.v[1] != d[1] || have.v[2] != d[2])
      return false;
    const std::size_t plane = (std::size_t)d[0] * (std::size_t)d[1];
    sharded_->forEachRank([&](int, kfusion::KinFu& kf) {  // every rank takes the layers it holds (own + halo)
      int z0, z1;
      kf.tsdf().heldLayers(z0, z1);
      kf.tsdf().data().upload(volume.data() + plane * (std::size_t)z0, plane * (std::size_t)(z1 - z0) * sizeof(std::uint32_t));
    });
    return true;
  }
  kfusion::cuda::TsdfVolume& vol = kinfu_->tsdf();
  const kfusion::Vec3i dims = vol.getDims();
  const auto have = volume.dims();
  if (have.v[0] != dims[0] || have.v[1] != dims[1] || have.v[2] != dims[2])
    return false;
  vol.data().upload(volume.data(), volume.size64() * sizeof(std::uint32_t));  // same size: the buffer is kept
  return true;
}

// yak_server.cpp:78-88
bool FusionServer::step(const Eigen::Affine3f& current_pose, const Eigen::Affine3f& last_pose, const cv::Mat& depth)
{
  Eigen::Affine3f s = current_pose * last_pose.inverse();
  s = s.inverse();
  if (sharded_)
    return sharded_->operator()(to_kf(s), to_kf(current_pose), to_kf(last_pose), depth.data, depth.step, depth.rows,
                                depth.cols);
  depthDevice_.upload(depth.data, depth.step, depth.rows, depth.cols);
  return kinfu_->operator()(to_kf(s), to_kf(current_pose), to_kf(last_pose), depthDevice_);
}

// yak_server.cpp:90-98 without the window
void FusionServer::downloadView()
{
  viewHost_.resize((std::size_t)viewDevice_.rows() * viewDevice_.cols() * 4);
  if (!viewHost_.empty())
    viewDevice_.download(viewHost_.data(), (std::size_t)viewDevice_.cols() * 4);
}
void FusionServer::display()  // yak_server.cpp:100-104
{
  if (sharded_)  // every rank holds the whole model maps: rank 0 renders them on its own thread
  {
    sharded_->forEachRank([&](int rank, kfusion::KinFu& kf) {
      if (rank != 0)
        return;
      kfusion::cuda::Image img;
      kf.renderImage(img, 3);
      viewRows_ = img.rows(), viewCols_ = img.cols();
      viewHost_.resize((std::size_t)img.rows() * img.cols() * 4);
      if (!viewHost_.empty())
        img.download(viewHost_.data(), (std::size_t)img.cols() * 4);
    });
    return;
  }
  kinfu_->renderImage(viewDevice_, 3);
  downloadView();
}
void FusionServer::display(const Eigen::Affine3f& pose)  // yak_server.cpp:106-110
{
  if (sharded_)
    throw std::runtime_error("FusionServer::display(pose): a fresh ray cast of a sharded volume is a collective of all "
                             "ranks; use display() (the tracker's current view) with YAK_HIP_DEVICES");
  kinfu_->renderImage(viewDevice_, to_kf(pose), 3);
  downloadView();
}
const std::vector<unsigned char>& FusionServer::view(int& rows, int& cols) const
{
  rows = sharded_ ? viewRows_ : viewDevice_.rows();
  cols = sharded_ ? viewCols_ : viewDevice_.cols();
  return viewHost_;
}
}  // namespace yak
