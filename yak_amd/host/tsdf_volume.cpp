// kfusion::cuda::TsdfVolume host side (reference yak/src/kfusion/tsdf_volume.cpp:19-145) and the
// imgproc wrappers (reference yak/src/kfusion/imgproc.cpp:3-80).
#include <yak/kfusion/cuda/imgproc.hpp>
#include <yak/kfusion/cuda/tsdf_volume.hpp>

#include <yakhip.h>

#include <algorithm>
#include <ostream>
#include <stdexcept>

#define YAK_CHECK(call)                                                       \
  do                                                                          \
  {                                                                           \
    if ((call) != 0)                                                          \
      ::kfusion::cuda::error(yakhip_last_error(), __FILE__, __LINE__, #call); \
  } while (0)

namespace kfusion
{
// reference precomp.cpp:7-19
Intr::Intr() : fx(0), fy(0), cx(0), cy(0) {}
Intr::Intr(float fx_, float fy_, float cx_, float cy_) : fx(fx_), fy(fy_), cx(cx_), cy(cy_) {}
Intr Intr::operator()(int level_index) const
{
  const int div = 1 << level_index;
  return Intr(fx / div, fy / div, cx / div, cy / div);
}
std::ostream& operator<<(std::ostream& os, const Intr& intr)
{
  return os << "([f = " << intr.fx << ", " << intr.fy << "] [cp = " << intr.cx << ", " << intr.cy << "])";
}

namespace cuda
{
// reference tsdf_volume.cpp:9-14 throws "Not implemented" from both; they are implemented here
float TsdfVolume::Entry::half2float(half value)
{
  const unsigned sign = (value & 0x8000u) << 16, exp = (value >> 10) & 0x1fu;
  unsigned mant = value & 0x3ffu, bits;
  if (exp == 0)
  {
    if (mant == 0)
      bits = sign;
    else
    {
      int e = -1;
      do
      {
        ++e;
        mant <<= 1;
      } while (!(mant & 0x400u));
      bits = sign | ((unsigned)(112 - e) << 23) | ((mant & 0x3ffu) << 13);
    }
  }
  else if (exp == 31)
    bits = sign | 0x7f800000u | (mant << 13);
  else
    bits = sign | ((exp + 112u) << 23) | (mant << 13);
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}
TsdfVolume::Entry::half TsdfVolume::Entry::float2half(float value)
{
  unsigned u;
  std::memcpy(&u, &value, 4);
  const unsigned sign = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
  if (a >= 0x7f800000u)
    return (half)(sign | 0x7c00u | (a > 0x7f800000u ? 0x200u : 0u));
  if (a >= 0x477ff000u)
    return (half)(sign | 0x7c00u);
  if (a < 0x33000001u)
    return (half)sign;
  const int exp = (int)(a >> 23);
  unsigned mant = a & 0x7fffffu, h;
  if (exp >= 113)
  {
    h = ((unsigned)(exp - 112) << 10) | (mant >> 13);
    const unsigned rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u)))
      ++h;
  }
  else
  {
    mant |= 0x800000u;
    const int shift = 126 - exp;
    h = mant >> shift;
    const unsigned rem = mant & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (h & 1u)))
      ++h;
  }
  return (half)(sign | h);
}

// reference tsdf_volume.cpp:19-30: defaults trunc 0.03, max weight 128, 3 m cube, identity pose
TsdfVolume::TsdfVolume(const Vec3i& dims)
  : held_z0_(0)
  , held_z1_(0)
  , blocking_integrate_(true)
  , map_stale_(false)
  , data_()
  , trunc_dist_(0.03f)
  , max_weight_(128)
  , dims_(dims)
  , size_(Vec3f::all(3.f))
  , pose_(Affine3f::Identity())
  , gradient_delta_factor_(0.75f)
  , raycast_step_factor_(0.75f)
{
  create(dims_);
}
TsdfVolume::~TsdfVolume() {}

// :34-41 (the reference sizes the buffer with an int product; size_t here)
void TsdfVolume::create(const Vec3i& dims)
{
  dims_ = dims;
  held_z0_ = held_z1_ = 0;  // forces the allocation in ensureSlab()
  setTruncDist(trunc_dist_);
  clear();
}

// z-slab sharding: rank r of w owns layers [Z r / w, Z (r + 1) / w) and holds the ray-cast halo around them
// (yakhip_raycast_halo: it depends on the truncation distance, the voxel size and the two ray-cast factors, which
// the reference's setters may change after create(): every entry point passes through here first).
void TsdfVolume::ownedLayers(int& z0, int& z1) const
{
  const Shard& sh = currentShard();
  z0 = (int)((long long)dims_[2] * sh.rank / sh.world);
  z1 = (int)((long long)dims_[2] * (sh.rank + 1) / sh.world);
}
void TsdfVolume::heldLayers(int& z0, int& z1) const
{
  ownedLayers(z0, z1);
  if (currentShard().world == 1)
    return;
  yakhip_volume probe = {};
  probe.data = (void*)256;  // not looked at
  const Vec3f vsz = getVoxelSize();
  for (int i = 0; i < 3; ++i)
  {
    probe.dims[i] = dims_[i];
    probe.voxel_size[i] = vsz[i];
  }
  probe.z0 = 0, probe.z1 = dims_[2], probe.trunc_dist = trunc_dist_, probe.max_weight = max_weight_;
  int halo = 0;
  YAK_CHECK(yakhip_raycast_halo(&probe, raycast_step_factor_, gradient_delta_factor_, &halo));
  z0 = std::max(0, z0 - halo);
  z1 = std::min(dims_[2], z1 + halo);
}
// The held range follows the halo, and the halo follows trunc_dist, the voxel size and the two ray-cast factors: the
// reference's setters never touch the voxels (tsdf_volume.cpp:54-78), so a range that changes under a fused model
// keeps the model -- the layers old and new range share are carried over, only the layers that are new to the rank
// start cleared (a halo that grew has not seen the earlier frames: fuse before growing it, or re-fuse), and the
// surface map is recomputed from the voxels.  Returns true when the layers were allocated fresh (all cleared).
bool TsdfVolume::ensureSlab()
{
  int z0, z1;
  heldLayers(z0, z1);
  if (z0 == held_z0_ && z1 == held_z1_ && !data_.empty())
    return false;
  const bool carry = !data_.empty() && held_z1_ > held_z0_;  // (create() sets the held range empty: fresh)
  const int old0 = held_z0_, old1 = held_z1_;
  CudaData old;
  old.swap(data_);
  held_z0_ = z0, held_z1_ = z1;
  const size_t plane = (size_t)dims_[0] * (size_t)dims_[1] * sizeof(int);
  data_.create(plane * (size_t)(z1 - z0));
  yakhip_volume v = {};
  v.data = data_.ptr<void>();
  const Vec3f vsz = getVoxelSize();
  for (int i = 0; i < 3; ++i)
  {
    v.dims[i] = dims_[i];
    v.voxel_size[i] = vsz[i];
  }
  v.z0 = z0, v.z1 = z1, v.trunc_dist = trunc_dist_, v.max_weight = max_weight_;
  // the surface map of these layers (include/yakhip.h): integrate keeps it, the ray cast skips free space by it
  size_t map_bytes = 0;
  YAK_CHECK(yakhip_surface_map_bytes(&v, &map_bytes));
  surface_map_.create(map_bytes);
  v.surface_map = surface_map_.ptr<void>();
  YAK_CHECK(yakhip_clear_volume(context(), &v));
  map_stale_ = false;
  const int c0 = std::max(z0, old0), c1 = std::min(z1, old1);
  if (!carry || c1 <= c0)
    return true;
  YAK_CHECK(yakhip_copy(context(), data_.ptr<char>() + plane * (size_t)(c0 - z0), old.ptr<char>() + plane * (size_t)(c0 - old0),
                        plane * (size_t)(c1 - c0)));
  YAK_CHECK(yakhip_surface_map_rebuild(context(), &v));
  return false;
}
yakhip_volume TsdfVolume::view() const
{
  yakhip_volume v = {};
  v.data = const_cast<void*>(data_.ptr<void>());
  const Vec3f vsz = getVoxelSize();
  for (int i = 0; i < 3; ++i)
  {
    v.dims[i] = dims_[i];
    v.voxel_size[i] = vsz[i];
  }
  v.z0 = held_z0_, v.z1 = held_z1_, v.trunc_dist = trunc_dist_, v.max_weight = max_weight_;
  v.surface_map = const_cast<void*>(surface_map_.ptr<void>());
  return v;
}
// Someone may have written voxels behind the class's back (the non-const data() handle, swap): the map is
// recomputed from the voxels before the next ray cast consults it.
// Staleness is sticky for as long as somebody outside still holds the storage (CudaData copies alias, and wrapped
// user memory is the user's): such a handle can be written through at any time, also after a ray cast has rebuilt
// the map, so every consult rebuilds until the last outside handle is gone (the reference puts no constraint on
// what a caller does with data(), tsdf_volume.cpp:50-51).
void TsdfVolume::freshSurfaceMap()
{
  if (!map_stale_)
    return;
  const yakhip_volume v = view();
  YAK_CHECK(yakhip_surface_map_rebuild(context(), &v));
  map_stale_ = data_.shared();
}
const void* TsdfVolume::ownedData(size_t& voxels) const
{
  int o0, o1;
  ownedLayers(o0, o1);
  const size_t plane = (size_t)dims_[0] * (size_t)dims_[1];
  voxels = plane * (size_t)(o1 - o0);
  return data_.ptr<unsigned>() + plane * (size_t)(o0 - held_z0_);
}
bool TsdfVolume::raycastOverflowed()
{
  const Shard& sh = currentShard();
  if (sh.world == 1 || !sh.comm)
    return false;
  unsigned status[2] = { 0, 0 };
  YAK_CHECK(yakhip_comm_status(context(), sh.comm, status));
  return status[1] != 0;
}
Vec3i TsdfVolume::getDims() const { return dims_; }
Vec3f TsdfVolume::getVoxelSize() const  // :45-48
{
  return Vec3f(size_[0] / dims_[0], size_[1] / dims_[1], size_[2] / dims_[2]);
}
const CudaData TsdfVolume::data() const { return data_; }
CudaData TsdfVolume::data()
{
  map_stale_ = true;  // a handle the caller can write through
  return data_;
}
Vec3f TsdfVolume::getSize() const { return size_; }
void TsdfVolume::setSize(const Vec3f& size)  // :54-58
{
  size_ = size;
  setTruncDist(trunc_dist_);
}
float TsdfVolume::getTruncDist() const { return trunc_dist_; }
void TsdfVolume::setTruncDist(float distance)  // :62-67
{
  const Vec3f vsz = getVoxelSize();
  const float max_coeff = std::max<float>(std::max<float>(vsz[0], vsz[1]), vsz[2]);
  trunc_dist_ = std::max(distance, 2.1f * max_coeff);
}
int TsdfVolume::getMaxWeight() const { return max_weight_; }
void TsdfVolume::setMaxWeight(int weight)
{
  // the weight lane is a signed 16-bit value on the reference's device path (device.hpp:69-77)
  max_weight_ = std::max(1, std::min(weight, 32767));
}
Affine3f TsdfVolume::getPose() const { return pose_; }
void TsdfVolume::setPose(const Affine3f& pose) { pose_ = pose; }
float TsdfVolume::getRaycastStepFactor() const { return raycast_step_factor_; }
void TsdfVolume::setRaycastStepFactor(float factor) { raycast_step_factor_ = factor; }
float TsdfVolume::getGradientDeltaFactor() const { return gradient_delta_factor_; }
void TsdfVolume::setGradientDeltaFactor(float factor) { gradient_delta_factor_ = factor; }
void TsdfVolume::swap(CudaData& data)
{
  data_.swap(data);
  map_stale_ = true;
}
void TsdfVolume::applyAffine(const Affine3f& affine) { pose_ = affine * pose_; }
void TsdfVolume::setBlockingIntegrate(bool blocking) { blocking_integrate_ = blocking; }

void TsdfVolume::clear()  // :80-87
{
  if (ensureSlab())
    return;  // freshly allocated layers are cleared
  const yakhip_volume v = view();
  YAK_CHECK(yakhip_clear_volume(context(), &v));
  map_stale_ = map_stale_ && data_.shared();  // a writable handle that is still out there can be written through again
}

// :89-108.  vol2cam = camera_pose^-1 * pose_; the device layer takes it as row-major R and t
// (precomp.hpp:15-30).  Blocking, like the reference (tsdf_volume.cu:139).
void TsdfVolume::integrate(const Dists& dists, const Affine3f& camera_motion_tform, const Intr& intr)
{
  const Affine3f vol2cam = camera_motion_tform.inv() * pose_;
  const Mat3f R = vol2cam.rotation();
  const Vec3f t = vol2cam.translation();
  const float Rf[9] = { R.val[0], R.val[1], R.val[2], R.val[3], R.val[4], R.val[5], R.val[6], R.val[7], R.val[8] };
  const float tf[3] = { t[0], t[1], t[2] };
  const float in[4] = { intr.fx, intr.fy, intr.cx, intr.cy };
  ensureSlab();
  // a rank integrates every layer it holds: its own and, redundantly, the halo (nothing is exchanged)
  const yakhip_volume v = view();
  YAK_CHECK(yakhip_integrate(context(), &v, dists.ptr(), dists.step(), dists.rows(), dists.cols(), Rf, tf, in));
  if (blocking_integrate_)
    YAK_CHECK(yakhip_ctx_synchronize(context()));
}

void TsdfVolume::prepareIntegrate(const Dists& dists)
{
  ensureSlab();
  const yakhip_volume v = view();
  YAK_CHECK(yakhip_integrate_prepare(context(), &v, dists.ptr(), dists.step(), dists.rows(), dists.cols()));
}

namespace
{
// :114-119 / :133-138: cam2vol = pose_^-1 * camera_pose, Rinv = its rotation inverted (DECOMP_SVD in
// the reference; Matx33f::inv here -- the same matrix up to rounding, parity unpinned: DESIGN.md)
struct RaycastPose
{
  float R[9], t[3], Rinv[9];
};
RaycastPose raycast_pose(const Affine3f& pose, const Affine3f& camera_pose)
{
  const Affine3f cam2vol = pose.inv() * camera_pose;
  const Mat3f R = cam2vol.rotation(), Rinv = cam2vol.rotation().inv(cv::DECOMP_SVD);
  const Vec3f t = cam2vol.translation();
  RaycastPose o;
  for (int i = 0; i < 9; ++i)
  {
    o.R[i] = R.val[i];
    o.Rinv[i] = Rinv.val[i];
  }
  for (int i = 0; i < 3; ++i)
    o.t[i] = t[i];
  return o;
}
}  // namespace

// :110-126
void TsdfVolume::raycast(const Affine3f& camera_pose, const Intr& intr, Depth& depth, Normals& normals)
{
  const RaycastPose rp = raycast_pose(pose_, camera_pose);
  const float in[4] = { intr.fx, intr.fy, intr.cx, intr.cy };
  ensureSlab();
  freshSurfaceMap();
  const yakhip_volume v = view();
  const Shard& sh = currentShard();
  if (sh.world > 1)
  {
    int o0, o1;
    ownedLayers(o0, o1);
    keys_.create(depth.rows(), depth.cols());
    YAK_CHECK(yakhip_raycast_sharded(context(), sh.comm, &v, o0, o1, rp.R, rp.t, rp.Rinv, in, raycast_step_factor_,
                                     gradient_delta_factor_, nullptr, 0, depth.ptr(), depth.step(), normals.ptr(),
                                     normals.step(), keys_.ptr(), keys_.step(), depth.rows(), depth.cols()));
    return;
  }
  YAK_CHECK(yakhip_raycast_depth(context(), &v, rp.R, rp.t, rp.Rinv, in, raycast_step_factor_, gradient_delta_factor_,
                                 depth.ptr(), depth.step(), normals.ptr(), normals.step(), depth.rows(), depth.cols()));
}
// :128-145
void TsdfVolume::raycast(const Affine3f& camera_pose, const Intr& intr, Cloud& points, Normals& normals)
{
  const RaycastPose rp = raycast_pose(pose_, camera_pose);
  const float in[4] = { intr.fx, intr.fy, intr.cx, intr.cy };
  ensureSlab();
  freshSurfaceMap();
  const yakhip_volume v = view();
  const Shard& sh = currentShard();
  if (sh.world > 1)
  {
    // slab march, composite over the ranks, resolve, composite of the settled pixels: one stream-ordered call
    // after which every rank holds the whole-volume maps, bit for bit (include/yakhip.h)
    int o0, o1;
    ownedLayers(o0, o1);
    keys_.create(points.rows(), points.cols());
    YAK_CHECK(yakhip_raycast_sharded(context(), sh.comm, &v, o0, o1, rp.R, rp.t, rp.Rinv, in, raycast_step_factor_,
                                     gradient_delta_factor_, points.ptr(), points.step(), nullptr, 0, normals.ptr(),
                                     normals.step(), keys_.ptr(), keys_.step(), points.rows(), points.cols()));
    return;
  }
  YAK_CHECK(yakhip_raycast_points(context(), &v, rp.R, rp.t, rp.Rinv, in, raycast_step_factor_, gradient_delta_factor_,
                                  points.ptr(), points.step(), normals.ptr(), normals.step(), points.rows(),
                                  points.cols()));
}
void TsdfVolume::raycastPyramid(const Affine3f& camera_pose, const Intr& intr, std::vector<Cloud>& points_pyr,
                                std::vector<Normals>& normals_pyr, int levels)
{
  if (levels < 1 || (int)points_pyr.size() < levels || (int)normals_pyr.size() < levels)
    ::kfusion::cuda::error("raycastPyramid: the pyramids hold fewer entries than `levels`", __FILE__, __LINE__, "raycastPyramid");
  Cloud& points = points_pyr[0];
  Normals& normals = normals_pyr[0];
  int done = 1;
  if (currentShard().world > 1 || levels < 2 || points.rows() < 4 || points.cols() < 4)
    raycast(camera_pose, intr, points, normals);
  else
  {
    const RaycastPose rp = raycast_pose(pose_, camera_pose);
    const float in[4] = { intr.fx, intr.fy, intr.cx, intr.cy };
    ensureSlab();
    freshSurfaceMap();
    const yakhip_volume v = view();
    done = levels >= 3 ? 3 : 2;
    for (int i = 1; i < done; ++i)
    {
      points_pyr[i].create(points_pyr[i - 1].rows() / 2, points_pyr[i - 1].cols() / 2);
      normals_pyr[i].create(points_pyr[i - 1].rows() / 2, points_pyr[i - 1].cols() / 2);
    }
    YAK_CHECK(yakhip_raycast_points_pyramid(context(), &v, rp.R, rp.t, rp.Rinv, in, raycast_step_factor_, gradient_delta_factor_,
                                            points.ptr(), points.step(), normals.ptr(), normals.step(), points.rows(),
                                            points.cols(), points_pyr[1].ptr(), points_pyr[1].step(), normals_pyr[1].ptr(),
                                            normals_pyr[1].step(), done == 3 ? points_pyr[2].ptr() : nullptr,
                                            done == 3 ? points_pyr[2].step() : 0, done == 3 ? normals_pyr[2].ptr() : nullptr,
                                            done == 3 ? normals_pyr[2].step() : 0));
  }
  int i = done;
  for (; i + 1 < levels; i += 2)  // two levels per launch
    resizePointsNormals2(points_pyr[i - 1], normals_pyr[i - 1], points_pyr[i], normals_pyr[i], points_pyr[i + 1], normals_pyr[i + 1]);
  if (i < levels)
    resizePointsNormals(points_pyr[i - 1], normals_pyr[i - 1], points_pyr[i], normals_pyr[i]);
}

// :147-168.  The volume pose is the affine the points are mapped by; the returned array is a view of the
// first min(n, capacity) elements of cloud_buffer (DeviceArray(T*, size) does not own).
DeviceArray<Point> TsdfVolume::fetchCloud(DeviceArray<Point>& cloud_buffer) const
{
  enum
  {
    DEFAULT_CLOUD_BUFFER_SIZE = 10 * 1000 * 1000
  };
  if (cloud_buffer.empty())
    cloud_buffer.create(DEFAULT_CLOUD_BUFFER_SIZE);

  const Mat3f R = pose_.rotation();
  const Vec3f t = pose_.translation();
  yakhip_volume v = view();
  if (map_stale_)
    v.surface_map = nullptr;  // (const: the map is not recomputed here, the extraction reads every voxel instead)
  unsigned long long n = 0;
  YAK_CHECK(yakhip_extract_cloud(context(), &v, R.val, t.val, cloud_buffer.ptr(), cloud_buffer.size(), &n));
  const size_t size = n < cloud_buffer.size() ? (size_t)n : cloud_buffer.size();
  return DeviceArray<Point>(cloud_buffer.ptr(), size);
}
// :170-183
void TsdfVolume::fetchNormals(const DeviceArray<Point>& cloud, DeviceArray<Normal>& normals) const
{
  normals.create(cloud.size());
  if (cloud.empty())
    return;
  const Mat3f R = pose_.rotation(), Rinv = pose_.rotation().inv(cv::DECOMP_SVD);
  const Vec3f t = pose_.translation();
  yakhip_volume v = view();
  v.z0 = 0, v.z1 = dims_[2], v.surface_map = nullptr;  // (whole volume only, as before)
  YAK_CHECK(yakhip_extract_normals(context(), &v, cloud.ptr(), cloud.size(), R.val, t.val, Rinv.val,
                                   gradient_delta_factor_, normals.ptr()));
}

// ---- imgproc wrappers ----
void computeDists(const Depth& depth, Dists& dists, const Intr& intr)  // imgproc.cpp:45-49
{
  dists.create(depth.rows(), depth.cols());
  const float in[4] = { intr.fx, intr.fy, intr.cx, intr.cy };
  YAK_CHECK(yakhip_compute_dists(context(), depth.ptr(), depth.step(), dists.ptr(), dists.step(), depth.rows(),
                                 depth.cols(), in));
}
void depthBilateralFilter(const Depth& in, Depth& out, int ksz, float sigma_spatial, float sigma_depth)  // imgproc.cpp:3-11
{
  out.create(in.rows(), in.cols());
  YAK_CHECK(yakhip_bilateral_filter(context(), in.ptr(), in.step(), out.ptr(), out.step(), in.rows(), in.cols(), ksz,
                                    sigma_spatial, sigma_depth));
}
void depthTruncation(Depth& depth, float threshold)  // imgproc.cpp:13
{
  YAK_CHECK(yakhip_truncate_depth(context(), depth.ptr(), depth.step(), depth.rows(), depth.cols(), threshold));
}
void depthBuildPyramid(const Depth& depth, Depth& pyramid, float sigma_depth)  // imgproc.cpp:15-19
{
  pyramid.create(depth.rows() / 2, depth.cols() / 2);
  YAK_CHECK(yakhip_depth_pyramid(context(), depth.ptr(), depth.step(), depth.rows(), depth.cols(), pyramid.ptr(),
                                 pyramid.step(), sigma_depth));
}
void computePointNormals(const Intr& intr, const Depth& depth, Cloud& points, Normals& normals)  // imgproc.cpp:33-43
{
  points.create(depth.rows(), depth.cols());
  normals.create(depth.rows(), depth.cols());
  const float in[4] = { intr.fx, intr.fy, intr.cx, intr.cy };
  YAK_CHECK(yakhip_compute_points_normals(context(), in, depth.ptr(), depth.step(), depth.rows(), depth.cols(),
                                          points.ptr(), points.step(), normals.ptr(), normals.step()));
}
void resizePointsNormals(const Cloud& points, const Normals& normals, Cloud& points_out, Normals& normals_out)
{  // imgproc.cpp:65-80
  points_out.create(points.rows() / 2, points.cols() / 2);
  normals_out.create(normals.rows() / 2, normals.cols() / 2);
  YAK_CHECK(yakhip_resize_points_normals(context(), points.ptr(), points.step(), normals.ptr(), normals.step(),
                                         points.rows(), points.cols(), points_out.ptr(), points_out.step(),
                                         normals_out.ptr(), normals_out.step()));
}
void resizePointsNormals2(const Cloud& points, const Normals& normals, Cloud& points_half, Normals& normals_half,
                          Cloud& points_quarter, Normals& normals_quarter)
{
  points_half.create(points.rows() / 2, points.cols() / 2);
  normals_half.create(normals.rows() / 2, normals.cols() / 2);
  points_quarter.create(points.rows() / 2 / 2, points.cols() / 2 / 2);
  normals_quarter.create(normals.rows() / 2 / 2, normals.cols() / 2 / 2);
  YAK_CHECK(yakhip_resize_points_normals2(context(), points.ptr(), points.step(), normals.ptr(), normals.step(), points.rows(),
                                          points.cols(), points_half.ptr(), points_half.step(), normals_half.ptr(),
                                          normals_half.step(), points_quarter.ptr(), points_quarter.step(),
                                          normals_quarter.ptr(), normals_quarter.step()));
}
void waitAllDefaultStream() { YAK_CHECK(yakhip_ctx_synchronize(context())); }  // imgproc.cpp:21

void renderImage(const Depth& depth, const Normals& normals, const Intr& intr, const Vec3f& light_pose, Image& image)
{  // imgproc.cpp:82-98
  image.create(depth.rows(), depth.cols());
  const float in[4] = { intr.fx, intr.fy, intr.cx, intr.fy };  // :92 passes fy where cy belongs; kept
  YAK_CHECK(yakhip_render_image_depth(context(), depth.ptr(), depth.step(), normals.ptr(), normals.step(), depth.rows(),
                                      depth.cols(), in, light_pose.val, image.ptr(), image.step()));
  waitAllDefaultStream();
}
void renderImage(const Cloud& points, const Normals& normals, const Intr& /*intr*/, const Vec3f& light_pose, Image& image)
{  // imgproc.cpp:100-116 (the points variant never uses its Reprojector, imgproc.cu:453-504)
  image.create(points.rows(), points.cols());
  YAK_CHECK(yakhip_render_image_points(context(), points.ptr(), points.step(), normals.ptr(), normals.step(),
                                       points.rows(), points.cols(), light_pose.val, image.ptr(), image.step()));
  waitAllDefaultStream();
}
void renderTangentColors(const Normals& normals, Image& image)  // imgproc.cpp:118-126
{
  image.create(normals.rows(), normals.cols());
  YAK_CHECK(yakhip_render_tangent_colors(context(), normals.ptr(), normals.step(), normals.rows(), normals.cols(),
                                         image.ptr(), image.step()));
  waitAllDefaultStream();
}
}  // namespace cuda
}  // namespace kfusion
