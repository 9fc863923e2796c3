// yak/kfusion/kinfu.hpp -- KinFuParams and the per-frame fusion loop
// (reference yak/include/yak/kfusion/kinfu.hpp:13-141; loop: yak/src/kfusion/kinfu.cpp:162-309).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include <yak/kfusion/cuda/projective_icp.hpp>
#include <yak/kfusion/cuda/tsdf_volume.hpp>
#include <yak/kfusion/types.hpp>

namespace kfusion
{
namespace cuda
{
// Names kept from the reference (kinfu.hpp:17-22); they report HIP devices.
KF_EXPORTS int getCudaEnabledDeviceCount();
KF_EXPORTS void setDevice(int device);
KF_EXPORTS std::string getDeviceName(int device);
KF_EXPORTS bool checkIfPreFermiGPU(int device);
KF_EXPORTS void printCudaDeviceInfo(int device);
KF_EXPORTS void printShortCudaDeviceInfo(int device);
}  // namespace cuda

/** Field-for-field the reference's parameter block (kinfu.hpp:25-86); defaults: kinfu.cpp:10-48. */
struct KF_EXPORTS KinFuParams
{
  static KinFuParams default_params();

  int cols;  // pixels
  int rows;  // pixels
  Intr intr;  // fx, fy, cx, cy

  Vec3i volume_dims;        // voxels
  float volume_resolution;  // metres per voxel
  Affine3f volume_pose;

  float bilateral_sigma_depth;    // metres
  float bilateral_sigma_spatial;  // pixels
  int bilateral_kernel_size;      // pixels

  float icp_truncate_depth_dist;  // metres, 0 = off
  float icp_dist_thres;           // metres
  float icp_angle_thres;          // radians
  std::vector<int> icp_iter_num;  // per pyramid level 0..3

  float tsdf_min_camera_movement;  // metres, 0 = integrate every frame
  float tsdf_trunc_dist;           // metres
  int tsdf_max_weight;             // frames

  float raycast_step_factor;    // x trunc_dist (kinfu.cpp:39 says "voxel sizes"; the code multiplies trunc_dist)
  float gradient_delta_factor;  // x voxel size
  Vec3f light_pose;             // metres

  bool use_pose_hints;            // take the externally supplied motion as the ICP start / the motion itself
  bool use_icp;                   // refine the pose against the raycast model
  bool update_via_sensor_motion;  // poses_ <- poses_.back() * motion  (else: <- supplied pose)
};

class KF_EXPORTS KinFu
{
public:
  typedef std::shared_ptr<KinFu> Ptr;

  explicit KinFu(const KinFuParams& params);

  const KinFuParams& params() const;
  KinFuParams& params();

  const cuda::TsdfVolume& tsdf() const;
  cuda::TsdfVolume& tsdf();

  const cuda::ProjectiveICP& icp() const;
  cuda::ProjectiveICP& icp();

  void resetPose();
  void resetVolume();

  /** One frame (reference kinfu.cpp:162-309).  Returns false on frame 0 (it integrates but cannot
   *  track yet, kinfu.cpp:200) and on ICP failure (which also resets volume and pose, :247-252). */
  bool operator()(const Affine3f& inputCameraMotion, const Affine3f& currentCameraPose,
                  const Affine3f& previousCameraPose, const cuda::Depth& depth,
                  const cuda::Image& image = cuda::Image());

  /** The model as the tracker last saw it (the ray cast of the latest pose): flags 1 = Phong view, 2 = normal
   *  colours, 3 = both side by side (image is then 2 * cols wide); kinfu.hpp:113, kinfu.cpp:311-335. */
  void renderImage(cuda::Image& image, int flags = 0);
  /** The same views of the volume ray-cast from `pose` (kinfu.hpp:114, kinfu.cpp:337-366). */
  void renderImage(cuda::Image& image, const Affine3f& pose, int flags = 0);

  Affine3f getCameraPose(int time = -1) const;

  /** The zero crossings of the volume as points in the frame the volume pose maps to (kinfu.hpp:116,
   *  kinfu.cpp:368-377; at most 10 M points, tsdf_volume.cpp:149-155). */
  std::vector<Point> downloadCloud() const;

  /** The model maps the next frame is tracked against (level 0 of the previous-frame pyramid: the latest ray cast),
   *  rows * cols float4 each, densely packed.  No counterpart in the reference (its maps never leave the device);
   *  used by the multi-GPU tests to compare ranks. */
  void downloadModelMaps(void* points_out, void* normals_out) const;

  /** True if operator() prepares a frame on the context's second stream (include/yakhip.h, yakhip_ctx_side_begin):
   *  a caller that uploads the frame itself does so after yakhip_ctx_side_begin, and operator() takes it from there. */
  bool overlapping() const { return overlap_; }
  /** Opens that section (nothing if not overlapping): call it before uploading the frame operator() is about to get,
   *  so that the copy does not queue up behind the integrate and ray cast of the frame before. */
  void beginFrame();

private:
  void allocate_buffers();

  int frame_counter_;
  KinFuParams params_;
  std::vector<Affine3f> poses_;

  cuda::Dists dists_, dists_alt_;  // in turn: frame f + 1's is written while frame f's is still being integrated
  bool overlap_;                   // pre-processing on the context's second stream, no wait at the end of a frame
  cuda::Frame curr_, prev_;
  cuda::Cloud points_;
  cuda::Normals normals_;
  cuda::Depth depths_;

  std::shared_ptr<cuda::TsdfVolume> volume_;
  std::shared_ptr<cuda::ProjectiveICP> icp_;
};
}  // namespace kfusion
