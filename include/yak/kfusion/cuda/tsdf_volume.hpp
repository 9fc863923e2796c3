// yak/kfusion/cuda/tsdf_volume.hpp -- host view of the TSDF volume: owns the voxel buffer, keeps the
// metric parameters, turns poses into the 12 floats the device layer takes.  Same interface as the
// reference class (yak/include/yak/kfusion/cuda/tsdf_volume.hpp:10-78; behaviour cited per method
// in yak_amd/host/tsdf_volume.cpp).
#pragma once
#include <vector>
#include <yak/kfusion/types.hpp>

struct yakhip_volume;

namespace kfusion
{
namespace cuda
{
class KF_EXPORTS TsdfVolume
{
public:
  explicit TsdfVolume(const Vec3i& dims);
  virtual ~TsdfVolume();

  void create(const Vec3i& dims);

  Vec3i getDims() const;
  Vec3f getVoxelSize() const;

  const CudaData data() const;
  CudaData data();

  Vec3f getSize() const;
  void setSize(const Vec3f& size);

  float getTruncDist() const;
  void setTruncDist(float distance);

  int getMaxWeight() const;
  void setMaxWeight(int weight);

  Affine3f getPose() const;
  void setPose(const Affine3f& pose);

  float getRaycastStepFactor() const;
  void setRaycastStepFactor(float factor);

  float getGradientDeltaFactor() const;
  void setGradientDeltaFactor(float factor);

  virtual void clear();
  virtual void applyAffine(const Affine3f& affine);
  virtual void integrate(const Dists& dists, const Affine3f& camera_motion_tform, const Intr& intr);
  virtual void raycast(const Affine3f& camera_pose, const Intr& intr, Depth& depth, Normals& normals);
  virtual void raycast(const Affine3f& camera_pose, const Intr& intr, Cloud& points, Normals& normals);

  /** raycast(points, normals) into level 0 of the two pyramids and their coarser levels (resizePointsNormals level by
   *  level, kinfu.cpp:294-306) -- on one GPU the first two of them come out of the ray cast's own launch
   *  (include/yakhip.h, yakhip_raycast_points_pyramid).  Same maps either way.  No counterpart in the reference. */
  void raycastPyramid(const Affine3f& camera_pose, const Intr& intr, std::vector<Cloud>& points_pyr,
                      std::vector<Normals>& normals_pyr, int levels);

  void swap(CudaData& data);

  /** integrate() waits for the device before it returns, as the reference's does (tsdf_volume.cu:139); a caller that
   *  keeps the ray-distance image alive until the work is done (KinFu does) may switch the wait off.  No counterpart
   *  in the reference. */
  void setBlockingIntegrate(bool blocking);
  /** The part of integrate()'s work on `dists` that needs no pose, ahead of time and on the stream that is current
   *  (include/yakhip.h, yakhip_integrate_prepare): the integrate() of the same image that follows launches the voxel
   *  walk only.  Optional; the voxels come out the same.  No counterpart in the reference. */
  void prepareIntegrate(const Dists& dists);

  // ---- z-slab sharding (no counterpart in the reference; world == 1 leaves everything above as it is) ----
  /** Layers [z0, z1) of the volume this object OWNS on the calling thread's rank (cuda::currentShard()). */
  void ownedLayers(int& z0, int& z1) const;
  /** Layers [z0, z1) it HOLDS: the owned ones plus the ray-cast halo, integrated redundantly.  data() is this range. */
  void heldLayers(int& z0, int& z1) const;
  /** Device pointer of the first OWNED layer and the number of owned voxels (a sharded downloadTSDF concatenates these). */
  const void* ownedData(size_t& voxels) const;
  /** True if the last sharded raycast() had more pixels to settle than its list holds (include/yakhip.h,
   *  yakhip_comm_status); KinFu then repeats that ray cast with room for all.  BLOCKING. */
  bool raycastOverflowed();

  DeviceArray<Point> fetchCloud(DeviceArray<Point>& cloud_buffer) const;
  void fetchNormals(const DeviceArray<Point>& cloud, DeviceArray<Normal>& normals) const;

  struct Entry
  {
    typedef unsigned short half;
    half tsdf;
    unsigned short weight;
    static float half2float(half value);
    static half float2half(float value);
  };

private:
  bool ensureSlab();  // (re)allocates the held layers when the sharding or the halo-relevant parameters changed; a fused model is carried over
  yakhip_volume view() const;  // the held layers as the C ABI takes them
  void freshSurfaceMap();      // recompute the surface map if voxels may have been written behind the class's back

  int held_z0_, held_z1_;          // layers in data_ (0, dims[2] on one GPU)
  bool blocking_integrate_;
  bool map_stale_;                 // data() / swap() handed the voxels out and a handle may still be alive: consults rebuild the map
  CudaData surface_map_;           // one bit per 16^3 brick of the held layers (include/yakhip.h: yakhip_volume.surface_map)
  DeviceArray2D<int> keys_;        // sharded ray cast: the march-ending pair of every pixel
  CudaData data_;
  float trunc_dist_;
  int max_weight_;
  Vec3i dims_;
  Vec3f size_;
  Affine3f pose_;
  float gradient_delta_factor_;
  float raycast_step_factor_;
};
}  // namespace cuda
}  // namespace kfusion
