// yak/kfusion/kinfu_sharded.hpp -- the KinFu frame loop over a z-slab sharded volume on several GPUs of one node.
// NO COUNTERPART IN THE REFERENCE (one process, one GPU: SURVEY.md 2a); it is north_star's multi-GPU mode with the
// host code kept in C++: one rank per GPU, each rank a worker thread that owns its context, its slab of the
// volume (own layers + ray-cast halo) and an ordinary kfusion::KinFu running the reference's loop
// (yak/src/kfusion/kinfu.cpp:162-309) on it.  What the ranks exchange happens inside the C ABI on each rank's
// stream (include/yakhip.h: yakhip_raycast_sharded, yakhip_icp_estimate_sharded) -- RCCL over xGMI when every
// rank has a GPU of its own, the in-process transport when ranks share a GPU (tests on a one-GPU box).
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <yak/kfusion/kinfu.hpp>

namespace kfusion
{
class KF_EXPORTS ShardedKinFu
{
public:
  typedef std::shared_ptr<ShardedKinFu> Ptr;

  /** devices[r] = the HIP device of rank r.  The same device may be listed more than once (ranks then share it and
   *  talk through host memory).  Throws std::runtime_error if a rank cannot be set up. */
  ShardedKinFu(const KinFuParams& params, const std::vector<int>& devices);
  ~ShardedKinFu();

  /** The devices listed in the environment variable YAK_HIP_DEVICES ("0,1,2,3"); empty if it is not set. */
  static std::vector<int> devicesFromEnvironment();

  int world() const { return (int)ranks_.size(); }
  const KinFuParams& params() const { return params_; }

  /** KinFu::operator() on every rank with the same host depth frame (uint16 millimetres, `step` bytes per row): every
   *  rank uploads it to its own GPU, pre-processes it redundantly (image-space, 0.07 ms), tracks with all-reduced
   *  ICP sums, integrates the layers it holds and ray-casts its slab; returns what KinFu::operator() returns (the
   *  same on every rank: they hold the same pose bit for bit). */
  bool operator()(const Affine3f& inputCameraMotion, const Affine3f& currentCameraPose, const Affine3f& previousCameraPose,
                  const void* depth_host, size_t step, int rows, int cols);

  void resetVolume();
  void resetPose();
  Affine3f getCameraPose(int time = -1);

  /** The whole volume, x fastest, z slowest, slabs concatenated in rank order (what downloadTSDF returns on one GPU).
   *  `voxels_out` needs dims[0] * dims[1] * dims[2] 32-bit words. */
  void downloadVolume(void* voxels_out);
  /** The model maps of rank `rank` (vertex and normal image of the latest ray cast, rows * cols float4 each):
   *  identical on every rank. */
  void downloadModelMaps(int rank, void* points_out, void* normals_out);

  /** Runs fn(rank, kinfu) on every rank's thread (its context and shard are current there) and waits. */
  void forEachRank(const std::function<void(int, KinFu&)>& fn);

private:
  struct Rank;
  void worker(Rank* r);
  std::string waitAll(bool abort_peers);
  void shutdown();

  KinFuParams params_;
  std::vector<Rank*> ranks_;
};
}  // namespace kfusion
