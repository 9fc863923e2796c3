// kfusion::ShardedKinFu -- one worker thread per rank, each running kfusion::KinFu on its z-slab
// (include/yak/kfusion/kinfu_sharded.hpp).  No counterpart in the reference.
#include <yak/kfusion/kinfu_sharded.hpp>

#include <yakhip.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <set>
#include <sstream>
#include <stdexcept>

namespace kfusion
{
struct ShardedKinFu::Rank
{
  int rank = 0, device = 0;
  yakhip_ctx* ctx = nullptr;
  yakhip_comm* comm = nullptr;
  std::unique_ptr<KinFu> kinfu;
  cuda::Depth depth;
  std::thread thread;
  std::mutex m;
  std::condition_variable cv;
  std::function<void()> job;  // set by the caller, run by the worker
  bool has_job = false, done = false, quit = false;
  std::string error;
};

std::vector<int> ShardedKinFu::devicesFromEnvironment()
{
  std::vector<int> out;
  const char* env = std::getenv("YAK_HIP_DEVICES");
  if (!env)
    return out;
  std::stringstream ss(env);
  std::string item;
  while (std::getline(ss, item, ','))
    if (!item.empty())
      out.push_back(std::atoi(item.c_str()));
  return out;
}

void ShardedKinFu::worker(Rank* r)
{
  for (;;)
  {
    std::function<void()> job;
    {
      std::unique_lock<std::mutex> lock(r->m);
      r->cv.wait(lock, [&] { return r->has_job || r->quit; });
      if (r->quit && !r->has_job)
        return;
      job = r->job;
    }
    std::string err;
    try
    {
      job();
    }
    catch (const std::exception& e)
    {
      err = e.what();
      // this rank cannot go on: peers that wait for it in an exchange must not wait for ever (local transport: they
      // wake with YAKHIP_EABORTED; RCCL: this communicator's own collectives end, waitAll() releases the peers')
      if (r->comm)
        yakhip_comm_abort(r->comm);
    }
    {
      std::lock_guard<std::mutex> lock(r->m);
      r->error = err;
      r->has_job = false;
      r->done = true;
    }
    r->cv.notify_all();
  }
}

void ShardedKinFu::forEachRank(const std::function<void(int, KinFu&)>& fn)
{
  for (Rank* r : ranks_)
  {
    std::lock_guard<std::mutex> lock(r->m);
    r->job = [fn, r] { fn(r->rank, *r->kinfu); };
    r->has_job = true;
    r->done = false;
    r->cv.notify_all();
  }
  const std::string first_error = waitAll(true);
  if (!first_error.empty())
    throw std::runtime_error(first_error);
}

// Waits for every rank's job.  Once one rank has failed, the ranks still at work may be blocked in a collective that
// the failed rank will never join: with abort_peers their communicators are aborted from here (ncclCommAbort may be
// called from another thread; the local transport has been woken by the failed rank's own abort already), so that
// they come back with an error instead of hanging.  Returns the first error ("" if none).  After a failure the
// communicators are dead and every later frame fails with YAKHIP_EABORTED: build a new ShardedKinFu.
std::string ShardedKinFu::waitAll(bool abort_peers)
{
  std::string first_error, first_echo;  // (echo: a peer's "the communicator was aborted", the consequence, not the cause)
  std::vector<bool> finished(ranks_.size(), false), aborted(ranks_.size(), false);
  size_t left = ranks_.size();
  bool failed = false;
  while (left > 0)
  {
    for (size_t i = 0; i < ranks_.size(); ++i)
    {
      if (finished[i])
        continue;
      Rank* r = ranks_[i];
      std::unique_lock<std::mutex> lock(r->m);
      // block on the first unfinished rank while all is well, poll once somebody has failed
      const bool got = r->cv.wait_for(lock, std::chrono::milliseconds(failed ? 1 : 20), [&] { return r->done; });
      if (got)
      {
        finished[i] = true;
        --left;
        if (!r->error.empty())
        {
          failed = true;
          const bool echo = r->error.find("left the group") != std::string::npos || r->error.find("was aborted") != std::string::npos;
          std::string& slot = echo ? first_echo : first_error;
          if (slot.empty())
            slot = "rank " + std::to_string(r->rank) + ": " + r->error;
        }
      }
      else if (failed && abort_peers && !aborted[i] && r->comm)
      {
        aborted[i] = true;
        lock.unlock();
        yakhip_comm_abort(r->comm);
      }
    }
  }
  return first_error.empty() ? first_echo : first_error;
}

ShardedKinFu::ShardedKinFu(const KinFuParams& params, const std::vector<int>& devices) : params_(params)
{
  const int world = (int)devices.size();
  if (world < 1)
    throw std::invalid_argument("ShardedKinFu: needs at least one device");
  // RCCL wants one rank per GPU; ranks that share a GPU (or YAK_HIP_COMM=local) talk through host memory
  const char* env = std::getenv("YAK_HIP_COMM");
  const bool distinct = std::set<int>(devices.begin(), devices.end()).size() == devices.size();
  const bool use_rccl = env ? std::string(env) == "rccl" : distinct;
  std::vector<yakhip_comm*> local((size_t)world, nullptr);
  char unique_id[YAKHIP_UNIQUE_ID_BYTES];
  if (use_rccl)
  {
    if (yakhip_comm_unique_id(unique_id, sizeof(unique_id)) != 0)
      throw std::runtime_error(std::string("ShardedKinFu: ") + yakhip_last_error());
  }
  else if (yakhip_comm_create_local(world, local.data()) != 0)
    throw std::runtime_error(std::string("ShardedKinFu: ") + yakhip_last_error());

  for (int i = 0; i < world; ++i)
  {
    Rank* r = new Rank();
    r->rank = i;
    r->device = devices[(size_t)i];
    r->comm = local[(size_t)i];
    ranks_.push_back(r);
    r->thread = std::thread([this, r] { worker(r); });
  }
  // set-up on every rank's own thread, concurrently (ncclCommInitRank returns when all ranks have joined)
  std::vector<std::string> errors((size_t)world);
  for (Rank* r : ranks_)
  {
    std::lock_guard<std::mutex> lock(r->m);
    r->job = [this, r, world, use_rccl, &unique_id] {
      if (yakhip_ctx_create(&r->ctx, r->device) != 0)
        throw std::runtime_error(yakhip_last_error());
      cuda::setDevice(r->device);
      cuda::setContext(r->ctx);
      if (use_rccl && yakhip_comm_create_rccl(r->ctx, &r->comm, unique_id, world, r->rank) != 0)
        throw std::runtime_error(yakhip_last_error());
      cuda::Shard sh = { r->comm, r->rank, world };
      cuda::setShard(sh);
      r->kinfu.reset(new KinFu(params_));
    };
    r->has_job = true;
    r->done = false;
    r->cv.notify_all();
  }
  // (with RCCL the communicators come into being inside these jobs: a rank that fails before ncclCommInitRank leaves
  // its peers inside that call, which no abort reaches -- set-up errors there are device errors that hit every rank)
  const std::string first_error = waitAll(!use_rccl);
  if (!first_error.empty())
  {
    shutdown();
    throw std::runtime_error("ShardedKinFu: " + first_error);
  }
}

ShardedKinFu::~ShardedKinFu() { shutdown(); }

void ShardedKinFu::shutdown()
{
  for (Rank* r : ranks_)
  {
    {
      std::lock_guard<std::mutex> lock(r->m);
      // the rank's objects live and die on its own thread (its context is current there)
      r->job = [r] {
        r->kinfu.reset();
        r->depth.release();
        if (r->comm)
          yakhip_comm_destroy(r->comm);
        r->comm = nullptr;
        if (r->ctx)
          yakhip_ctx_destroy(r->ctx);
        r->ctx = nullptr;
        cuda::setContext(nullptr);
      };
      r->has_job = true;
      r->done = false;
      r->quit = true;
    }
    r->cv.notify_all();
  }
  for (Rank* r : ranks_)
  {
    if (r->thread.joinable())
      r->thread.join();
    delete r;
  }
  ranks_.clear();
}

bool ShardedKinFu::operator()(const Affine3f& inputCameraMotion, const Affine3f& currentCameraPose,
                              const Affine3f& previousCameraPose, const void* depth_host, size_t step, int rows, int cols)
{
  std::vector<int> result(ranks_.size(), 0);
  forEachRank([&](int rank, KinFu& kf) {
    Rank* r = ranks_[(size_t)rank];
    kf.beginFrame();
    r->depth.upload(depth_host, step, rows, cols);
    result[(size_t)rank] = kf(inputCameraMotion, currentCameraPose, previousCameraPose, r->depth) ? 1 : 0;
  });
  for (size_t i = 1; i < result.size(); ++i)
    if (result[i] != result[0])
      throw std::runtime_error("ShardedKinFu: ranks disagree on the outcome of a frame");
  return result[0] != 0;
}

void ShardedKinFu::resetVolume()
{
  forEachRank([](int, KinFu& kf) { kf.resetVolume(); });
}
void ShardedKinFu::resetPose()
{
  forEachRank([](int, KinFu& kf) { kf.resetPose(); });
}
Affine3f ShardedKinFu::getCameraPose(int time)
{
  Affine3f pose;
  forEachRank([&](int rank, KinFu& kf) {
    if (rank == 0)
      pose = kf.getCameraPose(time);
  });
  return pose;
}

void ShardedKinFu::downloadVolume(void* voxels_out)
{
  const size_t plane = (size_t)params_.volume_dims[0] * (size_t)params_.volume_dims[1];
  forEachRank([&](int, KinFu& kf) {
    int z0, z1;
    kf.tsdf().ownedLayers(z0, z1);
    size_t voxels = 0;
    const void* src = kf.tsdf().ownedData(voxels);
    if (yakhip_download(cuda::context(), (char*)voxels_out + plane * (size_t)z0 * 4u, src, voxels * 4u) != 0)
      throw std::runtime_error(yakhip_last_error());
  });
}

void ShardedKinFu::downloadModelMaps(int rank, void* points_out, void* normals_out)
{
  forEachRank([&](int r, KinFu& kf) {
    if (r != rank)
      return;
    kf.downloadModelMaps(points_out, normals_out);
  });
}
}  // namespace kfusion
