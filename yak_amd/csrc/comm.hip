// comm.hip -- the exchange steps of the z-slab sharded volume behind the C ABI: communicators,
// all-reduce on the context's stream, and the composite of per-slab ray casts.  The reference has one GPU and
// no communication at all: nothing here has a counterpart to cite; the arithmetic being
// composited is the ray cast's (raycast.hip, yak/src/cuda/tsdf_volume.cu:288-372).
//
// Three transports behind one handle:
//   rccl    ncclAllReduce on the context's stream (RCCL over xGMI), one rank per GPU, ranks being processes or
//           threads.  librccl is opened on first use (dlopen), so the library still loads where only the HIP
//           runtime is installed and single-GPU callers never touch it.
//   local   ranks are threads of ONE process (each with its own context, on the same or on different GPUs);
//           the values meet in host memory behind a barrier.  For tests on a one-GPU box (RCCL refuses two
//           ranks on one device) and as a fallback; blocking.
//   custom  a caller-supplied all-reduce (e.g. over torch.distributed / gloo, MPI).
// Every transport hands every rank the same bits (the sum is formed once, in rank order, or by RCCL's ring,
// which broadcasts one result), so ranks that solve from all-reduced values stay in lockstep.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>

#include <rocprim/rocprim.hpp>

#include "yakhip_internal.h"

namespace yakhip
{
namespace
{
// ---- librccl, opened lazily ------------------------------------------------------------------------
struct Rccl
{
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

int rccl(Rccl** out)
{
  static std::mutex m;
  static Rccl lib;
  std::lock_guard<std::mutex> lock(m);
  if (!lib.handle)
  {
    const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char* n : names)
      if ((lib.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr)
        break;
    if (!lib.handle)
      return fail(YAKHIP_ENOGPU, "librccl could not be opened: %s", dlerror());
    lib.GetUniqueId = (decltype(lib.GetUniqueId))dlsym(lib.handle, "ncclGetUniqueId");
    lib.CommInitRank = (decltype(lib.CommInitRank))dlsym(lib.handle, "ncclCommInitRank");
    lib.CommDestroy = (decltype(lib.CommDestroy))dlsym(lib.handle, "ncclCommDestroy");
    lib.CommAbort = (decltype(lib.CommAbort))dlsym(lib.handle, "ncclCommAbort");
    lib.AllReduce = (decltype(lib.AllReduce))dlsym(lib.handle, "ncclAllReduce");
    lib.GetErrorString = (decltype(lib.GetErrorString))dlsym(lib.handle, "ncclGetErrorString");
    if (!lib.GetUniqueId || !lib.CommInitRank || !lib.CommDestroy || !lib.CommAbort || !lib.AllReduce || !lib.GetErrorString)
    {
      dlclose(lib.handle);
      lib.handle = nullptr;
      return fail(YAKHIP_ENOGPU, "librccl lacks an expected entry point");
    }
  }
  *out = &lib;
  return 0;
}

int check_nccl(Rccl* lib, ncclResult_t r, const char* what)
{
  if (r == ncclSuccess)
    return 0;
  return fail(1000 + (int)r, "%s: %s (ncclResult %d)", what, lib->GetErrorString(r), (int)r);
}

// ---- the local transport: ranks are threads of one process ------------------------------------------
struct LocalGroup
{
  std::mutex m;
  std::condition_variable cv;
  int world = 0, arrived = 0;
  unsigned long long generation = 0;
  std::vector<std::vector<uint32_t>> stage;  // one host vector per rank
  bool broken = false;

  // true when every rank arrived.  A barrier that completed stays completed: a peer that has gone on to leave the
  // group (destroy or abort set `broken`) before this rank woke up does not turn its finished exchange into an error.
  bool barrier()
  {
    std::unique_lock<std::mutex> lock(m);
    if (broken)
      return false;
    const unsigned long long g = generation;
    if (++arrived == world)
    {
      arrived = 0;
      ++generation;
      cv.notify_all();
      return true;
    }
    cv.wait(lock, [&] { return generation != g || broken; });
    return generation != g;
  }

  void leave()
  {
    std::lock_guard<std::mutex> lock(m);
    broken = true;  // ranks waiting in a barrier that can no longer complete give up instead of hanging
    cv.notify_all();
  }
};
}  // namespace
}  // namespace yakhip

struct yakhip_comm
{
  int kind;  // 0 rccl, 1 local, 2 custom
  int rank, world;
  ncclComm_t nccl;
  std::shared_ptr<yakhip::LocalGroup> group;
  yakhip_allreduce_fn fn;
  void* user;
  // device scratch of the ray-cast composite (allocated on first use, owned by the communicator)
  void* scratch;
  size_t scratch_bytes;
  std::atomic<bool> aborted;  // yakhip_comm_abort ran: every later collective fails, destroy frees what is left
};

namespace yakhip
{
int comm_allreduce(yakhip_ctx* ctx, yakhip_comm* comm, void* buf, size_t count, int op)
{
  if (comm->aborted.load())
    return fail(YAKHIP_EABORTED, "communicator of rank %d was aborted", comm->rank);
  if (comm->world == 1 && comm->kind != 0)
    return 0;
  if (comm->kind == 0)
  {
    Rccl* lib = nullptr;
    int rc = rccl(&lib);
    if (rc != 0)
      return rc;
    const ncclDataType_t dt = op == YAKHIP_SUM_F32 ? ncclFloat32 : ncclInt32;
    const ncclRedOp_t ro = op == YAKHIP_MIN_I32 ? ncclMin : ncclSum;
    return check_nccl(lib, lib->AllReduce(buf, buf, count, dt, ro, comm->nccl, ctx->stream), "ncclAllReduce");
  }
  if (comm->kind == 2)
  {
    const int rc = comm->fn(comm->user, buf, count, op, (void*)ctx->stream);
    return rc == 0 ? 0 : fail(rc, "custom all-reduce callback returned %d", rc);
  }
  // local: through host memory, summed in rank order by every rank (identical bits everywhere)
  LocalGroup& g = *comm->group;
  std::vector<uint32_t>& mine = g.stage[(size_t)comm->rank];
  mine.resize(count);
  YAKHIP_TRY(hipMemcpyAsync(mine.data(), buf, count * 4u, hipMemcpyDeviceToHost, ctx->stream));
  YAKHIP_TRY(hipStreamSynchronize(ctx->stream));
  if (!g.barrier())
    return fail(YAKHIP_EABORTED, "local communicator: a rank left the group");
  std::vector<uint32_t> out(count);
  for (size_t i = 0; i < count; ++i)
  {
    if (op == YAKHIP_SUM_F32)
    {
      float s = 0.f;
      for (int r = 0; r < comm->world; ++r)
      {
        float v;
        memcpy(&v, &g.stage[(size_t)r][i], 4);
        s = r == 0 ? v : s + v;
      }
      memcpy(&out[i], &s, 4);
    }
    else if (op == YAKHIP_MIN_I32)
    {
      int32_t s = (int32_t)g.stage[0][i];
      for (int r = 1; r < comm->world; ++r)
        s = std::min(s, (int32_t)g.stage[(size_t)r][i]);
      out[i] = (uint32_t)s;
    }
    else
    {
      uint32_t s = 0;
      for (int r = 0; r < comm->world; ++r)
        s += g.stage[(size_t)r][i];
      out[i] = s;
    }
  }
  if (!g.barrier())  // everybody has read every stage vector: they may be overwritten by the next call
    return fail(YAKHIP_EABORTED, "local communicator: a rank left the group");
  YAKHIP_TRY(hipMemcpyAsync(buf, out.data(), count * 4u, hipMemcpyHostToDevice, ctx->stream));
  return check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}

int comm_rank(const yakhip_comm* c) { return c->rank; }
int comm_world(const yakhip_comm* c) { return c->world; }

// ---- composite of the per-slab ray casts ---------------------------------------------------------
// keys / payload images are the outputs of yakhip_raycast_slab on this rank.  Step 1: best = MIN over ranks of
// the keys; every rank zeroes its payload where it does not hold the best key; SUM over ranks of the payload
// bit patterns (one non-zero term: exact).  Step 2 (after yakhip_raycast_resolve on the composited images):
// the pixels whose best key is odd -- the same list, in the same order, on every rank -- are composited once
// more; exactly one rank (the one whose layers hold the refined point) has turned such a key even.
constexpr int kKeyNone = 0x7fffffff;
// What a winner ships (round 3; seven words until then: vertex, normal, depth): the march time of its refined point and
// the normal.  Every rank rebuilds the point, the vertex and the depth from the time with the ray cast's own
// instructions (raycast.hip: composite_unpack_kernel).
constexpr int kPayloadWords = 4;

__device__ __forceinline__ int32_t* key_at(const CompositeImages& im, int y, int x)
{
  return (int32_t*)(im.keys + (size_t)y * im.keys_pitch) + x;
}

// payload words of pixel i, zeroed unless this rank holds the best key.  keys (image) / ts: what this rank's slab ray
// cast left per pixel; best: the keys after the MIN over the ranks (the ray cast wrote this rank's into it: no packing pass).
__global__ __launch_bounds__(256) void mask_pack_kernel(const uint8_t* normals, uint32_t normals_pitch, const uint8_t* keys,
                                                        uint32_t keys_pitch, int rows, int cols, const float* ts,
                                                        const int32_t* best, uint32_t* packed)
{
  const int i = (int)(blockIdx.x * 256u + threadIdx.x);
  const int n = rows * cols;
  if (i >= n)
    return;
  const int y = i / cols, x = i % cols;
  const int32_t b = best[i];
  uint32_t w[kPayloadWords] = { 0, 0, 0, 0 };
  if (b != kKeyNone && ((const int32_t*)(keys + (size_t)y * keys_pitch))[x] == b)
  {
    w[0] = __builtin_bit_cast(uint32_t, ts[i]);
    if ((b & 1) == 0)  // (an odd key: the normals image holds the point, which the time gives as well)
    {
      const uint4 q = ((const uint4*)(normals + (size_t)y * normals_pitch))[x];
      w[1] = q.x, w[2] = q.y, w[3] = q.z;
    }
  }
#pragma unroll
  for (int k = 0; k < kPayloadWords; ++k)
    packed[(size_t)k * n + i] = w[k];
}

struct OddKey
{
  const int32_t* best;
  __device__ bool operator()(int i) const { return best[i] != kKeyNone && (best[i] & 1); }
};

// deferred pixels after the resolve: entry j = {outcome, normal} of list[j], zeros beyond the list and on the ranks that
// did not settle the pixel.  outcome 1: the normal; 2: the normal came out NaN and the hit is discarded
// (tsdf_volume.cu:340-348) -- vertex and depth, which every rank holds since the first step, then go too.
constexpr int kEntryWords = 4;
__global__ __launch_bounds__(256) void gather_deferred_kernel(CompositeImages im, const int32_t* best, const int* list,
                                                               const unsigned* count, int cap, uint32_t* entries,
                                                               unsigned* overflow)
{
  const int j = (int)(blockIdx.x * 256u + threadIdx.x);
  if (j == 0 && *count > (unsigned)cap)
    atomicAdd(overflow, 1u);
  if (j >= cap)
    return;
  uint32_t w[kEntryWords] = { 0, 0, 0, 0 };
  if ((unsigned)j < *count)
  {
    const int i = list[j], y = i / im.cols, x = i % im.cols;
    const int32_t k = *key_at(im, y, x);
    if ((k & 1) == 0 && k == best[i] - 1)  // this rank's resolve settled the pixel
    {
      const uint4 q = ((const uint4*)(im.normals + (size_t)y * im.normals_pitch))[x];
      const bool discarded = (q.x & 0x7fffffffu) > 0x7f800000u;
      w[0] = discarded ? 2u : 1u;
      if (!discarded)
        w[1] = q.x, w[2] = q.y, w[3] = q.z;
    }
  }
#pragma unroll
  for (int k = 0; k < kEntryWords; ++k)
    entries[(size_t)k * cap + j] = w[k];
}

__global__ __launch_bounds__(256) void scatter_deferred_kernel(CompositeImages im, const int32_t* best, const int* list,
                                                                const unsigned* count, int cap, const uint32_t* entries)
{
  const int j = (int)(blockIdx.x * 256u + threadIdx.x);
  if (j >= cap || (unsigned)j >= *count)
    return;
  const uint32_t outcome = entries[j];
  if (outcome == 0u)
    return;  // no slab holds the refined point (it left the volume): the pixel stays as the first step left it
  const int i = list[j], y = i / im.cols, x = i % im.cols;
  const uint32_t qnan = 0x7fc00000u;
  if (outcome == 1u)
    ((uint4*)(im.normals + (size_t)y * im.normals_pitch))[x] =
        uint4{ entries[(size_t)1 * cap + j], entries[(size_t)2 * cap + j], entries[(size_t)3 * cap + j], 0u };
  else
  {
    ((uint4*)(im.normals + (size_t)y * im.normals_pitch))[x] = uint4{ qnan, qnan, qnan, qnan };
    if (im.points)
      ((uint4*)(im.points + (size_t)y * im.points_pitch))[x] = uint4{ qnan, qnan, qnan, qnan };
    if (im.depth)
      ((uint16_t*)(im.depth + (size_t)y * im.depth_pitch))[x] = (uint16_t)0;
  }
  *key_at(im, y, x) = best[i] - 1;
}

// A NaN vertex with a NaN normal is how yakhip_raycast_resolve discards a hit; its payload words are NaN bit
// patterns and travel through the integer sum like any others.

struct CompositeScratch
{
  int32_t* best;       // the keys: this rank's own after the slab ray cast (RaycastParams::key_out), the smallest over the
                       // ranks after the MIN (the keys image keeps this rank's own until the composite is done)
  float* ts;           // this rank's march times (RaycastParams::ts_out)
  uint32_t* packed;
  int* list;
  unsigned* count;     // [0] list length, [1] overflow events (accumulated until read)
  uint32_t* entries;
  void* select_temp;
  size_t select_bytes;
  int cap;
};

static int composite_scratch(yakhip_ctx* ctx, yakhip_comm* comm, int n, int cap_request, CompositeScratch& s)
{
  s.cap = cap_request > 0 ? cap_request : (n / 64 > 4096 ? n / 64 : 4096);
  if (s.cap > n)
    s.cap = n;
  size_t select_bytes = 0;
  OddKey pred{ nullptr };
  YAKHIP_TRY(rocprim::select(nullptr, select_bytes, rocprim::counting_iterator<int>(0), (int*)nullptr, (unsigned*)nullptr,
                             (size_t)n, pred, ctx->stream));
  auto up = [](size_t v) { return (v + 255u) & ~(size_t)255u; };
  const size_t need = 256u + 2 * up((size_t)n * 4u) + up((size_t)n * 4u * kPayloadWords) + up((size_t)n * 4u) +
                      up((size_t)s.cap * 4u * kEntryWords) + up(select_bytes);
  if (comm->scratch_bytes < need)
  {
    if (comm->scratch)
    {
      YAKHIP_TRY(hipStreamSynchronize(ctx->stream));
      YAKHIP_TRY(hipFree(comm->scratch));
      comm->scratch = nullptr;
      comm->scratch_bytes = 0;
    }
    YAKHIP_TRY(hipMalloc(&comm->scratch, need));
    comm->scratch_bytes = need;
    YAKHIP_TRY(hipMemsetAsync(comm->scratch, 0, need, ctx->stream));
  }
  char* p = (char*)comm->scratch;
  s.count = (unsigned*)p, p += 256;
  s.best = (int32_t*)p, p += up((size_t)n * 4u);
  s.ts = (float*)p, p += up((size_t)n * 4u);
  s.packed = (uint32_t*)p, p += up((size_t)n * 4u * kPayloadWords);
  s.list = (int*)p, p += up((size_t)n * 4u);
  s.entries = (uint32_t*)p, p += up((size_t)s.cap * 4u * kEntryWords);
  s.select_temp = p;
  s.select_bytes = select_bytes;
  return 0;
}

int composite_buffers(yakhip_ctx* ctx, yakhip_comm* comm, int n, int cap_request, int32_t** keys_flat, float** ts_flat)
{
  CompositeScratch s;
  const int rc = composite_scratch(ctx, comm, n, cap_request, s);
  if (rc != 0)
    return rc;
  *keys_flat = s.best, *ts_flat = s.ts;
  return 0;
}

// rp: the parameters the slab ray cast of this frame ran with (its key_out / ts_out are composite_buffers' arrays)
int composite_first(yakhip_ctx* ctx, yakhip_comm* comm, const RaycastParams& rp, int cap_request)
{
  const int n = rp.rows * rp.cols;
  CompositeScratch s;
  int rc = composite_scratch(ctx, comm, n, cap_request, s);
  if (rc != 0)
    return rc;
  if (rp.key_out != s.best || rp.ts_out != s.ts)
    return fail(YAKHIP_EINVAL, "composite: the slab ray cast did not write into the communicator's buffers");
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if ((rc = comm_allreduce(ctx, comm, s.best, (size_t)n, YAKHIP_MIN_I32)) != 0)
    return rc;
  hipLaunchKernelGGL(mask_pack_kernel, dim3(blocks), dim3(256), 0, ctx->stream, rp.normals, rp.normals_pitch, rp.keys,
                     rp.keys_pitch, rp.rows, rp.cols, s.ts, s.best, s.packed);
  if ((rc = comm_allreduce(ctx, comm, s.packed, (size_t)n * kPayloadWords, YAKHIP_SUM_I32)) != 0)
    return rc;
  if ((rc = launch_composite_unpack(ctx, rp, s.best, s.packed)) != 0)
    return rc;
  // the pixels still to be resolved, in pixel order (the same list on every rank)
  OddKey pred{ s.best };
  size_t bytes = s.select_bytes;
  YAKHIP_TRY(rocprim::select(s.select_temp, bytes, rocprim::counting_iterator<int>(0), s.list, s.count, (size_t)n, pred,
                             ctx->stream));
  return check(hipGetLastError(), "composite kernels");
}

int composite_second(yakhip_ctx* ctx, yakhip_comm* comm, const CompositeImages& im, int cap_request)
{
  const int n = im.rows * im.cols;
  CompositeScratch s;
  int rc = composite_scratch(ctx, comm, n, cap_request, s);
  if (rc != 0)
    return rc;
  const unsigned blocks = (unsigned)((s.cap + 255) / 256);
  hipLaunchKernelGGL(gather_deferred_kernel, dim3(blocks), dim3(256), 0, ctx->stream, im, s.best, s.list, s.count, s.cap,
                     s.entries, s.count + 1);
  if ((rc = comm_allreduce(ctx, comm, s.entries, (size_t)s.cap * kEntryWords, YAKHIP_SUM_I32)) != 0)
    return rc;
  hipLaunchKernelGGL(scatter_deferred_kernel, dim3(blocks), dim3(256), 0, ctx->stream, im, s.best, s.list, s.count, s.cap,
                     s.entries);
  return check(hipGetLastError(), "composite kernels");
}

int composite_status(yakhip_ctx* ctx, yakhip_comm* comm, unsigned out[2])
{
  out[0] = out[1] = 0;
  if (!comm->scratch)
    return 0;
  YAKHIP_TRY(hipMemcpyAsync(out, comm->scratch, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
  YAKHIP_TRY(hipMemsetAsync((char*)comm->scratch + 4, 0, 4, ctx->stream));
  return check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}
}  // namespace yakhip

using namespace yakhip;

extern "C" {

int yakhip_comm_unique_id(void* id_out, size_t bytes)
{
  if (!id_out || bytes < sizeof(ncclUniqueId))
    return fail(YAKHIP_EINVAL, "comm_unique_id: needs a buffer of at least %zu bytes", sizeof(ncclUniqueId));
  Rccl* lib = nullptr;
  int rc = rccl(&lib);
  if (rc != 0)
    return rc;
  ncclUniqueId id;
  if ((rc = check_nccl(lib, lib->GetUniqueId(&id), "ncclGetUniqueId")) != 0)
    return rc;
  memset(id_out, 0, bytes);
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

static yakhip_comm* new_comm(int kind, int rank, int world)
{
  yakhip_comm* c = new yakhip_comm();
  c->kind = kind, c->rank = rank, c->world = world;
  c->nccl = nullptr, c->fn = nullptr, c->user = nullptr, c->scratch = nullptr, c->scratch_bytes = 0;
  c->aborted.store(false);
  return c;
}

int yakhip_comm_create_rccl(yakhip_ctx* ctx, yakhip_comm** out, const void* unique_id, int world, int rank)
{
  if (!ctx || !out || !unique_id || world < 1 || rank < 0 || rank >= world)
    return fail(YAKHIP_EINVAL, "comm_create_rccl: bad argument");
  *out = nullptr;
  YAKHIP_TRY(hipSetDevice(ctx->device));
  Rccl* lib = nullptr;
  int rc = rccl(&lib);
  if (rc != 0)
    return rc;
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclComm_t nc = nullptr;
  if ((rc = check_nccl(lib, lib->CommInitRank(&nc, world, id, rank), "ncclCommInitRank")) != 0)
    return rc;
  yakhip_comm* c = new_comm(0, rank, world);
  c->nccl = nc;
  *out = c;
  return 0;
}

int yakhip_comm_create_local(int world, yakhip_comm** comms_out)
{
  if (world < 1 || !comms_out)
    return fail(YAKHIP_EINVAL, "comm_create_local: bad argument");
  auto g = std::make_shared<LocalGroup>();
  g->world = world;
  g->stage.resize((size_t)world);
  for (int r = 0; r < world; ++r)
  {
    comms_out[r] = new_comm(1, r, world);
    comms_out[r]->group = g;
  }
  return 0;
}

int yakhip_comm_create_custom(yakhip_comm** out, int world, int rank, yakhip_allreduce_fn fn, void* user)
{
  if (!out || !fn || world < 1 || rank < 0 || rank >= world)
    return fail(YAKHIP_EINVAL, "comm_create_custom: bad argument");
  yakhip_comm* c = new_comm(2, rank, world);
  c->fn = fn, c->user = user;
  *out = c;
  return 0;
}

// The failure path: a rank that cannot go on (an exception mid-frame) aborts its communicator so that its peers do
// not wait for it for ever.  Local transport: the group is marked broken and every rank waiting in a barrier
// returns YAKHIP_EABORTED.  RCCL: ncclCommAbort ends the collectives of THIS communicator that are in flight (its
// kernels stop polling for the missing peer and the stream drains); peers blocked on the rank that failed are
// released by aborting THEIR communicators too, which any thread may do (ShardedKinFu::forEachRank does it from the
// thread that dispatched the frame).  After an abort every collective on the communicator fails; destroy it.
int yakhip_comm_abort(yakhip_comm* comm)
{
  if (!comm)
    return fail(YAKHIP_EINVAL, "comm_abort: null communicator");
  if (comm->aborted.exchange(true))
    return 0;
  int rc = 0;
  if (comm->kind == 0 && comm->nccl)
  {
    Rccl* lib = nullptr;
    if ((rc = rccl(&lib)) == 0)
      rc = check_nccl(lib, lib->CommAbort(comm->nccl), "ncclCommAbort");
  }
  if (comm->kind == 1 && comm->group)
    comm->group->leave();
  return rc;
}

int yakhip_comm_destroy(yakhip_comm* comm)
{
  if (!comm)
    return 0;
  int rc = 0;
  if (comm->scratch)
    (void)hipFree(comm->scratch);
  if (comm->kind == 0 && comm->nccl && !comm->aborted.load())  // ncclCommAbort has freed an aborted communicator
  {
    Rccl* lib = nullptr;
    if ((rc = rccl(&lib)) == 0)
      rc = check_nccl(lib, lib->CommDestroy(comm->nccl), "ncclCommDestroy");
  }
  if (comm->kind == 1 && comm->group)
    comm->group->leave();
  delete comm;
  return rc;
}

int yakhip_comm_info(const yakhip_comm* comm, int* rank, int* world)
{
  if (!comm)
    return fail(YAKHIP_EINVAL, "comm_info: null communicator");
  if (rank)
    *rank = comm->rank;
  if (world)
    *world = comm->world;
  return 0;
}

int yakhip_comm_allreduce(yakhip_ctx* ctx, yakhip_comm* comm, void* buf_dev, size_t count, int op)
{
  if (!ctx || !comm || !buf_dev)
    return fail(YAKHIP_EINVAL, "comm_allreduce: null argument");
  if (op != YAKHIP_SUM_F32 && op != YAKHIP_MIN_I32 && op != YAKHIP_SUM_I32)
    return fail(YAKHIP_EINVAL, "comm_allreduce: unknown operation %d", op);
  if (((uintptr_t)buf_dev & 3) != 0)
    return fail(YAKHIP_EALIGN, "comm_allreduce: buffer not 4-byte aligned");
  YAKHIP_TRY(hipSetDevice(ctx->device));
  if (count == 0)
    return 0;
  return comm_allreduce(ctx, comm, buf_dev, count, op);
}

}  // extern "C"
