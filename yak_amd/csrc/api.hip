// api.hip -- the extern "C" surface declared in include/yakhip.h: argument checking, context,
// device memory, and dispatch to the kernel launchers.  No kernels here.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <thread>

#include "solve6.hpp"
#include "yakhip_internal.h"

namespace yakhip
{
static thread_local char g_error[512] = "";

int fail(int code, const char* fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
  return code;
}

int check(hipError_t e, const char* what)
{
  if (e == hipSuccess)
    return 0;
  snprintf(g_error, sizeof(g_error), "%s: %s (hipError %d)", what, hipGetErrorString(e), (int)e);
  return (int)e;
}

int timing_begin(yakhip_ctx* ctx, std::pair<hipEvent_t, hipEvent_t>** ev)
{
  *ev = nullptr;
  if (!ctx->kernel_timing)
    return 0;
  if (ctx->timing_used == ctx->timing_events.size())
  {
    std::pair<hipEvent_t, hipEvent_t> e;
    YAKHIP_TRY(hipEventCreate(&e.first));
    YAKHIP_TRY(hipEventCreate(&e.second));
    ctx->timing_events.push_back(e);
  }
  *ev = &ctx->timing_events[ctx->timing_used++];
  return check(hipEventRecord((*ev)->first, ctx->stream), "hipEventRecord");
}

int timing_end(yakhip_ctx* ctx, std::pair<hipEvent_t, hipEvent_t>* ev)
{
  return ev ? check(hipEventRecord(ev->second, ctx->stream), "hipEventRecord") : 0;
}

// Makes the context's device current for the calling thread.
static int enter(const yakhip_ctx* ctx)
{
  if (!ctx)
    return fail(YAKHIP_EINVAL, "null context");
  return check(hipSetDevice(ctx->device), "hipSetDevice");
}

static int check_volume(const yakhip_volume* v)
{
  if (!v || !v->data)
    return fail(YAKHIP_EINVAL, "volume: null pointer");
  if (v->dims[0] <= 0 || v->dims[1] <= 0 || v->dims[2] <= 0)
    return fail(YAKHIP_EINVAL, "volume: dims must be positive, got %d x %d x %d", v->dims[0], v->dims[1], v->dims[2]);
  if (v->z0 < 0 || v->z0 >= v->z1 || v->z1 > v->dims[2])
    return fail(YAKHIP_EINVAL, "volume: slab [%d, %d) is not inside [0, %d)", v->z0, v->z1, v->dims[2]);
  if ((uintptr_t)v->data % 4 != 0)
    return fail(YAKHIP_EALIGN, "volume: data pointer must be 4-byte aligned");
  // in-plane offsets are 32-bit on the device; a slab is addressed with 64-bit offsets
  if ((uint64_t)v->dims[0] * (uint64_t)v->dims[1] > 0x7fffffffull)
    return fail(YAKHIP_ERANGE, "volume: one z layer exceeds 2^31 voxels");
  if ((uintptr_t)v->surface_map % 4 != 0)
    return fail(YAKHIP_EALIGN, "volume: surface map pointer must be 4-byte aligned");
  return 0;
}

// 32-bit words of a slab's surface map (yakhip.h): one bit per 16^3 brick of its layers, then one per 64^3 cell
static size_t surface_map_words(const yakhip_volume* v)
{
  return surface_brick_words(v->dims[0], v->dims[1], v->z1 - v->z0) + surface_cell_words(v->dims[0], v->dims[1], v->z1 - v->z0);
}
}  // namespace yakhip

using namespace yakhip;

extern "C" {

const char* yakhip_last_error(void) { return g_error; }

int yakhip_device_count(int* count)
{
  if (!count)
    return fail(YAKHIP_EINVAL, "null count");
  *count = 0;
  hipError_t e = hipGetDeviceCount(count);
  if (e == hipErrorNoDevice)
  {
    *count = 0;
    return 0;
  }
  return check(e, "hipGetDeviceCount");
}

int yakhip_device_name(int device, char* buf, size_t buflen)
{
  if (!buf || buflen == 0)
    return fail(YAKHIP_EINVAL, "null buffer");
  hipDeviceProp_t prop;
  YAKHIP_TRY(hipGetDeviceProperties(&prop, device));
  snprintf(buf, buflen, "%s (%s)", prop.name, prop.gcnArchName);
  return 0;
}

int yakhip_ctx_create(yakhip_ctx** out, int device)
{
  if (!out)
    return fail(YAKHIP_EINVAL, "null out pointer");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(YAKHIP_ENOGPU, "no HIP device visible (%s)", hipGetErrorString(e));
  if (device < 0 || device >= n)
    return fail(YAKHIP_EINVAL, "device %d out of range [0, %d)", device, n);
  YAKHIP_TRY(hipSetDevice(device));
  yakhip_ctx* ctx = new yakhip_ctx();
  ctx->device = device;
  ctx->z_chunk = 0;
  ctx->nontemporal = 1;
  ctx->force_vx = 0;
  ctx->zu = 0;
  ctx->group_test = 1;
  ctx->brick_test = 1;
  ctx->chunk_order = -1;
  ctx->surface_skip = 1;
  ctx->raycast_bands = -1;
  ctx->raycast_segments = 0;
  ctx->batch_as_singles = 0;
  if (const char* e = getenv("YAK_HIP_RAYCAST_SEGMENTS"))  // (tuning / tests: 1, 2 or 4 waves per ray)
    if ((e[0] == '1' || e[0] == '2' || e[0] == '4') && e[1] == 0)
      ctx->raycast_segments = e[0] - '0';
  ctx->chunk_min = 0;
  ctx->chunk_ramp = 0;
  ctx->checkpoint = 1;
  ctx->tile_shift = 0;
  ctx->tile_side = 1;
  ctx->mc_z_chunk = 0;
  ctx->thr = nullptr;
  ctx->thr_bytes = 0;
  ctx->ckpt = nullptr;
  ctx->ckpt_bytes = 0;
  ctx->static_scratch = nullptr;
  ctx->brick_stats = nullptr;
  ctx->classes = nullptr;
  ctx->classes_bytes = 0;
  ctx->classes_n = 0;
  ctx->stats_host = nullptr;
  ctx->launch_seq = ctx->classes_seq = 0;
  ctx->launches_one = ctx->launches_two = 0;
  memset(ctx->seq_signature, 0, sizeof(ctx->seq_signature));
  ctx->cptab = nullptr;
  ctx->cptab_bytes = 0;
  ctx->defer_mixed = 1;
  if (const char* e = getenv("YAK_HIP_DEFER_MIXED"))  // (tuning / tests: 0 never, 1 large launches, 2 every launch)
    if (e[0] >= '0' && e[0] <= '2' && e[1] == 0)
      ctx->defer_mixed = e[0] - '0';
  ctx->mixed_waves = 0;
  ctx->refine = 1;
  ctx->num_cus = 0;
  ctx->thr_read[0] = ctx->thr_read[1] = nullptr;
  ctx->prepare_used = 0;
  ctx->thr_read_valid[0] = ctx->thr_read_valid[1] = 0;
  ctx->thr_slot_read = -1;
  ctx->icp_scratch = nullptr;
  ctx->icp_scratch_bytes = 0;
  ctx->icp_pinned = nullptr;
  ctx->icp_ticket = 0;
  ctx->icp_poll = 1;
  ctx->prepared.valid = false;
  ctx->prepared_next = 0;
  ctx->mc_scratch = nullptr;
  ctx->mc_scratch_bytes = 0;
  ctx->weld_scratch = nullptr;
  ctx->icp_sums = nullptr;
  ctx->weld_scratch_bytes = 0;
  ctx->mc_pinned = nullptr;
  ctx->stage[0] = ctx->stage[1] = nullptr;
  ctx->stage_done[0] = ctx->stage_done[1] = nullptr;
  ctx->kernel_timing = 0;
  ctx->composite_capacity = 0;
  ctx->timing_used = 0;
  int rc = check(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking), "hipStreamCreateWithFlags");
  if (rc != 0)
  {
    delete ctx;
    return rc;
  }
  ctx->stream = ctx->own_stream;
  ctx->side_stream = nullptr;
  ctx->side_done = nullptr;
  ctx->main_mark = nullptr;
  ctx->main_marked = 0;
  ctx->main_stream = nullptr;
  ctx->main_stream_saved = 0;
  // set-up for yakhip_integrate_static: this device's address of the library's file-scope scratch array,
  // asked for once per context with the context's device current (each GPU holds its own copy)
  if ((rc = static_scratch_address(&ctx->static_scratch)) != 0)
  {
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return rc;
  }
  if (hipHostMalloc((void**)&ctx->stats_host, 64, hipHostMallocDefault) == hipSuccess)
    memset(ctx->stats_host, 0, 64);
  else
    ctx->stats_host = nullptr;  // (then every large launch is one kernel, as before)
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0)
      cus = 256;
    ctx->num_cus = cus;
  }
  *out = ctx;
  return 0;
}

int yakhip_ctx_destroy(yakhip_ctx* ctx)
{
  if (!ctx)
    return 0;
  int rc = enter(ctx);
  if (rc == 0)
  {
    rc = check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    if (ctx->thr)
      (void)hipFree(ctx->thr);
    if (ctx->ckpt)
      (void)hipFree(ctx->ckpt);
    if (ctx->brick_stats)
      (void)hipFree(ctx->brick_stats);
    if (ctx->classes)
      (void)hipFree(ctx->classes);
    if (ctx->stats_host)
      (void)hipHostFree(ctx->stats_host);
    if (ctx->cptab)
      (void)hipFree(ctx->cptab);
    for (int i = 0; i < 2; ++i)
      if (ctx->thr_read[i])
        (void)hipEventDestroy(ctx->thr_read[i]);
    if (ctx->icp_scratch)
      (void)hipFree(ctx->icp_scratch);
    if (ctx->icp_sums)
      (void)hipFree(ctx->icp_sums);
    if (ctx->icp_pinned)
      (void)hipHostFree(ctx->icp_pinned);
    if (ctx->mc_scratch)
      (void)hipFree(ctx->mc_scratch);
    if (ctx->weld_scratch)
      (void)hipFree(ctx->weld_scratch);
    if (ctx->mc_pinned)
      (void)hipHostFree(ctx->mc_pinned);
    for (int i = 0; i < 2; ++i)
    {
      if (ctx->stage[i])
        (void)hipHostFree(ctx->stage[i]);
      if (ctx->stage_done[i])
        (void)hipEventDestroy(ctx->stage_done[i]);
    }
    for (auto& ev : ctx->timing_events)
    {
      (void)hipEventDestroy(ev.first);
      (void)hipEventDestroy(ev.second);
    }
    if (ctx->side_stream)
    {
      (void)hipStreamSynchronize(ctx->side_stream);
      (void)hipStreamDestroy(ctx->side_stream);
      (void)hipEventDestroy(ctx->side_done);
      (void)hipEventDestroy(ctx->main_mark);
    }
    int rc2 = check(hipStreamDestroy(ctx->own_stream), "hipStreamDestroy");
    if (rc == 0)
      rc = rc2;
  }
  delete ctx;
  return rc;
}

int yakhip_ctx_set_stream(yakhip_ctx* ctx, void* hip_stream)
{
  if (!ctx)
    return fail(YAKHIP_EINVAL, "null context");
  if (ctx->main_stream_saved)
    return fail(YAKHIP_EINVAL, "ctx_set_stream: between ctx_side_begin and ctx_side_end");
  ctx->stream = (hipStream_t)hip_stream;
  return 0;
}

int yakhip_ctx_use_own_stream(yakhip_ctx* ctx)
{
  if (!ctx)
    return fail(YAKHIP_EINVAL, "null context");
  if (ctx->main_stream_saved)
    return fail(YAKHIP_EINVAL, "ctx_use_own_stream: between ctx_side_begin and ctx_side_end");
  ctx->stream = ctx->own_stream;
  return 0;
}

int yakhip_ctx_stream(yakhip_ctx* ctx, void** hip_stream)
{
  if (!ctx || !hip_stream)
    return fail(YAKHIP_EINVAL, "ctx_stream: null argument");
  *hip_stream = (void*)ctx->stream;
  return 0;
}

int yakhip_ctx_side_begin(yakhip_ctx* ctx)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (ctx->main_stream_saved)
    return 0;  // already there: the section goes on until yakhip_ctx_side_end
  if (!ctx->side_stream)
  {
    YAKHIP_TRY(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
    YAKHIP_TRY(hipEventCreateWithFlags(&ctx->side_done, hipEventDisableTiming));
    YAKHIP_TRY(hipEventCreateWithFlags(&ctx->main_mark, hipEventDisableTiming));
  }
  // the side stream may run ahead of the main one by what has been enqueued there since the last side_end, no more
  if (ctx->main_marked)
    YAKHIP_TRY(hipStreamWaitEvent(ctx->side_stream, ctx->main_mark, 0));
  ctx->main_stream = ctx->stream;
  ctx->main_stream_saved = 1;
  ctx->stream = ctx->side_stream;
  return 0;
}

int yakhip_ctx_side_end(yakhip_ctx* ctx)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!ctx->main_stream_saved)
    return fail(YAKHIP_EINVAL, "ctx_side_end: not on the side stream");
  ctx->stream = ctx->main_stream;
  ctx->main_stream_saved = 0;
  YAKHIP_TRY(hipEventRecord(ctx->side_done, ctx->side_stream));
  YAKHIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->side_done, 0));
  ctx->main_marked = 1;
  return check(hipEventRecord(ctx->main_mark, ctx->stream), "hipEventRecord");
}

int yakhip_ctx_synchronize(yakhip_ctx* ctx)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  return check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}

int yakhip_ctx_set_option(yakhip_ctx* ctx, const char* key, int value)
{
  if (!ctx || !key)
    return fail(YAKHIP_EINVAL, "null context or key");
  if (strcmp(key, "z_chunk") == 0)
  {
    if (value < 0)
      return fail(YAKHIP_EINVAL, "z_chunk must be >= 0");
    ctx->z_chunk = value;
    return 0;
  }
  if (strcmp(key, "nontemporal") == 0)
  {
    ctx->nontemporal = value != 0;
    return 0;
  }
  if (strcmp(key, "vx") == 0)
  {
    if (value != 0 && value != 1 && value != 2 && value != 4)
      return fail(YAKHIP_EINVAL, "vx must be 0, 1, 2 or 4");
    ctx->force_vx = value;
    return 0;
  }
  if (strcmp(key, "kernel_timing") == 0)
  {
    ctx->kernel_timing = value != 0;
    ctx->timing_used = 0;  // restart the accumulation
    return 0;
  }
  if (strcmp(key, "group_test") == 0)
  {
    ctx->group_test = value != 0;
    return 0;
  }
  if (strcmp(key, "brick_test") == 0)
  {
    ctx->brick_test = value != 0;
    return 0;
  }
  if (strcmp(key, "chunk_order") == 0)
  {
    ctx->chunk_order = (int)value;
    return 0;
  }
  if (strcmp(key, "icp_poll") == 0)
  {
    ctx->icp_poll = value != 0;
    return 0;
  }
  if (strcmp(key, "raycast_bands") == 0)
  {
    ctx->raycast_bands = (int)value;  // -1 automatic, 0 image order, 1 XCD bands, >= 2 scattered with about this stride
    return 0;
  }
  if (strcmp(key, "raycast_segments") == 0)
  {
    if (value != 0 && value != 1 && value != 2 && value != 4)
      return fail(YAKHIP_EINVAL, "set_option(raycast_segments): 0 (default), 1, 2 or 4");
    ctx->raycast_segments = (int)value;
    return 0;
  }
  if (strcmp(key, "surface_skip") == 0)
  {
    ctx->surface_skip = value != 0;
    return 0;
  }
  if (strcmp(key, "chunk_min") == 0)
  {
    ctx->chunk_min = (int)value;
    return 0;
  }
  if (strcmp(key, "chunk_ramp") == 0)
  {
    ctx->chunk_ramp = (int)value;
    return 0;
  }
  if (strcmp(key, "brick_stats") == 0)
  {
    int rc = enter(ctx);
    if (rc != 0)
      return rc;
    if (value && !ctx->brick_stats)
    {
      YAKHIP_TRY(hipMalloc((void**)&ctx->brick_stats, 4 * sizeof(uint32_t)));
      YAKHIP_TRY(hipMemsetAsync(ctx->brick_stats, 0, 4 * sizeof(uint32_t), ctx->stream));
    }
    else if (!value && ctx->brick_stats)
    {
      YAKHIP_TRY(hipStreamSynchronize(ctx->stream));
      YAKHIP_TRY(hipFree(ctx->brick_stats));
      ctx->brick_stats = nullptr;
    }
    return 0;
  }
  if (strcmp(key, "checkpoint") == 0)
  {
    ctx->checkpoint = value != 0;
    return 0;
  }
  if (strcmp(key, "defer_mixed") == 0)
  {
    if (value < 0 || value > 2)
      return fail(YAKHIP_EINVAL, "defer_mixed must be 0 (never), 1 (large launches) or 2 (every launch with the brick test)");
    ctx->defer_mixed = value;
    return 0;
  }
  if (strcmp(key, "mixed_waves") == 0)
  {
    if (value < 0 || value > 32)
      return fail(YAKHIP_EINVAL, "mixed_waves must be 0 (default) or 1..32");
    ctx->mixed_waves = value;
    return 0;
  }
  if (strcmp(key, "refine") == 0)
  {
    ctx->refine = value != 0;
    return 0;
  }
  if (strcmp(key, "tile") == 0)
  {
    if (value != 0 && (value < 3 || value > 6))
      return fail(YAKHIP_EINVAL, "tile must be 0 (default) or 3..6");
    ctx->tile_shift = value;
    return 0;
  }
  if (strcmp(key, "composite_capacity") == 0)
  {
    if (value < 0)
      return fail(YAKHIP_EINVAL, "composite_capacity must be >= 0");
    ctx->composite_capacity = value;
    return 0;
  }
  if (strcmp(key, "tile_side") == 0)
  {
    ctx->tile_side = value != 0;
    return 0;
  }
  if (strcmp(key, "mc_z_chunk") == 0)
  {
    if (value < 0)
      return fail(YAKHIP_EINVAL, "mc_z_chunk must be >= 0");
    ctx->mc_z_chunk = value;
    return 0;
  }
  if (strcmp(key, "zu") == 0)
  {
    if (value != 0 && value != 1 && value != 2 && value != 4)
      return fail(YAKHIP_EINVAL, "zu must be 0, 1, 2 or 4");
    ctx->zu = value;
    return 0;
  }
  return fail(YAKHIP_EINVAL, "unknown option '%s'", key);
}

int yakhip_ctx_kernel_time(yakhip_ctx* ctx, double* total_ms, int* launches)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!total_ms || !launches)
    return fail(YAKHIP_EINVAL, "kernel_time: null out pointer");
  double sum = 0.0;
  for (size_t i = 0; i < ctx->timing_used; ++i)
  {
    YAKHIP_TRY(hipEventSynchronize(ctx->timing_events[i].second));
    float ms = 0.f;
    YAKHIP_TRY(hipEventElapsedTime(&ms, ctx->timing_events[i].first, ctx->timing_events[i].second));
    sum += ms;
  }
  *total_ms = sum;
  *launches = (int)ctx->timing_used;
  ctx->timing_used = 0;
  return 0;
}

int yakhip_ctx_kernel_times(yakhip_ctx* ctx, float* ms_out, float* start_ms_out, int capacity, int* launches)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!launches || capacity < 0 || (capacity > 0 && !ms_out))
    return fail(YAKHIP_EINVAL, "kernel_times: null out pointer or negative capacity");
  for (size_t i = 0; i < ctx->timing_used; ++i)
  {
    YAKHIP_TRY(hipEventSynchronize(ctx->timing_events[i].second));
    if ((int)i >= capacity)
      continue;
    YAKHIP_TRY(hipEventElapsedTime(&ms_out[i], ctx->timing_events[i].first, ctx->timing_events[i].second));
    if (start_ms_out)
      YAKHIP_TRY(hipEventElapsedTime(&start_ms_out[i], ctx->timing_events[0].first, ctx->timing_events[i].first));
  }
  *launches = (int)ctx->timing_used;
  ctx->timing_used = 0;
  return 0;
}

int yakhip_ctx_launch_counts(yakhip_ctx* ctx, unsigned long long counts[2])
{
  if (!ctx || !counts)
    return fail(YAKHIP_EINVAL, "launch_counts: null context or out pointer");
  counts[0] = ctx->launches_one;
  counts[1] = ctx->launches_two;
  return 0;
}

int yakhip_ctx_brick_stats(yakhip_ctx* ctx, unsigned long long counts[3])
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!counts)
    return fail(YAKHIP_EINVAL, "brick_stats: null out pointer");
  if (!ctx->brick_stats)
    return fail(YAKHIP_EINVAL, "brick_stats: switch the \"brick_stats\" option on first");
  uint32_t host[4] = { 0, 0, 0, 0 };
  YAKHIP_TRY(hipMemcpyAsync(host, ctx->brick_stats, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
  YAKHIP_TRY(hipMemsetAsync(ctx->brick_stats, 0, sizeof(host), ctx->stream));
  YAKHIP_TRY(hipStreamSynchronize(ctx->stream));
  counts[0] = host[1];  // dead
  counts[1] = host[2];  // front
  counts[2] = host[0];  // mixed
  return 0;
}

int yakhip_selftest_division(yakhip_ctx* ctx, unsigned long long counts[4])
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!counts)
    return fail(YAKHIP_EINVAL, "selftest: null out pointer");
  return launch_selftest_division(ctx, counts);
}

// ---- device memory ------------------------------------------------------------------------------

int yakhip_malloc(yakhip_ctx* ctx, void** ptr, size_t bytes)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!ptr || bytes == 0)
    return fail(YAKHIP_EINVAL, "malloc: null out pointer or zero size");
  *ptr = nullptr;
  return check(hipMalloc(ptr, bytes), "hipMalloc");
}

int yakhip_malloc_pitch(yakhip_ctx* ctx, void** ptr, size_t* pitch_bytes, size_t width_bytes, size_t rows)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!ptr || !pitch_bytes || width_bytes == 0 || rows == 0)
    return fail(YAKHIP_EINVAL, "malloc_pitch: null out pointer or zero extent");
  // rows padded to 256 bytes: every row start is aligned for 16-byte vector access
  const size_t pitch = (width_bytes + 255u) & ~(size_t)255u;
  *ptr = nullptr;
  rc = check(hipMalloc(ptr, pitch * rows), "hipMalloc");
  if (rc == 0)
    *pitch_bytes = pitch;
  return rc;
}

int yakhip_free(yakhip_ctx* ctx, void* ptr)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!ptr)
    return 0;
  return check(hipFree(ptr), "hipFree");
}

// Host <-> device copies go through two pinned staging buffers of the context.  The caller's memory is
// pageable (std::vector, cv::Mat), and the runtime's own path for that is slow where it matters here: a
// pitched copy from pageable memory is done row by row (4 ms for one 640 x 480 depth frame, ten times the
// rest of the frame's work); staged, the same frame takes 0.1 ms.  Rows are packed into the buffer by the
// CPU while the previous buffer's DMA runs.
static constexpr size_t kStageBytes = 8u << 20;

static int stage_buffers(yakhip_ctx* ctx)
{
  for (int i = 0; i < 2; ++i)
  {
    if (!ctx->stage[i])
      YAKHIP_TRY(hipHostMalloc(&ctx->stage[i], kStageBytes, hipHostMallocDefault));
    if (!ctx->stage_done[i])
      YAKHIP_TRY(hipEventCreateWithFlags(&ctx->stage_done[i], hipEventDisableTiming));
  }
  return 0;
}

// rows x width bytes between pitched host memory and pitched device memory, blocking
// The host side of a staged piece: rows of `width` bytes between a staging buffer and the caller's memory.  A large piece
// is split over a few threads: one thread copies a fresh 512 MiB destination in 45 ms (a page fault per 4 KiB), four
// take 11.5 ms (tools/memcpy_bench.py) -- the DMA beside it needs about as long.
static void host_copy(uint8_t* dst, size_t dst_pitch, const uint8_t* src, size_t src_pitch, size_t width, size_t rows)
{
  const size_t total = width * rows;
  unsigned hw = std::thread::hardware_concurrency();
  const unsigned n_threads = total < (4u << 20) || hw < 2 ? 1u : (hw < 4 ? hw : 4u);
  auto part = [=](unsigned k) {
    if (rows >= n_threads)
    {
      const size_t r0 = rows * k / n_threads, r1 = rows * (k + 1) / n_threads;
      for (size_t r = r0; r < r1; ++r)
        memcpy(dst + r * dst_pitch, src + r * src_pitch, width);
    }
    else  // a few long rows: every thread takes its share of each
    {
      const size_t c0 = width * k / n_threads, c1 = width * (k + 1) / n_threads;
      for (size_t r = 0; r < rows; ++r)
        memcpy(dst + r * dst_pitch + c0, src + r * src_pitch + c0, c1 - c0);
    }
  };
  if (n_threads == 1)
    return part(0);
  std::thread helpers[3];
  unsigned started = 1;
  try
  {
    for (; started < n_threads; ++started)
      helpers[started - 1] = std::thread(part, started);
  }
  catch (...)  // no thread to be had: the caller does the parts that found none (nothing may be thrown across the C ABI)
  {
  }
  part(0);
  for (unsigned k = started; k < n_threads; ++k)
    part(k);
  for (unsigned k = 1; k < started; ++k)
    helpers[k - 1].join();
}

// yakhip_integrate_prepare remembers an image by its address: a library call that writes [dst, dst + bytes) makes
// what was prepared from that image stale
static void forget_prepared(yakhip_ctx* ctx, const void* dst, size_t bytes)
{
  const uint8_t* img = (const uint8_t*)ctx->prepared.dists;
  if (ctx->prepared.valid && img + (size_t)ctx->prepared.pitch * (size_t)ctx->prepared.rows > (const uint8_t*)dst &&
      img < (const uint8_t*)dst + bytes)
    ctx->prepared.valid = false;
}

static int staged_copy(yakhip_ctx* ctx, uint8_t* dev, size_t dev_pitch, uint8_t* host, size_t host_pitch, size_t width,
                       size_t rows, bool to_device)
{
  int rc = stage_buffers(ctx);
  if (rc != 0)
    return rc;
  if (to_device)
    forget_prepared(ctx, dev, dev_pitch * (rows - 1) + width);
  if (dev_pitch == width && host_pitch == width)
  {  // contiguous on both sides: one long row, cut into pieces below
    width *= rows;
    rows = 1;
    dev_pitch = host_pitch = width;
  }
  // a piece = whole rows if a row fits the buffer, else a part of one row
  const size_t rows_per_piece = width <= kStageBytes ? kStageBytes / width : 1;
  const size_t cols_per_piece = width <= kStageBytes ? width : kStageBytes;
  struct Piece
  {
    size_t row, col, nrows, ncols;
  };
  Piece pending = { 0, 0, 0, 0 };  // download: the piece whose DMA was issued last and is not unpacked yet
  int n = 0;
  for (size_t row = 0; row < rows; row += rows_per_piece)
    for (size_t col = 0; col < width; col += cols_per_piece, ++n)
    {
      const Piece pc = { row, col, rows - row < rows_per_piece ? rows - row : rows_per_piece,
                         width - col < cols_per_piece ? width - col : cols_per_piece };
      uint8_t* buf = (uint8_t*)ctx->stage[n & 1];
      if (n >= 2)
        YAKHIP_TRY(hipEventSynchronize(ctx->stage_done[n & 1]));  // the DMA that used this buffer two pieces ago
      if (to_device)
      {
        host_copy(buf, pc.ncols, host + pc.row * host_pitch + pc.col, host_pitch, pc.ncols, pc.nrows);
        if (pc.nrows == 1 || dev_pitch == pc.ncols)
          YAKHIP_TRY(hipMemcpyAsync(dev + pc.row * dev_pitch + pc.col, buf, pc.nrows * pc.ncols, hipMemcpyHostToDevice,
                                    ctx->stream));
        else
          YAKHIP_TRY(hipMemcpy2DAsync(dev + pc.row * dev_pitch + pc.col, dev_pitch, buf, pc.ncols, pc.ncols, pc.nrows,
                                      hipMemcpyHostToDevice, ctx->stream));
        YAKHIP_TRY(hipEventRecord(ctx->stage_done[n & 1], ctx->stream));
      }
      else
      {
        if (pc.nrows == 1 || dev_pitch == pc.ncols)
          YAKHIP_TRY(hipMemcpyAsync(buf, dev + pc.row * dev_pitch + pc.col, pc.nrows * pc.ncols, hipMemcpyDeviceToHost,
                                    ctx->stream));
        else
          YAKHIP_TRY(hipMemcpy2DAsync(buf, pc.ncols, dev + pc.row * dev_pitch + pc.col, dev_pitch, pc.ncols, pc.nrows,
                                      hipMemcpyDeviceToHost, ctx->stream));
        YAKHIP_TRY(hipEventRecord(ctx->stage_done[n & 1], ctx->stream));
        if (n >= 1)
        {  // unpack the previous piece while this one is in flight
          YAKHIP_TRY(hipEventSynchronize(ctx->stage_done[(n - 1) & 1]));
          const uint8_t* prev = (const uint8_t*)ctx->stage[(n - 1) & 1];
          host_copy(host + pending.row * host_pitch + pending.col, host_pitch, prev, pending.ncols, pending.ncols, pending.nrows);
        }
        pending = pc;
      }
    }
  YAKHIP_TRY(hipStreamSynchronize(ctx->stream));  // blocking, as device_memory.cpp:129-138, 247-256
  if (!to_device && n >= 1)
  {
    const uint8_t* prev = (const uint8_t*)ctx->stage[(n - 1) & 1];
    host_copy(host + pending.row * host_pitch + pending.col, host_pitch, prev, pending.ncols, pending.ncols, pending.nrows);
  }
  return 0;
}

int yakhip_upload(yakhip_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!dst_dev || !src_host)
    return fail(YAKHIP_EINVAL, "upload: null pointer");
  if (bytes == 0)
    return 0;
  return staged_copy(ctx, (uint8_t*)dst_dev, bytes, (uint8_t*)const_cast<void*>(src_host), bytes, bytes, 1, true);
}

int yakhip_download(yakhip_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!dst_host || !src_dev)
    return fail(YAKHIP_EINVAL, "download: null pointer");
  if (bytes == 0)
    return 0;
  return staged_copy(ctx, (uint8_t*)const_cast<void*>(src_dev), bytes, (uint8_t*)dst_host, bytes, bytes, 1, false);
}

int yakhip_copy(yakhip_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!dst_dev || !src_dev)
    return fail(YAKHIP_EINVAL, "copy: null pointer");
  if (bytes == 0)
    return 0;
  forget_prepared(ctx, dst_dev, bytes);
  YAKHIP_TRY(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}

int yakhip_copy_2d(yakhip_ctx* ctx, void* dst_dev, size_t dst_pitch, const void* src_dev, size_t src_pitch,
                   size_t width_bytes, size_t rows)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!dst_dev || !src_dev)
    return fail(YAKHIP_EINVAL, "copy_2d: null pointer");
  if (dst_pitch < width_bytes || src_pitch < width_bytes)
    return fail(YAKHIP_EINVAL, "copy_2d: pitch smaller than a row");
  if (width_bytes == 0 || rows == 0)
    return 0;
  forget_prepared(ctx, dst_dev, dst_pitch * (rows - 1) + width_bytes);
  YAKHIP_TRY(hipMemcpy2DAsync(dst_dev, dst_pitch, src_dev, src_pitch, width_bytes, rows, hipMemcpyDeviceToDevice,
                              ctx->stream));
  return check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}

int yakhip_upload_2d(yakhip_ctx* ctx, void* dst_dev, size_t dst_pitch, const void* src_host, size_t src_pitch,
                     size_t width_bytes, size_t rows)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!dst_dev || !src_host)
    return fail(YAKHIP_EINVAL, "upload_2d: null pointer");
  if (width_bytes > dst_pitch || width_bytes > src_pitch)
    return fail(YAKHIP_EINVAL, "upload_2d: row width exceeds a pitch");
  if (width_bytes == 0 || rows == 0)
    return 0;
  return staged_copy(ctx, (uint8_t*)dst_dev, dst_pitch, (uint8_t*)const_cast<void*>(src_host), src_pitch, width_bytes,
                     rows, true);
}

int yakhip_download_2d(yakhip_ctx* ctx, void* dst_host, size_t dst_pitch, const void* src_dev, size_t src_pitch,
                       size_t width_bytes, size_t rows)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!dst_host || !src_dev)
    return fail(YAKHIP_EINVAL, "download_2d: null pointer");
  if (width_bytes > dst_pitch || width_bytes > src_pitch)
    return fail(YAKHIP_EINVAL, "download_2d: row width exceeds a pitch");
  if (width_bytes == 0 || rows == 0)
    return 0;
  return staged_copy(ctx, (uint8_t*)const_cast<void*>(src_dev), src_pitch, (uint8_t*)dst_host, dst_pitch, width_bytes,
                     rows, false);
}

// ---- hot path -----------------------------------------------------------------------------------

int yakhip_clear_volume(yakhip_ctx* ctx, const yakhip_volume* vol)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if ((rc = check_volume(vol)) != 0)
    return rc;
  const size_t n = (size_t)vol->dims[0] * (size_t)vol->dims[1] * (size_t)(vol->z1 - vol->z0);
  if (vol->surface_map)
    YAKHIP_TRY(hipMemsetAsync(vol->surface_map, 0, surface_map_words(vol) * 4u, ctx->stream));
  return launch_clear(ctx, (uint32_t*)vol->data, n);
}

int yakhip_surface_map_bytes(const yakhip_volume* vol, size_t* bytes)
{
  int rc = check_volume(vol);
  if (rc != 0)
    return rc;
  if (!bytes)
    return fail(YAKHIP_EINVAL, "surface_map_bytes: null output");
  *bytes = surface_map_words(vol) * 4u;
  return 0;
}

int yakhip_surface_map_rebuild(yakhip_ctx* ctx, const yakhip_volume* vol)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if ((rc = check_volume(vol)) != 0)
    return rc;
  if (!vol->surface_map)
    return fail(YAKHIP_EINVAL, "surface_map_rebuild: the volume has no surface map");
  return launch_surface_scan(ctx, (const uint32_t*)vol->data, vol->dims[0], vol->dims[1], vol->z1 - vol->z0,
                             (uint32_t*)vol->surface_map);
}

static int check_image(const char* who, const void* ptr, size_t pitch, int rows, int cols, size_t elem)
{
  if (!ptr)
    return fail(YAKHIP_EINVAL, "%s: null image pointer", who);
  if (rows <= 0 || cols <= 0)
    return fail(YAKHIP_EINVAL, "%s: image must be non-empty, got %d x %d", who, rows, cols);
  if (pitch < (size_t)cols * elem)
    return fail(YAKHIP_EINVAL, "%s: pitch %zu smaller than a row (%zu bytes)", who, pitch, (size_t)cols * elem);
  if (pitch % elem != 0 || (uintptr_t)ptr % elem != 0)
    return fail(YAKHIP_EALIGN, "%s: pointer and pitch must be multiples of %zu bytes", who, elem);
  if (pitch > 0xffffffffull)
    return fail(YAKHIP_ERANGE, "%s: pitch too large", who);
  return 0;
}

int yakhip_compute_dists(yakhip_ctx* ctx, const void* depth_dev, size_t depth_pitch, void* dists_dev,
                         size_t dists_pitch, int rows, int cols, const float intr[4])
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!intr)
    return fail(YAKHIP_EINVAL, "compute_dists: null intrinsics");
  if ((rc = check_image("compute_dists(depth)", depth_dev, depth_pitch, rows, cols, 2)) != 0)
    return rc;
  if ((rc = check_image("compute_dists(dists)", dists_dev, dists_pitch, rows, cols, 2)) != 0)
    return rc;
  if (rows > 65535)
    return fail(YAKHIP_ERANGE, "compute_dists: more than 65535 rows");
  forget_prepared(ctx, dists_dev, dists_pitch * (size_t)(rows - 1) + (size_t)cols * 2u);
  return launch_compute_dists(ctx, (const uint8_t*)depth_dev, depth_pitch, (uint8_t*)dists_dev, dists_pitch, rows,
                              cols, intr[0], intr[1], intr[2], intr[3]);
}

static int fill_integrate_params(IntegrateParams& p, const yakhip_volume* vol, const void* dists_dev, size_t dists_pitch,
                                 int rows, int cols, const float R[9], const float t[3], const float intr[4])
{
  int rc;
  if (!R || !t || !intr)
    return fail(YAKHIP_EINVAL, "integrate: null pose or intrinsics");
  if ((rc = check_image("integrate(dists)", dists_dev, dists_pitch, rows, cols, 2)) != 0)
    return rc;
  if (vol->max_weight < 1 || vol->max_weight > 32767)
    return fail(YAKHIP_ERANGE, "integrate: max_weight %d outside [1, 32767] (the weight lane is int16)",
                vol->max_weight);
  if (!(vol->trunc_dist > 0.f))
    return fail(YAKHIP_EINVAL, "integrate: trunc_dist must be positive");

  memset(&p, 0, sizeof(p));  // (everything the launcher fills in later starts as "none")
  p.vol = (uint32_t*)vol->data;
  p.bricks = (uint32_t*)vol->surface_map;
  p.bricks_x = (vol->dims[0] + 15) >> 4;
  p.bricks_y = (vol->dims[1] + 15) >> 4;
  p.bricks_z = (vol->z1 - vol->z0 + 15) >> 4;
  p.bricks_words = (uint32_t)surface_brick_words(vol->dims[0], vol->dims[1], vol->z1 - vol->z0);
  p.dists = (const uint8_t*)dists_dev;
  p.thr = nullptr;  // thr, coarse, cw/ch and the grp_* constants are filled by the launcher
  p.coarse = nullptr;
  p.cw = p.ch = 0;
  p.grp_enable = 0;
  p.grp_margin = p.grp_cols_hi = p.grp_rows_hi = p.grp_span = p.grp_inv_bk = p.grp_bias = p.grp_zmin = 0.f;
  p.dists_pitch = (uint32_t)dists_pitch;
  p.rows = rows;
  p.cols = cols;
  p.X = vol->dims[0];
  p.Y = vol->dims[1];
  p.z0 = vol->z0;
  p.z1 = vol->z1;
  p.z_chunk = 0;
  p.tile_shift = 6;
  p.tile_side = 1;
  p.plane = (uint32_t)vol->dims[0] * (uint32_t)vol->dims[1];
  p.vsx = vol->voxel_size[0];
  p.vsy = vol->voxel_size[1];
  p.vsz = vol->voxel_size[2];
  p.trunc = vol->trunc_dist;
  p.trunc_inv = 1.f / vol->trunc_dist;  // tsdf_volume.cu:125, formed on the host
  p.trunc_hi = vol->trunc_dist * 0x1.008p+0f;  // 1 + 2^-9
  p.max_weight = vol->max_weight;
  memcpy(p.R, R, sizeof(p.R));
  memcpy(p.t, t, sizeof(p.t));
  p.fx = intr[0];
  p.fy = intr[1];
  p.cx = intr[2];
  p.cy = intr[3];
  return 0;
}

int yakhip_integrate(yakhip_ctx* ctx, const yakhip_volume* vol, const void* dists_dev, size_t dists_pitch, int rows,
                     int cols, const float R[9], const float t[3], const float intr[4])
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if ((rc = check_volume(vol)) != 0)
    return rc;
  IntegrateParams p;
  if ((rc = fill_integrate_params(p, vol, dists_dev, dists_pitch, rows, cols, R, t, intr)) != 0)
    return rc;
  return launch_integrate(ctx, p);
}

int yakhip_integrate_prepare(yakhip_ctx* ctx, const yakhip_volume* vol, const void* dists_dev, size_t dists_pitch, int rows,
                             int cols)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if ((rc = check_volume(vol)) != 0)
    return rc;
  static const float eye[9] = { 1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f }, zero[3] = { 0.f, 0.f, 0.f }, unit[4] = { 1.f, 1.f, 0.f, 0.f };
  IntegrateParams p;
  if ((rc = fill_integrate_params(p, vol, dists_dev, dists_pitch, rows, cols, eye, zero, unit)) != 0)
    return rc;
  return launch_integrate_prepare(ctx, p);
}

int yakhip_integrate_static(yakhip_ctx* ctx, const yakhip_volume* vol, const void* dists_dev, size_t dists_pitch,
                            int rows, int cols, const float R[9], const float t[3], const float intr[4])
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if ((rc = check_volume(vol)) != 0)
    return rc;
  IntegrateParams p;
  if ((rc = fill_integrate_params(p, vol, dists_dev, dists_pitch, rows, cols, R, t, intr)) != 0)
    return rc;
  return launch_integrate(ctx, p, true);
}

int yakhip_integrate_batch(yakhip_ctx* ctx, const yakhip_volume* vol, const yakhip_frame* frames, int n_frames)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if ((rc = check_volume(vol)) != 0)
    return rc;
  if (!frames || n_frames < 1)
    return fail(YAKHIP_EINVAL, "integrate_batch: needs at least one frame");
  std::vector<IntegrateParams> ps((size_t)n_frames);
  for (int i = 0; i < n_frames; ++i)
  {
    const yakhip_frame& f = frames[i];
    if ((rc = fill_integrate_params(ps[(size_t)i], vol, f.dists, f.dists_pitch, f.rows, f.cols, f.R, f.t, f.intr)) != 0)
      return rc;
    if (f.rows != frames[0].rows || f.cols != frames[0].cols)
      return fail(YAKHIP_EINVAL, "integrate_batch: frame %d is %d x %d, frame 0 is %d x %d", i, f.cols, f.rows,
                  frames[0].cols, frames[0].rows);
  }
  return launch_integrate_batch(ctx, ps.data(), n_frames);
}

// layers of halo a slab needs on either side so that a march-ending pair and its refinement read held voxels
// only: the next sample is at most ceil(step / voxel) + 1 layers from the owned one, the trilinear taps one
// more, the gradient taps ceil(delta / voxel) more
static int raycast_halo_layers(const yakhip_volume* vol, float step_factor, float gradient_delta_factor)
{
  const float vs = vol->voxel_size[2];
  const double step = (double)(vol->trunc_dist * step_factor) / vs, delta = (double)(vs * gradient_delta_factor) / vs;
  const double h = ceil(step) + ceil(delta) + 2.0;
  return h < 1e9 ? (int)h : 1000000000;
}

// half- and quarter-size outputs of yakhip_raycast_points_pyramid (the quarter size may be absent)
struct RaycastPyramid
{
  void *points[2], *normals[2];
  size_t points_pitch[2], normals_pitch[2];
};

// keys == nullptr: the whole volume (vol->z0 = 0, vol->z1 = dims[2]).  Otherwise slab mode: vol holds the
// layers [z0, z1), of which [z_own0, z_own1) are the slab's own.
static int raycast_common(yakhip_ctx* ctx, const yakhip_volume* vol, const float R[9], const float t[3],
                          const float Rinv[9], const float intr[4], float step_factor, float gradient_delta_factor,
                          void* points, size_t points_pitch, void* depth, size_t depth_pitch, void* normals,
                          size_t normals_pitch, int rows, int cols, void* keys = nullptr, size_t keys_pitch = 0,
                          int z_own0 = 0, int z_own1 = 0, const RaycastPyramid* pyr = nullptr, int32_t* key_out = nullptr,
                          float* ts_out = nullptr, RaycastParams* params_out = nullptr)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if ((rc = check_volume(vol)) != 0)
    return rc;
  if (!keys && (vol->z0 != 0 || vol->z1 != vol->dims[2]))
    return fail(YAKHIP_EINVAL, "raycast: needs the whole volume, got the slab [%d, %d) of %d layers", vol->z0, vol->z1,
                vol->dims[2]);
  if (!R || !t || !Rinv || !intr)
    return fail(YAKHIP_EINVAL, "raycast: null pose or intrinsics");
  if (!(vol->trunc_dist > 0.f) || !(step_factor > 0.f) || !(gradient_delta_factor > 0.f))
    return fail(YAKHIP_EINVAL, "raycast: trunc_dist, step factor and gradient delta factor must be positive");
  if ((rc = check_image("raycast(normals)", normals, normals_pitch, rows, cols, 16)) != 0)
    return rc;
  if (points && (rc = check_image("raycast(points)", points, points_pitch, rows, cols, 16)) != 0)
    return rc;
  if (depth && (rc = check_image("raycast(depth)", depth, depth_pitch, rows, cols, 2)) != 0)
    return rc;
  if (keys)
  {
    if ((rc = check_image("raycast(keys)", keys, keys_pitch, rows, cols, 4)) != 0)
      return rc;
    if (!(vol->z0 <= z_own0 && z_own0 < z_own1 && z_own1 <= vol->z1))
      return fail(YAKHIP_EINVAL, "raycast_slab: own layers [%d, %d) must be a non-empty part of the held layers [%d, %d)",
                  z_own0, z_own1, vol->z0, vol->z1);
    const int h = raycast_halo_layers(vol, step_factor, gradient_delta_factor);
    if (z_own0 - vol->z0 < (z_own0 < h ? z_own0 : h) || vol->z1 - z_own1 < (vol->dims[2] - z_own1 < h ? vol->dims[2] - z_own1 : h))
      return fail(YAKHIP_EINVAL, "raycast_slab: own layers [%d, %d) of held [%d, %d) leave less than the %d halo layers needed",
                  z_own0, z_own1, vol->z0, vol->z1, h);
  }

  RaycastParams p;
  p.vol = (const uint32_t*)vol->data;
  p.X = vol->dims[0], p.Y = vol->dims[1], p.Z = vol->dims[2];
  p.vsx = vol->voxel_size[0], p.vsy = vol->voxel_size[1], p.vsz = vol->voxel_size[2];
  // tsdf_volume.cu:399-402 (host arithmetic of device::raycast), :306 (box_max)
  p.ivx = 1.f / p.vsx, p.ivy = 1.f / p.vsy, p.ivz = 1.f / p.vsz;
  p.gdx = p.vsx * gradient_delta_factor, p.gdy = p.vsy * gradient_delta_factor, p.gdz = p.vsz * gradient_delta_factor;
  p.bx = p.vsx * (float)p.X - p.vsx, p.by = p.vsy * (float)p.Y - p.vsy, p.bz = p.vsz * (float)p.Z - p.vsz;
  p.time_step = vol->trunc_dist * step_factor;
  memcpy(p.R, R, sizeof(p.R));
  memcpy(p.t, t, sizeof(p.t));
  memcpy(p.Rinv, Rinv, sizeof(p.Rinv));
  p.finvx = 1.f / intr[0], p.finvy = 1.f / intr[1];  // precomp.cpp:73-76
  p.cx = intr[2], p.cy = intr[3];
  p.rows = rows, p.cols = cols;
  p.points = (uint8_t*)points, p.normals = (uint8_t*)normals, p.depth = (uint8_t*)depth;
  p.points_pitch = (uint32_t)points_pitch, p.normals_pitch = (uint32_t)normals_pitch;
  p.depth_pitch = (uint32_t)depth_pitch;
  for (int l = 0; l < 2; ++l)
  {
    p.pyr_points[l] = p.pyr_normals[l] = nullptr;
    p.pyr_points_pitch[l] = p.pyr_normals_pitch[l] = 0;
  }
  if (pyr)
  {
    int r = rows, c = cols;
    for (int l = 0; l < 2 && pyr->points[l]; ++l)
    {
      r /= 2, c /= 2;  // imgproc.cu:386-387
      if (r < 1 || c < 1)
        return fail(YAKHIP_EINVAL, "raycast_points_pyramid: a %d x %d image has no level %d", cols, rows, l + 1);
      if ((rc = check_image("raycast_points_pyramid(points)", pyr->points[l], pyr->points_pitch[l], r, c, 16)) != 0)
        return rc;
      if ((rc = check_image("raycast_points_pyramid(normals)", pyr->normals[l], pyr->normals_pitch[l], r, c, 16)) != 0)
        return rc;
      p.pyr_points[l] = (uint8_t*)pyr->points[l], p.pyr_normals[l] = (uint8_t*)pyr->normals[l];
      p.pyr_points_pitch[l] = (uint32_t)pyr->points_pitch[l], p.pyr_normals_pitch[l] = (uint32_t)pyr->normals_pitch[l];
    }
  }
  p.keys = (uint8_t*)keys, p.keys_pitch = (uint32_t)keys_pitch;
  p.z_base = vol->z0, p.z_held = vol->z1 - vol->z0, p.z_own0 = z_own0, p.z_own1 = z_own1;
  p.key_out = key_out, p.ts_out = ts_out;
  p.skip = (const uint32_t*)vol->surface_map;
  p.skip_x = (vol->dims[0] + 15) >> 4, p.skip_y = (vol->dims[1] + 15) >> 4;
  const size_t words = surface_brick_words(vol->dims[0], vol->dims[1], vol->z1 - vol->z0);
  p.skip_words = words <= 0x3fffffffull ? (int)words : 0x3fffffff;
  p.cell_x = (p.skip_x + 3) >> 2, p.cell_y = (p.skip_y + 3) >> 2, p.cell_z = (((vol->z1 - vol->z0 + 15) >> 4) + 3) >> 2;
  p.cell_words = (int)surface_cell_words(vol->dims[0], vol->dims[1], vol->z1 - vol->z0);
  // Crossing free space in one go (raycast.hip) rests on the marched samples -- positions and times built by
  // repeated rounded additions -- staying close to the straight line: every addition is off by at most half an
  // ulp of the largest value it can reach (a coordinate inside the volume; the time from the camera to the far
  // corner), over at most diagonal / step additions.  Positions must stay within an eighth of a voxel, times
  // within a quarter of a step (the jump keeps two steps and 16 voxels in hand).
  {
    const float diag = sqrtf(p.bx * p.bx + p.by * p.by + p.bz * p.bz);
    const float vmin = fminf(fminf(p.vsx, p.vsy), p.vsz), vmax = fmaxf(fmaxf(p.vsx, p.vsy), p.vsz);
    const float pos_reach = fmaxf(fmaxf(p.bx, p.by), p.bz) + 2.f * vmax;
    const float time_reach = sqrtf(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]) + diag;
    const float steps = diag / p.time_step + 8.f;
    p.cell_jump = steps < 1e6f && time_reach < 1e6f && steps * 0x1p-23f * pos_reach < 0.125f * vmin &&
                  steps * 0x1p-23f * time_reach < 0.25f * p.time_step;
  }
  p.vol_bytes = (unsigned long long)vol->dims[0] * (unsigned long long)vol->dims[1] * (unsigned long long)(vol->z1 - vol->z0) * 4ull;
  if (params_out)
    *params_out = p;
  return launch_raycast(ctx, p);
}

int yakhip_raycast_points(yakhip_ctx* ctx, const yakhip_volume* vol, const float R[9], const float t[3],
                          const float Rinv[9], const float intr[4], float step_factor, float gradient_delta_factor,
                          void* points_dev, size_t points_pitch, void* normals_dev, size_t normals_pitch, int rows,
                          int cols)
{
  if (!points_dev)
    return fail(YAKHIP_EINVAL, "raycast(points): null image pointer");
  return raycast_common(ctx, vol, R, t, Rinv, intr, step_factor, gradient_delta_factor, points_dev, points_pitch,
                        nullptr, 0, normals_dev, normals_pitch, rows, cols);
}

int yakhip_raycast_points_pyramid(yakhip_ctx* ctx, const yakhip_volume* vol, const float R[9], const float t[3],
                                  const float Rinv[9], const float intr[4], float step_factor, float gradient_delta_factor,
                                  void* points_dev, size_t points_pitch, void* normals_dev, size_t normals_pitch, int rows,
                                  int cols, void* points1_dev, size_t points1_pitch, void* normals1_dev,
                                  size_t normals1_pitch, void* points2_dev, size_t points2_pitch, void* normals2_dev,
                                  size_t normals2_pitch)
{
  if (!points_dev || !points1_dev || !normals1_dev)
    return fail(YAKHIP_EINVAL, "raycast_points_pyramid: null image pointer");
  if ((points2_dev == nullptr) != (normals2_dev == nullptr))
    return fail(YAKHIP_EINVAL, "raycast_points_pyramid: the quarter-size maps come as a pair or not at all");
  RaycastPyramid pyr = { { points1_dev, points2_dev }, { normals1_dev, normals2_dev }, { points1_pitch, points2_pitch },
                         { normals1_pitch, normals2_pitch } };
  return raycast_common(ctx, vol, R, t, Rinv, intr, step_factor, gradient_delta_factor, points_dev, points_pitch,
                        nullptr, 0, normals_dev, normals_pitch, rows, cols, nullptr, 0, 0, 0, &pyr);
}

int yakhip_raycast_depth(yakhip_ctx* ctx, const yakhip_volume* vol, const float R[9], const float t[3],
                         const float Rinv[9], const float intr[4], float step_factor, float gradient_delta_factor,
                         void* depth_dev, size_t depth_pitch, void* normals_dev, size_t normals_pitch, int rows,
                         int cols)
{
  if (!depth_dev)
    return fail(YAKHIP_EINVAL, "raycast(depth): null image pointer");
  return raycast_common(ctx, vol, R, t, Rinv, intr, step_factor, gradient_delta_factor, nullptr, 0, depth_dev,
                        depth_pitch, normals_dev, normals_pitch, rows, cols);
}

int yakhip_raycast_halo(const yakhip_volume* vol, float step_factor, float gradient_delta_factor, int* layers)
{
  int rc = check_volume(vol);
  if (rc != 0)
    return rc;
  if (!layers || !(vol->trunc_dist > 0.f) || !(step_factor > 0.f) || !(gradient_delta_factor > 0.f))
    return fail(YAKHIP_EINVAL, "raycast_halo: null output or non-positive trunc_dist / factors");
  *layers = raycast_halo_layers(vol, step_factor, gradient_delta_factor);
  return 0;
}

int yakhip_raycast_slab(yakhip_ctx* ctx, const yakhip_volume* vol, int z_own0, int z_own1, const float R[9],
                        const float t[3], const float Rinv[9], const float intr[4], float step_factor,
                        float gradient_delta_factor, void* points_dev, size_t points_pitch, void* depth_dev,
                        size_t depth_pitch, void* normals_dev, size_t normals_pitch, void* keys_dev, size_t keys_pitch,
                        int rows, int cols)
{
  if (!keys_dev)
    return fail(YAKHIP_EINVAL, "raycast_slab: null keys image");
  return raycast_common(ctx, vol, R, t, Rinv, intr, step_factor, gradient_delta_factor, points_dev, points_pitch,
                        depth_dev, depth_pitch, normals_dev, normals_pitch, rows, cols, keys_dev, keys_pitch, z_own0,
                        z_own1);
}

int yakhip_raycast_resolve(yakhip_ctx* ctx, const yakhip_volume* vol, int z_own0, int z_own1, const float Rinv[9],
                           float gradient_delta_factor, void* points_dev, size_t points_pitch, void* depth_dev,
                           size_t depth_pitch, void* normals_dev, size_t normals_pitch, void* keys_dev, size_t keys_pitch,
                           int rows, int cols)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if ((rc = check_volume(vol)) != 0)
    return rc;
  if (!Rinv || !(gradient_delta_factor > 0.f))
    return fail(YAKHIP_EINVAL, "raycast_resolve: null rotation or non-positive gradient delta factor");
  if ((rc = check_image("raycast_resolve(normals)", normals_dev, normals_pitch, rows, cols, 16)) != 0)
    return rc;
  if ((rc = check_image("raycast_resolve(keys)", keys_dev, keys_pitch, rows, cols, 4)) != 0)
    return rc;
  if (points_dev && (rc = check_image("raycast_resolve(points)", points_dev, points_pitch, rows, cols, 16)) != 0)
    return rc;
  if (depth_dev && (rc = check_image("raycast_resolve(depth)", depth_dev, depth_pitch, rows, cols, 2)) != 0)
    return rc;
  if (!(vol->z0 <= z_own0 && z_own0 < z_own1 && z_own1 <= vol->z1))
    return fail(YAKHIP_EINVAL, "raycast_resolve: own layers [%d, %d) must be a non-empty part of the held layers [%d, %d)",
                z_own0, z_own1, vol->z0, vol->z1);
  // the taps around a point of an own layer: ceil(delta / voxel) + 2 layers either way
  const int h = (int)ceil((double)gradient_delta_factor) + 2;
  if (z_own0 - vol->z0 < (z_own0 < h ? z_own0 : h) || vol->z1 - z_own1 < (vol->dims[2] - z_own1 < h ? vol->dims[2] - z_own1 : h))
    return fail(YAKHIP_EINVAL, "raycast_resolve: own layers [%d, %d) of held [%d, %d) leave less than the %d halo layers needed",
                z_own0, z_own1, vol->z0, vol->z1, h);

  RaycastParams p = {};
  memset(&p, 0, sizeof(p));
  p.vol = (const uint32_t*)vol->data;
  p.X = vol->dims[0], p.Y = vol->dims[1], p.Z = vol->dims[2];
  p.vsx = vol->voxel_size[0], p.vsy = vol->voxel_size[1], p.vsz = vol->voxel_size[2];
  p.ivx = 1.f / p.vsx, p.ivy = 1.f / p.vsy, p.ivz = 1.f / p.vsz;
  p.gdx = p.vsx * gradient_delta_factor, p.gdy = p.vsy * gradient_delta_factor, p.gdz = p.vsz * gradient_delta_factor;
  memcpy(p.Rinv, Rinv, sizeof(p.Rinv));
  p.rows = rows, p.cols = cols;
  p.points = (uint8_t*)points_dev, p.normals = (uint8_t*)normals_dev, p.depth = (uint8_t*)depth_dev;
  p.points_pitch = (uint32_t)points_pitch, p.normals_pitch = (uint32_t)normals_pitch;
  p.depth_pitch = (uint32_t)depth_pitch;
  p.keys = (uint8_t*)keys_dev, p.keys_pitch = (uint32_t)keys_pitch;
  p.z_base = vol->z0, p.z_held = vol->z1 - vol->z0, p.z_own0 = z_own0, p.z_own1 = z_own1;
  return launch_raycast_resolve(ctx, p);
}

// ---- cloud extraction ---------------------------------------------------------------------------

int yakhip_extract_cloud(yakhip_ctx* ctx, const yakhip_volume* vol, const float R[9], const float t[3], void* points_dev,
                         unsigned long long capacity_points, unsigned long long* n_points_host)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if ((rc = check_volume(vol)) != 0)
    return rc;
  if (!R || !t || !n_points_host)
    return fail(YAKHIP_EINVAL, "extract_cloud: null pose or count pointer");
  if (points_dev && ((uintptr_t)points_dev & 15) != 0)
    return fail(YAKHIP_EALIGN, "extract_cloud: points buffer not 16-byte aligned");
  return launch_extract_cloud(ctx, (const uint32_t*)vol->data, vol->dims[0], vol->dims[1], vol->z0, vol->z1 - vol->z0,
                              vol->voxel_size, R, t, (float*)points_dev, capacity_points, n_points_host,
                              (const uint32_t*)vol->surface_map);
}

int yakhip_extract_normals(yakhip_ctx* ctx, const yakhip_volume* vol, const void* points_dev, unsigned long long n_points,
                           const float R[9], const float t[3], const float Rinv[9], float gradient_delta_factor,
                           void* normals_dev)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if ((rc = check_volume(vol)) != 0)
    return rc;
  if (vol->z0 != 0 || vol->z1 != vol->dims[2])
    return fail(YAKHIP_EINVAL, "extract_normals: needs the whole volume, got the slab [%d, %d) of %d layers", vol->z0,
                vol->z1, vol->dims[2]);
  if (!R || !t || !Rinv || !(gradient_delta_factor > 0.f))
    return fail(YAKHIP_EINVAL, "extract_normals: null pose or non-positive gradient delta factor");
  if (n_points == 0)
    return 0;
  if (!points_dev || !normals_dev)
    return fail(YAKHIP_EINVAL, "extract_normals: null buffer");
  if ((((uintptr_t)points_dev | (uintptr_t)normals_dev) & 15) != 0)
    return fail(YAKHIP_EALIGN, "extract_normals: buffers not 16-byte aligned");
  RaycastParams p = {};
  memset(&p, 0, sizeof(p));
  p.vol = (const uint32_t*)vol->data;
  p.X = vol->dims[0], p.Y = vol->dims[1], p.Z = vol->dims[2];
  p.vsx = vol->voxel_size[0], p.vsy = vol->voxel_size[1], p.vsz = vol->voxel_size[2];
  p.ivx = 1.f / p.vsx, p.ivy = 1.f / p.vsy, p.ivz = 1.f / p.vsz;  // tsdf_volume.cu:684-689
  p.gdx = p.vsx * gradient_delta_factor, p.gdy = p.vsy * gradient_delta_factor, p.gdz = p.vsz * gradient_delta_factor;
  memcpy(p.R, R, sizeof(p.R));
  memcpy(p.t, t, sizeof(p.t));
  memcpy(p.Rinv, Rinv, sizeof(p.Rinv));
  return launch_extract_normals(ctx, p, points_dev, normals_dev, n_points);
}

// ---- tracking side: depth pre-processing and ICP ----------------------------------------------

int yakhip_bilateral_filter(yakhip_ctx* ctx, const void* src_dev, size_t src_pitch, void* dst_dev, size_t dst_pitch,
                            int rows, int cols, int kernel_size, float sigma_spatial, float sigma_depth)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if ((rc = check_image("bilateral(src)", src_dev, src_pitch, rows, cols, 2)) != 0)
    return rc;
  if ((rc = check_image("bilateral(dst)", dst_dev, dst_pitch, rows, cols, 2)) != 0)
    return rc;
  if (src_dev == dst_dev)
    return fail(YAKHIP_EINVAL, "bilateral: in-place filtering is not supported");
  if (kernel_size < 1 || kernel_size > 255 || !(sigma_spatial > 0.f) || !(sigma_depth > 0.f))
    return fail(YAKHIP_EINVAL, "bilateral: kernel size must be in [1, 255] and both sigmas positive");
  return launch_bilateral(ctx, (const uint8_t*)src_dev, src_pitch, (uint8_t*)dst_dev, dst_pitch, rows, cols,
                          kernel_size, sigma_spatial, sigma_depth);
}

int yakhip_truncate_depth(yakhip_ctx* ctx, void* depth_dev, size_t depth_pitch, int rows, int cols, float max_dist)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if ((rc = check_image("truncate_depth", depth_dev, depth_pitch, rows, cols, 2)) != 0)
    return rc;
  return launch_truncate_depth(ctx, (uint8_t*)depth_dev, depth_pitch, rows, cols, max_dist);
}

int yakhip_depth_pyramid(yakhip_ctx* ctx, const void* src_dev, size_t src_pitch, int src_rows, int src_cols,
                         void* dst_dev, size_t dst_pitch, float sigma_depth)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if ((rc = check_image("depth_pyramid(src)", src_dev, src_pitch, src_rows, src_cols, 2)) != 0)
    return rc;
  if (src_rows < 2 || src_cols < 2)
    return fail(YAKHIP_EINVAL, "depth_pyramid: source must be at least 2 x 2");
  if ((rc = check_image("depth_pyramid(dst)", dst_dev, dst_pitch, src_rows / 2, src_cols / 2, 2)) != 0)
    return rc;
  return launch_depth_pyramid(ctx, (const uint8_t*)src_dev, src_pitch, src_rows, src_cols, (uint8_t*)dst_dev,
                              dst_pitch, sigma_depth);
}

int yakhip_compute_points_normals(yakhip_ctx* ctx, const float intr[4], const void* depth_dev, size_t depth_pitch,
                                  int rows, int cols, void* points_dev, size_t points_pitch, void* normals_dev,
                                  size_t normals_pitch)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!intr)
    return fail(YAKHIP_EINVAL, "compute_points_normals: null intrinsics");
  if ((rc = check_image("compute_points_normals(depth)", depth_dev, depth_pitch, rows, cols, 2)) != 0)
    return rc;
  if ((rc = check_image("compute_points_normals(points)", points_dev, points_pitch, rows, cols, 16)) != 0)
    return rc;
  if ((rc = check_image("compute_points_normals(normals)", normals_dev, normals_pitch, rows, cols, 16)) != 0)
    return rc;
  return launch_points_normals(ctx, intr, (const uint8_t*)depth_dev, depth_pitch, rows, cols, (uint8_t*)points_dev,
                               points_pitch, (uint8_t*)normals_dev, normals_pitch);
}

int yakhip_resize_points_normals(yakhip_ctx* ctx, const void* points_dev, size_t points_pitch, const void* normals_dev,
                                 size_t normals_pitch, int src_rows, int src_cols, void* points_out_dev,
                                 size_t points_out_pitch, void* normals_out_dev, size_t normals_out_pitch)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if ((rc = check_image("resize_points_normals(points)", points_dev, points_pitch, src_rows, src_cols, 16)) != 0)
    return rc;
  if ((rc = check_image("resize_points_normals(normals)", normals_dev, normals_pitch, src_rows, src_cols, 16)) != 0)
    return rc;
  if (src_rows < 2 || src_cols < 2)
    return fail(YAKHIP_EINVAL, "resize_points_normals: source must be at least 2 x 2");
  const int rows = src_rows / 2, cols = src_cols / 2;
  if ((rc = check_image("resize_points_normals(points_out)", points_out_dev, points_out_pitch, rows, cols, 16)) != 0)
    return rc;
  if ((rc = check_image("resize_points_normals(normals_out)", normals_out_dev, normals_out_pitch, rows, cols, 16)) != 0)
    return rc;
  return launch_resize_points_normals(ctx, (const uint8_t*)points_dev, points_pitch, (const uint8_t*)normals_dev,
                                      normals_pitch, src_rows, src_cols, (uint8_t*)points_out_dev, points_out_pitch,
                                      (uint8_t*)normals_out_dev, normals_out_pitch);
}

int yakhip_resize_points_normals2(yakhip_ctx* ctx, const void* points_dev, size_t points_pitch, const void* normals_dev,
                                  size_t normals_pitch, int src_rows, int src_cols, void* points1_dev, size_t points1_pitch,
                                  void* normals1_dev, size_t normals1_pitch, void* points2_dev, size_t points2_pitch,
                                  void* normals2_dev, size_t normals2_pitch)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if ((rc = check_image("resize_points_normals2(points)", points_dev, points_pitch, src_rows, src_cols, 16)) != 0)
    return rc;
  if ((rc = check_image("resize_points_normals2(normals)", normals_dev, normals_pitch, src_rows, src_cols, 16)) != 0)
    return rc;
  if (src_rows < 4 || src_cols < 4)
    return fail(YAKHIP_EINVAL, "resize_points_normals2: source must be at least 4 x 4");
  const int rows1 = src_rows / 2, cols1 = src_cols / 2, rows2 = rows1 / 2, cols2 = cols1 / 2;
  if ((rc = check_image("resize_points_normals2(points, half size)", points1_dev, points1_pitch, rows1, cols1, 16)) != 0)
    return rc;
  if ((rc = check_image("resize_points_normals2(normals, half size)", normals1_dev, normals1_pitch, rows1, cols1, 16)) != 0)
    return rc;
  if ((rc = check_image("resize_points_normals2(points, quarter size)", points2_dev, points2_pitch, rows2, cols2, 16)) != 0)
    return rc;
  if ((rc = check_image("resize_points_normals2(normals, quarter size)", normals2_dev, normals2_pitch, rows2, cols2, 16)) != 0)
    return rc;
  return launch_resize_points_normals2(ctx, (const uint8_t*)points_dev, points_pitch, (const uint8_t*)normals_dev, normals_pitch,
                                       src_rows, src_cols, (uint8_t*)points1_dev, points1_pitch, (uint8_t*)normals1_dev,
                                       normals1_pitch, (uint8_t*)points2_dev, points2_pitch, (uint8_t*)normals2_dev,
                                       normals2_pitch);
}

// ---- debug display -----------------------------------------------------------------------------

static int render_image(yakhip_ctx* ctx, bool from_depth, const void* src_dev, size_t src_pitch, const void* normals_dev,
                        size_t normals_pitch, int rows, int cols, const float intr[4], const float light_pose[3],
                        void* image_dev, size_t image_pitch)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!light_pose || (from_depth && !intr))
    return fail(YAKHIP_EINVAL, "render_image: null light position or intrinsics");
  if (from_depth && (intr[0] == 0.f || intr[1] == 0.f))
    return fail(YAKHIP_EINVAL, "render_image: zero focal length");
  if ((rc = check_image(from_depth ? "render_image(depth)" : "render_image(points)", src_dev, src_pitch, rows, cols,
                        from_depth ? 2 : 16)) != 0)
    return rc;
  if ((rc = check_image("render_image(normals)", normals_dev, normals_pitch, rows, cols, 16)) != 0)
    return rc;
  if ((rc = check_image("render_image(image)", image_dev, image_pitch, rows, cols, 4)) != 0)
    return rc;
  return launch_render_image(ctx, from_depth, (const uint8_t*)src_dev, src_pitch, (const uint8_t*)normals_dev,
                             normals_pitch, rows, cols, intr, light_pose, (uint8_t*)image_dev, image_pitch);
}

int yakhip_render_image_points(yakhip_ctx* ctx, const void* points_dev, size_t points_pitch, const void* normals_dev,
                               size_t normals_pitch, int rows, int cols, const float light_pose[3], void* image_dev,
                               size_t image_pitch)
{
  return render_image(ctx, false, points_dev, points_pitch, normals_dev, normals_pitch, rows, cols, nullptr, light_pose,
                      image_dev, image_pitch);
}

int yakhip_render_image_depth(yakhip_ctx* ctx, const void* depth_dev, size_t depth_pitch, const void* normals_dev,
                              size_t normals_pitch, int rows, int cols, const float intr[4], const float light_pose[3],
                              void* image_dev, size_t image_pitch)
{
  return render_image(ctx, true, depth_dev, depth_pitch, normals_dev, normals_pitch, rows, cols, intr, light_pose,
                      image_dev, image_pitch);
}

int yakhip_render_tangent_colors(yakhip_ctx* ctx, const void* normals_dev, size_t normals_pitch, int rows, int cols,
                                 void* image_dev, size_t image_pitch)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if ((rc = check_image("render_tangent_colors(normals)", normals_dev, normals_pitch, rows, cols, 16)) != 0)
    return rc;
  if ((rc = check_image("render_tangent_colors(image)", image_dev, image_pitch, rows, cols, 4)) != 0)
    return rc;
  return launch_tangent_colors(ctx, (const uint8_t*)normals_dev, normals_pitch, rows, cols, (uint8_t*)image_dev,
                               image_pitch);
}

int yakhip_icp_accumulate(yakhip_ctx* ctx, const float R[9], const float t[3], const float intr[4], int level,
                          float dist_thres, float angle_thres, const void* vcurr_dev, size_t vcurr_pitch,
                          const void* ncurr_dev, size_t ncurr_pitch, const void* vprev_dev, size_t vprev_pitch,
                          const void* nprev_dev, size_t nprev_pitch, int rows, int cols, float sums_host[27],
                          int counts_host[6])
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!R || !t || !intr || !sums_host)
    return fail(YAKHIP_EINVAL, "icp_accumulate: null pose, intrinsics or output");
  if (level < 0 || level > 15)
    return fail(YAKHIP_EINVAL, "icp_accumulate: pyramid level %d outside [0, 15]", level);
  if ((rc = check_image("icp_accumulate(vcurr)", vcurr_dev, vcurr_pitch, rows, cols, 16)) != 0)
    return rc;
  if ((rc = check_image("icp_accumulate(ncurr)", ncurr_dev, ncurr_pitch, rows, cols, 16)) != 0)
    return rc;
  if ((rc = check_image("icp_accumulate(vprev)", vprev_dev, vprev_pitch, rows, cols, 16)) != 0)
    return rc;
  if ((rc = check_image("icp_accumulate(nprev)", nprev_dev, nprev_pitch, rows, cols, 16)) != 0)
    return rc;
  return launch_icp_accumulate(ctx, R, t, intr, level, dist_thres, angle_thres, (const uint8_t*)vcurr_dev, vcurr_pitch,
                               (const uint8_t*)ncurr_dev, ncurr_pitch, (const uint8_t*)vprev_dev, vprev_pitch,
                               (const uint8_t*)nprev_dev, nprev_pitch, rows, cols, sums_host, counts_host, 0, rows,
                               nullptr);
}

int yakhip_icp_accumulate_rows(yakhip_ctx* ctx, const float R[9], const float t[3], const float intr[4], int level,
                               float dist_thres, float angle_thres, const void* vcurr_dev, size_t vcurr_pitch,
                               const void* ncurr_dev, size_t ncurr_pitch, const void* vprev_dev, size_t vprev_pitch,
                               const void* nprev_dev, size_t nprev_pitch, int rows, int cols, int row_begin,
                               int row_end, float* sums_dev)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!R || !t || !intr || !sums_dev)
    return fail(YAKHIP_EINVAL, "icp_accumulate_rows: null pose, intrinsics or output");
  if (((uintptr_t)sums_dev & 3) != 0)
    return fail(YAKHIP_EALIGN, "icp_accumulate_rows: output not 4-byte aligned");
  if (level < 0 || level > 15)
    return fail(YAKHIP_EINVAL, "icp_accumulate_rows: pyramid level %d outside [0, 15]", level);
  if ((rc = check_image("icp_accumulate_rows(vcurr)", vcurr_dev, vcurr_pitch, rows, cols, 16)) != 0)
    return rc;
  if ((rc = check_image("icp_accumulate_rows(ncurr)", ncurr_dev, ncurr_pitch, rows, cols, 16)) != 0)
    return rc;
  if ((rc = check_image("icp_accumulate_rows(vprev)", vprev_dev, vprev_pitch, rows, cols, 16)) != 0)
    return rc;
  if ((rc = check_image("icp_accumulate_rows(nprev)", nprev_dev, nprev_pitch, rows, cols, 16)) != 0)
    return rc;
  if (!(0 <= row_begin && row_begin <= row_end && row_end <= rows))
    return fail(YAKHIP_EINVAL, "icp_accumulate_rows: rows [%d, %d) outside the image's %d", row_begin, row_end, rows);
  return launch_icp_accumulate(ctx, R, t, intr, level, dist_thres, angle_thres, (const uint8_t*)vcurr_dev, vcurr_pitch,
                               (const uint8_t*)ncurr_dev, ncurr_pitch, (const uint8_t*)vprev_dev, vprev_pitch,
                               (const uint8_t*)nprev_dev, nprev_pitch, rows, cols, nullptr, nullptr, row_begin, row_end,
                               sums_dev);
}

// One linearisation's host side, the reference's own control flow (projective_icp.cpp:212-228): unpack the 27
// sums, determinant test, SVD solve, Tinc * corrected.  Returns false where estimateTransform returns false.
static bool host_solve_step(const float sums[27], float Rc[9], float tc[3])
{
  double A[36], b[6], r[6];
  int shift = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 7; ++j)
    {
      const double value = sums[shift++];
      if (j == 6)
        b[i] = value;
      else
        A[j * 6 + i] = A[i * 6 + j] = value;
    }
  const double det = yak::host::det6(A);
  if (fabs(det) < 1e-15 || det != det)
    return false;
  yak::host::solve6_svd_symmetric(A, b, r);
  // Affine3f(rvec, t) and Tinc * corrected, in the arithmetic of include/yak/kfusion/compat/linalg.hpp
  const float rv[3] = { (float)r[0], (float)r[1], (float)r[2] };
  float Ri[9] = { 1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f };
  const double th = sqrt((double)rv[0] * rv[0] + (double)rv[1] * rv[1] + (double)rv[2] * rv[2]);
  if (th > 2.220446049250313e-16)
  {
    const double kx = rv[0] / th, ky = rv[1] / th, kz = rv[2] / th, c = cos(th), s = sin(th), v = 1.0 - c;
    Ri[0] = (float)(c + kx * kx * v), Ri[1] = (float)(kx * ky * v - kz * s), Ri[2] = (float)(kx * kz * v + ky * s);
    Ri[3] = (float)(ky * kx * v + kz * s), Ri[4] = (float)(c + ky * ky * v), Ri[5] = (float)(ky * kz * v - kx * s);
    Ri[6] = (float)(kz * kx * v - ky * s), Ri[7] = (float)(kz * ky * v + kx * s), Ri[8] = (float)(c + kz * kz * v);
  }
  float Rn[9], tn[3];
  for (int rr = 0; rr < 3; ++rr)
  {
    for (int cc = 0; cc < 3; ++cc)
      Rn[3 * rr + cc] = Ri[3 * rr + 0] * Rc[0 + cc] + Ri[3 * rr + 1] * Rc[3 + cc] + Ri[3 * rr + 2] * Rc[6 + cc];
    tn[rr] = (Ri[3 * rr + 0] * tc[0] + Ri[3 * rr + 1] * tc[1] + Ri[3 * rr + 2] * tc[2]) + (float)r[3 + rr];
  }
  memcpy(Rc, Rn, 9 * sizeof(float));
  memcpy(tc, tn, 3 * sizeof(float));
  return true;
}

int yakhip_icp_estimate(yakhip_ctx* ctx, const float R[9], const float t[3], const float intr[4], float dist_thres,
                        float angle_thres, const yakhip_icp_level* levels, int n_levels, float R_out[9], float t_out[3],
                        int* ok)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!R || !t || !intr || !levels || !R_out || !t_out || !ok)
    return fail(YAKHIP_EINVAL, "icp_estimate: null argument");
  if (n_levels < 1 || n_levels > 16)
    return fail(YAKHIP_EINVAL, "icp_estimate: %d pyramid levels outside [1, 16]", n_levels);
  for (int l = 0; l < n_levels; ++l)
  {
    const yakhip_icp_level& L = levels[l];
    if (L.iterations < 0)
      return fail(YAKHIP_EINVAL, "icp_estimate: negative iteration count on level %d", l);
    if ((rc = check_image("icp_estimate(vcurr)", L.vcurr, L.vcurr_pitch, L.rows, L.cols, 16)) != 0 ||
        (rc = check_image("icp_estimate(ncurr)", L.ncurr, L.ncurr_pitch, L.rows, L.cols, 16)) != 0 ||
        (rc = check_image("icp_estimate(vprev)", L.vprev, L.vprev_pitch, L.rows, L.cols, 16)) != 0 ||
        (rc = check_image("icp_estimate(nprev)", L.nprev, L.nprev_pitch, L.rows, L.cols, 16)) != 0)
      return rc;
  }
  *ok = 1;
  int status = 0, iterations = 0;
  if ((rc = launch_icp_estimate(ctx, R, t, intr, dist_thres, angle_thres, levels, n_levels, R_out, t_out, &status,
                                &iterations)) != 0)
    return rc;
  if (status == 0)
    return 0;

  // Slow path, the reference's own control flow (projective_icp.cpp:195-231): one blocking accumulation
  // per linearisation, determinant test, SVD solve and composition on the host.
  float Rc[9], tc[3];
  memcpy(Rc, R, sizeof(Rc));
  memcpy(tc, t, sizeof(tc));
  for (int l = n_levels - 1; l >= 0; --l)
  {
    const yakhip_icp_level& L = levels[l];
    for (int it = 0; it < L.iterations; ++it)
    {
      float sums[27];
      if ((rc = launch_icp_accumulate(ctx, Rc, tc, intr, l, dist_thres, angle_thres, (const uint8_t*)L.vcurr,
                                      L.vcurr_pitch, (const uint8_t*)L.ncurr, L.ncurr_pitch, (const uint8_t*)L.vprev,
                                      L.vprev_pitch, (const uint8_t*)L.nprev, L.nprev_pitch, L.rows, L.cols, sums,
                                      nullptr, 0, L.rows, nullptr)) != 0)
        return rc;
      if (!host_solve_step(sums, Rc, tc))
      {
        *ok = 0;
        memcpy(R_out, Rc, sizeof(Rc));
        memcpy(t_out, tc, sizeof(tc));
        return 0;
      }
    }
  }
  memcpy(R_out, Rc, sizeof(Rc));
  memcpy(t_out, tc, sizeof(tc));
  return 0;
}


static int check_icp_levels(const char* who, const yakhip_icp_level* levels, int n_levels)
{
  int rc;
  if (n_levels < 1 || n_levels > 16)
    return fail(YAKHIP_EINVAL, "%s: %d pyramid levels outside [1, 16]", who, n_levels);
  for (int l = 0; l < n_levels; ++l)
  {
    const yakhip_icp_level& L = levels[l];
    if (L.iterations < 0)
      return fail(YAKHIP_EINVAL, "%s: negative iteration count on level %d", who, l);
    if ((rc = check_image(who, L.vcurr, L.vcurr_pitch, L.rows, L.cols, 16)) != 0 ||
        (rc = check_image(who, L.ncurr, L.ncurr_pitch, L.rows, L.cols, 16)) != 0 ||
        (rc = check_image(who, L.vprev, L.vprev_pitch, L.rows, L.cols, 16)) != 0 ||
        (rc = check_image(who, L.nprev, L.nprev_pitch, L.rows, L.cols, 16)) != 0)
      return rc;
  }
  return 0;
}

int yakhip_icp_estimate_sharded(yakhip_ctx* ctx, yakhip_comm* comm, const float R[9], const float t[3],
                                const float intr[4], float dist_thres, float angle_thres, const yakhip_icp_level* levels,
                                int n_levels, float R_out[9], float t_out[3], int* ok)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!comm || !R || !t || !intr || !levels || !R_out || !t_out || !ok)
    return fail(YAKHIP_EINVAL, "icp_estimate_sharded: null argument");
  if ((rc = check_icp_levels("icp_estimate_sharded", levels, n_levels)) != 0)
    return rc;
  *ok = 1;
  int status = 0, iterations = 0;
  if ((rc = launch_icp_estimate_sharded(ctx, comm, R, t, intr, dist_thres, angle_thres, levels, n_levels, R_out, t_out,
                                        &status, &iterations)) != 0)
    return rc;
  if (status == 0)
    return 0;
  // Slow path, taken by every rank alike (they hold the same state): per linearisation the band's 33 values,
  // their all-reduce, a wait, and the reference's host-side determinant test and SVD solve.
  int rank = 0, world = 1;
  yakhip_comm_info(comm, &rank, &world);
  if (!ctx->icp_sums)  // (kept in the context: this path runs once per frame for as long as a frame needs it)
    YAKHIP_TRY(hipMalloc((void**)&ctx->icp_sums, 64 * sizeof(float)));
  float* sums_dev = ctx->icp_sums;
  float Rc[9], tc[3], host[33];
  memcpy(Rc, R, sizeof(Rc));
  memcpy(tc, t, sizeof(tc));
  for (int l = n_levels - 1; l >= 0 && rc == 0; --l)
  {
    const yakhip_icp_level& L = levels[l];
    const int r0 = (int)((long long)L.rows * rank / world), r1 = (int)((long long)L.rows * (rank + 1) / world);
    for (int it = 0; it < L.iterations && rc == 0; ++it)
    {
      if ((rc = launch_icp_accumulate(ctx, Rc, tc, intr, l, dist_thres, angle_thres, (const uint8_t*)L.vcurr, L.vcurr_pitch,
                                      (const uint8_t*)L.ncurr, L.ncurr_pitch, (const uint8_t*)L.vprev, L.vprev_pitch,
                                      (const uint8_t*)L.nprev, L.nprev_pitch, L.rows, L.cols, nullptr, nullptr, r0, r1,
                                      sums_dev)) != 0 ||
          (rc = comm_allreduce(ctx, comm, sums_dev, 33, YAKHIP_SUM_F32)) != 0 ||
          (rc = check(hipMemcpyAsync(host, sums_dev, sizeof(host), hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync")) != 0 ||
          (rc = check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize")) != 0)
        break;
      if (!host_solve_step(host, Rc, tc))
      {
        *ok = 0;
        l = -1;  // leave both loops with the estimate reached before the failing linearisation
        break;
      }
    }
  }
  memcpy(R_out, Rc, sizeof(Rc));
  memcpy(t_out, tc, sizeof(tc));
  return rc;
}

int yakhip_raycast_sharded(yakhip_ctx* ctx, yakhip_comm* comm, const yakhip_volume* vol, int z_own0, int z_own1,
                           const float R[9], const float t[3], const float Rinv[9], const float intr[4],
                           float step_factor, float gradient_delta_factor, void* points_dev, size_t points_pitch,
                           void* depth_dev, size_t depth_pitch, void* normals_dev, size_t normals_pitch, void* keys_dev,
                           size_t keys_pitch, int rows, int cols)
{
  if (!comm)
    return fail(YAKHIP_EINVAL, "raycast_sharded: null communicator");
  if (!keys_dev)
    return fail(YAKHIP_EINVAL, "raycast_sharded: null keys image");
  if (rows < 1 || cols < 1 || (long long)rows * cols > 0x7fffffffll / 32)
    return fail(YAKHIP_ERANGE, "raycast_sharded: image of %d x %d pixels", cols, rows);
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  // the slab's ray cast leaves, beside its images, every pixel's key and march time in the communicator's scratch:
  // that is what the composite exchanges (comm.hip)
  int32_t* keys_flat = nullptr;
  float* ts_flat = nullptr;
  if ((rc = composite_buffers(ctx, comm, rows * cols, ctx->composite_capacity, &keys_flat, &ts_flat)) != 0)
    return rc;
  RaycastParams rp;
  if ((rc = raycast_common(ctx, vol, R, t, Rinv, intr, step_factor, gradient_delta_factor, points_dev, points_pitch, depth_dev,
                           depth_pitch, normals_dev, normals_pitch, rows, cols, keys_dev, keys_pitch, z_own0, z_own1, nullptr,
                           keys_flat, ts_flat, &rp)) != 0)
    return rc;
  CompositeImages im;
  im.points = (uint8_t*)points_dev, im.normals = (uint8_t*)normals_dev, im.depth = (uint8_t*)depth_dev;
  im.keys = (uint8_t*)keys_dev;
  im.points_pitch = (uint32_t)points_pitch, im.normals_pitch = (uint32_t)normals_pitch;
  im.depth_pitch = (uint32_t)depth_pitch, im.keys_pitch = (uint32_t)keys_pitch;
  im.rows = rows, im.cols = cols;
  if ((rc = composite_first(ctx, comm, rp, ctx->composite_capacity)) != 0)
    return rc;
  if ((rc = yakhip_raycast_resolve(ctx, vol, z_own0, z_own1, Rinv, gradient_delta_factor, points_dev, points_pitch,
                                   depth_dev, depth_pitch, normals_dev, normals_pitch, keys_dev, keys_pitch, rows, cols)) != 0)
    return rc;
  return composite_second(ctx, comm, im, ctx->composite_capacity);
}

int yakhip_comm_status(yakhip_ctx* ctx, yakhip_comm* comm, unsigned status[2])
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!comm || !status)
    return fail(YAKHIP_EINVAL, "comm_status: null argument");
  return composite_status(ctx, comm, status);
}

// ---- surface extraction ------------------------------------------------------------------------

int yakhip_marching_cubes(yakhip_ctx* ctx, const yakhip_volume* vol, int min_weight, double scale, void* triangles_dev,
                          unsigned long long capacity_triangles, unsigned long long* n_triangles_host)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if ((rc = check_volume(vol)) != 0)
    return rc;
  if (!n_triangles_host)
    return fail(YAKHIP_EINVAL, "marching_cubes: null count pointer");
  if (triangles_dev && (uintptr_t)triangles_dev % 4 != 0)
    return fail(YAKHIP_EALIGN, "marching_cubes: triangle buffer must be 4-byte aligned");
  return launch_marching_cubes(ctx, (const uint32_t*)vol->data, vol->dims[0], vol->dims[1], vol->z0, vol->z1 - vol->z0,
                               min_weight, scale, (float*)triangles_dev, triangles_dev ? capacity_triangles : 0,
                               n_triangles_host, (const uint32_t*)vol->surface_map, 0);
}

int yakhip_marching_cubes_layers(yakhip_ctx* ctx, const yakhip_volume* vol, int z_first, int z_last, int min_weight,
                                 double scale, void* triangles_dev, unsigned long long capacity_triangles,
                                 unsigned long long* n_triangles_host)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if ((rc = check_volume(vol)) != 0)
    return rc;
  if (!n_triangles_host)
    return fail(YAKHIP_EINVAL, "marching_cubes_layers: null count pointer");
  if (triangles_dev && (uintptr_t)triangles_dev % 4 != 0)
    return fail(YAKHIP_EALIGN, "marching_cubes_layers: triangle buffer must be 4-byte aligned");
  if (!(vol->z0 <= z_first && z_first < z_last && z_last <= vol->z1))
    return fail(YAKHIP_EINVAL, "marching_cubes_layers: layers [%d, %d) are not part of the slab [%d, %d)", z_first, z_last,
                vol->z0, vol->z1);
  const size_t plane = (size_t)vol->dims[0] * (size_t)vol->dims[1];
  return launch_marching_cubes(ctx, (const uint32_t*)vol->data + plane * (size_t)(z_first - vol->z0), vol->dims[0],
                               vol->dims[1], z_first, z_last - z_first, min_weight, scale, (float*)triangles_dev,
                               triangles_dev ? capacity_triangles : 0, n_triangles_host, (const uint32_t*)vol->surface_map,
                               z_first - vol->z0);
}

int yakhip_weld_vertices(yakhip_ctx* ctx, const void* triangles_dev, unsigned long long n_triangles, void* vertices_dev,
                         void* index_dev, unsigned long long* n_vertices_host)
{
  int rc = enter(ctx);
  if (rc != 0)
    return rc;
  if (!n_vertices_host)
    return fail(YAKHIP_EINVAL, "weld_vertices: null count pointer");
  *n_vertices_host = 0;
  if (n_triangles == 0)
    return 0;
  if (!triangles_dev || !vertices_dev || !index_dev)
    return fail(YAKHIP_EINVAL, "weld_vertices: null buffer");
  if ((uintptr_t)triangles_dev % 4 != 0 || (uintptr_t)vertices_dev % 4 != 0 || (uintptr_t)index_dev % 4 != 0)
    return fail(YAKHIP_EALIGN, "weld_vertices: buffers must be 4-byte aligned");
  if (n_triangles > 0x3fffffffull / 3ull)
    return fail(YAKHIP_ERANGE, "weld_vertices: %llu triangles, at most %llu", n_triangles, 0x7fffffffull / 3ull);
  return launch_weld(ctx, (const uint32_t*)triangles_dev, (uint32_t)(n_triangles * 3ull), (uint32_t*)vertices_dev,
                     (uint32_t*)index_dev, n_vertices_host);
}

}  // extern "C"
