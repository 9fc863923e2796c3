/*
 * yakhip.h -- C ABI of the MI355X (gfx950) TSDF fusion engine for Yak.
 *
 * This is the drop-in seam: each entry point replaces one free function of the
 * reference's device layer `kfusion::device::*` (yak/include/yak/kfusion/internal.hpp:124-193)
 * or one method of its device-memory layer (yak/src/kfusion/device_memory.cpp).  The host
 * classes of the reference (kfusion::cuda::TsdfVolume, KinFu, yak::FusionServer) already
 * reduce everything to the plain pointers, dims and 3x4 pose floats that cross this boundary.
 *
 * Conventions
 *   - plain C: pointers, sizes, POD structs.  No C++ types, no exceptions, no process globals.
 *   - every call returns 0 on success, a negative YAKHIP_E* for a rejected argument, or a
 *     positive hipError_t.  yakhip_last_error() gives the text (thread-local).  Nothing here
 *     ever calls exit() (the reference's cudaSafeCall does: device_memory.cpp:7-11).
 *   - all device work is enqueued on the context's stream and returns immediately; call
 *     yakhip_ctx_synchronize() where the reference had cudaDeviceSynchronize()
 *     (tsdf_volume.cu:139, imgproc.cpp:21).
 *   - one context = one GPU = one caller thread at a time (same contract as yak_server.h:22).
 *   - R is a row-major 3x3 rotation, t a translation, exactly the `device::Aff3f` the
 *     reference builds with device_cast (yak/include/yak/kfusion/precomp.hpp:15-30).
 *   - intr = {fx, fy, cx, cy} (kfusion::Intr, yak/include/yak/kfusion/types.hpp:18-25).
 *
 * All file:line citations are relative to /root/reference/.
 */
#ifndef YAKHIP_H_
#define YAKHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YAKHIP_OK 0
#define YAKHIP_EINVAL (-1)   /* bad argument (null pointer, non-positive size, bad range) */
#define YAKHIP_EALIGN (-2)   /* pointer / pitch not aligned as required */
#define YAKHIP_ENOGPU (-3)   /* no usable HIP device */
#define YAKHIP_ERANGE (-4)   /* value outside what the kernels support (e.g. max_weight > 32767) */
#define YAKHIP_EABORTED (-5) /* the communicator was aborted, or a rank of its group left (yakhip_comm_abort) */

typedef struct yakhip_ctx yakhip_ctx;

/*
 * A z-slab view of a TSDF volume: replaces `kfusion::device::TsdfVolume`
 * (yak/include/yak/kfusion/internal.hpp:42-67).
 *
 * Voxel = 4 bytes: bytes 0-1 IEEE half tsdf, bytes 2-3 int16 weight
 * (yak/include/yak/kfusion/cuda/device.hpp:68-81; yak/include/yak/kfusion/tsdf_container.h:45-49).
 * Layout x fastest, z slowest: offset(x,y,z) = x + y*dims[0] + z*dims[0]*dims[1] (device.hpp:21-24).
 *
 * `data` points at layer z0; the view owns layers [z0, z1) of a volume whose full
 * extent is dims[].  A single-GPU volume is the slab z0 = 0, z1 = dims[2].  A slab
 * is one contiguous byte range, so concatenating slabs in z order is the whole volume.
 */
typedef struct yakhip_volume
{
  void* data;           /* device pointer, 4-byte aligned (16-byte aligned enables the wide path) */
  int dims[3];          /* full volume extent in voxels: X, Y, Z */
  int z0, z1;           /* layers held by `data`: 0 <= z0 < z1 <= dims[2] */
  float voxel_size[3];  /* metres per voxel (TsdfVolume::getVoxelSize, yak/src/kfusion/tsdf_volume.cpp:45-48) */
  float trunc_dist;     /* metres (tsdf_volume.cpp:62-67) */
  int max_weight;       /* 1..32767 */
  void* surface_map;    /* NULL, or device memory of yakhip_surface_map_bytes() bytes (4-byte aligned): see below */
} yakhip_volume;

/* The surface map of a slab (no counterpart in the reference; optional): one bit per brick of
 * YAKHIP_SURFACE_BRICK^3 voxels of the slab's layers [z0, z1) -- bit (bx + BX * (by + BY * bz)) of an array of
 * 32-bit words, BX = ceil(X / 16), BY = ceil(Y / 16), bz counted from layer z0 -- set when the brick MAY hold a voxel
 * with a negative tsdf; then, from the next word on, one bit per cell of 4 x 4 x 4 bricks in the same order, set
 * when a marked brick lies in the cell or touches it from a neighbouring cell (the ray cast crosses the free space in
 * front of the first marked cell without looking at its samples).  The ray cast can only stop between two samples of which one is negative
 * (yak/src/cuda/tsdf_volume.cu:326-331), so it does not fetch the samples that fall into bricks whose bit is
 * clear; the images are the same bit for bit, with or without a map.
 *   yakhip_integrate / _static / _batch   set the bits of the bricks in which a negative measurement was folded
 *                                         into a voxel (nothing else can make a voxel negative);
 *   yakhip_clear_volume                   clears the map with the voxels;
 *   yakhip_raycast_* (and _slab/_sharded), yakhip_marching_cubes[_layers] and yakhip_extract_cloud consult it;
 *   yakhip_surface_map_rebuild            recomputes it from the voxels.
 * The map belongs to one (data, z0, z1): pass the same three with it every time.  Whoever writes voxels by other
 * means (an upload, a copy, a kernel of his own) calls yakhip_surface_map_rebuild before the next ray cast, or
 * passes NULL.  A stale SET bit is harmless. */
#define YAKHIP_SURFACE_BRICK 16
int yakhip_surface_map_bytes(const yakhip_volume* vol, size_t* bytes);

/* ---- context ------------------------------------------------------------------------------ */

/* Replaces kfusion::cuda::setDevice (yak/src/kfusion/core.cpp:23) + implicit default-stream use.
 * Creates a non-blocking stream owned by the context. */
int yakhip_ctx_create(yakhip_ctx** out, int device);
int yakhip_ctx_destroy(yakhip_ctx* ctx);
/* Borrow a caller-owned hipStream_t (e.g. the host framework's current stream).  NULL is HIP's
 * default (null) stream, which is what the reference launches on. */
int yakhip_ctx_set_stream(yakhip_ctx* ctx, void* hip_stream);
/* Go back to the context's own non-blocking stream. */
int yakhip_ctx_use_own_stream(yakhip_ctx* ctx);
/* The hipStream_t calls are currently enqueued on (the context's own, a borrowed one, or the side stream below):
 * for callers that record their own events or launch their own kernels in order with the library's. */
int yakhip_ctx_stream(yakhip_ctx* ctx, void** hip_stream);
/* A second stream of the context (no counterpart in the reference, which launches everything on the default
 * stream).  Calls made between yakhip_ctx_side_begin and yakhip_ctx_side_end are enqueued on it: they may run
 * at the same time as what the context's stream was given since the PREVIOUS yakhip_ctx_side_end (they wait for
 * everything before that), and everything enqueued AFTER yakhip_ctx_side_end runs after both.  For work that does
 * not touch what those calls read or write -- a fusion loop prepares frame f + 1 (ray distances, bilateral
 * filter, pyramids, point / normal maps) while the integrate and ray cast of frame f are still running; buffers
 * written there and read by frame f's main-stream work (the ray-distance image) alternate between two.
 * Blocking calls made in between wait for the side stream only.  A second yakhip_ctx_side_begin inside a
 * section does nothing: whoever calls yakhip_ctx_side_end next ends it (a server uploads the frame in the section
 * that its tracker then continues and ends). */
int yakhip_ctx_side_begin(yakhip_ctx* ctx);
int yakhip_ctx_side_end(yakhip_ctx* ctx);
/* Replaces cuda::waitAllDefaultStream (yak/src/kfusion/imgproc.cpp:21) and the
 * cudaDeviceSynchronize at yak/src/cuda/tsdf_volume.cu:139. */
int yakhip_ctx_synchronize(yakhip_ctx* ctx);
/* Replaces getCudaEnabledDeviceCount / getDeviceName (yak/src/kfusion/core.cpp:8-21, 25-32). */
int yakhip_device_count(int* count);
int yakhip_device_name(int device, char* buf, size_t buflen);
const char* yakhip_last_error(void);
/* Engine knobs that are not part of the reference's API (results never depend on them).  Keys:
 * "z_chunk" (layers walked by one workgroup, 0 = auto), "nontemporal" (0/1), "vx" (cap on voxels per
 * lane: 0 = widest, 1, 2, 4), "group_test" (0/1: coarse culling test of whole voxel runs), "brick_test" (0/1:
 * whole workgroup bricks that cannot update, or update everywhere with tsdf 1, are settled before they are walked),
 * "checkpoint" (0/1: slabs with z0 >= 64 start from column positions computed in closed form instead of
 * replaying z0 additions per column), "tile" (0 or 3..6: wave footprint in the layer), "tile_side" (0/1: a workgroup's waves stacked in y / side by side in x), "mc_z_chunk"
 * (marching-cubes tile depth, 0 = auto), "kernel_timing" (0/1, see yakhip_ctx_kernel_time), "chunk_order" (-1 auto, 0 z-chunks dispatched
 * shallow to deep, 1 shallow / deep in turn, 3 deepest first), "chunk_min" / "chunk_ramp" (graded z-chunks of a volume walked from
 * layer 0: a chunk that starts at layer zs is about zs / ramp layers long, at least chunk_min; 0 = auto, ramp -1 = all alike),
 * "icp_poll" (0/1: yakhip_icp_estimate polls pinned memory for its result / copies and
 * synchronises), "surface_skip" (0/1: the ray cast consults / ignores yakhip_volume.surface_map),
 * "raycast_segments" (waves per ray where the surface map is consulted: 1 (= 0, the default) one wave marches the whole ray, 2 or 4 cut
 * its sample pairs into segments marched side by side, the first pair that ends a march wins -- same images, measured slower: DESIGN 3.6;
 * also YAK_HIP_RAYCAST_SEGMENTS in the environment),
 * "brick_stats" (0/1, see yakhip_ctx_brick_stats), "composite_capacity" (list slots of yakhip_raycast_sharded's second step, 0 = auto),
 * "defer_mixed" (how yakhip_integrate launches: 0 one kernel, every workgroup walking its own bricks; 2 two kernels wherever the
 * 16-byte layout allows -- the brick test for the whole grid first, then the live bricks' parts dealt out evenly to a fixed number
 * of workgroups; 1, the default, one or the other by what the brick test found in the last frames of this volume, see
 * yakhip_ctx_launch_counts), "mixed_waves" (workgroups per CU of the second kernel, 0 = auto), "refine" (0/1: the second kernel
 * tests a part again layer by layer before walking it),
 * "zu" (ignored).  The environment variable YAK_HIP_DEFER_MIXED (0/1/2) sets "defer_mixed" for every new context.
 * Unknown key or value -> YAKHIP_EINVAL. */
int yakhip_ctx_set_option(yakhip_ctx* ctx, const char* key, int value);
/* Measurement aid with no counterpart in the reference (its only timers are the never-invoked
 * ScopeTime/SampledScopeTime, yak/src/kfusion/core.cpp:236-260).  After
 * yakhip_ctx_set_option(ctx, "kernel_timing", 1) every integrate / raycast / marching-cubes kernel launch is bracketed by a
 * hipEvent pair on the context's stream; this call waits for them, returns the summed device time of
 * those launches (milliseconds) and their count, and restarts the accumulation. */
int yakhip_ctx_kernel_time(yakhip_ctx* ctx, double* total_ms, int* launches);
/* The same accumulation launch by launch (also without counterpart in the reference): ms_out[i] = device time
 * of the i-th bracketed launch, start_ms_out[i] (may be NULL) = time from the first bracketed launch's start to
 * this one's start, for the first min(*launches, capacity) launches; *launches = how many were bracketed.
 * Restarts the accumulation.  bench.py prints this series so that slow launches can be told from the mean. */
int yakhip_ctx_kernel_times(yakhip_ctx* ctx, float* ms_out, float* start_ms_out, int capacity, int* launches);
/* Measurement aid, no counterpart in the reference: after yakhip_ctx_set_option(ctx, "brick_stats", 1) the
 * single-frame integrate kernel counts how its whole-brick test classified the wave bricks it looked at;
 * this call waits, returns counts = {dead (no voxel can update: skipped), front (every voxel updates with
 * tsdf 1: streamed), mixed (walked voxel by voxel)} since the last call and restarts the count. */
int yakhip_ctx_brick_stats(yakhip_ctx* ctx, unsigned long long counts[3]);
/* Measurement aid, no counterpart in the reference: counts = {yakhip_integrate launches of this context that ran as one
 * kernel, ... as two kernels} since the context was created ("defer_mixed" above; the voxels are the same either way). */
int yakhip_ctx_launch_counts(yakhip_ctx* ctx, unsigned long long counts[2]);
/* Device self-test with no counterpart in the reference: the integrate kernel replaces IEEE '/' by a
 * v_rcp_f32 + Newton + Markstein sequence; this checks on the GPU at hand that (a) the refined
 * reciprocal is the correctly rounded 1/d for all 2^23 mantissas at 8 exponents and (b) the quotient
 * equals n / d on ~10^9 pseudo-random pairs.  counts = {reciprocals checked, mismatches, quotients
 * checked, mismatches}; both mismatch counts must be 0 for the kernel's bit-exactness argument. */
int yakhip_selftest_division(yakhip_ctx* ctx, unsigned long long counts[4]);

/* ---- device memory: replaces DeviceMemory / DeviceMemory2D
 *      (yak/src/kfusion/device_memory.cpp:85-103 create, :129-138 upload/download,
 *       :196-215 create 2-D pitched, :244-256 upload/download 2-D) ----------------------------- */
int yakhip_malloc(yakhip_ctx* ctx, void** ptr, size_t bytes);
int yakhip_malloc_pitch(yakhip_ctx* ctx, void** ptr, size_t* pitch_bytes, size_t width_bytes, size_t rows);
int yakhip_free(yakhip_ctx* ctx, void* ptr);
int yakhip_upload(yakhip_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int yakhip_download(yakhip_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int yakhip_upload_2d(yakhip_ctx* ctx, void* dst_dev, size_t dst_pitch, const void* src_host, size_t src_pitch,
                     size_t width_bytes, size_t rows);
int yakhip_download_2d(yakhip_ctx* ctx, void* dst_host, size_t dst_pitch, const void* src_dev, size_t src_pitch,
                       size_t width_bytes, size_t rows);

/* Device-to-device, blocking: replaces the cudaMemcpy / cudaMemcpy2D of DeviceMemory[2D]::copyTo
 * (device_memory.cpp:105-114, 233-242).  Ranges must not overlap. */
int yakhip_copy(yakhip_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);
int yakhip_copy_2d(yakhip_ctx* ctx, void* dst_dev, size_t dst_pitch, const void* src_dev, size_t src_pitch,
                   size_t width_bytes, size_t rows);

/* ---- hot path ----------------------------------------------------------------------------- */

/* Replaces kfusion::device::clear_volume (internal.hpp:125; yak/src/cuda/tsdf_volume.cu:14-40):
 * every voxel of the slab <- pack_tsdf(0, 0) = 0x00000000. */
int yakhip_clear_volume(yakhip_ctx* ctx, const yakhip_volume* vol);

/* vol->surface_map <- the bricks of vol's layers that hold a voxel with a negative tsdf (one sweep over the
 * voxels).  Not blocking.  No counterpart in the reference. */
int yakhip_surface_map_rebuild(yakhip_ctx* ctx, const yakhip_volume* vol);

/* Replaces kfusion::device::compute_dists (internal.hpp:150; yak/src/cuda/imgproc.cu:259-283):
 * dists(y,x) = half_rn(depth_mm(y,x) * sqrt(xl^2 + yl^2 + 1) * 0.001), xl = (x - cx)/fx.
 * depth: uint16 millimetres, dists: IEEE half bits; both row-pitched (pitch in bytes, even). */
int yakhip_compute_dists(yakhip_ctx* ctx, const void* depth_dev, size_t depth_pitch, void* dists_dev,
                         size_t dists_pitch, int rows, int cols, const float intr[4]);

/* Replaces kfusion::device::integrate (internal.hpp:126; yak/src/cuda/tsdf_volume.cu:51-140):
 * projective TSDF update of every voxel of the slab from one ray-distance image.
 * R,t = vol2cam (volume frame -> camera frame) as built at yak/src/kfusion/tsdf_volume.cpp:97-104.
 * Bit-identical, voxel for voxel, to the same layers of a whole-volume run (the camera-space
 * position is accumulated from z = 0 exactly as tsdf_volume.cu:76 does). */
int yakhip_integrate(yakhip_ctx* ctx, const yakhip_volume* vol, const void* dists_dev, size_t dists_pitch, int rows,
                     int cols, const float R[9], const float t[3], const float intr[4]);
/* Optional, no counterpart in the reference: the part of yakhip_integrate's work that does not depend on the pose
 * (per-pixel band thresholds and the coarse planes its culling tests read), ahead of time and on whatever stream is
 * current -- e.g. between yakhip_ctx_side_begin / _end, beside the previous frame's work.  The next yakhip_integrate
 * call with the same image (pointer, pitch, size; not written in between -- a yakhip_compute_dists, upload or copy into
 * it through this context discards what was prepared, other writers must not touch it) and truncation distance then launches its
 * walk only; any other integrate call in between discards what was prepared.  Two calls in a row use two scratch
 * areas in turn, so frame f + 1 may be prepared while frame f is still being integrated.  Slabs that start from
 * column checkpoints (z0 >= 64) take nothing from it.  Results are the same bit for bit.  The integrate call must be
 * ordered behind this one like behind any producer of its image: same stream, or after yakhip_ctx_side_end. */
int yakhip_integrate_prepare(yakhip_ctx* ctx, const yakhip_volume* vol, const void* dists_dev, size_t dists_pitch, int rows,
                             int cols);

/* yakhip_integrate for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for code
 * under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.  Same arguments, same
 * kernels and the same voxels, bit for bit, as yakhip_integrate; the per-frame scratch (band thresholds,
 * coarse minima, and the column checkpoints of a slab with z0 >= 64) lives in a file-scope __device__ array of
 * the library instead of context-owned buffers -- about 20 MiB per GPU, held until the process ends, its
 * address taken once in yakhip_ctx_create -- and the call itself performs no hipMalloc / hipFree / event
 * creation (the "kernel_timing" option is ignored here).
 * Limits: rows * cols <= YAKHIP_STATIC_MAX_PIXELS, else YAKHIP_ERANGE; a slab with z0 >= 64 whose plane
 * dims[0] * dims[1] exceeds YAKHIP_STATIC_MAX_COLUMNS replays the reference's additions from layer 0
 * (same result, slower start).  Every call on one GPU shares that GPU's array, in every context: calls that
 * overlap on one GPU (two streams, two threads) race; calls on one stream are ordered and safe. */
#define YAKHIP_STATIC_MAX_PIXELS (1u << 20)
#define YAKHIP_STATIC_MAX_COLUMNS (1u << 20)
int yakhip_integrate_static(yakhip_ctx* ctx, const yakhip_volume* vol, const void* dists_dev, size_t dists_pitch,
                            int rows, int cols, const float R[9], const float t[3], const float intr[4]);

/* One camera frame of a batch: its ray-distance image (as yakhip_compute_dists writes it), vol2cam pose
 * and intrinsics -- the arguments kfusion::device::integrate takes per frame (internal.hpp:126). */
typedef struct yakhip_frame
{
  const void* dists;   /* device pointer, IEEE half bits */
  size_t dists_pitch;  /* bytes */
  int rows, cols;
  float R[9], t[3];    /* vol2cam, row-major R */
  float intr[4];       /* fx, fy, cx, cy */
} yakhip_frame;

/* Multi-camera batched integrate (no single counterpart in the reference, which
 * integrates one frame per call: yak/src/cuda/tsdf_volume.cu:119-140).  Folds n_frames frames into
 * every voxel of the slab in the order given, during one sweep of the volume per group of up to 4
 * frames: each touched voxel is read and written once per group instead of once per frame.  The result
 * is bit-identical to calling yakhip_integrate n_frames times in that order.  All frames of a call
 * must have the same image size.  n_frames >= 1. */
int yakhip_integrate_batch(yakhip_ctx* ctx, const yakhip_volume* vol, const yakhip_frame* frames, int n_frames);

/* ---- ray casting ------------------------------------------------------- */

/* Replaces kfusion::device::raycast, Points overload (internal.hpp:137-144;
 * yak/src/cuda/tsdf_volume.cu:288-352, 411-425): one ray per pixel through the whole volume
 * (vol->z0 must be 0 and vol->z1 = dims[2]); first + -> - zero crossing of the nearest-voxel samples,
 * refined by trilinear interpolation; vertex and normal are stored in the CAMERA frame.
 * R,t = cam2vol and Rinv = its inverse rotation, as built at yak/src/kfusion/tsdf_volume.cpp:133-136.
 * points / normals: rows x cols float4 {x, y, z, 0}, row pitch in bytes (multiple of 16); pixels
 * without a hit hold quiet NaN in all four lanes (tsdf_volume.cu:298).
 * step_factor, gradient_delta_factor: TsdfVolume::raycast_step_factor_ / gradient_delta_factor_
 * (time step = trunc_dist * step_factor, gradient delta = voxel_size * gradient_delta_factor). */
int yakhip_raycast_points(yakhip_ctx* ctx, const yakhip_volume* vol, const float R[9], const float t[3],
                          const float Rinv[9], const float intr[4], float step_factor, float gradient_delta_factor,
                          void* points_dev, size_t points_pitch, void* normals_dev, size_t normals_pitch, int rows,
                          int cols);

/* Replaces kfusion::device::raycast, Depth overload (internal.hpp:128-135;
 * yak/src/cuda/tsdf_volume.cu:220-285, 395-409): as above, but stores depth(y,x) =
 * (uint16)(vertex.z * 1000) millimetres, 0 where there is no hit. */
int yakhip_raycast_depth(yakhip_ctx* ctx, const yakhip_volume* vol, const float R[9], const float t[3],
                         const float Rinv[9], const float intr[4], float step_factor, float gradient_delta_factor,
                         void* depth_dev, size_t depth_pitch, void* normals_dev, size_t normals_pitch, int rows,
                         int cols);

/* yakhip_raycast_points plus the pyramid KinFu builds from its result (kinfu.cpp:294-306: resizePointsNormals level
 * by level, imgproc.cu:348-395), written by the same launch: the half-size maps (rows / 2 x cols / 2) and, unless
 * points2_dev and normals2_dev are both null, the quarter-size maps.  Bit for bit the maps of yakhip_raycast_points
 * followed by yakhip_resize_points_normals per level.  Whole volume only. */
int yakhip_raycast_points_pyramid(yakhip_ctx* ctx, const yakhip_volume* vol, const float R[9], const float t[3],
                                  const float Rinv[9], const float intr[4], float step_factor, float gradient_delta_factor,
                                  void* points_dev, size_t points_pitch, void* normals_dev, size_t normals_pitch, int rows,
                                  int cols, void* points1_dev, size_t points1_pitch, void* normals1_dev,
                                  size_t normals1_pitch, void* points2_dev, size_t points2_pitch, void* normals2_dev,
                                  size_t normals2_pitch);

/* Ray casting over a z-slab of a volume that is spread over several GPUs (option 1).  No
 * counterpart in the reference, which ray-casts one whole volume (tsdf_volume.cu:288-352); the arithmetic
 * is that kernel's.  `vol` describes the layers this GPU HOLDS, [vol->z0, vol->z1): its own layers
 * [z_own0, z_own1) plus a halo of at least yakhip_raycast_halo() layers on either side (fewer only where
 * the volume ends).  Every GPU marches every ray over the same global sample sequence and examines the
 * sample pairs whose first sample falls into one of its own layers.  Outputs, per pixel:
 *   keys (int32 image, pitch multiple of 4): 2 * k + f, k = index of the first pair among the slab's own
 *     that ends the march (zero crossing, or the - to + abort of tsdf_volume.cu:328-329); 0x7fffffff if no
 *     own pair ends the march.  f = 1: the hit's refined point lies outside the held layers (the secant
 *     step of tsdf_volume.cu:338 extrapolates when the trilinear values at the two samples have the same
 *     sign), so its normal is still to be computed -- see yakhip_raycast_resolve;
 *   points / depth / normals: as yakhip_raycast_points / _depth write them for that pair (NaN resp. 0
 *     for an abort or no event); for f = 1 the vertex / depth is final unless the normal turns out NaN,
 *     and the normals image holds the refined point in VOLUME coordinates.  points_dev or depth_dev may
 *     be NULL.
 * Composite: the pixel's result is the payload of the slab with the smallest key (keys of different
 * slabs never tie).  Then every slab calls yakhip_raycast_resolve on the composited images and the
 * pixels it changed (its key went from odd to even) are composited once more.  The result equals the
 * whole-volume raycast bit for bit.  yak_amd/shard.py:composite_raycast does both steps with
 * all-reduces. */
int yakhip_raycast_slab(yakhip_ctx* ctx, const yakhip_volume* vol, int z_own0, int z_own1, const float R[9],
                        const float t[3], const float Rinv[9], const float intr[4], float step_factor,
                        float gradient_delta_factor, void* points_dev, size_t points_pitch, void* depth_dev,
                        size_t depth_pitch, void* normals_dev, size_t normals_pitch, void* keys_dev, size_t keys_pitch,
                        int rows, int cols);
/* Second step of the slab composite.  points / depth / normals / keys: the COMPOSITED images (the same on
 * every GPU).  For every pixel with an odd key whose refined point (read from the normals image) lies
 * in one of this slab's own layers: compute the normal there (tsdf_volume.cu:354-372), write it -- or,
 * if it is NaN, discard the hit as tsdf_volume.cu:340-348 does: NaN vertex and normal, depth 0 -- and
 * clear the key's low bit.  Other pixels are left alone.  Needs ceil(gradient delta / voxel_z) + 2 halo
 * layers, which yakhip_raycast_slab's halo covers.  Not blocking. */
int yakhip_raycast_resolve(yakhip_ctx* ctx, const yakhip_volume* vol, int z_own0, int z_own1, const float Rinv[9],
                           float gradient_delta_factor, void* points_dev, size_t points_pitch, void* depth_dev,
                           size_t depth_pitch, void* normals_dev, size_t normals_pitch, void* keys_dev, size_t keys_pitch,
                           int rows, int cols);
/* Halo layers yakhip_raycast_slab needs: ceil(step / voxel_z) + ceil(gradient delta / voxel_z) + 2.
 * vol->data, z0, z1 are not looked at beyond the usual validity checks. */
int yakhip_raycast_halo(const yakhip_volume* vol, float step_factor, float gradient_delta_factor, int* layers);

/* ---- the exchange steps of a z-slab sharded volume ----------------------------------
 * No counterpart in the reference, which has one GPU and no communication.  A communicator
 * joins `world` ranks, one per context: GPUs of one node as processes or as threads of one process.
 * integrate needs none of this (every rank integrates its own layers of the same frame); the tracking side
 * has two exchange steps, both enqueued on the context's stream without a host wait when the transport is RCCL:
 * the composite of the per-slab ray casts and the sum of the per-rank ICP normal equations. */
typedef struct yakhip_comm yakhip_comm;
#define YAKHIP_SUM_F32 0
#define YAKHIP_MIN_I32 1
#define YAKHIP_SUM_I32 2
#define YAKHIP_UNIQUE_ID_BYTES 128
/* RCCL transport (ncclAllReduce over xGMI on the context's stream).  One rank calls yakhip_comm_unique_id and
 * hands the 128 bytes to the others (the caller's business: a pipe, a file, MPI, torch.distributed); then every
 * rank calls yakhip_comm_create_rccl with its own context -- concurrently, it returns when all have joined.
 * librccl is opened on first use; YAKHIP_ENOGPU if it cannot be.  RCCL accepts one rank per GPU. */
int yakhip_comm_unique_id(void* id_out, size_t bytes);
int yakhip_comm_create_rccl(yakhip_ctx* ctx, yakhip_comm** out, const void* unique_id, int world, int rank);
/* Ranks that are threads of ONE process, on the same GPU or on different ones: the values meet in host memory
 * behind a barrier (blocking; every rank must make the same calls in the same order, each from its own
 * thread).  Fills comms_out[0 .. world).  For tests on a one-GPU box and as a fallback transport. */
int yakhip_comm_create_local(int world, yakhip_comm** comms_out);
/* A caller-supplied all-reduce (over MPI, torch.distributed, ...): fn(user, buf_dev, count, op, hip_stream) must
 * leave the reduction over all ranks of the `count` 4-byte values at buf_dev (op = YAKHIP_SUM_F32 / MIN_I32 /
 * SUM_I32) visible to work enqueued later on hip_stream, with the SAME bits on every rank; returns 0 or an
 * error code (passed through). */
typedef int (*yakhip_allreduce_fn)(void* user, void* buf_dev, size_t count, int op, void* hip_stream);
int yakhip_comm_create_custom(yakhip_comm** out, int world, int rank, yakhip_allreduce_fn fn, void* user);
int yakhip_comm_destroy(yakhip_comm* comm);
/* The failure path (no counterpart in the reference: one GPU, no peers).  A rank that cannot go on aborts its
 * communicator instead of leaving its peers waiting: local transport -- every rank waiting in an exchange of the
 * group returns YAKHIP_EABORTED; RCCL -- ncclCommAbort: the collectives of THIS communicator that are in flight stop
 * waiting for their peers and its stream drains (results undefined).  May be called from any thread, also while
 * another thread is inside a collective on the communicator (that is the point: the thread that dispatched the
 * frame aborts the communicators of the ranks still blocked on the one that failed).  Afterwards every collective on
 * the communicator returns YAKHIP_EABORTED; yakhip_comm_destroy is the only call left.  Idempotent. */
int yakhip_comm_abort(yakhip_comm* comm);
int yakhip_comm_info(const yakhip_comm* comm, int* rank, int* world);
/* In-place all-reduce of `count` 4-byte values in device memory, ordered on the context's stream. */
int yakhip_comm_allreduce(yakhip_ctx* ctx, yakhip_comm* comm, void* buf_dev, size_t count, int op);

/* Ray cast of a sharded volume in one call: yakhip_raycast_slab on this rank's layers, the composite over the
 * ranks (all-reduce MIN over the keys, all-reduce SUM over the payload bits of the winners),
 * yakhip_raycast_resolve, and the composite of the few pixels that call settled (one more all-reduce over a list
 * of fixed capacity).  Arguments as yakhip_raycast_slab; on return (stream-ordered, not waited for) points /
 * depth / normals hold, on EVERY rank, what yakhip_raycast_points / _depth give on the whole volume, bit for
 * bit, and keys the composited keys.  Per call: 3 collectives -- 4 B per pixel (the keys), then 16 B per pixel (the
 * winner's march time and normal: every rank rebuilds the vertex and the depth from the time with the ray cast's own
 * arithmetic; 28 B until round 3), then 16 B per list slot (capacity max(4096, pixels / 64); yakhip_comm_status
 * reports a frame that needed more).  At 640 x 480: 1.2 MB + 4.9 MB + 77 KB per rank and frame.
 * The communicator owns the scratch (about 32 B per pixel), allocated on first use. */
int yakhip_raycast_sharded(yakhip_ctx* ctx, yakhip_comm* comm, const yakhip_volume* vol, int z_own0, int z_own1,
                           const float R[9], const float t[3], const float Rinv[9], const float intr[4],
                           float step_factor, float gradient_delta_factor, void* points_dev, size_t points_pitch,
                           void* depth_dev, size_t depth_pitch, void* normals_dev, size_t normals_pitch, void* keys_dev,
                           size_t keys_pitch, int rows, int cols);
/* BLOCKING.  status[0] = pixels the last yakhip_raycast_sharded call had to settle in its second step,
 * status[1] = calls since the last query whose list overflowed its capacity (their surplus pixels hold the
 * refined point instead of the normal: repeat such a frame's ray cast after
 * yakhip_ctx_set_option(ctx, "composite_capacity", pixels)). */
int yakhip_comm_status(yakhip_ctx* ctx, yakhip_comm* comm, unsigned status[2]);

/* ---- tracking side: depth pre-processing and projective ICP ------------
 * Depth images: uint16 millimetres; point / normal maps: float4 {x, y, z, w}, NaN = no data.  All
 * row-pitched (pitch in bytes; multiple of 2 resp. 16). */

/* Replaces kfusion::device::bilateralFilter (internal.hpp:146; yak/src/cuda/imgproc.cu:10-56).
 * sigma_depth in metres, sigma_spatial and kernel_size in pixels.  src != dst. */
int yakhip_bilateral_filter(yakhip_ctx* ctx, const void* src_dev, size_t src_pitch, void* dst_dev, size_t dst_pitch,
                            int rows, int cols, int kernel_size, float sigma_spatial, float sigma_depth);
/* Replaces kfusion::device::truncateDepth (imgproc.cu:66-85): depth > max_dist (metres) -> 0, in place. */
int yakhip_truncate_depth(yakhip_ctx* ctx, void* depth_dev, size_t depth_pitch, int rows, int cols, float max_dist);
/* Replaces kfusion::device::depthPyr (imgproc.cu:94-136): dst is (src_rows / 2) x (src_cols / 2). */
int yakhip_depth_pyramid(yakhip_ctx* ctx, const void* src_dev, size_t src_pitch, int src_rows, int src_cols,
                         void* dst_dev, size_t dst_pitch, float sigma_depth);
/* Replaces kfusion::device::computePointNormals (internal.hpp:160; imgproc.cu:210-250):
 * back-projects every pixel and takes the normal of the triangle (x,y), (x+1,y), (x,y+1).
 * intr = the intrinsics of THIS pyramid level (Intr::operator()(level), yak/src/kfusion/precomp.cpp:10-14). */
int yakhip_compute_points_normals(yakhip_ctx* ctx, const float intr[4], const void* depth_dev, size_t depth_pitch,
                                  int rows, int cols, void* points_dev, size_t points_pitch, void* normals_dev,
                                  size_t normals_pitch);
/* Replaces kfusion::device::resizePointsNormals (internal.hpp:157; imgproc.cu:348-394): 2 x 2 box
 * average where all four points exist; outputs are (src_rows / 2) x (src_cols / 2). */
int yakhip_resize_points_normals(yakhip_ctx* ctx, const void* points_dev, size_t points_pitch, const void* normals_dev,
                                 size_t normals_pitch, int src_rows, int src_cols, void* points_out_dev,
                                 size_t points_out_pitch, void* normals_out_dev, size_t normals_out_pitch);
/* The same twice in one launch, for the three-level model pyramid after a ray cast (kinfu.cpp:294-306 calls
 * resizePointsNormals level by level): half-size maps (src / 2) and quarter-size maps ((src / 2) / 2), bit for bit
 * those of two calls.  No counterpart of its own in the reference. */
int yakhip_resize_points_normals2(yakhip_ctx* ctx, const void* points_dev, size_t points_pitch, const void* normals_dev,
                                  size_t normals_pitch, int src_rows, int src_cols, void* points1_dev, size_t points1_pitch,
                                  void* normals1_dev, size_t normals1_pitch, void* points2_dev, size_t points2_pitch,
                                  void* normals2_dev, size_t normals2_pitch);
/* Replaces kfusion::device::ComputeIcpHelper::operator() (Points overload, internal.hpp:113;
 * yak/src/cuda/proj_icp.cu:74-102, 105-387, 416-439) together with the stream synchronisation of
 * StreamHelper::get (yak/src/kfusion/projective_icp.cpp:42-61): BLOCKING.  For the pose estimate R,t
 * (current -> previous camera) it finds the projective correspondence of every pixel of vcurr in
 * vprev, rejects by distance / normal angle, and returns
 *   sums_host[27]  = sum of row[i] * row[j], i <= j < 7, i < 6, row = [cross(s, n), n, n.(d - s)],
 *                    in the order StreamHelper::get unpacks (upper triangle of A row by row, b interleaved);
 *   counts_host[6] = pixels accepted, rejected as 40 (no source point), 80 (behind / outside the image),
 *                    120 (no destination point), 160 (too far), 200 (normals disagree); may be NULL
 *                    (no counterpart in the reference; its mask output is commented out, proj_icp.cu:350).
 * intr = LEVEL-0 intrinsics, divided by 1 << level as setLevelIntr does (projective_icp.cpp:16-22);
 * dist_thres in metres, angle_thres in radians (ComputeIcpHelper ctor, projective_icp.cpp:10-14).
 * The additions are done in a fixed order: the result is the same on every run. */
int yakhip_icp_accumulate(yakhip_ctx* ctx, const float R[9], const float t[3], const float intr[4], int level,
                          float dist_thres, float angle_thres, const void* vcurr_dev, size_t vcurr_pitch,
                          const void* ncurr_dev, size_t ncurr_pitch, const void* vprev_dev, size_t vprev_pitch,
                          const void* nprev_dev, size_t nprev_pitch, int rows, int cols, float sums_host[27],
                          int counts_host[6]);

/* The same sums over the rows [row_begin, row_end) of the CURRENT maps only (the previous maps are read
 * wherever the correspondences fall), left on the device and NOT waited for: sums_dev receives 33 floats,
 * the 27 sums followed by the 6 counts (exact as floats below 2^24).  No counterpart in the reference;
 * it is the piece SURVEY.md 8e's image-space split needs: every GPU takes a band of rows, the 33 values
 * are added over the GPUs (an all-reduce, yak_amd/shard.py:allreduce_icp_sums), every GPU solves the
 * same 6 x 6 system.  The order of the additions differs from the one-GPU sum, so A and b agree to
 * rounding (1e-4 relative), not bit for bit.  row_begin == row_end gives zeros. */
int yakhip_icp_accumulate_rows(yakhip_ctx* ctx, const float R[9], const float t[3], const float intr[4], int level,
                               float dist_thres, float angle_thres, const void* vcurr_dev, size_t vcurr_pitch,
                               const void* ncurr_dev, size_t ncurr_pitch, const void* vprev_dev, size_t vprev_pitch,
                               const void* nprev_dev, size_t nprev_pitch, int rows, int cols, int row_begin,
                               int row_end, float* sums_dev);

/* One pyramid level of an ICP problem: the current frame's maps, the previous (model) maps, their size
 * and the number of linearisations to run on it (ProjectiveICP::iters_, projective_icp.cpp:87-96). */
typedef struct yakhip_icp_level
{
  const void *vcurr, *ncurr, *vprev, *nprev; /* float4 maps, device */
  size_t vcurr_pitch, ncurr_pitch, vprev_pitch, nprev_pitch;
  int rows, cols;
  int iterations;
} yakhip_icp_level;

/* Replaces the body of kfusion::cuda::ProjectiveICP::estimateTransform (Points variant,
 * yak/src/kfusion/projective_icp.cpp:175-232) in one call: coarse to fine over levels[n_levels-1] .. levels[0]
 * (levels[l] is pyramid level l: intrinsics divided by 1 << l), `iterations` linearisations each, starting
 * from the pose R,t (current -> previous camera).  The reference returns to the host after every
 * linearisation (stream sync, cv::determinant, cv::solve, compose); here accumulation, reduction, the
 * 6 x 6 solve and the pose update are chained on the device and the host waits once.  The device solve
 * is a double-precision Cholesky factorisation, used only while it provably equals the reference's
 * SVD solve (no singular value below OpenCV's cut-off, determinant clear of the 1e-15 test); otherwise
 * the call repeats the estimate with per-linearisation host solves (LU determinant, Jacobi SVD).
 * *ok = 0 reproduces estimateTransform returning false (|det| < 1e-15 or NaN): R_out,t_out then hold
 * the estimate reached before the failing linearisation.  BLOCKING.  Parity of the solve with OpenCV is
 * unpinned (the reference pins no OpenCV version); see DESIGN.md. */
int yakhip_icp_estimate(yakhip_ctx* ctx, const float R[9], const float t[3], const float intr[4], float dist_thres,
                        float angle_thres, const yakhip_icp_level* levels, int n_levels, float R_out[9], float t_out[3],
                        int* ok);

/* yakhip_icp_estimate over the ranks of a communicator: every rank passes the SAME maps (the composited
 * model maps, the replicated current maps) and accumulates the rows [rows * rank / world, rows * (rank + 1) /
 * world) of every level; the 33 sums of each linearisation are all-reduced on the stream, every rank solves
 * the same 6 x 6 system from the same bits and ends with the same pose.  With the RCCL transport the whole
 * schedule (per linearisation: accumulate launch, 33-wave reduction, ncclAllReduce of 132 bytes) is enqueued
 * back to back and the host waits once.  BLOCKING (returns the pose).  Sums agree with the one-GPU estimate to
 * rounding (the order of the additions differs), counts exactly. */
int yakhip_icp_estimate_sharded(yakhip_ctx* ctx, yakhip_comm* comm, const float R[9], const float t[3],
                                const float intr[4], float dist_thres, float angle_thres, const yakhip_icp_level* levels,
                                int n_levels, float R_out[9], float t_out[3], int* ok);

/* ---- surface extraction -------------------------------------------------- */

/* GPU counterpart of yak::marchingCubesCPU's core (yak/src/mc/marching_cubes.cpp:22-194: getOffset,
 * processCube and the triple loop of makeMesh; case table yak/src/mc/marching_cubes_tables.cpp).
 * Every cube whose 8 corners lie inside the slab's layers [z0, z1) and carry weight >= min_weight
 * (weight read as UNSIGNED 16 bit, tsdf_container.h:45-49) is triangulated at iso level 0 of
 * 5 * tsdf.  A triangle is 9 floats {v0.xyz, v1.xyz, v2.xyz} in voxel units times `scale`
 * (MarchingCubesParameters::scale, yak/include/yak/mc/marching_cubes.h:9-16), z counted from layer 0
 * of the whole volume.  BLOCKING.  *n_triangles_host always receives the number of triangles the
 * slab produces; they are written only if triangles_dev != NULL and capacity_triangles is large enough
 * -- call once with NULL to size the buffer, or pass a generous one and compare.
 * The triangle SET equals the CPU routine's; the order is this library's (deterministic: by 63 x 4 x 8
 * cube tile, then z, row, x within the tile, then case-table order), the reference's being
 * nondeterministic (OpenMP critical section, :163-171).
 * Multi-GPU: give each rank's slab one extra layer of overlap (z1 = next slab's z0 + 1) and the
 * per-rank sets are disjoint and complete.  No vertex merging here (MarchingCubesParameters::clean is
 * VTK's vtkCleanPolyData in the reference, :204-216; the host layer does it). */
int yakhip_marching_cubes(yakhip_ctx* ctx, const yakhip_volume* vol, int min_weight, double scale, void* triangles_dev,
                          unsigned long long capacity_triangles, unsigned long long* n_triangles_host);
/* The same over the layers [z_first, z_last) of the slab `vol` (z0 <= z_first < z_last <= z1): the cubes
 * z_first .. z_last - 2.  For a rank that holds more layers than it meshes (its ray-cast halo): the slab's surface
 * map, which belongs to the whole slab, still applies.  With a surface map (either call) the count pass reads only
 * where a marked brick is near -- a cube has triangles only if one of its corners is negative --; the triangles are
 * the same set in the same order. */
int yakhip_marching_cubes_layers(yakhip_ctx* ctx, const yakhip_volume* vol, int z_first, int z_last, int min_weight,
                                 double scale, void* triangles_dev, unsigned long long capacity_triangles,
                                 unsigned long long* n_triangles_host);

/* The exact part of MarchingCubesParameters::clean (yak/src/mc/marching_cubes.cpp:204-216: vtkCleanPolyData,
 * point merging on): corners of the triangle soup with the same three bit patterns become one vertex.
 * triangles_dev: n_triangles * 9 floats as yakhip_marching_cubes writes them.  vertices_dev (3 floats per
 * vertex, room for 3 * n_triangles of them) receives the distinct points in the order of their first appearance
 * in the triangle list (the order the reference's incremental point locator numbers them in; deterministic:
 * yakhip_marching_cubes fixes every triangle's slot); index_dev (one uint32 per corner, 3 * n_triangles)
 * receives each corner's vertex.  Dropping triangles that repeat an index is left to the caller (one linear
 * pass).  Both output buffers 4-byte aligned.  BLOCKING; its scratch (about 20 bytes per corner) is kept in the
 * context and grows with the largest call.  At most 2^30 / 3 triangles.  (Until round 3 the points came out in
 * ascending order of their bit patterns, which cost three sorts.) */
int yakhip_weld_vertices(yakhip_ctx* ctx, const void* triangles_dev, unsigned long long n_triangles, void* vertices_dev,
                         void* index_dev, unsigned long long* n_vertices_host);

/* ---- cloud extraction --------------------------------------------------- */

/* Replaces kfusion::device::extractCloud (internal.hpp:163; yak/src/cuda/tsdf_volume.cu:464-665, 764-780):
 * every voxel with weight != 0 and tsdf != 1 gives one point for each of its +x, +y, +z neighbours that
 * also has weight != 0, tsdf != 1 and the opposite strict sign -- the linear zero crossing between the two
 * voxel CENTRES ((i + 0.5) * voxel_size) -- mapped by aff = (R, t), the volume pose
 * (yak/src/kfusion/tsdf_volume.cpp:147-168).  points_dev: float4 {x, y, z, 0}, 16-byte aligned.
 * BLOCKING.  *n_points_host receives the number of points the slab holds; the first
 * min(n, capacity_points) of them are written (the reference stops at its buffer's size the same way,
 * :639-642); points_dev == NULL only counts.  The point SET equals the reference's; the order is this
 * library's (deterministic), the reference's comes from a global atomic counter (:616).
 * A slab [z0, z1) gives the points of the voxels z0 .. z1 - 2: overlap slabs by one layer as for
 * yakhip_marching_cubes. */
int yakhip_extract_cloud(yakhip_ctx* ctx, const yakhip_volume* vol, const float R[9], const float t[3], void* points_dev,
                         unsigned long long capacity_points, unsigned long long* n_points_host);

/* Replaces kfusion::device::extractNormals (internal.hpp:164-165; tsdf_volume.cu:672-763, 782-797): for
 * every point (in the frame of aff = (R, t)) the central-difference gradient of the trilinear tsdf at
 * Rinv * (p - t), rotated by R and normalised; quiet NaN where the nearest voxel is within 2 of a face of
 * the volume.  Whole volume only.  normals_dev: float4 {x, y, z, 0}.  Not blocking. */
int yakhip_extract_normals(yakhip_ctx* ctx, const yakhip_volume* vol, const void* points_dev, unsigned long long n_points,
                           const float R[9], const float t[3], const float Rinv[9], float gradient_delta_factor,
                           void* normals_dev);

/* ---- debug display ------------------------------------------------------
 * Images: 4 bytes per pixel {b, g, r, 0} (kfusion::RGB, yak/include/yak/kfusion/types.hpp:39-49), row
 * pitch a multiple of 4.  Not blocking. */

/* Replaces kfusion::device::renderImage, Points overload (internal.hpp:167-171; yak/src/cuda/imgproc.cu:453-504,
 * 517-524): Phong shading (ambient 0.3, diffuse 0.5, specular 0.2 with power 20, white light at light_pose,
 * eye at the origin) of the camera-frame vertex / normal maps; pixels whose point is NaN get the vertical
 * backdrop gradient.  The specular power is v^20 by five multiplications; the reference's __powf is
 * the CUDA hardware's exp2(20 log2 v) approximation, so grey levels may differ by one (parity unpinned). */
int yakhip_render_image_points(yakhip_ctx* ctx, const void* points_dev, size_t points_pitch, const void* normals_dev,
                               size_t normals_pitch, int rows, int cols, const float light_pose[3], void* image_dev,
                               size_t image_pitch);
/* Replaces kfusion::device::renderImage, Depth overload (internal.hpp:162-166; imgproc.cu:400-451, 508-515):
 * the same with the vertex re-projected from depth (uint16 mm) by intr = {fx, fy, cx, cy} -- the
 * Reprojector's arguments as given (the reference's host wrapper passes fy for cy, imgproc.cpp:92). */
int yakhip_render_image_depth(yakhip_ctx* ctx, const void* depth_dev, size_t depth_pitch, const void* normals_dev,
                              size_t normals_pitch, int rows, int cols, const float intr[4], const float light_pose[3],
                              void* image_dev, size_t image_pitch);
/* Replaces kfusion::device::renderTangentColors (internal.hpp:172; imgproc.cu:530-561):
 * r, g, b = (uchar)((5 - n.x * 3.5) * 25.5), (5 - n.y * 2.5) * 25.5, (5 - n.z * 3.5) * 25.5; the float ->
 * uchar conversion truncates and saturates, NaN -> 0, as the GPU conversion does. */
int yakhip_render_tangent_colors(yakhip_ctx* ctx, const void* normals_dev, size_t normals_pitch, int rows, int cols,
                                 void* image_dev, size_t image_pitch);

#ifdef __cplusplus
}
#endif
#endif /* YAKHIP_H_ */
