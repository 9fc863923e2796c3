// yakhip_internal.h -- shared between the HIP translation units of libyakhip.so (not installed).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>
#include <vector>

#include "yakhip.h"

struct yakhip_ctx
{
  int device;
  hipStream_t own_stream;
  hipStream_t stream;  // own_stream or a borrowed one (between ctx_side_begin / _end: side_stream)
  hipStream_t side_stream, main_stream;  // yakhip_ctx_side_begin: a second stream, created on first use; the stream to go back to
  hipEvent_t side_done, main_mark;  // end of the side work; the main stream's position at the last side_end
  int main_stream_saved, main_marked;
  int z_chunk;         // 0 = auto
  int nontemporal;     // volume stream bypasses cache retention
  int force_vx;        // 0 = widest the layout allows; 1/2/4 caps the voxels per lane
  int zu;              // (unused; kept for option compatibility)
  int group_test;      // coarse group test on (default) / off
  int chunk_order;     // -1 automatic; 0 ascending, 1 shallow / deep in turn, 3 deepest first (tuning)
  int chunk_min;       // 0 automatic; shortest chunk of a graded table, layers (tuning)
  int chunk_ramp;      // 0 automatic (6); a chunk starting at layer zs is about zs / ramp layers long; -1: all chunks alike (tuning)
  int raycast_bands;   // ray-cast tiles: -1 automatic, 1 each XCD a contiguous band of the image, 0 image order, >= 2 scattered
  int surface_skip;    // the ray cast consults a volume's surface map (default) / ignores it
  int batch_as_singles;  // set while yakhip_integrate_batch runs its frames as single-frame launches (integrate.hip)
  int raycast_segments;  // waves per ray with the surface map: 0 the default, 1 one wave marches the whole ray, 2 or 4
  int brick_test;      // classify whole bricks (dead / front / mixed) before walking them (default) / off
  int checkpoint;      // deep slabs start from closed-form column positions (default) / replay from layer 0
  int tile_shift;      // 0 = default wave footprint, else 3..6 (tuning)
  int tile_side;       // the four waves of a workgroup lie side by side along x (default) / stacked in y
  int mc_z_chunk;      // 0 = default cube layers per marching-cubes tile (tuning)
  void* thr;           // scratch: per-pixel band thresholds of the frame being integrated
  size_t thr_bytes;
  // yakhip_integrate_prepare: the prepass of a frame already run (on whatever stream was current) into slot
  // prepared.slot of thr (which then holds two); the next yakhip_integrate of the same image takes it from there
  struct
  {
    const void* dists;
    uint32_t pitch;
    int rows, cols;
    float trunc_hi;
    int slot;
    bool valid;
    bool fine;  // the fine cells were filled too (what a two-kernel launch needs)
  } prepared;
  int prepared_next;   // slot the next prepare call writes (0 / 1 in turn)
  void* ckpt;          // scratch: column positions at layer z0 of a deep slab (integrate.hip, checkpoint_kernel)
  size_t ckpt_bytes;
  uint8_t* classes;    // between the two kernels of a large launch: a class byte per wave brick (IntegrateParams::classes)
  size_t classes_bytes;
  uint64_t classes_n;  // ... bricks of the last launch that used it (the bytes behind them are zero)
  unsigned long long launches_one, launches_two;  // yakhip_ctx_launch_counts
  uint32_t seq_signature[8];  // which volume the last launches were of
  uint32_t launch_seq, classes_seq;  // launches that could have used it so far; the one whose classes it holds
  uint32_t* stats_host; // pinned: {front bricks, mixed bricks, wave bricks, launch number} of an earlier launch (integrate.hip, prepare_list)
  float* cptab;        // ... and the column checkpoints (IntegrateParams::cptab)
  size_t cptab_bytes;
  int defer_mixed;     // 1 (default): mixed bricks of large launches go through that list; 0: walked where they are classified
  int mixed_waves;     // waves per SIMD of the second kernel (0 = default)
  int refine;          // the second kernel looks at a mixed brick again, in up to eight pieces along z (default) / walks it whole
  int num_cus;
  hipEvent_t thr_read[2];  // after the last launch that read slot 0 / 1 of thr (recorded once yakhip_integrate_prepare is in use)
  int prepare_used, thr_read_valid[2];
  int thr_slot_read;   // half of thr the launch being set up reads (-1: all of it)
  uint32_t* brick_stats; // device counters {mixed, dead, front, -}, allocated by the "brick_stats" option
  void* static_scratch;  // this device's copy of integrate.hip's file-scope scratch array (yakhip_integrate_static)
  void* icp_scratch;   // ICP: workgroup partials + the reduced 33 values (device)
  size_t icp_scratch_bytes;
  float* icp_sums;     // sharded ICP, host-solve path: the band's 33 values between accumulate and all-reduce (device)
  void* icp_pinned;    // ICP: pinned host mirror of the 33 values / the state; +128 the state and +192 the ticket icp_solve_kernel writes
  unsigned icp_ticket; // tickets handed out to yakhip_icp_estimate calls
  int icp_poll;        // the host polls that ticket (default) / copies and synchronises
  void* mc_scratch;    // marching cubes: tile counts, tile offsets, total (device)
  size_t mc_scratch_bytes;
  void* weld_scratch;  // yakhip_weld_vertices: hash table, per-corner slots, sort arrays (device; grown on demand)
  size_t weld_scratch_bytes;
  void* mc_pinned;     // marching cubes: pinned host copy of the total
  void* stage[2];      // host <-> device copies: two pinned staging buffers and the events that free them
  hipEvent_t stage_done[2];
  int composite_capacity;  // list slots of the sharded ray cast's second composite step (0 = auto)
  int kernel_timing;   // record a hipEvent pair around every integrate_kernel launch
  std::vector<std::pair<hipEvent_t, hipEvent_t>> timing_events;
  size_t timing_used;
};

namespace yakhip
{
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Records the message for yakhip_last_error() and returns `code`.
int fail(int code, const char* fmt, ...);
// hipError -> return code (0 stays 0); records file/line text on failure.
int check(hipError_t e, const char* what);

// "kernel_timing" option: bracket the dominant kernel of a call with an event pair on the context's
// stream.  begin() hands back nullptr when the option is off; end(nullptr) is a no-op.
int timing_begin(yakhip_ctx* ctx, std::pair<hipEvent_t, hipEvent_t>** ev);
int timing_end(yakhip_ctx* ctx, std::pair<hipEvent_t, hipEvent_t>* ev);

#define YAKHIP_TRY(expr)                                   \
  do                                                       \
  {                                                        \
    int _rc = ::yakhip::check((expr), #expr);              \
    if (_rc != 0)                                          \
      return _rc;                                          \
  } while (0)

// Everything the integrate kernel needs, passed by value in the kernarg segment so the
// pose / intrinsics / sizes live in SGPRs (wave-uniform), not in LDS or VGPRs.
struct IntegrateParams
{
  uint32_t* vol;         // layer z0 of the slab
  const uint8_t* dists;  // half image, row pitch in bytes
  const f32x2* thr;      // per-pixel {A2, B2} band thresholds, dense rows x cols, and one more entry {-1, -1}
  uint32_t thr_none;     // ... its index (rows * cols): what a voxel that sees no pixel gathers
  const float* coarse;   // coarse minima of A2 (group test), cw x ch cells
  const float* ckpt;     // nullptr, or camera-space x | y | z of every column at layer z0 (3 planes of X*Y floats)
  const float* cellmin;  // brick test: min of A2 / max of the B2 word over each 16 x 16 pixel cell, ccw x cch cells
  const float* cellmax;
  int ccw, cch;
  const float* finemin;  // the same two planes over 4 x 4 pixel cells, fcw x fch: the second look at a mixed brick, piece by piece
  const float* finemax;
  int fcw, fch;
  uint8_t* classes;      // nullptr, or where the brick test leaves its findings: a byte per wave brick (parts to do | 0x80 if
                         // front; 0 = dead), brick = (chunk * grid_x + plane tile) * 4 + wave
  int two_kernels;       // 0: one kernel, every workgroup walks its own bricks.  1: classify_kernel leaves the classes,
                         // integrate_list_kernel deals the live bricks' parts out evenly (integrate.hip)
  uint32_t n_bricks, grid_x;  // wave bricks of the launch; plane tiles (workgroups of the one-kernel grid per chunk)
  float* cptab;          // two kernels: camera-space positions of every column at the layers z0 + k * cp_stride, k < cp_count
  int cp_ffwd;           // the table starts from closed-form positions at z0 (a deep slab) / from z0 replayed steps
  int cp_stride, cp_count;  // (plane floats of x, then y, then z per k), written by extra workgroups of the first kernel, so
                         // that the second starts a part from the checkpoint below it instead of replaying from layer 0
  uint32_t* bricks;      // nullptr, or the slab's surface map: one bit per 16^3 brick, bricks_x x bricks_y x bricks_z,
  int bricks_x, bricks_y, bricks_z;  // then (bricks_words 32-bit words on) one bit per cell of 4 x 4 x 4 bricks
  uint32_t bricks_words;
  uint32_t* brick_stats; // nullptr, or counters {mixed, dead, front} of classified wave bricks ("brick_stats" option)
  int brick_enable;      // classify each workgroup's brick (x-y tile times z-chunk) before walking it
  float brick_eps;       // bound on |replayed fp32 position - ideal position| per camera coordinate (metres)
  float brick_zmin;      // bricks nearer than this (camera z) are not classified
  int cw, ch;
  int grp_enable;
  float grp_margin, grp_cols_hi, grp_rows_hi, grp_span, grp_inv_bk, grp_bias, grp_zmin;
  uint32_t dists_pitch;
  int rows, cols;
  int X, Y;              // in-plane extent
  int z0, z1;            // owned layers
  int z_chunk;           // longest chunk of the launch, layers (the chunks themselves: ChunkTable)
  int tile_shift;        // log2 of the lanes a wave lays along x (6 = layer order); integrate.hip lane_voxels()
  int tile_side;         // waves of a workgroup side by side along x (1) or stacked in y (0)
  uint32_t plane;        // X * Y
  float vsx, vsy, vsz;
  float trunc, trunc_inv;
  float trunc_hi;        // trunc * (1 + 2^-9): slack threshold of the sqrt-free band tests
  int max_weight;
  float R[9];
  float t[3];
  float fx, fy, cx, cy;
};

// The z-chunks of an integrate launch in dispatch order, passed by value beside IntegrateParams: workgroups with
// blockIdx.y == j walk layers [z0 + first[j], z0 + first[j] + layers[j]).  integrate.hip, chunk_table().
constexpr int kMaxChunks = 128;
struct ChunkTable
{
  uint32_t first[kMaxChunks];
  uint16_t layers[kMaxChunks];
};

// Surface map, second level: after brick (bx, by, bz) of a bricks_x x bricks_y x bricks_z grid has been marked, mark
// the cells (4 x 4 x 4 bricks, bits from word `words` on) that hold it or are touched by it from next door.
__device__ __forceinline__ void surface_mark_cells(uint32_t* map, uint32_t words, int bricks_x, int bricks_y, int bricks_z,
                                                   int bx, int by, int bz)
{
  const int ncx = (bricks_x + 3) >> 2, ncy = (bricks_y + 3) >> 2, ncz = (bricks_z + 3) >> 2;
  const int cx = bx >> 2, cy = by >> 2, cz = bz >> 2;
  const int x0 = (bx & 3) == 0 && cx > 0 ? cx - 1 : cx, x1 = (bx & 3) == 3 && cx + 1 < ncx ? cx + 1 : cx;
  const int y0 = (by & 3) == 0 && cy > 0 ? cy - 1 : cy, y1 = (by & 3) == 3 && cy + 1 < ncy ? cy + 1 : cy;
  const int z0 = (bz & 3) == 0 && cz > 0 ? cz - 1 : cz, z1 = (bz & 3) == 3 && cz + 1 < ncz ? cz + 1 : cz;
  for (int z = z0; z <= z1; ++z)
    for (int y = y0; y <= y1; ++y)
      for (int x = x0; x <= x1; ++x)
      {
        const uint32_t c = (uint32_t)x + (uint32_t)ncx * ((uint32_t)y + (uint32_t)ncy * (uint32_t)z);
        uint32_t* word = map + words + (c >> 5);
        const uint32_t bit = 1u << (c & 31u);
        if (!(__builtin_nontemporal_load(word) & bit))  // (the steady state: set long ago; a stale line costs one atomic)
          atomicOr(word, bit);
      }
}

// Ray casting (raycast.hip).  Everything wave-uniform travels in the kernarg segment.
struct RaycastParams
{
  const uint32_t* vol;  // whole volume, layer 0
  int X, Y, Z;
  float vsx, vsy, vsz;  // voxel_size
  float ivx, ivy, ivz;  // 1 / voxel_size            (tsdf_volume.cu:402)
  float gdx, gdy, gdz;  // voxel_size * grad factor  (tsdf_volume.cu:401)
  float bx, by, bz;     // box_max = voxel_size * dims - voxel_size  (tsdf_volume.cu:399, 306)
  float time_step;      // trunc_dist * step factor  (tsdf_volume.cu:400)
  float R[9], t[3], Rinv[9];
  float finvx, finvy, cx, cy;
  int rows, cols;
  uint8_t* points;      // float4 image or nullptr
  uint8_t* normals;     // float4 image
  uint8_t* depth;       // uint16 image or nullptr
  uint32_t points_pitch, normals_pitch, depth_pitch;
  // yakhip_raycast_points_pyramid: half- and quarter-size maps (resizePointsNormals of the result, once and twice),
  // written by the same launch; level[0] null = none, level[1] null = the half size only.  Whole volume only.
  uint8_t *pyr_points[2], *pyr_normals[2];
  uint32_t pyr_points_pitch[2], pyr_normals_pitch[2];
  // slab mode (keys != nullptr): vol starts at layer z_base and holds z_held layers, of which the global
  // layers [z_own0, z_own1) are this slab's own; keys = int32 image of march-ending pair indices
  uint8_t* keys;
  uint32_t keys_pitch;
  int z_base, z_held, z_own0, z_own1;
  // yakhip_raycast_sharded only (else null): per pixel, row-major without padding, the key once more and the march time
  // of the hit's refined point (NaN without a hit) -- what the composite ships instead of the vertex (comm.hip)
  int32_t* key_out;
  float* ts_out;
  // surface map of the held layers (yakhip_volume.surface_map) or nullptr: one bit per 16^3 brick, skip_x x skip_y
  // bricks per brick layer, skip_words 32-bit words in all
  const uint32_t* skip;
  int skip_x, skip_y, skip_words;  // bricks per row / column, 32-bit words of brick bits
  int cell_x, cell_y, cell_z;      // cells (4 x 4 x 4 bricks) per row / column / in z; their bits follow the brick bits
  int cell_words;
  int cell_jump;                   // 1: the march may cross the free space in front of the first marked cell in one go
  unsigned long long vol_bytes;    // bytes of the held layers
};

// 32-bit words of a slab's surface map (yakhip.h): one bit per 16^3 brick of its layers
inline size_t surface_brick_words(int X, int Y, int layers)
{
  return ((size_t)((X + 15) >> 4) * (size_t)((Y + 15) >> 4) * (size_t)((layers + 15) >> 4) + 31u) / 32u;
}
// ... followed by one bit per cell of 64^3 voxels (4 x 4 x 4 bricks): set when a marked brick lies in the cell or
// touches it from a neighbouring cell, i.e. clear = no marked brick within 16 voxels of the cell
inline size_t surface_cell_words(int X, int Y, int layers)
{
  return ((size_t)((X + 63) >> 6) * (size_t)((Y + 63) >> 6) * (size_t)((layers + 63) >> 6) + 31u) / 32u;
}

// use_static: thresholds / checkpoints go to the file-scope array instead of context-owned buffers, and the
// call neither allocates nor creates events (yakhip_integrate_static)
int launch_integrate(yakhip_ctx* ctx, const IntegrateParams& p, bool use_static = false);
// the pose-independent part of launch_integrate's set-up, ahead of time (yakhip_integrate_prepare)
int launch_integrate_prepare(yakhip_ctx* ctx, const IntegrateParams& p);
// device address of the CURRENT device's copy of that array (hipGetSymbolAddress; once per context)
int static_scratch_address(void** out);
// n_frames >= 1 frames applied in order to the same slab; thr / coarse / grp_* / z_chunk are filled here
int launch_integrate_batch(yakhip_ctx* ctx, const IntegrateParams* frames, int n_frames);
int launch_raycast(yakhip_ctx* ctx, const RaycastParams& p);
int launch_raycast_resolve(yakhip_ctx* ctx, const RaycastParams& p);
int launch_extract_normals(yakhip_ctx* ctx, const RaycastParams& p, const void* points, void* normals, unsigned long long n);  // slab mode, after the composite
int launch_clear(const yakhip_ctx* ctx, uint32_t* data, size_t n_voxels);
int launch_surface_scan(yakhip_ctx* ctx, const uint32_t* vol, int X, int Y, int layers, uint32_t* map);  // raycast.hip
int launch_selftest_division(yakhip_ctx* ctx, unsigned long long counts[4]);
int launch_compute_dists(const yakhip_ctx* ctx, const uint8_t* depth, size_t depth_pitch, uint8_t* dists,
                         size_t dists_pitch, int rows, int cols, float fx, float fy, float cx, float cy);

// frameproc.hip
int launch_bilateral(const yakhip_ctx* ctx, const uint8_t* src, size_t src_pitch, uint8_t* dst, size_t dst_pitch,
                     int rows, int cols, int kernel_size, float sigma_spatial, float sigma_depth);
int launch_truncate_depth(const yakhip_ctx* ctx, uint8_t* depth, size_t pitch, int rows, int cols, float max_dist);
int launch_depth_pyramid(const yakhip_ctx* ctx, const uint8_t* src, size_t src_pitch, int src_rows, int src_cols,
                         uint8_t* dst, size_t dst_pitch, float sigma_depth);
int launch_points_normals(const yakhip_ctx* ctx, const float intr[4], const uint8_t* depth, size_t depth_pitch, int rows,
                          int cols, uint8_t* points, size_t points_pitch, uint8_t* normals, size_t normals_pitch);
int launch_resize_points_normals2(const yakhip_ctx* ctx, const uint8_t* points, size_t points_pitch, const uint8_t* normals,
                                  size_t normals_pitch, int src_rows, int src_cols, uint8_t* points1, size_t points1_pitch,
                                  uint8_t* normals1, size_t normals1_pitch, uint8_t* points2, size_t points2_pitch,
                                  uint8_t* normals2, size_t normals2_pitch);
int launch_resize_points_normals(const yakhip_ctx* ctx, const uint8_t* points, size_t points_pitch,
                                 const uint8_t* normals, size_t normals_pitch, int src_rows, int src_cols,
                                 uint8_t* points_out, size_t points_out_pitch, uint8_t* normals_out,
                                 size_t normals_out_pitch);
// debug display (imgproc.cu:400-561).  reproj_args = {fx, fy, cx, cy} of the Reprojector (depth variant only)
int launch_render_image(const yakhip_ctx* ctx, bool from_depth, const uint8_t* src, size_t src_pitch, const uint8_t* normals,
                        size_t normals_pitch, int rows, int cols, const float reproj_args[4], const float light[3],
                        uint8_t* image, size_t image_pitch);
int launch_tangent_colors(const yakhip_ctx* ctx, const uint8_t* normals, size_t normals_pitch, int rows, int cols,
                          uint8_t* image, size_t image_pitch);
// icp.hip (blocking: returns after the 27 sums and 6 counters are on the host)
int launch_icp_accumulate(yakhip_ctx* ctx, const float R[9], const float t[3], const float intr[4], int level,
                          float dist_thres, float angle_thres, const uint8_t* vcurr, size_t vcurr_pitch,
                          const uint8_t* ncurr, size_t ncurr_pitch, const uint8_t* vprev, size_t vprev_pitch,
                          const uint8_t* nprev, size_t nprev_pitch, int rows, int cols, float sums[27], int counts[6],
                          int row_begin, int row_end, float* sums_dev);
// (rows [row_begin, row_end) of the current maps; sums_dev != nullptr: leave the 33 values there and do not wait)

// mc.hip (blocking).  `vol` points at layer z0; `layers` voxel layers are present, so layers - 1 cube
// layers are meshed.  Only counts when triangles == nullptr or the count exceeds `capacity`.
int launch_weld(yakhip_ctx* ctx, const uint32_t* corners, uint32_t n, uint32_t* vertices, uint32_t* index,
                unsigned long long* n_vertices_host);
int launch_marching_cubes(yakhip_ctx* ctx, const uint32_t* vol, int X, int Y, int z0, int layers, int min_weight,
                          double scale, float* triangles, unsigned long long capacity, unsigned long long* n_out,
                          const uint32_t* map = nullptr, int map_z0 = 0);
// (map: the surface map of the slab these layers are part of, whose layer map_z0 is `vol`'s first; nullptr: read everything)
int launch_extract_cloud(yakhip_ctx* ctx, const uint32_t* vol, int X, int Y, int z0, int layers, const float vs[3],
                         const float R[9], const float t[3], float* points, unsigned long long capacity,
                         unsigned long long* n_out, const uint32_t* map = nullptr);

// icp.hip: the device-resident schedule.  status_out: 0 = done, 2 = some system needs the host path.
int launch_icp_estimate(yakhip_ctx* ctx, const float R[9], const float t[3], const float intr[4], float dist_thres,
                        float angle_thres, const yakhip_icp_level* levels, int n_levels, float R_out[9], float t_out[3],
                        int* status_out, int* iterations_out);

// comm.hip: the exchange steps of the sharded volume
struct CompositeImages
{
  uint8_t *points, *normals, *depth, *keys;  // points or depth may be null
  uint32_t points_pitch, normals_pitch, depth_pitch, keys_pitch;
  int rows, cols;
};
int comm_allreduce(yakhip_ctx* ctx, yakhip_comm* comm, void* buf, size_t count, int op);  // on ctx->stream
int comm_rank(const yakhip_comm* c);
int comm_world(const yakhip_comm* c);
// (scratch of the communicator for an image of n pixels: where the slab ray cast leaves its keys and march times)
int composite_buffers(yakhip_ctx* ctx, yakhip_comm* comm, int n, int cap_request, int32_t** keys_flat, float** ts_flat);
int composite_first(yakhip_ctx* ctx, yakhip_comm* comm, const RaycastParams& rp, int cap_request);
int composite_second(yakhip_ctx* ctx, yakhip_comm* comm, const CompositeImages& im, int cap_request);
// raycast.hip: the composited pixel from {march time, normal}: the vertex and the depth are rebuilt with the ray cast's own
// arithmetic; keys <- best
int launch_composite_unpack(yakhip_ctx* ctx, const RaycastParams& p, const int32_t* best, const uint32_t* packed);
int composite_status(yakhip_ctx* ctx, yakhip_comm* comm, unsigned out[2]);
// icp.hip: the device-resident schedule over row bands, the 33 sums of every linearisation all-reduced on the stream
int launch_icp_estimate_sharded(yakhip_ctx* ctx, yakhip_comm* comm, const float R[9], const float t[3], const float intr[4],
                                float dist_thres, float angle_thres, const yakhip_icp_level* levels, int n_levels,
                                float R_out[9], float t_out[3], int* status_out, int* iterations_out);

// ---- device helpers shared by kernels --------------------------------------------------------
__device__ __forceinline__ float half_bits_to_float(uint32_t bits)
{
  return (float)__builtin_bit_cast(_Float16, (unsigned short)bits);  // v_cvt_f32_f16: exact
}

// fp32 -> fp16, round-to-nearest-even, of a value that has ALREADY been rounded to fp32
// (__float2half_rn semantics).  The empty asm hides the producer from instruction selection:
// without it hipcc folds `(half)(a * b)` into v_fma_mixlo_f16, which rounds the exact product
// once, straight to fp16 -- measured on gfx950 to differ from mul-then-convert in ~1e-4 of pixels.
__device__ __forceinline__ uint32_t float_to_half_bits(float f)
{
  asm("" : "+v"(f));
  return (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)f);  // v_cvt_f16_f32
}

// temp_utils.hpp:36-39 -- the nesting order is part of the result
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
  return __builtin_fmaf(ax, bx, __builtin_fmaf(ay, by, az * bz));
}
}  // namespace yakhip
