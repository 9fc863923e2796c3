// raycast.hip -- volume ray casting for gfx950: replaces kfusion::device::raycast
// (yak/src/cuda/tsdf_volume.cu:149-425).  Citations are relative to /root/reference/.
//
// Arithmetic follows the reference statement by statement (fp32, no contraction: this TU is built
// with -ffp-contract=off and correctly rounded '/' and sqrtf), so the result is bit-identical to the
// CPU oracle; what differs is the schedule:
//   * a wave owns an 8 x 8 pixel tile (not a 32 x 1 row segment of a 32 x 8 block): neighbouring rays
//     sample neighbouring voxels, so one wave-wide gather touches ~8 short row segments, not 64 lines;
//   * the march positions do not depend on the fetched values, so U samples are put in flight at once
//     and then examined in order -- the loop-carried dependency is the exit test only;
//   * lanes that found a zero crossing leave the march and do the 64 trilinear taps together after it;
//   * workgroup ids are remapped so each XCD (own L2) walks a contiguous band of the image.
//
// Slab mode (SLAB = true; SURVEY.md 8e option 1, no counterpart in the reference): the volume is cut in
// z-slabs over several GPUs, every GPU holds its own layers plus a halo.  Each GPU marches ALL rays over
// the same global sample sequence (the positions and march times are pure arithmetic, replayed without
// fetches where the slab holds nothing) and examines the sample pairs whose first sample falls into a
// layer it owns.  It reports the index k of the first pair that ends the march (zero crossing or the
// - to + abort) with the hit's vertex and normal; the pair with the smallest k over all GPUs is the one
// the whole-volume march stops at, so "min k wins" composites the slabs into the same image, bit for bit.
// One case needs a second step: the secant refinement of a hit extrapolates when the trilinear values at
// the two bracketing samples have the same sign, and the refined point can then lie any distance along the
// ray, outside every halo.  The bracketing taps are always held, so the owner of the pair still computes
// the point exactly; it stores the point (volume frame) in place of the normal and sets the key's low bit,
// and after the composite the slab that owns the point's layer computes the normal there
// (raycast_resolve_kernel) -- about 1 hit in 1500 on the orbit scene.
#include "yakhip_internal.h"

namespace yakhip
{
namespace
{
constexpr int kUnrollPlain = 4;
#ifndef YAK_RAYCAST_UNROLL_SKIP
#define YAK_RAYCAST_UNROLL_SKIP 8
#endif
constexpr int kDefaultSegments = 1;  // waves per ray when the surface map is consulted (raycast_kernel, SEG)
// 48 KiB of LDS at most for the map (three workgroups a CU then; up to 32 KiB -- 512^3: 4 KiB -- five): a 1024^3 volume
// needs 8192 words of brick bits + 128 of cell bits, a 2048 x 2048 x 276 slab of configuration 5 9376
constexpr int kMaxSkipWords = 12288;

template <bool IDX32>
__device__ __forceinline__ float voxel_tsdf(const RaycastParams& p, int x, int y, int z)
{
  // only the half (low 16 bits of the little-endian voxel word) is read: device.hpp:78-81
  if constexpr (IDX32)
  {
    const uint32_t idx = (uint32_t)x + (uint32_t)p.X * ((uint32_t)y + (uint32_t)p.Y * (uint32_t)z);
    return half_bits_to_float(*(const unsigned short*)((const uint8_t*)p.vol + (size_t)(idx << 2)));
  }
  else
  {
    const size_t idx = (size_t)x + (size_t)p.X * ((size_t)y + (size_t)p.Y * (size_t)z);
    return half_bits_to_float(*(const unsigned short*)(p.vol + idx));
  }
}

// Slab mode: z is a layer of the whole volume, p.vol starts at layer p.z_base and holds p.z_held layers.
// A layer that is not held reads as 0 and clears `held`.
template <bool IDX32, bool SLAB>
__device__ __forceinline__ float voxel_tsdf_at(const RaycastParams& p, int x, int y, int z, bool& held)
{
  if constexpr (!SLAB)
    return voxel_tsdf<IDX32>(p, x, y, z);
  else
  {
    const int zl = z - p.z_base;
    if ((unsigned)zl >= (unsigned)p.z_held)
    {
      held = false;
      return 0.f;
    }
    return voxel_tsdf<IDX32>(p, x, y, zl);
  }
}

__device__ __forceinline__ int clampi(int v, int hi) { return min(max(v, 0), hi); }

// tsdf_volume.cu:209-217: nearest voxel, round-half-even.  The reference does not bounds-test here;
// the clamp keeps the speculative look-ahead fetches inside the allocation and changes nothing where
// the reference's read is defined.  Slab mode: `owned` = the sample's layer belongs to this slab.
template <bool IDX32, bool SLAB>
__device__ __forceinline__ float fetch_tsdf(const RaycastParams& p, float px, float py, float pz, bool& owned, bool& held)
{
  const int x = clampi((int)__builtin_rintf(px * p.ivx), p.X - 1);
  const int y = clampi((int)__builtin_rintf(py * p.ivy), p.Y - 1);
  const int z = clampi((int)__builtin_rintf(pz * p.ivz), p.Z - 1);
  if constexpr (SLAB)
    owned = z >= p.z_own0 && z < p.z_own1;
  return voxel_tsdf_at<IDX32, SLAB>(p, x, y, z, held);
}

// The same with the surface map (one bit per 16^3 brick of the held layers, a copy in LDS): a sample in a brick
// that holds no negative voxel is not fetched.  It reads as +0 with `skipped` set: its true value is >= 0, which
// is all the march needs to know unless the other sample of its pair is negative -- then the march fetches it
// after all.  No branch: the voxel is read through a buffer descriptor over the held layers, and a lane that
// skips passes an offset beyond it (2^31), for which the hardware returns 0 without touching memory.
constexpr int kSkipShift = 4;  // YAKHIP_SURFACE_BRICK = 16
typedef __amdgpu_buffer_rsrc_t rsrc_t;

// min(max(v, 0), hi) for a wave-uniform hi >= 0 in one instruction (the compiler cannot know hi >= 0 and emits two)
__device__ __forceinline__ int clamp_index(int v, int hi)
{
  int r;
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "s"(hi));
  return r;
}
// (index arithmetic in 24-bit multiply-adds, full rate where the 32-bit ones run at a quarter: the launcher takes this
// variant only for X, Y < 2^24 and Y * held layers <= 2^24)
template <bool SLAB, bool BIG>
__device__ __forceinline__ float fetch_tsdf_skip(const RaycastParams& p, rsrc_t rs, const uint32_t* map_lds, float px, float py,
                                                 float pz, bool& owned, bool& held, bool& skipped)
{
  const int x = clamp_index((int)__builtin_rintf(px * p.ivx), p.X - 1);
  const int y = clamp_index((int)__builtin_rintf(py * p.ivy), p.Y - 1);
  const int z = clamp_index((int)__builtin_rintf(pz * p.ivz), p.Z - 1);
  int zl = z;
  bool in = true;
  if constexpr (SLAB)
  {
    owned = z >= p.z_own0 && z < p.z_own1;
    zl = z - p.z_base;
    in = (unsigned)zl < (unsigned)p.z_held;
    held = held && in;
    zl = in ? zl : 0;
  }
  const uint32_t b = (uint32_t)(x >> kSkipShift) +
                     __umul24((uint32_t)p.skip_x, (uint32_t)(y >> kSkipShift) + __umul24((uint32_t)p.skip_y, (uint32_t)(zl >> kSkipShift)));
  const bool marked = (map_lds[b >> 5] >> (b & 31u)) & 1u;
  skipped = in && !marked;
  const uint32_t idx = (uint32_t)x + __umul24((uint32_t)p.X, (uint32_t)y + __umul24((uint32_t)p.Y, (uint32_t)zl));
  if constexpr (BIG)
  {
    // held layers of 2 GiB and more (a 1024^3 volume: 4 GiB) do not fit the descriptor's offsets: a load under the
    // lane's predicate instead (the voxel index still fits 32 bits: the launcher sees to it)
    uint32_t word = 0;
    if (in && marked)
      word = p.vol[(size_t)idx];
    return half_bits_to_float(word & 0xffffu);
  }
  else
  {
    const uint32_t word = __builtin_amdgcn_raw_buffer_load_b32(rs, in && marked ? idx << 2 : 0x80000000u, 0, 0);
    return half_bits_to_float(word & 0xffffu);
  }
}

// The brick bit alone: is the sample's brick marked?  (fetch_tsdf_skip's own index arithmetic.)  Slab mode: a sample in a
// layer that is not held has no brick and reads 0 (`in` cleared); `owned` as in fetch_tsdf_skip.
template <bool SLAB>
__device__ __forceinline__ bool sample_brick_marked(const RaycastParams& p, const uint32_t* map_lds, float px, float py, float pz,
                                                    bool& in, bool& owned)
{
  const int x = clamp_index((int)__builtin_rintf(px * p.ivx), p.X - 1);
  const int y = clamp_index((int)__builtin_rintf(py * p.ivy), p.Y - 1);
  const int z = clamp_index((int)__builtin_rintf(pz * p.ivz), p.Z - 1);
  int zl = z;
  in = true;
  if constexpr (SLAB)
  {
    owned = z >= p.z_own0 && z < p.z_own1;
    zl = z - p.z_base;
    in = (unsigned)zl < (unsigned)p.z_held;
    zl = in ? zl : 0;
  }
  const uint32_t b = (uint32_t)(x >> kSkipShift) +
                     __umul24((uint32_t)p.skip_x, (uint32_t)(y >> kSkipShift) + __umul24((uint32_t)p.skip_y, (uint32_t)(zl >> kSkipShift)));
  return in && ((map_lds[b >> 5] >> (b & 31u)) & 1u);
}

// tsdf_volume.cu:167-192: trilinear tap, argument in voxel units
template <bool IDX32, bool SLAB>
__device__ float interpolate(const RaycastParams& p, float cfx, float cfy, float cfz, bool& held)
{
  const float fx = __builtin_floorf(cfx), fy = __builtin_floorf(cfy), fz = __builtin_floorf(cfz);
  if (!(fx >= 0.f && fx < (float)(p.X - 1) && fy >= 0.f && fy < (float)(p.Y - 1) && fz >= 0.f && fz < (float)(p.Z - 1)))
    return __builtin_nanf("");
  const int gx = (int)fx, gy = (int)fy, gz = (int)fz;
  const float a = cfx - (float)gx, b = cfy - (float)gy, c = cfz - (float)gz;
  const float v000 = voxel_tsdf_at<IDX32, SLAB>(p, gx + 0, gy + 0, gz + 0, held);
  const float v001 = voxel_tsdf_at<IDX32, SLAB>(p, gx + 0, gy + 0, gz + 1, held);
  const float v010 = voxel_tsdf_at<IDX32, SLAB>(p, gx + 0, gy + 1, gz + 0, held);
  const float v011 = voxel_tsdf_at<IDX32, SLAB>(p, gx + 0, gy + 1, gz + 1, held);
  const float v100 = voxel_tsdf_at<IDX32, SLAB>(p, gx + 1, gy + 0, gz + 0, held);
  const float v101 = voxel_tsdf_at<IDX32, SLAB>(p, gx + 1, gy + 0, gz + 1, held);
  const float v110 = voxel_tsdf_at<IDX32, SLAB>(p, gx + 1, gy + 1, gz + 0, held);
  const float v111 = voxel_tsdf_at<IDX32, SLAB>(p, gx + 1, gy + 1, gz + 1, held);
  float tsdf = 0.f;
  tsdf += v000 * (1.f - a) * (1.f - b) * (1.f - c);
  tsdf += v001 * (1.f - a) * (1.f - b) * c;
  tsdf += v010 * (1.f - a) * b * (1.f - c);
  tsdf += v011 * (1.f - a) * b * c;
  tsdf += v100 * a * (1.f - b) * (1.f - c);
  tsdf += v101 * a * (1.f - b) * c;
  tsdf += v110 * a * b * (1.f - c);
  tsdf += v111 * a * b * c;
  return tsdf;
}

// temp_utils.hpp:103 -- v * rsqrt(dot(v, v)), with rsqrt restated as 1 / sqrt
__device__ __forceinline__ void normalize3(float& x, float& y, float& z)
{
  const float s = 1.0f / __builtin_sqrtf(dot3(x, y, z, x, y, z));
  x *= s;
  y *= s;
  z *= s;
}

// compute_normal, tsdf_volume.cu:354-372 (before the rotation into the camera frame)
template <bool IDX32, bool SLAB>
__device__ __forceinline__ void surface_gradient(const RaycastParams& p, float px, float py, float pz, float& gx, float& gy,
                                                 float& gz, bool& held)
{
  const float Fx1 = interpolate<IDX32, SLAB>(p, (px + p.gdx) * p.ivx, py * p.ivy, pz * p.ivz, held);
  const float Fx2 = interpolate<IDX32, SLAB>(p, (px - p.gdx) * p.ivx, py * p.ivy, pz * p.ivz, held);
  const float Fy1 = interpolate<IDX32, SLAB>(p, px * p.ivx, (py + p.gdy) * p.ivy, pz * p.ivz, held);
  const float Fy2 = interpolate<IDX32, SLAB>(p, px * p.ivx, (py - p.gdy) * p.ivy, pz * p.ivz, held);
  const float Fz1 = interpolate<IDX32, SLAB>(p, px * p.ivx, py * p.ivy, (pz + p.gdz) * p.ivz, held);
  const float Fz2 = interpolate<IDX32, SLAB>(p, px * p.ivx, py * p.ivy, (pz - p.gdz) * p.ivz, held);
  gx = (Fx1 - Fx2) / p.gdx;
  gy = (Fy1 - Fy2) / p.gdy;
  gz = (Fz1 - Fz2) / p.gdz;
  normalize3(gx, gy, gz);
}

#ifdef YAK_RAYCAST_TIMING
// diagnostic build only (make EXTRA=-DYAK_RAYCAST_TIMING; tools/bench_raycast.py prints the sums): shader-clock cycles
// per phase of every wave of the last launch
constexpr int kRcSlots = 8192;
__device__ unsigned long long g_rc_time[kRcSlots][10];
#define RC_CLOCK(var) var = __builtin_readcyclecounter()
#define RC_COUNT(var) var += 1
#else
#define RC_CLOCK(var)
#define RC_COUNT(var)
#endif
// imgproc.cu:348-380 between lanes: the lane holds the top-left value of a 2 x 2 block, the other three are held
// `right` lanes further on, `down` lanes further on and both.  Same products and sums, in the same order, as
// resize_pixel (frameproc.hip); lanes whose partners are outside the image or the wave get values nobody stores.
__device__ __forceinline__ void average_block(int lane, int right, int down, float (&v)[3], float (&n)[3])
{
  float v01[3], v10[3], v11[3], n01[3], n10[3], n11[3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
  {
    v01[k] = __shfl(v[k], lane + right, 64), v10[k] = __shfl(v[k], lane + down, 64), v11[k] = __shfl(v[k], lane + down + right, 64);
    n01[k] = __shfl(n[k], lane + right, 64), n10[k] = __shfl(n[k], lane + down, 64), n11[k] = __shfl(n[k], lane + down + right, 64);
  }
  const float prod = v[0] * v01[0] * v10[0] * v11[0];
  const bool whole = !(prod != prod);
  const float qnan = __builtin_nanf("");
#pragma unroll
  for (int k = 0; k < 3; ++k)
  {
    v[k] = whole ? (v[k] + v01[k] + v10[k] + v11[k]) * 0.25f : qnan;
    n[k] = whole ? (n[k] + n01[k] + n10[k] + n11[k]) * 0.25f : qnan;
  }
}

// SEG > 1 (whole volume with the surface map only): the march of every ray is cut into SEG segments of its pair
// sequence and SEG waves of the workgroup march one each -- the rule of the slab composite inside one GPU.  A wave
// replays the march's arithmetic up to its first pair without fetching anything, examines its own pairs as the
// whole march would have, and the smallest pair index that ends the march wins (tsdf_volume.cu:320-351 stops at the
// first such pair): same image, the longest ray costs a SEG-th.  The workgroup then covers 4 / SEG tiles of 8 x 8.
// tsdf_volume.cu:300-301; device.hpp:57-62 with z = 1: the direction of pixel (x, y)'s ray in the volume frame
// (one function for the ray cast and for the composite that rebuilds vertices from march times: the same instructions)
__device__ __forceinline__ void pixel_ray(const RaycastParams& p, int x, int y, float& dx, float& dy, float& dz)
{
  const float rx = ((float)x - p.cx) * p.finvx;
  const float ry = ((float)y - p.cy) * p.finvy;
  dx = dot3(p.R[0], p.R[1], p.R[2], rx, ry, 1.f);
  dy = dot3(p.R[3], p.R[4], p.R[5], rx, ry, 1.f);
  dz = dot3(p.R[6], p.R[7], p.R[8], rx, ry, 1.f);
  normalize3(dx, dy, dz);
}

// the vertex (camera frame) of the point at march time Ts on that ray, and its depth in millimetres: tsdf_volume.cu:338,
// 346 / 279-280 as raycast_kernel states them
__device__ __forceinline__ void ray_vertex(const RaycastParams& p, float dx, float dy, float dz, float Ts, float& px, float& py,
                                           float& pz, float& vx, float& vy, float& vz)
{
  px = p.t[0] + dx * Ts, py = p.t[1] + dy * Ts, pz = p.t[2] + dz * Ts;
  const float wx = px - p.t[0], wy = py - p.t[1], wz = pz - p.t[2];
  vx = dot3(p.Rinv[0], p.Rinv[1], p.Rinv[2], wx, wy, wz);
  vy = dot3(p.Rinv[3], p.Rinv[4], p.Rinv[5], wx, wy, wz);
  vz = dot3(p.Rinv[6], p.Rinv[7], p.Rinv[8], wx, wy, wz);
}
__device__ __forceinline__ unsigned short vertex_depth_mm(float vz)
{  // static_cast<ushort>(vertex.z * 1000): truncating, saturating conversion
  const float mm = vz * 1000.f;
  return (unsigned short)(mm >= 65535.f ? 65535 : (mm > 0.f ? (int)mm : 0));
}

template <bool IDX32, bool SLAB, bool SKIP, int SEG>
__global__ __launch_bounds__(256, 5) void raycast_kernel(const RaycastParams p, int tiles_x, int n_tiles, int per_xcd)
{
  static_assert(SEG == 1 || (SKIP && !SLAB && (SEG == 2 || SEG == 4)), "segments: whole volume with the surface map");
  constexpr int kNone = 0x7fffffff;
  constexpr int kSub = SEG > 1 ? 4 / SEG : 1;             // tiles of 8 x 8 pixels per workgroup
  __shared__ int seg_key[SEG > 1 ? kSub * 64 : 1];        // per pixel: the smallest pair index that ended a march so far (x 2)
  __shared__ float seg_out[SEG > 1 ? kSub * 7 * 64 : 1];  // the winner's vertex, normal and hit flag
  // XCD-aware tile order: consecutive workgroup ids go round-robin over the 8 XCDs; give each XCD a
  // contiguous run of tiles (a band of the image) so that its L2 sees one part of the volume.
  const int bid = (int)blockIdx.x;
  // per_xcd < 0: tiles in a scattered order (stride -per_xcd, coprime to the tile count): with the surface map
  // the cost of a tile is its hits, which come in patches of the image
  const int tile = per_xcd > 0 ? (bid & 7) * per_xcd + (bid >> 3) : (per_xcd < 0 ? (int)(((long long)bid * -per_xcd) % n_tiles) : bid);
  if (tile >= n_tiles)
    return;
#ifdef YAK_RAYCAST_TIMING
  unsigned long long rc_t0, rc_t1, rc_t2, rc_t3, rc_trips = 0, rc_real = 0, rc_real_cycles = 0, rc_trip0 = 0, rc_trip1 = 0;
  RC_CLOCK(rc_t0);
#endif
  extern __shared__ uint32_t map_lds[];
  // The surface map's words go to LDS; their loads are sent off here and waited for after the ray has been set up
  // (a sqrt and five correctly rounded divisions: the longest stretch of the kernel without a memory access).
  constexpr int kMapRegs = 8;  // words per thread in flight (a 512^3 volume's map: 1040 words, five a thread)
  uint32_t map_regs[kMapRegs];
  if constexpr (SKIP)
  {
#pragma unroll
    for (int j = 0; j < kMapRegs; ++j)
    {
      const int i = (int)threadIdx.x + j * 256;
      map_regs[j] = i < p.skip_words + p.cell_words ? p.skip[i] : 0u;
    }
  }
  const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  const int sub = SEG > 1 ? wave / SEG : 0, seg = SEG > 1 ? wave % SEG : 0;
  const int x = SEG > 1 ? tile_x * (8 * kSub) + sub * 8 + (lane & 7) : tile_x * 16 + (wave & 1) * 8 + (lane & 7);
  const int y = SEG > 1 ? tile_y * 8 + (lane >> 3) : tile_y * 16 + (wave >> 1) * 8 + (lane >> 3);
  const bool valid = x < p.cols && y < p.rows;  // (a lane outside the image stays until the barrier below)

  RC_CLOCK(rc_t1);
  RC_CLOCK(rc_t2);
  float dx, dy, dz;
  pixel_ray(p, x, y, dx, dy, dz);
  const float ox = p.t[0], oy = p.t[1], oz = p.t[2];

  // tsdf_volume.cu:149-165, 306-314
  const float irx = 1.f / dx, iry = 1.f / dy, irz = 1.f / dz;
  const float tbx = irx * (0.f - ox), tby = iry * (0.f - oy), tbz = irz * (0.f - oz);
  const float ttx = irx * (p.bx - ox), tty = iry * (p.by - oy), ttz = irz * (p.bz - oz);
  const float tmnx = fminf(ttx, tbx), tmny = fminf(tty, tby), tmnz = fminf(ttz, tbz);
  const float tmxx = fmaxf(ttx, tbx), tmxy = fmaxf(tty, tby), tmxz = fmaxf(ttz, tbz);
  float tmin = fmaxf(fmaxf(tmnx, tmny), fmaxf(tmnx, tmnz));
  float tmax = fminf(fminf(tmxx, tmxy), fminf(tmxx, tmxz));
  tmin = fmaxf(0.f, tmin);

  // 0 = marching, 1 = zero crossing found, 2 = finished without one
  int state = tmin < tmax && valid ? 0 : 2;

  // tsdf_volume.cu:316-318
  const float ts = p.time_step;
  tmax -= ts;
  const float sx = dx * ts, sy = dy * ts, sz = dz * ts;
  float nx = ox + dx * tmin, ny = oy + dy * tmin, nz = oz + dz * tmin;
  float tcurr = tmin;
  float hcx = 0.f, hcy = 0.f, hcz = 0.f, hnx = 0.f, hny = 0.f, hnz = 0.f;  // the bracketing samples of a hit

  // tsdf_volume.cu:320-351, kUnroll samples in flight per trip.  Slab mode: a pair is examined only if
  // its first sample lies in an owned layer; `pair` counts the pairs of the global march.
  int pair = 0;
  // this wave's pairs [k_lo, k_hi) of the march: the same split in every wave of a pixel (same arithmetic, same lane)
  int k_hi = kNone;
  if constexpr (SEG > 1)
  {
    const float span = (tmax - tmin) * __builtin_amdgcn_rcpf(ts);  // about the number of pairs inside the volume
    const int n_tot = state == 0 ? (int)fminf(fmaxf(span, 0.f), 1.0e6f) + 1 : 0;
    const int k_lo = n_tot * seg / SEG;
    k_hi = seg == SEG - 1 ? kNone : n_tot * (seg + 1) / SEG;
    // up to its first pair a wave replays what the march does between events: the time test, the rounded additions
    f32x2 pxy = { nx, ny }, pzt = { nz, tcurr };
    const f32x2 sxy = { sx, sy }, szt = { sz, ts };
    int done = 0;
    if (state == 0)
      for (; done < k_lo; ++done)
      {
        if (!(pzt.y < tmax))
        {
          state = 2;
          break;
        }
        pxy += sxy;
        pzt += szt;
      }
    nx = pxy.x, ny = pxy.y, nz = pzt.x, tcurr = pzt.y;
    pair = done;
  }
  if constexpr (SKIP)
  {
#pragma unroll
    for (int j = 0; j < kMapRegs; ++j)
    {
      const int i = (int)threadIdx.x + j * 256;
      if (i < p.skip_words + p.cell_words)
        map_lds[i] = map_regs[j];
    }
    for (int i = (int)threadIdx.x + kMapRegs * 256; i < p.skip_words + p.cell_words; i += 256)
      map_lds[i] = p.skip[i];
    if constexpr (SEG > 1)
      if ((int)threadIdx.x < kSub * 64)
        seg_key[threadIdx.x] = kNone;
    __syncthreads();
  }
  if constexpr (SEG == 1)
    if (!valid)
      return;  // (SEG > 1: the lane stays for the workgroup's later barriers, marches nothing and writes nothing)
  if (state == 0)
  {
    bool own_next = true, scratch = true, skip_next = false;
    float tsdf_next;
    // (SKIP only; held layers below 2 GiB are read through it, larger ones -- the IDX32 = false variants -- by plain loads)
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.vol), 0, IDX32 ? (uint32_t)p.vol_bytes : 0u, 0x00020000);
    if constexpr (SKIP)
      tsdf_next = fetch_tsdf_skip<SLAB, !IDX32>(p, rs, map_lds, nx, ny, nz, own_next, scratch, skip_next);
    else
      tsdf_next = fetch_tsdf<IDX32, SLAB>(p, nx, ny, nz, own_next, scratch);
    if constexpr (SKIP)
    {
      // Free space in one go.  Walk the cells of 64^3 voxels the straight ray passes through, from its entry into
      // the volume, until one is marked (a marked brick in it or within 16 voxels of it) or the ray leaves: up to
      // there every sample lies in an unmarked brick, so every pair is a pair of values >= 0 and cannot end the
      // march (tsdf_volume.cu:326-331).  Only the march's own arithmetic is replayed for those pairs -- the same
      // rounded additions of position and time, the same tcurr < tmax test per step -- two steps short of the
      // marked cell; the marched samples stay within an eighth of a voxel of the straight line (cell_jump,
      // api.hip), far inside the 16 voxels the cell bits keep clear.  Slab mode: layers that are not held count as
      // free (their samples read 0 and their pairs are another slab's).
      if (p.cell_jump && (skip_next || (SLAB && !scratch)))
      {
        // voxel-index coordinates shifted by one half: floor(u / 64) is the cell of the voxel that rint() picks
        const float zb = SLAB ? (float)p.z_base : 0.f;
        const float ux = nx * p.ivx + 0.5f, uy = ny * p.ivy + 0.5f, uz = nz * p.ivz + 0.5f - zb;
        const float gx = dx * p.ivx, gy = dy * p.ivy, gz = dz * p.ivz;  // d(u) / dt
        int cxi = clampi((int)__builtin_floorf(ux * (1.f / 64.f)), p.cell_x - 1);
        int cyi = clampi((int)__builtin_floorf(uy * (1.f / 64.f)), p.cell_y - 1);
        int czi = (int)__builtin_floorf(uz * (1.f / 64.f));
        if constexpr (!SLAB)
          czi = clampi(czi, p.cell_z - 1);
        const float igx = __builtin_amdgcn_rcpf(gx), igy = __builtin_amdgcn_rcpf(gy), igz = __builtin_amdgcn_rcpf(gz);
        const float big = 3.0e38f;
        // time at which the ray crosses into the next cell along each axis, and the time between crossings
        // (t0: the march time of the sample the walk starts from -- tmin, or a later segment's first sample)
        const float t0 = tcurr;
        float tx = gx > 0.f ? t0 + ((float)((cxi + 1) << 6) - ux) * igx : (gx < 0.f ? t0 + ((float)(cxi << 6) - ux) * igx : big);
        float ty = gy > 0.f ? t0 + ((float)((cyi + 1) << 6) - uy) * igy : (gy < 0.f ? t0 + ((float)(cyi << 6) - uy) * igy : big);
        float tz = gz > 0.f ? t0 + ((float)((czi + 1) * 64) - uz) * igz : (gz < 0.f ? t0 + ((float)(czi * 64) - uz) * igz : big);
        const float ex = 64.f * __builtin_fabsf(igx), ey = 64.f * __builtin_fabsf(igy), ez = 64.f * __builtin_fabsf(igz);
        const int stx = gx > 0.f ? 1 : -1, sty = gy > 0.f ? 1 : -1, stz = gz > 0.f ? 1 : -1;
        float t_free = t0;
        for (int guard = 0; guard < 4096; ++guard)
        {
          if ((unsigned)cxi >= (unsigned)p.cell_x || (unsigned)cyi >= (unsigned)p.cell_y)
            break;  // the ray has left the volume
          // (slab mode: the cell layer next to the held ones answers for its held neighbour -- a sample an eighth of
          // a voxel off the straight line may already lie in a held layer; cells farther out are free)
          const int czl = SLAB && czi == -1 ? 0 : (SLAB && czi == p.cell_z ? p.cell_z - 1 : czi);
          const bool held_z = (unsigned)czl < (unsigned)p.cell_z;
          if (!SLAB && !held_z)
            break;
          if (held_z)
          {
            const uint32_t c = (uint32_t)cxi + (uint32_t)p.cell_x * ((uint32_t)cyi + (uint32_t)p.cell_y * (uint32_t)czl);
            if ((map_lds[p.skip_words + (c >> 5)] >> (c & 31u)) & 1u)
              break;
          }
          // into the next cell
          if (tx <= ty && tx <= tz)
            t_free = tx, tx += ex, cxi += stx;
          else if (ty <= tz)
            t_free = ty, ty += ey, cyi += sty;
          else
            t_free = tz, tz += ez, czi += stz;
          if (!(t_free < tmax))
            break;
        }
        const float room = (fminf(t_free, tmax) - t0) / ts - 2.f;
        int steps = room > 0.f ? (int)fminf(room, 1.0e6f) : 0;
        if constexpr (SEG > 1)
          steps = min(steps, max(k_hi - pair, 0));  // (not beyond the segment: its end is examined in the march)
        f32x2 pxy = { nx, ny }, pzt = { nz, tcurr };
        const f32x2 sxy = { sx, sy }, szt = { sz, ts };
        int done = 0;
        // four steps at a time while all four pass the time test (the times rise: the fourth step's test decides),
        // then step by step -- the same additions in the same order either way
        while (done + 4 <= steps)
        {
          const f32x2 a1 = pxy + sxy, b1 = pzt + szt;
          const f32x2 a2 = a1 + sxy, b2 = b1 + szt;
          const f32x2 a3 = a2 + sxy, b3 = b2 + szt;
          if (!(b3.y < tmax))
            break;
          pxy = a3 + sxy, pzt = b3 + szt;
          done += 4;
        }
        for (; done < steps; ++done)
        {
          if (!(pzt.y < tmax))
          {
            state = 2;
            break;
          }
          pxy += sxy;
          pzt += szt;
        }
        nx = pxy.x, ny = pxy.y, nz = pzt.x, tcurr = pzt.y;
        pair += done;
        if (done > 0)
          skip_next = true;  // the sample now stood on reads as +0: unmarked brick, or a layer that is not held
        if constexpr (SLAB)
        {
          const int zn = clampi((int)__builtin_rintf(nz * p.ivz), p.Z - 1);
          own_next = zn >= p.z_own0 && zn < p.z_own1;
        }
      }
    }
    RC_CLOCK(rc_t2);
    bool look_first = true;  // (wave-uniform)
    while (state == 0)
    {
      RC_COUNT(rc_trips);
      RC_CLOCK(rc_trip0);
      if constexpr (SEG > 1)
      {
        // an earlier segment's march has ended at a pair before this one: nothing here can win any more
        if (__hip_atomic_load(&seg_key[sub * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 2 * pair)
        {
          state = 2;
          break;
        }
      }
      constexpr int kUnroll = SKIP ? YAK_RAYCAST_UNROLL_SKIP : kUnrollPlain;
      if constexpr (SKIP)
      {
        // A look at the bricks before anything else.  Inside a marked cell most samples still lie in unmarked bricks
        // (six trips in seven fetch nothing): if none of this trip's samples, in any lane still marching, lies in a
        // marked brick and the sample stood on is not negative, every value of the trip reads as +0 and no pair can end
        // the march (tsdf_volume.cu:326-331 needs a negative sample) -- only the march's own arithmetic is left, the
        // time test and the rounded additions.  (Measured: working out the voxel offsets here as well, for the full
        // trip to reuse, costs the quiet trips more than it saves the others.)
        float bx = nx, by = ny, bz = nz;
        bool marked = tsdf_next < 0.f;
        bool last_in = true, last_owned = true;  // (slab mode: a sample outside the held layers reads 0, like a skipped one)
        if (look_first)
        {
#pragma unroll
          for (int k = 0; k < kUnroll; ++k)
          {
            bx += sx;
            by += sy;
            bz += sz;
            marked = marked | sample_brick_marked<SLAB>(p, map_lds, bx, by, bz, last_in, last_owned);
          }
        }
        // (a wave that has just fetched does not look first: beside a surface one trip after the other fetches; it
        // looks again after a full trip that found every sample of every lane in an unmarked brick)
        if (look_first && __ballot(marked) == 0ull)
        {
          // the times of the trip's steps: rising, so the last test decides them all
          float tl = tcurr;
#pragma unroll
          for (int k = 0; k < kUnroll - 1; ++k)
            tl += ts;
          if (tl < tmax && (SEG == 1 || pair + kUnroll <= k_hi))
            tcurr = tl + ts, pair += kUnroll;
          else
          {
#pragma unroll
            for (int k = 0; k < kUnroll; ++k)
            {
              const bool late = !(tcurr < tmax) | (SEG > 1 && pair >= k_hi);
              const bool on = (state == 0) & !late;
              tcurr = on ? tcurr + ts : tcurr;
              pair += on ? 1 : 0;
              state = (state == 0 && late) ? 2 : state;
            }
          }
          nx = bx, ny = by, nz = bz;
          tsdf_next = 0.f, skip_next = last_in;  // an unmarked brick's sample: +0, not fetched (not held: 0 for good)
          own_next = last_owned;
          continue;
        }
        look_first = false;
      }
      float qx[kUnroll], qy[kUnroll], qz[kUnroll], f[kUnroll];
      bool own[kUnroll], held[kUnroll], skip[kUnroll];
      float ax = nx, ay = ny, az = nz;
#pragma unroll
      for (int k = 0; k < kUnroll; ++k)
      {
        ax += sx;
        ay += sy;
        az += sz;
        qx[k] = ax, qy[k] = ay, qz[k] = az;
        own[k] = true, held[k] = true, skip[k] = false;
        if constexpr (SKIP)
          f[k] = fetch_tsdf_skip<SLAB, !IDX32>(p, rs, map_lds, ax, ay, az, own[k], held[k], skip[k]);
        else
          f[k] = fetch_tsdf<IDX32, SLAB>(p, ax, ay, az, own[k], held[k]);
      }
      if constexpr (SKIP)
      {
        bool all_skipped = true;
#pragma unroll
        for (int k = 0; k < kUnroll; ++k)
          all_skipped = all_skipped & (skip[k] | (SLAB && !held[k]));
        look_first = __ballot(!all_skipped) == 0ull;
      }
      if constexpr (SKIP)
      {
        // A skipped sample reads as +0 and is >= 0; its value matters only beside a negative one (tsdf_volume.cu:326-331:
        // a pair ends the march only if one of its samples is negative).  A lane with such a pair in this trip fetches
        // all the trip's samples for good, in one go: one wait for the wave.
        bool beside_negative = (skip_next & (f[0] < 0.f)) | (skip[0] & (tsdf_next < 0.f));
#pragma unroll
        for (int k = 1; k < kUnroll; ++k)
          beside_negative = beside_negative | (skip[k - 1] & (f[k] < 0.f)) | (skip[k] & (f[k - 1] < 0.f));
        if (beside_negative)
        {
          bool o = true, h = true;
          tsdf_next = fetch_tsdf<IDX32, SLAB>(p, nx, ny, nz, o, h);
          skip_next = false;
#pragma unroll
          for (int k = 0; k < kUnroll; ++k)
          {
            f[k] = fetch_tsdf<IDX32, SLAB>(p, qx[k], qy[k], qz[k], o, h);
            skip[k] = false;
          }
        }
        // The pairs in order, without branches (the lanes of a wave end their marches at different pairs): `ended` is
        // the first pair that ends the march, by running out of time (2), a zero crossing (1) or the - to + abort (3).
        float fc = tsdf_next;
        bool own_c = own_next;
        int ended = -1;
#pragma unroll
        for (int k = 0; k < kUnroll; ++k)
        {
          // (& and |, not && and ||: masks, no short cuts)
          const bool marching = state == 0;
          const bool late = !(tcurr < tmax) | (SEG > 1 && pair >= k_hi);  // (a segment's end: "no event", like the ray's)
          const bool abort = own_c & (fc < 0.f) & (f[k] > 0.f), crossing = own_c & (fc > 0.f) & (f[k] < 0.f);
          const int verdict = late ? 2 : (abort ? 3 : (crossing ? 1 : 0));
          const bool on = marching & !(late | abort | crossing);
          tcurr = on ? tcurr + ts : tcurr;
          pair += on ? 1 : 0;
          ended = (marching & !on) ? k : ended;
          state = marching ? verdict : state;
          fc = f[k], own_c = own[k];
        }
        if (state == 1)
        {
          // the bracketing samples of the hit: pair `ended` = samples ended - 1 and ended of this trip
          hcx = nx, hcy = ny, hcz = nz;
#pragma unroll
          for (int k = 0; k < kUnroll; ++k)
          {
            hcx = ended == k + 1 ? qx[k] : hcx, hcy = ended == k + 1 ? qy[k] : hcy, hcz = ended == k + 1 ? qz[k] : hcz;
            hnx = ended == k ? qx[k] : hnx, hny = ended == k ? qy[k] : hny, hnz = ended == k ? qz[k] : hnz;
          }
        }
      }
      else
      {
        // pair by pair, as the loads arrive (this variant waits for memory, not for the vector unit)
        float cx = nx, cy = ny, cz = nz, fc = tsdf_next;
        bool own_c = own_next;
#pragma unroll
        for (int k = 0; k < kUnroll; ++k)
        {
          if (state == 0)
          {
            if (!(tcurr < tmax))
              state = 2;
            else if (own_c && fc < 0.f && f[k] > 0.f)
              state = 3;
            else if (own_c && fc > 0.f && f[k] < 0.f)
            {
              state = 1;
              hcx = cx, hcy = cy, hcz = cz;
              hnx = qx[k], hny = qy[k], hnz = qz[k];
            }
            else
            {
              tcurr += ts;
              ++pair;
            }
          }
          cx = qx[k], cy = qy[k], cz = qz[k], fc = f[k], own_c = own[k];
        }
      }
      nx = qx[kUnroll - 1], ny = qy[kUnroll - 1], nz = qz[kUnroll - 1];
      tsdf_next = f[kUnroll - 1], own_next = own[kUnroll - 1], skip_next = skip[kUnroll - 1];
#ifdef YAK_RAYCAST_TIMING
      if constexpr (SKIP)
      {
        bool fetched = false;
#pragma unroll
        for (int k = 0; k < kUnroll; ++k)
          fetched = fetched || !skip[k];
        RC_CLOCK(rc_trip1);
        if (__ballot(fetched) != 0ull)  // the wave waited for memory in this trip
          rc_real += 1, rc_real_cycles += rc_trip1 - rc_trip0;
      }
#endif
    }
  }

  RC_CLOCK(rc_t3);
  const float qnan = __builtin_nanf("");
  float vx = qnan, vy = qnan, vz = qnan, vw = qnan, n0 = qnan, n1 = qnan, n2 = qnan, n3 = qnan;
  bool hit = false, deferred = false;
  float ts_hit = qnan;  // the march time of the refined point, for the composite over slabs
  bool winner = true;
  if constexpr (SEG > 1)
  {
    // the pair that ended this wave's march against the other segments' of the same pixel: the smallest ends the ray
    const int key = (state == 1 || state == 3) ? 2 * pair : kNone;
    if (key != kNone)
      atomicMin(&seg_key[sub * 64 + lane], key);
    __syncthreads();
    winner = valid && seg_key[sub * 64 + lane] == key && (key != kNone || seg == 0);  // (no event anywhere: segment 0 reports that)
  }
  if (state == 1 && winner)
  {
    // tsdf_volume.cu:333-348
    bool held_bracket = true, held_normal = true;
    const float Ft = interpolate<IDX32, SLAB>(p, hcx * p.ivx, hcy * p.ivy, hcz * p.ivz, held_bracket);
    const float Ftdt = interpolate<IDX32, SLAB>(p, hnx * p.ivx, hny * p.ivy, hnz * p.ivz, held_bracket);
    const float Ts = tcurr - (ts * Ft) / (Ftdt - Ft);
    float px, py, pz, hx, hy, hz;
    ray_vertex(p, dx, dy, dz, Ts, px, py, pz, hx, hy, hz);
    float gx, gy, gz;
    surface_gradient<IDX32, SLAB>(p, px, py, pz, gx, gy, gz, held_normal);
    const float prod = gx * gy * gz;
    deferred = SLAB && !held_normal;
    if (deferred || !(prod != prod))
    {
      hit = true;
      ts_hit = Ts;
      vx = hx, vy = hy, vz = hz;
      vw = 0.f;
      if (deferred)
        n0 = px, n1 = py, n2 = pz;  // the point itself, for the slab that holds its surroundings
      else
      {
        n0 = dot3(p.Rinv[0], p.Rinv[1], p.Rinv[2], gx, gy, gz);
        n1 = dot3(p.Rinv[3], p.Rinv[4], p.Rinv[5], gx, gy, gz);
        n2 = dot3(p.Rinv[6], p.Rinv[7], p.Rinv[8], gx, gy, gz);
      }
      n3 = 0.f;
    }
  }

  if constexpr (SEG > 1)
  {
    // the winner's pixel goes to the tile's first wave, which writes the tile (and averages it for the pyramid)
    float* out = &seg_out[sub * 7 * 64 + lane];
    if (winner)
    {
      out[0 * 64] = vx, out[1 * 64] = vy, out[2 * 64] = vz, out[3 * 64] = n0, out[4 * 64] = n1, out[5 * 64] = n2;
      out[6 * 64] = hit ? 1.f : 0.f;
    }
    __syncthreads();
    if (seg != 0 || !valid)
      return;
    vx = out[0 * 64], vy = out[1 * 64], vz = out[2 * 64], n0 = out[3 * 64], n1 = out[4 * 64], n2 = out[5 * 64];
    hit = out[6 * 64] != 0.f;
    vw = hit ? 0.f : qnan, n3 = hit ? 0.f : qnan;
  }
  // tsdf_volume.cu:298 (defaults) and :346-347 / :279-280 (hit)
  float4* nrow = (float4*)(p.normals + (size_t)y * p.normals_pitch);
  nrow[x] = make_float4(n0, n1, n2, n3);
  if (p.points)
  {
    float4* prow = (float4*)(p.points + (size_t)y * p.points_pitch);
    prow[x] = make_float4(vx, vy, vz, vw);
  }
  if constexpr (!SLAB)
  {
    // The pyramid of the maps (kinfu.cpp:294-306 calls resizePointsNormals level by level after the ray cast) from
    // the registers of the wave: it holds 8 x 8 pixels, lane = 8 * (y & 7) + (x & 7), so every 2 x 2 block of the
    // full-size maps and every 2 x 2 block of the half-size ones lies in one wave.  (All lanes inside the image
    // are here together: the march is left lane by lane, but its loop ends before this point.)
    if (p.pyr_points[0])
    {
      float v[3] = { vx, vy, vz }, n[3] = { n0, n1, n2 };
      average_block(lane, 1, 8, v, n);
      const int rows1 = p.rows >> 1, cols1 = p.cols >> 1;  // imgproc.cu:386-387
      if ((lane & 9) == 0 && (y >> 1) < rows1 && (x >> 1) < cols1)
      {
        ((float4*)(p.pyr_points[0] + (size_t)(y >> 1) * p.pyr_points_pitch[0]))[x >> 1] = make_float4(v[0], v[1], v[2], 0.f);
        ((float4*)(p.pyr_normals[0] + (size_t)(y >> 1) * p.pyr_normals_pitch[0]))[x >> 1] = make_float4(n[0], n[1], n[2], 0.f);
      }
      if (p.pyr_points[1])
      {
        average_block(lane, 2, 16, v, n);
        if ((lane & 27) == 0 && (y >> 2) < (rows1 >> 1) && (x >> 2) < (cols1 >> 1))
        {
          ((float4*)(p.pyr_points[1] + (size_t)(y >> 2) * p.pyr_points_pitch[1]))[x >> 2] = make_float4(v[0], v[1], v[2], 0.f);
          ((float4*)(p.pyr_normals[1] + (size_t)(y >> 2) * p.pyr_normals_pitch[1]))[x >> 2] = make_float4(n[0], n[1], n[2], 0.f);
        }
      }
    }
  }
  if (p.depth)
  {
    const unsigned short mm_out = hit ? vertex_depth_mm(vz) : (unsigned short)0;
    unsigned short* drow = (unsigned short*)(p.depth + (size_t)y * p.depth_pitch);
    drow[x] = mm_out;
  }
#ifdef YAK_RAYCAST_TIMING
  {
    const unsigned long long rc_t4 = __builtin_readcyclecounter();
    const unsigned long long hits = __builtin_popcountll(__ballot(state == 1));
    // the march loop is left lane by lane: the wave's figures are those of the lane that stayed longest
    for (int off = 32; off >= 1; off >>= 1)
    {
      rc_trips = max(rc_trips, (unsigned long long)__shfl_xor((unsigned)rc_trips, off, 64));
      rc_real = max(rc_real, (unsigned long long)__shfl_xor((unsigned)rc_real, off, 64));
      rc_real_cycles = max(rc_real_cycles, (unsigned long long)__shfl_xor((unsigned)rc_real_cycles, off, 64));
    }
    if (lane == 0)  // (a wave whose lane 0 is outside the image reports nothing)
    {
      unsigned long long* row = g_rc_time[((int)blockIdx.x * 4 + wave) & (kRcSlots - 1)];
      row[0] = rc_t1 - rc_t0, row[1] = rc_t2 - rc_t1, row[2] = rc_t3 - rc_t2, row[3] = rc_t4 - rc_t3;
      row[4] = rc_trips, row[5] = 1ull, row[6] = hits, row[7] = rc_t4 - rc_t0, row[8] = rc_real, row[9] = rc_real_cycles;
    }
  }
#endif
  if constexpr (SLAB)
  {
    // the pair that ended the march here (zero crossing = 1, abort = 3), or "none": 2 * k + (1 if the
    // normal is still to be computed where the refined point lies)
    int* krow = (int*)(p.keys + (size_t)y * p.keys_pitch);
    const int key = (state == 1 || state == 3) ? 2 * pair + (deferred ? 1 : 0) : 0x7fffffff;
    krow[x] = key;
    if (p.key_out)  // yakhip_raycast_sharded: what its composite exchanges
    {
      p.key_out[(size_t)y * p.cols + x] = key;
      p.ts_out[(size_t)y * p.cols + x] = ts_hit;
    }
  }
}

// The composite of the slab ray casts (comm.hip), last step: pixel i's winner sent {march time, normal}; the refined
// point is ray origin + direction * time and the vertex its image in the camera frame, by the instructions the ray
// cast itself ran (pixel_ray, ray_vertex), the depth follows from the vertex.  An odd key (the normal is still to be
// computed by the slab that holds the point: raycast_resolve_kernel) gets the point in place of the normal, as
// raycast_kernel leaves it.  A NaN time: the march ended without a hit (the - to + abort, a NaN normal).
__global__ __launch_bounds__(256) void composite_unpack_kernel(const RaycastParams p, const int32_t* best, const uint32_t* packed)
{
  const int i = (int)(blockIdx.x * 256u + threadIdx.x);
  const int n = p.rows * p.cols;
  if (i >= n)
    return;
  const int y = i / p.cols, x = i - y * p.cols;
  const int32_t b = best[i];
  const float qnan = __builtin_nanf("");
  const float Ts = __builtin_bit_cast(float, packed[i]);
  float4 v = make_float4(qnan, qnan, qnan, qnan), nr = v;
  unsigned short mm = 0;
  if (b != 0x7fffffff && !(Ts != Ts))
  {
    float dx, dy, dz, px, py, pz, vx, vy, vz;
    pixel_ray(p, x, y, dx, dy, dz);
    ray_vertex(p, dx, dy, dz, Ts, px, py, pz, vx, vy, vz);
    v = make_float4(vx, vy, vz, 0.f);
    nr = (b & 1) ? make_float4(px, py, pz, 0.f)
                 : make_float4(__builtin_bit_cast(float, packed[(size_t)n + i]), __builtin_bit_cast(float, packed[2 * (size_t)n + i]),
                               __builtin_bit_cast(float, packed[3 * (size_t)n + i]), 0.f);
    mm = vertex_depth_mm(vz);
  }
  if (p.points)
    ((float4*)(p.points + (size_t)y * p.points_pitch))[x] = v;
  ((float4*)(p.normals + (size_t)y * p.normals_pitch))[x] = nr;
  if (p.depth)
    ((unsigned short*)(p.depth + (size_t)y * p.depth_pitch))[x] = mm;
  ((int*)(p.keys + (size_t)y * p.keys_pitch))[x] = b;
}

// Slab mode, after the composite: the pixels whose key is odd hold a refined point (volume frame, in the
// normals image) instead of a normal.  The slab that owns the point's layer computes the normal as the
// whole-volume kernel would have (tsdf_volume.cu:340-348: a NaN normal discards the hit), writes it and
// clears the key's low bit; the other slabs leave the pixel alone.
template <bool IDX32>
__global__ __launch_bounds__(256) void raycast_resolve_kernel(const RaycastParams p)
{
  const int x = (int)(blockIdx.x * 64 + (threadIdx.x & 63)), y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (x >= p.cols || y >= p.rows)
    return;
  int* krow = (int*)(p.keys + (size_t)y * p.keys_pitch);
  const int key = krow[x];
  if (key == 0x7fffffff || (key & 1) == 0)
    return;
  float4* nrow = (float4*)(p.normals + (size_t)y * p.normals_pitch);
  const float4 q = nrow[x];
  // the layer the point lies in, clamped; a point at infinity or NaN reads nothing (every tap is out of
  // bounds) and goes to the slab of layer 0 resp. Z - 1
  const float lz = __builtin_floorf(q.z * p.ivz);
  const int zp = !(lz >= 0.f) ? 0 : (lz >= (float)(p.Z - 1) ? p.Z - 1 : (int)lz);
  if (zp < p.z_own0 || zp >= p.z_own1)
    return;
  float gx, gy, gz;
  bool held = true;
  surface_gradient<IDX32, true>(p, q.x, q.y, q.z, gx, gy, gz, held);
  const float prod = gx * gy * gz;
  if (!(prod != prod))
    nrow[x] = make_float4(dot3(p.Rinv[0], p.Rinv[1], p.Rinv[2], gx, gy, gz), dot3(p.Rinv[3], p.Rinv[4], p.Rinv[5], gx, gy, gz),
                          dot3(p.Rinv[6], p.Rinv[7], p.Rinv[8], gx, gy, gz), 0.f);
  else
  {
    const float qnan = __builtin_nanf("");
    nrow[x] = make_float4(qnan, qnan, qnan, qnan);
    if (p.points)
      ((float4*)(p.points + (size_t)y * p.points_pitch))[x] = make_float4(qnan, qnan, qnan, qnan);
    if (p.depth)
      ((unsigned short*)(p.depth + (size_t)y * p.depth_pitch))[x] = 0;
  }
  krow[x] = key - 1;
}

// kfusion::device::extractNormals (tsdf_volume.cu:672-763, 782-797): normals of extracted cloud points.
// points are in the frame the cloud was extracted into (aff = p.R, p.t); the gradient is taken in the
// volume frame and rotated back.  No counterpart of raycast's NaN test: a NaN gradient stays NaN.
template <bool IDX32>
__global__ __launch_bounds__(256) void extract_normals_kernel(const RaycastParams p, const float4* pts, float4* out,
                                                              unsigned long long n)
{
  const unsigned long long idx = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
  if (idx >= n)
    return;
  const float qnan = __builtin_nanf("");
  float nx = qnan, ny = qnan, nz = qnan;
  const float4 q = pts[idx];
  // :715  point = Rinv * (tr(p) - aff.t)
  const float wx = q.x - p.t[0], wy = q.y - p.t[1], wz = q.z - p.t[2];
  const float px = dot3(p.Rinv[0], p.Rinv[1], p.Rinv[2], wx, wy, wz), py = dot3(p.Rinv[3], p.Rinv[4], p.Rinv[5], wx, wy, wz),
              pz = dot3(p.Rinv[6], p.Rinv[7], p.Rinv[8], wx, wy, wz);
  // :694-701, 718  nearest voxel (round half even) at least 2 away from every face; compared as floats so
  // that NaN and huge coordinates fail the test instead of overflowing a conversion
  const float gx = __builtin_rintf(px * p.ivx), gy = __builtin_rintf(py * p.ivy), gz = __builtin_rintf(pz * p.ivz);
  if (gx > 1.f && gy > 1.f && gz > 1.f && gx < (float)(p.X - 2) && gy < (float)(p.Y - 2) && gz < (float)(p.Z - 2))
  {
    bool held = true;
    const float Fx1 = interpolate<IDX32, false>(p, (px + p.gdx) * p.ivx, py * p.ivy, pz * p.ivz, held);
    const float Fx2 = interpolate<IDX32, false>(p, (px - p.gdx) * p.ivx, py * p.ivy, pz * p.ivz, held);
    const float Fy1 = interpolate<IDX32, false>(p, px * p.ivx, (py + p.gdy) * p.ivy, pz * p.ivz, held);
    const float Fy2 = interpolate<IDX32, false>(p, px * p.ivx, (py - p.gdy) * p.ivy, pz * p.ivz, held);
    const float Fz1 = interpolate<IDX32, false>(p, px * p.ivx, py * p.ivy, (pz + p.gdz) * p.ivz, held);
    const float Fz2 = interpolate<IDX32, false>(p, px * p.ivx, py * p.ivy, (pz - p.gdz) * p.ivz, held);
    const float dx = (Fx1 - Fx2) / p.gdx, dy = (Fy1 - Fy2) / p.gdy, dz = (Fz1 - Fz2) / p.gdz;
    // :753  n = normalized(aff.R * n)
    nx = dot3(p.R[0], p.R[1], p.R[2], dx, dy, dz);
    ny = dot3(p.R[3], p.R[4], p.R[5], dx, dy, dz);
    nz = dot3(p.R[6], p.R[7], p.R[8], dx, dy, dz);
    normalize3(nx, ny, nz);
  }
  out[idx] = make_float4(nx, ny, nz, 0.f);
}
// yakhip_surface_map_rebuild: the surface map of a slab from its voxels.  One workgroup per brick: thread t reads
// the 16 voxels of row (y = t & 15, z = t >> 4) of the brick; a voxel is negative if its half has the sign bit
// and a non-zero magnitude.
__global__ __launch_bounds__(256) void surface_scan_kernel(const uint32_t* vol, int X, int Y, int layers, int bricks_x,
                                                           int bricks_y, int bricks_z, uint32_t* map, uint32_t brick_words)
{
  const uint32_t b = blockIdx.x;
  const int bx = (int)(b % (uint32_t)bricks_x), by = (int)((b / (uint32_t)bricks_x) % (uint32_t)bricks_y),
            bz = (int)(b / ((uint32_t)bricks_x * (uint32_t)bricks_y));
  const int y = by * 16 + ((int)threadIdx.x & 15), z = bz * 16 + ((int)threadIdx.x >> 4);
  bool neg = false;
  if (y < Y && z < layers)
  {
    const uint32_t* row = vol + ((size_t)z * (size_t)Y + (size_t)y) * (size_t)X;
    for (int x = bx * 16; x < min(bx * 16 + 16, X); ++x)
    {
      const uint32_t w = row[x];
      neg = neg || ((w & 0x8000u) && (w & 0x7fffu));
    }
  }
  __shared__ int any;
  if (threadIdx.x == 0)
    any = 0;
  __syncthreads();
  if (__builtin_amdgcn_ballot_w64(neg) != 0 && (threadIdx.x & 63u) == 0)
    any = 1;
  __syncthreads();
  if (threadIdx.x == 0 && any)
  {
    atomicOr(map + (b >> 5), 1u << (b & 31u));
    surface_mark_cells(map, brick_words, bricks_x, bricks_y, bricks_z, bx, by, bz);
  }
}
}  // namespace

#ifdef YAK_RAYCAST_TIMING
// sums over the waves of the last launch (entry 7: the longest wave); reset clears the rows
extern "C" int yakhip_debug_raycast_timing(unsigned long long out[10], int reset)
{
  static unsigned long long rows[kRcSlots][10];
  void* at = nullptr;
  if (hipDeviceSynchronize() != hipSuccess || hipGetSymbolAddress(&at, HIP_SYMBOL(g_rc_time)) != hipSuccess ||
      hipMemcpy(rows, at, sizeof(rows), hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  for (int k = 0; k < 10; ++k)
    out[k] = 0;
  for (int w = 0; w < kRcSlots; ++w)
    for (int k = 0; k < 10; ++k)
      out[k] = k == 7 ? (rows[w][k] > out[k] ? rows[w][k] : out[k]) : out[k] + rows[w][k];
  if (reset && hipMemset(at, 0, sizeof(rows)) != hipSuccess)
    return -1;
  return 0;
}
// the rows themselves (kRcSlots x 10), for a closer look at the waves that take longest
extern "C" int yakhip_debug_raycast_rows(unsigned long long* out, int max_rows)
{
  void* at = nullptr;
  const int n = max_rows < kRcSlots ? max_rows : kRcSlots;
  if (hipDeviceSynchronize() != hipSuccess || hipGetSymbolAddress(&at, HIP_SYMBOL(g_rc_time)) != hipSuccess ||
      hipMemcpy(out, at, (size_t)n * 10 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  return n;
}
#endif
int launch_surface_scan(yakhip_ctx* ctx, const uint32_t* vol, int X, int Y, int layers, uint32_t* map)
{
  const int bx = (X + 15) >> 4, by = (Y + 15) >> 4, bz = (layers + 15) >> 4;
  const unsigned long long bricks = (unsigned long long)bx * by * bz;
  if (bricks > 0x7fffffffull)
    return fail(YAKHIP_ERANGE, "surface_map_rebuild: too many bricks");
  const size_t brick_words = surface_brick_words(X, Y, layers);
  YAKHIP_TRY(hipMemsetAsync(map, 0, (brick_words + surface_cell_words(X, Y, layers)) * 4u, ctx->stream));
  hipLaunchKernelGGL(surface_scan_kernel, dim3((unsigned)bricks), dim3(256), 0, ctx->stream, vol, X, Y, layers, bx, by, bz, map,
                     (uint32_t)brick_words);
  return check(hipGetLastError(), "surface_scan_kernel launch");
}

int launch_extract_normals(yakhip_ctx* ctx, const RaycastParams& p, const void* points, void* normals, unsigned long long n)
{
  if (n == 0)
    return 0;
  const unsigned long long blocks = (n + 255ull) / 256ull;
  if (blocks > 0x7fffffffull)
    return fail(YAKHIP_ERANGE, "extract_normals: too many points");
  const bool idx32 = (unsigned long long)p.X * (unsigned long long)p.Y * (unsigned long long)p.Z <= (1ull << 30);
  if (idx32)
    hipLaunchKernelGGL((extract_normals_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, p,
                       (const float4*)points, (float4*)normals, n);
  else
    hipLaunchKernelGGL((extract_normals_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, p,
                       (const float4*)points, (float4*)normals, n);
  return check(hipGetLastError(), "extract_normals_kernel launch");
}

int launch_raycast_resolve(yakhip_ctx* ctx, const RaycastParams& p)
{
  const dim3 grid((unsigned)((p.cols + 63) / 64), (unsigned)((p.rows + 3) / 4));
  if (grid.y > 65535u)
    return fail(YAKHIP_ERANGE, "raycast_resolve: image too tall");
  const bool idx32 = (unsigned long long)p.X * (unsigned long long)p.Y * (unsigned long long)p.z_held <= (1ull << 30);
  if (idx32)
    hipLaunchKernelGGL((raycast_resolve_kernel<true>), grid, dim3(256), 0, ctx->stream, p);
  else
    hipLaunchKernelGGL((raycast_resolve_kernel<false>), grid, dim3(256), 0, ctx->stream, p);
  return check(hipGetLastError(), "raycast_resolve_kernel launch");
}

int launch_composite_unpack(yakhip_ctx* ctx, const RaycastParams& p, const int32_t* best, const uint32_t* packed)
{
  const unsigned blocks = (unsigned)(((long long)p.rows * p.cols + 255) / 256);
  hipLaunchKernelGGL(composite_unpack_kernel, dim3(blocks), dim3(256), 0, ctx->stream, p, best, packed);
  return check(hipGetLastError(), "composite_unpack_kernel launch");
}

int launch_raycast(yakhip_ctx* ctx, const RaycastParams& p)
{
  const bool slab = p.keys != nullptr;
  // the surface map travels in LDS: volumes whose map is too large for that are marched without it
  // (and it does its index arithmetic in 24 bits: fetch_tsdf_skip)
  // (held layers of 2 GiB and more take the variants that index in 64 bits -- IDX32 = false -- whose sample fetches are
  // plain loads; the voxel index itself must fit 32 bits)
  const unsigned long long held_voxels = (unsigned long long)p.X * (unsigned long long)p.Y * (unsigned long long)p.z_held;
  const bool skip = p.skip != nullptr && p.skip_words + p.cell_words <= kMaxSkipWords && held_voxels <= 0xffffffffull &&
                    p.X < (1 << 24) && p.Y < (1 << 24) && (long long)p.Y * p.z_held <= (1ll << 24) && ctx->surface_skip;
  // segments per ray (raycast_kernel, SEG): whole volume with the surface map only
  const int seg = skip && !slab ? (ctx->raycast_segments > 0 ? ctx->raycast_segments : kDefaultSegments) : 1;
  const int tile_w = seg == 1 ? 16 : 32 / seg, tile_h = seg == 1 ? 16 : 8;  // pixels per workgroup
  const int tiles_x = (p.cols + tile_w - 1) / tile_w, tiles_y = (p.rows + tile_h - 1) / tile_h;
  const long long n_tiles_ll = (long long)tiles_x * tiles_y;
  if (n_tiles_ll > 0x7ffffff0ll / 8)
    return fail(YAKHIP_ERANGE, "raycast: image too large");
  const int n_tiles = (int)n_tiles_ll;
  const int per_xcd = (n_tiles + 7) / 8;
  const dim3 grid((unsigned)(per_xcd * 8));
  // (32-bit byte offsets: up to 2^30 voxels by plain loads, below 2^29 -- 2 GiB -- where the samples come through the
  // buffer descriptor, whose skipped lanes pass the offset 2^31)
  const bool idx32 = (unsigned long long)p.X * (unsigned long long)p.Y * (unsigned long long)(slab ? p.z_held : p.Z) <=
                     (skip ? (1ull << 29) - 1ull : (1ull << 30));

  std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
  int rc = timing_begin(ctx, &ev);
  if (rc != 0)
    return rc;
  const size_t lds = skip ? (size_t)(p.skip_words + p.cell_words) * 4u : 0;
#define YAK_RAYCAST_LAUNCH(I, S)                                                                                              \
  do                                                                                                                         \
  {                                                                                                                          \
    if (skip && !S && seg == 4)                                                                                              \
      hipLaunchKernelGGL((raycast_kernel<I, false, true, 4>), grid, dim3(256), lds, ctx->stream, p, tiles_x, n_tiles, per_xcd_arg); \
    else if (skip && !S && seg == 2)                                                                                         \
      hipLaunchKernelGGL((raycast_kernel<I, false, true, 2>), grid, dim3(256), lds, ctx->stream, p, tiles_x, n_tiles, per_xcd_arg); \
    else if (skip)                                                                                                           \
      hipLaunchKernelGGL((raycast_kernel<I, S, true, 1>), grid, dim3(256), lds, ctx->stream, p, tiles_x, n_tiles, per_xcd_arg);    \
    else                                                                                                                     \
      hipLaunchKernelGGL((raycast_kernel<I, S, false, 1>), grid, dim3(256), lds, ctx->stream, p, tiles_x, n_tiles, per_xcd_arg);   \
  } while (0)
  // XCD bands pay while the march fetches (each L2 sees one part of the volume: 0.057 against 0.070 ms along z);
  // with the surface map a tile costs its hits, which come in patches of the image, and plain image order spreads
  // them best (0.056-0.066 ms against 0.064-0.089 in bands, 0.060-0.072 scattered)
  const int bands = ctx->raycast_bands >= 0 ? ctx->raycast_bands : (skip ? 0 : 1);
  int per_xcd_arg = bands == 1 ? per_xcd : 0;
  if (bands >= 2)
  {
    int stride = bands;
    auto gcd = [](int a, int b) { while (b) { const int r = a % b; a = b; b = r; } return a; };
    while (gcd(stride, n_tiles) != 1)
      ++stride;
    per_xcd_arg = -stride;
  }
  if (slab && idx32)
    YAK_RAYCAST_LAUNCH(true, true);
  else if (slab)
    YAK_RAYCAST_LAUNCH(false, true);
  else if (idx32)
    YAK_RAYCAST_LAUNCH(true, false);
  else
    YAK_RAYCAST_LAUNCH(false, false);
#undef YAK_RAYCAST_LAUNCH
  if ((rc = timing_end(ctx, ev)) != 0)
    return rc;
  return check(hipGetLastError(), "raycast_kernel launch");
}
}  // namespace yakhip
