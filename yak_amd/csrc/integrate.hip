// integrate.hip -- TsdfVolume::integrate for gfx950 (MI355X), hand-written HIP.
//
// Replaces kfusion::device::integrate + TsdfIntegrator (reference yak/src/cuda/tsdf_volume.cu:51-140)
// and clear_volume (:14-40).  Same per-voxel results (bit for bit against the scalar oracle),
// different decomposition:
//
//   reference (CUDA)                         here (CDNA4)
//   ---------------------------------------  ----------------------------------------------------
//   thread = one (x,y) column, serial         lane = VX (4) consecutive x voxels = one 16-byte
//   512-step z loop, 4-byte ld/st             global_load/store_dwordx4; a wave64 covers 1 KiB of
//                                             one x-row run per instruction
//   grid = 1024 blocks over (x,y)             grid = (plane tiles) x (z chunks)
//   pose/intrinsics in a by-value struct      the same, read once into SGPRs (wave-uniform)
//   dists through a texture reference         per-pixel band thresholds precomputed once per frame
//                                             (8 B/pixel, L2-resident), half image only in the band
//   3 approximate divides + sqrt per voxel    1 v_rcp + Markstein divides, no sqrt outside the band
//   ld/st.global.cs                           non-temporal load/store, next trip's voxels prefetched
//
// The kernel is VALU-issue bound before it is HBM bound on this chip (one wave64 VALU instruction
// per ~4 cycles per SIMD, measured), so the arithmetic is organised to issue as few instructions
// per voxel as bit-exactness allows; DESIGN.md ("integrate kernel") has the accounting.
//
// Bit parity with the scalar oracle:
//  * the reference accumulates the camera-space position with one float addition per z step
//    (vc += zstep, tsdf_volume.cu:76), so vc(z) != vc0 + z*zstep in floating point.  A workgroup that
//    starts at layer zs REPLAYS the zs additions before its first voxel;
//  * all FMAs are explicit, the file is compiled with -ffp-contract=off;
//  * every quotient is the correctly rounded IEEE quotient (see div_by below);
//  * the square root is only needed, and only evaluated (correctly rounded), inside the truncation
//    band; outside it two sufficient tests with explicit rounding slack decide the voxel.
#include <cmath>

#include "yakhip_internal.h"

#define YAK_FFWD_FN __host__ __device__ inline
#include "ffwd.hpp"

namespace yakhip
{
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

constexpr int kDefaultTileShift = 6;  // wave footprint in a layer, see lane_voxels()
constexpr int kMixedWavesPerSimd = 5;  // size of integrate_list_kernel's grid: workgroups per CU
constexpr uint64_t kTwoKernelMinVoxels = 1ull << 25;  // smaller slabs are one kernel unless "defer_mixed" is 2
constexpr int kPartLayers = 8;         // a listed brick is walked in parts of this many layers (at most four parts)
#ifndef YAK_CP_STRIDE
#define YAK_CP_STRIDE 32
#endif
constexpr int kCheckpointStride = YAK_CP_STRIDE;  // layers between the column checkpoints of a launch with a list
// layers per part of a brick of `layers` layers: kPartLayers, or what makes four parts of a longer brick (even)
__host__ __device__ inline int part_layers(int layers)
{
  const int quarter = (((layers + 3) >> 2) + 1) & ~1;
  return quarter > kPartLayers ? quarter : kPartLayers;
}

template <int VX>
struct Vec;
template <>
struct Vec<4>
{
  typedef u32x4 type;
};
template <>
struct Vec<2>
{
  typedef u32x2 type;
};
template <>
struct Vec<1>
{
  typedef uint32_t type;
};

template <int VX>
__device__ __forceinline__ uint32_t vget(const typename Vec<VX>::type& v, int i)
{
  return v[i];
}
template <>
__device__ __forceinline__ uint32_t vget<1>(const uint32_t& v, int)
{
  return v;
}
template <int VX>
__device__ __forceinline__ void vset(typename Vec<VX>::type& v, int i, uint32_t x)
{
  v[i] = x;
}
template <>
__device__ __forceinline__ void vset<1>(uint32_t& v, int, uint32_t x)
{
  v = x;
}

// ---- correctly rounded fp32 division in 3 + 3 instructions -------------------------------------
// refined_rcp(d) = v_rcp_f32 + one Newton step.  On gfx950 it equals the correctly rounded 1/d for
// EVERY fp32 mantissa (checked exhaustively on the device by yakhip_selftest_division, tests/test_division_gpu.py; the
// arithmetic is exponent-invariant while nothing under/overflows).  With y = RN(1/d) and
// q0 = RN(n*y), Markstein's theorem gives RN(q0 + (n - d*q0)*y) = RN(n/d): the correctly rounded
// quotient, i.e. bit-identical to IEEE '/'.  Callers keep d inside [2^-20, 2^20]; numerators so
// small that the residual would underflow cannot change an observable result here.
__device__ __forceinline__ float refined_rcp(float d)
{
  const float r = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r, 1.f);
  return __builtin_fmaf(e, r, r);
}

__device__ __forceinline__ float div_by(float n, float d, float r /* refined_rcp(d) */)
{
  const float q = n * r;
  const float e = __builtin_fmaf(-d, q, n);
  return __builtin_fmaf(e, r, q);
}

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// 2^-20 <= v <= 2^20 (negative, zero, NaN fail)
__device__ __forceinline__ bool in_div_range(float v)
{
  return (__float_as_uint(v) - 0x35800000u) <= (0x49800000u - 0x35800000u);
}

__device__ __forceinline__ int floor_to_int(float v)
{
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(v));  // floor, saturating; no builtin exposes it
  return r;
}

// One voxel, every case, exactly as the reference writes it: tsdf_volume.cu:78-96.
// Used for the voxels the fast path cannot settle (truncation band, denominators outside the
// Markstein range).  Returns true when the voxel takes an update.
__device__ __forceinline__ bool measure_exact(const IntegrateParams& p, float vcx, float vcy, float vcz, float& tsdf)
{
  // :89 tests 'vc.z <= 0' after the fetch; testing it first skips exactly the same voxels
  if (!(vcz > 0.f))
    return false;
  const float coox = __builtin_fmaf(p.fx, vcx / vcz, p.cx);  // Projector, device.hpp:46-52
  const float cooy = __builtin_fmaf(p.fy, vcy / vcz, p.cy);
  // :83 (written so that a NaN coordinate is rejected instead of indexing the image)
  if (!(coox >= 0.f && cooy >= 0.f && coox < (float)p.cols && cooy < (float)p.rows))
    return false;
  // :87 point fetch at (floor(coo.x), floor(coo.y)); coo >= 0 here so truncation == floor
  const uint32_t px = (uint32_t)(int)coox, py = (uint32_t)(int)cooy;
  const float Dp = half_bits_to_float(*(const uint16_t*)(p.dists + (py * p.dists_pitch + px * 2u)));
  if (Dp == 0.f)
    return false;
  const float sdf = Dp - __builtin_sqrtf(dot3(vcx, vcy, vcz, vcx, vcy, vcz));  // :92
  if (!(sdf >= -p.trunc))                                                      // :94
    return false;
  tsdf = __builtin_fminf(1.f, sdf * p.trunc_inv);  // :96
  return true;
}

// Weighted running average of one packed voxel: tsdf_volume.cu:98-106, device.hpp:69-77.
__device__ __forceinline__ uint32_t fuse(uint32_t voxel, float tsdf, int max_weight)
{
  const float tsdf_prev = half_bits_to_float(voxel & 0xffffu);
  const int weight_prev = (int)(int16_t)(uint16_t)(voxel >> 16);
  const float num = __builtin_fmaf(tsdf_prev, (float)weight_prev, tsdf);
  const float den = (float)(weight_prev + 1);
  float tsdf_new;
  if (weight_prev >= 0)  // den in [1, 32768]
    tsdf_new = div_by(num, den, refined_rcp(den));
  else
    tsdf_new = num / den;  // only reachable from a corrupt volume; same arithmetic as the oracle
  const int weight_new = min(weight_prev + 1, max_weight);
  return float_to_half_bits(tsdf_new) | ((uint32_t)(uint16_t)weight_new << 16);
}

// ---- per-pixel band thresholds -------------------------------------------------------------------
// For a voxel at squared camera distance d2 that sees a pixel with ray length Dp:
//   d2 <= A2(Dp)  =>  Dp - |vc| >= trunc*(1 + 2^-10)  =>  sdf >= -trunc and fmin(1, sdf/trunc) == 1
//   d2 >  B2(Dp)  =>  Dp - |vc| <  -trunc             =>  no update
// with A2 = a*a*(1 - 2^-17), a = fma(Dp, 1 - 2^-20, -trunc*(1 + 2^-9)) (when a > 0), and
//      B2 = b*b*(1 + 2^-17), b = fma(Dp, 1 + 2^-20,  trunc*(1 + 2^-9)).
// The slack factors dominate every rounding in forming A2, B2, d2, sqrt and the subtraction, so both
// tests are sufficient (never wrong); what they leave undecided is a shell ~1.002*trunc thick around
// the surface, which goes through measure_exact().  Pixels with no return (Dp == 0, tsdf_volume.cu:89)
// or NaN get A2 = B2 = -1: never "in front", always "behind".  Derivation: DESIGN.md, "Band tests".
__device__ __forceinline__ f32x2 band_thresholds(float Dp, float trunc_hi)
{
  f32x2 out = { -1.f, -1.f };
  if (Dp != 0.f && Dp == Dp)
  {
    const float a = __builtin_fmaf(Dp, 0x1.ffffep-1f, -trunc_hi);
    if (a > 0.f)
      out.x = (a * a) * 0x1.fffep-1f;  // 1 - 2^-16: conservative (the proof needs 2^-17; the rest is
                                       // spent by the group test, which bounds d2 by its end voxels)
    const float b = __builtin_fmaf(Dp, 0x1.00001p+0f, trunc_hi);
    if (b >= 0.f)
      out.y = (b * b) * 0x1.0001p+0f;  // 1 + 2^-16
  }
  return out;
}

// A slab that starts at layer z0 of a tall volume needs every column's camera-space position after the z0
// rounded additions the reference performed below it.  Replaying them in the integrate kernel costs z0 steps
// per chunk start -- for the last of 8 ranks of a 1024-layer volume more than the rank's own work, and all
// workgroups do it at the same time, with the memory system idle.  Here they are done once per column and
// frame, in closed form (ffwd.hpp: bit-identical to the loop), by extra workgroups of the prepass launch,
// and the integrate kernels start from that.
constexpr int kCheckpointFrom = 64;  // slabs starting below this layer replay from 0 as before
constexpr int kBrickCell = 16;       // pixels per side of a brick-test cell (prepass_kernel, classify_brick)
constexpr int kFineShift = 2;        // ... and of the cells a mixed brick's pieces are tested against: 4 x 4 pixels
constexpr int kFineCell = 1 << kFineShift;
constexpr int kFineMaxCells = 64;    // a pixel box of more fine cells than this is looked up in the 16-pixel cells

__device__ __forceinline__ void checkpoint_column(const IntegrateParams& p, float* out, uint32_t col)
{
  if (col >= p.plane)
    return;
  const int y = (int)(col / (uint32_t)p.X), x = (int)(col - (uint32_t)y * (uint32_t)p.X);
  // tsdf_volume.cu:69, 71-72, exactly as integrate_kernel starts a column
  const float vxx = (float)x * p.vsx, vxy = (float)y * p.vsy;
  const float vx0 = dot3(p.R[0], p.R[1], p.R[2], vxx, vxy, 0.f) + p.t[0];
  const float vy0 = dot3(p.R[3], p.R[4], p.R[5], vxx, vxy, 0.f) + p.t[1];
  const float vz0 = dot3(p.R[6], p.R[7], p.R[8], vxx, vxy, 0.f) + p.t[2];
  out[col] = yak_ffwd_accumulate(vx0, p.R[2] * p.vsz, p.z0);
  out[p.plane + col] = yak_ffwd_accumulate(vy0, p.R[5] * p.vsz, p.z0);
  out[2u * p.plane + col] = yak_ffwd_accumulate(vz0, p.R[8] * p.vsz, p.z0);
}

// One launch per frame, two jobs (both read only the half ray-distance image):
//   blocks [0, thr_blocks)   per-pixel {A2, B2}                       -> thr[rows][cols]
//   (block numbers after the checkpoint workgroups, if any, which come first)
//   blocks [thr_blocks, ..)  coarse[by][bx] = min A2 over the pixel window
//                            [bk*bx, bk*bx + 2*bk) x [bk*by, bk*by + 2*bk) clipped to the image
//                            (one wave per cell).  Any pixel box no larger than bk x bk whose low
//                            corner lies in cell (bx, by) is inside that window.
__global__ __launch_bounds__(256) void prepass_kernel(const uint8_t* dists, uint32_t dists_pitch, int rows, int cols,
                                                      float trunc_hi, f32x2* thr, float* coarse, int cw, int ch,
                                                      int bk, uint32_t thr_blocks_x, uint32_t thr_blocks,
                                                      uint32_t ckpt_blocks, const IntegrateParams cp,
                                                      float* ckpt_out, float* cellmin, float* cellmax, int ccw, int cch,
                                                      uint32_t coarse_blocks, uint32_t brick_blocks, uint32_t fine_blocks,
                                                      const uint32_t* prev_classes, uint32_t prev_bricks, uint32_t prev_seq,
                                                      uint32_t* stats_out)
{
  // the column checkpoints of a deep slab are the long job of this launch: its workgroups come first, the
  // image's fill in behind them (ckpt_blocks = 0 for a slab that replays from layer 0)
  if (blockIdx.x < ckpt_blocks)
  {
    checkpoint_column(cp, ckpt_out, blockIdx.x * 256u + threadIdx.x);
    return;
  }
  const uint32_t bid = blockIdx.x - ckpt_blocks;
  if (bid < thr_blocks)
  {
    const int x = (int)((bid % thr_blocks_x) * 256u + threadIdx.x);
    const int y = (int)(bid / thr_blocks_x);
    if (x >= cols)
      return;
    const uint32_t bits = *(const uint16_t*)(dists + (size_t)y * dists_pitch + (size_t)x * 2u);
    f32x2 ab = band_thresholds(half_bits_to_float(bits), trunc_hi);
    // second word: B2 rounded UP to its upper 16 bits (still a sufficient "behind" threshold, 2^-8 looser)
    // with the pixel's half ray length in the lower 16 -- read as a float it is >= that rounded B2, so
    // the kernel compares d2 against the word as it is, and the band path finds Dp without a second gather
    if (ab.y >= 0.f)
      ab.y = __uint_as_float(((__float_as_uint(ab.y) + 0xffffu) & 0xffff0000u) | bits);
    thr[(size_t)y * cols + x] = ab;
    if (x == 0 && y == 0)  // the entry behind the image: what a voxel that sees no pixel gathers (IntegrateParams::thr_none)
      thr[(size_t)rows * cols] = f32x2{ -1.f, -1.f };
    return;
  }
  const int lane = (int)(threadIdx.x & 63u);
  if (bid >= thr_blocks + coarse_blocks + brick_blocks + fine_blocks)
  {
    // One more workgroup, when the launcher asks for it: what the PREVIOUS launch's brick test found -- front and mixed
    // wave bricks, from the class bytes it left -- written to pinned host memory, where the launcher reads it (whenever
    // it gets there: a hint) to choose between the one-kernel and the two-kernel launch for a later frame.
    const uint32_t groups = (prev_bricks + 1023u) >> 10;
    uint32_t front = 0, live = 0;
    for (uint32_t i0 = 0; i0 < groups; i0 += 8u)  // (eight loads in flight: this workgroup must not outlast the others)
    {
      uint32_t w[8];
#pragma unroll
      for (uint32_t k = 0; k < 8u; ++k)
        w[k] = i0 + k < groups ? prev_classes[(i0 + k) * 256u + threadIdx.x] : 0u;
#pragma unroll
      for (uint32_t k = 0; k < 8u; ++k)
      {
        front += (uint32_t)__builtin_popcount(w[k] & 0x80808080u);
        const uint32_t u = w[k] & 0x07070707u;
        live += (uint32_t)__builtin_popcount((u | (u >> 1) | (u >> 2)) & 0x01010101u);
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
    {
      front += __shfl_xor(front, off);
      live += __shfl_xor(live, off);
    }
    __shared__ uint32_t part[2][4];
    if (lane == 0)
      part[0][threadIdx.x >> 6] = front, part[1][threadIdx.x >> 6] = live;
    __syncthreads();
    if (threadIdx.x == 0)
    {
      front = part[0][0] + part[0][1] + part[0][2] + part[0][3];
      live = part[1][0] + part[1][1] + part[1][2] + part[1][3];
      __builtin_nontemporal_store(front, stats_out + 0);
      __builtin_nontemporal_store(live - front, stats_out + 1);
      __builtin_nontemporal_store(prev_bricks, stats_out + 2);
      __builtin_nontemporal_store(prev_seq, stats_out + 3);  // (last: the launcher takes the four words as they are)
    }
    return;
  }
  if (bid >= thr_blocks + coarse_blocks + brick_blocks)
  {
    // the brick test's planes once more over kFineCell x kFineCell pixel cells (one lane per cell): what the second
    // kernel tests the pieces of a mixed brick against (cp.finemin / cp.finemax, cp.fcw x cp.fch cells)
    const int c = (int)(bid - thr_blocks - coarse_blocks - brick_blocks) * 256 + (int)threadIdx.x;
    if (c >= cp.fcw * cp.fch)
      return;
    const int cx = c % cp.fcw, cy = c / cp.fcw;
    float lo = __builtin_inff(), hi = -__builtin_inff();
    for (int j = 0; j < kFineCell; ++j)
      for (int i = 0; i < kFineCell; ++i)
      {
        const int x = kFineCell * cx + i, y = kFineCell * cy + j;
        if (x < cols && y < rows)
        {
          const uint32_t bits = *(const uint16_t*)(dists + (size_t)y * dists_pitch + (size_t)x * 2u);
          f32x2 ab = band_thresholds(half_bits_to_float(bits), trunc_hi);
          if (ab.y >= 0.f)  // the word the kernel compares d2 with, as below
            ab.y = __uint_as_float(((__float_as_uint(ab.y) + 0xffffu) & 0xffff0000u) | bits);
          lo = __builtin_fminf(lo, ab.x);
          hi = __builtin_fmaxf(hi, ab.y);
        }
      }
    const_cast<float*>(cp.finemin)[c] = lo;
    const_cast<float*>(cp.finemax)[c] = hi;
    return;
  }
  if (bid >= thr_blocks + coarse_blocks)
  {
    // brick test planes: min A2 / max B2 word over the pixels of each kBrickCell x kBrickCell cell (one wave per
    // cell, no overlap: a brick's pixel box queries every cell it touches)
    const int c = (int)(bid - thr_blocks - coarse_blocks) * 4 + (int)(threadIdx.x >> 6);
    if (c >= ccw * cch)
      return;
    const int cx = c % ccw, cy = c / ccw;
    float lo = __builtin_inff(), hi = -__builtin_inff();
    for (int i = lane; i < kBrickCell * kBrickCell; i += 64)
    {
      const int x = kBrickCell * cx + i % kBrickCell, y = kBrickCell * cy + i / kBrickCell;
      if (x < cols && y < rows)
      {
        const uint32_t bits = *(const uint16_t*)(dists + (size_t)y * dists_pitch + (size_t)x * 2u);
        f32x2 ab = band_thresholds(half_bits_to_float(bits), trunc_hi);
        if (ab.y >= 0.f)  // the word the kernel compares d2 with (see above): >= B2
          ab.y = __uint_as_float(((__float_as_uint(ab.y) + 0xffffu) & 0xffff0000u) | bits);
        lo = __builtin_fminf(lo, ab.x);
        hi = __builtin_fmaxf(hi, ab.y);
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
    {
      lo = __builtin_fminf(lo, __shfl_xor(lo, off));
      hi = __builtin_fmaxf(hi, __shfl_xor(hi, off));
    }
    if (lane == 0)
    {
      cellmin[c] = lo;
      cellmax[c] = hi;
    }
    return;
  }
  const int cell = (int)(bid - thr_blocks) * 4 + (int)(threadIdx.x >> 6);
  if (cell >= cw * ch)
    return;
  const int bx = cell % cw, by = cell / cw;
  const int w = 2 * bk;
  float m = __builtin_inff();
  for (int i = lane; i < w * w; i += 64)
  {
    const int x = bk * bx + i % w, y = bk * by + i / w;
    if (x < cols && y < rows)
    {
      const float Dp = half_bits_to_float(*(const uint16_t*)(dists + (size_t)y * dists_pitch + (size_t)x * 2u));
      m = __builtin_fminf(m, band_thresholds(Dp, trunc_hi).x);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    m = __builtin_fminf(m, __shfl_xor(m, off));
  if (lane == 0)
    coarse[cell] = m;
}

#ifdef YAK_INTEGRATE_TIMING
// diagnostic build only (make EXTRA=-DYAK_INTEGRATE_TIMING; tools/bench_orbit_waves.py): per wave brick of the last launch
// {class + 1, shader-clock cycles in all, cycles to the end of the brick test, replay cycles,
//  layers walked | duration in 10 ns ticks << 16, (HW_ID | XCC << 16) << 40 | start on the device-wide 100 MHz clock}
constexpr int kItSlots = 1 << 17;
__device__ unsigned long long g_it_time[kItSlots][6];
// ... and per walked brick, shader-clock cycles spent in {stage 1, settle, the wait for voxels the look-ahead did not bring,
// fuse, store + look-ahead}, then the number of {such waits, layers that took the per-voxel path, layers with band voxels}
__device__ unsigned long long g_it_phase[kItSlots][8];
struct ItPhase
{
  unsigned long long t[8];
};
#define IT_CLOCK(var) var = __builtin_readcyclecounter()
#define IT_PHASE_PARAM , ItPhase& itp
#define IT_PHASE_ARG , itp
#define IT_LAP(i)                                              \
  do                                                           \
  {                                                            \
    const unsigned long long it_n = __builtin_readcyclecounter(); \
    itp.t[i] += it_n - it_lap;                                 \
    it_lap = it_n;                                             \
  } while (0)
#define IT_LAP_START unsigned long long it_lap = __builtin_readcyclecounter()
#define IT_COUNT(i, cond) itp.t[i] += (cond) ? 1 : 0
#define IT_BEGIN                                                        \
  unsigned long long it_t0, it_t1 = 0, it_t3 = 0;                       \
  const unsigned long long it_r0 = __builtin_amdgcn_s_memrealtime();    \
  IT_CLOCK(it_t0)
#define IT_ROW(slot_, cls_, t_cls_, t_replay_, layers_)                                                                  \
  do                                                                                                                     \
  {                                                                                                                      \
    if ((threadIdx.x & 63u) == 0)                                                                                        \
    {                                                                                                                    \
      unsigned hw, xcc;                                                                                                  \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                                   \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                                 \
      hw = (hw & 0xffffu) | ((xcc & 0xfu) << 16);                                                                        \
      unsigned long long* row = g_it_time[(size_t)(slot_) & (kItSlots - 1)];                                             \
      const unsigned long long it_now = __builtin_readcyclecounter();                                                    \
      row[0] = (unsigned long long)(cls_) + 1, row[1] = it_now - it_t0, row[2] = (t_cls_) - it_t0, row[3] = (t_replay_); \
      row[4] = (unsigned long long)(layers_) | ((__builtin_amdgcn_s_memrealtime() - it_r0) << 16);                       \
      row[5] = ((unsigned long long)hw << 40) | (it_r0 & 0xffffffffffull);                                               \
    }                                                                                                                    \
  } while (0)
#else
#define IT_CLOCK(var)
#define IT_BEGIN
#define IT_ROW(slot_, cls_, t_cls_, t_replay_, layers_)
#define IT_PHASE_PARAM
#define IT_PHASE_ARG
#define IT_LAP(i)
#define IT_LAP_START
#define IT_COUNT(i, cond)
#endif

// ---- the kernel ------------------------------------------------------------------------------------
// What the brick test (classify_box below) says of a box of voxels.
enum BrickClass
{
  kBrickMixed = 0,
  kBrickDead = 1,
  kBrickFront = 2
};
// Per-lane predicates are kept as 64-bit wave masks in SGPRs (ballot / inverse ballot): combining
// them is scalar work, and a rare case anywhere in the wave is one scalar branch.
typedef unsigned long long mask_t;
__device__ __forceinline__ mask_t ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }
__device__ __forceinline__ bool lanes(mask_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// Camera-space positions of the VX voxels of one lane at one layer, as voxel pairs for packed math.
// With four voxels the pairs are (v0, v3) and (v1, v2): pair 0 then holds the two ends of the run,
// which is all the group test needs.
template <int VX>
struct Pos
{
  static constexpr int NP = (VX + 1) / 2;
  f32x2 x[NP], y[NP], z[NP];
  // voxel index held by half h of pair k (>= VX: the slot shadows voxel 0 and is never stored)
  static constexpr int voxel(int k, int h) { return VX == 4 ? (k == 0 ? (h == 0 ? 0 : 3) : (h == 0 ? 1 : 2)) : 2 * k + h; }
  static constexpr int pair_of(int v) { return VX == 4 ? (v == 0 || v == 3 ? 0 : 1) : v / 2; }
  static constexpr int half_of(int v) { return VX == 4 ? (v == 0 || v == 1 ? 0 : 1) : v % 2; }
};

// What one layer's projection stage leaves for its update stage (one loop trip later).
template <int VX>
struct Probe
{
  f32x2 ab[VX];       // gathered {A2, B2} thresholds of the pixel each voxel sees; {-1, -1} ("never in front, always
                      // behind", what a pixel without a return has) for a voxel that projects outside the image or
                      // whose vc.z is outside the Markstein range: no mask per voxel has to travel with them
  float d2[VX];       // dot(vc, vc)
  mask_t slow_any;    // lanes with a voxel whose vc.z is outside [2^-20, 2^20]: only measure_exact() can decide it
  mask_t pass;        // lanes whose whole voxel run was settled by the group test (all VX voxels hit, tsdf 1)
  bool detailed;      // wave-uniform: the per-voxel fields above were produced for this layer
};

// Stage 1 of layer z: project VX voxels, start the threshold gathers.
template <int VX>
__device__ __forceinline__ void project(const IntegrateParams& p, const Pos<VX>& c, Probe<VX>& o, f32x2 fx2,
                                        f32x2 fy2, f32x2 cx2, f32x2 cy2)
{
#pragma unroll
  for (int k = 0; k < Pos<VX>::NP; ++k)
  {
    // Projector (device.hpp:46-52) for two voxels at once: correctly rounded vc.xy / vc.z
    f32x2 r = { __builtin_amdgcn_rcpf(c.z[k].x), __builtin_amdgcn_rcpf(c.z[k].y) };
    r = pk_fma(pk_fma(-c.z[k], r, f32x2{ 1.f, 1.f }), r, r);
    f32x2 qx = c.x[k] * r;
    qx = pk_fma(pk_fma(-c.z[k], qx, c.x[k]), r, qx);
    f32x2 qy = c.y[k] * r;
    qy = pk_fma(pk_fma(-c.z[k], qy, c.y[k]), r, qy);
    const f32x2 coox = pk_fma(fx2, qx, cx2);
    const f32x2 cooy = pk_fma(fy2, qy, cy2);
    // dot(vc, vc), temp_utils.hpp:36-39
    const f32x2 d2 = pk_fma(c.x[k], c.x[k], pk_fma(c.y[k], c.y[k], c.z[k] * c.z[k]));
#pragma unroll
    for (int h = 0; h < 2; ++h)
    {
      const int v = Pos<VX>::voxel(k, h);
      if (v >= VX)
        continue;
      const mask_t fast = ballot(in_div_range(c.z[k][h]));
      const uint32_t px = (uint32_t)floor_to_int(coox[h]), py = (uint32_t)floor_to_int(cooy[h]);
      // :83 -- floor(coo) inside the image <=> 0 <= coo < size
      const mask_t inside = fast & ballot(px < (uint32_t)p.cols) & ballot(py < (uint32_t)p.rows);
      const uint32_t idx = lanes(inside) ? __umul24(py, (uint32_t)p.cols) + px : p.thr_none;
      o.ab[v] = p.thr[idx];
      o.d2[v] = d2[h];
      o.slow_any |= ~fast;
    }
  }
}

// ---- group test: settle a lane's whole run of voxels with one coarse lookup ---------------------
// The four voxels of a lane are collinear in camera space, so their pixels lie in the box spanned
// by the pixels of the two end voxels, and their squared distances are bounded by the larger of the
// two ends'.  If that pixel box (grown by a margin that covers every rounding below) lies inside the
// image, is at most one coarse cell wide, and max(d2_ends) <= the minimum of A2 over the coarse
// window that contains it, then every voxel of the run is in front of the surface by more than the
// truncation distance: all four take the update with tsdf == 1 -- the same conclusion the per-voxel
// path reaches, with approximate (v_rcp, no refinement) projections.  Anything else, anywhere in the
// wave, sends the wave down the exact per-voxel path.  Derivation and margins: DESIGN.md, "Group test".
__device__ __forceinline__ mask_t group_test(const IntegrateParams& p, f32x2 x, f32x2 y, f32x2 z,
                                             const float* coarse_lds, f32x2 fx2, f32x2 fy2, f32x2 cx2, f32x2 cy2)
{
  const f32x2 rz = { __builtin_amdgcn_rcpf(z.x), __builtin_amdgcn_rcpf(z.y) };
  const f32x2 u = pk_fma(fx2, x * rz, cx2);
  const f32x2 w = pk_fma(fy2, y * rz, cy2);
  const f32x2 d2 = pk_fma(x, x, pk_fma(y, y, z * z));
  const float ulo = __builtin_fminf(u.x, u.y), uhi = __builtin_fmaxf(u.x, u.y);
  const float wlo = __builtin_fminf(w.x, w.y), whi = __builtin_fmaxf(w.x, w.y);
  const float d2max = __builtin_fmaxf(d2.x, d2.y);
  const float zmin = __builtin_fminf(z.x, z.y);
  const mask_t ok = ballot(ulo >= p.grp_margin) & ballot(uhi < p.grp_cols_hi) & ballot(wlo >= p.grp_margin) &
                    ballot(whi < p.grp_rows_hi) & ballot(uhi - ulo <= p.grp_span) & ballot(whi - wlo <= p.grp_span) &
                    ballot(zmin >= p.grp_zmin);
  // coarse cell of the low corner: floor((lo - margin) / bk)
  const int bx = floor_to_int(__builtin_fmaf(ulo, p.grp_inv_bk, p.grp_bias));
  const int by = floor_to_int(__builtin_fmaf(wlo, p.grp_inv_bk, p.grp_bias));
  const uint32_t idx = lanes(ok) ? __umul24((uint32_t)by, (uint32_t)p.cw) + (uint32_t)bx : 0u;
  return ok & ballot(d2max <= coarse_lds[idx]);
}

// Stage 1 of one layer: group test first; the exact per-voxel projection only if some lane needs it.
// lcls (wave-uniform): what is already known of this layer of the wave's brick -- kBrickDead: no voxel of it can take
// an update, kBrickFront: every voxel takes one with tsdf 1 (classify_pieces); neither needs a projection.
template <int VX>
__device__ __forceinline__ void stage1(const IntegrateParams& p, const Pos<VX>& c, Probe<VX>& o,
                                       const float* coarse_lds, mask_t active, f32x2 fx2, f32x2 fy2, f32x2 cx2,
                                       f32x2 cy2, int lcls = kBrickMixed)
{
  if (lcls != kBrickMixed)
  {
    o.detailed = false;
    o.pass = lcls == kBrickFront ? active : 0;
    o.slow_any = 0;
    return;
  }
  mask_t pass = 0;
  if constexpr (VX == 4)
    if (p.grp_enable)
      pass = group_test(p, c.x[0], c.y[0], c.z[0], coarse_lds, fx2, fy2, cx2, cy2);
  o.detailed = (active & ~pass) != 0;
  o.pass = o.detailed ? 0 : pass;
  o.slow_any = 0;
  if (o.detailed)
  {
    project<VX>(p, c, o, fx2, fy2, cx2, cy2);
    o.slow_any &= active;
  }
}

template <int VX>
__device__ __forceinline__ void advance(const Pos<VX>& from, Pos<VX>& to, f32x2 sx, f32x2 sy, f32x2 sz)
{
#pragma unroll
  for (int k = 0; k < Pos<VX>::NP; ++k)
  {
    to.x[k] = from.x[k] + sx;
    to.y[k] = from.y[k] + sy;
    to.z[k] = from.z[k] + sz;
  }
}

// ---- volume access through a buffer descriptor ------------------------------------------------
// Every volume load/store of the hot loop is a raw buffer op on a descriptor that covers exactly
// this workgroup's z-chunk.  The lane part of the address (the voxel group inside the layer) never
// changes, the layer offset is a scalar (soffset), and a lane that must NOT access memory passes
// voffset = 2^31 (beyond the descriptor): the hardware range check then returns zeros / drops the store.  So each op
// is issued unconditionally -- no exec-mask branch, a static count for the compiler's s_waitcnt --
// and costs no address arithmetic per layer.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
// 2 GiB: past the end of any descriptor the launcher builds (chunks are kept below 2 GiB), and far
// enough from 2^32 that adding the scalar layer offset cannot wrap back into range.
constexpr uint32_t kNoAccess = 0x80000000u;

template <int VX, bool NT>
__device__ __forceinline__ typename Vec<VX>::type buf_load(rsrc_t rs, uint32_t voff, uint32_t soff)
{
  constexpr int aux = NT ? 2 : 0;  // bit 1 = nt
  if constexpr (VX == 4)
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, aux));
  else if constexpr (VX == 2)
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, aux));
  else
    return __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, aux);
}

template <int VX, bool NT>
__device__ __forceinline__ void buf_store(typename Vec<VX>::type v, rsrc_t rs, uint32_t voff, uint32_t soff)
{
  constexpr int aux = NT ? 2 : 0;
  if constexpr (VX == 4)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v), rs, voff, soff, aux);
  else if constexpr (VX == 2)
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(unsigned int)))) unsigned int, v), rs, voff, soff, aux);
  else
    __builtin_amdgcn_raw_buffer_store_b32(v, rs, voff, soff, aux);
  // A 16-byte store reads its data VGPRs for some cycles after issue.  hipcc's hazard recogniser exempts
  // buffer stores whose soffset is an SGPR, and in the batched kernel it placed a v_cndmask into the
  // first data register right behind such a store: on gfx950 that value, not the voxel, reached memory
  // (512^3, timing dependent, only the lanes read last).  The asm keeps the data registers live -- so
  // nothing may be allocated over them -- until the idle states have passed.
  if constexpr (VX == 4)
    asm volatile("s_nop 7" ::"v"(v));
}

// Decide every voxel of a lane's run at one layer from what stage 1 left: which voxels take an update
// (hit), which of those with a measurement of exactly 1.0 (unit), and the measurement (tsdf).
template <int VX>
__device__ __forceinline__ void settle(const IntegrateParams& p, const Pos<VX>& c, const Probe<VX>& pr, mask_t live_in,
                                       mask_t (&hit)[VX], mask_t (&unit)[VX], float (&tsdf)[VX], mask_t& negative)
{
  const mask_t live_m = live_in & ballot(true);  // (lanes that own no voxels are not part of any mask)
#pragma unroll
  for (int v = 0; v < VX; ++v)
  {
    hit[v] = unit[v] = pr.pass & live_m;  // the group test settled the whole run (or nothing, if detailed)
    tsdf[v] = 1.f;
  }
  if (pr.detailed)  // wave-uniform
  {
    mask_t rare_any = 0;
    mask_t rare[VX];
#pragma unroll
    for (int v = 0; v < VX; ++v)
    {
      // (a voxel outside the image, or one only measure_exact() can decide, carries {-1, -1}: not in front, and
      //  behind unless its d2 is a NaN -- then the band path below finds a ray length of 0 and no update)
      const mask_t front = ballot(pr.d2[v] <= pr.ab[v].x) & live_m;
      const mask_t behind = ballot(pr.d2[v] > pr.ab[v].y);
      hit[v] = front;
      unit[v] = front;
      rare[v] = ~front & ~behind & live_m;
      rare_any |= rare[v];
    }
    rare_any |= pr.slow_any & live_m;
    if (rare_any != 0)  // wave-uniform branch
    {
      // Inside the truncation band the reference's expression is evaluated as written (tsdf_volume.cu:92-96)
      // on what stage 1 already holds: d2 = dot(vc, vc) bit for bit, and the pixel's ray length in the low
      // half of the threshold word.  Straight-line for all VX voxels; the masks pick the lanes it is for.
#pragma unroll
      for (int v = 0; v < VX; ++v)
      {
        const float Dp = half_bits_to_float(__float_as_uint(pr.ab[v].y) & 0xffffu);
        const float sdf = Dp - __builtin_sqrtf(pr.d2[v]);
        const float t = __builtin_fminf(1.f, sdf * p.trunc_inv);
        const mask_t in_band = rare[v] & ballot(sdf >= -p.trunc);
        tsdf[v] = lanes(in_band) ? t : tsdf[v];
        hit[v] |= in_band;
        unit[v] |= in_band & ballot(t == 1.f);
        negative |= in_band & ballot(t < 0.f);  // (surface map: the only way a voxel turns negative)
      }
      if ((pr.slow_any & live_m) != 0)  // wave-uniform: vc.z outside the Markstein range, practically never
      {
        // Masks are wave-wide values: they may only be updated in wave-uniform control flow.  Inside
        // the per-lane regions below only per-lane variables change; the ballots come after the join.
        // (Such a voxel carries {-1, -1}, so the band path above found a ray length of 0 for it and left it alone.)
        bool exact_hit[VX];
#pragma unroll
        for (int v = 0; v < VX; ++v)
        {
          exact_hit[v] = false;
          const float vz = c.z[Pos<VX>::pair_of(v)][Pos<VX>::half_of(v)];
          if (!in_div_range(vz) && lanes(live_m))
          {
            float t = 1.f;
            exact_hit[v] = measure_exact(p, c.x[Pos<VX>::pair_of(v)][Pos<VX>::half_of(v)],
                                         c.y[Pos<VX>::pair_of(v)][Pos<VX>::half_of(v)], vz, t);
            tsdf[v] = exact_hit[v] ? t : tsdf[v];
          }
        }
#pragma unroll
        for (int v = 0; v < VX; ++v)
        {
          hit[v] |= ballot(exact_hit[v]);
          unit[v] |= ballot(exact_hit[v] && tsdf[v] == 1.f);
          negative |= ballot(exact_hit[v] && tsdf[v] < 0.f);
        }
      }
    }
  }
}

// Fold one measurement per voxel into the packed voxels `buf` -> `out` (tsdf_volume.cu:98-106).
template <int VX>
__device__ __forceinline__ void fuse_run(const typename Vec<VX>::type& buf, typename Vec<VX>::type& out,
                                         const mask_t (&hit)[VX], const mask_t (&unit)[VX], const float (&tsdf)[VX],
                                         int max_weight, uint32_t max_w_hi)
{
  mask_t general[VX];
  mask_t general_any = 0;
#pragma unroll
  for (int v = 0; v < VX; ++v)
  {
    const uint32_t old = vget<VX>(buf, v);
    // free space seen again: tsdf_prev == 1.0, weight >= 0, new measurement 1.0
    //   => (1*w + 1) / (w + 1) == 1 exactly; only the weight moves: min(w + 1, max_weight)
    const mask_t same = hit[v] & unit[v] & ballot((old & 0x8000ffffu) == 0x00003c00u);
    const u16x2 bumped = __builtin_bit_cast(u16x2, old) + u16x2{ 0, 1 };
    const u16x2 capped = __builtin_elementwise_min(bumped, __builtin_bit_cast(u16x2, max_w_hi));
    vset<VX>(out, v, lanes(same) ? __builtin_bit_cast(uint32_t, capped) : old);
    general[v] = hit[v] & ~same;
    general_any |= general[v];
  }
  if (general_any != 0)  // wave-uniform
  {
#pragma unroll
    for (int v = 0; v < VX; ++v)
      if (lanes(general[v]))
        vset<VX>(out, v, fuse(vget<VX>(buf, v), tsdf[v], max_weight));
  }
}

// Stage 2 of layer z: settle every voxel, fuse, write back; then fetch the voxels two layers ahead
// into the same register buffer.
//   soff        byte offset of this layer inside the chunk's buffer
//   ahead_ok    the layer two steps on is still inside the chunk
template <int VX, bool NT>
__device__ __forceinline__ void update(const IntegrateParams& p, const Pos<VX>& c, const Probe<VX>& pr,
                                       typename Vec<VX>::type& buf, mask_t& buf_valid, rsrc_t rs, uint32_t voff,
                                       uint32_t soff, uint32_t layer_bytes, bool live, bool ahead_ok,
                                       uint32_t max_w_hi, mask_t& band IT_PHASE_PARAM)
{
  typedef typename Vec<VX>::type vec_t;
  mask_t hit[VX], unit[VX];
  float tsdf[VX];
  mask_t hit_any = 0;
  IT_LAP_START;
  IT_COUNT(6, pr.detailed);
  settle<VX>(p, c, pr, live ? ~0ull : 0ull, hit, unit, tsdf, band);
  IT_LAP(1);
#pragma unroll
  for (int v = 0; v < VX; ++v)
    hit_any |= hit[v];

  // the reference reads only updated voxels (tsdf_volume.cu:98-100): demand-load what the look-ahead
  // did not bring (first two layers of a chunk, and lanes whose updates start at this layer)
  const mask_t need = hit_any & ~buf_valid;
  IT_COUNT(5, need != 0);
  if (need != 0)  // wave-uniform, rare after the first trips
  {
    const vec_t got = buf_load<VX, NT>(rs, lanes(need) ? voff : kNoAccess, soff);
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): settle here so the hot path keeps its counted waits
#pragma unroll
    for (int v = 0; v < VX; ++v)
      vset<VX>(buf, v, lanes(need) ? vget<VX>(got, v) : vget<VX>(buf, v));
  }

  IT_LAP(2);
  vec_t out = buf;
  fuse_run<VX>(buf, out, hit, unit, tsdf, p.max_weight, max_w_hi);
  IT_LAP(3);
  // (an access that is switched off for the whole wave -- a layer past the chunk's end -- also gets a scalar
  // offset inside the descriptor: voffset 2^31 is then out of range whether the hardware compares
  // voffset >= num_records - soffset or voffset + soffset >= num_records)
  buf_store<VX, NT>(out, rs, lanes(hit_any) ? voff : kNoAccess, live ? soff : 0u);

  // look-ahead: updates are coherent along z, so a lane that updated this layer will very likely
  // update the layer two steps on (this buffer's next use); fetch it now
  buf_valid = ahead_ok ? hit_any : 0;
  buf = buf_load<VX, NT>(rs, lanes(buf_valid) ? voff : kNoAccess, ahead_ok ? soff + 2u * layer_bytes : 0u);
  IT_LAP(4);
}

// ---- surface map (yakhip_volume.surface_map) --------------------------------------------------------
// One bit per brick of 16 x 16 x 16 voxels of the slab: set = the brick may hold a voxel with a negative tsdf.
// The ray cast skips the fetches of samples in bricks whose bit is clear (raycast.hip).  A voxel can only turn
// negative when a negative measurement is folded into it (an average of values >= 0 is >= 0), and measurements
// are negative only behind the surface inside the truncation band: the walk collects the lanes that took one
// (`band`, a wave mask in SGPRs, updated on the band path of settle() only) and sets their brick's bit when it
// leaves the brick's layers.
// A second level (one bit per cell of 4 x 4 x 4 bricks, set with the first marking of a brick in or next to the cell)
// lets the ray cast cross the free space in front of the first marked cell in one go.
// Bits are only ever set here; yakhip_clear_volume clears them, yakhip_surface_map_rebuild recomputes them.
constexpr int kBrickShift = 4;  // YAKHIP_SURFACE_BRICK = 16

__device__ __forceinline__ void mark_bricks(const IntegrateParams& p, mask_t band, int x0, int y, int z_rel)
{
  if (band == 0)  // wave-uniform
    return;
  if (lanes(band))
  {
    const int bx = x0 >> kBrickShift, by = y >> kBrickShift, bz = z_rel >> kBrickShift;
    const uint32_t b = (uint32_t)bx + (uint32_t)p.bricks_x * ((uint32_t)by + (uint32_t)p.bricks_y * (uint32_t)bz);
    uint32_t* word = p.bricks + (b >> 5);
    const uint32_t bit = 1u << (b & 31u);
    if (!(__builtin_nontemporal_load(word) & bit))  // (a stale line only costs a redundant atomic)
      atomicOr(word, bit);
  }
}

// The second level for the lanes that marked a brick during a walk over the layers [zs_rel, ze_rel): after the walk
// (the loop's registers are dead by then), for every brick layer the walk touched -- a superset of the bricks
// actually marked, which is all a "may hold" bit has to be.
__device__ __forceinline__ void mark_cells(const IntegrateParams& p, mask_t marked, int x0, int y, int zs_rel, int ze_rel)
{
  if (marked == 0)  // wave-uniform
    return;
  if (lanes(marked))
    for (int bz = zs_rel >> kBrickShift; bz <= (ze_rel - 1) >> kBrickShift; ++bz)
      surface_mark_cells(p.bricks, p.bricks_words, p.bricks_x, p.bricks_y, p.bricks_z, x0 >> kBrickShift, y >> kBrickShift, bz);
}

// after the layers z, z + 1 (z - z0 even or odd) have been settled: flush `band` if the next layer starts another brick
// or the walk ends
__device__ __forceinline__ void flush_bricks(const IntegrateParams& p, mask_t& band, mask_t& marked, int x0, int y, int z,
                                             int step, int end)
{
  const int r = z - p.z0;
  if (((r + step) >> kBrickShift) != (r >> kBrickShift) || z + step >= end)
  {
    marked |= band;
    mark_bricks(p, band, x0, y, r);
    if (step > 1 && ((r + step - 1) >> kBrickShift) != (r >> kBrickShift) && z + step - 1 < end)
      mark_bricks(p, band, x0, y, r + step - 1);  // (an odd chunk start: the pair of layers straddles two bricks)
    band = 0;
  }
}

// Which voxels of a layer a lane owns.  tile_shift == 6: lanes in layer order, a wave = 64 * VX voxels of
// one x-row.  Smaller shifts fold the wave into a compact tile -- 2^shift lanes (2^shift * VX voxels,
// still >= 128 contiguous bytes) along x times 64 >> shift rows, the four waves of a workgroup stacked
// in y -- so that a surface cutting through the layer sends fewer waves down the per-voxel path
// (a wave takes it as soon as ONE lane is not settled by the group test).
template <int VX>
__device__ __forceinline__ bool lane_voxels(const IntegrateParams& p, uint32_t block_x, uint32_t wave, uint32_t& elem, int& x0,
                                            int& y)
{
  // (block_x, wave: the workgroup's place in the plane and the wave's in the workgroup -- blockIdx.x and
  // threadIdx.x >> 6 in the kernel that walks the grid, a listed brick's in integrate_mixed_kernel)
  const uint32_t lane = threadIdx.x & 63u;
  if (p.tile_shift >= 6)
  {
    const uint32_t group = block_x * 256u + wave * 64u + lane;  // VX-voxel group inside one z layer
    if (group >= p.plane / VX)
      return false;
    elem = group * VX;
    y = (int)(elem / (uint32_t)p.X);
    x0 = (int)(elem - (uint32_t)y * (uint32_t)p.X);
    return true;
  }
  // the launcher only picks a tile that divides the layer: every lane owns voxels
  const uint32_t lanes_x = 1u << p.tile_shift, rows_wave = 64u >> p.tile_shift;
  if (p.tile_side)
  {
    // the four waves side by side along x: the workgroup covers 4 * lanes_x * VX voxels of rows_wave rows, so
    // each row is one long contiguous run (memory sees what the layer-order footprint gives it) while every
    // wave's own patch stays compact (the brick test and the per-voxel path work wave by wave)
    const uint32_t blocks_per_row = (uint32_t)p.X / (4u * lanes_x * VX);
    const uint32_t bx = block_x % blocks_per_row, by = block_x / blocks_per_row;
    x0 = (int)(((bx * 4u + wave) * lanes_x + (lane & (lanes_x - 1u))) * VX);
    y = (int)(by * rows_wave + (lane >> p.tile_shift));
  }
  else
  {
    const uint32_t blocks_per_row = (uint32_t)p.X / (lanes_x * VX);
    const uint32_t bx = block_x % blocks_per_row, by = block_x / blocks_per_row;
    x0 = (int)((bx * lanes_x + (lane & (lanes_x - 1u))) * VX);
    y = (int)(by * 4u * rows_wave + wave * rows_wave + (lane >> p.tile_shift));
  }
  elem = (uint32_t)y * (uint32_t)p.X + (uint32_t)x0;
  return true;
}

// ---- brick test: settle a whole workgroup x z-chunk before it touches anything ------------------
// A wave owns a brick of voxels: its x-y footprint in the layer times the layers [zs, ze) of its chunk.
// The brick is convex, so in front of the camera its pixels lie in the bounding box of its 8 projected corners
// and its squared distances between the distance to its camera-space bounding box and the farthest corner.
// Three outcomes, each a sufficient condition for what the per-voxel path would decide voxel by voxel:
//   dead   no voxel can take an update: the brick is behind the camera (tsdf_volume.cu:89), projects outside
//          the image (:83), or lies behind the surface by more than the truncation distance at every pixel
//          it can see -- min d2 > max over the box of the pixels' B2 word (:94; a pixel without a return has
//          B2 = -1).  The wave exits: no replay, no memory access (and no LDS fill if no wave of the workgroup
//          is mixed).
//   front  every voxel takes the update with tsdf = 1: the box lies inside the image and max d2 <= min A2
//          over it.  The wave streams the chunk: load, weight bump (or general fuse), store.
//   mixed  anything else: the per-voxel path.
// The positions the kernel works with are the reference's replayed fp32 sums, not the ideal affine images:
// brick_eps bounds their distance per coordinate (host: (z1 + 32) * 2^-22 * P, P = largest |camera
// coordinate| in the volume -- each of the <= z1 + 8 roundings is below 2^-24 * 2P) and every bound below is
// widened by it; the pixel box by the matching projection error + 1/32 px for this routine's own roundings.

// Voxel index ranges of wave `wave` of workgroup `block_x` in a layer (see lane_voxels); false: the wave owns nothing.
template <int VX>
__device__ __forceinline__ bool wave_footprint(const IntegrateParams& p, uint32_t block_x, uint32_t wave, int& xlo, int& xhi,
                                               int& ylo, int& yhi)
{
  if (p.tile_shift >= 6)
  {
    const uint32_t e0 = (block_x * 4u + wave) * (64u * VX);
    uint32_t e1 = e0 + (64u * VX - 1u);
    const bool owns = e0 < p.plane;
    if (e1 >= p.plane)
      e1 = p.plane - 1u;  // (a wave that starts beyond the layer owns nothing: its class is never used)
    ylo = (int)(e0 / (uint32_t)p.X);
    yhi = (int)(e1 / (uint32_t)p.X);
    xlo = ylo == yhi ? (int)(e0 - (uint32_t)ylo * (uint32_t)p.X) : 0;
    xhi = ylo == yhi ? (int)(e1 - (uint32_t)yhi * (uint32_t)p.X) : p.X - 1;
    return owns;
  }
  const uint32_t lanes_x = 1u << p.tile_shift, rows_wave = 64u >> p.tile_shift;
  if (p.tile_side)
  {
    const uint32_t blocks_per_row = (uint32_t)p.X / (4u * lanes_x * VX);
    const uint32_t bx = block_x % blocks_per_row, by = block_x / blocks_per_row;
    xlo = (int)((bx * 4u + wave) * lanes_x * VX);
    ylo = (int)(by * rows_wave);
  }
  else
  {
    const uint32_t blocks_per_row = (uint32_t)p.X / (lanes_x * VX);
    const uint32_t bx = block_x % blocks_per_row, by = block_x / blocks_per_row;
    xlo = (int)(bx * lanes_x * VX);
    ylo = (int)((by * 4u + wave) * rows_wave);
  }
  xhi = xlo + (int)(lanes_x * VX) - 1;
  yhi = ylo + (int)rows_wave - 1;
  return true;
}

// The test itself for the box of voxels [xlo, xhi] x [ylo, yhi] x [zlo, zhi] (inclusive), done by the GROUP lanes of the
// calling lane's group (GROUP = 8 or 64; the box may differ from group to group).  FINE: the pixel box is looked up
// in the 4-pixel cells when it covers no more than kFineMaxCells of them, which leaves far less of a brick undecided
// beside a silhouette or the horizon (tools/sim_bricks.py: 16-pixel cells call 14.5 % of an orbit frame's wave bricks
// mixed, the voxels themselves 7 %).  Returns the class of the lane's group.
template <int GROUP, bool FINE>
__device__ __forceinline__ int classify_box(const IntegrateParams& p, int xlo, int xhi, int ylo, int yhi, int zlo, int zhi)
{
  // corner (lane & 7) of the box in camera space (every lane group of 8 computes the same 8 corners)
  const uint32_t lane = threadIdx.x & 63u;
  const float vx = (float)((lane & 1u) ? xhi : xlo) * p.vsx;
  const float vy = (float)((lane & 2u) ? yhi : ylo) * p.vsy;
  const float vz = (float)((lane & 4u) ? zhi : zlo) * p.vsz;
  const float cx = dot3(p.R[0], p.R[1], p.R[2], vx, vy, vz) + p.t[0];
  const float cy = dot3(p.R[3], p.R[4], p.R[5], vx, vy, vz) + p.t[1];
  const float cz = dot3(p.R[6], p.R[7], p.R[8], vx, vy, vz) + p.t[2];
  const float e = p.brick_eps;
  float xmin = cx, xmax = cx, ymin = cy, ymax = cy, zmin = cz, zmax = cz;
  const float ax = __builtin_fabsf(cx) + e, ay = __builtin_fabsf(cy) + e, az = __builtin_fabsf(cz) + e;
  float d2max = __builtin_fmaf(ax, ax, __builtin_fmaf(ay, ay, az * az));
  // projection of the corner (only meaningful, and only used, when the whole brick has z > brick_zmin; the hardware
  // reciprocal is good to 1 ulp, 1e-4 px in a 1000 px image: inside the 1/32 px the box is widened by for such things)
  const float iz = __builtin_amdgcn_rcpf(cz);
  float umin = __builtin_fmaf(p.fx, cx * iz, p.cx), vmin = __builtin_fmaf(p.fy, cy * iz, p.cy);
  float umax = umin, vmax = vmin;
#pragma unroll
  for (int off = 1; off < 8; off <<= 1)
  {
    xmin = __builtin_fminf(xmin, __shfl_xor(xmin, off));
    xmax = __builtin_fmaxf(xmax, __shfl_xor(xmax, off));
    ymin = __builtin_fminf(ymin, __shfl_xor(ymin, off));
    ymax = __builtin_fmaxf(ymax, __shfl_xor(ymax, off));
    zmin = __builtin_fminf(zmin, __shfl_xor(zmin, off));
    zmax = __builtin_fmaxf(zmax, __shfl_xor(zmax, off));
    umin = __builtin_fminf(umin, __shfl_xor(umin, off));
    umax = __builtin_fmaxf(umax, __shfl_xor(umax, off));
    vmin = __builtin_fminf(vmin, __shfl_xor(vmin, off));
    vmax = __builtin_fmaxf(vmax, __shfl_xor(vmax, off));
    d2max = __builtin_fmaxf(d2max, __shfl_xor(d2max, off));
  }
  xmin -= e, ymin -= e, zmin -= e, xmax += e, ymax += e, zmax += e;
  int cls = kBrickMixed;
  if (zmax <= 0.f)
    cls = kBrickDead;  // every voxel has vc.z <= 0
  else if (zmin >= p.brick_zmin)
  {
    const float xabs = __builtin_fmaxf(__builtin_fabsf(xmin), __builtin_fabsf(xmax));
    const float yabs = __builtin_fmaxf(__builtin_fabsf(ymin), __builtin_fabsf(ymax));
    const float rz = __builtin_amdgcn_rcpf(zmin - e);  // (the factor 1.01 below covers its ulp)
    const float du = __builtin_fmaf(p.fx * e * (1.f + xabs * rz), rz * 1.01f, 1.f / 32.f);
    const float dv = __builtin_fmaf(p.fy * e * (1.f + yabs * rz), rz * 1.01f, 1.f / 32.f);
    umin -= du, umax += du, vmin -= dv, vmax += dv;
    const float fc = (float)p.cols, fr = (float)p.rows;
    if (umax < 0.f || vmax < 0.f || umin >= fc || vmin >= fr)
      cls = kBrickDead;  // every voxel projects outside the image
    else
    {
      // cells the box touches, clipped to the image
      const int px0 = (int)__builtin_fmaxf(__builtin_floorf(umin), 0.f), px1 = (int)__builtin_fminf(__builtin_floorf(umax), fc - 1.f);
      const int py0 = (int)__builtin_fmaxf(__builtin_floorf(vmin), 0.f), py1 = (int)__builtin_fminf(__builtin_floorf(vmax), fr - 1.f);
      int cx0 = px0 / kBrickCell, cy0 = py0 / kBrickCell;
      int ncx = px1 / kBrickCell - cx0 + 1, ncy = py1 / kBrickCell - cy0 + 1;
      int pitch = p.ccw;
      const float *lo_plane = p.cellmin, *hi_plane = p.cellmax;
      if constexpr (FINE)
      {
        const int fx0 = px0 >> kFineShift, fy0 = py0 >> kFineShift;
        const int nfx = (px1 >> kFineShift) - fx0 + 1, nfy = (py1 >> kFineShift) - fy0 + 1;
        if (nfx * nfy <= kFineMaxCells)
          cx0 = fx0, cy0 = fy0, ncx = nfx, ncy = nfy, pitch = p.fcw, lo_plane = p.finemin, hi_plane = p.finemax;
      }
      const int n = ncx * ncy;
      if (n <= 256)
      {
        float amin = __builtin_inff(), bmax = -__builtin_inff();
        // (four cells per trip, their loads issued together: the trips of a piece are round trips to L2 one after
        //  the other otherwise, the larger part of the test's time)
        const float rncx = __builtin_amdgcn_rcpf((float)ncx);
        for (int i0 = (int)(lane & (GROUP - 1)); i0 < n; i0 += 4 * GROUP)
        {
          float lo[4], hi[4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
          {
            const int i = i0 + u * GROUP;
            const bool on = i < n;
            // i / ncx for 0 <= i < 65536, ncx <= 256: the rounded-down float quotient is exact or one too large
            int yy = (int)((float)i * rncx);
            yy -= yy * ncx > i ? 1 : 0;
            yy += (yy + 1) * ncx <= i ? 1 : 0;
            const int xx = i - yy * ncx;
            const int idx = on ? (cy0 + yy) * pitch + cx0 + xx : cy0 * pitch + cx0;
            lo[u] = lo_plane[idx];
            hi[u] = hi_plane[idx];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
          {
            amin = __builtin_fminf(amin, lo[u]);  // (a lane past the end re-reads the box's first cell: harmless)
            bmax = __builtin_fmaxf(bmax, hi[u]);
          }
        }
#pragma unroll
        for (int off = GROUP / 2; off > 0; off >>= 1)
        {
          amin = __builtin_fminf(amin, __shfl_xor(amin, off));
          bmax = __builtin_fmaxf(bmax, __shfl_xor(bmax, off));
        }
        // squared distance from the camera to the brick's (widened) camera-space bounding box
        const float mx = xmin > 0.f ? xmin : (xmax < 0.f ? -xmax : 0.f);
        const float my = ymin > 0.f ? ymin : (ymax < 0.f ? -ymax : 0.f);
        const float d2min = __builtin_fmaf(mx, mx, __builtin_fmaf(my, my, zmin * zmin)) * 0x1.ffep-1f;  // 1 - 2^-11
        const bool inside = umin >= 0.5f && vmin >= 0.5f && umax < fc - 0.5f && vmax < fr - 0.5f;
        if (d2min > bmax)
          cls = kBrickDead;  // behind the surface (or no return) at every pixel the brick can see
        else if (inside && d2max * 0x1.002p+0f <= amin)
          cls = kBrickFront;  // in front of the surface by more than the truncation distance, everywhere
      }
    }
  }
  return cls;
}

// GROUP = 64: the calling wave settles its own brick, all lanes sharing the work.  GROUP = 8: every group of eight
// lanes of the calling wave settles the brick of wave (lane / 8) of the workgroup (groups 4..7 repeat 0..3), so one
// wave does the test for all four -- a quarter of the instructions, and on an orbit frame, where 82 % of the waves
// are dead, the test is a sixth of all the kernel issues.  Returns the class of the lane's group.
template <int VX, int GROUP>
__device__ __forceinline__ int classify_brick(const IntegrateParams& p, uint32_t block_x, int zs, int ze)
{
  int xlo, xhi, ylo, yhi;
  const uint32_t wave = GROUP == 64 ? threadIdx.x >> 6 : (threadIdx.x >> 3) & 3u;
  wave_footprint<VX>(p, block_x, wave, xlo, xhi, ylo, yhi);
  const int cls = classify_box<GROUP, false>(p, xlo, xhi, ylo, yhi, zs, ze - 1);
  if constexpr (GROUP == 64)
    return __builtin_amdgcn_readfirstlane(cls);
  else
    return cls;
}

// A mixed brick once more, in up to eight pieces along z of 1 << shift layers each -- piece g by lanes 8g .. 8g + 7,
// against the fine cells.  A piece's pixel box is a fraction of the brick's and the planes it is looked up in are
// sixteen times finer, so most of what the first test left undecided next to a silhouette, the horizon or the image
// border is settled here: on an orbit frame 40 % of a mixed brick's layers (tools/sim_bricks.py).  Returns the lanes of
// the dead and of the front pieces as wave masks (all eight lanes of a group alike; groups past the brick's last
// layer count as dead).
template <int VX>
__device__ __forceinline__ void classify_pieces(const IntegrateParams& p, uint32_t block_x, uint32_t wave, int zs, int ze,
                                                int shift, mask_t& dead, mask_t& front)
{
  int xlo, xhi, ylo, yhi;
  wave_footprint<VX>(p, block_x, wave, xlo, xhi, ylo, yhi);
  const int g = (int)((threadIdx.x & 63u) >> 3);
  const int zlo = zs + (g << shift);
  int zhi = zlo + (1 << shift) - 1;
  if (zhi > ze - 1)
    zhi = ze - 1;
  const bool empty = zlo >= ze;
  const int cls = classify_box<8, true>(p, xlo, xhi, ylo, yhi, empty ? zs : zlo, empty ? zs : zhi);
  dead = ballot(empty || cls == kBrickDead);
  front = ballot(!empty && cls == kBrickFront);
}

// A "front" brick: every voxel of the chunk takes the measurement 1.0.  Plain streaming: two layers in flight.
template <int VX, bool NT>
__device__ __forceinline__ void walk_front(const IntegrateParams& p, rsrc_t rs, uint32_t voff, uint32_t layer_bytes,
                                           int layers, mask_t active, uint32_t max_w_hi)
{
  typedef typename Vec<VX>::type vec_t;
  mask_t hit[VX], unit[VX];
  float tsdf[VX];
#pragma unroll
  for (int v = 0; v < VX; ++v)
  {
    hit[v] = unit[v] = active;
    tsdf[v] = 1.f;
  }
  vec_t b0 = buf_load<VX, NT>(rs, voff, 0u);
  vec_t b1 = buf_load<VX, NT>(rs, layers > 1 ? voff : kNoAccess, layers > 1 ? layer_bytes : 0u);
  uint32_t soff = 0;
  for (int i = 0; i < layers; i += 2, soff += 2u * layer_bytes)
  {
    vec_t o0 = b0;
    fuse_run<VX>(b0, o0, hit, unit, tsdf, p.max_weight, max_w_hi);
    buf_store<VX, NT>(o0, rs, voff, soff);
    const bool a2 = i + 2 < layers, a1 = i + 1 < layers, a3 = i + 3 < layers;
    b0 = buf_load<VX, NT>(rs, a2 ? voff : kNoAccess, a2 ? soff + 2u * layer_bytes : 0u);
    vec_t o1 = b1;
    fuse_run<VX>(b1, o1, hit, unit, tsdf, p.max_weight, max_w_hi);
    buf_store<VX, NT>(o1, rs, a1 ? voff : kNoAccess, a1 ? soff + layer_bytes : 0u);
    b1 = buf_load<VX, NT>(rs, a3 ? voff : kNoAccess, a3 ? soff + 3u * layer_bytes : 0u);
  }
}

// The same with DEPTH layers in flight, for launches where front bricks are few and far between (a launch with a list:
// a surface-rich frame): a wave that streams alone is bound by the round trip to memory, one per two layers above --
// 24 layers took a front wave 15 to 37 us, the longest jobs of the first kernel.
template <int VX, bool NT, int DEPTH>
__device__ __forceinline__ void walk_front_deep(const IntegrateParams& p, rsrc_t rs, uint32_t voff, uint32_t layer_bytes,
                                                int layers, mask_t active, uint32_t max_w_hi)
{
  typedef typename Vec<VX>::type vec_t;
  mask_t hit[VX], unit[VX];
  float tsdf[VX];
#pragma unroll
  for (int v = 0; v < VX; ++v)
  {
    hit[v] = unit[v] = active;
    tsdf[v] = 1.f;
  }
  vec_t b[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    b[d] = buf_load<VX, NT>(rs, d < layers ? voff : kNoAccess, d < layers ? (uint32_t)d * layer_bytes : 0u);
  uint32_t soff = 0;
  for (int i = 0; i < layers; i += DEPTH, soff += (uint32_t)DEPTH * layer_bytes)
  {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
    {
      const bool on = i + d < layers, ahead = i + d + DEPTH < layers;
      vec_t o = b[d];
      fuse_run<VX>(b[d], o, hit, unit, tsdf, p.max_weight, max_w_hi);
      buf_store<VX, NT>(o, rs, on ? voff : kNoAccess, on ? soff + (uint32_t)d * layer_bytes : 0u);
      b[d] = buf_load<VX, NT>(rs, ahead ? voff : kNoAccess, ahead ? soff + (uint32_t)(d + DEPTH) * layer_bytes : 0u);
    }
  }
}


// The per-voxel walk of the layers [ws, we) of wave `wave` of workgroup `block_x`: VX voxels per lane along x.  The z
// loop is software-pipelined two layers deep with two register sets (A, B): while layer z is being settled and
// written, layer z+1 is already tested/projected (its threshold gathers in flight), and the voxels of layers
// z+1 / z+2 are being fetched.
// piece_dead / piece_front (classify_pieces; 0 / 0 = nothing known): the class of the pieces of 1 << piece_shift layers
// counted from layer zs, eight mask bits per piece.
template <int VX, bool NT, bool PIECES>
__device__ __forceinline__ void walk_mixed(const IntegrateParams& p, const float* coarse_lds, uint32_t block_x, uint32_t wave,
                                           int zs, int ws, int we, mask_t piece_dead, mask_t piece_front, int piece_shift
#ifdef YAK_INTEGRATE_TIMING
                                           , unsigned long long& it_replay, size_t it_slot
#endif
)
{
  typedef typename Vec<VX>::type vec_t;
  uint32_t elem;  // x + y*X of the lane's first voxel; X % VX == 0 so no row straddle
  int x0, y;
  if (!lane_voxels<VX>(p, block_x, wave, elem, x0, y))
    return;
  const mask_t active = ballot(true);

  // :69
  const float zsx = p.R[2] * p.vsz, zsy = p.R[5] * p.vsz, zsz = p.R[8] * p.vsz;
  f32x2 sx = { zsx, zsx }, sy = { zsy, zsy }, sz = { zsz, zsz };
  asm volatile("" : "+v"(sx), "+v"(sy), "+v"(sz));  // keep the splats resident (no per-use rebuild)

  // :71-72 (a pair slot that holds no voxel, VX == 1, shadows voxel 0 and is never stored)
  Pos<VX> A, B;
  const float vxy = (float)y * p.vsy;
#pragma unroll
  for (int k = 0; k < Pos<VX>::NP; ++k)
#pragma unroll
    for (int h = 0; h < 2; ++h)
    {
      const int v = Pos<VX>::voxel(k, h);
      const float vxx = (float)(x0 + (v < VX ? v : 0)) * p.vsx;
      A.x[k][h] = dot3(p.R[0], p.R[1], p.R[2], vxx, vxy, 0.f) + p.t[0];
      A.y[k][h] = dot3(p.R[3], p.R[4], p.R[5], vxx, vxy, 0.f) + p.t[1];
      A.z[k][h] = dot3(p.R[6], p.R[7], p.R[8], vxx, vxy, 0.f) + p.t[2];
    }

  // Replay the additions the reference performed on layers [0, ws) (tsdf_volume.cu:76), ws of them here -- from layer 0, or
  // for a slab deep inside a tall volume from the column positions at z0 that checkpoint_kernel left.
  int replay = ws;
  if (PIECES && p.cptab)
  {
    // ... or, in a launch with a list, from the checkpoint below ws that the first kernel wrote
    if constexpr (VX == 4)
    {
      const int k = (ws - p.z0) / p.cp_stride;
      const float* at = p.cptab + (size_t)k * 3u * p.plane + elem;
      const f32x4 cx = *(const f32x4*)at, cy = *(const f32x4*)(at + p.plane), cz = *(const f32x4*)(at + 2u * p.plane);
      A.x[0] = f32x2{ cx[0], cx[3] }, A.x[1] = f32x2{ cx[1], cx[2] };
      A.y[0] = f32x2{ cy[0], cy[3] }, A.y[1] = f32x2{ cy[1], cy[2] };
      A.z[0] = f32x2{ cz[0], cz[3] }, A.z[1] = f32x2{ cz[1], cz[2] };
      replay = ws - p.z0 - k * p.cp_stride;
    }
  }
  else if (p.ckpt)
  {
#pragma unroll
    for (int k = 0; k < Pos<VX>::NP; ++k)
#pragma unroll
      for (int h = 0; h < 2; ++h)
      {
        const int v = Pos<VX>::voxel(k, h);
        const uint32_t col = elem + (uint32_t)(v < VX ? v : 0);
        A.x[k][h] = p.ckpt[col];
        A.y[k][h] = p.ckpt[p.plane + col];
        A.z[k][h] = p.ckpt[2u * p.plane + col];
      }
    replay = ws - p.z0;
  }
#ifdef YAK_INTEGRATE_TIMING
  unsigned long long it_a, it_b;
  IT_CLOCK(it_a);
#endif
  for (int i = 0; i < replay; ++i)
    advance<VX>(A, A, sx, sy, sz);
#ifdef YAK_INTEGRATE_TIMING
  IT_CLOCK(it_b);
  it_replay = it_b - it_a;
#endif

  const f32x2 fx2 = { p.fx, p.fx }, fy2 = { p.fy, p.fy }, cx2 = { p.cx, p.cx }, cy2 = { p.cy, p.cy };
  const uint32_t max_w_hi = ((uint32_t)p.max_weight << 16) | 0xffffu;

  // descriptor over the walked layers [ws, we); the launcher keeps a chunk below 2 GiB
  const uint32_t layer_bytes = p.plane * 4u;
  const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.vol + (size_t)(ws - p.z0) * p.plane, 0,
                                                      (uint32_t)(we - ws) * layer_bytes, 0x00020000);
  const uint32_t voff = elem * 4u;
  // what classify_pieces said of layer z (wave-uniform); layers past the walk's end count as dead: stage 1 runs a
  // layer ahead, and a dead layer costs it nothing
  auto layer_class = [&](int z) -> int {
    if constexpr (!PIECES)
      return kBrickMixed;  // (stage 1 is then not guarded at the walk's end: a projection past `we` reads valid
                           //  threshold texels and is discarded)
    if (z >= we)
      return kBrickDead;
    const int bit = ((z - zs) >> piece_shift) << 3;
    return ((piece_dead >> bit) & 1ull) ? kBrickDead : (((piece_front >> bit) & 1ull) ? kBrickFront : kBrickMixed);
  };

  // Prologue, shaped like one steady-state half trip so that the compiler's s_waitcnt counting sees
  // the same queue at loop entry as on the back edge: [voxel fetch] [stage 1] [store] [voxel fetch].
  // The first two layers are fetched for every lane (nothing is known about them yet); the store is
  // a no-op (every lane out of range).
  Probe<VX> PA, PB;
  mask_t validA = ~0ull, validB = ws + 1 < we ? ~0ull : 0ull;
  vec_t bufA = buf_load<VX, NT>(rs, voff, 0u);
  __builtin_amdgcn_sched_barrier(0);
  stage1<VX>(p, A, PA, coarse_lds, active, fx2, fy2, cx2, cy2, layer_class(ws));
  __builtin_amdgcn_sched_barrier(0);
  buf_store<VX, NT>(bufA, rs, kNoAccess, 0u);
  vec_t bufB = buf_load<VX, NT>(rs, validB ? voff : kNoAccess, validB ? layer_bytes : 0u);
  __builtin_amdgcn_sched_barrier(0);
  uint32_t soff = 0;
  mask_t band = 0;  // lanes that folded a negative measurement into a voxel since the last flush (surface map)
  mask_t marked = 0;  // ... at any time during this walk
#ifdef YAK_INTEGRATE_TIMING
  ItPhase itp = {};
#endif
  for (int z = ws; z < we; z += 2, soff += 2u * layer_bytes)
  {
    // layer z lives in set A, layer z+1 in set B.  Stage 1 runs one layer ahead.
    {
      IT_LAP_START;
      advance<VX>(A, B, sx, sy, sz);
      stage1<VX>(p, B, PB, coarse_lds, active, fx2, fy2, cx2, cy2, layer_class(z + 1));
      IT_LAP(0);
    }
    update<VX, NT>(p, A, PA, bufA, validA, rs, voff, soff, layer_bytes, true, z + 2 < we, max_w_hi, band IT_PHASE_ARG);
    {
      IT_LAP_START;
      advance<VX>(B, A, sx, sy, sz);
      stage1<VX>(p, A, PA, coarse_lds, active, fx2, fy2, cx2, cy2, layer_class(z + 2));
      IT_LAP(0);
    }
    update<VX, NT>(p, B, PB, bufB, validB, rs, voff, soff + layer_bytes, layer_bytes, z + 1 < we, z + 3 < we,
                   max_w_hi, band IT_PHASE_ARG);
    if (p.bricks)
      flush_bricks(p, band, marked, x0, y, z, 2, we);
  }
  if (p.bricks)
    mark_cells(p, marked, x0, y, ws - p.z0, we - p.z0);
#ifdef YAK_INTEGRATE_TIMING
  if ((threadIdx.x & 63u) == 0)
    for (int i = 0; i < 8; ++i)
      g_it_phase[it_slot & (kItSlots - 1)][i] = itp.t[i];
#endif
}

// Column checkpoints of a launch with a list (IntegrateParams::cptab): the camera-space positions of columns
// col .. col + 3 at the layers z0, z0 + cp_stride, ... exactly as the walk would have them there.
__device__ __forceinline__ void checkpoint_table(const IntegrateParams& p, uint32_t col)
{
  if (col >= p.plane)
    return;
  const int y = (int)(col / (uint32_t)p.X), x0 = (int)(col - (uint32_t)y * (uint32_t)p.X);
  const float zsx = p.R[2] * p.vsz, zsy = p.R[5] * p.vsz, zsz = p.R[8] * p.vsz;  // :69
  const f32x2 sx = { zsx, zsx }, sy = { zsy, zsy }, sz = { zsz, zsz };
  Pos<4> A;
  const float vxy = (float)y * p.vsy;
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int h = 0; h < 2; ++h)
    {
      const int v = Pos<4>::voxel(k, h);
      if (p.ckpt)
      {
        A.x[k][h] = p.ckpt[col + v], A.y[k][h] = p.ckpt[p.plane + col + v], A.z[k][h] = p.ckpt[2u * p.plane + col + v];
      }
      else
      {
        const float vxx = (float)(x0 + v) * p.vsx;  // :71-72
        A.x[k][h] = dot3(p.R[0], p.R[1], p.R[2], vxx, vxy, 0.f) + p.t[0];
        A.y[k][h] = dot3(p.R[3], p.R[4], p.R[5], vxx, vxy, 0.f) + p.t[1];
        A.z[k][h] = dot3(p.R[6], p.R[7], p.R[8], vxx, vxy, 0.f) + p.t[2];
      }
    }
  if (p.cp_ffwd)
  {
    // a deep slab: the positions at z0 in closed form (ffwd.hpp: bit for bit the z0 rounded additions), as
    // checkpoint_column has them for the one-kernel launch
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int h = 0; h < 2; ++h)
      {
        A.x[k][h] = yak_ffwd_accumulate(A.x[k][h], zsx, p.z0);
        A.y[k][h] = yak_ffwd_accumulate(A.y[k][h], zsy, p.z0);
        A.z[k][h] = yak_ffwd_accumulate(A.z[k][h], zsz, p.z0);
      }
  }
  else if (!p.ckpt)
    for (int i = 0; i < p.z0; ++i)  // (a slab that starts a few layers up)
      advance<4>(A, A, sx, sy, sz);
  float* at = p.cptab + col;
  for (int k = 0; k < p.cp_count; ++k, at += 3u * p.plane)
  {
    *(f32x4*)at = f32x4{ A.x[0][0], A.x[1][0], A.x[1][1], A.x[0][1] };
    *(f32x4*)(at + p.plane) = f32x4{ A.y[0][0], A.y[1][0], A.y[1][1], A.y[0][1] };
    *(f32x4*)(at + 2u * p.plane) = f32x4{ A.z[0][0], A.z[1][0], A.z[1][1], A.z[0][1] };
    if (k + 1 < p.cp_count)
      for (int i = 0; i < p.cp_stride; ++i)
        advance<4>(A, A, sx, sy, sz);
  }
}

// The launch's grid walks (plane tiles) x (z chunks).  Every workgroup settles its four wave bricks first: a dead
// brick ends there, a front brick is streamed, a mixed one is walked voxel by voxel.  This is the whole launch for
// frames that are mostly free space (the wall scene: the roofline case) and for small volumes; a surface-rich frame
// of a large volume goes through the two kernels below instead.
template <int VX, bool NT>
__global__ __launch_bounds__(256, 4) void integrate_kernel(const IntegrateParams p, const ChunkTable chunks)
{
  IT_BEGIN;
  extern __shared__ __attribute__((aligned(16))) float coarse_lds[];

  // The chunk this workgroup walks comes from the launcher's table (chunk_table below): layers [zs, ze), in the
  // order the launcher wants them dispatched.
  const int zs = p.z0 + __builtin_amdgcn_readfirstlane((int)chunks.first[blockIdx.y]);
  const int ze = zs + __builtin_amdgcn_readfirstlane((int)chunks.layers[blockIdx.y]);
  const uint32_t wave = threadIdx.x >> 6;
#ifdef YAK_INTEGRATE_TIMING
  const size_t it_slot = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave;
#endif

  // brick test: the class of every wave's brick, settled by the first wave for all four (eight lanes per brick) and
  // handed over through LDS
  __shared__ int wave_class[4];
  int cls = kBrickMixed;
  bool fill = VX == 4 && p.grp_enable;
  if (p.brick_enable)
  {
    if (threadIdx.x < 64u)
    {
      const int c = classify_brick<VX, 8>(p, blockIdx.x, zs, ze);
      if (threadIdx.x < 32u && (threadIdx.x & 7u) == 0)
      {
        wave_class[threadIdx.x >> 3] = c;
        if (p.classes)  // what this launch found, for the launcher's choice of kernels for a later frame (class bytes
        {               // as classify_kernel leaves them)
          int xlo, xhi, ylo, yhi;
          const bool owns = wave_footprint<VX>(p, blockIdx.x, threadIdx.x >> 3, xlo, xhi, ylo, yhi);
          const int pl = part_layers(ze - zs);
          const uint32_t parts = (uint32_t)((ze - zs + pl - 1) / pl);
          p.classes[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4u + (threadIdx.x >> 3)] =
              (uint8_t)(c == kBrickDead || !owns ? 0u : (c == kBrickFront ? parts | 0x80u : parts));
        }
      }
    }
    __syncthreads();
    cls = __builtin_amdgcn_readfirstlane(wave_class[wave]);
    // The coarse threshold image (a few KiB, read once per workgroup into LDS) is only needed by mixed bricks.
    fill = fill && (wave_class[0] == kBrickMixed || wave_class[1] == kBrickMixed || wave_class[2] == kBrickMixed ||
                    wave_class[3] == kBrickMixed);
  }
  if (p.brick_stats && (threadIdx.x & 63u) == 0)  // measurement aid, off by default
    atomicAdd(&p.brick_stats[cls], 1u);
  if (fill)
  {
    const int n = p.cw * p.ch;
    for (int i = (int)threadIdx.x; i < n; i += 256)
      coarse_lds[i] = p.coarse[i];
    __syncthreads();
  }
  IT_CLOCK(it_t1);
  if (cls == kBrickDead)
  {
    IT_ROW(it_slot, cls, it_t1, 0, 0);
    return;
  }
  if (cls == kBrickFront)
  {
    uint32_t elem;
    int x0, y;
    if (!lane_voxels<VX>(p, blockIdx.x, wave, elem, x0, y))
      return;
    const uint32_t lb = p.plane * 4u;
    const rsrc_t rsf = __builtin_amdgcn_make_buffer_rsrc(p.vol + (size_t)(zs - p.z0) * p.plane, 0,
                                                         (uint32_t)(ze - zs) * lb, 0x00020000);
    walk_front<VX, NT>(p, rsf, elem * 4u, lb, ze - zs, ballot(true), ((uint32_t)p.max_weight << 16) | 0xffffu);
    IT_ROW(it_slot, cls, it_t1, 0, ze - zs);
    return;
  }
#ifdef YAK_INTEGRATE_TIMING
  walk_mixed<VX, NT, false>(p, coarse_lds, blockIdx.x, wave, zs, zs, ze, 0, 0, 31, it_t3, it_slot);
#else
  walk_mixed<VX, NT, false>(p, coarse_lds, blockIdx.x, wave, zs, zs, ze, 0, 0, 31);  // (nothing known of its pieces)
#endif
  IT_ROW(it_slot, cls, it_t1, it_t3, ze - zs);
}

// ---- a launch in two kernels: surface-rich frames of large volumes --------------------------------
// Mixed bricks are few (4 000 of an orbit frame's 30 000 wave bricks), cost ten to forty times a front brick, and sit
// where the surfaces are.  Walked in place by the grid above, all of them are resident within the first eighth of
// the launch -- ten to a SIMD here, none there, every one a chain of 24 layers behind a replay of up to 500 steps --
// and the launch lasts as long as its most loaded SIMD (per-wave clocks of an orbit view: SIMDs end anywhere between
// 68 and 125 us of 125).  So the work is cut finer and dealt out evenly:
//   classify_kernel        the brick test for every wave brick of the grid above, eight lanes per brick, the classes
//                          left in memory (a byte per brick); and, by extra workgroups, the camera-space positions of
//                          every column every cp_stride layers, so that nobody replays from layer 0 again
//   integrate_list_kernel  a fixed number of workgroups; each reads all the classes, numbers the live bricks' parts
//                          (kPartLayers layers each) the same way as every other workgroup does, keeps every G-th one
//                          -- no list in memory, no atomics, every workgroup the same count -- and its four waves take
//                          them from a counter in LDS as they come free.
// The voxels come out word for word as from the grid above: the same walk over the same layers from the same positions.
constexpr uint32_t kClassUnits = 7u, kClassFront = 0x80u;  // a class byte: parts to do (0 = dead) | front

template <int VX>
__global__ __launch_bounds__(256) void classify_kernel(const IntegrateParams p, const ChunkTable chunks, uint32_t rows_x,
                                                       uint32_t cp_rows)
{
  static_assert(VX == 4, "the two-kernel launch is for the 16-byte layout");
  // A slab that starts at layer 0 gets its column checkpoints from here, for every column, by cp_rows workgroups in
  // front of the others (dispatched first: the launch's longest jobs, 512 steps one after the other; they run beside
  // the brick tests, which leave the vector units idle).  A deep slab's start from closed-form positions costs
  // more per column: checkpoint_tiles_kernel makes them afterwards, for the columns that need them.
  if (blockIdx.x < cp_rows)
  {
    checkpoint_table(p, (blockIdx.x * 256u + threadIdx.x) * 4u);
    return;
  }
  // rows_x workgroups per chunk, 32 wave bricks each
  const uint32_t id = blockIdx.x - cp_rows;
  const uint32_t cj = id / rows_x, in_chunk = (id - cj * rows_x) * 32u + (threadIdx.x >> 3);
  const int zs = p.z0 + __builtin_amdgcn_readfirstlane((int)chunks.first[cj]);
  const int ze = zs + __builtin_amdgcn_readfirstlane((int)chunks.layers[cj]);
  if (in_chunk >= p.grid_x * 4u)
    return;
  int xlo, xhi, ylo, yhi;
  const bool owns = wave_footprint<VX>(p, in_chunk >> 2, in_chunk & 3u, xlo, xhi, ylo, yhi);
  int cls = classify_box<8, false>(p, xlo, xhi, ylo, yhi, zs, ze - 1);
  if (!owns)
    cls = kBrickDead;
  if ((threadIdx.x & 7u) == 0)
  {
    const int pl = part_layers(ze - zs);
    const uint32_t parts = (uint32_t)((ze - zs + pl - 1) / pl);  // <= 4
    p.classes[(size_t)cj * p.grid_x * 4u + in_chunk] =
        (uint8_t)(cls == kBrickDead ? 0u : (cls == kBrickFront ? parts | kClassFront : parts));
    if (p.brick_stats)  // measurement aid, off by default
      atomicAdd(&p.brick_stats[cls], 1u);
  }
}

// Between the two: the column checkpoints, for the columns that need them -- those of a wave footprint with a mixed
// brick in some chunk (front bricks are streamed: no positions).  One workgroup per plane tile, lane = four columns,
// positions as the walk forms and advances them (tsdf_volume.cu:69-76), stored every cp_stride layers.  On a slab
// of configuration 5 a fifth of the columns; the table for all of them was 200 MB written per frame.
template <int VX>
__global__ __launch_bounds__(256) void checkpoint_tiles_kernel(const IntegrateParams p, uint32_t n_chunks)
{
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  bool mixed = false;
  for (uint32_t cj = lane; cj < n_chunks; cj += 64u)
  {
    const uint32_t c = p.classes[((size_t)cj * p.grid_x + blockIdx.x) * 4u + wave];
    mixed = mixed || ((c & kClassUnits) != 0 && (c & kClassFront) == 0);
  }
  if (ballot(mixed) == 0)
    return;
  uint32_t elem;
  int x0, y;
  if (!lane_voxels<VX>(p, blockIdx.x, wave, elem, x0, y))
    return;
  checkpoint_table(p, elem);
}

template <int VX, bool NT>
__global__ __launch_bounds__(256, 5) void integrate_list_kernel(const IntegrateParams p_arg, const ChunkTable chunks, int refine,
                                                                uint32_t list_cap)
{
  extern __shared__ __attribute__((aligned(16))) float coarse_lds[];
  // The parameters are read through a pointer the compiler cannot see through, renewed for every part: otherwise
  // it loads all of them once, in front of the loop over the parts, and keeps them in scalar registers through the
  // walk, which then spills a hundred of them (measured: 128 spilled SGPRs and 37 VGPRs to scratch; 24 / 0 so).
  // p_arg is the kernel's first argument: it starts the kernarg segment.
  typedef const __attribute__((address_space(4))) IntegrateParams* params_ptr;
  params_ptr pq = (params_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(pq));
#define p (*(const IntegrateParams*)pq)
  const int coarse_n = VX == 4 && p.grp_enable ? p.cw * p.ch : 0;
  uint32_t* my_parts = (uint32_t*)(coarse_lds + coarse_n);  // this workgroup's share, list_cap entries
  __shared__ uint32_t wave_sum[4], share_n, share_next;
  for (int i = (int)threadIdx.x; i < coarse_n; i += 256)
    coarse_lds[i] = p.coarse[i];

  // Number the parts of the live bricks, the same way in every workgroup: thread t counts the bricks 4 t .. 4 t + 3 of
  // every group of 1024 (one word of four class bytes per group), the threads' counts are summed up in thread order,
  // and a part whose number is g modulo G belongs to workgroup g.
  const uint32_t G = gridDim.x, g = blockIdx.x;
  const uint32_t groups = (p.n_bricks + 1023u) >> 10;
  const uint32_t* words = (const uint32_t*)p.classes;  // (padded with zeros to whole groups)
  uint32_t mine = 0;
  for (uint32_t i = 0; i < groups; ++i)
  {
    const uint32_t w = words[i * 256u + threadIdx.x] & 0x07070707u;
    mine += (w * 0x01010101u) >> 24;  // the four counts added up
  }
  uint32_t incl = mine;  // inclusive scan over the wave, then over the four waves
#pragma unroll
  for (int off = 1; off < 64; off <<= 1)
  {
    const uint32_t up = __shfl_up(incl, off);
    if ((int)(threadIdx.x & 63u) >= off)
      incl += up;
  }
  if ((threadIdx.x & 63u) == 63u)
    wave_sum[threadIdx.x >> 6] = incl;
  if (threadIdx.x == 0)
    share_next = 0;
  __syncthreads();
  uint32_t number = incl - mine;  // of this thread's first part
  for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w)
    number += wave_sum[w];
  const uint32_t total = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
  if (total == 0)  // (every wave of the grid finds the same: nothing to do)
    return;
  if (mine != 0)
  {
    // this thread's parts carry the numbers [number, number + mine): those that are g modulo G are this workgroup's
    // (mine < G for every volume that is not huge: at most one)
    const uint32_t r = number % G;
    uint32_t target = number + (g >= r ? g - r : g + G - r);
    uint32_t i = 0, seen = number;  // parts of the words before word i
    while (target < number + mine)
    {
      uint32_t w = 0, cnt = 0;
      for (; i < groups; ++i)
      {
        w = words[i * 256u + threadIdx.x];
        cnt = ((w & 0x07070707u) * 0x01010101u) >> 24;
        if (seen + cnt > target)
          break;
        seen += cnt;
      }
      uint32_t at = seen;  // the part is in word i: which brick, which part of it
      for (uint32_t k = 0; k < 4u; ++k)
      {
        const uint32_t c = (w >> (8u * k)) & 0xffu, u = c & kClassUnits;
        if (at + u > target)
        {
          const uint32_t slot = target / G;
          if (slot < list_cap)
            my_parts[slot] = ((i * 256u + threadIdx.x) * 4u + k) | ((target - at) << 28) | ((c & kClassFront) ? 1u << 30 : 0u);
          break;
        }
        at += u;
      }
      target += G;
    }
  }
  if (threadIdx.x == 0)
    share_n = total / G + (g < total - (total / G) * G ? 1u : 0u);
  __syncthreads();
  const uint32_t n_mine = share_n < list_cap ? share_n : list_cap;  // (list_cap holds the worst case: see the launcher)

  // The four waves take the workgroup's parts off a counter in LDS as they come free.  (One counter in memory for
  // the whole grid cost 100 us per launch: atomics on one word take about 6 ns each on this part, whoever issues
  // them.  Fixed shares per wave: two or three parts each, and the launch lasts as long as three.)
  const bool first_lane = (threadIdx.x & 63u) == 0;
  for (;;)
  {
    uint32_t k = 0;
    if (first_lane)
      k = atomicAdd(&share_next, 1u);
    k = __builtin_amdgcn_readfirstlane(k);
    if (k >= n_mine)
      break;
    asm volatile("" : "+s"(pq));
    IT_BEGIN;
    const uint32_t item = __builtin_amdgcn_readfirstlane(my_parts[k]);
    const uint32_t brick = item & 0x0fffffffu, part = (item >> 28) & 3u;
    const bool is_front = (item >> 30) & 1u;
    const uint32_t wave = brick & 3u, tile = brick >> 2, cj = tile / p.grid_x, block_x = tile - cj * p.grid_x;
    const int cs = p.z0 + __builtin_amdgcn_readfirstlane((int)chunks.first[cj]);
    const int ce = cs + __builtin_amdgcn_readfirstlane((int)chunks.layers[cj]);
    const int pl = part_layers(ce - cs);
    const int zs = cs + (int)part * pl;  // this part's layers of the brick
    const int ze = zs + pl < ce ? zs + pl : ce;
#ifdef YAK_INTEGRATE_TIMING
    const size_t it_slot = (size_t)brick * 4 + part;
#endif
    mask_t dead = 0, front = 0;
    int shift = 31;
    int ws = zs, we = ze;
    bool stream = is_front;
    if (refine && !is_front)
    {
      // the part once more, layer by layer against the fine cells
      shift = 0;
      while ((8 << shift) < ze - zs)
        ++shift;
      classify_pieces<VX>(p, block_x, wave, zs, ze, shift, dead, front);
      if (dead == ~0ull)
      {
        IT_CLOCK(it_t1);
        IT_ROW(it_slot, kBrickDead, it_t1, 0, 0);
        continue;
      }
      // cut the dead pieces off both ends (the replay then stops at the first layer that matters)
      const int first = __builtin_ctzll(~dead) >> 3, last = (63 - __builtin_clzll(~dead)) >> 3;
      ws = zs + (first << shift);
      we = zs + ((last + 1) << shift);
      if (we > ze)
        we = ze;
      // front from the first live piece to the last: the part is streamed
      const mask_t upto = last == 7 ? ~0ull : (1ull << ((last + 1) << 3)) - 1ull;
      const mask_t range = upto & ~((1ull << (first << 3)) - 1ull);
      stream = (front & range) == range;
    }
    IT_CLOCK(it_t1);
    if (stream)
    {
      uint32_t elem;
      int x0, y;
      if (lane_voxels<VX>(p, block_x, wave, elem, x0, y))
      {
        const uint32_t lb = p.plane * 4u;
        const rsrc_t rsf = __builtin_amdgcn_make_buffer_rsrc(p.vol + (size_t)(ws - p.z0) * p.plane, 0,
                                                             (uint32_t)(we - ws) * lb, 0x00020000);
        walk_front_deep<VX, NT, 4>(p, rsf, elem * 4u, lb, we - ws, ballot(true), ((uint32_t)p.max_weight << 16) | 0xffffu);
      }
      IT_ROW(it_slot, kBrickFront, it_t1, 0, we - ws);
      continue;
    }
#ifdef YAK_INTEGRATE_TIMING
    walk_mixed<VX, NT, true>(p, coarse_lds, block_x, wave, zs, ws, we, dead, front, shift, it_t3, it_slot);
#else
    walk_mixed<VX, NT, true>(p, coarse_lds, block_x, wave, zs, ws, we, dead, front, shift);
#endif
    IT_ROW(it_slot, kBrickMixed, it_t1, it_t3, we - ws);
  }
#undef p
}
#ifdef YAK_INTEGRATE_TIMING
extern "C" int yakhip_debug_integrate_phases(unsigned long long* out, int max_rows)
{
  void* at = nullptr;
  const int n = max_rows < kItSlots ? max_rows : kItSlots;
  if (hipDeviceSynchronize() != hipSuccess || hipGetSymbolAddress(&at, HIP_SYMBOL(g_it_phase)) != hipSuccess ||
      hipMemcpy(out, at, (size_t)n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemset(at, 0, (size_t)n * 8 * sizeof(unsigned long long)) != hipSuccess)
    return -1;
  return n;
}
extern "C" int yakhip_debug_integrate_rows(unsigned long long* out, int max_rows)
{
  void* at = nullptr;
  const int n = max_rows < kItSlots ? max_rows : kItSlots;
  if (hipDeviceSynchronize() != hipSuccess || hipGetSymbolAddress(&at, HIP_SYMBOL(g_it_time)) != hipSuccess ||
      hipMemcpy(out, at, (size_t)n * 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemset(at, 0, (size_t)n * 6 * sizeof(unsigned long long)) != hipSuccess)
    return -1;
  return n;
}
#endif

template <int VX>
static int launch_variant(yakhip_ctx* ctx, const IntegrateParams& p, const ChunkTable& chunks, dim3 grid, bool use_static)
{
  const size_t coarse_lds = VX == 4 && p.grp_enable ? (size_t)p.cw * p.ch * sizeof(float) : 0;
  std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
  int rc = use_static ? 0 : timing_begin(ctx, &ev);  // measurement aid: one event pair around this launch's kernels
  if (rc != 0)                                       // (it may create events: not on the allocation-free path)
    return rc;
  if (!use_static)
    (p.two_kernels ? ctx->launches_two : ctx->launches_one) += 1;
  if constexpr (VX == 4)
    if (p.two_kernels)
    {
      const uint32_t rows_x = (grid.x * 4u + 31u) / 32u;
      const uint32_t cp_rows = p.cp_ffwd ? 0u : grid.x;
      hipLaunchKernelGGL((classify_kernel<VX>), dim3(cp_rows + rows_x * grid.y), dim3(256), 0, ctx->stream, p, chunks, rows_x, cp_rows);
      if (p.cp_ffwd)
        hipLaunchKernelGGL((checkpoint_tiles_kernel<VX>), dim3(grid.x), dim3(256), 0, ctx->stream, p, grid.y);
      const int per_simd = ctx->mixed_waves > 0 ? ctx->mixed_waves : kMixedWavesPerSimd;
      const uint32_t G = (uint32_t)(ctx->num_cus * per_simd);
      const uint32_t list_cap = (p.n_bricks * 4u + G - 1u) / G + 1u;  // (every brick live, four parts each)
      const size_t lds2 = coarse_lds + (size_t)list_cap * sizeof(uint32_t);
      if (ctx->nontemporal)
        hipLaunchKernelGGL((integrate_list_kernel<VX, true>), dim3(G), dim3(256), lds2, ctx->stream, p, chunks, ctx->refine, list_cap);
      else
        hipLaunchKernelGGL((integrate_list_kernel<VX, false>), dim3(G), dim3(256), lds2, ctx->stream, p, chunks, ctx->refine, list_cap);
      if ((rc = timing_end(ctx, ev)) != 0)
        return rc;
      return check(hipGetLastError(), "integrate_list_kernel launch");
    }
  if (ctx->nontemporal)
    hipLaunchKernelGGL((integrate_kernel<VX, true>), grid, dim3(256), coarse_lds, ctx->stream, p, chunks);
  else
    hipLaunchKernelGGL((integrate_kernel<VX, false>), grid, dim3(256), coarse_lds, ctx->stream, p, chunks);
  if ((rc = timing_end(ctx, ev)) != 0)
    return rc;
  return check(hipGetLastError(), "integrate_kernel launch");
}

// ---- scratch of the allocation-free entry point (yakhip_integrate_static) -----------------------
// One array per GPU, held until the process ends: [band thresholds | coarse minima] of one frame, then the
// column checkpoints of a deep slab.  Host code never uses the array's name as an address (it is a host-side
// stand-in): static_scratch_address() asks the runtime for the device address once per device and the
// launchers pass that pointer on, exactly as they pass the context-owned buffers of yakhip_integrate.
// Nothing in it has to start at zero: the prepass of a call writes every word the call's kernel reads.
constexpr size_t kStaticMaxPixels = YAKHIP_STATIC_MAX_PIXELS;    // image of the frame
constexpr size_t kStaticMaxColumns = YAKHIP_STATIC_MAX_COLUMNS;  // X * Y up to which a deep slab gets checkpoints
constexpr size_t kStaticCoarseCap = 32768;
constexpr size_t kStaticCellCap = 1u << 20;  // both brick-test planes (8 B per 16 x 16 pixel cell)
constexpr size_t kStaticFrameBytes = kStaticMaxPixels * sizeof(f32x2) + 256u + kStaticCoarseCap + kStaticCellCap;
constexpr size_t kStaticCkptBytes = kStaticMaxColumns * 3u * sizeof(float);
static __device__ __align__(256) unsigned char scratch[kStaticFrameBytes + kStaticCkptBytes];

int static_scratch_address(void** out)
{
  void* ptr = nullptr;
  YAKHIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));  // of the CURRENT device's copy
  if (!ptr)
    return fail(YAKHIP_EINVAL, "hipGetSymbolAddress(scratch) returned a null pointer");
  *out = ptr;
  return 0;
}

// scratch for the column checkpoints of slot `slot` (see checkpoint_column); sets p.ckpt or leaves it null
static int prepare_checkpoint(yakhip_ctx* ctx, IntegrateParams& p, int slot, int slots, bool use_static)
{
  p.ckpt = nullptr;
  if (p.z0 < kCheckpointFrom || !ctx->checkpoint)
    return 0;
  if (use_static)
  {
    // a plane beyond the array's share replays from layer 0 instead: same voxels, slower start
    if ((size_t)p.plane <= kStaticMaxColumns)
      p.ckpt = (const float*)((char*)ctx->static_scratch + kStaticFrameBytes);
    return 0;
  }
  const size_t slot_bytes = ((size_t)p.plane * 3u * sizeof(float) + 255u) & ~(size_t)255u;
  if (ctx->ckpt_bytes < slot_bytes * (size_t)slots)
  {
    if (ctx->ckpt)
    {
      YAKHIP_TRY(hipStreamSynchronize(ctx->stream));
      YAKHIP_TRY(hipFree(ctx->ckpt));
      ctx->ckpt = nullptr;
      ctx->ckpt_bytes = 0;
    }
    YAKHIP_TRY(hipMalloc(&ctx->ckpt, slot_bytes * (size_t)slots));
    ctx->ckpt_bytes = slot_bytes * (size_t)slots;
  }
  p.ckpt = (const float*)((char*)ctx->ckpt + slot_bytes * (size_t)slot);
  return 0;
}

// Per-frame set-up shared by the single-frame and the batched launch: the prepass kernel (band
// thresholds + coarse minima) into slot `slot` of the context's scratch, and the group-test constants.
// All slots of one launch have the same geometry (same image size, same coarse cap).
// ahead = true (launch_integrate_prepare): only the prepass, into the slot prepare calls take in turn, noted in
// ctx->prepared; a later single-frame call for the same image finds it there and launches no prepass of its own.
// what the findings belong to: the slab's extent (they carry over between the launches of a volume whatever their chunking)
static uint32_t volume_signature(const IntegrateParams& p)
{
  uint32_t h = 2166136261u;
  for (uint32_t v : { (uint32_t)p.X, (uint32_t)p.Y, (uint32_t)p.z0, (uint32_t)p.z1 })
    h = (h ^ v) * 16777619u;
  return h | 1u;
}

// The brick test's findings of a recent launch of this volume, if there are any: shares of front and of mixed wave bricks.
static bool recent_findings(const yakhip_ctx* ctx, const IntegrateParams& p, double* front, double* mixed)
{
  if (!ctx->stats_host)
    return false;
  const volatile uint32_t* st = ctx->stats_host;
  const uint32_t f = st[0], m = st[1], n = st[2], seq = st[3];
  if (n == 0 || ctx->launch_seq - seq > 4u || ctx->seq_signature[seq & 7u] != volume_signature(p))
    return false;
  *front = (double)f / n;
  *mixed = (double)m / n;
  return true;
}

// widest vector the layout allows: rows must not straddle a lane's group and the slab base,
// every row start and every layer start must be aligned to the vector size
static int widest_vector(const yakhip_ctx* ctx, const IntegrateParams& p)
{
  const uintptr_t base = (uintptr_t)p.vol;
  int vx = 1;
  if (p.X % 4 == 0 && base % 16 == 0)
    vx = 4;
  else if (p.X % 2 == 0 && base % 8 == 0)
    vx = 2;
  if (ctx->force_vx > 0 && ctx->force_vx < vx)
    vx = ctx->force_vx;
  return vx;
}

// P = largest |camera coordinate| in the slab, f = larger focal length: what the culling tests' margins are made of
static void pose_extent(const IntegrateParams& p, float* P_out, float* f_out)
{
  float P = 0.f;
  for (int i = 0; i < 3; ++i)
  {
    const float ext = fabsf(p.R[3 * i + 0]) * p.vsx * (float)p.X + fabsf(p.R[3 * i + 1]) * p.vsy * (float)p.Y +
                      fabsf(p.R[3 * i + 2]) * p.vsz * (float)p.z1 + fabsf(p.t[i]);
    P = fmaxf(P, ext);
  }
  *P_out = P;
  *f_out = fmaxf(fabsf(p.fx), fabsf(p.fy));
}

// the brick test assumes an ordinary pinhole (positive focal lengths, image below 32k px) and a finite pose
static bool brick_test_allowed(const yakhip_ctx* ctx, const IntegrateParams& p)
{
  float P, f;
  pose_extent(p, &P, &f);
  return ctx->brick_test && p.fx > 0.f && p.fy > 0.f && p.cols < 32768 && p.rows < 32768 && std::isfinite(P) && std::isfinite(f) &&
         std::isfinite(p.cx) && std::isfinite(p.cy);
}

// Which launch (decided before anything is enqueued, so that the prepass does what that launch needs and no more):
// the two kernels pay where mixed bricks are many and front bricks few (an orbit frame at 512^3: 0.109 -> 0.093 ms); a
// frame that is mostly free space streams best from the one-kernel grid (the wall scene: 0.182 against 0.25 ms).  The
// launcher goes by what the brick test found in an earlier frame of this volume -- the previous launch's class
// bytes, summed up by this launch's prepass into pinned memory and read here without waiting, so it is two frames
// old or older -- and starts with one kernel.  Either way the voxels are the same.  A caller that enqueues frame after
// frame without ever waiting gets one kernel, as before: the findings must be those of one of the last four launches
// to count (a frame loop that waits once per frame, as tracking does, sees the launch before the previous one).
// pose_known = false (yakhip_integrate_prepare): whether the frame being prepared is likely to go that way.
static bool want_two_kernels(const yakhip_ctx* ctx, const IntegrateParams& p, int vx, bool use_static, bool pose_known = true)
{
  // (the entry point that allocates nothing keeps to one kernel: the checkpoints alone are 50 MB for a 512^2 plane)
  if (use_static || !ctx->defer_mixed || vx != 4 || (pose_known && !brick_test_allowed(ctx, p)) || (!pose_known && !ctx->brick_test))
    return false;
  const uint64_t voxels = (uint64_t)p.plane * (uint64_t)(p.z1 - p.z0);
  if (voxels / 256u >= (1u << 26))  // (wave bricks are numbered in 28 bits; chunks are at least 8 layers)
    return false;
  if (ctx->defer_mixed == 2)
    return true;
  if (ctx->batch_as_singles)  // (a batch that runs as single-frame launches has decided for all its frames: below)
    return voxels >= kTwoKernelMinVoxels;
  double front = 0, mixed = 0;
  return voxels >= kTwoKernelMinVoxels && recent_findings(ctx, p, &front, &mixed) && mixed >= 0.03 && front <= 2.0 * mixed;
}

// Once yakhip_integrate_prepare is in use (a prepass may then run on another stream than the walks), every launch that
// reads the context's threshold scratch leaves an event behind for the half (or halves) of it that it read, and a
// prepass ahead of time waits for the event of the half it is about to write.  In the matched order -- prepare(f + 1)
// beside integrate(f), each in its own half -- that event was recorded behind integrate(f - 1) and is long past;
// any other order of calls (another image in between, a batch, a skipped prepare) is now merely slower, where it
// used to let the prepass rewrite thresholds under a running walk.
static int note_thr_read(yakhip_ctx* ctx, int slot /* -1: all of it */)
{
  if (!ctx->prepare_used)
    return 0;
  for (int s = 0; s < 2; ++s)
    if (slot < 0 || slot == s)
    {
      YAKHIP_TRY(hipEventRecord(ctx->thr_read[s], ctx->stream));
      ctx->thr_read_valid[s] = 1;
    }
  return 0;
}

// two: the launch that follows is the two-kernel one -- the prepass then fills the fine cells as well and leaves the
// closed-form column checkpoints of a deep slab out (checkpoint_tiles_kernel makes the ones it needs); ahead of time
// (no launch follows yet) the same for a frame that is likely to go that way.
static int prepare_frame(yakhip_ctx* ctx, IntegrateParams& p, int slot, int slots, size_t coarse_cap_bytes,
                         bool use_static = false, bool ahead = false, bool two = false)
{
  bool reuse = false;
  if (ahead)
  {
    slots = 2, slot = ctx->prepared_next;
    ctx->prepared.valid = false;
    if (!ctx->prepare_used)
    {
      // first use: the events, and one wait for everything the walks' stream holds so far (launches before this
      // one left no event)
      for (int s = 0; s < 2; ++s)
        YAKHIP_TRY(hipEventCreateWithFlags(&ctx->thr_read[s], hipEventDisableTiming));
      ctx->prepare_used = 1;
      hipStream_t walks = ctx->main_stream_saved ? ctx->main_stream : ctx->stream;
      if (walks != ctx->stream)
      {
        YAKHIP_TRY(hipEventRecord(ctx->thr_read[0], walks));
        YAKHIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->thr_read[0], 0));
      }
    }
    else if (ctx->thr_read_valid[slot])
      YAKHIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->thr_read[slot], 0));
  }
  else if (ctx->prepared.valid)
  {
    // (any other use of the scratch overwrites what was prepared)
    reuse = !use_static && slots == 1 && ctx->prepared.dists == (const void*)p.dists && ctx->prepared.pitch == p.dists_pitch &&
            ctx->prepared.rows == p.rows && ctx->prepared.cols == p.cols && ctx->prepared.trunc_hi == p.trunc_hi &&
            (two ? ctx->prepared.fine : (p.z0 < kCheckpointFrom || !ctx->checkpoint));
    ctx->prepared.valid = false;
    if (reuse)
      slots = 2, slot = ctx->prepared.slot;
  }
  if (!ahead && !use_static)
    ctx->thr_slot_read = reuse ? slot : -1;  // (what the launch that follows reads: note_thr_read)
  // ---- per-frame prepass into context-owned scratch: per-pixel band thresholds (8 B/pixel) and
  //      the coarse minima the group test looks up (one float per bk x bk pixel cell) ----
  int bk = 8;
  while ((size_t)((p.cols + bk - 1) / bk) * ((p.rows + bk - 1) / bk) * sizeof(float) > coarse_cap_bytes && bk < 1024)
    bk *= 2;
  p.cw = (p.cols + bk - 1) / bk;
  p.ch = (p.rows + bk - 1) / bk;
  const size_t thr_bytes = (((size_t)p.rows * (size_t)p.cols + 1u) * sizeof(f32x2) + 15u) & ~(size_t)15u;  // + the entry {-1, -1}
  p.thr_none = (uint32_t)p.rows * (uint32_t)p.cols;
  const size_t coarse_bytes = (size_t)p.cw * p.ch * sizeof(float);
  p.ccw = (p.cols + kBrickCell - 1) / kBrickCell;
  p.cch = (p.rows + kBrickCell - 1) / kBrickCell;
  const size_t cell_bytes = (size_t)p.ccw * p.cch * sizeof(float);  // per plane
  p.fcw = (p.cols + kFineCell - 1) / kFineCell;
  p.fch = (p.rows + kFineCell - 1) / kFineCell;
  const size_t fine_bytes = (size_t)p.fcw * p.fch * sizeof(float);  // per plane
  const size_t slot_bytes = (thr_bytes + coarse_bytes + 2u * cell_bytes + 2u * fine_bytes + 255u) & ~(size_t)255u;
  if (use_static)
  {
    if (!ctx->static_scratch)
      return fail(YAKHIP_EINVAL, "integrate_static: the context holds no address of the static scratch");
    if ((size_t)p.rows * (size_t)p.cols > kStaticMaxPixels || coarse_bytes > kStaticCoarseCap ||
        2u * cell_bytes + 2u * fine_bytes > kStaticCellCap)
      return fail(YAKHIP_ERANGE, "integrate_static: a %d x %d image exceeds the static scratch (at most %zu pixels)",
                  p.cols, p.rows, kStaticMaxPixels);
  }
  else if (ctx->thr_bytes < slot_bytes * (size_t)slots)
  {
    if (reuse)
      return fail(YAKHIP_EINVAL, "integrate: the prepared frame's scratch is gone");  // (cannot happen: it was sized then)
    if (slot != 0 && !ahead)
      return fail(YAKHIP_EINVAL, "integrate: frames of one batch must have the same image size");
    if (ctx->thr)
    {
      YAKHIP_TRY(hipStreamSynchronize(ctx->stream));
      YAKHIP_TRY(hipFree(ctx->thr));
      ctx->thr = nullptr;
      ctx->thr_bytes = 0;
    }
    YAKHIP_TRY(hipMalloc(&ctx->thr, slot_bytes * (size_t)slots));
    ctx->thr_bytes = slot_bytes * (size_t)slots;
  }
  char* base = use_static ? (char*)ctx->static_scratch : (char*)ctx->thr + slot_bytes * (size_t)slot;
  p.thr = (const f32x2*)base;
  p.coarse = (const float*)(base + thr_bytes);
  p.cellmin = (const float*)(base + thr_bytes + coarse_bytes);
  p.cellmax = (const float*)(base + thr_bytes + coarse_bytes + cell_bytes);
  p.finemin = (const float*)(base + thr_bytes + coarse_bytes + 2u * cell_bytes);
  p.finemax = (const float*)(base + thr_bytes + coarse_bytes + 2u * cell_bytes + fine_bytes);
  if (ahead || two)
    p.ckpt = nullptr;
  else
  {
    const int rc = prepare_checkpoint(ctx, p, slot, slots, use_static);
    if (rc != 0)
      return rc;
  }
  if (!reuse)
  {
    const uint32_t thr_blocks_x = (uint32_t)((p.cols + 255) / 256);
    const uint32_t thr_blocks = thr_blocks_x * (uint32_t)p.rows;
    const uint32_t cell_blocks = (uint32_t)((p.cw * p.ch + 3) / 4);
    const uint32_t ckpt_blocks = p.ckpt ? (p.plane + 255u) / 256u : 0u;
    const uint32_t brick_blocks = (uint32_t)((p.ccw * p.cch + 3) / 4);
    const uint32_t fine_blocks = two ? (uint32_t)((p.fcw * p.fch + 255) / 256) : 0u;  // (only the two-kernel launch looks at them)
    // (one more workgroup sums up the class bytes of the launch before this one: see prepass_kernel.  A prepass ahead of
    // time runs beside that launch: it then sums bytes of which some are still those of the launch before -- each byte
    // is a whole class either way, and "what the last two frames looked like" is all the findings are used for.  Without
    // this a loop that prepares every frame never had findings, and every frame of it ran as one kernel.)
    const uint32_t stats_blocks =
        !use_static && (ahead || slot == 0) && ctx->classes && ctx->classes_n != 0 && ctx->stats_host ? 1u : 0u;
    hipLaunchKernelGGL(prepass_kernel, dim3(thr_blocks + cell_blocks + brick_blocks + fine_blocks + stats_blocks + ckpt_blocks),
                       dim3(256), 0, ctx->stream, p.dists, p.dists_pitch, p.rows, p.cols, p.trunc_hi, (f32x2*)base,
                       (float*)(base + thr_bytes), p.cw, p.ch, bk, thr_blocks_x, thr_blocks, ckpt_blocks, p,
                       const_cast<float*>(p.ckpt), const_cast<float*>(p.cellmin), const_cast<float*>(p.cellmax), p.ccw,
                       p.cch, cell_blocks, brick_blocks, fine_blocks, (const uint32_t*)ctx->classes, (uint32_t)ctx->classes_n,
                       ctx->classes_seq, ctx->stats_host);
    YAKHIP_TRY(hipGetLastError());
  }
  if (ahead)
  {
    ctx->prepared.dists = (const void*)p.dists, ctx->prepared.pitch = p.dists_pitch, ctx->prepared.rows = p.rows;
    ctx->prepared.cols = p.cols, ctx->prepared.trunc_hi = p.trunc_hi, ctx->prepared.slot = slot, ctx->prepared.valid = true;
    ctx->prepared.fine = two;
    ctx->prepared_next = slot ^ 1;
    return 0;
  }

  // ---- group-test constants ("Group test") ----
  // margin: 1/64 px covers the approximate projections (v_rcp, unrefined: < 2e-3 px inside a 32k px
  // image) and the deviation of the rounded voxel positions from a straight line, which is at most
  // f * 2^-23 * P / z px (P = largest |camera coordinate| in the volume); requiring
  // z >= f * P * 2^-15 keeps that below a quarter of the margin.
  {
    const float margin = 1.f / 64.f;
    float P, f;
    pose_extent(p, &P, &f);
    p.grp_margin = margin;
    p.grp_cols_hi = (float)p.cols - margin;
    p.grp_rows_hi = (float)p.rows - margin;
    p.grp_span = (float)bk - 2.f * margin;
    p.grp_inv_bk = 1.f / (float)bk;
    p.grp_bias = -margin / (float)bk;
    p.grp_zmin = fmaxf(f * P * 0x1p-15f, 0x1p-10f);
    // the test assumes an ordinary pinhole (positive focal lengths, image below 32k px) and finite pose
    p.grp_enable = ctx->group_test && p.fx > 0.f && p.fy > 0.f && p.cols < 32768 && p.rows < 32768 &&
                   std::isfinite(P) && std::isfinite(f) && coarse_bytes <= coarse_cap_bytes;
    // brick test (classify_brick): position slack and the nearest depth it is trusted at
    p.brick_eps = (float)(p.z1 + 32) * 0x1p-22f * P;
    p.brick_zmin = fmaxf(p.grp_zmin, 8.f * p.brick_eps);
    p.brick_stats = use_static ? nullptr : ctx->brick_stats;
    p.brick_enable = brick_test_allowed(ctx, p);
  }
  return 0;
}

// The memory of the brick test's findings -- a class byte per wave brick, padded with zeros to a multiple of 1024 -- and,
// for a launch in two kernels (p.two_kernels, decided by want_two_kernels), of the column checkpoints.
static int prepare_list(yakhip_ctx* ctx, IntegrateParams& p, uint64_t wave_bricks, uint32_t blocks_x, int vx, bool use_static)
{
  p.classes = nullptr;
  p.n_bricks = 0;
  p.grid_x = blocks_x;
  p.cptab = nullptr;
  p.cp_stride = p.cp_count = p.cp_ffwd = 0;
  if (use_static || !ctx->defer_mixed || !p.brick_enable || vx != 4 || wave_bricks >= (1u << 28))
  {
    if (p.two_kernels)
      return fail(YAKHIP_EINVAL, "integrate: a two-kernel launch was planned for a layout that has none");  // (cannot happen)
    return 0;
  }
  ctx->launch_seq += 1;
  ctx->seq_signature[ctx->launch_seq & 7u] = volume_signature(p);
  const size_t bytes_cls = ((size_t)wave_bricks + 1023u) & ~(size_t)1023u;
  if (ctx->classes_bytes < bytes_cls)
  {
    if (ctx->classes)
    {
      YAKHIP_TRY(hipStreamSynchronize(ctx->stream));
      YAKHIP_TRY(hipFree(ctx->classes));
      ctx->classes = nullptr;
      ctx->classes_bytes = 0;
    }
    YAKHIP_TRY(hipMalloc((void**)&ctx->classes, bytes_cls));
    ctx->classes_bytes = bytes_cls;
    ctx->classes_n = 0;
  }
  if (ctx->classes_n != wave_bricks)  // (the padding behind the last brick must read as dead)
  {
    YAKHIP_TRY(hipMemsetAsync(ctx->classes, 0, ctx->classes_bytes, ctx->stream));
    ctx->classes_n = wave_bricks;
  }
  p.classes = ctx->classes;
  p.n_bricks = (uint32_t)wave_bricks;
  ctx->classes_seq = ctx->launch_seq;  // (the launch these class bytes will be of)
  if (!p.two_kernels)
    return 0;
  // column checkpoints every cp_stride layers, at most 256 MiB of them
  const int layers = p.z1 - p.z0;
  int stride = kCheckpointStride;
  while ((size_t)((layers + stride - 1) / stride) * 3u * p.plane * sizeof(float) > ((size_t)256 << 20) && stride < layers)
    stride *= 2;
  const int count = (layers + stride - 1) / stride;
  const size_t bytes = (size_t)count * 3u * p.plane * sizeof(float);
  if (ctx->cptab_bytes < bytes)
  {
    if (ctx->cptab)
    {
      YAKHIP_TRY(hipStreamSynchronize(ctx->stream));
      YAKHIP_TRY(hipFree(ctx->cptab));
      ctx->cptab = nullptr;
      ctx->cptab_bytes = 0;
    }
    YAKHIP_TRY(hipMalloc((void**)&ctx->cptab, bytes));
    ctx->cptab_bytes = bytes;
  }
  p.cptab = ctx->cptab;
  p.cp_stride = stride;
  p.cp_count = count;
  p.cp_ffwd = p.z0 >= kCheckpointFrom && ctx->checkpoint;  // (as a one-kernel launch of the slab would start: prepare_checkpoint)
  return 0;
}

// A volume that replays from layer 0 and has layers below 4 MiB gets chunks of graded length, deepest first
// (chunk_table below).
static bool graded_chunks(const yakhip_ctx* ctx, const IntegrateParams& p)
{
  return ctx->chunk_ramp >= 0 && !p.ckpt && p.z0 == 0 && (uint64_t)p.plane * 4u < (4ull << 20);
}

// z_chunk and grid for a plane of `blocks_x` workgroups.  Short chunks balance the load: the cost of a
// chunk depends on how much surface it holds (per-voxel path, band arithmetic), and with a few long
// chunks per column the last workgroups of a surface-rich frame run alone (measured at 512^3: 171-layer
// chunks 0.61 ms on the orbit scene, 32-layer chunks 0.29 ms; the wall scene is 0.193 ms either way).
// What limits shortening is the replay of layers [0, zs), which grows with the chunk count: keep at most
// 16 chunks per column, at least 32 layers each, unless the grid would be too small to fill the chip.
static int choose_chunk(const yakhip_ctx* ctx, const IntegrateParams& p, uint32_t blocks_x, uint32_t want_blocks,
                        int min_chunk, int* chunk_out, uint32_t* nchunks_out)
{
  const int layers = p.z1 - p.z0;
  const bool graded = graded_chunks(ctx, p);
  int chunk = ctx->z_chunk;
  if (chunk <= 0)
  {
    // (a volume walked from layer 0 in a graded table, see chunk_table: 24-layer chunks at 512 layers, measured)
    chunk = graded ? (layers / 21) & ~1 : (layers + 15) / 16;
    if (chunk < min_chunk)
      chunk = min_chunk;
    // A chunk that starts at layer zs replays zs steps before its own `chunk` layers.  While the average
    // replay stays below about 8 steps per layer of real work it hides behind the memory traffic (a whole
    // 512-layer volume in 32-layer chunks: 256 / 32); a slab deep inside a tall volume needs longer chunks
    // for that (measured, rank 7 of 8 at 1024^3, layers [896, 1024): 0.264 ms with 32-layer chunks).
    const int for_replay = ((p.ckpt || p.two_kernels ? 0 : p.z0) + layers / 2) / (graded ? 12 : 8);  // (with a checkpoint the replay starts at z0)
    if (chunk < for_replay)
    {
      const int n = layers / for_replay > 1 ? layers / for_replay : 1;
      chunk = (layers + n - 1) / n;
    }
    while (chunk > 16 && chunk > for_replay && (uint64_t)blocks_x * (uint64_t)((layers + chunk - 1) / chunk) < want_blocks)
      chunk = chunk / 2 < 16 ? 16 : chunk / 2;
  }
  if (chunk > layers)
    chunk = layers;
  // one buffer descriptor spans a chunk: keep it below 2 GiB (see kNoAccess)
  const uint64_t layer_bytes = (uint64_t)p.plane * 4u;
  const uint64_t max_chunk = 0x7fffffffull / layer_bytes;
  if (max_chunk < 1)
    return fail(YAKHIP_ERANGE, "integrate: one z layer exceeds 2 GiB");
  if ((uint64_t)chunk > max_chunk)
    chunk = (int)max_chunk;
  const uint32_t nchunks = (uint32_t)((layers + chunk - 1) / chunk);
  if (nchunks > 65535u)
    return fail(YAKHIP_ERANGE, "integrate: %u z-chunks exceed the grid limit; raise z_chunk", nchunks);
  *chunk_out = chunk;
  *nchunks_out = nchunks;
  return 0;
}

// The chunks of a launch, in dispatch order (grid.y index -> layers).  Returns their number (0: they do not fit).
//
// Two things decide how long a frame's last workgroups run on alone, and with it 5-10 % of the kernel time:
//   order   a workgroup costs its replay (zs steps) plus its walk, so the deepest chunks are the longest jobs:
//           they go first (longest job first), the shallow ones fill the gaps at the end;
//   size    where the replay is short a chunk can be short too without the replay taking over: a chunk that
//           starts at layer zs is about zs / ramp layers long, between min_layers and `chunk`.  The short,
//           cheap chunks are the ones dispatched last, so the tail is made of small jobs.
// Measured at 512^3 (kernel time, 32-layer chunks shallow / deep in turn -> this table): see DESIGN.md 3.1.
// Only for a volume that replays from layer 0 and has layers below 4 MiB: a slab that starts from a checkpoint
// has short replays everywhere and streams better in order (rank 7 of 8 at 1024^3: 0.201 ms in order, 0.216
// alternating), and so do the 4 MiB layers of wide slabs (rank 0: 0.196 / 0.206).
static uint32_t chunk_table(const yakhip_ctx* ctx, const IntegrateParams& p, int chunk, int min_layers, ChunkTable* out)
{
  const int layers = p.z1 - p.z0;
  const bool from_zero = !p.ckpt && p.z0 == 0 && (uint64_t)p.plane * 4u < (4ull << 20);
  const int order = ctx->chunk_order >= 0 ? ctx->chunk_order : (from_zero ? 3 : 0);
  const int ramp = ctx->chunk_ramp > 0 ? ctx->chunk_ramp : 6;
  if (ctx->chunk_min > 0)
    min_layers = ctx->chunk_min;
  bool graded = graded_chunks(ctx, p) && min_layers < chunk;
  uint32_t first[kMaxChunks];
  uint16_t count[kMaxChunks];
  uint32_t n;
  for (;;)
  {
    n = 0;
    int z = 0;
    while (z < layers && n < (uint32_t)kMaxChunks)
    {
      int len = chunk;
      if (graded)
      {
        len = (z / ramp + 1) & ~1;  // even: the walk takes layers in pairs
        len = len < min_layers ? min_layers : (len > chunk ? chunk : len);
      }
      if (len > layers - z)
        len = layers - z;
      first[n] = (uint32_t)z;
      count[n] = (uint16_t)len;
      ++n;
      z += len;
    }
    if (z >= layers)
      break;
    // too many chunks for the table: plain chunks, as long as it takes (choose_chunk keeps them below 65536 layers)
    graded = false;
    chunk = (layers + kMaxChunks - 1) / kMaxChunks > chunk ? (layers + kMaxChunks - 1) / kMaxChunks : chunk + 1;
  }
  if ((uint64_t)chunk * p.plane * 4u > 0x7fffffffull)
    return 0;  // (one buffer descriptor spans a chunk, see choose_chunk; no volume that fits the card gets here)
  for (uint32_t j = 0; j < n; ++j)
  {
    uint32_t i = j;  // 0: ascending
    if (order == 1)
      i = (j & 1u) ? n - 1u - (j >> 1) : j >> 1;  // shallow and deep in turn
    else if (order == 3)
      i = n - 1u - j;  // deepest first
    out->first[j] = first[i];
    out->layers[j] = count[i];
  }
  return n;
}

// Largest compact tile (see lane_voxels) that divides the layer exactly; 6 = plain layer order.
static int choose_tile(const yakhip_ctx* ctx, const IntegrateParams& p, int vx)
{
  // with the brick test a compact footprint (32 x 32 voxels for four voxels per lane) makes bricks that are
  // small in every direction: their pixel boxes are small and most of them are settled whole
  int shift = ctx->tile_shift > 0 ? ctx->tile_shift : (p.brick_enable ? 4 : kDefaultTileShift);
  for (; shift < 6; ++shift)
  {
    const uint32_t tile_x = (p.tile_side ? 4u : 1u) * (1u << shift) * (uint32_t)vx;
    const uint32_t tile_y = (p.tile_side ? 1u : 4u) * (64u >> shift);
    if ((uint32_t)p.X % tile_x == 0 && (uint32_t)p.Y % tile_y == 0)
      break;
  }
  return shift;
}

int launch_integrate(yakhip_ctx* ctx, const IntegrateParams& p_in, bool use_static)
{
  IntegrateParams p = p_in;
  const int vx = widest_vector(ctx, p);
  p.two_kernels = want_two_kernels(ctx, p, vx, use_static);
  int rc = prepare_frame(ctx, p, 0, 1, 32768, use_static, false, p.two_kernels != 0);
  if (rc != 0)
    return rc;

  const uint32_t groups = p.plane / (uint32_t)vx;
  const uint32_t blocks_x = (groups + 255u) / 256u;
  p.tile_side = ctx->tile_side;
  p.tile_shift = choose_tile(ctx, p, vx);
  int chunk = 0;
  uint32_t nchunks = 0;
  // Layers of 4 MiB and more (the slabs of a 1024-wide volume) stream better in shorter walks: with no
  // arithmetic at all an in-place sweep of 1024 x 1024 x 128 takes 201 us in 32-layer walks and 188 us in
  // 16-layer walks (tools/rmw.hip; 512 x 512 x 512: 183 us); this kernel 0.203 -> 0.198 ms.
  // Not beyond that: with layers of 16 MiB (a 2048-wide slab) every walk is 4096 workgroups, each with its own
  // set-up (19 KB of coarse minima into LDS, checkpoint, descriptor), and on frames with surfaces -- where the
  // kernel is bound by instructions, not by the sweep -- 16-layer walks cost more than they stream better:
  // four single-frame calls on the orbit rig over a 2048 x 2048 x 256 slab take 7.9 ms in 16-layer walks, 6.7 in
  // 32-layer and 6.1 in 64-layer ones (tools/bench_c5.py --z-chunk), the bare sweep 209 / 231 us per 2^27 voxels.
  const uint64_t layer_bytes = (uint64_t)p.plane * 4u;
  const int min_chunk = layer_bytes >= (4ull << 20) && layer_bytes < (16ull << 20) ? 16 : (graded_chunks(ctx, p) ? 24 : 32);
  if ((rc = choose_chunk(ctx, p, blocks_x, 768u, min_chunk, &chunk, &nchunks)) != 0)
    return rc;
  p.z_chunk = chunk;
  ChunkTable chunks;
  const dim3 grid(blocks_x, chunk_table(ctx, p, chunk, 8, &chunks));
  if (grid.y == 0)
    return fail(YAKHIP_ERANGE, "integrate: the volume needs more than %d z-chunks of at most 2 GiB", kMaxChunks);
  if ((rc = prepare_list(ctx, p, (uint64_t)grid.x * grid.y * 4u, grid.x, vx, use_static)) != 0)
    return rc;
  switch (vx)
  {
    case 4:
      rc = launch_variant<4>(ctx, p, chunks, grid, use_static);
      break;
    case 2:
      rc = launch_variant<2>(ctx, p, chunks, grid, use_static);
      break;
    default:
      rc = launch_variant<1>(ctx, p, chunks, grid, use_static);
      break;
  }
  if (rc == 0 && !use_static)
    rc = note_thr_read(ctx, ctx->thr_slot_read);
  return rc;
}

int launch_integrate_prepare(yakhip_ctx* ctx, const IntegrateParams& p_in)
{
  IntegrateParams p = p_in;
  const bool two = want_two_kernels(ctx, p, widest_vector(ctx, p), false, false);
  if (!two && p.z0 >= kCheckpointFrom && ctx->checkpoint)
    return 0;  // a deep slab's one-kernel prepass also computes its column checkpoints from the pose: nothing to do ahead of time
  return prepare_frame(ctx, p, 0, 1, 32768, false, true, two);
}

// ---- multi-camera batched integrate ---------------------------------------------------------------
// K frames (each with its own ray-distance image, pose and intrinsics) are folded into every voxel in
// frame order k = 0..K-1 during ONE sweep of the volume: the voxel is read once and written once
// (8 B per touched voxel instead of 8 B per frame), and the result is bit-identical to K consecutive
// yakhip_integrate calls because each frame's measurement depends only on the voxel's position and
// that frame's image, and the running average is applied in the same order.  (Extension of
// tsdf_volume.cu:51-110)
//
// Same lane layout, exact replay, group test, band thresholds and buffer addressing as the
// single-frame kernel; per layer the K frames are settled one after the other while the voxels of
// the next layer are already in flight.
constexpr int kBatchMax = 4;
struct BatchParams
{
  IntegrateParams f[kBatchMax];  // f[k].vol / X / Y / z0 / z1 / z_chunk / plane / voxel sizes / trunc agree
};

template <int K, bool NT>
__global__ __launch_bounds__(256, 2) void integrate_batch_kernel(const BatchParams bp, const ChunkTable chunks)
{
  constexpr int VX = 4;
  typedef Vec<VX>::type vec_t;
  extern __shared__ __attribute__((aligned(16))) float coarse_lds[];
  const IntegrateParams& p = bp.f[0];

  const int cells = p.cw * p.ch;  // the same for every frame: same image size, same cell size
  const int zs = p.z0 + __builtin_amdgcn_readfirstlane((int)chunks.first[blockIdx.y]);  // as the single-frame kernel
  const int ze = zs + __builtin_amdgcn_readfirstlane((int)chunks.layers[blockIdx.y]);

  // the wave's brick against every frame (classify_brick): a frame that cannot touch it is skipped for the
  // whole chunk, one that updates all of it with tsdf 1 needs no projection; only mixed frames are walked
  // voxel by voxel (and only they need positions, the replay and the coarse minima in LDS)
  __shared__ int wave_mixed[4];
  int cls[K];
  bool any_mixed = false, all_dead = true;
#pragma unroll
  for (int k = 0; k < K; ++k)
  {
    cls[k] = p.brick_enable ? classify_brick<VX, 64>(bp.f[k], blockIdx.x, zs, ze) : kBrickMixed;
    any_mixed = any_mixed || cls[k] == kBrickMixed;
    all_dead = all_dead && cls[k] == kBrickDead;
  }
  bool fill = p.grp_enable;
  if (p.brick_enable)
  {
    if (p.classes && (threadIdx.x & 63u) == 0)  // what the first frame's test found: the launcher's findings (prepare_list)
      p.classes[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4u + (threadIdx.x >> 6)] =
          (uint8_t)(cls[0] == kBrickDead ? 0u : (cls[0] == kBrickFront ? 0x81u : 1u));
    if ((threadIdx.x & 63u) == 0)
      wave_mixed[threadIdx.x >> 6] = any_mixed;
    __syncthreads();
    fill = fill && (wave_mixed[0] || wave_mixed[1] || wave_mixed[2] || wave_mixed[3]);
  }
  if (fill)
  {
#pragma unroll
    for (int k = 0; k < K; ++k)
      for (int i = (int)threadIdx.x; i < cells; i += 256)
        coarse_lds[k * cells + i] = bp.f[k].coarse[i];
    __syncthreads();
  }
  if (all_dead)
    return;

  uint32_t elem;  // x + y*X of the lane's first voxel; X % VX == 0 so no row straddle
  int x0, y;
  if (!lane_voxels<VX>(p, blockIdx.x, threadIdx.x >> 6, elem, x0, y))
    return;
  const mask_t active = ballot(true);

  Pos<VX> pos[K];
  const float vxy = (float)y * p.vsy;
#pragma unroll
  for (int k = 0; k < K; ++k)
  {
    if (cls[k] != kBrickMixed)  // wave-uniform
      continue;
    const IntegrateParams& f = bp.f[k];
#pragma unroll
    for (int j = 0; j < Pos<VX>::NP; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h)
      {
        const float vxx = (float)(x0 + Pos<VX>::voxel(j, h)) * p.vsx;
        pos[k].x[j][h] = dot3(f.R[0], f.R[1], f.R[2], vxx, vxy, 0.f) + f.t[0];
        pos[k].y[j][h] = dot3(f.R[3], f.R[4], f.R[5], vxx, vxy, 0.f) + f.t[1];
        pos[k].z[j][h] = dot3(f.R[6], f.R[7], f.R[8], vxx, vxy, 0.f) + f.t[2];
      }
    // replay of the reference's per-layer additions on [0, zs), per frame (tsdf_volume.cu:69, 76)
    const float zx = f.R[2] * p.vsz, zy = f.R[5] * p.vsz, zz = f.R[8] * p.vsz;
    const f32x2 sx = { zx, zx }, sy = { zy, zy }, sz = { zz, zz };
    int replay = zs;
    if (f.ckpt)  // (all frames of a batch have one, or none: it depends on z0 only)
    {
#pragma unroll
      for (int j = 0; j < Pos<VX>::NP; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h)
        {
          const uint32_t col = elem + (uint32_t)Pos<VX>::voxel(j, h);
          pos[k].x[j][h] = f.ckpt[col];
          pos[k].y[j][h] = f.ckpt[p.plane + col];
          pos[k].z[j][h] = f.ckpt[2u * p.plane + col];
        }
      replay = zs - p.z0;
    }
    for (int i = 0; i < replay; ++i)
      advance<VX>(pos[k], pos[k], sx, sy, sz);
  }

  const uint32_t max_w_hi = ((uint32_t)p.max_weight << 16) | 0xffffu;
  const uint32_t layer_bytes = p.plane * 4u;
  const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.vol + (size_t)(zs - p.z0) * p.plane, 0,
                                                      (uint32_t)(ze - zs) * layer_bytes, 0x00020000);
  const uint32_t voff = elem * 4u;

  mask_t band = 0, marked = 0;  // (surface map, see the single-frame kernel)
  mask_t have = ~0ull;  // lanes whose `cur` holds this layer's voxels
  vec_t cur = buf_load<VX, NT>(rs, voff, 0u);
  uint32_t soff = 0;
  for (int z = zs; z < ze; ++z, soff += layer_bytes)
  {
    vec_t out = cur;
    mask_t any = 0;
#pragma unroll
    for (int k = 0; k < K; ++k)
    {
      if (cls[k] == kBrickDead)  // wave-uniform
        continue;
      const IntegrateParams& f = bp.f[k];
      mask_t hit[VX], unit[VX];
      float tsdf[VX];
      if (cls[k] == kBrickFront)
      {
#pragma unroll
        for (int v = 0; v < VX; ++v)
        {
          hit[v] = unit[v] = active;
          tsdf[v] = 1.f;
        }
      }
      else
      {
        const f32x2 fx2 = { f.fx, f.fx }, fy2 = { f.fy, f.fy }, cx2 = { f.cx, f.cx }, cy2 = { f.cy, f.cy };
        Probe<VX> pr;
        stage1<VX>(f, pos[k], pr, coarse_lds + k * cells, active, fx2, fy2, cx2, cy2);
        settle<VX>(f, pos[k], pr, ~0ull, hit, unit, tsdf, band);
        const float zx = f.R[2] * p.vsz, zy = f.R[5] * p.vsz, zz = f.R[8] * p.vsz;
        advance<VX>(pos[k], pos[k], f32x2{ zx, zx }, f32x2{ zy, zy }, f32x2{ zz, zz });
      }
      const mask_t hk = hit[0] | hit[1] | hit[2] | hit[3];
      // only updated voxels are read (tsdf_volume.cu:98-100): fetch what the look-ahead did not bring.
      // A lane that needs it has not been touched by an earlier frame of this layer, so `out` is free.
      const mask_t need = hk & ~have;
      if (need != 0)  // wave-uniform
      {
        const vec_t got = buf_load<VX, NT>(rs, lanes(need) ? voff : kNoAccess, soff);
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
#pragma unroll
        for (int v = 0; v < VX; ++v)
          vset<VX>(out, v, lanes(need) ? vget<VX>(got, v) : vget<VX>(out, v));
        have |= need;
      }
      if (hk != 0)  // wave-uniform
      {
        const vec_t prev = out;
        fuse_run<VX>(prev, out, hit, unit, tsdf, p.max_weight, max_w_hi);
      }
      any |= hk;
    }
    buf_store<VX, NT>(out, rs, lanes(any) ? voff : kNoAccess, soff);
    // look-ahead: a lane that updated this layer very likely updates the next one
    have = z + 1 < ze ? any : 0;
    cur = buf_load<VX, NT>(rs, lanes(have) ? voff : kNoAccess, z + 1 < ze ? soff + layer_bytes : 0u);
    if (p.bricks)
      flush_bricks(p, band, marked, x0, y, z, 1, ze);
  }
  if (p.bricks)
    mark_cells(p, marked, x0, y, zs - p.z0, ze - p.z0);
}

template <int K>
static int launch_batch_variant(yakhip_ctx* ctx, const BatchParams& bp, const ChunkTable& chunks, dim3 grid, size_t lds)
{
  std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
  int rc = timing_begin(ctx, &ev);
  if (rc != 0)
    return rc;
  if (ctx->nontemporal)
    hipLaunchKernelGGL((integrate_batch_kernel<K, true>), grid, dim3(256), lds, ctx->stream, bp, chunks);
  else
    hipLaunchKernelGGL((integrate_batch_kernel<K, false>), grid, dim3(256), lds, ctx->stream, bp, chunks);
  if ((rc = timing_end(ctx, ev)) != 0)
    return rc;
  return check(hipGetLastError(), "integrate_batch_kernel launch");
}

int launch_integrate_batch(yakhip_ctx* ctx, const IntegrateParams* frames, int n_frames)
{
  // the wide layout is the only one the batched kernel has; anything else (and a lone frame) goes
  // through the single-frame kernel, frame by frame -- the same result by definition
  const IntegrateParams& p0 = frames[0];
  const bool wide = p0.X % 4 == 0 && (uintptr_t)p0.vol % 16 == 0 && ctx->force_vx != 1 && ctx->force_vx != 2;
  int done = 0;
  while (done < n_frames)
  {
    const int k = wide ? (n_frames - done < kBatchMax ? n_frames - done : kBatchMax) : 1;
    if (k == 1)
    {
      const int rc = launch_integrate(ctx, frames[done]);
      if (rc != 0)
        return rc;
      done += 1;
      continue;
    }
    // A sweep of K frames pays 8 B per touched voxel once instead of K times -- where the frames are mostly free space.
    // Where they are mostly surface (mixed bricks outnumbering front bricks two to one in a recent launch of this
    // volume: recent_findings) the walk is bound by arithmetic, not by traffic, and K single-frame launches, each in two
    // kernels, are faster (512^3 orbit: 0.094 against 0.146 ms per frame; a 2048 x 2048 x 256 slab of configuration 5,
    // four cameras: 1.49 against 1.77 ms).  The voxels are the same either way, by definition of the sweep.
    {
      double front = 0, mixed = 0;
      if (ctx->defer_mixed == 1 && (uint64_t)p0.plane * (uint64_t)(p0.z1 - p0.z0) >= (1ull << 25) &&
          recent_findings(ctx, p0, &front, &mixed) && mixed >= 0.03 && front <= 0.5 * mixed)
      {
        // (one decision for the K frames: the findings the later ones would look at are those of launches this call
        // has only just enqueued -- not there yet, or too old by then -- and a frame that falls back to one kernel costs
        // 0.109 instead of 0.095 ms)
        ctx->batch_as_singles = 1;
        int rc = 0;
        for (int i = 0; i < k && rc == 0; ++i)
          rc = launch_integrate(ctx, frames[done + i]);
        ctx->batch_as_singles = 0;
        if (rc != 0)
          return rc;
        done += k;
        continue;
      }
    }
    BatchParams bp;
    const size_t coarse_cap = 32768 / (size_t)k;  // all K coarse images share the workgroup's LDS
    for (int i = 0; i < k; ++i)
    {
      bp.f[i] = frames[done + i];
      bp.f[i].two_kernels = 0;  // (the sweep is one kernel)
      if (bp.f[i].rows != p0.rows || bp.f[i].cols != p0.cols)
        return fail(YAKHIP_EINVAL, "integrate_batch: frames must have the same image size");
      const int rc = prepare_frame(ctx, bp.f[i], i, k, coarse_cap);
      if (rc != 0)
        return rc;
    }
    for (int i = k; i < kBatchMax; ++i)
      bp.f[i] = bp.f[0];
    // the group test is per workgroup: on for the batch only if every frame allows it
    bool grp = true, brick = true;
    for (int i = 0; i < k; ++i)
    {
      grp = grp && bp.f[i].grp_enable;
      brick = brick && bp.f[i].brick_enable;
    }
    for (int i = 0; i < kBatchMax; ++i)
    {
      bp.f[i].grp_enable = grp;
      bp.f[i].brick_enable = brick;
    }
    const uint32_t blocks_x = (p0.plane / 4u + 255u) / 256u;
    int chunk = 0;
    uint32_t nchunks = 0;
    // the batched kernel replays K frames per chunk start: a little longer chunks pay (measured, 512^3)
    int rc = choose_chunk(ctx, bp.f[0], blocks_x, 1024u, 48, &chunk, &nchunks);
    if (rc != 0)
      return rc;
    bp.f[0].tile_side = ctx->tile_side;
    const int tile = choose_tile(ctx, bp.f[0], 4);
    for (int i = 0; i < kBatchMax; ++i)
    {
      bp.f[i].z_chunk = chunk;
      bp.f[i].tile_shift = tile;
      bp.f[i].tile_side = ctx->tile_side;
    }
    ChunkTable chunks;
    const dim3 grid(blocks_x, chunk_table(ctx, bp.f[0], chunk, 16, &chunks));
    if (grid.y == 0)
      return fail(YAKHIP_ERANGE, "integrate: the volume needs more than %d z-chunks of at most 2 GiB", kMaxChunks);
    // (the sweep leaves the first frame's brick classes behind, as every launch does: the findings above)
    bp.f[0].two_kernels = 0;
    if ((rc = prepare_list(ctx, bp.f[0], (uint64_t)grid.x * grid.y * 4u, grid.x, 4, false)) != 0)
      return rc;
    const size_t lds = grp ? (size_t)k * bp.f[0].cw * bp.f[0].ch * sizeof(float) : 0;
    switch (k)
    {
      case 2:
        rc = launch_batch_variant<2>(ctx, bp, chunks, grid, lds);
        break;
      case 3:
        rc = launch_batch_variant<3>(ctx, bp, chunks, grid, lds);
        break;
      default:
        rc = launch_batch_variant<4>(ctx, bp, chunks, grid, lds);
        break;
    }
    if (rc != 0)
      return rc;
    if ((rc = note_thr_read(ctx, -1)) != 0)
      return rc;
    done += k;
  }
  return 0;
}

// ---- self-test of the division building blocks (yakhip_selftest_division) -----------------------
__global__ __launch_bounds__(256) void selftest_rcp_kernel(unsigned long long* bad)
{
  // every fp32 mantissa at several exponents: refined_rcp(d) must be the correctly rounded 1/d
  const uint32_t m = blockIdx.x * 256u + threadIdx.x;  // 0 .. 2^23-1
  const uint32_t exps[8] = { 107u, 113u, 120u, 126u, 127u, 128u, 137u, 147u };  // 2^-20 .. 2^20
  unsigned long long n = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k)
  {
    const float d = __uint_as_float((exps[k] << 23) | m);
    n += __float_as_uint(refined_rcp(d)) != __float_as_uint(1.0f / d);
  }
  if (n)
    atomicAdd(bad, n);
}

__global__ __launch_bounds__(256) void selftest_div_kernel(unsigned long long* bad, int iters, uint64_t seed)
{
  // pseudo-random numerators in +-[2^-30, 2^16] over denominators in [2^-20, 2^20] and over the
  // integers 1..32768 (the running-average divisor): div_by must equal IEEE '/'
  uint64_t s = seed + (uint64_t)(blockIdx.x * 256u + threadIdx.x) * 0x9E3779B97F4A7C15ull;
  unsigned long long n = 0;
  for (int i = 0; i < iters; ++i)
  {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const uint32_t a = (uint32_t)(s >> 32);
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const uint32_t b = (uint32_t)(s >> 32);
    const float num = __uint_as_float((a & 0x807fffffu) | ((97u + (a >> 23) % 47u) << 23));
    const float d1 = __uint_as_float((b & 0x007fffffu) | ((107u + (b >> 23) % 41u) << 23));
    const float d2 = (float)(1u + (b % 32768u));
    n += __float_as_uint(div_by(num, d1, refined_rcp(d1))) != __float_as_uint(num / d1);
    n += __float_as_uint(div_by(num, d2, refined_rcp(d2))) != __float_as_uint(num / d2);
  }
  if (n)
    atomicAdd(bad, n);
}

int launch_selftest_division(yakhip_ctx* ctx, unsigned long long counts[4])
{
  unsigned long long* dev = nullptr;
  YAKHIP_TRY(hipMalloc(&dev, 2 * sizeof(unsigned long long)));
  YAKHIP_TRY(hipMemsetAsync(dev, 0, 2 * sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(selftest_rcp_kernel, dim3((1u << 23) / 256u), dim3(256), 0, ctx->stream, dev);
  const int blocks = 2048, iters = 1024;
  hipLaunchKernelGGL(selftest_div_kernel, dim3(blocks), dim3(256), 0, ctx->stream, dev + 1, iters, 0x5eedull);
  unsigned long long host[2] = { ~0ull, ~0ull };
  int rc = check(hipMemcpyAsync(host, dev, sizeof(host), hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
  if (rc == 0)
    rc = check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
  (void)hipFree(dev);
  counts[0] = 8ull << 23;                                  // reciprocals checked
  counts[1] = host[0];                                     // ... that differ from 1.0f / d
  counts[2] = 2ull * (unsigned long long)blocks * 256ull * (unsigned long long)iters;  // quotients checked
  counts[3] = host[1];                                     // ... that differ from n / d
  return rc;
}

// ---- clear: tsdf_volume.cu:14-40.  pack_tsdf(0.f, 0) is the all-zero word. ----------------------
__global__ __launch_bounds__(256) void clear_kernel_x4(u32x4* data, size_t n4)
{
  const u32x4 zero = { 0u, 0u, 0u, 0u };
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256u)
    __builtin_nontemporal_store(zero, data + i);
}

__global__ __launch_bounds__(256) void clear_kernel_x1(uint32_t* data, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u)
    data[i] = 0u;
}

int launch_clear(const yakhip_ctx* ctx, uint32_t* data, size_t n_voxels)
{
  // head (to 16-byte alignment) and tail with the scalar kernel, body with 16-byte stores
  const uintptr_t addr = (uintptr_t)data;
  size_t head = ((16u - (addr & 15u)) & 15u) / 4u;
  if (head > n_voxels)
    head = n_voxels;
  const size_t body4 = (n_voxels - head) / 4u;
  const size_t tail = n_voxels - head - body4 * 4u;

  if (head)
    hipLaunchKernelGGL(clear_kernel_x1, dim3(1), dim3(256), 0, ctx->stream, data, head);
  if (body4)
  {
    size_t blocks = (body4 + 255u) / 256u;
    if (blocks > 8192u)
      blocks = 8192u;
    hipLaunchKernelGGL(clear_kernel_x4, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (u32x4*)(data + head),
                       body4);
  }
  if (tail)
    hipLaunchKernelGGL(clear_kernel_x1, dim3(1), dim3(256), 0, ctx->stream, data + head + body4 * 4u, tail);
  return check(hipGetLastError(), "clear_kernel launch");
}
}  // namespace yakhip
