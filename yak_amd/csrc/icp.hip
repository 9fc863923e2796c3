// icp.hip -- one linearisation of point-to-plane projective ICP on gfx950.
// Replaces kfusion::device::ComputeIcpHelper::operator() (Points path) and its two kernels
// (yak/src/cuda/proj_icp.cu:74-102 find_coresp, :105-341 partial_reduce, :343-387 kernels).
// Citations are relative to /root/reference/.
//
// The reference reduces each of the 27 products with its own 256-wide shared-memory tree and
// __syncthreads pair (27 serial block reductions per 32 x 8 tile).  Here a lane walks several pixels
// and keeps the 27 sums (+ 6 outcome counters, exact in fp32) in registers; a wave folds them with
// DPP adds (no LDS), the four waves of a workgroup meet once in LDS, and the workgroup partials are added by
// a second tiny kernel -- or, in the device-resident loop, by the next linearisation's launch.  Every addition happens in a fixed order, so the result is deterministic
// run to run (no float atomics).
#include "yakhip_internal.h"

#include <chrono>

namespace yakhip
{
namespace
{
#ifndef YAK_ICP_NB
#define YAK_ICP_NB 1
#endif
constexpr int kValues = 33;        // 27 products + accepted + rejected{40,80,120,160,200}
constexpr int kPixelsPerLane = 4;  // a workgroup covers 1024 consecutive pixels (measured in the loop: 2 / 4 / 8 pixels per lane give an ICP stage of 0.217 / 0.195 / 0.222 ms)

// Running state of a device-resident ICP (yakhip_icp_estimate): the pose estimate and why the loop
// stopped, if it did.  Lives in the context's scratch; the host reads it once at the end.
struct IcpState
{
  float R[9], t[3];
  int status;     // 0 = running / converged schedule, 2 = a system needs the host's SVD path
  int iterations; // linearisations applied
};

struct IcpParams
{
  // device-resident loop (yakhip_icp_estimate).  state != nullptr: the pose comes from there, and nothing
  // happens once its status != 0.  prev_partial != nullptr as well: *state is the pose BEFORE the previous
  // linearisation, whose workgroup partials are prev_partial; every workgroup first finishes that
  // linearisation (reduce, solve, pose update -- all workgroups compute the same bits) and workgroup 0
  // records the result in *state_out for the next launch.
  const IcpState* state;
  IcpState* state_out;
  const float* prev_partial;
  int prev_blocks;
  float R[9], t[3];
  float fx, fy, cx, cy;  // level intrinsics, projective_icp.cpp:16-22
  float frows, fcols;    // projective_icp.cpp:200-201
  float min_cosine, dist2_thres;
  const uint8_t *vcurr, *ncurr, *vprev, *nprev;
  uint32_t vcurr_pitch, ncurr_pitch, vprev_pitch, nprev_pitch;
  int rows, cols;  // rows of the current maps to walk, starting at row0 (frows / fcols describe the whole image)
  int row0;
};

// (the offset in 32 bits from a 24-bit multiply: one full-rate instruction where the 64-bit form takes two at a
// quarter of the rate, on the path between the pose and the gather; the launchers check rows, pitch < 2^24 and
// rows * pitch < 2^32)
__device__ __forceinline__ float4 f4_at(const uint8_t* base, uint32_t pitch, int y, int x)
{
  return *(const float4*)(base + (__umul24((uint32_t)y, pitch) + ((uint32_t)x << 4)));
}

// Sum over the 64 lanes of a wave without touching LDS: six v_add_f32 with DPP operands (quad swaps, row
// half mirror, row mirror, then the two row broadcasts of gfx9).  The total is valid in lane 63.  Measured
// here, per launch: the __shfl_xor butterfly (ds_bpermute_b32, ~100 cycles each, 198 of them for the 33
// values of a wave) took 2.2 us of a 9 us kernel.  The order of the additions is fixed, like the butterfly's.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_term(float v)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_lane63(float v)
{
  v += dpp_term<0xb1, 0xf>(v);   // quad_perm [1, 0, 3, 2]
  v += dpp_term<0x4e, 0xf>(v);   // quad_perm [2, 3, 0, 1]
  v += dpp_term<0x141, 0xf>(v);  // row_half_mirror: 8 lanes
  v += dpp_term<0x140, 0xf>(v);  // row_mirror: 16 lanes
  v += dpp_term<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3: 32 lanes
  v += dpp_term<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3: lane 63 holds all 64
  return v;
}

// Final reduction + solve + pose update of one linearisation, without leaving the device: the
// reference goes back to the host after every one of its 19 linearisations per frame
// (projective_icp.cpp:206-229: stream sync, cv::determinant, cv::solve, compose).
//
// reduce_partials: wave w of N_WAVES adds the workgroup partials of values w, w + N_WAVES, ... -- lane j
// takes partials j, j + 64, ... in order, then the wave sum: the sum of a value does not depend on
// N_WAVES.  Results in LDS; the caller synchronises.
template <int N_WAVES, int DEEP>
__device__ __forceinline__ void reduce_partials_deep(const float* partial, int n_blocks, float* sums)
{
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  constexpr int kVals = (kValues + N_WAVES - 1) / N_WAVES;
  float v[kVals];
#pragma unroll
  for (int i = 0; i < kVals; ++i)
    v[i] = 0.f;
  for (int b0 = lane; b0 < n_blocks; b0 += 64 * DEEP)
  {
    float x[kVals][DEEP];
#pragma unroll
    for (int i = 0; i < kVals; ++i)
    {
      const int k = wave + i * N_WAVES;
#pragma unroll
      for (int u = 0; u < DEEP; ++u)
      {
        const int b = b0 + 64 * u;  // 33 * n_blocks < 2^31 (the launchers bound the image size)
        x[i][u] = k < kValues && b < n_blocks ? partial[(uint32_t)(k * n_blocks + b)] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < DEEP; ++u)
      if (b0 + 64 * u < n_blocks)
      {
#pragma unroll
        for (int i = 0; i < kVals; ++i)
          v[i] += x[i][u];
      }
  }
#pragma unroll
  for (int i = 0; i < kVals; ++i)
  {
    const float r = wave_sum_lane63(v[i]);
    const int k = wave + i * N_WAVES;
    if (lane == 63 && k < kValues)
      sums[k] = r;
  }
}

template <int N_WAVES>
__device__ __forceinline__ void reduce_partials(const float* partial, int n_blocks, float* sums)
{
  // The partials were written by the previous launch's workgroups on every XCD, so each load here is a
  // trip to the fabric (about a microsecond): a wave issues the loads of ALL its values before it adds
  // anything -- one trip for up to 512 workgroups -- and then adds in the fixed order (which does not depend
  // on the depth chosen here: lane j always takes partials j, j + 64, ...).  The coarse pyramid levels have
  // few workgroups: they get a shallower body instead of dozens of loads that are predicated off.
  if (n_blocks <= 64)  // wave-uniform
    reduce_partials_deep<N_WAVES, 1>(partial, n_blocks, sums);
  else if (n_blocks <= 128)
    reduce_partials_deep<N_WAVES, 2>(partial, n_blocks, sums);
  else
    reduce_partials_deep<N_WAVES, 8>(partial, n_blocks, sums);
}

// 1 / sqrt(x) in double for normal positive x: the hardware estimate and two Newton steps (an ulp or two
// from the correctly rounded value).  The IEEE division / square-root sequences cost a dozen dependent
// instructions each and the solve below is one serial chain of them; with 1 / L_ii in hand every division
// of the factorisation and the substitutions is a multiplication.
__device__ __forceinline__ double fast_rsqrt(double x)
{
  double r = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  r = __builtin_fma(r, __builtin_fma(-h * r, r, 0.5), r);
  return __builtin_fma(r, __builtin_fma(-h * r, r, 0.5), r);
}

// solve_update: every lane that calls it computes the same thing (a pure function of sums and st).
// The normal matrix is symmetric positive definite whenever the reference's own checks pass, so a
// Cholesky factorisation in double gives the solution cv::solve's SVD gives, as long as the SVD would
// not truncate a singular value.  That is certified, not assumed: lambda_min >= det / (trace / 5)^5
// (AM-GM on the other five eigenvalues) or, failing that, lambda_min >= 1 / trace(A^-1) (the triangular
// inverse of the factor, rare path) must exceed OpenCV's cut-off 2 * FLT_EPSILON * trace four times, and det
// must be clear of the 1e-15 test; anything else sets status = 2 and the host's LU / SVD path
// (solve6.hpp) redoes the estimate.
__device__ __forceinline__ void solve_update(const float* sums, IcpState& st)
{
  // Double precision with the fused multiply-adds written out (they halve the dependent chain): this translation
  // unit is compiled without contraction, so the arithmetic is the same in whichever kernel the function is
  // inlined -- the chain of launches and the one-launch schedule give the same bits.
  // StreamHelper::get, projective_icp.cpp:42-61.  Every loop below has constant bounds and is unrolled,
  // so the 6 x 6 system lives in registers (double precision).
  double A[6][6], b[6];
  {
    int shift = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 7; ++j)
      {
        const double value = sums[shift++];
        if (j == 6)
          b[i] = value;
        else
          A[j][i] = A[i][j] = value;
      }
  }
  double trace = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i)
    trace += A[i][i];

  // Cholesky A = L L^T (lower triangle in place), det = prod L_ii^2; 1 / L_ii kept for the substitutions
  double det = 1.0, inv[6];
  bool spd = trace == trace;
#pragma unroll
  for (int j = 0; j < 6; ++j)
  {
    double d = A[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k)
      d = __builtin_fma(-A[j][k], A[j][k], d);
    spd = spd && d > 0.0;
    const double dd = spd ? d : 1.0;
    const double rs = fast_rsqrt(dd);
    double l = dd * rs;
    l = __builtin_fma(__builtin_fma(-l, l, dd), 0.5 * rs, l);
    A[j][j] = l;
    inv[j] = rs;
    det *= d;
#pragma unroll
    for (int i = j + 1; i < 6; ++i)
    {
      double v = A[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k)
        v = __builtin_fma(-A[i][k], A[j][k], v);
      A[i][j] = v * rs;
    }
  }
  const double m = trace * 0.2;
  const double cutoff = 2.0 * 1.1920928955078125e-7 * trace * 4.0;  // OpenCV's singular-value cut-off, with a margin of 4
  bool certified = spd && det > 1e-12 && det > cutoff * (m * m * m * m * m);
  if (!certified && spd && det > 1e-12)
  {
    // The AM-GM bound is loose when the eigenvalues are spread (views where one direction is weakly
    // constrained: 11 frames of the 300-frame orbit).  Rare path, the tight bound: lambda_min >= 1 / trace(A^-1)
    // = 1 / ||L^-1||_F^2 (within a factor of 6 of lambda_min), from the triangular inverse of the factor at hand.
    double M[6][6], frob = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j)
    {
      M[j][j] = inv[j];
      frob = __builtin_fma(inv[j], inv[j], frob);
#pragma unroll
      for (int i = j + 1; i < 6; ++i)
      {
        double v = 0.0;
#pragma unroll
        for (int k = j; k < i; ++k)
          v = __builtin_fma(-A[i][k], M[k][j], v);
        M[i][j] = v * inv[i];
        frob = __builtin_fma(M[i][j], M[i][j], frob);
      }
    }
    certified = frob * cutoff < 1.0;
  }
  if (!certified)
  {
    st.status = 2;
    return;
  }
  double y[6], r[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
  {
    double v = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k)
      v = __builtin_fma(-A[i][k], y[k], v);
    y[i] = v * inv[i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i)
  {
    double v = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k)
      v = __builtin_fma(-A[k][i], r[k], v);
    r[i] = v * inv[i];
  }

  // Tinc = Affine3f(rvec, tvec) (Rodrigues), corrected = Tinc * corrected: projective_icp.cpp:227-228,
  // in the float arithmetic of the host type (include/yak/kfusion/compat/linalg.hpp)
  const float rv[3] = { (float)r[0], (float)r[1], (float)r[2] };
  const float tinc[3] = { (float)r[3], (float)r[4], (float)r[5] };
  float Ri[9] = { 1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f };
  const double th2 = __builtin_fma((double)rv[2], (double)rv[2], __builtin_fma((double)rv[1], (double)rv[1], (double)rv[0] * rv[0]));
  if (th2 > 2.220446049250313e-16 * 2.220446049250313e-16)  // theta > DBL_EPSILON
  {
    const double inv_th = fast_rsqrt(th2), th = th2 * inv_th;
    double s, c;
    sincos(th, &s, &c);
    const double kx = rv[0] * inv_th, ky = rv[1] * inv_th, kz = rv[2] * inv_th, v = 1.0 - c;
    const double xx = kx * kx * v, yy = ky * ky * v, zz = kz * kz * v, xy = kx * ky * v, xz = kx * kz * v, yz = ky * kz * v;
    Ri[0] = (float)(c + xx), Ri[1] = (float)__builtin_fma(-kz, s, xy), Ri[2] = (float)__builtin_fma(ky, s, xz);
    Ri[3] = (float)__builtin_fma(kz, s, xy), Ri[4] = (float)(c + yy), Ri[5] = (float)__builtin_fma(-kx, s, yz);
    Ri[6] = (float)__builtin_fma(-ky, s, xz), Ri[7] = (float)__builtin_fma(kx, s, yz), Ri[8] = (float)(c + zz);
  }
  float Rn[9], tn[3];
#pragma unroll
  for (int rr = 0; rr < 3; ++rr)
  {
#pragma unroll
    for (int cc = 0; cc < 3; ++cc)
      Rn[3 * rr + cc] = Ri[3 * rr + 0] * st.R[0 + cc] + Ri[3 * rr + 1] * st.R[3 + cc] + Ri[3 * rr + 2] * st.R[6 + cc];
    tn[rr] = (Ri[3 * rr + 0] * st.t[0] + Ri[3 * rr + 1] * st.t[1] + Ri[3 * rr + 2] * st.t[2]) + tinc[rr];
  }
#pragma unroll
  for (int i = 0; i < 9; ++i)
    st.R[i] = Rn[i];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    st.t[i] = tn[i];
  st.iterations += 1;
}

__device__ __forceinline__ void load_state(const IcpState* src, IcpState& st)  // wave-uniform: scalar loads
{
#pragma unroll
  for (int i = 0; i < 9; ++i)
    st.R[i] = src->R[i];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    st.t[i] = src->t[i];
  st.status = src->status;
  st.iterations = src->iterations;
}

// The pixels of one lane of a 256-lane block: block `block` covers the 1024 consecutive pixels from block * 1024.
struct CurrentMaps
{
  bool live[kPixelsPerLane];
  float4 s0[kPixelsPerLane], nc[kPixelsPerLane];
};

// The loads of the current maps do not depend on the pose: they are issued before the pose is worked out and
// fly meanwhile.
__device__ __forceinline__ void load_current(const IcpParams& p, int block, int tid, CurrentMaps& c)
{
  const int n_pixels = p.rows * p.cols;
  const int last = n_pixels > 0 ? n_pixels - 1 : 0;  // (an empty band of the sharded schedule: pixel (row0, 0), a valid row)
  // Row and column of the block's first pixel: one integer division per wave (`block` is the same in all its lanes).
  // The lanes' pixels lie at most 1023 further on: their rows and columns come from a float quotient of numbers
  // below 2^24, put right by one either way -- a handful of full-rate instructions per pixel where the 32-bit
  // division takes some thirty, a third of them at a quarter of the rate, in front of the first load.
  const int first = min(__builtin_amdgcn_readfirstlane(block) * (256 * kPixelsPerLane), last);
  const int row_first = first / p.cols, col_first = first - row_first * p.cols;
  const float inv_cols = __builtin_amdgcn_rcpf((float)p.cols);
#pragma unroll
  for (int i = 0; i < kPixelsPerLane; ++i)
  {
    const int pix = first + tid + i * 256;
    c.live[i] = pix < n_pixels;
    const int col = col_first + ((c.live[i] ? pix : last) - first);  // a dead lane re-reads the band's last pixel
    int q = (int)((float)col * inv_cols);
    int x = col - (int)__umul24((uint32_t)q, (uint32_t)p.cols);
    q += x < 0 ? -1 : (x >= p.cols ? 1 : 0);
    x += x < 0 ? p.cols : (x >= p.cols ? -p.cols : 0);
    const int y = row_first + q + p.row0;
    c.s0[i] = f4_at(p.vcurr, p.vcurr_pitch, y, x);
    c.nc[i] = f4_at(p.ncurr, p.ncurr_pitch, y, x);
  }
}

// The four pixels of a lane are independent until their sums meet, and the kernel is latency-bound (one
// or two waves per SIMD, each pixel a chain load -> project -> gather): all four current-map loads go
// out together (load_current), then all four gathers, then the arithmetic -- pixel by pixel in the old order, so
// the sums are bit for bit what the one-pixel-at-a-time loop gave.
__device__ __forceinline__ void accumulate_pixels(const IcpParams& p, const float (&R)[9], const float (&t)[3],
                                                  const CurrentMaps& c, float (&acc)[kValues])
{
  float4 d[kPixelsPerLane], nd[kPixelsPerLane];
  int code[kPixelsPerLane];
  float sx[kPixelsPerLane], sy[kPixelsPerLane], sz[kPixelsPerLane];
#pragma unroll
  for (int i = 0; i < kPixelsPerLane; ++i)
  {
    // find_coresp, proj_icp.cu:74-102, up to the fetch of the destination point
    code[i] = 0;
    sx[i] = sy[i] = sz[i] = 0.f;
    int px = 0, py = 0;
    if (c.s0[i].x != c.s0[i].x)
      code[i] = 40;
    else
    {
      sx[i] = dot3(R[0], R[1], R[2], c.s0[i].x, c.s0[i].y, c.s0[i].z) + t[0];
      sy[i] = dot3(R[3], R[4], R[5], c.s0[i].x, c.s0[i].y, c.s0[i].z) + t[1];
      sz[i] = dot3(R[6], R[7], R[8], c.s0[i].x, c.s0[i].y, c.s0[i].z) + t[2];
      if (sz[i] <= 0.f)
        code[i] = 80;
      else
      {
        const float coox = __builtin_fmaf(p.fx, sx[i] / sz[i], p.cx), cooy = __builtin_fmaf(p.fy, sy[i] / sz[i], p.cy);
        if (!(coox >= 0.f && cooy >= 0.f && coox < p.fcols && cooy < p.frows))
          code[i] = 80;
        else
          px = (int)__builtin_floorf(coox), py = (int)__builtin_floorf(cooy);  // point-filtered fetch
      }
    }
    d[i] = f4_at(p.vprev, p.vprev_pitch, py, px);  // pixel (0, 0) for a rejected one: read and ignored
    nd[i] = f4_at(p.nprev, p.nprev_pitch, py, px);
  }
#pragma unroll
  for (int i = 0; i < kPixelsPerLane; ++i)
  {
    if (!c.live[i])
      break;
    int cd = code[i];
    if (cd == 0)
    {
      if (d[i].x != d[i].x)
        cd = 120;
      else
      {
        const float ex = sx[i] - d[i].x, ey = sy[i] - d[i].y, ez = sz[i] - d[i].z;
        if (dot3(ex, ey, ez, ex, ey, ez) > p.dist2_thres)
          cd = 160;
        else
        {
          const float nsx = dot3(R[0], R[1], R[2], c.nc[i].x, c.nc[i].y, c.nc[i].z);
          const float nsy = dot3(R[3], R[4], R[5], c.nc[i].x, c.nc[i].y, c.nc[i].z);
          const float nsz = dot3(R[6], R[7], R[8], c.nc[i].x, c.nc[i].y, c.nc[i].z);
          const float cosine = __builtin_fabsf(dot3(nsx, nsy, nsz, nd[i].x, nd[i].y, nd[i].z));
          if (cosine < p.min_cosine)
            cd = 200;
        }
      }
    }

    if (cd == 0)
    {
      // proj_icp.cu:354-359  row = [cross(s, n), n, dot(n, d - s)]
      float row[7];
      row[0] = sy[i] * nd[i].z - sz[i] * nd[i].y;
      row[1] = sz[i] * nd[i].x - sx[i] * nd[i].z;
      row[2] = sx[i] * nd[i].y - sy[i] * nd[i].x;
      row[3] = nd[i].x, row[4] = nd[i].y, row[5] = nd[i].z;
      row[6] = dot3(nd[i].x, nd[i].y, nd[i].z, d[i].x - sx[i], d[i].y - sy[i], d[i].z - sz[i]);
      int k = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 7; ++b)
          acc[k++] += row[a] * row[b];  // order of proj_icp.cu:122-335
      acc[27] += 1.f;
    }
    else
    {
      acc[28] += cd == 40 ? 1.f : 0.f;
      acc[29] += cd == 80 ? 1.f : 0.f;
      acc[30] += cd == 120 ? 1.f : 0.f;
      acc[31] += cd == 160 ? 1.f : 0.f;
      acc[32] += cd == 200 ? 1.f : 0.f;
    }
  }
}

// The 33 sums of a 256-lane block from its lanes' sums: wave sums (in lane 63), then the block's four waves through
// LDS (`lds` is this block's [4][kValues]); the caller synchronises, then lane k < kValues of the block holds value k
// in block_value().
__device__ __forceinline__ void block_sums_to_lds(float (&acc)[kValues], float (*lds)[kValues], int tid)
{
#pragma unroll
  for (int k = 0; k < kValues; ++k)
    acc[k] = wave_sum_lane63(acc[k]);
  const int lane = tid & 63, wave = tid >> 6;
  if (lane == 63)
  {
#pragma unroll
    for (int k = 0; k < kValues; ++k)
      lds[wave][k] = acc[k];
  }
}
__device__ __forceinline__ float block_value(const float (*lds)[kValues], int k)
{
  return ((lds[0][k] + lds[1][k]) + lds[2][k]) + lds[3][k];
}

// NB blocks of 256 lanes per workgroup; the block sums and their order are the same for every NB.  Measured in the
// fusion loop, same box: NB = 1 2 340-2 370 frames/s, NB = 2 (half as many workgroups of 512 lanes) 2 280-2 320.
template <int NB>
__global__ __launch_bounds__(256 * NB) void icp_accumulate_kernel(const IcpParams p, float* partial, int n_blocks)
{
  float acc[kValues];
#pragma unroll
  for (int k = 0; k < kValues; ++k)
    acc[k] = 0.f;

  const int tid = (int)threadIdx.x & 255, sub = (int)threadIdx.x >> 8;
  const int block = (int)blockIdx.x * NB + sub;
  const bool exists = block < n_blocks;
  CurrentMaps cur;
  load_current(p, exists ? block : 0, tid, cur);
  float R[9], t[3];
  if (p.state)
  {
    IcpState st;
    load_state(p.state, st);
    if (p.prev_partial)
    {
      // not gated by the status: the partial loads then leave together with the state's scalar loads (one
      // trip to the fabric instead of two); after a stop they read stale scratch, which nobody uses
      __shared__ float prev_sums[kValues];
      reduce_partials<4 * NB>(p.prev_partial, p.prev_blocks, prev_sums);
      __syncthreads();
      if (st.status == 0)
        solve_update(prev_sums, st);
      if (blockIdx.x == 0 && threadIdx.x == 0)
        *p.state_out = st;
    }
    if (st.status != 0)
      return;
#pragma unroll
    for (int i = 0; i < 9; ++i)
      R[i] = st.R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      t[i] = st.t[i];
  }
  else
  {
#pragma unroll
    for (int i = 0; i < 9; ++i)
      R[i] = p.R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      t[i] = p.t[i];
    // the first launch of a device-resident estimate: the pose came with the launch (no copy command in front of
    // the chain); it goes into the state slot the next launch starts from
    if (p.state_out && blockIdx.x == 0 && threadIdx.x == 0)
    {
      IcpState st;
#pragma unroll
      for (int i = 0; i < 9; ++i)
        st.R[i] = p.R[i];
#pragma unroll
      for (int i = 0; i < 3; ++i)
        st.t[i] = p.t[i];
      st.status = 0, st.iterations = 0;
      *p.state_out = st;
    }
  }

  if (exists)
    accumulate_pixels(p, R, t, cur, acc);

  __shared__ float lds[NB][4][kValues];
  block_sums_to_lds(acc, lds[sub], tid);
  __syncthreads();
  if (tid < kValues && exists)
    partial[(size_t)tid * n_blocks + block] = block_value(lds[sub], tid);
}

// One wave per value: lane j adds partials j, j + 64, ... in order, then the butterfly.
__global__ __launch_bounds__(64) void icp_final_kernel(const float* partial, int n_blocks, float* out)
{
  const int k = (int)blockIdx.x;
  float v = 0.f;
  for (int b = (int)threadIdx.x; b < n_blocks; b += 64)
    v += partial[(size_t)k * n_blocks + b];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1)
    v += __shfl_xor(v, off, 64);
  if (threadIdx.x == 0)
    out[k] = v;
}
// The last linearisation of a device-resident estimate has no accumulate launch behind it to finish it:
// one workgroup does (state_in, partial) -> state_out.
// host_out / host_ticket (pinned host memory, may be null): the result is also written straight to the host,
// followed by the call's ticket -- the host polls the ticket instead of waiting for a copy and a stream
// synchronisation behind this kernel (launch_icp_estimate).
__global__ __launch_bounds__(1024) void icp_solve_kernel(const float* partial, int n_blocks, const IcpState* state_in,
                                                         IcpState* state_out, IcpState* host_out, unsigned* host_ticket,
                                                         unsigned ticket)
{
  __shared__ float sums[kValues];
  IcpState st;
  load_state(state_in, st);
  reduce_partials<16>(partial, n_blocks, sums);
  __syncthreads();
  if (st.status == 0)
    solve_update(sums, st);
  if (threadIdx.x == 0)
  {
    *state_out = st;
    if (host_out)
    {
      *host_out = st;
      __hip_atomic_store(host_ticket, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
}  // namespace

static int icp_scratch(yakhip_ctx* ctx, size_t need);

// what the kernels' 24-bit index arithmetic needs (f4_at, load_current): far beyond any camera
static bool icp_fits(int rows, int cols, size_t pitch_a, size_t pitch_b, size_t pitch_c, size_t pitch_d)
{
  size_t pitch = pitch_a > pitch_b ? pitch_a : pitch_b;
  pitch = pitch > pitch_c ? pitch : pitch_c;
  pitch = pitch > pitch_d ? pitch : pitch_d;
  return (long long)rows * cols <= 0x7fffffffll - 2048 && rows < (1 << 24) && cols < (1 << 22) && pitch < (1u << 24) &&
         (unsigned long long)rows * pitch < (1ull << 32);
}

int launch_icp_accumulate(yakhip_ctx* ctx, const float R[9], const float t[3], const float intr[4], int level,
                          float dist_thres, float angle_thres, const uint8_t* vcurr, size_t vcurr_pitch,
                          const uint8_t* ncurr, size_t ncurr_pitch, const uint8_t* vprev, size_t vprev_pitch,
                          const uint8_t* nprev, size_t nprev_pitch, int rows, int cols, float sums[27], int counts[6],
                          int row_begin, int row_end, float* sums_dev)
{
  IcpParams p;
  p.state = nullptr, p.state_out = nullptr, p.prev_partial = nullptr, p.prev_blocks = 0;
  for (int i = 0; i < 9; ++i)
    p.R[i] = R[i];
  for (int i = 0; i < 3; ++i)
    p.t[i] = t[i];
  const int div = 1 << level;  // projective_icp.cpp:16-22
  p.fx = intr[0] / div, p.fy = intr[1] / div, p.cx = intr[2] / div, p.cy = intr[3] / div;
  p.frows = (float)rows, p.fcols = (float)cols;
  p.min_cosine = (float)cos((double)angle_thres);  // projective_icp.cpp:12-13
  p.dist2_thres = dist_thres * dist_thres;
  p.vcurr = vcurr, p.ncurr = ncurr, p.vprev = vprev, p.nprev = nprev;
  p.vcurr_pitch = (uint32_t)vcurr_pitch, p.ncurr_pitch = (uint32_t)ncurr_pitch;
  p.vprev_pitch = (uint32_t)vprev_pitch, p.nprev_pitch = (uint32_t)nprev_pitch;
  p.rows = row_end - row_begin, p.cols = cols, p.row0 = row_begin;

  const long long n_pixels = (long long)p.rows * cols;
  if (!icp_fits(rows, cols, vcurr_pitch, ncurr_pitch, vprev_pitch, nprev_pitch))
    return fail(YAKHIP_ERANGE, "icp: image too large");
  if (n_pixels == 0)  // an empty row range (a rank with no rows of a coarse level): all sums are zero
    return check(hipMemsetAsync(sums_dev, 0, kValues * sizeof(float), ctx->stream), "hipMemsetAsync");
  const int n_blocks = (int)((n_pixels + 256 * kPixelsPerLane - 1) / (256 * kPixelsPerLane));

  // context-owned scratch: block partials + the 33 results on the device, a pinned host mirror
  const size_t need = ((size_t)n_blocks * kValues + kValues) * sizeof(float);
  int rc = icp_scratch(ctx, need);
  if (rc != 0)
    return rc;
  float* partial = (float*)ctx->icp_scratch;
  float* result = partial + (size_t)n_blocks * kValues;
  hipLaunchKernelGGL(icp_accumulate_kernel<1>, dim3((unsigned)n_blocks), dim3(256), 0, ctx->stream, p, partial, n_blocks);
  YAKHIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(icp_final_kernel, dim3(kValues), dim3(64), 0, ctx->stream, partial, n_blocks,
                     sums_dev ? sums_dev : result);
  YAKHIP_TRY(hipGetLastError());
  if (sums_dev)
    return 0;  // rows variant: the 33 values stay on the device, nothing to wait for
  YAKHIP_TRY(hipMemcpyAsync(ctx->icp_pinned, result, kValues * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  YAKHIP_TRY(hipStreamSynchronize(ctx->stream));  // StreamHelper::get, projective_icp.cpp:44
  const float* h = (const float*)ctx->icp_pinned;
  for (int k = 0; k < 27; ++k)
    sums[k] = h[k];
  if (counts)
    for (int k = 0; k < 6; ++k)
      counts[k] = (int)h[27 + k];
  return 0;
}

static int icp_scratch(yakhip_ctx* ctx, size_t need)
{
  if (ctx->icp_scratch_bytes < need)
  {
    if (ctx->icp_scratch)
    {
      YAKHIP_TRY(hipStreamSynchronize(ctx->stream));
      YAKHIP_TRY(hipFree(ctx->icp_scratch));
    }
    ctx->icp_scratch = nullptr;
    ctx->icp_scratch_bytes = 0;
    YAKHIP_TRY(hipMalloc(&ctx->icp_scratch, need));
    ctx->icp_scratch_bytes = need;
  }
  if (!ctx->icp_pinned)
    YAKHIP_TRY(hipHostMalloc(&ctx->icp_pinned, 256, hipHostMallocMapped | hipHostMallocCoherent));  // (the device writes to it: icp_solve_kernel)
  return 0;
}

// The whole coarse-to-fine schedule of ProjectiveICP::estimateTransform (projective_icp.cpp:175-232)
// enqueued back to back, ONE launch per linearisation: launch j first finishes linearisation j - 1 (every
// workgroup reduces its partials and solves -- redundant, identical, and a few microseconds of latency
// that a separate launch would also have cost, without that launch's gap), then accumulates its own.  The
// pose travels through a chain of state slots (launch j reads slot j - 1 and workgroup 0 writes slot j),
// the partials through two alternating buffers, so no launch writes what a workgroup of the same launch
// reads.  A single-workgroup launch finishes the last linearisation; the host waits once, at the end.
// Measured per frame at 640 x 480, 10/5/4 linearisations: 0.34 ms with a reduce+solve launch behind every
// accumulate launch (that launch alone took 13 us: five dependent trips to the fabric for the partials and
// an IEEE-division Cholesky on one lane), 0.28 ms with this schedule, 0.22 ms with DPP wave sums; a launch now
// takes 9-10 us whatever the level -- a chain of state + partials load, solve, gather, sums, store -- and
// the gaps 1-2 us.
// (Folding the reduce+solve into the SAME accumulate launch -- last workgroup by ticket -- was measured and is
// slower, 0.52 ms: the device-scope release/acquire fences every workgroup then needs write back and
// invalidate the XCD's L2, which costs more than a launch.)
int launch_icp_estimate(yakhip_ctx* ctx, const float R[9], const float t[3], const float intr[4], float dist_thres,
                        float angle_thres, const yakhip_icp_level* levels, int n_levels, float R_out[9], float t_out[3],
                        int* status_out, int* iterations_out)
{
  int max_blocks = 1;
  long long n_launches = 0;
  for (int l = 0; l < n_levels; ++l)
  {
    const long long n_pixels = (long long)levels[l].rows * levels[l].cols;
    if (!icp_fits(levels[l].rows, levels[l].cols, levels[l].vcurr_pitch, levels[l].ncurr_pitch, levels[l].vprev_pitch,
                  levels[l].nprev_pitch))
      return fail(YAKHIP_ERANGE, "icp: image too large");
    const int nb = (int)((n_pixels + 256 * kPixelsPerLane - 1) / (256 * kPixelsPerLane));
    if (nb > max_blocks)
      max_blocks = nb;
    n_launches += levels[l].iterations > 0 ? levels[l].iterations : 0;
  }
  if (n_launches > 100000)
    return fail(YAKHIP_ERANGE, "icp: more than 100000 linearisations");
  // scratch: partials A | partials B | state slots 0 .. n_launches
  const size_t partial_bytes = ((size_t)max_blocks * kValues * sizeof(float) + 255u) & ~(size_t)255u;
  int rc = icp_scratch(ctx, 2 * partial_bytes + ((size_t)n_launches + 1) * sizeof(IcpState));
  if (rc != 0)
    return rc;
  float* partial[2] = { (float*)ctx->icp_scratch, (float*)((char*)ctx->icp_scratch + partial_bytes) };
  IcpState* slots = (IcpState*)((char*)ctx->icp_scratch + 2 * partial_bytes);

  IcpState* host = (IcpState*)ctx->icp_pinned;
  for (int i = 0; i < 9; ++i)
    host->R[i] = R[i];
  for (int i = 0; i < 3; ++i)
    host->t[i] = t[i];
  host->status = 0;
  host->iterations = 0;
  if (n_launches > 0)
  {
    int j = 0, prev_blocks = 0;
    for (int l = n_levels - 1; l >= 0; --l)
    {
      const yakhip_icp_level& L = levels[l];
      if (L.iterations <= 0)
        continue;
      IcpParams p;
      for (int i = 0; i < 9; ++i)
        p.R[i] = 0.f;
      for (int i = 0; i < 3; ++i)
        p.t[i] = 0.f;
      const int div = 1 << l;  // projective_icp.cpp:16-22
      p.fx = intr[0] / div, p.fy = intr[1] / div, p.cx = intr[2] / div, p.cy = intr[3] / div;
      p.frows = (float)L.rows, p.fcols = (float)L.cols;
      p.min_cosine = (float)cos((double)angle_thres);
      p.dist2_thres = dist_thres * dist_thres;
      p.vcurr = (const uint8_t*)L.vcurr, p.ncurr = (const uint8_t*)L.ncurr;
      p.vprev = (const uint8_t*)L.vprev, p.nprev = (const uint8_t*)L.nprev;
      p.vcurr_pitch = (uint32_t)L.vcurr_pitch, p.ncurr_pitch = (uint32_t)L.ncurr_pitch;
      p.vprev_pitch = (uint32_t)L.vprev_pitch, p.nprev_pitch = (uint32_t)L.nprev_pitch;
      p.rows = L.rows, p.cols = L.cols, p.row0 = 0;
      const int n_blocks = (int)(((long long)L.rows * L.cols + 256 * kPixelsPerLane - 1) / (256 * kPixelsPerLane));
      for (int it = 0; it < L.iterations; ++it, ++j)
      {
        // launch 0 takes the pose from its arguments and records it in slot 0; launch j > 0 turns slot j - 1 and
        // launch j - 1's partials into slot j
        p.state = j == 0 ? nullptr : &slots[j - 1];
        p.state_out = &slots[j];
        if (j == 0)
        {
          for (int i = 0; i < 9; ++i)
            p.R[i] = R[i];
          for (int i = 0; i < 3; ++i)
            p.t[i] = t[i];
        }
        p.prev_partial = j == 0 ? nullptr : partial[(j - 1) & 1];
        p.prev_blocks = prev_blocks;
        hipLaunchKernelGGL(icp_accumulate_kernel<YAK_ICP_NB>, dim3((unsigned)((n_blocks + YAK_ICP_NB - 1) / YAK_ICP_NB)), dim3(256 * YAK_ICP_NB), 0, ctx->stream, p, partial[j & 1],
                           n_blocks);
        prev_blocks = n_blocks;
      }
      YAKHIP_TRY(hipGetLastError());
    }
    // j == n_launches: the last launch read slot max(j - 2, 0) and (for j > 1) wrote slot j - 1
    // The result comes back through pinned memory the last kernel writes itself: its state, then this call's
    // ticket (release, system scope).  The host polls the ticket -- a few microseconds after the kernel's last
    // store instead of a copy command, its completion signal and the wake-up of a sleeping thread (measured
    // in the fusion loop: see DESIGN.md 3.7).  A ticket that does not come within two seconds is looked for
    // the ordinary way, which also brings any error of the stream to light.
    IcpState* result = (IcpState*)((char*)ctx->icp_pinned + 128);
    unsigned* ticket_at = (unsigned*)((char*)ctx->icp_pinned + 192);
    const bool poll = ctx->icp_poll != 0;
    const unsigned ticket = ++ctx->icp_ticket;
    hipLaunchKernelGGL(icp_solve_kernel, dim3(1), dim3(1024), 0, ctx->stream, partial[(j - 1) & 1], prev_blocks,
                       &slots[j - 1], &slots[j], poll ? result : nullptr, ticket_at, ticket);
    YAKHIP_TRY(hipGetLastError());
    bool arrived = false;
    if (poll)
    {
      const auto t0 = std::chrono::steady_clock::now();
      for (unsigned spins = 1; !arrived; ++spins)
      {
        arrived = __atomic_load_n(ticket_at, __ATOMIC_ACQUIRE) == ticket;
        if (!arrived && (spins & 0xfffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2))
          break;
      }
      if (arrived)
        *host = *result;
    }
    if (!arrived)
    {
      YAKHIP_TRY(hipMemcpyAsync(host, &slots[j], sizeof(IcpState), hipMemcpyDeviceToHost, ctx->stream));
      YAKHIP_TRY(hipStreamSynchronize(ctx->stream));
    }
  }
  for (int i = 0; i < 9; ++i)
    R_out[i] = host->R[i];
  for (int i = 0; i < 3; ++i)
    t_out[i] = host->t[i];
  *status_out = host->status;
  *iterations_out = host->iterations;
  return 0;
}
// The same schedule over the ranks of a communicator (image-space split).  Rank r walks the rows
// [rows * r / world, rows * (r + 1) / world) of every level; per linearisation j:
//   accumulate launch   first finishes linearisation j - 1 from the ALL-REDUCED 33 values of that linearisation
//                       (they stand in for "one workgroup's partials": prev_blocks = 1), then accumulates its
//                       band with the new pose into its own workgroup partials;
//   final launch        33 waves fold those partials into 33 values;
//   all-reduce          SUM of the 33 floats over the ranks, on the stream (132 bytes: latency only).
// Every rank solves from the same bits, so the poses stay identical without ever being exchanged.  The 33-value
// buffers alternate like the partials of the one-GPU schedule.  A rank whose band is empty on a coarse level
// contributes zeros.
int launch_icp_estimate_sharded(yakhip_ctx* ctx, yakhip_comm* comm, const float R[9], const float t[3], const float intr[4],
                                float dist_thres, float angle_thres, const yakhip_icp_level* levels, int n_levels,
                                float R_out[9], float t_out[3], int* status_out, int* iterations_out)
{
  const int world = comm_world(comm), rank = comm_rank(comm);
  int max_blocks = 1;
  long long n_launches = 0;
  for (int l = 0; l < n_levels; ++l)
  {
    const long long n_pixels = (long long)levels[l].rows * levels[l].cols;
    if (!icp_fits(levels[l].rows, levels[l].cols, levels[l].vcurr_pitch, levels[l].ncurr_pitch, levels[l].vprev_pitch,
                  levels[l].nprev_pitch))
      return fail(YAKHIP_ERANGE, "icp: image too large");
    const int nb = (int)((n_pixels + 256 * kPixelsPerLane - 1) / (256 * kPixelsPerLane));
    if (nb > max_blocks)
      max_blocks = nb;
    n_launches += levels[l].iterations > 0 ? levels[l].iterations : 0;
  }
  if (n_launches > 100000)
    return fail(YAKHIP_ERANGE, "icp: more than 100000 linearisations");
  // scratch: partials | sums A | sums B | state slots 0 .. n_launches
  const size_t partial_bytes = ((size_t)max_blocks * kValues * sizeof(float) + 255u) & ~(size_t)255u;
  int rc = icp_scratch(ctx, partial_bytes + 512 + ((size_t)n_launches + 1) * sizeof(IcpState));
  if (rc != 0)
    return rc;
  float* partial = (float*)ctx->icp_scratch;
  float* sums[2] = { (float*)((char*)ctx->icp_scratch + partial_bytes), (float*)((char*)ctx->icp_scratch + partial_bytes + 256) };
  IcpState* slots = (IcpState*)((char*)ctx->icp_scratch + partial_bytes + 512);

  IcpState* host = (IcpState*)ctx->icp_pinned;
  for (int i = 0; i < 9; ++i)
    host->R[i] = R[i];
  for (int i = 0; i < 3; ++i)
    host->t[i] = t[i];
  host->status = 0;
  host->iterations = 0;
  if (n_launches > 0)
  {
    YAKHIP_TRY(hipMemcpyAsync(&slots[0], host, sizeof(IcpState), hipMemcpyHostToDevice, ctx->stream));
    int j = 0;
    for (int l = n_levels - 1; l >= 0; --l)
    {
      const yakhip_icp_level& L = levels[l];
      if (L.iterations <= 0)
        continue;
      IcpParams p;
      for (int i = 0; i < 9; ++i)
        p.R[i] = 0.f;
      for (int i = 0; i < 3; ++i)
        p.t[i] = 0.f;
      const int div = 1 << l;  // projective_icp.cpp:16-22
      p.fx = intr[0] / div, p.fy = intr[1] / div, p.cx = intr[2] / div, p.cy = intr[3] / div;
      p.frows = (float)L.rows, p.fcols = (float)L.cols;
      p.min_cosine = (float)cos((double)angle_thres);
      p.dist2_thres = dist_thres * dist_thres;
      p.vcurr = (const uint8_t*)L.vcurr, p.ncurr = (const uint8_t*)L.ncurr;
      p.vprev = (const uint8_t*)L.vprev, p.nprev = (const uint8_t*)L.nprev;
      p.vcurr_pitch = (uint32_t)L.vcurr_pitch, p.ncurr_pitch = (uint32_t)L.ncurr_pitch;
      p.vprev_pitch = (uint32_t)L.vprev_pitch, p.nprev_pitch = (uint32_t)L.nprev_pitch;
      const int r0 = (int)((long long)L.rows * rank / world), r1 = (int)((long long)L.rows * (rank + 1) / world);
      p.rows = r1 - r0, p.cols = L.cols, p.row0 = r0;
      // at least one workgroup even for an empty band: the launch also carries the solve of the previous
      // linearisation and the state slot every later launch reads (an empty band accumulates nothing: rows = 0)
      const long long band = (long long)p.rows * L.cols;
      const int n_blocks = band > 0 ? (int)((band + 256 * kPixelsPerLane - 1) / (256 * kPixelsPerLane)) : 1;
      for (int it = 0; it < L.iterations; ++it, ++j)
      {
        p.state = j == 0 ? &slots[0] : &slots[j - 1];
        p.state_out = &slots[j];
        p.prev_partial = j == 0 ? nullptr : sums[(j - 1) & 1];
        p.prev_blocks = 1;
        hipLaunchKernelGGL(icp_accumulate_kernel<1>, dim3((unsigned)n_blocks), dim3(256), 0, ctx->stream, p, partial, n_blocks);
        hipLaunchKernelGGL(icp_final_kernel, dim3(kValues), dim3(64), 0, ctx->stream, partial, n_blocks, sums[j & 1]);
        YAKHIP_TRY(hipGetLastError());
        if ((rc = comm_allreduce(ctx, comm, sums[j & 1], kValues, YAKHIP_SUM_F32)) != 0)
          return rc;
      }
    }
    hipLaunchKernelGGL(icp_solve_kernel, dim3(1), dim3(1024), 0, ctx->stream, sums[(j - 1) & 1], 1, &slots[j - 1], &slots[j], nullptr, nullptr, 0u);
    YAKHIP_TRY(hipGetLastError());
    YAKHIP_TRY(hipMemcpyAsync(host, &slots[j], sizeof(IcpState), hipMemcpyDeviceToHost, ctx->stream));
    YAKHIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  for (int i = 0; i < 9; ++i)
    R_out[i] = host->R[i];
  for (int i = 0; i < 3; ++i)
    t_out[i] = host->t[i];
  *status_out = host->status;
  *iterations_out = host->iterations;
  return 0;
}
}  // namespace yakhip
