// mc.hip -- marching cubes over the TSDF volume on gfx950.
// GPU counterpart of yak::marchingCubesCPU's core (yak/src/mc/marching_cubes.cpp:22-194: getOffset,
// cubeOffsets, processCube, the triple loop of makeMesh).  Citations are relative to /root/reference/.
//
// Same cubes, same case table, same vertex arithmetic, so the triangle SET equals the CPU result
// (the reference's own output order is nondeterministic: OpenMP + a critical section, :163-171).
// Schedule: classify+count -> exclusive scan -> classify+emit, both volume passes streaming:
//   * a wave owns a run of 63 cubes along x (64 voxels: lane i holds the voxels at x0 + i, the cube
//     corners at x + 1 come from lane i + 1 by a cross-lane move, no second load) on one cube row,
//     and walks a chunk of z layers keeping the previous layer in registers: 2 dword loads per lane
//     and layer, both 256-byte row segments;
//   * the four waves of a workgroup take four consecutive rows, so the row shared by neighbours is
//     served by the CU's L1;
//   * the output position of every triangle is fixed by the scan (tile order, then z, then x, then the
//     case table's order): the result is deterministic, no atomics.
#include "yakhip_internal.h"

#define YAK_MC_TABLE_DECL __constant__ const
#include "mc_case_table.h"

namespace yakhip
{
namespace
{
constexpr int kRun = 63;  // cubes per wave along x

struct McParams
{
  const uint32_t* vol;  // layer z0 of the slab
  int X, Y;
  int z0;               // global index of the first layer held
  int cube_layers;      // cube layers to process: global z in [z0, z0 + cube_layers)
  int nx, ny;           // wave tiles along x, cube rows
  int z_chunk, nzc;
  int min_weight;
  double scale;
};

// corner c of the cube sits at (x + dx, y + dy, z + dz): marching_cubes.cpp:33-37
constexpr unsigned kDxMask = 0x33u, kDyMask = 0x99u, kDzMask = 0xF0u;

__device__ __forceinline__ unsigned voxel_code(uint32_t word, int min_weight)
{
  // bit 0: 5 * tsdf < 0 (marching_cubes.cpp:46, 81-96); bit 1: weight >= min_weight (:51-74, u16 weight)
  const float v = half_bits_to_float(word & 0xffffu);
  return (v < 0.f ? 1u : 0u) | ((int)(word >> 16) >= min_weight ? 2u : 0u);
}

// One z step of one wave: from the codes of the four own-x corners and the four x+1 corners.
__device__ __forceinline__ unsigned cube_index(unsigned a0, unsigned b0, unsigned a1, unsigned b1, unsigned a0n,
                                               unsigned b0n, unsigned a1n, unsigned b1n, bool& valid)
{
  valid = ((a0 & b0 & a1 & b1 & a0n & b0n & a1n & b1n) & 2u) != 0;
  return (b0n & 1u) | ((a0n & 1u) << 1) | ((a0 & 1u) << 2) | ((b0 & 1u) << 3) | ((b1n & 1u) << 4) | ((a1n & 1u) << 5) |
         ((a1 & 1u) << 6) | ((b1 & 1u) << 7);
}

template <bool EMIT>
__global__ __launch_bounds__(256) void mc_kernel(const McParams p, uint32_t* counts, const unsigned long long* offsets,
                                                 float* triangles)
{
  __shared__ uint64_t table[256];
  __shared__ float vals[4][8][64];
  table[threadIdx.x] = yak_mc_case_table[threadIdx.x];
  __syncthreads();

  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  const int xt = (int)blockIdx.x, y = (int)blockIdx.y * 4 + wave, zc = (int)blockIdx.z;
  if (y >= p.ny)
    return;
  const size_t tile = ((size_t)zc * p.ny + y) * p.nx + xt;
  unsigned long long out = 0;
  if (EMIT)
  {
    if (counts[tile] == 0)
      return;
    out = offsets[tile];
  }

  const int x = xt * kRun + lane;
  const bool in_x = x < p.X;                         // this lane holds a voxel
  const bool has_cube = lane < kRun && x + 1 < p.X;  // ... and owns the cube [x, x + 1]
  const size_t plane = (size_t)p.X * p.Y;
  const uint32_t* row_a = p.vol + (size_t)y * p.X + (in_x ? x : 0);
  const uint32_t* row_b = row_a + p.X;
  const int zl0 = zc * p.z_chunk, zl1 = min(zl0 + p.z_chunk, p.cube_layers);  // local cube layers

  uint32_t wa0 = 0, wb0 = 0;
  if (in_x)
  {
    wa0 = row_a[(size_t)zl0 * plane];
    wb0 = row_b[(size_t)zl0 * plane];
  }
  uint32_t wa0n = __shfl_down(wa0, 1, 64), wb0n = __shfl_down(wb0, 1, 64);
  unsigned n_mine = 0;

  for (int zl = zl0; zl < zl1; ++zl)
  {
    uint32_t wa1 = 0, wb1 = 0;
    if (in_x)
    {
      wa1 = row_a[(size_t)(zl + 1) * plane];
      wb1 = row_b[(size_t)(zl + 1) * plane];
    }
    const uint32_t wa1n = __shfl_down(wa1, 1, 64), wb1n = __shfl_down(wb1, 1, 64);

    bool valid;
    const unsigned ci = cube_index(voxel_code(wa0, p.min_weight), voxel_code(wb0, p.min_weight),
                                   voxel_code(wa1, p.min_weight), voxel_code(wb1, p.min_weight),
                                   voxel_code(wa0n, p.min_weight), voxel_code(wb0n, p.min_weight),
                                   voxel_code(wa1n, p.min_weight), voxel_code(wb1n, p.min_weight), valid);
    const uint64_t entry = (has_cube && valid) ? table[ci] : 0;
    const unsigned nt = (unsigned)(entry >> 60);

    if (!EMIT)
      n_mine += nt;
    else if (__builtin_amdgcn_ballot_w64(nt != 0) != 0)
    {
      // exclusive scan of nt over the wave: where this lane's triangles go
      unsigned incl = nt;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1)
      {
        const unsigned up = __shfl_up(incl, off, 64);
        if (lane >= off)
          incl += up;
      }
      const unsigned total = __shfl(incl, 63, 64);
      if (nt != 0)
      {
        // corner values, marching_cubes.cpp:43-74: val = 5.0f * float(half)
        vals[wave][0][lane] = 5.0f * half_bits_to_float(wb0n & 0xffffu);
        vals[wave][1][lane] = 5.0f * half_bits_to_float(wa0n & 0xffffu);
        vals[wave][2][lane] = 5.0f * half_bits_to_float(wa0 & 0xffffu);
        vals[wave][3][lane] = 5.0f * half_bits_to_float(wb0 & 0xffffu);
        vals[wave][4][lane] = 5.0f * half_bits_to_float(wb1n & 0xffffu);
        vals[wave][5][lane] = 5.0f * half_bits_to_float(wa1n & 0xffffu);
        vals[wave][6][lane] = 5.0f * half_bits_to_float(wa1 & 0xffffu);
        vals[wave][7][lane] = 5.0f * half_bits_to_float(wb1 & 0xffffu);
        float* dst = triangles + (out + (incl - nt)) * 9ull;
        const int gz = p.z0 + zl;
        for (unsigned k = 0; k < 3 * nt; ++k)
        {
          const unsigned e = (unsigned)(entry >> (4 * k)) & 15u;
          const unsigned c0 = yak_mc_edge_corners[e][0], c1 = yak_mc_edge_corners[e][1];
          const float v0 = vals[wave][c0][lane], v1 = vals[wave][c1][lane];
          // getOffset, :22-31: the double quotient of two floats rounded to float IS the fp32 quotient
          // (53 >= 2 * 24 + 2), so the correctly rounded fp32 '/' gives the same value
          const float delta = v1 - v0;
          const float offset = delta == 0.f ? 0.5f : (0.0f - v0) / delta;
          // :121-129  edge = v0_pos + offset * (v1_pos - v0_pos); the product is exact (factor 0 or +-1)
          const float p0x = (float)(x + (int)((kDxMask >> c0) & 1u)), p1x = (float)(x + (int)((kDxMask >> c1) & 1u));
          const float p0y = (float)(y + (int)((kDyMask >> c0) & 1u)), p1y = (float)(y + (int)((kDyMask >> c1) & 1u));
          const float p0z = (float)(gz + (int)((kDzMask >> c0) & 1u)), p1z = (float)(gz + (int)((kDzMask >> c1) & 1u));
          // :175-183  s * double(v), narrowed to float when the mesh cloud is filled
          dst[3 * k + 0] = (float)(p.scale * (double)(p0x + offset * (p1x - p0x)));
          dst[3 * k + 1] = (float)(p.scale * (double)(p0y + offset * (p1y - p0y)));
          dst[3 * k + 2] = (float)(p.scale * (double)(p0z + offset * (p1z - p0z)));
        }
      }
      out += total;
    }
    wa0 = wa1, wb0 = wb1, wa0n = wa1n, wb0n = wb1n;
  }

  if (!EMIT)
  {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
      n_mine += __shfl_xor(n_mine, off, 64);
    if (lane == 0)
      counts[tile] = n_mine;
  }
}

// Exclusive scan of the tile counts (one workgroup; the tile count is small next to the volume).
__global__ __launch_bounds__(1024) void mc_scan_kernel(const uint32_t* counts, unsigned long long* offsets, size_t n,
                                                       unsigned long long* total)
{
  __shared__ unsigned long long wave_sum[16];
  __shared__ unsigned long long carry;
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  if (threadIdx.x == 0)
    carry = 0;
  __syncthreads();
  for (size_t base = 0; base < n; base += 1024)
  {
    const size_t i = base + threadIdx.x;
    const unsigned long long c = i < n ? counts[i] : 0;
    unsigned long long incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
    {
      const unsigned long long up = __shfl_up(incl, off, 64);
      if (lane >= off)
        incl += up;
    }
    if (lane == 63)
      wave_sum[wave] = incl;
    __syncthreads();
    unsigned long long before = carry;
    for (int w = 0; w < wave; ++w)
      before += wave_sum[w];
    if (i < n)
      offsets[i] = before + incl - c;
    __syncthreads();
    if (threadIdx.x == 1023)
      carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0)
    *total = carry;
}
}  // namespace

int launch_marching_cubes(yakhip_ctx* ctx, const uint32_t* vol, int X, int Y, int z0, int layers, int min_weight,
                          double scale, float* triangles, unsigned long long capacity, unsigned long long* n_out)
{
  *n_out = 0;
  if (X < 2 || Y < 2 || layers < 2)
    return 0;  // no cube fits (marching_cubes.cpp:164-168: the loops do not run)
  McParams p;
  p.vol = vol;
  p.X = X, p.Y = Y, p.z0 = z0;
  p.cube_layers = layers - 1;
  p.nx = (X - 1 + kRun - 1) / kRun;
  p.ny = Y - 1;
  p.z_chunk = 32;
  p.nzc = (p.cube_layers + p.z_chunk - 1) / p.z_chunk;
  if (p.nzc > 65535)
  {
    p.z_chunk = (p.cube_layers + 65534) / 65535;
    p.nzc = (p.cube_layers + p.z_chunk - 1) / p.z_chunk;
  }
  p.min_weight = min_weight;
  p.scale = scale;
  const unsigned gy = (unsigned)((p.ny + 3) / 4);
  if (gy > 65535u)
    return fail(YAKHIP_ERANGE, "marching_cubes: more than 262140 voxel rows");
  const size_t n_tiles = (size_t)p.nx * p.ny * p.nzc;

  // context-owned scratch: counts (u32) | offsets (u64) | total (u64), and a pinned host word
  const size_t counts_bytes = (n_tiles * sizeof(uint32_t) + 7) & ~(size_t)7;
  const size_t need = counts_bytes + (n_tiles + 1) * sizeof(unsigned long long);
  if (ctx->mc_scratch_bytes < need)
  {
    if (ctx->mc_scratch)
      YAKHIP_TRY(hipFree(ctx->mc_scratch));
    ctx->mc_scratch = nullptr;
    ctx->mc_scratch_bytes = 0;
    YAKHIP_TRY(hipMalloc(&ctx->mc_scratch, need));
    ctx->mc_scratch_bytes = need;
  }
  if (!ctx->mc_pinned)
    YAKHIP_TRY(hipHostMalloc(&ctx->mc_pinned, sizeof(unsigned long long), hipHostMallocDefault));
  uint32_t* counts = (uint32_t*)ctx->mc_scratch;
  unsigned long long* offsets = (unsigned long long*)((uint8_t*)ctx->mc_scratch + counts_bytes);
  unsigned long long* total = offsets + n_tiles;

  const dim3 grid((unsigned)p.nx, gy, (unsigned)p.nzc);
  std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
  int rc = timing_begin(ctx, &ev);
  if (rc != 0)
    return rc;
  hipLaunchKernelGGL((mc_kernel<false>), grid, dim3(256), 0, ctx->stream, p, counts, offsets, (float*)nullptr);
  if ((rc = timing_end(ctx, ev)) != 0)
    return rc;
  YAKHIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(mc_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, counts, offsets, n_tiles, total);
  YAKHIP_TRY(hipGetLastError());
  YAKHIP_TRY(hipMemcpyAsync(ctx->mc_pinned, total, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  YAKHIP_TRY(hipStreamSynchronize(ctx->stream));
  const unsigned long long n = *(const unsigned long long*)ctx->mc_pinned;
  *n_out = n;
  if (!triangles || n == 0 || n > capacity)
    return 0;  // count only (or the caller's buffer is too small: it sees n > capacity)

  if ((rc = timing_begin(ctx, &ev)) != 0)
    return rc;
  hipLaunchKernelGGL((mc_kernel<true>), grid, dim3(256), 0, ctx->stream, p, counts, offsets, triangles);
  if ((rc = timing_end(ctx, ev)) != 0)
    return rc;
  YAKHIP_TRY(hipGetLastError());
  return check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}
}  // namespace yakhip
