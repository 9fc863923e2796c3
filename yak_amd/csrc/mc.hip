// mc.hip -- marching cubes over the TSDF volume on gfx950.
// GPU counterpart of yak::marchingCubesCPU's core (yak/src/mc/marching_cubes.cpp:22-194: getOffset,
// cubeOffsets, processCube, the triple loop of makeMesh).  Citations are relative to /root/reference/.
//
// Same cubes, same case table, same vertex arithmetic, so the triangle SET equals the CPU result
// (the reference's own output order is nondeterministic: OpenMP + a critical section, :163-171).
// Schedule: classify+count -> exclusive scan -> classify+emit, both volume passes streaming:
//   * a wave owns a tile of 63 cubes along x (64 voxels: lane i holds the voxels at x0 + i, the cube
//     corners at x + 1 come from lane i + 1 by a cross-lane move, no second load) times 4 cube rows
//     (5 voxel rows: 1.25 dword loads per cube and layer, each a 256-byte row segment), and walks a
//     chunk of z layers keeping the previous layer in registers;
//   * the four waves of a workgroup stack their tiles in y, so of the 17 voxel rows a workgroup needs
//     only one is shared with the next workgroup (1.06x the volume from the fabric; a wave per row
//     pair measured 1.77x);
//   * the output position of every triangle is fixed by the scan (tile order, then z, row, x, then the
//     case table's order): the result is deterministic, no atomics.  The emit pass is latency-bound
//     (few waves hold surface, each walks its cubes serially), so lanes store their triangles
//     directly: staging a step's triangles in LDS for full-row stores measured slower.
#include "yakhip_internal.h"

#define YAK_MC_TABLE_DECL __constant__ const
#include "mc_case_table.h"

namespace yakhip
{
namespace
{
constexpr int kRun = 63;   // cubes per wave along x
constexpr int kRows = 4;   // cube rows per wave
constexpr size_t kScanBlock = 1024;  // tile counts scanned by one workgroup

struct McParams
{
  const uint32_t* vol;  // layer z0 of the slab
  int X, Y;
  int z0;               // global index of the first layer held
  int cube_layers;      // cube layers to process: global z in [z0, z0 + cube_layers)
  int nx, ny;           // wave tiles along x and y
  int z_chunk, nzc;     // cube layers a wave walks in the count pass, and how many such chunks
  int z_sub, nzs;       // cube layers per tile (what is counted, scanned and emitted as a unit); divides z_chunk
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

// Codes of one voxel row at one x: bits 0-1 own voxel, bits 2-3 the voxel at x + 1; `lo` = layer z,
// `hi` = layer z + 1.  Rows a (y) and b (y + 1) give the cube's 8 corners.
__device__ __forceinline__ unsigned cube_index(unsigned a_lo, unsigned b_lo, unsigned a_hi, unsigned b_hi, bool& valid)
{
  valid = ((a_lo & (a_lo >> 2) & b_lo & (b_lo >> 2) & a_hi & (a_hi >> 2) & b_hi & (b_hi >> 2)) & 2u) != 0;
  return ((b_lo >> 2) & 1u) | (((a_lo >> 2) & 1u) << 1) | ((a_lo & 1u) << 2) | ((b_lo & 1u) << 3) |
         (((b_hi >> 2) & 1u) << 4) | (((a_hi >> 2) & 1u) << 5) | ((a_hi & 1u) << 6) | ((b_hi & 1u) << 7);
}

template <bool EMIT>
__global__ __launch_bounds__(256) void mc_kernel(const McParams p, uint32_t* counts, const uint32_t* local_offsets,
                                                 const unsigned long long* block_bases, float* triangles)
{
  __shared__ uint64_t table[256];
  __shared__ float vals[EMIT ? 4 : 1][8][64];
  table[threadIdx.x] = yak_mc_case_table[threadIdx.x];
  __syncthreads();

  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  // the count pass walks z_chunk layers and closes a tile every z_sub layers; the emit pass takes one
  // tile per wave, so that the few waves that hold surface are short and many
  const int xt = (int)blockIdx.x, yt = (int)blockIdx.y * 4 + wave, zc = (int)blockIdx.z;
  if (yt >= p.ny)
    return;
  const int walk = EMIT ? p.z_sub : p.z_chunk;
  const size_t tile_stride = (size_t)p.ny * p.nx;
  size_t tile = ((size_t)zc * (size_t)(walk / p.z_sub) * p.ny + yt) * p.nx + xt;  // first tile of this wave
  unsigned long long out = 0;
  if (EMIT)
  {
    if (counts[tile] == 0)
      return;
    out = block_bases[tile / kScanBlock] + local_offsets[tile];
  }

  const int x = xt * kRun + lane;
  const int y0 = yt * kRows;
  const bool in_x = x < p.X;                         // this lane holds a voxel
  const bool has_cube = lane < kRun && x + 1 < p.X;  // ... and owns the cubes [x, x + 1]
  const size_t plane = (size_t)p.X * p.Y;
  const int zl0 = zc * walk, zl1 = min(zl0 + walk, p.cube_layers);  // local cube layers

  // voxel rows y0 .. y0 + 4 (clamped to the volume: a clamped row only feeds cube rows that do not exist)
  const uint32_t* rowp[kRows + 1];
#pragma unroll
  for (int r = 0; r <= kRows; ++r)
    rowp[r] = p.vol + (size_t)min(y0 + r, p.Y - 1) * p.X + (in_x ? x : 0);

  uint32_t w_lo[kRows + 1], n_lo[kRows + 1];  // layer z: own word, the word at x + 1
#pragma unroll
  for (int r = 0; r <= kRows; ++r)
  {
    w_lo[r] = in_x ? rowp[r][(size_t)zl0 * plane] : 0u;
    n_lo[r] = __shfl_down(w_lo[r], 1, 64);
  }
  unsigned n_mine = 0;

  for (int zl = zl0; zl < zl1; ++zl)
  {
    uint32_t w_hi[kRows + 1], n_hi[kRows + 1];
    unsigned c_lo[kRows + 1], c_hi[kRows + 1];
#pragma unroll
    for (int r = 0; r <= kRows; ++r)
      w_hi[r] = in_x ? rowp[r][(size_t)(zl + 1) * plane] : 0u;
#pragma unroll
    for (int r = 0; r <= kRows; ++r)
    {
      n_hi[r] = __shfl_down(w_hi[r], 1, 64);
      c_lo[r] = voxel_code(w_lo[r], p.min_weight) | (voxel_code(n_lo[r], p.min_weight) << 2);
      c_hi[r] = voxel_code(w_hi[r], p.min_weight) | (voxel_code(n_hi[r], p.min_weight) << 2);
    }

#pragma unroll
    for (int r = 0; r < kRows; ++r)
    {
      bool valid;
      const unsigned ci = cube_index(c_lo[r], c_lo[r + 1], c_hi[r], c_hi[r + 1], valid);
      const bool exists = has_cube && y0 + r + 1 < p.Y;
      const uint64_t entry = (exists && valid) ? table[ci] : 0;
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
          vals[wave][0][lane] = 5.0f * half_bits_to_float(n_lo[r + 1] & 0xffffu);
          vals[wave][1][lane] = 5.0f * half_bits_to_float(n_lo[r] & 0xffffu);
          vals[wave][2][lane] = 5.0f * half_bits_to_float(w_lo[r] & 0xffffu);
          vals[wave][3][lane] = 5.0f * half_bits_to_float(w_lo[r + 1] & 0xffffu);
          vals[wave][4][lane] = 5.0f * half_bits_to_float(n_hi[r + 1] & 0xffffu);
          vals[wave][5][lane] = 5.0f * half_bits_to_float(n_hi[r] & 0xffffu);
          vals[wave][6][lane] = 5.0f * half_bits_to_float(w_hi[r] & 0xffffu);
          vals[wave][7][lane] = 5.0f * half_bits_to_float(w_hi[r + 1] & 0xffffu);
          float* dst = triangles + (out + (incl - nt)) * 9ull;
          const int gy = y0 + r, gz = p.z0 + zl;
          for (unsigned k = 0; k < 3 * nt; ++k)
          {
            const unsigned e = (unsigned)(entry >> (4 * k)) & 15u;
            const unsigned c0 = (unsigned)(YAK_MC_EDGE_A >> (4 * e)) & 15u, c1 = (unsigned)(YAK_MC_EDGE_B >> (4 * e)) & 15u;
            const float v0 = vals[wave][c0][lane], v1 = vals[wave][c1][lane];
            // getOffset, :22-31: the double quotient of two floats rounded to float IS the fp32 quotient
            // (53 >= 2 * 24 + 2), so the correctly rounded fp32 '/' gives the same value
            const float delta = v1 - v0;
            const float offset = delta == 0.f ? 0.5f : (0.0f - v0) / delta;
            // :121-129  edge = v0_pos + offset * (v1_pos - v0_pos); the product is exact (factor 0 or +-1)
            const float p0x = (float)(x + (int)((kDxMask >> c0) & 1u)), p1x = (float)(x + (int)((kDxMask >> c1) & 1u));
            const float p0y = (float)(gy + (int)((kDyMask >> c0) & 1u)), p1y = (float)(gy + (int)((kDyMask >> c1) & 1u));
            const float p0z = (float)(gz + (int)((kDzMask >> c0) & 1u)), p1z = (float)(gz + (int)((kDzMask >> c1) & 1u));
            // :175-183  s * double(v), narrowed to float when the mesh cloud is filled
            dst[3 * k + 0] = (float)(p.scale * (double)(p0x + offset * (p1x - p0x)));
            dst[3 * k + 1] = (float)(p.scale * (double)(p0y + offset * (p1y - p0y)));
            dst[3 * k + 2] = (float)(p.scale * (double)(p0z + offset * (p1z - p0z)));
          }
        }
        out += total;
      }
    }
#pragma unroll
    for (int r = 0; r <= kRows; ++r)
    {
      w_lo[r] = w_hi[r];
      n_lo[r] = n_hi[r];
    }
    if (!EMIT && ((zl + 1 - zl0) % p.z_sub == 0 || zl + 1 == zl1))  // wave-uniform: a tile is complete
    {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1)
        n_mine += __shfl_xor(n_mine, off, 64);
      if (lane == 0)
        counts[tile] = n_mine;
      tile += tile_stride;
      n_mine = 0;
    }
  }
}

// Exclusive scan of the tile counts in two small launches: every workgroup scans kScanBlock counts
// (4 per thread, one 16-byte load) into offsets local to the block and publishes the block's sum; one
// workgroup then scans the block sums.  The emit pass adds the two.
__global__ __launch_bounds__(256) void mc_scan_local_kernel(const uint32_t* counts, uint32_t* local_offsets, size_t n,
                                                            unsigned long long* block_sums)
{
  __shared__ uint32_t wave_sum[4];
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  const size_t base = (size_t)blockIdx.x * kScanBlock + (size_t)threadIdx.x * 4;
  uint32_t c[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    c[i] = base + i < n ? counts[base + i] : 0u;
  const uint32_t mine = c[0] + c[1] + c[2] + c[3];
  uint32_t incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1)
  {
    const uint32_t up = __shfl_up(incl, off, 64);
    if (lane >= off)
      incl += up;
  }
  if (lane == 63)
    wave_sum[wave] = incl;
  __syncthreads();
  uint32_t before = incl - mine;
  for (int w = 0; w < wave; ++w)
    before += wave_sum[w];
#pragma unroll
  for (int i = 0; i < 4; ++i)
  {
    if (base + i < n)
      local_offsets[base + i] = before;
    before += c[i];
  }
  if (threadIdx.x == 255)
    block_sums[blockIdx.x] = before;
}

__global__ __launch_bounds__(1024) void mc_scan_blocks_kernel(unsigned long long* block_sums, size_t n_blocks,
                                                              unsigned long long* total)
{
  __shared__ unsigned long long wave_sum[16];
  __shared__ unsigned long long carry;
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  if (threadIdx.x == 0)
    carry = 0;
  __syncthreads();
  for (size_t base = 0; base < n_blocks; base += 1024)  // one trip up to a million tiles
  {
    const size_t i = base + threadIdx.x;
    const unsigned long long c = i < n_blocks ? block_sums[i] : 0;
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
    if (i < n_blocks)
      block_sums[i] = before + incl - c;  // in place: the sums become exclusive bases
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
  p.ny = (Y - 1 + kRows - 1) / kRows;
  // tiles of 4 cube layers, walked 4 at a time by the count pass (measured at 512^3: the count pass wants
  // long walks, 0.18 ms at 32 layers vs 0.28 at 8; the emit pass wants many short waves, 0.21 vs 0.53 ms)
  p.z_sub = ctx->mc_z_chunk > 0 ? ctx->mc_z_chunk : 4;
  while ((p.cube_layers + p.z_sub - 1) / p.z_sub > 65535)
    p.z_sub *= 2;
  p.z_chunk = 4 * p.z_sub;
  p.nzs = (p.cube_layers + p.z_sub - 1) / p.z_sub;
  p.nzc = (p.cube_layers + p.z_chunk - 1) / p.z_chunk;
  p.min_weight = min_weight;
  p.scale = scale;
  const unsigned gy = (unsigned)((p.ny + 3) / 4);
  if (gy > 65535u)
    return fail(YAKHIP_ERANGE, "marching_cubes: more than a million voxel rows");
  const size_t n_tiles = (size_t)p.nx * p.ny * p.nzs;

  // context-owned scratch: counts (u32) | local offsets (u32) | block bases (u64) | total (u64), and a
  // pinned host word
  const size_t n_blocks = (n_tiles + kScanBlock - 1) / kScanBlock;
  const size_t counts_bytes = (n_tiles * sizeof(uint32_t) + 15) & ~(size_t)15;
  const size_t need = 2 * counts_bytes + (n_blocks + 1) * sizeof(unsigned long long);
  if (ctx->mc_scratch_bytes < need)
  {
    if (ctx->mc_scratch)
    {
      YAKHIP_TRY(hipStreamSynchronize(ctx->stream));
      YAKHIP_TRY(hipFree(ctx->mc_scratch));
    }
    ctx->mc_scratch = nullptr;
    ctx->mc_scratch_bytes = 0;
    YAKHIP_TRY(hipMalloc(&ctx->mc_scratch, need));
    ctx->mc_scratch_bytes = need;
  }
  if (!ctx->mc_pinned)
    YAKHIP_TRY(hipHostMalloc(&ctx->mc_pinned, sizeof(unsigned long long), hipHostMallocDefault));
  uint32_t* counts = (uint32_t*)ctx->mc_scratch;
  uint32_t* local_offsets = (uint32_t*)((uint8_t*)ctx->mc_scratch + counts_bytes);
  unsigned long long* block_bases = (unsigned long long*)((uint8_t*)ctx->mc_scratch + 2 * counts_bytes);
  unsigned long long* total = block_bases + n_blocks;

  const dim3 grid((unsigned)p.nx, gy, (unsigned)p.nzc), grid_emit((unsigned)p.nx, gy, (unsigned)p.nzs);
  std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
  int rc = timing_begin(ctx, &ev);  // one event pair around count + scan (+ the emit pass below gets its own)
  if (rc != 0)
    return rc;
  hipLaunchKernelGGL((mc_kernel<false>), grid, dim3(256), 0, ctx->stream, p, counts, local_offsets, block_bases,
                     (float*)nullptr);
  YAKHIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(mc_scan_local_kernel, dim3((unsigned)n_blocks), dim3(256), 0, ctx->stream, counts, local_offsets,
                     n_tiles, block_bases);
  hipLaunchKernelGGL(mc_scan_blocks_kernel, dim3(1), dim3(1024), 0, ctx->stream, block_bases, n_blocks, total);
  if ((rc = timing_end(ctx, ev)) != 0)
    return rc;
  YAKHIP_TRY(hipGetLastError());
  YAKHIP_TRY(hipMemcpyAsync(ctx->mc_pinned, total, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  YAKHIP_TRY(hipStreamSynchronize(ctx->stream));
  const unsigned long long n = *(const unsigned long long*)ctx->mc_pinned;
  *n_out = n;
  if (!triangles || n == 0 || n > capacity)
    return 0;  // count only (or the caller's buffer is too small: it sees n > capacity)

  if ((rc = timing_begin(ctx, &ev)) != 0)
    return rc;
  hipLaunchKernelGGL((mc_kernel<true>), grid_emit, dim3(256), 0, ctx->stream, p, counts, local_offsets, block_bases,
                     triangles);
  if ((rc = timing_end(ctx, ev)) != 0)
    return rc;
  YAKHIP_TRY(hipGetLastError());
  return check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}
}  // namespace yakhip
