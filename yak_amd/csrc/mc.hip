// mc.hip -- marching cubes over the TSDF volume on gfx950.
// GPU counterpart of yak::marchingCubesCPU's core (yak/src/mc/marching_cubes.cpp:22-194: getOffset,
// cubeOffsets, processCube, the triple loop of makeMesh).  Citations are relative to /root/reference/.
//
// Same cubes, same case table, same vertex arithmetic, so the triangle SET equals the CPU result
// (the reference's own output order is nondeterministic: OpenMP + a critical section, :163-171).
// Schedule: classify+count -> exclusive scan -> classify+emit, both volume passes streaming:
//   * a wave owns a tile of 63 cubes along x (64 voxels: lane i holds the voxels at x0 + i, the cube
//     corners at x + 1 come from lane i + 1) times 4 cube rows (5 voxel rows: 1.25 dword loads per cube
//     and layer, each a 256-byte row segment), and walks a chunk of z layers keeping the previous layer
//     in registers and the next one in flight;
//   * the four waves of a workgroup stack their tiles in y, and in the count pass every XCD gets a
//     contiguous run of workgroups, so that the 128-byte lines and voxel rows that neighbouring tiles
//     share meet in one L2: 1.12x the volume from the fabric (FETCH_SIZE; round-robin workgroups 1.66x);
//   * classification is done on wave masks and starts with the question "is any voxel of these two
//     layers negative": free and unseen space, most of the volume, costs five compares a layer.  The
//     count pass is bound by memory (0.11 ms for 512 MiB; a plain streaming read takes 0.08);
//   * the output position of every triangle is fixed by the scan (tile order, then the order in which the
//     tile's cubes are queued: z, row, x, then the case table's order): deterministic, no atomics;
//   * the emit pass takes the tiles that hold triangles, queues their surface cubes in LDS and works
//     them off 64 at a time, a cube per lane and a triangle per step, whatever rows and layers the
//     cubes came from (row by row a curved surface kept 4 of 64 lanes busy: 0.19 ms against 0.09).
#include "yakhip_internal.h"
#include <algorithm>

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
  // surface map of the slab `vol` is part of (yakhip_volume.surface_map), or nullptr: bit (bx + map_x * (by + map_y * bz))
  // set = the brick of 16^3 voxels may hold a negative voxel; vol's first layer is layer map_z0 of that slab
  const uint32_t* map;
  int map_x, map_y, map_z0;
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

// EMIT: launched over the tiles that hold triangles only -- wave k takes list entry k = {tile, first slot}
// (mc_compact_kernel), n_live of them; the count pass over every tile of the slab.
template <bool EMIT>
__global__ __launch_bounds__(256) void mc_kernel(const McParams p, uint32_t* counts, const unsigned long long* list,
                                                 unsigned long long n_live, float* triangles)
{
  // The case table in LDS.  The emit pass keeps a copy per wave and fills it only after the wave knows that
  // its tile holds triangles: nearly all of its waves leave at once, and a workgroup barrier behind a
  // table load would keep each of them resident for two memory round trips.
  __shared__ uint64_t table_lds[EMIT ? 4 : 1][256];
  // emit pass: per wave, a ring of cubes that hold triangles, field-major (4 words of corner halves, the
  // case entry, the cube's place in the tile)
  __shared__ uint32_t queue[EMIT ? 4 : 1][7][EMIT ? 128 : 1];
  if (!EMIT)
  {
    table_lds[0][threadIdx.x] = yak_mc_case_table[threadIdx.x];
    __syncthreads();
  }

  // (the wave index through readfirstlane: the compiler cannot see that it is uniform over the wave, and
  // everything derived from it -- the tile, the row tests, the surface masks -- should live in scalars)
  const int lane = (int)threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  // the count pass walks z_chunk layers and closes a tile every z_sub layers; the emit pass takes one
  // tile per wave, so that the few waves that hold surface are short and many
  // Workgroups go to the 8 XCDs round-robin and each XCD has its own L2.  A 63-cube tile row is 256
  // bytes at a 252-byte pitch, so x-neighbours share a 128-byte line (and y-neighbours a voxel row):
  // with neighbours on different XCDs the count pass fetched 1.66x the volume (FETCH_SIZE).  Give each
  // XCD a contiguous run of tiles instead, x fastest, so that the shared lines meet in one L2.
  const unsigned n_wg = gridDim.x, per_xcd = n_wg / 8u;
  // (count pass only: the emit pass's work sits in the few tiles that hold surface, and round-robin
  // spreads those over the XCDs; contiguous runs measured 0.31 ms against 0.21)
  const unsigned wg =
      !EMIT && blockIdx.x < per_xcd * 8u ? (blockIdx.x % 8u) * per_xcd + blockIdx.x / 8u : blockIdx.x;
  const unsigned gy = (unsigned)(p.ny + 3) / 4u;
  const int walk = EMIT ? p.z_sub : p.z_chunk;
  const size_t tile_stride = (size_t)p.ny * p.nx;
  int xt, yt, zc;
  size_t tile;  // first tile of this wave
  unsigned long long out = 0;
  if (EMIT)
  {
    const unsigned long long k = (unsigned long long)blockIdx.x * 4ull + (unsigned long long)wave;
    if (k >= n_live)
      return;
    tile = (size_t)list[2 * k];
    out = list[2 * k + 1];
    xt = (int)(tile % (size_t)p.nx), yt = (int)((tile / (size_t)p.nx) % (size_t)p.ny), zc = (int)(tile / tile_stride);
  }
  else
  {
    xt = (int)(wg % (unsigned)p.nx), yt = (int)((wg / (unsigned)p.nx) % gy) * 4 + wave, zc = (int)(wg / ((unsigned)p.nx * gy));
    if (yt >= p.ny)
      return;
    tile = ((size_t)zc * (size_t)(walk / p.z_sub) * p.ny + yt) * p.nx + xt;
  }
  if (!EMIT && p.map)
  {
    // A cube has triangles only if one of its corners is negative (cases 0 and 255 have none), and a brick whose
    // bit is clear holds no negative voxel: if none of the <= 5 x 2 x 3 bricks that this wave's voxels lie in is
    // marked, its tiles are empty and nothing is read.
    const int vx0 = xt * kRun, vx1 = min(vx0 + kRun, p.X - 1), vy0 = yt * kRows, vy1 = min(vy0 + kRows, p.Y - 1);
    const int vz0 = zc * walk, vz1 = min(vz0 + walk, p.cube_layers);  // voxel layers vz0 .. vz1 (cube layer z uses z and z + 1)
    const int bx0 = vx0 >> 4, nbx = (vx1 >> 4) - bx0 + 1, by0 = vy0 >> 4, nby = (vy1 >> 4) - by0 + 1;
    const int bz0 = (vz0 + p.map_z0) >> 4, nbz = ((vz1 + p.map_z0) >> 4) - bz0 + 1;
    bool marked = false;
    for (int i = lane; i < nbx * nby * nbz; i += 64)
    {
      const int ix = i % nbx, iy = (i / nbx) % nby, iz = i / (nbx * nby);
      const uint32_t b = (uint32_t)(bx0 + ix) + (uint32_t)p.map_x * ((uint32_t)(by0 + iy) + (uint32_t)p.map_y * (uint32_t)(bz0 + iz));
      marked = marked || ((p.map[b >> 5] >> (b & 31u)) & 1u);
    }
    if (__builtin_amdgcn_ballot_w64(marked) == 0)
    {
      const int n_sub = (vz1 - vz0 + p.z_sub - 1) / p.z_sub;
      if (lane < n_sub)
        counts[tile + (size_t)lane * tile_stride] = 0;
      return;
    }
  }
  if (EMIT)
  {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      table_lds[wave][lane + 64 * i] = yak_mc_case_table[lane + 64 * i];
    // (read back by this wave only: LDS operations of one wave complete in order)
  }
  const uint64_t* table = table_lds[EMIT ? wave : 0];

  const int x = xt * kRun + lane;
  const int y0 = yt * kRows;
  const bool in_x = x < p.X;                         // this lane holds a voxel
  const bool has_cube = lane < kRun && x + 1 < p.X;  // ... and owns the cubes [x, x + 1]
  const size_t plane = (size_t)p.X * p.Y;
  const int zl0 = zc * walk, zl1 = min(zl0 + walk, p.cube_layers);  // local cube layers

  // voxel rows y0 .. y0 + 4 (clamped to the volume: a clamped row only feeds cube rows that do not exist),
  // as dword offsets within a layer; the layer base is a scalar
  uint32_t row_off[kRows + 1];
#pragma unroll
  for (int r = 0; r <= kRows; ++r)
    row_off[r] = (uint32_t)min(y0 + r, p.Y - 1) * (uint32_t)p.X + (uint32_t)(in_x ? x : 0);  // past the row: voxel 0

  // Classification runs on wave masks (one bit per lane, scalar registers).  The CU's scalar unit issues
  // no faster than a SIMD's vector unit, so masks are not free either, and most of the volume is free or
  // unseen space: a tile layer costs five compares (which voxels are negative) and a few scalar
  // operations unless one of its two voxel layers holds a negative voxel.  Only then are the masks
  // combined per cube (all 8 corners valid and of mixed sign: cases 0 and 255 and invalid cubes have no
  // triangle), and only the cube rows that hold surface go on to the per-lane case lookup.
  const uint64_t cube_lanes = __builtin_amdgcn_ballot_w64(has_cube);
  const uint64_t voxel_lanes = __builtin_amdgcn_ballot_w64(in_x);
  // weight test of voxel_code as one unsigned compare of the whole word
  const uint32_t weight_thr = p.min_weight <= 0 ? 0u : (uint32_t)p.min_weight << 16;
  uint64_t row_lanes[kRows + 1];  // lanes whose voxel in row r exists and can be valid at all
#pragma unroll
  for (int r = 0; r <= kRows; ++r)
    row_lanes[r] = (y0 + r < p.Y && p.min_weight <= 65535) ? voxel_lanes : 0;
  struct Masks
  {
    uint64_t neg[kRows + 1], val[kRows + 1];
  };
  // (lanes past the row and rows past the volume hold some other voxel: never valid, sign ignored)
  auto negative = [&](uint32_t w) { return __builtin_amdgcn_ballot_w64(half_bits_to_float(w & 0xffffu) < 0.f); };
  auto any_negative = [&](const uint32_t(&w)[kRows + 1]) {
    uint64_t any = 0;
#pragma unroll
    for (int r = 0; r <= kRows; ++r)
      any |= negative(w[r]);
    return any != 0;
  };
  auto classify = [&](const uint32_t(&w)[kRows + 1]) {
    Masks m;
#pragma unroll
    for (int r = 0; r <= kRows; ++r)
    {
      m.neg[r] = negative(w[r]);
      m.val[r] = __builtin_amdgcn_ballot_w64(w[r] >= weight_thr) & row_lanes[r];
    }
    return m;
  };

  const uint32_t* layer = p.vol + (size_t)zl0 * plane;  // wave-uniform
  uint32_t w_lo[kRows + 1];                             // layer z
#pragma unroll
  for (int r = 0; r <= kRows; ++r)
    w_lo[r] = layer[row_off[r]];
  // the loads run one layer ahead of the classification (two layers in flight per wave): with the
  // branches below the compiler does not move them up by itself
  layer += plane;
  uint32_t w_ahead[kRows + 1];
#pragma unroll
  for (int r = 0; r <= kRows; ++r)
    w_ahead[r] = layer[row_off[r]];
  bool lo_negative = any_negative(w_lo);
  unsigned n_mine = 0;

  unsigned q_head = 0, q_count = 0;  // wave-uniform
  auto drain = [&](unsigned n) {
    __builtin_amdgcn_wave_barrier();
    const unsigned slot = (q_head + (unsigned)lane) & 127u;
    const bool mine = (unsigned)lane < n;
    const uint64_t entry = mine ? ((uint64_t)queue[wave][5][slot] << 32) | queue[wave][4][slot] : 0;
    const unsigned nt = (unsigned)(entry >> 60);
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
      const unsigned place = queue[wave][6][slot];
      const int cx = xt * kRun + (int)(place & 63u), gy = y0 + (int)((place >> 6) & 3u), gz = p.z0 + zl0 + (int)(place >> 8);
      auto vertex = [&](unsigned e, float(&v)[3]) {
        const unsigned c0 = (unsigned)(YAK_MC_EDGE_A >> (4 * e)) & 15u, c1 = (unsigned)(YAK_MC_EDGE_B >> (4 * e)) & 15u;
        // corner values, marching_cubes.cpp:43-74: val = 5.0f * float(half)
        const float v0 = 5.0f * half_bits_to_float((queue[wave][c0 >> 1][slot] >> (16u * (c0 & 1u))) & 0xffffu);
        const float v1 = 5.0f * half_bits_to_float((queue[wave][c1 >> 1][slot] >> (16u * (c1 & 1u))) & 0xffffu);
        // getOffset, :22-31: the double quotient of two floats rounded to float IS the fp32 quotient
        // (53 >= 2 * 24 + 2), so the correctly rounded fp32 '/' gives the same value
        const float delta = v1 - v0;
        const float offset = delta == 0.f ? 0.5f : (0.0f - v0) / delta;
        // :121-129  edge = v0_pos + offset * (v1_pos - v0_pos); the product is exact (factor 0 or +-1)
        const float p0x = (float)(cx + (int)((kDxMask >> c0) & 1u)), p1x = (float)(cx + (int)((kDxMask >> c1) & 1u));
        const float p0y = (float)(gy + (int)((kDyMask >> c0) & 1u)), p1y = (float)(gy + (int)((kDyMask >> c1) & 1u));
        const float p0z = (float)(gz + (int)((kDzMask >> c0) & 1u)), p1z = (float)(gz + (int)((kDzMask >> c1) & 1u));
        // :175-183  s * double(v), narrowed to float when the mesh cloud is filled
        v[0] = (float)(p.scale * (double)(p0x + offset * (p1x - p0x)));
        v[1] = (float)(p.scale * (double)(p0y + offset * (p1y - p0y)));
        v[2] = (float)(p.scale * (double)(p0z + offset * (p1z - p0z)));
      };
      // a triangle per step: its three vertices are independent work (the chain of LDS reads, the
      // division and the double products overlap), stored as three 12-byte pieces
      struct __attribute__((packed, aligned(4))) Vertex
      {
        float v[3];
      };
      Vertex* dst = (Vertex*)(triangles + (out + (incl - nt)) * 9ull);
      uint64_t edges = entry;
      for (unsigned t = 0; t < nt; ++t, edges >>= 12, dst += 3)
      {
        Vertex a, b, c;
        vertex((unsigned)edges & 15u, a.v);
        vertex((unsigned)(edges >> 4) & 15u, b.v);
        vertex((unsigned)(edges >> 8) & 15u, c.v);
        dst[0] = a;
        dst[1] = b;
        dst[2] = c;
      }
    }
    __builtin_amdgcn_wave_barrier();
    out += total;
    q_head = (q_head + n) & 127u;
    q_count -= n;
  };

  for (int zl = zl0; zl < zl1; ++zl)
  {
    uint32_t w_hi[kRows + 1];
    if (zl + 2 <= p.cube_layers)  // the last voxel layer of the slab at most
      layer += plane;
#pragma unroll
    for (int r = 0; r <= kRows; ++r)
    {
      w_hi[r] = w_ahead[r];
      w_ahead[r] = layer[row_off[r]];
    }
    const bool hi_negative = any_negative(w_hi);
    const bool near_surface = lo_negative || hi_negative;  // wave-uniform
    lo_negative = hi_negative;
    Masks m_lo, m_hi;
    if (near_surface)
    {
      m_lo = classify(w_lo);
      m_hi = classify(w_hi);
    }

#pragma unroll
    for (int r = 0; r < kRows; ++r)
    {
      if (!near_surface)
        break;
      const uint64_t all4 = m_lo.neg[r] & m_lo.neg[r + 1] & m_hi.neg[r] & m_hi.neg[r + 1];
      const uint64_t any4 = m_lo.neg[r] | m_lo.neg[r + 1] | m_hi.neg[r] | m_hi.neg[r + 1];
      const uint64_t val4 = m_lo.val[r] & m_lo.val[r + 1] & m_hi.val[r] & m_hi.val[r + 1];
      // ... and over the voxels at x and x + 1
      const uint64_t surface = (any4 | (any4 >> 1)) & ~(all4 & (all4 >> 1)) & val4 & (val4 >> 1) & cube_lanes;
      if (surface == 0)  // wave-uniform
        continue;

      // the words at x + 1, and the cube's case: marching_cubes.cpp:43-96
      const uint32_t na_lo = __shfl_down(w_lo[r], 1, 64), nb_lo = __shfl_down(w_lo[r + 1], 1, 64);
      const uint32_t na_hi = __shfl_down(w_hi[r], 1, 64), nb_hi = __shfl_down(w_hi[r + 1], 1, 64);
      const unsigned a_lo = voxel_code(w_lo[r], p.min_weight) | (voxel_code(na_lo, p.min_weight) << 2);
      const unsigned b_lo = voxel_code(w_lo[r + 1], p.min_weight) | (voxel_code(nb_lo, p.min_weight) << 2);
      const unsigned a_hi = voxel_code(w_hi[r], p.min_weight) | (voxel_code(na_hi, p.min_weight) << 2);
      const unsigned b_hi = voxel_code(w_hi[r + 1], p.min_weight) | (voxel_code(nb_hi, p.min_weight) << 2);
      bool valid_cube;
      const unsigned ci = cube_index(a_lo, b_lo, a_hi, b_hi, valid_cube);
      const uint64_t entry = ((surface >> lane) & 1u) && valid_cube ? table[ci] : 0;
      const unsigned nt = (unsigned)(entry >> 60);

      if (!EMIT)
        n_mine += nt;
      else
      {
        // queue the cubes that hold triangles; they are worked off 64 at a time, whichever rows and layers
        // they come from (a curved surface crosses a row in a handful of cubes: emitting row by row kept
        // a few lanes of the wave busy)
        const uint64_t holders = __builtin_amdgcn_ballot_w64(nt != 0);
        if (nt != 0)
        {
          const unsigned before = __builtin_amdgcn_mbcnt_hi((uint32_t)(holders >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)holders, 0u));
          const unsigned slot = (q_head + q_count + before) & 127u;
          queue[wave][0][slot] = (nb_lo & 0xffffu) | (na_lo << 16);          // corners 0, 1
          queue[wave][1][slot] = (w_lo[r] & 0xffffu) | (w_lo[r + 1] << 16);  // corners 2, 3
          queue[wave][2][slot] = (nb_hi & 0xffffu) | (na_hi << 16);          // corners 4, 5
          queue[wave][3][slot] = (w_hi[r] & 0xffffu) | (w_hi[r + 1] << 16);  // corners 6, 7
          queue[wave][4][slot] = (uint32_t)entry;
          queue[wave][5][slot] = (uint32_t)(entry >> 32);
          queue[wave][6][slot] = (unsigned)lane | ((unsigned)r << 6) | ((unsigned)(zl - zl0) << 8);
        }
        q_count += (unsigned)__builtin_popcountll(holders);
        if (q_count >= 64u)
          drain(64u);
      }
    }
#pragma unroll
    for (int r = 0; r <= kRows; ++r)
      w_lo[r] = w_hi[r];
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
  if (EMIT && q_count != 0)
    drain(q_count);
}

// Exclusive scan of the tile counts in two small launches: every workgroup scans kScanBlock counts
// (4 per thread, one 16-byte load) into offsets local to the block and publishes the block's sum; one
// workgroup then scans the block sums.  The emit pass adds the two.
// With live_local / block_live (marching cubes): the same scan over "this tile holds triangles", packed into the upper
// half of the scanned word (a block holds 1024 tiles, its triangles stay far below 2^32) -- the emit pass is launched
// over the live tiles only (mc_compact_kernel).
__global__ __launch_bounds__(256) void mc_scan_local_kernel(const uint32_t* counts, uint32_t* local_offsets, size_t n,
                                                            unsigned long long* block_sums, uint32_t* live_local,
                                                            unsigned long long* block_live)
{
  __shared__ unsigned long long wave_sum[4];
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  const size_t base = (size_t)blockIdx.x * kScanBlock + (size_t)threadIdx.x * 4;
  uint32_t c[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    c[i] = base + i < n ? counts[base + i] : 0u;
  auto packed = [](uint32_t v) { return (unsigned long long)v | ((unsigned long long)(v != 0u) << 32); };
  const unsigned long long mine = packed(c[0]) + packed(c[1]) + packed(c[2]) + packed(c[3]);
  unsigned long long incl = mine;
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
  unsigned long long before = incl - mine;
  for (int w = 0; w < wave; ++w)
    before += wave_sum[w];
#pragma unroll
  for (int i = 0; i < 4; ++i)
  {
    if (base + i < n)
    {
      local_offsets[base + i] = (uint32_t)before;
      if (live_local)
        live_local[base + i] = (uint32_t)(before >> 32);
    }
    before += packed(c[i]);
  }
  if (threadIdx.x == 255)
  {
    block_sums[blockIdx.x] = before & 0xffffffffull;
    if (block_live)
      block_live[blockIdx.x] = before >> 32;
  }
}

// The live tiles in tile order, each with the slot of its first triangle: entry k = {tile, slot} of the k-th tile that
// holds triangles.  One thread per tile, after both scans.
__global__ __launch_bounds__(256) void mc_compact_kernel(const uint32_t* counts, const uint32_t* local_offsets,
                                                         const unsigned long long* block_bases, const uint32_t* live_local,
                                                         const unsigned long long* block_live, size_t n,
                                                         unsigned long long* list)
{
  const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= n || counts[i] == 0u)
    return;
  const size_t b = i / kScanBlock;
  const unsigned long long k = block_live[b] + live_local[i];
  list[2 * k] = i;
  list[2 * k + 1] = block_bases[b] + local_offsets[i];
}

__global__ __launch_bounds__(1024) void mc_scan_blocks_kernel(unsigned long long* block_sums, size_t n_blocks,
                                                              unsigned long long* total)
{
  // (blockIdx.x == 1: the same over the live-tile sums, which follow the triangle sums; its total follows `total`)
  block_sums += (size_t)blockIdx.x * n_blocks;
  total += blockIdx.x;
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
// ---- cloud extraction: kfusion::device::extractCloud (yak/src/cuda/tsdf_volume.cu:464-665, 764-780) ----------
// Every voxel with weight != 0 and tsdf != 1 gives a point for each of its +x, +y, +z neighbours that also has
// weight != 0, tsdf != 1 and the opposite strict sign: the zero crossing between the two voxel centres,
// mapped by the volume pose.  The reference appends through a global atomic counter (:616), so its order
// changes from run to run; here the same count -> scan -> emit schedule as marching cubes fixes every
// point's slot (tile, z, row, x, then dx, dy, dz).  Same wave tile too: 63 owner voxels along x (lane i + 1
// supplies the +x neighbour) by 4 owner rows (a fifth row supplies +y), walking z with the next layer in
// registers; a tile layer with no negative voxel cannot hold a sign change and costs five compares.
struct CloudParams
{
  const uint32_t* vol;  // first held layer
  int X, Y;
  int z0;               // global index of the first held layer
  int pair_layers;      // layers z that have z + 1 inside the slab
  int nx, ny;           // wave tiles along x and y
  int z_chunk, nzc, z_sub, nzs;
  float vsx, vsy, vsz;
  float R[9], t[3];
  unsigned long long capacity;
  // surface map of the slab (yakhip_volume.surface_map) or nullptr, as in McParams; vol's first layer is the map's
  const uint32_t* map;
  int map_x, map_y;
};

// "W != 0 && F != 1.f" (tsdf_volume.cu:524, 535, ...) and the value
__device__ __forceinline__ bool cloud_voxel(uint32_t word, float& F)
{
  F = half_bits_to_float(word & 0xffffu);
  return (word >> 16) != 0 && F != 1.f;
}

template <bool EMIT>
__global__ __launch_bounds__(256) void cloud_kernel(const CloudParams p, uint32_t* counts, const unsigned long long* list,
                                                    unsigned long long n_live, float4* points)
{
  const int lane = (int)threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  // count pass: every XCD gets a contiguous run of workgroups (shared lines and rows meet in one L2); the
  // emit pass keeps round-robin so that the few tiles with surface spread over the XCDs (see mc_kernel)
  const unsigned n_wg = gridDim.x, per_xcd = n_wg / 8u;
  const unsigned wg =
      !EMIT && blockIdx.x < per_xcd * 8u ? (blockIdx.x % 8u) * per_xcd + blockIdx.x / 8u : blockIdx.x;
  const unsigned gy = (unsigned)(p.ny + 3) / 4u;
  const int walk = EMIT ? p.z_sub : p.z_chunk;
  const size_t tile_stride = (size_t)p.ny * p.nx;
  int xt, yt, zc;
  size_t tile;
  unsigned long long out = 0;
  if (EMIT)  // (launched over the tiles that hold points: wave k takes entry k of the list, as mc_kernel's emit pass)
  {
    const unsigned long long k = (unsigned long long)blockIdx.x * 4ull + (unsigned long long)wave;
    if (k >= n_live)
      return;
    tile = (size_t)list[2 * k];
    out = list[2 * k + 1];
    xt = (int)(tile % (size_t)p.nx), yt = (int)((tile / (size_t)p.nx) % (size_t)p.ny), zc = (int)(tile / tile_stride);
  }
  else
  {
    xt = (int)(wg % (unsigned)p.nx), yt = (int)((wg / (unsigned)p.nx) % gy) * 4 + wave, zc = (int)(wg / ((unsigned)p.nx * gy));
    if (yt >= p.ny)
      return;
    tile = ((size_t)zc * (size_t)(walk / p.z_sub) * p.ny + yt) * p.nx + xt;
  }
  if (!EMIT && p.map)
  {
    // A point needs a pair of voxels of opposite strict sign, one of them negative, and a brick whose bit is clear
    // holds no negative voxel: if none of the bricks this wave's voxels lie in (owners and their +x, +y, +z
    // neighbours) is marked, its tiles are empty and nothing is read (as in mc_kernel).
    const int vx0 = xt * kRun, vx1 = min(vx0 + kRun, p.X - 1), vy0 = yt * kRows, vy1 = min(vy0 + kRows, p.Y - 1);
    const int vz0 = zc * walk, vz1 = min(vz0 + walk, p.pair_layers);  // pair layer z uses layers z and z + 1
    const int bx0 = vx0 >> 4, nbx = (vx1 >> 4) - bx0 + 1, by0 = vy0 >> 4, nby = (vy1 >> 4) - by0 + 1;
    const int bz0 = vz0 >> 4, nbz = (vz1 >> 4) - bz0 + 1;
    bool marked = false;
    for (int i = lane; i < nbx * nby * nbz; i += 64)
    {
      const int ix = i % nbx, iy = (i / nbx) % nby, iz = i / (nbx * nby);
      const uint32_t b = (uint32_t)(bx0 + ix) + (uint32_t)p.map_x * ((uint32_t)(by0 + iy) + (uint32_t)p.map_y * (uint32_t)(bz0 + iz));
      marked = marked || ((p.map[b >> 5] >> (b & 31u)) & 1u);
    }
    if (__builtin_amdgcn_ballot_w64(marked) == 0)
    {
      const int n_sub = (vz1 - vz0 + p.z_sub - 1) / p.z_sub;
      if (lane < n_sub)
        counts[tile + (size_t)lane * tile_stride] = 0;
      return;
    }
  }
  const int x = xt * kRun + lane, y0 = yt * kRows;
  const bool in_x = x < p.X;
  const bool owner = lane < kRun && in_x;       // this lane's voxels are counted here
  const bool has_x1 = owner && x + 1 < p.X;     // ... and have a +x neighbour (in lane + 1)
  const size_t plane = (size_t)p.X * p.Y;
  const int zl0 = zc * walk, zl1 = min(zl0 + walk, p.pair_layers);

  uint32_t row_off[kRows + 1];
#pragma unroll
  for (int r = 0; r <= kRows; ++r)
    row_off[r] = (uint32_t)min(y0 + r, p.Y - 1) * (uint32_t)p.X + (uint32_t)(in_x ? x : 0);
  auto any_negative = [&](const uint32_t(&w)[kRows + 1]) {
    uint64_t any = 0;
#pragma unroll
    for (int r = 0; r <= kRows; ++r)
      any |= __builtin_amdgcn_ballot_w64(half_bits_to_float(w[r] & 0xffffu) < 0.f);
    return any != 0;
  };

  const uint32_t* layer = p.vol + (size_t)zl0 * plane;
  uint32_t w_lo[kRows + 1], w_ahead[kRows + 1];
#pragma unroll
  for (int r = 0; r <= kRows; ++r)
    w_lo[r] = layer[row_off[r]];
  layer += plane;
#pragma unroll
  for (int r = 0; r <= kRows; ++r)
    w_ahead[r] = layer[row_off[r]];
  bool lo_negative = any_negative(w_lo);
  unsigned n_mine = 0;

  for (int zl = zl0; zl < zl1; ++zl)
  {
    uint32_t w_hi[kRows + 1];
    if (zl + 2 <= p.pair_layers)
      layer += plane;
#pragma unroll
    for (int r = 0; r <= kRows; ++r)
    {
      w_hi[r] = w_ahead[r];
      w_ahead[r] = layer[row_off[r]];
    }
    const bool hi_negative = any_negative(w_hi);
    const bool near_surface = lo_negative || hi_negative;  // wave-uniform: a sign change needs a negative voxel
    lo_negative = hi_negative;

#pragma unroll
    for (int r = 0; r < kRows; ++r)
    {
      if (!near_surface)
        break;
      // tsdf_volume.cu:519-591 for the voxel (x, y0 + r, z0 + zl)
      float F, Fx, Fy, Fz;
      const bool ok = cloud_voxel(w_lo[r], F) && owner && y0 + r < p.Y;
      const bool okx = cloud_voxel(__shfl_down(w_lo[r], 1, 64), Fx) && has_x1;
      const bool oky = cloud_voxel(w_lo[r + 1], Fy) && y0 + r + 1 < p.Y;
      const bool okz = cloud_voxel(w_hi[r], Fz);
      const bool cx = ok && okx && ((F > 0.f && Fx < 0.f) || (F < 0.f && Fx > 0.f));
      const bool cy = ok && oky && ((F > 0.f && Fy < 0.f) || (F < 0.f && Fy > 0.f));
      const bool cz = ok && okz && ((F > 0.f && Fz < 0.f) || (F < 0.f && Fz > 0.f));
      const unsigned cnt = (unsigned)cx + (unsigned)cy + (unsigned)cz;
      if (!EMIT)
        n_mine += cnt;
      else if (__builtin_amdgcn_ballot_w64(cnt != 0) != 0)  // wave-uniform
      {
        unsigned incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1)
        {
          const unsigned up = __shfl_up(incl, off, 64);
          if (lane >= off)
            incl += up;
        }
        const unsigned total = __shfl(incl, 63, 64);
        unsigned long long slot = out + (incl - cnt);
        // :508-509, 526  voxel centre
        const float Vx = ((float)x + 0.5f) * p.vsx, Vy = ((float)(y0 + r) + 0.5f) * p.vsy,
                    Vz = ((float)(p.z0 + zl) + 0.5f) * p.vsz;
        auto put = [&](float px, float py, float pz) {  // aff * p, device.hpp:106-111
          if (slot < p.capacity)
            points[slot] = make_float4(dot3(p.R[0], p.R[1], p.R[2], px, py, pz) + p.t[0],
                                       dot3(p.R[3], p.R[4], p.R[5], px, py, pz) + p.t[1],
                                       dot3(p.R[6], p.R[7], p.R[8], px, py, pz) + p.t[2], 0.f);
          ++slot;
        };
        // :539-545  p = (V * |Fn| + Vn * |F|) * (1 / (|F| + |Fn|)) along the axis of the neighbour
        const float aF = __builtin_fabsf(F);
        if (cx)
        {
          const float aN = __builtin_fabsf(Fx), d_inv = 1.f / (aF + aN);
          put((Vx * aN + (Vx + p.vsx) * aF) * d_inv, Vy, Vz);
        }
        if (cy)
        {
          const float aN = __builtin_fabsf(Fy), d_inv = 1.f / (aF + aN);
          put(Vx, (Vy * aN + (Vy + p.vsy) * aF) * d_inv, Vz);
        }
        if (cz)
        {
          const float aN = __builtin_fabsf(Fz), d_inv = 1.f / (aF + aN);
          put(Vx, Vy, (Vz * aN + (Vz + p.vsz) * aF) * d_inv);
        }
        out += total;
      }
    }
#pragma unroll
    for (int r = 0; r <= kRows; ++r)
      w_lo[r] = w_hi[r];
    if (!EMIT && ((zl + 1 - zl0) % p.z_sub == 0 || zl + 1 == zl1))  // a tile is complete
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
}  // namespace

// scratch of the count -> scan -> emit passes (kept in the context, grown on demand)
struct TileScratch
{
  uint32_t *counts, *local_offsets, *live_local;
  unsigned long long *block_bases, *block_live;  // (block_live follows block_bases: mc_scan_blocks_kernel's second block)
  unsigned long long* total;                     // [0] triangles / points, [1] live tiles
  unsigned long long* list;                      // marching cubes: {tile, first slot} per live tile
  size_t n_blocks;
};
static int tile_scratch(yakhip_ctx* ctx, size_t n_tiles, bool with_list, TileScratch* out);

int launch_marching_cubes(yakhip_ctx* ctx, const uint32_t* vol, int X, int Y, int z0, int layers, int min_weight,
                          double scale, float* triangles, unsigned long long capacity, unsigned long long* n_out,
                          const uint32_t* map, int map_z0)
{
  *n_out = 0;
  if (X < 2 || Y < 2 || layers < 2)
    return 0;  // no cube fits (marching_cubes.cpp:164-168: the loops do not run)
  if ((unsigned long long)X * (unsigned long long)Y > (1ull << 30))
    return fail(YAKHIP_ERANGE, "marching_cubes: a layer of more than 2^30 voxels");  // 32-bit offsets within a layer
  McParams p;
  p.vol = vol;
  p.X = X, p.Y = Y, p.z0 = z0;
  p.cube_layers = layers - 1;
  p.nx = (X - 1 + kRun - 1) / kRun;
  p.ny = (Y - 1 + kRows - 1) / kRows;
  // tiles of 8 cube layers; the count pass walks 16 layers (two tiles) per wave.  Measured at 512^3 on the
  // orbit scene: count 0.117 ms for walks of 16, 0.17 for 32 (too few waves); emit 0.090 ms at 8 layers
  // per tile, 0.094 at 4, 0.115 at 2 and at 16 (short tiles fill the cube queue badly, long ones leave
  // too few waves)
  // (round 3, with the emit pass launched over the live tiles only: emit 0.074 / 0.066 / 0.045 / 0.045 / 0.054 ms at
  // 16 / 8 / 4 / 2 / 1 layers per tile for 0.96 M triangles -- short tiles cost nothing any more where they are empty --
  // and count + scan 0.120 / 0.120 / 0.122 / 0.129 / 0.147)
  p.z_sub = ctx->mc_z_chunk > 0 ? ctx->mc_z_chunk : 4;
  p.z_chunk = p.z_sub * std::max(1, 16 / p.z_sub);
  p.nzs = (p.cube_layers + p.z_sub - 1) / p.z_sub;
  p.nzc = (p.cube_layers + p.z_chunk - 1) / p.z_chunk;
  p.min_weight = min_weight;
  p.scale = scale;
  p.map = ctx->surface_skip ? map : nullptr;
  p.map_x = (X + 15) >> 4, p.map_y = (Y + 15) >> 4, p.map_z0 = map_z0;
  const unsigned gy = (unsigned)((p.ny + 3) / 4);
  if (gy > 65535u)
    return fail(YAKHIP_ERANGE, "marching_cubes: more than a million voxel rows");
  const size_t n_tiles = (size_t)p.nx * p.ny * p.nzs;

  TileScratch ts;
  {
    const int rc = tile_scratch(ctx, n_tiles, true, &ts);
    if (rc != 0)
      return rc;
  }

  const size_t wgs_layer = (size_t)p.nx * gy;
  if (wgs_layer * (size_t)p.nzs > 0x7fffffffull)
    return fail(YAKHIP_ERANGE, "marching_cubes: more than 2^31 workgroups");
  const dim3 grid((unsigned)(wgs_layer * p.nzc));
  std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
  int rc = timing_begin(ctx, &ev);  // one event pair around count + scan (+ the emit pass below gets its own)
  if (rc != 0)
    return rc;
  hipLaunchKernelGGL((mc_kernel<false>), grid, dim3(256), 0, ctx->stream, p, ts.counts, (const unsigned long long*)nullptr,
                     0ull, (float*)nullptr);
  YAKHIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(mc_scan_local_kernel, dim3((unsigned)ts.n_blocks), dim3(256), 0, ctx->stream, ts.counts, ts.local_offsets,
                     n_tiles, ts.block_bases, ts.live_local, ts.block_live);
  hipLaunchKernelGGL(mc_scan_blocks_kernel, dim3(2), dim3(1024), 0, ctx->stream, ts.block_bases, ts.n_blocks, ts.total);
  // the tiles that hold triangles, in tile order, each with its first slot: the emit pass is launched over these alone
  hipLaunchKernelGGL(mc_compact_kernel, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, ctx->stream, ts.counts,
                     ts.local_offsets, ts.block_bases, ts.live_local, ts.block_live, n_tiles, ts.list);
  if ((rc = timing_end(ctx, ev)) != 0)
    return rc;
  YAKHIP_TRY(hipGetLastError());
  YAKHIP_TRY(hipMemcpyAsync(ctx->mc_pinned, ts.total, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  YAKHIP_TRY(hipStreamSynchronize(ctx->stream));
  const unsigned long long n = ((const unsigned long long*)ctx->mc_pinned)[0];
  const unsigned long long n_live = ((const unsigned long long*)ctx->mc_pinned)[1];
  *n_out = n;
  if (!triangles || n == 0 || n > capacity)
    return 0;  // count only (or the caller's buffer is too small: it sees n > capacity)

  if ((rc = timing_begin(ctx, &ev)) != 0)
    return rc;
  hipLaunchKernelGGL((mc_kernel<true>), dim3((unsigned)((n_live + 3) / 4)), dim3(256), 0, ctx->stream, p, ts.counts, ts.list,
                     n_live, triangles);
  if ((rc = timing_end(ctx, ev)) != 0)
    return rc;
  YAKHIP_TRY(hipGetLastError());
  return check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}
// counts (u32) | local offsets (u32) | live offsets (u32) | block bases (u64) | live block bases (u64) | totals (2 x u64) |
// the live-tile list (2 x u64 per tile, marching cubes only)
static int tile_scratch(yakhip_ctx* ctx, size_t n_tiles, bool with_list, TileScratch* out)
{
  const size_t n_blocks = (n_tiles + kScanBlock - 1) / kScanBlock;
  const size_t counts_bytes = (n_tiles * sizeof(uint32_t) + 15) & ~(size_t)15;
  const size_t fixed = 3 * counts_bytes + (2 * n_blocks + 2) * sizeof(unsigned long long);
  const size_t need = fixed + (with_list ? n_tiles * 2 * sizeof(unsigned long long) : 0);
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
    YAKHIP_TRY(hipHostMalloc(&ctx->mc_pinned, 2 * sizeof(unsigned long long), hipHostMallocDefault));
  uint8_t* base = (uint8_t*)ctx->mc_scratch;
  out->counts = (uint32_t*)base;
  out->local_offsets = (uint32_t*)(base + counts_bytes);
  out->live_local = (uint32_t*)(base + 2 * counts_bytes);
  out->block_bases = (unsigned long long*)(base + 3 * counts_bytes);
  out->block_live = out->block_bases + n_blocks;
  out->total = out->block_live + n_blocks;
  out->list = (unsigned long long*)(base + fixed);
  out->n_blocks = n_blocks;
  return 0;
}

int launch_extract_cloud(yakhip_ctx* ctx, const uint32_t* vol, int X, int Y, int z0, int layers, const float vs[3],
                         const float R[9], const float t[3], float* points, unsigned long long capacity,
                         unsigned long long* n_out, const uint32_t* map)
{
  *n_out = 0;
  if (layers < 2)
    return 0;  // tsdf_volume.cu:514: the z loop stops one layer short
  if ((unsigned long long)X * (unsigned long long)Y > (1ull << 30))
    return fail(YAKHIP_ERANGE, "extract_cloud: a layer of more than 2^30 voxels");
  CloudParams p;
  p.vol = vol;
  p.X = X, p.Y = Y, p.z0 = z0;
  p.pair_layers = layers - 1;
  p.nx = (X + kRun - 1) / kRun;
  p.ny = (Y + kRows - 1) / kRows;
  p.z_sub = 8;
  p.z_chunk = 16;
  p.nzs = (p.pair_layers + p.z_sub - 1) / p.z_sub;
  p.nzc = (p.pair_layers + p.z_chunk - 1) / p.z_chunk;
  p.vsx = vs[0], p.vsy = vs[1], p.vsz = vs[2];
  for (int i = 0; i < 9; ++i)
    p.R[i] = R[i];
  for (int i = 0; i < 3; ++i)
    p.t[i] = t[i];
  p.capacity = points ? capacity : 0;
  p.map = ctx->surface_skip ? map : nullptr;
  p.map_x = (X + 15) >> 4, p.map_y = (Y + 15) >> 4;
  const unsigned gy = (unsigned)((p.ny + 3) / 4);
  const size_t n_tiles = (size_t)p.nx * p.ny * p.nzs, wgs_layer = (size_t)p.nx * gy;
  if (wgs_layer * (size_t)p.nzs > 0x7fffffffull)
    return fail(YAKHIP_ERANGE, "extract_cloud: more than 2^31 workgroups");
  TileScratch ts;
  int rc = tile_scratch(ctx, n_tiles, true, &ts);
  if (rc != 0)
    return rc;

  std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
  if ((rc = timing_begin(ctx, &ev)) != 0)
    return rc;
  hipLaunchKernelGGL((cloud_kernel<false>), dim3((unsigned)(wgs_layer * p.nzc)), dim3(256), 0, ctx->stream, p, ts.counts,
                     (const unsigned long long*)nullptr, 0ull, (float4*)nullptr);
  YAKHIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(mc_scan_local_kernel, dim3((unsigned)ts.n_blocks), dim3(256), 0, ctx->stream, ts.counts, ts.local_offsets,
                     n_tiles, ts.block_bases, ts.live_local, ts.block_live);
  hipLaunchKernelGGL(mc_scan_blocks_kernel, dim3(2), dim3(1024), 0, ctx->stream, ts.block_bases, ts.n_blocks, ts.total);
  hipLaunchKernelGGL(mc_compact_kernel, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, ctx->stream, ts.counts,
                     ts.local_offsets, ts.block_bases, ts.live_local, ts.block_live, n_tiles, ts.list);
  if ((rc = timing_end(ctx, ev)) != 0)
    return rc;
  YAKHIP_TRY(hipGetLastError());
  YAKHIP_TRY(hipMemcpyAsync(ctx->mc_pinned, ts.total, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  YAKHIP_TRY(hipStreamSynchronize(ctx->stream));
  const unsigned long long n = ((const unsigned long long*)ctx->mc_pinned)[0];
  const unsigned long long n_live = ((const unsigned long long*)ctx->mc_pinned)[1];
  *n_out = n;
  if (!points || n == 0 || capacity == 0)
    return 0;
  if ((rc = timing_begin(ctx, &ev)) != 0)
    return rc;
  // (the emit pass over the tiles that hold points only, like marching cubes')
  hipLaunchKernelGGL((cloud_kernel<true>), dim3((unsigned)((n_live + 3) / 4)), dim3(256), 0, ctx->stream, p, ts.counts, ts.list,
                     n_live, (float4*)points);
  if ((rc = timing_end(ctx, ev)) != 0)
    return rc;
  YAKHIP_TRY(hipGetLastError());
  return check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}
}  // namespace yakhip
