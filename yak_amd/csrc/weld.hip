// weld.hip -- exact vertex merging of a triangle soup on gfx950 (yakhip_weld_vertices).
// The reference cleans its marching-cubes mesh with VTK's vtkCleanPolyData (yak/src/mc/marching_cubes.cpp:204-216,
// MarchingCubesParameters::clean); VTK is not part of this tree, and its merging of the bit-identical corners
// marching cubes produces on shared cube edges is restated as: corners with the same three bit patterns are
// one point.  On the host that is a sort of 3 corners per triangle (120 ms for 0.8 M triangles on four
// threads).  Here (round 3): the corners meet in a hash table, where the FIRST corner of the list with a given point
// ends up standing for it (atomicMin over the corner numbers: no race in the outcome); a scan over "this corner is
// the first of its point" numbers the points in the order of their first appearance in the triangle list -- the
// order vtkCleanPolyData's incremental point locator gives them too -- and every corner reads the number of the
// corner that stands for its point.  No sort.  (Round 2 sorted all corners three times, on (x, y, z) bit patterns:
// 1.05 ms per million triangles; a sort of the distinct points only, behind the same table, 0.45 ms -- 48 launches
// of the library's sort passes at 5 us each.)  The scratch lives in the context.
// Citations are relative to /root/reference/.
#include <string.h>

#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "yakhip_internal.h"

namespace yakhip
{
namespace
{
constexpr uint32_t kEmpty = 0xffffffffu;

__device__ __forceinline__ uint32_t point_hash(uint32_t x, uint32_t y, uint32_t z)
{
  uint32_t h = x * 0x9e3779b1u;
  h = (h ^ (h >> 15)) + y * 0x85ebca77u;
  h = (h ^ (h >> 13)) + z * 0xc2b2ae3du;
  h ^= h >> 16;
  h *= 0x7feb352du;
  return h ^ (h >> 15);
}

// Corner i goes to the first slot from its hash on that is empty or holds a corner with the same three words; the
// slot keeps the smallest corner number that came to it.  slot_of[i] = that slot.
__global__ __launch_bounds__(256) void weld_insert_kernel(const uint32_t* corners, uint32_t n, uint32_t* table, uint32_t mask,
                                                          uint32_t* slot_of)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n)
    return;
  const uint32_t x = corners[3ull * i], y = corners[3ull * i + 1], z = corners[3ull * i + 2];
  uint32_t h = point_hash(x, y, z) & mask;
  for (;;)  // (the table has at least twice as many slots as corners: an empty slot is always reached)
  {
    uint32_t held = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (held == kEmpty)
    {
      held = atomicCAS(&table[h], kEmpty, i);
      if (held == kEmpty)
        break;
    }
    // (whoever holds the slot now or later has the words of the corner seen here: only equal points replace each other)
    const uint32_t* q = corners + 3ull * held;
    if (q[0] == x && q[1] == y && q[2] == z)
    {
      if (i < held)
        atomicMin(&table[h], i);
      break;
    }
    h = (h + 1u) & mask;
  }
  slot_of[i] = h;
}

// 1 where corner i is the first of the list with its point
struct FirstOfItsPoint
{
  const uint32_t *table, *slot_of;
  __device__ uint32_t operator()(uint32_t i) const { return table[slot_of[i]] == i ? 1u : 0u; }
};

// number[i] = how many points appeared before corner i.  The corner that stands for a point writes the vertex; every
// corner takes the number of the corner that stands for its point.
__global__ __launch_bounds__(256) void weld_emit_kernel(const uint32_t* corners, const uint32_t* table, const uint32_t* slot_of,
                                                        const uint32_t* number, uint32_t* vertices, uint32_t* index, uint32_t n,
                                                        uint32_t* n_points)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n)
    return;
  const uint32_t first = table[slot_of[i]], id = number[first];
  index[i] = id;
  if (first == i)
  {
    vertices[3ull * id + 0] = corners[3ull * i + 0];
    vertices[3ull * id + 1] = corners[3ull * i + 1];
    vertices[3ull * id + 2] = corners[3ull * i + 2];
  }
  if (i == n - 1u)
    *n_points = number[i] + (first == i ? 1u : 0u);
}
}  // namespace

int launch_weld(yakhip_ctx* ctx, const uint32_t* corners, uint32_t n, uint32_t* vertices, uint32_t* index,
                unsigned long long* n_vertices_host)
{
  *n_vertices_host = 0;
  if (n == 0)
    return 0;
  hipStream_t s = ctx->stream;
  uint32_t slots = 1024;
  while (slots < 2u * n && slots < 0x80000000u)
    slots <<= 1;
  // the table, two arrays of n words (slot per corner, points before each corner), the count of points, the library's
  // scratch -- kept in the context, grown on demand
  const size_t arr = ((size_t)n * 4u + 255u) & ~(size_t)255u, table_bytes = (size_t)slots * 4u;
  char* base = nullptr;
  auto flags = [&](const uint32_t* table, const uint32_t* slot_of) {
    return rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0u), FirstOfItsPoint{ table, slot_of });
  };
  size_t scan_bytes = 0;
  YAKHIP_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, flags(nullptr, nullptr), (uint32_t*)nullptr, 0u, (size_t)n,
                                     rocprim::plus<uint32_t>(), s));
  const size_t tmp_bytes = scan_bytes + 256;
  const size_t need = table_bytes + 2 * arr + 256 + tmp_bytes;
  if (ctx->weld_scratch_bytes < need)
  {
    if (ctx->weld_scratch)
    {
      YAKHIP_TRY(hipStreamSynchronize(s));
      YAKHIP_TRY(hipFree(ctx->weld_scratch));
    }
    ctx->weld_scratch = nullptr;
    ctx->weld_scratch_bytes = 0;
    YAKHIP_TRY(hipMalloc(&ctx->weld_scratch, need));
    ctx->weld_scratch_bytes = need;
  }
  if (!ctx->mc_pinned)
    YAKHIP_TRY(hipHostMalloc(&ctx->mc_pinned, 2 * sizeof(unsigned long long), hipHostMallocDefault));
  base = (char*)ctx->weld_scratch;
  uint32_t* table = (uint32_t*)base;
  uint32_t* slot_of = (uint32_t*)(base + table_bytes);
  uint32_t* number = (uint32_t*)(base + table_bytes + arr);
  uint32_t* n_points_dev = (uint32_t*)(base + table_bytes + 2 * arr);
  void* tmp = base + table_bytes + 2 * arr + 256;
  const dim3 grid((n + 255u) / 256u), wg(256);

  YAKHIP_TRY(hipMemsetAsync(table, 0xff, table_bytes, s));
  hipLaunchKernelGGL(weld_insert_kernel, grid, wg, 0, s, corners, n, table, slots - 1u, slot_of);
  YAKHIP_TRY(hipGetLastError());
  size_t bytes = tmp_bytes;
  YAKHIP_TRY(rocprim::exclusive_scan(tmp, bytes, flags(table, slot_of), number, 0u, (size_t)n, rocprim::plus<uint32_t>(), s));
  hipLaunchKernelGGL(weld_emit_kernel, grid, wg, 0, s, corners, table, slot_of, number, vertices, index, n, n_points_dev);
  YAKHIP_TRY(hipGetLastError());
  YAKHIP_TRY(hipMemcpyAsync(ctx->mc_pinned, n_points_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  YAKHIP_TRY(hipStreamSynchronize(s));
  *n_vertices_host = *(const uint32_t*)ctx->mc_pinned;
  return 0;
}
}  // namespace yakhip
