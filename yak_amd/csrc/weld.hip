// weld.hip -- exact vertex merging of a triangle soup on gfx950 (yakhip_weld_vertices).
// The reference cleans its marching-cubes mesh with VTK's vtkCleanPolyData (yak/src/mc/marching_cubes.cpp:204-216,
// MarchingCubesParameters::clean); VTK is not part of this tree, and its merging of the bit-identical corners
// marching cubes produces on shared cube edges is restated as: corners with the same three bit patterns are
// one point.  On the host that is a sort of 3 corners per triangle (120 ms for 0.8 M triangles on four
// threads); here it is three stable radix-sort passes (z, then y, then x bits: together the lexicographic
// order on (x, y, z, corner) the host routine used), a head flag per run, a scan and a scatter.
// Citations are relative to /root/reference/.
#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "yakhip_internal.h"

namespace yakhip
{
namespace
{
// keys[i] = component `comp` of the corner that sits at place i of the current order (identity if order == nullptr)
__global__ __launch_bounds__(256) void weld_keys_kernel(const uint32_t* corners, const uint32_t* order, int comp,
                                                        uint32_t* keys, uint32_t* identity, uint32_t n)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n)
    return;
  const uint32_t c = order ? order[i] : i;
  keys[i] = corners[3ull * c + (uint32_t)comp];
  if (identity)
    identity[i] = i;
}

// head[i] = 1 where place i starts a run of equal points
__global__ __launch_bounds__(256) void weld_heads_kernel(const uint32_t* corners, const uint32_t* order, uint32_t* head,
                                                         uint32_t n)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n)
    return;
  uint32_t h = 1;
  if (i > 0)
  {
    const uint32_t* a = corners + 3ull * order[i];
    const uint32_t* b = corners + 3ull * order[i - 1];
    h = a[0] != b[0] || a[1] != b[1] || a[2] != b[2];
  }
  head[i] = h;
}

// vertex id of place i = (number of heads up to and including i) - 1
__global__ __launch_bounds__(256) void weld_emit_kernel(const uint32_t* corners, const uint32_t* order,
                                                        const uint32_t* head, const uint32_t* scan, uint32_t* vertices,
                                                        uint32_t* index, uint32_t n)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n)
    return;
  const uint32_t c = order[i], id = scan[i] - 1u;
  index[c] = id;
  if (head[i])
  {
    vertices[3ull * id + 0] = corners[3ull * c + 0];
    vertices[3ull * id + 1] = corners[3ull * c + 1];
    vertices[3ull * id + 2] = corners[3ull * c + 2];
  }
}

struct DeviceBlock  // freed on every way out of the launcher
{
  void* p = nullptr;
  ~DeviceBlock()
  {
    if (p)
      (void)hipFree(p);
  }
};
}  // namespace

int launch_weld(yakhip_ctx* ctx, const uint32_t* corners, uint32_t n, uint32_t* vertices, uint32_t* index,
                unsigned long long* n_vertices_host)
{
  *n_vertices_host = 0;
  if (n == 0)
    return 0;
  hipStream_t s = ctx->stream;
  size_t sort_bytes = 0, scan_bytes = 0;
  YAKHIP_TRY(rocprim::radix_sort_pairs(nullptr, sort_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                       (uint32_t*)nullptr, (size_t)n, 0u, 32u, s));
  YAKHIP_TRY(rocprim::inclusive_scan(nullptr, scan_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n,
                                     rocprim::plus<uint32_t>(), s));
  const size_t tmp_bytes = (sort_bytes > scan_bytes ? sort_bytes : scan_bytes) + 256;
  // five arrays of n words (keys in / out, order in / out, head flags; the scan reuses a key array) + library scratch
  const size_t arr = ((size_t)n * 4u + 255u) & ~(size_t)255u;
  DeviceBlock block;
  YAKHIP_TRY(hipMalloc(&block.p, 5 * arr + tmp_bytes));
  char* base = (char*)block.p;
  uint32_t* keys_a = (uint32_t*)base;
  uint32_t* keys_b = (uint32_t*)(base + arr);
  uint32_t* ord_a = (uint32_t*)(base + 2 * arr);
  uint32_t* ord_b = (uint32_t*)(base + 3 * arr);
  uint32_t* head = (uint32_t*)(base + 4 * arr);
  void* tmp = base + 5 * arr;
  const dim3 grid((n + 255u) / 256u), wg(256);

  // least significant component first; every pass is stable, so the passes compose to the order on (x, y, z, corner)
  const uint32_t* order = nullptr;
  for (int comp = 2; comp >= 0; --comp)
  {
    hipLaunchKernelGGL(weld_keys_kernel, grid, wg, 0, s, corners, order, comp, keys_a, order ? nullptr : ord_a, n);
    YAKHIP_TRY(hipGetLastError());
    size_t bytes = tmp_bytes;
    YAKHIP_TRY(rocprim::radix_sort_pairs(tmp, bytes, keys_a, keys_b, ord_a, ord_b, (size_t)n, 0u, 32u, s));
    uint32_t* t = ord_a;
    ord_a = ord_b, ord_b = t;
    order = ord_a;
  }
  hipLaunchKernelGGL(weld_heads_kernel, grid, wg, 0, s, corners, order, head, n);
  YAKHIP_TRY(hipGetLastError());
  size_t bytes = tmp_bytes;
  YAKHIP_TRY(rocprim::inclusive_scan(tmp, bytes, head, keys_a, (size_t)n, rocprim::plus<uint32_t>(), s));
  hipLaunchKernelGGL(weld_emit_kernel, grid, wg, 0, s, corners, order, head, keys_a, vertices, index, n);
  YAKHIP_TRY(hipGetLastError());
  uint32_t total = 0;
  YAKHIP_TRY(hipMemcpyAsync(&total, keys_a + (n - 1), sizeof(total), hipMemcpyDeviceToHost, s));
  YAKHIP_TRY(hipStreamSynchronize(s));
  *n_vertices_host = total;
  return 0;
}
}  // namespace yakhip
