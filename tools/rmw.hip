// In-place read-modify-write streaming ceiling on this box (same access pattern as integrate).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT, int UNROLL>
__global__ __launch_bounds__(256) void rmw(u32x4* p, size_t n, size_t stride_elems, int iters)
{
  // each thread walks `iters` layers separated by stride (like the z loop), UNROLL in flight
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x + (size_t)blockIdx.y * iters * stride_elems;
  for (int k = 0; k < iters; k += UNROLL) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride_elems) : p[i + u * stride_elems];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { v[u] += 0x10000u; if (NT) __builtin_nontemporal_store(v[u], p + i + u * stride_elems); else p[i + u * stride_elems] = v[u]; }
    i += UNROLL * stride_elems;
  }
}
template <bool NT, int UNROLL>
__global__ __launch_bounds__(256) void rmw_yfast(u32x4* p, size_t n, size_t stride_elems, int iters)
{
  size_t i = (size_t)blockIdx.y * 256 + threadIdx.x + (size_t)blockIdx.x * iters * stride_elems;
  for (int k = 0; k < iters; k += UNROLL) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride_elems) : p[i + u * stride_elems];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { v[u] += 0x10000u; if (NT) __builtin_nontemporal_store(v[u], p + i + u * stride_elems); else p[i + u * stride_elems] = v[u]; }
    i += UNROLL * stride_elems;
  }
}
// workgroups in groups of G tiles: within a group the tile index runs fastest, then the chunk, then the group
template <bool NT, int UNROLL>
__global__ __launch_bounds__(256) void rmw_grouped(u32x4* p, size_t n, size_t stride_elems, int iters, int nchunks, int G)
{
  const unsigned id = blockIdx.x, per = (unsigned)G * (unsigned)nchunks;
  const unsigned g = id / per, rem = id % per, chunk = rem / (unsigned)G, tile = g * (unsigned)G + rem % (unsigned)G;
  size_t i = (size_t)tile * 256 + threadIdx.x + (size_t)chunk * iters * stride_elems;
  for (int k = 0; k < iters; k += UNROLL) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride_elems) : p[i + u * stride_elems];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { v[u] += 0x10000u; if (NT) __builtin_nontemporal_store(v[u], p + i + u * stride_elems); else p[i + u * stride_elems] = v[u]; }
    i += UNROLL * stride_elems;
  }
}
template <bool NT>
__global__ __launch_bounds__(256) void copy(const u32x4* a, u32x4* b, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    u32x4 v = NT ? __builtin_nontemporal_load(a + i) : a[i];
    if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v;
  }
}
template <typename F> float timeit(F f, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int r = 0; r < reps; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main() {
  const size_t bytes = 512ull << 20, n = bytes / 16;
  u32x4 *a, *b; CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes));
  CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, bytes));
  const size_t plane = 512 * 512 / 4;  // 16-byte elements per layer
  for (int chunk : {32, 64, 128, 256}) {
    dim3 grid(plane / 256, 512 / chunk);
    float t1 = timeit([&] { hipLaunchKernelGGL((rmw<true, 1>), grid, dim3(256), 0, 0, a, n, plane, chunk); }, 20);
    float t2 = timeit([&] { hipLaunchKernelGGL((rmw<true, 2>), grid, dim3(256), 0, 0, a, n, plane, chunk); }, 20);
    float t4 = timeit([&] { hipLaunchKernelGGL((rmw<true, 4>), grid, dim3(256), 0, 0, a, n, plane, chunk); }, 20);
    float p2 = timeit([&] { hipLaunchKernelGGL((rmw<false, 2>), grid, dim3(256), 0, 0, a, n, plane, chunk); }, 20);
    printf("rmw in place chunk=%3d blocks=%5d: nt u1 %.1f us (%.0f GB/s)  nt u2 %.1f us (%.0f GB/s)  nt u4 %.1f us (%.0f GB/s)  plain u2 %.1f us (%.0f GB/s)\n", chunk, grid.x * grid.y,
           t1 * 1e3, 2 * bytes / t1 / 1e6, t2 * 1e3, 2 * bytes / t2 / 1e6, t4 * 1e3, 2 * bytes / t4 / 1e6, p2 * 1e3, 2 * bytes / p2 / 1e6);
  }
  // the same bytes as slabs of wider planes (ranks of a multi-GPU volume): layer stride 1, 2, 4, 16 MiB
  for (size_t side_x : {512, 1024, 1024, 2048}) {
    static int call = 0;
    const size_t side_y = (call++ == 1) ? 512 : side_x;
    const size_t pl = side_x * side_y / 4, layers = n / pl;
    for (int chunk : {16, 32}) {
      if ((size_t)chunk > layers) continue;
      dim3 grid(pl / 256, layers / chunk);
      float t2 = timeit([&] { hipLaunchKernelGGL((rmw<true, 2>), grid, dim3(256), 0, 0, a, n, pl, chunk); }, 20);
      // chunk index fastest: consecutive workgroups take the same tile of different chunks
      float t2y = timeit([&] { hipLaunchKernelGGL((rmw_yfast<true, 2>), dim3(grid.y, grid.x), dim3(256), 0, 0, a, n, pl, chunk); }, 20);
      for (int G : {64, 128, 256, 512}) {
        if ((size_t)G > grid.x) continue;
        float tg = timeit([&] { hipLaunchKernelGGL((rmw_grouped<true, 2>), dim3(grid.x * grid.y), dim3(256), 0, 0, a, n, pl, chunk, (int)grid.y, G); }, 20);
        printf("   grouped G=%3d: %.1f us (%.0f GB/s)\n", G, tg * 1e3, 2 * bytes / tg / 1e6);
      }
      printf("rmw plane %zux%zu (%zu layers) chunk=%2d: tile-fastest %.1f us (%.0f GB/s)  chunk-fastest %.1f us (%.0f GB/s)\n", side_x, side_y,
             layers, chunk, t2 * 1e3, 2 * bytes / t2 / 1e6, t2y * 1e3, 2 * bytes / t2y / 1e6);
    }
  }
  for (int blocks : {2048, 4096, 8192, 16384}) {
    float c1 = timeit([&] { hipLaunchKernelGGL((copy<true>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, 20);
    float c2 = timeit([&] { hipLaunchKernelGGL((copy<false>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, 20);
    printf("copy a->b blocks=%5d: nt %.1f us (%.0f GB/s)  plain %.1f us (%.0f GB/s)\n", blocks, c1 * 1e3, 2 * bytes / c1 / 1e6, c2 * 1e3, 2 * bytes / c2 / 1e6);
  }
  float m = timeit([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, 20);
  printf("hipMemcpy d2d: %.1f us (%.0f GB/s)\n", m * 1e3, 2 * bytes / m / 1e6);
  return 0;
}
