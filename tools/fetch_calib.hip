// fetch_calib.hip -- what rocprofv3's FETCH_SIZE reports for dword-per-lane reads on gfx950.
// MI355X_MICROARCH.md calibrates FETCH_SIZE only for 16-byte-per-lane streaming reads (it reports half
// the bytes); marching cubes reads one dword per lane, a 256-byte row segment per wave, at a 252-byte
// pitch.  Three kernels over the same 512 MiB buffer (every byte is read at least once in each):
//   rows_aligned  : wave w reads bytes [256 w, 256 w + 256)                       -> 1.000x by construction
//   rows_pitch252 : a 2048-byte row is covered by 9 waves at a 252-byte pitch, x-neighbours in
//                   consecutive workgroups (different XCDs)                        -> what does the fabric see?
//   rows_pitch252_xcd : the same with each XCD given a contiguous run of workgroups
// build: hipcc --offload-arch=gfx950 -O3 -o tools/fetch_calib tools/fetch_calib.hip
// run:   rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- tools/fetch_calib
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

constexpr size_t kBytes = 512ull << 20;
constexpr int kRowBytes = 2048, kTiles = 9;  // one 512-voxel row, 63-cube tiles

__global__ __launch_bounds__(256) void rows_aligned(const uint32_t* buf, uint32_t* sink)
{
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t v = buf[wave * 64 + (threadIdx.x & 63)];
  if (v == 0x12345678u)
    *sink = v;
}

template <bool XCD>
__global__ __launch_bounds__(256) void rows_pitch252(const uint32_t* buf, uint32_t* sink)
{
  // workgroup = 4 consecutive rows of one x tile; consecutive workgroups = consecutive x tiles
  const unsigned n = gridDim.x, per = n / 8u;
  const unsigned wg = XCD && blockIdx.x < per * 8u ? (blockIdx.x % 8u) * per + blockIdx.x / 8u : blockIdx.x;
  const unsigned xt = wg % kTiles;
  const size_t row = (size_t)(wg / kTiles) * 4 + (threadIdx.x >> 6);
  const unsigned x = min(xt * 63u + (threadIdx.x & 63u), 511u);
  const uint32_t v = buf[row * (kRowBytes / 4) + x];
  if (v == 0x12345678u)
    *sink = v;
}

// read-only streaming ceilings: every workgroup reads PER consecutive 4 KiB (16 B/lane) or 1 KiB (4 B/lane)
// pieces, all loads issued before the first use
template <typename T, int PER>
__global__ __launch_bounds__(256) void stream_read(const T* buf, uint32_t* sink)
{
  const T* p = buf + (size_t)blockIdx.x * PER * 256 + threadIdx.x;
  T v[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i)
    v[i] = p[(size_t)i * 256];
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i)
    if constexpr (sizeof(T) == 16)
      acc |= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    else
      acc |= v[i];
  if (acc == 0x12345678u)
    *sink = acc;
}

int main()
{
  uint32_t *buf, *sink;
  if (hipMalloc(&buf, kBytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess)
    return 1;
  hipMemset(buf, 0, kBytes);
  hipDeviceSynchronize();
  const unsigned rows = (unsigned)(kBytes / kRowBytes);
  for (int rep = 0; rep < 3; ++rep)
  {
    hipLaunchKernelGGL(rows_aligned, dim3((unsigned)(kBytes / 1024)), dim3(256), 0, 0, buf, sink);
    hipLaunchKernelGGL(rows_pitch252<false>, dim3(rows / 4 * kTiles), dim3(256), 0, 0, buf, sink);
    hipLaunchKernelGGL(rows_pitch252<true>, dim3(rows / 4 * kTiles), dim3(256), 0, 0, buf, sink);
    hipLaunchKernelGGL((stream_read<uint4, 4>), dim3((unsigned)(kBytes / (4096 * 4))), dim3(256), 0, 0, (const uint4*)buf, sink);
    hipLaunchKernelGGL((stream_read<uint4, 16>), dim3((unsigned)(kBytes / (4096 * 16))), dim3(256), 0, 0, (const uint4*)buf, sink);
    hipLaunchKernelGGL((stream_read<uint32_t, 16>), dim3((unsigned)(kBytes / (1024 * 16))), dim3(256), 0, 0, buf, sink);
    hipLaunchKernelGGL((stream_read<uint32_t, 64>), dim3((unsigned)(kBytes / (1024 * 64))), dim3(256), 0, 0, buf, sink);
  }
  if (hipDeviceSynchronize() != hipSuccess)
    return 2;
  printf("buffer %zu bytes\n", kBytes);
  return 0;
}
