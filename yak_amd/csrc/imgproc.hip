// imgproc.hip -- depth pre-processing kernels for gfx950.
//
// compute_dists replaces kfusion::device::compute_dists (reference yak/src/cuda/imgproc.cu:259-283):
// depth in uint16 millimetres -> ray length in metres, rounded to IEEE half.  This image is the
// only measurement the integrate kernel sees, so its fp16 quantisation is part of the spec.
// Compiled with -ffp-contract=off: the source has no FMA here and neither has the oracle.
#include "yakhip_internal.h"

namespace yakhip
{
// One lane = two horizontally adjacent pixels (one 4-byte load, one 4-byte store) when the row
// base and pitch allow it; a wave64 then covers 128 pixels of one row.
template <int PX>
__global__ __launch_bounds__(256) void compute_dists_kernel(const uint8_t* depth, uint32_t depth_pitch,
                                                            uint8_t* dists, uint32_t dists_pitch, int rows, int cols,
                                                            float finvx, float finvy, float cx, float cy)
{
  const int x0 = (int)(blockIdx.x * 256u + threadIdx.x) * PX;
  const int y = (int)blockIdx.y;
  if (x0 >= cols || y >= rows)
    return;

  const uint16_t* drow = (const uint16_t*)(depth + (size_t)y * depth_pitch);
  uint16_t* orow = (uint16_t*)(dists + (size_t)y * dists_pitch);
  const float yl = ((float)y - cy) * finvy;

  uint32_t in_bits;
  if (PX == 2)
    in_bits = *(const uint32_t*)(drow + x0);
  else
    in_bits = drow[x0];

  uint32_t out_bits = 0;
#pragma unroll
  for (int i = 0; i < PX; ++i)
  {
    const float d = (float)((in_bits >> (16 * i)) & 0xffffu);
    const float xl = ((float)(x0 + i) - cx) * finvx;
    const float lambda = __builtin_sqrtf(xl * xl + yl * yl + 1.f);
    out_bits |= float_to_half_bits(d * lambda * 0.001f) << (16 * i);
  }

  if (PX == 2)
    *(uint32_t*)(orow + x0) = out_bits;
  else
    orow[x0] = (uint16_t)out_bits;
}

int launch_compute_dists(const yakhip_ctx* ctx, const uint8_t* depth, size_t depth_pitch, uint8_t* dists,
                         size_t dists_pitch, int rows, int cols, float fx, float fy, float cx, float cy)
{
  const float finvx = 1.f / fx, finvy = 1.f / fy;  // imgproc.cu:281, formed on the host
  const bool pairs = cols % 2 == 0 && depth_pitch % 4 == 0 && dists_pitch % 4 == 0 && (uintptr_t)depth % 4 == 0 &&
                     (uintptr_t)dists % 4 == 0;
  if (pairs)
  {
    const dim3 grid((unsigned)((cols / 2 + 255) / 256), (unsigned)rows);
    hipLaunchKernelGGL((compute_dists_kernel<2>), grid, dim3(256), 0, ctx->stream, depth, (uint32_t)depth_pitch,
                       dists, (uint32_t)dists_pitch, rows, cols, finvx, finvy, cx, cy);
  }
  else
  {
    const dim3 grid((unsigned)((cols + 255) / 256), (unsigned)rows);
    hipLaunchKernelGGL((compute_dists_kernel<1>), grid, dim3(256), 0, ctx->stream, depth, (uint32_t)depth_pitch,
                       dists, (uint32_t)dists_pitch, rows, cols, finvx, finvy, cx, cy);
  }
  return check(hipGetLastError(), "compute_dists_kernel launch");
}
}  // namespace yakhip
