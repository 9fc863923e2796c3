// icp.hip -- one linearisation of point-to-plane projective ICP on gfx950.
// Replaces kfusion::device::ComputeIcpHelper::operator() (Points path) and its two kernels
// (yak/src/cuda/proj_icp.cu:74-102 find_coresp, :105-341 partial_reduce, :343-387 kernels).
// Citations are relative to /root/reference/.
//
// The reference reduces each of the 27 products with its own 256-wide shared-memory tree and
// __syncthreads pair (27 serial block reductions per 32 x 8 tile).  Here a lane walks several pixels
// and keeps the 27 sums (+ 6 outcome counters, exact in fp32) in registers; a wave folds them with
// cross-lane shuffles, the four waves of a workgroup meet once in LDS, and a second tiny kernel adds
// the workgroup partials.  Every addition happens in a fixed order, so the result is deterministic
// run to run (no float atomics).
#include "yakhip_internal.h"

namespace yakhip
{
namespace
{
constexpr int kValues = 33;        // 27 products + accepted + rejected{40,80,120,160,200}
constexpr int kPixelsPerLane = 4;  // a workgroup covers 1024 consecutive pixels

struct IcpParams
{
  float R[9], t[3];
  float fx, fy, cx, cy;  // level intrinsics, projective_icp.cpp:16-22
  float frows, fcols;    // projective_icp.cpp:200-201
  float min_cosine, dist2_thres;
  const uint8_t *vcurr, *ncurr, *vprev, *nprev;
  uint32_t vcurr_pitch, ncurr_pitch, vprev_pitch, nprev_pitch;
  int rows, cols;
};

__device__ __forceinline__ float4 f4_at(const uint8_t* base, uint32_t pitch, int y, int x)
{
  return ((const float4*)(base + (size_t)y * pitch))[x];
}

__global__ __launch_bounds__(256) void icp_accumulate_kernel(const IcpParams p, float* partial, int n_blocks)
{
  float acc[kValues];
#pragma unroll
  for (int k = 0; k < kValues; ++k)
    acc[k] = 0.f;

  const int n_pixels = p.rows * p.cols;
  const int base = (int)blockIdx.x * (256 * kPixelsPerLane) + (int)threadIdx.x;
  for (int i = 0; i < kPixelsPerLane; ++i)
  {
    const int pix = base + i * 256;
    if (pix >= n_pixels)
      break;
    const int y = pix / p.cols, x = pix - y * p.cols;

    // find_coresp, proj_icp.cu:74-102
    int code = 0;
    float row[7];
    const float4 s0 = f4_at(p.vcurr, p.vcurr_pitch, y, x);
    float sx = 0.f, sy = 0.f, sz = 0.f;
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f), nd = d;
    if (s0.x != s0.x)
      code = 40;
    else
    {
      sx = dot3(p.R[0], p.R[1], p.R[2], s0.x, s0.y, s0.z) + p.t[0];
      sy = dot3(p.R[3], p.R[4], p.R[5], s0.x, s0.y, s0.z) + p.t[1];
      sz = dot3(p.R[6], p.R[7], p.R[8], s0.x, s0.y, s0.z) + p.t[2];
      if (sz <= 0.f)
        code = 80;
      else
      {
        const float coox = __builtin_fmaf(p.fx, sx / sz, p.cx), cooy = __builtin_fmaf(p.fy, sy / sz, p.cy);
        if (!(coox >= 0.f && cooy >= 0.f && coox < p.fcols && cooy < p.frows))
          code = 80;
        else
        {
          const int px = (int)__builtin_floorf(coox), py = (int)__builtin_floorf(cooy);  // point-filtered fetch
          d = f4_at(p.vprev, p.vprev_pitch, py, px);
          if (d.x != d.x)
            code = 120;
          else
          {
            const float ex = sx - d.x, ey = sy - d.y, ez = sz - d.z;
            if (dot3(ex, ey, ez, ex, ey, ez) > p.dist2_thres)
              code = 160;
            else
            {
              const float4 nc = f4_at(p.ncurr, p.ncurr_pitch, y, x);
              const float nsx = dot3(p.R[0], p.R[1], p.R[2], nc.x, nc.y, nc.z);
              const float nsy = dot3(p.R[3], p.R[4], p.R[5], nc.x, nc.y, nc.z);
              const float nsz = dot3(p.R[6], p.R[7], p.R[8], nc.x, nc.y, nc.z);
              nd = f4_at(p.nprev, p.nprev_pitch, py, px);
              const float cosine = __builtin_fabsf(dot3(nsx, nsy, nsz, nd.x, nd.y, nd.z));
              if (cosine < p.min_cosine)
                code = 200;
            }
          }
        }
      }
    }

    if (code == 0)
    {
      // proj_icp.cu:354-359  row = [cross(s, n), n, dot(n, d - s)]
      row[0] = sy * nd.z - sz * nd.y;
      row[1] = sz * nd.x - sx * nd.z;
      row[2] = sx * nd.y - sy * nd.x;
      row[3] = nd.x, row[4] = nd.y, row[5] = nd.z;
      row[6] = dot3(nd.x, nd.y, nd.z, d.x - sx, d.y - sy, d.z - sz);
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
      acc[28] += code == 40 ? 1.f : 0.f;
      acc[29] += code == 80 ? 1.f : 0.f;
      acc[30] += code == 120 ? 1.f : 0.f;
      acc[31] += code == 160 ? 1.f : 0.f;
      acc[32] += code == 200 ? 1.f : 0.f;
    }
  }

  // wave64 butterfly, then the four waves through LDS
#pragma unroll
  for (int k = 0; k < kValues; ++k)
  {
    float v = acc[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
      v += __shfl_xor(v, off, 64);
    acc[k] = v;
  }
  __shared__ float lds[4][kValues];
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  if (lane == 0)
  {
#pragma unroll
    for (int k = 0; k < kValues; ++k)
      lds[wave][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < kValues)
  {
    const int k = (int)threadIdx.x;
    partial[(size_t)k * n_blocks + blockIdx.x] = ((lds[0][k] + lds[1][k]) + lds[2][k]) + lds[3][k];
  }
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
}  // namespace

int launch_icp_accumulate(yakhip_ctx* ctx, const float R[9], const float t[3], const float intr[4], int level,
                          float dist_thres, float angle_thres, const uint8_t* vcurr, size_t vcurr_pitch,
                          const uint8_t* ncurr, size_t ncurr_pitch, const uint8_t* vprev, size_t vprev_pitch,
                          const uint8_t* nprev, size_t nprev_pitch, int rows, int cols, float sums[27], int counts[6])
{
  IcpParams p;
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
  p.rows = rows, p.cols = cols;

  const long long n_pixels = (long long)rows * cols;
  if (n_pixels > 0x7fffffffll - 2048)
    return fail(YAKHIP_ERANGE, "icp: image too large");
  const int n_blocks = (int)((n_pixels + 256 * kPixelsPerLane - 1) / (256 * kPixelsPerLane));

  // context-owned scratch: block partials + the 33 results on the device, a pinned host mirror
  const size_t need = ((size_t)n_blocks * kValues + kValues) * sizeof(float);
  if (ctx->icp_scratch_bytes < need)
  {
    if (ctx->icp_scratch)
      YAKHIP_TRY(hipFree(ctx->icp_scratch));
    ctx->icp_scratch = nullptr;
    ctx->icp_scratch_bytes = 0;
    YAKHIP_TRY(hipMalloc(&ctx->icp_scratch, need));
    ctx->icp_scratch_bytes = need;
  }
  if (!ctx->icp_pinned)
    YAKHIP_TRY(hipHostMalloc(&ctx->icp_pinned, kValues * sizeof(float), hipHostMallocDefault));

  float* partial = (float*)ctx->icp_scratch;
  float* result = partial + (size_t)n_blocks * kValues;
  hipLaunchKernelGGL(icp_accumulate_kernel, dim3((unsigned)n_blocks), dim3(256), 0, ctx->stream, p, partial, n_blocks);
  YAKHIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(icp_final_kernel, dim3(kValues), dim3(64), 0, ctx->stream, partial, n_blocks, result);
  YAKHIP_TRY(hipGetLastError());
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
}  // namespace yakhip
