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
  const IcpState* state;  // non-null: take the pose from here (and do nothing once status != 0)
  float R[9], t[3];
  float fx, fy, cx, cy;  // level intrinsics, projective_icp.cpp:16-22
  float frows, fcols;    // projective_icp.cpp:200-201
  float min_cosine, dist2_thres;
  const uint8_t *vcurr, *ncurr, *vprev, *nprev;
  uint32_t vcurr_pitch, ncurr_pitch, vprev_pitch, nprev_pitch;
  int rows, cols;  // rows of the current maps to walk, starting at row0 (frows / fcols describe the whole image)
  int row0;
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

  float R[9], t[3];
  if (p.state)  // wave-uniform: scalar loads
  {
    if (p.state->status != 0)
      return;
#pragma unroll
    for (int i = 0; i < 9; ++i)
      R[i] = p.state->R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      t[i] = p.state->t[i];
  }
  else
  {
#pragma unroll
    for (int i = 0; i < 9; ++i)
      R[i] = p.R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      t[i] = p.t[i];
  }

  const int n_pixels = p.rows * p.cols;
  const int base = (int)blockIdx.x * (256 * kPixelsPerLane) + (int)threadIdx.x;
  for (int i = 0; i < kPixelsPerLane; ++i)
  {
    const int pix = base + i * 256;
    if (pix >= n_pixels)
      break;
    const int yr = pix / p.cols, x = pix - yr * p.cols, y = yr + p.row0;

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
      sx = dot3(R[0], R[1], R[2], s0.x, s0.y, s0.z) + t[0];
      sy = dot3(R[3], R[4], R[5], s0.x, s0.y, s0.z) + t[1];
      sz = dot3(R[6], R[7], R[8], s0.x, s0.y, s0.z) + t[2];
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
              const float nsx = dot3(R[0], R[1], R[2], nc.x, nc.y, nc.z);
              const float nsy = dot3(R[3], R[4], R[5], nc.x, nc.y, nc.z);
              const float nsz = dot3(R[6], R[7], R[8], nc.x, nc.y, nc.z);
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
// Final reduction + solve + pose update of one linearisation, without leaving the device: the
// reference goes back to the host after every one of its 19 linearisations per frame
// (projective_icp.cpp:206-229: stream sync, cv::determinant, cv::solve, compose).
// One workgroup: wave w adds the workgroup partials of values w, w + 16, w + 32 in a fixed order; then
// one lane solves the 6 x 6 system.  The normal matrix is symmetric positive definite whenever the
// reference's own checks pass, so a Cholesky factorisation in double gives the solution cv::solve's SVD
// gives, as long as the SVD would not truncate a singular value.  That is certified, not assumed:
// lambda_min >= det / (trace / 5)^5 (AM-GM on the other five eigenvalues) must exceed OpenCV's cut-off
// 2 * FLT_EPSILON * trace, and det must be clear of the 1e-15 test; anything else sets status = 2 and the
// host's LU / SVD path (solve6.hpp) redoes the estimate.
__global__ __launch_bounds__(1024) void icp_solve_kernel(const float* partial, int n_blocks, IcpState* st, float* sums_out)
{
  __shared__ float sums[kValues];
  if (st->status != 0)
    return;
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  for (int k = wave; k < kValues; k += 16)
  {
    float v = 0.f;
    for (int b = lane; b < n_blocks; b += 64)
      v += partial[(size_t)k * n_blocks + b];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
      v += __shfl_xor(v, off, 64);
    if (lane == 0)
      sums[k] = v;
  }
  __syncthreads();
  if (threadIdx.x != 0)
    return;
  for (int k = 0; k < kValues; ++k)
    sums_out[k] = sums[k];

  // StreamHelper::get, projective_icp.cpp:42-61.  Every loop below has constant bounds and is unrolled,
  // so the 6 x 6 system lives in registers (one lane, double precision).
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

  // Cholesky A = L L^T (lower triangle in place), det = prod L_ii^2
  double det = 1.0;
  bool spd = trace == trace;
#pragma unroll
  for (int j = 0; j < 6; ++j)
  {
    double d = A[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k)
      d -= A[j][k] * A[j][k];
    spd = spd && d > 0.0;
    const double l = sqrt(spd ? d : 1.0);
    A[j][j] = l;
    det *= d;
#pragma unroll
    for (int i = j + 1; i < 6; ++i)
    {
      double v = A[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k)
        v -= A[i][k] * A[j][k];
      A[i][j] = v / l;
    }
  }
  const double m = trace / 5.0;
  const bool certified = spd && det > 1e-12 && det > 2.0 * 1.1920928955078125e-7 * trace * (m * m * m * m * m) * 4.0;
  if (!certified)
  {
    st->status = 2;
    return;
  }
  double y[6], r[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
  {
    double v = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k)
      v -= A[i][k] * y[k];
    y[i] = v / A[i][i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i)
  {
    double v = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k)
      v -= A[k][i] * r[k];
    r[i] = v / A[i][i];
  }

  // Tinc = Affine3f(rvec, tvec) (Rodrigues), corrected = Tinc * corrected: projective_icp.cpp:227-228,
  // in the float arithmetic of the host type (include/yak/kfusion/compat/linalg.hpp)
  const float rv[3] = { (float)r[0], (float)r[1], (float)r[2] };
  const float tinc[3] = { (float)r[3], (float)r[4], (float)r[5] };
  float Ri[9] = { 1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f };
  const double th = sqrt((double)rv[0] * rv[0] + (double)rv[1] * rv[1] + (double)rv[2] * rv[2]);
  if (th > 2.220446049250313e-16)
  {
    const double kx = rv[0] / th, ky = rv[1] / th, kz = rv[2] / th, c = cos(th), s = sin(th), v = 1.0 - c;
    Ri[0] = (float)(c + kx * kx * v), Ri[1] = (float)(kx * ky * v - kz * s), Ri[2] = (float)(kx * kz * v + ky * s);
    Ri[3] = (float)(ky * kx * v + kz * s), Ri[4] = (float)(c + ky * ky * v), Ri[5] = (float)(ky * kz * v - kx * s);
    Ri[6] = (float)(kz * kx * v - ky * s), Ri[7] = (float)(kz * ky * v + kx * s), Ri[8] = (float)(c + kz * kz * v);
  }
  float Rn[9], tn[3];
#pragma unroll
  for (int rr = 0; rr < 3; ++rr)
  {
#pragma unroll
    for (int cc = 0; cc < 3; ++cc)
      Rn[3 * rr + cc] = Ri[3 * rr + 0] * st->R[0 + cc] + Ri[3 * rr + 1] * st->R[3 + cc] + Ri[3 * rr + 2] * st->R[6 + cc];
    tn[rr] = (Ri[3 * rr + 0] * st->t[0] + Ri[3 * rr + 1] * st->t[1] + Ri[3 * rr + 2] * st->t[2]) + tinc[rr];
  }
#pragma unroll
  for (int i = 0; i < 9; ++i)
    st->R[i] = Rn[i];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    st->t[i] = tn[i];
  st->iterations += 1;
}
}  // namespace

static int icp_scratch(yakhip_ctx* ctx, size_t need);

int launch_icp_accumulate(yakhip_ctx* ctx, const float R[9], const float t[3], const float intr[4], int level,
                          float dist_thres, float angle_thres, const uint8_t* vcurr, size_t vcurr_pitch,
                          const uint8_t* ncurr, size_t ncurr_pitch, const uint8_t* vprev, size_t vprev_pitch,
                          const uint8_t* nprev, size_t nprev_pitch, int rows, int cols, float sums[27], int counts[6],
                          int row_begin, int row_end, float* sums_dev)
{
  IcpParams p;
  p.state = nullptr;
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
  if ((long long)rows * cols > 0x7fffffffll - 2048)
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
  hipLaunchKernelGGL(icp_accumulate_kernel, dim3((unsigned)n_blocks), dim3(256), 0, ctx->stream, p, partial, n_blocks);
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
    YAKHIP_TRY(hipHostMalloc(&ctx->icp_pinned, 256, hipHostMallocDefault));
  return 0;
}

// The whole coarse-to-fine schedule of ProjectiveICP::estimateTransform (projective_icp.cpp:175-232)
// enqueued back to back: per linearisation one accumulate launch and one reduce+solve launch, the pose
// handed from one to the next in device memory; the host waits once, at the end.
// (Folding the reduce+solve into the accumulate launch -- last workgroup by ticket -- was measured and is
// slower, 0.52 vs 0.34 ms per frame: the device-scope release/acquire fences every workgroup then needs
// write back and invalidate the XCD's L2, which costs more than the second launch.)
int launch_icp_estimate(yakhip_ctx* ctx, const float R[9], const float t[3], const float intr[4], float dist_thres,
                        float angle_thres, const yakhip_icp_level* levels, int n_levels, float R_out[9], float t_out[3],
                        int* status_out, int* iterations_out)
{
  int max_blocks = 1;
  for (int l = 0; l < n_levels; ++l)
  {
    const long long n_pixels = (long long)levels[l].rows * levels[l].cols;
    if (n_pixels > 0x7fffffffll - 2048)
      return fail(YAKHIP_ERANGE, "icp: image too large");
    const int nb = (int)((n_pixels + 256 * kPixelsPerLane - 1) / (256 * kPixelsPerLane));
    if (nb > max_blocks)
      max_blocks = nb;
  }
  const size_t partial_bytes = ((size_t)max_blocks * kValues + kValues) * sizeof(float);
  const size_t state_off = (partial_bytes + 255u) & ~(size_t)255u;
  int rc = icp_scratch(ctx, state_off + sizeof(IcpState));
  if (rc != 0)
    return rc;
  float* partial = (float*)ctx->icp_scratch;
  float* sums_dev = partial + (size_t)max_blocks * kValues;
  IcpState* st = (IcpState*)((char*)ctx->icp_scratch + state_off);

  IcpState init;
  for (int i = 0; i < 9; ++i)
    init.R[i] = R[i];
  for (int i = 0; i < 3; ++i)
    init.t[i] = t[i];
  init.status = 0;
  init.iterations = 0;
  IcpState* host = (IcpState*)ctx->icp_pinned;
  *host = init;
  YAKHIP_TRY(hipMemcpyAsync(st, host, sizeof(IcpState), hipMemcpyHostToDevice, ctx->stream));

  for (int l = n_levels - 1; l >= 0; --l)
  {
    const yakhip_icp_level& L = levels[l];
    IcpParams p;
    p.state = st;
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
    for (int it = 0; it < L.iterations; ++it)
    {
      hipLaunchKernelGGL(icp_accumulate_kernel, dim3((unsigned)n_blocks), dim3(256), 0, ctx->stream, p, partial, n_blocks);
      hipLaunchKernelGGL(icp_solve_kernel, dim3(1), dim3(1024), 0, ctx->stream, partial, n_blocks, st, sums_dev);
    }
    YAKHIP_TRY(hipGetLastError());
  }
  YAKHIP_TRY(hipMemcpyAsync(host, st, sizeof(IcpState), hipMemcpyDeviceToHost, ctx->stream));
  YAKHIP_TRY(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 9; ++i)
    R_out[i] = host->R[i];
  for (int i = 0; i < 3; ++i)
    t_out[i] = host->t[i];
  *status_out = host->status;
  *iterations_out = host->iterations;
  return 0;
}
}  // namespace yakhip
