// frameproc.hip -- per-frame depth pre-processing for the tracking side of the KinFu loop, gfx950.
// Replaces kfusion::device::{bilateralFilter, truncateDepth, depthPyr, computePointNormals,
// resizePointsNormals} (yak/src/cuda/imgproc.cu:10-56, 66-85, 94-136, 210-250, 348-394).
// Citations are relative to /root/reference/.
//
// These are small image-space kernels (640 x 480 and its pyramid): they are latency-, not
// bandwidth-bound, so the only design rule is "wide rows, one pass": a wave covers 64 consecutive
// pixels of a row, every tap goes through the vector L1, and every statement keeps the reference's
// fp32 expression order (this TU is built with -ffp-contract=off).
#include "yakhip_internal.h"

namespace yakhip
{
namespace
{
__device__ __forceinline__ int u16_at(const uint8_t* base, uint32_t pitch, int y, int x)
{
  return ((const unsigned short*)(base + (size_t)y * pitch))[x];
}

// imgproc.cu:10-42.  KSZ > 0: the kernel size is known at compile time, and a workgroup whose 64 x 4 tile and
// window lie inside the image takes the taps of a window as one batch of loads (the general loop has per-lane
// bounds, so every tap waits for its own load: 49 trips to the L1 in a row, 20 us a frame) and then the same
// statements in the same order.  KSZ == 0: any size, the loop as written.
__device__ __forceinline__ void bilateral_tap(int value, int depth, int dx, int dy, float sigma_spatial2_inv_half,
                                              float sigma_depth2_inv_half, float& sum1, float& sum2)
{
  const float space2 = (float)(dx * dx + dy * dy);
  const int dd = value - depth;
  const float color2 = (float)(int)((unsigned)dd * (unsigned)dd);  // int product, wraps as the source's does
  const float weight = expf(-(space2 * sigma_spatial2_inv_half + color2 * sigma_depth2_inv_half));
  sum1 += (float)depth * weight;
  sum2 += weight;
}

template <int KSZ>
__global__ __launch_bounds__(256) void bilateral_kernel(const uint8_t* src, uint32_t src_pitch, uint8_t* dst,
                                                        uint32_t dst_pitch, int rows, int cols, int ksz,
                                                        float sigma_spatial2_inv_half, float sigma_depth2_inv_half)
{
  const int x = (int)(blockIdx.x * 64u + (threadIdx.x & 63u));
  const int y = (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
  float sum1 = 0.f, sum2 = 0.f;
  bool done = false;
  if constexpr (KSZ > 0)
  {
    const int bx0 = (int)(blockIdx.x * 64u), by0 = (int)(blockIdx.y * 4u);
    // every pixel of the tile: max(. - KSZ / 2, 0) and min(. - KSZ / 2 + KSZ, size - 1) do not clip
    if (bx0 - KSZ / 2 >= 0 && bx0 + 63 - KSZ / 2 + KSZ <= cols - 1 && by0 - KSZ / 2 >= 0 &&
        by0 + 3 - KSZ / 2 + KSZ <= rows - 1)  // workgroup-uniform
    {
      const int value = u16_at(src, src_pitch, y, x);
      int d[KSZ][KSZ];
#pragma unroll
      for (int j = 0; j < KSZ; ++j)
#pragma unroll
        for (int i = 0; i < KSZ; ++i)
          d[j][i] = u16_at(src, src_pitch, y - KSZ / 2 + j, x - KSZ / 2 + i);
#pragma unroll
      for (int j = 0; j < KSZ; ++j)
#pragma unroll
        for (int i = 0; i < KSZ; ++i)
          bilateral_tap(value, d[j][i], KSZ / 2 - i, KSZ / 2 - j, sigma_spatial2_inv_half, sigma_depth2_inv_half, sum1,
                        sum2);
      done = true;
    }
  }
  if (!done)
  {
    if (x >= cols || y >= rows)
      return;
    const int value = u16_at(src, src_pitch, y, x);
    const int tx = min(x - ksz / 2 + ksz, cols - 1);
    const int ty = min(y - ksz / 2 + ksz, rows - 1);
    for (int cy = max(y - ksz / 2, 0); cy < ty; ++cy)
      for (int cx = max(x - ksz / 2, 0); cx < tx; ++cx)
        bilateral_tap(value, u16_at(src, src_pitch, cy, cx), x - cx, y - cy, sigma_spatial2_inv_half,
                      sigma_depth2_inv_half, sum1, sum2);
  }
  const float q = sum1 / sum2;
  // __float2int_rn: round half even; NaN (an all-underflow window) -> 0
  const int r = q != q ? 0 : (int)__builtin_rintf(q);
  ((unsigned short*)(dst + (size_t)y * dst_pitch))[x] = (unsigned short)r;
}

// imgproc.cu:66-74
__global__ __launch_bounds__(256) void truncate_depth_kernel(uint8_t* depth, uint32_t pitch, int rows, int cols,
                                                             unsigned short max_dist)
{
  const int x = (int)(blockIdx.x * 64u + (threadIdx.x & 63u));
  const int y = (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
  if (x >= cols || y >= rows)
    return;
  unsigned short* p = (unsigned short*)(depth + (size_t)y * pitch) + x;
  if (*p > max_dist)
    *p = 0;
}

// imgproc.cu:94-123
__global__ __launch_bounds__(256) void pyramid_kernel(const uint8_t* src, uint32_t src_pitch, int src_rows, int src_cols,
                                                      uint8_t* dst, uint32_t dst_pitch, int rows, int cols,
                                                      float sigma_depth_mult3)
{
  const int x = (int)(blockIdx.x * 64u + (threadIdx.x & 63u));
  const int y = (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
  if (x >= cols || y >= rows)
    return;
  const int D = 5;
  const int center = u16_at(src, src_pitch, 2 * y, 2 * x);
  const int tx = min(2 * x - D / 2 + D, src_cols - 1);
  const int ty = min(2 * y - D / 2 + D, src_rows - 1);
  int sum = 0, count = 0;
  for (int cy = max(0, 2 * y - D / 2); cy < ty; ++cy)
    for (int cx = max(0, 2 * x - D / 2); cx < tx; ++cx)
    {
      const int val = u16_at(src, src_pitch, cy, cx);
      if ((float)abs(val - center) < sigma_depth_mult3)
      {
        sum += val;
        ++count;
      }
    }
  ((unsigned short*)(dst + (size_t)y * dst_pitch))[x] = (unsigned short)(count == 0 ? 0 : sum / count);
}

// device.hpp:57-62
__device__ __forceinline__ void reproj(float finvx, float finvy, float cx, float cy, int u, int v, float z, float& ox,
                                       float& oy)
{
  ox = z * ((float)u - cx) * finvx;
  oy = z * ((float)v - cy) * finvy;
}

// imgproc.cu:210-239
__global__ __launch_bounds__(256) void points_normals_kernel(float finvx, float finvy, float cx, float cy,
                                                             const uint8_t* depth, uint32_t depth_pitch, int rows,
                                                             int cols, uint8_t* points, uint32_t points_pitch,
                                                             uint8_t* normals, uint32_t normals_pitch)
{
  const int x = (int)(blockIdx.x * 64u + (threadIdx.x & 63u));
  const int y = (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
  if (x >= cols || y >= rows)
    return;
  const float qnan = __builtin_nanf("");
  float4 p = make_float4(qnan, qnan, qnan, qnan), n = p;
  if (x < cols - 1 && y < rows - 1)
  {
    const float z00 = (float)u16_at(depth, depth_pitch, y, x) * 0.001f;
    const float z01 = (float)u16_at(depth, depth_pitch, y, x + 1) * 0.001f;
    const float z10 = (float)u16_at(depth, depth_pitch, y + 1, x) * 0.001f;
    if (z00 * z01 * z10 != 0.f)
    {
      float v00x, v00y, v01x, v01y, v10x, v10y;
      reproj(finvx, finvy, cx, cy, x, y, z00, v00x, v00y);
      reproj(finvx, finvy, cx, cy, x + 1, y, z01, v01x, v01y);
      reproj(finvx, finvy, cx, cy, x, y + 1, z10, v10x, v10y);
      const float ax = v01x - v00x, ay = v01y - v00y, az = z01 - z00;
      const float bx = v10x - v00x, by = v10y - v00y, bz = z10 - z00;
      const float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;  // temp_utils.hpp:105-108
      const float s = 1.0f / __builtin_sqrtf(dot3(nx, ny, nz, nx, ny, nz));             // normalized, :103
      n = make_float4(-(nx * s), -(ny * s), -(nz * s), 0.f);
      p = make_float4(v00x, v00y, z00, 0.f);
    }
  }
  ((float4*)(points + (size_t)y * points_pitch))[x] = p;
  ((float4*)(normals + (size_t)y * normals_pitch))[x] = n;
}

__device__ __forceinline__ float4 f4_at(const uint8_t* base, uint32_t pitch, int y, int x)
{
  return ((const float4*)(base + (size_t)y * pitch))[x];
}

// imgproc.cu:348-380: pixel (y, x) of the half-size maps from the 2 x 2 block of the source
__device__ __forceinline__ void resize_pixel(const uint8_t* vsrc, uint32_t vsrc_pitch, const uint8_t* nsrc, uint32_t nsrc_pitch,
                                             int y, int x, float4& v, float4& n)
{
  const float qnan = __builtin_nanf("");
  v = make_float4(qnan, qnan, qnan, 0.f), n = v;
  const int xs = 2 * x, ys = 2 * y;
  const float4 d00 = f4_at(vsrc, vsrc_pitch, ys + 0, xs + 0);
  const float4 d01 = f4_at(vsrc, vsrc_pitch, ys + 0, xs + 1);
  const float4 d10 = f4_at(vsrc, vsrc_pitch, ys + 1, xs + 0);
  const float4 d11 = f4_at(vsrc, vsrc_pitch, ys + 1, xs + 1);
  const float prod = d00.x * d01.x * d10.x * d11.x;
  if (!(prod != prod))
  {
    v = make_float4((d00.x + d01.x + d10.x + d11.x) * 0.25f, (d00.y + d01.y + d10.y + d11.y) * 0.25f,
                    (d00.z + d01.z + d10.z + d11.z) * 0.25f, 0.f);
    const float4 n00 = f4_at(nsrc, nsrc_pitch, ys + 0, xs + 0);
    const float4 n01 = f4_at(nsrc, nsrc_pitch, ys + 0, xs + 1);
    const float4 n10 = f4_at(nsrc, nsrc_pitch, ys + 1, xs + 0);
    const float4 n11 = f4_at(nsrc, nsrc_pitch, ys + 1, xs + 1);
    n = make_float4((n00.x + n01.x + n10.x + n11.x) * 0.25f, (n00.y + n01.y + n10.y + n11.y) * 0.25f,
                    (n00.z + n01.z + n10.z + n11.z) * 0.25f, 0.f);
  }
}

__global__ __launch_bounds__(256) void resize_points_normals_kernel(const uint8_t* vsrc, uint32_t vsrc_pitch,
                                                                    const uint8_t* nsrc, uint32_t nsrc_pitch,
                                                                    uint8_t* vdst, uint32_t vdst_pitch, uint8_t* ndst,
                                                                    uint32_t ndst_pitch, int rows, int cols)
{
  const int x = (int)(blockIdx.x * 64u + (threadIdx.x & 63u));
  const int y = (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
  if (x >= cols || y >= rows)
    return;
  float4 v, n;
  resize_pixel(vsrc, vsrc_pitch, nsrc, nsrc_pitch, y, x, v, n);
  ((float4*)(vdst + (size_t)y * vdst_pitch))[x] = v;
  ((float4*)(ndst + (size_t)y * ndst_pitch))[x] = n;
}

// Two pyramid levels in one launch (the model maps after a ray cast: kinfu.cpp:294-306 calls resizePointsNormals
// level by level; here the second call's launch and the gap in front of it are saved): a lane owns one pixel of the
// quarter-size maps, computes the 2 x 2 half-size pixels under it as resize_pixel does, stores them, and
// averages them again -- the arithmetic of the second call on the values of the first, so both levels are bit for
// bit those of two calls.  Half-size pixels that no quarter-size pixel covers (odd half-size dimensions) are
// computed by the lanes just beyond the quarter-size image.
__global__ __launch_bounds__(256) void resize_points_normals2_kernel(const uint8_t* vsrc, uint32_t vsrc_pitch, const uint8_t* nsrc,
                                                                     uint32_t nsrc_pitch, uint8_t* v1, uint32_t v1_pitch,
                                                                     uint8_t* n1, uint32_t n1_pitch, int rows1, int cols1,
                                                                     uint8_t* v2, uint32_t v2_pitch, uint8_t* n2,
                                                                     uint32_t n2_pitch, int rows2, int cols2)
{
  const int x = (int)(blockIdx.x * 64u + (threadIdx.x & 63u));
  const int y = (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
  float4 v[2][2], n[2][2];
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx)
    {
      const int y1 = 2 * y + dy, x1 = 2 * x + dx;
      if (y1 < rows1 && x1 < cols1)
      {
        resize_pixel(vsrc, vsrc_pitch, nsrc, nsrc_pitch, y1, x1, v[dy][dx], n[dy][dx]);
        ((float4*)(v1 + (size_t)y1 * v1_pitch))[x1] = v[dy][dx];
        ((float4*)(n1 + (size_t)y1 * n1_pitch))[x1] = n[dy][dx];
      }
    }
  if (x >= cols2 || y >= rows2)
    return;
  const float qnan = __builtin_nanf("");
  float4 vo = make_float4(qnan, qnan, qnan, 0.f), no = vo;
  const float prod = v[0][0].x * v[0][1].x * v[1][0].x * v[1][1].x;
  if (!(prod != prod))
  {
    vo = make_float4((v[0][0].x + v[0][1].x + v[1][0].x + v[1][1].x) * 0.25f, (v[0][0].y + v[0][1].y + v[1][0].y + v[1][1].y) * 0.25f,
                     (v[0][0].z + v[0][1].z + v[1][0].z + v[1][1].z) * 0.25f, 0.f);
    no = make_float4((n[0][0].x + n[0][1].x + n[1][0].x + n[1][1].x) * 0.25f, (n[0][0].y + n[0][1].y + n[1][0].y + n[1][1].y) * 0.25f,
                     (n[0][0].z + n[0][1].z + n[1][0].z + n[1][1].z) * 0.25f, 0.f);
  }
  ((float4*)(v2 + (size_t)y * v2_pitch))[x] = vo;
  ((float4*)(n2 + (size_t)y * n2_pitch))[x] = no;
}

// ---- debug display: imgproc.cu:400-561 ----

// __saturatef: NaN -> 0, clamp to [0, 1]
__device__ __forceinline__ float saturate(float v) { return v > 0.f ? (v < 1.f ? v : 1.f) : 0.f; }

// v^20 by squaring -- the same five products in the oracle, so the two agree bit for bit; the reference's
// __powf (imgproc.cu:441) is the hardware's exp2(20 * log2 v) approximation, which no CPU or AMD routine
// reproduces to the bit
__device__ __forceinline__ float pow20(float v)
{
  const float v2 = v * v, v4 = v2 * v2, v5 = v4 * v, v10 = v5 * v5;
  return v10 * v10;
}

__device__ __forceinline__ uint32_t grey(float c) { return (uint32_t)(saturate(c) * 255.f) * 0x010101u; }

// :414-418 / :467-471 -- the backdrop where there is no surface: a vertical gradient, stored b, g, r, 0
__device__ __forceinline__ uint32_t backdrop(int y, int rows)
{
  const float w = (float)y / (float)rows;
  const float b = (4.f / 255.f) * (1 - w) + (236.f / 255.f) * w;
  const float g = (2.f / 255.f) * (1 - w) + (120.f / 255.f) * w;
  return (uint32_t)(saturate(b) * 255.f) | (uint32_t)(saturate(g) * 255.f) << 8 | (uint32_t)(saturate(g) * 255.f) << 16;
}

// :422-443 / :475-496 -- Phong with Ka 0.3, Kd 0.5, Ks 0.2, power 20, white light at light_pose, eye at 0
__device__ __forceinline__ uint32_t phong(float px, float py, float pz, float4 n, float lx, float ly, float lz)
{
  float Lx = lx - px, Ly = ly - py, Lz = lz - pz;
  float s = 1.0f / __builtin_sqrtf(dot3(Lx, Ly, Lz, Lx, Ly, Lz));
  Lx *= s, Ly *= s, Lz *= s;
  float Vx = 0.f - px, Vy = 0.f - py, Vz = 0.f - pz;
  s = 1.0f / __builtin_sqrtf(dot3(Vx, Vy, Vz, Vx, Vy, Vz));
  Vx *= s, Vy *= s, Vz *= s;
  const float nl = dot3(n.x, n.y, n.z, Lx, Ly, Lz);
  float Rx = 2 * n.x * nl - Lx, Ry = 2 * n.y * nl - Ly, Rz = 2 * n.z * nl - Lz;
  s = 1.0f / __builtin_sqrtf(dot3(Rx, Ry, Rz, Rx, Ry, Rz));
  Rx *= s, Ry *= s, Rz *= s;
  const float rv = dot3(Rx, Ry, Rz, Vx, Vy, Vz);
  const float Ix = 0.3f + 0.5f * (nl > 0.f ? nl : 0.f) + 0.2f * pow20(rv > 0.f ? rv : 0.f);
  return grey(Ix);
}

// DEPTH = false: imgproc.cu:453-504 (points), true: :400-451 (depth in mm through the reprojector)
template <bool DEPTH>
__global__ __launch_bounds__(256) void render_image_kernel(const uint8_t* src, uint32_t src_pitch, const uint8_t* normals,
                                                           uint32_t normals_pitch, int rows, int cols, float finvx,
                                                           float finvy, float cx, float cy, float lx, float ly, float lz,
                                                           uint8_t* image, uint32_t image_pitch)
{
  const int x = (int)(blockIdx.x * 64u + (threadIdx.x & 63u));
  const int y = (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
  if (x >= cols || y >= rows)
    return;
  uint32_t out;
  if (DEPTH)
  {
    const int d = u16_at(src, src_pitch, y, x);
    if (d == 0)
      out = backdrop(y, rows);
    else
    {
      const float z = (float)d * 0.001f;
      float px, py;
      reproj(finvx, finvy, cx, cy, x, y, z, px, py);
      out = phong(px, py, z, f4_at(normals, normals_pitch, y, x), lx, ly, lz);
    }
  }
  else
  {
    const float4 p = f4_at(src, src_pitch, y, x);
    out = p.x != p.x ? backdrop(y, rows) : phong(p.x, p.y, p.z, f4_at(normals, normals_pitch, y, x), lx, ly, lz);
  }
  ((uint32_t*)(image + (size_t)y * image_pitch))[x] = out;
}

// float -> unsigned char as the GPU conversion of imgproc.cu:545-547 does it: truncate, NaN -> 0; a unit
// normal gives 38 .. 216, anything else is clamped to the byte
__device__ __forceinline__ uint32_t tangent_byte(float n, float k)
{
  const float v = (5.f - n * k) * 25.5f;
  return v > 0.f ? (v < 255.f ? (uint32_t)v : 255u) : 0u;
}

// imgproc.cu:530-550
__global__ __launch_bounds__(256) void tangent_colors_kernel(const uint8_t* normals, uint32_t normals_pitch, int rows,
                                                             int cols, uint8_t* image, uint32_t image_pitch)
{
  const int x = (int)(blockIdx.x * 64u + (threadIdx.x & 63u));
  const int y = (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
  if (x >= cols || y >= rows)
    return;
  const float4 n = f4_at(normals, normals_pitch, y, x);
  const uint32_t r = tangent_byte(n.x, 3.5f), g = tangent_byte(n.y, 2.5f), b = tangent_byte(n.z, 3.5f);
  ((uint32_t*)(image + (size_t)y * image_pitch))[x] = b | g << 8 | r << 16;
}

inline dim3 image_grid(int rows, int cols) { return dim3((unsigned)((cols + 63) / 64), (unsigned)((rows + 3) / 4)); }
}  // namespace

int launch_bilateral(const yakhip_ctx* ctx, const uint8_t* src, size_t src_pitch, uint8_t* dst, size_t dst_pitch,
                     int rows, int cols, int kernel_size, float sigma_spatial, float sigma_depth)
{
  sigma_depth *= 1000;  // imgproc.cu:48 metres -> mm
  const float ss = 0.5f / (sigma_spatial * sigma_spatial), sd = 0.5f / (sigma_depth * sigma_depth);  // :54
  if (kernel_size == 7)  // the size KinFu uses (kinfu.cpp:28)
    hipLaunchKernelGGL(bilateral_kernel<7>, image_grid(rows, cols), dim3(256), 0, ctx->stream, src, (uint32_t)src_pitch,
                       dst, (uint32_t)dst_pitch, rows, cols, kernel_size, ss, sd);
  else
    hipLaunchKernelGGL(bilateral_kernel<0>, image_grid(rows, cols), dim3(256), 0, ctx->stream, src, (uint32_t)src_pitch,
                       dst, (uint32_t)dst_pitch, rows, cols, kernel_size, ss, sd);
  return check(hipGetLastError(), "bilateral_kernel launch");
}

int launch_truncate_depth(const yakhip_ctx* ctx, uint8_t* depth, size_t pitch, int rows, int cols, float max_dist)
{
  const float mm = max_dist * 1000.f;  // imgproc.cu:83, saturating truncation as the GPU conversion does
  const unsigned short lim = (unsigned short)(mm >= 65535.f ? 65535 : (mm > 0.f ? (int)mm : 0));
  hipLaunchKernelGGL(truncate_depth_kernel, image_grid(rows, cols), dim3(256), 0, ctx->stream, depth, (uint32_t)pitch,
                     rows, cols, lim);
  return check(hipGetLastError(), "truncate_depth_kernel launch");
}

int launch_depth_pyramid(const yakhip_ctx* ctx, const uint8_t* src, size_t src_pitch, int src_rows, int src_cols,
                         uint8_t* dst, size_t dst_pitch, float sigma_depth)
{
  const int rows = src_rows / 2, cols = src_cols / 2;  // imgproc.cpp:17
  hipLaunchKernelGGL(pyramid_kernel, image_grid(rows, cols), dim3(256), 0, ctx->stream, src, (uint32_t)src_pitch,
                     src_rows, src_cols, dst, (uint32_t)dst_pitch, rows, cols, sigma_depth * 1000 * 3);  // :129,:134
  return check(hipGetLastError(), "pyramid_kernel launch");
}

int launch_points_normals(const yakhip_ctx* ctx, const float intr[4], const uint8_t* depth, size_t depth_pitch, int rows,
                          int cols, uint8_t* points, size_t points_pitch, uint8_t* normals, size_t normals_pitch)
{
  hipLaunchKernelGGL(points_normals_kernel, image_grid(rows, cols), dim3(256), 0, ctx->stream, 1.f / intr[0],
                     1.f / intr[1], intr[2], intr[3], depth, (uint32_t)depth_pitch, rows, cols, points,
                     (uint32_t)points_pitch, normals, (uint32_t)normals_pitch);
  return check(hipGetLastError(), "points_normals_kernel launch");
}

int launch_resize_points_normals(const yakhip_ctx* ctx, const uint8_t* points, size_t points_pitch,
                                 const uint8_t* normals, size_t normals_pitch, int src_rows, int src_cols,
                                 uint8_t* points_out, size_t points_out_pitch, uint8_t* normals_out,
                                 size_t normals_out_pitch)
{
  const int rows = src_rows / 2, cols = src_cols / 2;  // imgproc.cu:386-387
  hipLaunchKernelGGL(resize_points_normals_kernel, image_grid(rows, cols), dim3(256), 0, ctx->stream, points,
                     (uint32_t)points_pitch, normals, (uint32_t)normals_pitch, points_out, (uint32_t)points_out_pitch,
                     normals_out, (uint32_t)normals_out_pitch, rows, cols);
  return check(hipGetLastError(), "resize_points_normals_kernel launch");
}

int launch_resize_points_normals2(const yakhip_ctx* ctx, const uint8_t* points, size_t points_pitch, const uint8_t* normals,
                                  size_t normals_pitch, int src_rows, int src_cols, uint8_t* points1, size_t points1_pitch,
                                  uint8_t* normals1, size_t normals1_pitch, uint8_t* points2, size_t points2_pitch,
                                  uint8_t* normals2, size_t normals2_pitch)
{
  const int rows1 = src_rows / 2, cols1 = src_cols / 2, rows2 = rows1 / 2, cols2 = cols1 / 2;  // imgproc.cu:386-387, twice
  hipLaunchKernelGGL(resize_points_normals2_kernel, image_grid((rows1 + 1) / 2, (cols1 + 1) / 2), dim3(256), 0, ctx->stream, points,
                     (uint32_t)points_pitch, normals, (uint32_t)normals_pitch, points1, (uint32_t)points1_pitch, normals1,
                     (uint32_t)normals1_pitch, rows1, cols1, points2, (uint32_t)points2_pitch, normals2,
                     (uint32_t)normals2_pitch, rows2, cols2);
  return check(hipGetLastError(), "resize_points_normals2_kernel launch");
}

int launch_render_image(const yakhip_ctx* ctx, bool from_depth, const uint8_t* src, size_t src_pitch, const uint8_t* normals,
                        size_t normals_pitch, int rows, int cols, const float reproj_args[4], const float light[3],
                        uint8_t* image, size_t image_pitch)
{
  // internal.hpp:80 Reprojector(fx, fy, cx, cy): finv = 1 / f
  const float finvx = from_depth ? 1.f / reproj_args[0] : 0.f, finvy = from_depth ? 1.f / reproj_args[1] : 0.f;
  const float cx = from_depth ? reproj_args[2] : 0.f, cy = from_depth ? reproj_args[3] : 0.f;
  if (from_depth)
    hipLaunchKernelGGL(render_image_kernel<true>, image_grid(rows, cols), dim3(256), 0, ctx->stream, src,
                       (uint32_t)src_pitch, normals, (uint32_t)normals_pitch, rows, cols, finvx, finvy, cx, cy, light[0],
                       light[1], light[2], image, (uint32_t)image_pitch);
  else
    hipLaunchKernelGGL(render_image_kernel<false>, image_grid(rows, cols), dim3(256), 0, ctx->stream, src,
                       (uint32_t)src_pitch, normals, (uint32_t)normals_pitch, rows, cols, finvx, finvy, cx, cy, light[0],
                       light[1], light[2], image, (uint32_t)image_pitch);
  return check(hipGetLastError(), "render_image_kernel launch");
}

int launch_tangent_colors(const yakhip_ctx* ctx, const uint8_t* normals, size_t normals_pitch, int rows, int cols,
                          uint8_t* image, size_t image_pitch)
{
  hipLaunchKernelGGL(tangent_colors_kernel, image_grid(rows, cols), dim3(256), 0, ctx->stream, normals,
                     (uint32_t)normals_pitch, rows, cols, image, (uint32_t)image_pitch);
  return check(hipGetLastError(), "tangent_colors_kernel launch");
}
}  // namespace yakhip
