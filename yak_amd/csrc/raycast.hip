// raycast.hip -- volume ray casting for gfx950: replaces kfusion::device::raycast
// (yak/src/cuda/tsdf_volume.cu:149-425).  Citations are relative to /root/reference/.
//
// Arithmetic follows the reference statement by statement (fp32, no contraction: this TU is built
// with -ffp-contract=off and correctly rounded '/' and sqrtf), so the result is bit-identical to the
// CPU oracle; what differs is the schedule:
//   * a wave owns an 8 x 8 pixel tile (not a 32 x 1 row segment of a 32 x 8 block): neighbouring rays
//     sample neighbouring voxels, so one wave-wide gather touches ~8 short row segments, not 64 lines;
//   * the march positions do not depend on the fetched values, so U samples are put in flight at once
//     and then examined in order -- the loop-carried dependency is the exit test only;
//   * lanes that found a zero crossing leave the march and do the 64 trilinear taps together after it;
//   * workgroup ids are remapped so each XCD (own L2) walks a contiguous band of the image.
#include "yakhip_internal.h"

namespace yakhip
{
namespace
{
constexpr int kUnroll = 4;

template <bool IDX32>
__device__ __forceinline__ float voxel_tsdf(const RaycastParams& p, int x, int y, int z)
{
  // only the half (low 16 bits of the little-endian voxel word) is read: device.hpp:78-81
  if constexpr (IDX32)
  {
    const uint32_t idx = (uint32_t)x + (uint32_t)p.X * ((uint32_t)y + (uint32_t)p.Y * (uint32_t)z);
    return half_bits_to_float(*(const unsigned short*)((const uint8_t*)p.vol + (size_t)(idx << 2)));
  }
  else
  {
    const size_t idx = (size_t)x + (size_t)p.X * ((size_t)y + (size_t)p.Y * (size_t)z);
    return half_bits_to_float(*(const unsigned short*)(p.vol + idx));
  }
}

__device__ __forceinline__ int clampi(int v, int hi) { return min(max(v, 0), hi); }

// tsdf_volume.cu:209-217: nearest voxel, round-half-even.  The reference does not bounds-test here;
// the clamp keeps the speculative look-ahead fetches inside the allocation and changes nothing where
// the reference's read is defined.
template <bool IDX32>
__device__ __forceinline__ float fetch_tsdf(const RaycastParams& p, float px, float py, float pz)
{
  const int x = clampi((int)__builtin_rintf(px * p.ivx), p.X - 1);
  const int y = clampi((int)__builtin_rintf(py * p.ivy), p.Y - 1);
  const int z = clampi((int)__builtin_rintf(pz * p.ivz), p.Z - 1);
  return voxel_tsdf<IDX32>(p, x, y, z);
}

// tsdf_volume.cu:167-192: trilinear tap, argument in voxel units
template <bool IDX32>
__device__ float interpolate(const RaycastParams& p, float cfx, float cfy, float cfz)
{
  const float fx = __builtin_floorf(cfx), fy = __builtin_floorf(cfy), fz = __builtin_floorf(cfz);
  if (!(fx >= 0.f && fx < (float)(p.X - 1) && fy >= 0.f && fy < (float)(p.Y - 1) && fz >= 0.f && fz < (float)(p.Z - 1)))
    return __builtin_nanf("");
  const int gx = (int)fx, gy = (int)fy, gz = (int)fz;
  const float a = cfx - (float)gx, b = cfy - (float)gy, c = cfz - (float)gz;
  const float v000 = voxel_tsdf<IDX32>(p, gx + 0, gy + 0, gz + 0);
  const float v001 = voxel_tsdf<IDX32>(p, gx + 0, gy + 0, gz + 1);
  const float v010 = voxel_tsdf<IDX32>(p, gx + 0, gy + 1, gz + 0);
  const float v011 = voxel_tsdf<IDX32>(p, gx + 0, gy + 1, gz + 1);
  const float v100 = voxel_tsdf<IDX32>(p, gx + 1, gy + 0, gz + 0);
  const float v101 = voxel_tsdf<IDX32>(p, gx + 1, gy + 0, gz + 1);
  const float v110 = voxel_tsdf<IDX32>(p, gx + 1, gy + 1, gz + 0);
  const float v111 = voxel_tsdf<IDX32>(p, gx + 1, gy + 1, gz + 1);
  float tsdf = 0.f;
  tsdf += v000 * (1.f - a) * (1.f - b) * (1.f - c);
  tsdf += v001 * (1.f - a) * (1.f - b) * c;
  tsdf += v010 * (1.f - a) * b * (1.f - c);
  tsdf += v011 * (1.f - a) * b * c;
  tsdf += v100 * a * (1.f - b) * (1.f - c);
  tsdf += v101 * a * (1.f - b) * c;
  tsdf += v110 * a * b * (1.f - c);
  tsdf += v111 * a * b * c;
  return tsdf;
}

// temp_utils.hpp:103 -- v * rsqrt(dot(v, v)), with rsqrt restated as 1 / sqrt
__device__ __forceinline__ void normalize3(float& x, float& y, float& z)
{
  const float s = 1.0f / __builtin_sqrtf(dot3(x, y, z, x, y, z));
  x *= s;
  y *= s;
  z *= s;
}

template <bool IDX32>
__global__ __launch_bounds__(256) void raycast_kernel(const RaycastParams p, int tiles_x, int n_tiles, int per_xcd)
{
  // XCD-aware tile order: consecutive workgroup ids go round-robin over the 8 XCDs; give each XCD a
  // contiguous run of tiles (a band of the image) so that its L2 sees one part of the volume.
  const int bid = (int)blockIdx.x;
  const int tile = (bid & 7) * per_xcd + (bid >> 3);
  if (tile >= n_tiles)
    return;
  const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  const int x = tile_x * 16 + (wave & 1) * 8 + (lane & 7);
  const int y = tile_y * 16 + (wave >> 1) * 8 + (lane >> 3);
  if (x >= p.cols || y >= p.rows)
    return;

  // tsdf_volume.cu:300-301; device.hpp:57-62 with z = 1
  const float rx = ((float)x - p.cx) * p.finvx;
  const float ry = ((float)y - p.cy) * p.finvy;
  float dx = dot3(p.R[0], p.R[1], p.R[2], rx, ry, 1.f);
  float dy = dot3(p.R[3], p.R[4], p.R[5], rx, ry, 1.f);
  float dz = dot3(p.R[6], p.R[7], p.R[8], rx, ry, 1.f);
  normalize3(dx, dy, dz);
  const float ox = p.t[0], oy = p.t[1], oz = p.t[2];

  // tsdf_volume.cu:149-165, 306-314
  const float irx = 1.f / dx, iry = 1.f / dy, irz = 1.f / dz;
  const float tbx = irx * (0.f - ox), tby = iry * (0.f - oy), tbz = irz * (0.f - oz);
  const float ttx = irx * (p.bx - ox), tty = iry * (p.by - oy), ttz = irz * (p.bz - oz);
  const float tmnx = fminf(ttx, tbx), tmny = fminf(tty, tby), tmnz = fminf(ttz, tbz);
  const float tmxx = fmaxf(ttx, tbx), tmxy = fmaxf(tty, tby), tmxz = fmaxf(ttz, tbz);
  float tmin = fmaxf(fmaxf(tmnx, tmny), fmaxf(tmnx, tmnz));
  float tmax = fminf(fminf(tmxx, tmxy), fminf(tmxx, tmxz));
  tmin = fmaxf(0.f, tmin);

  // 0 = marching, 1 = zero crossing found, 2 = finished without one
  int state = tmin < tmax ? 0 : 2;

  // tsdf_volume.cu:316-318
  const float ts = p.time_step;
  tmax -= ts;
  const float sx = dx * ts, sy = dy * ts, sz = dz * ts;
  float nx = ox + dx * tmin, ny = oy + dy * tmin, nz = oz + dz * tmin;
  float tcurr = tmin;
  float hcx = 0.f, hcy = 0.f, hcz = 0.f, hnx = 0.f, hny = 0.f, hnz = 0.f;  // the bracketing samples of a hit

  // tsdf_volume.cu:320-351, kUnroll samples in flight per trip
  if (state == 0)
  {
    float tsdf_next = fetch_tsdf<IDX32>(p, nx, ny, nz);
    while (state == 0)
    {
      float qx[kUnroll], qy[kUnroll], qz[kUnroll], f[kUnroll];
      float ax = nx, ay = ny, az = nz;
#pragma unroll
      for (int k = 0; k < kUnroll; ++k)
      {
        ax += sx;
        ay += sy;
        az += sz;
        qx[k] = ax, qy[k] = ay, qz[k] = az;
        f[k] = fetch_tsdf<IDX32>(p, ax, ay, az);
      }
      float cx = nx, cy = ny, cz = nz, fc = tsdf_next;
#pragma unroll
      for (int k = 0; k < kUnroll; ++k)
      {
        if (state == 0)
        {
          if (!(tcurr < tmax))
            state = 2;
          else if (fc < 0.f && f[k] > 0.f)
            state = 2;
          else if (fc > 0.f && f[k] < 0.f)
          {
            state = 1;
            hcx = cx, hcy = cy, hcz = cz;
            hnx = qx[k], hny = qy[k], hnz = qz[k];
          }
          else
            tcurr += ts;
        }
        cx = qx[k], cy = qy[k], cz = qz[k], fc = f[k];
      }
      nx = cx, ny = cy, nz = cz, tsdf_next = fc;
    }
  }

  const float qnan = __builtin_nanf("");
  float vx = qnan, vy = qnan, vz = qnan, vw = qnan, n0 = qnan, n1 = qnan, n2 = qnan, n3 = qnan;
  bool hit = false;
  if (state == 1)
  {
    // tsdf_volume.cu:333-348
    const float Ft = interpolate<IDX32>(p, hcx * p.ivx, hcy * p.ivy, hcz * p.ivz);
    const float Ftdt = interpolate<IDX32>(p, hnx * p.ivx, hny * p.ivy, hnz * p.ivz);
    const float Ts = tcurr - (ts * Ft) / (Ftdt - Ft);
    const float px = ox + dx * Ts, py = oy + dy * Ts, pz = oz + dz * Ts;

    // compute_normal, tsdf_volume.cu:354-372
    const float Fx1 = interpolate<IDX32>(p, (px + p.gdx) * p.ivx, py * p.ivy, pz * p.ivz);
    const float Fx2 = interpolate<IDX32>(p, (px - p.gdx) * p.ivx, py * p.ivy, pz * p.ivz);
    const float Fy1 = interpolate<IDX32>(p, px * p.ivx, (py + p.gdy) * p.ivy, pz * p.ivz);
    const float Fy2 = interpolate<IDX32>(p, px * p.ivx, (py - p.gdy) * p.ivy, pz * p.ivz);
    const float Fz1 = interpolate<IDX32>(p, px * p.ivx, py * p.ivy, (pz + p.gdz) * p.ivz);
    const float Fz2 = interpolate<IDX32>(p, px * p.ivx, py * p.ivy, (pz - p.gdz) * p.ivz);
    float gx = (Fx1 - Fx2) / p.gdx;
    float gy = (Fy1 - Fy2) / p.gdy;
    float gz = (Fz1 - Fz2) / p.gdz;
    normalize3(gx, gy, gz);
    const float prod = gx * gy * gz;
    if (!(prod != prod))
    {
      hit = true;
      n0 = dot3(p.Rinv[0], p.Rinv[1], p.Rinv[2], gx, gy, gz);
      n1 = dot3(p.Rinv[3], p.Rinv[4], p.Rinv[5], gx, gy, gz);
      n2 = dot3(p.Rinv[6], p.Rinv[7], p.Rinv[8], gx, gy, gz);
      n3 = 0.f;
      const float wx = px - p.t[0], wy = py - p.t[1], wz = pz - p.t[2];
      vx = dot3(p.Rinv[0], p.Rinv[1], p.Rinv[2], wx, wy, wz);
      vy = dot3(p.Rinv[3], p.Rinv[4], p.Rinv[5], wx, wy, wz);
      vz = dot3(p.Rinv[6], p.Rinv[7], p.Rinv[8], wx, wy, wz);
      vw = 0.f;
    }
  }

  // tsdf_volume.cu:298 (defaults) and :346-347 / :279-280 (hit)
  float4* nrow = (float4*)(p.normals + (size_t)y * p.normals_pitch);
  nrow[x] = make_float4(n0, n1, n2, n3);
  if (p.points)
  {
    float4* prow = (float4*)(p.points + (size_t)y * p.points_pitch);
    prow[x] = make_float4(vx, vy, vz, vw);
  }
  if (p.depth)
  {
    unsigned short mm_out = 0;
    if (hit)
    {  // static_cast<ushort>(vertex.z * 1000): truncating, saturating conversion
      const float mm = vz * 1000.f;
      mm_out = (unsigned short)(mm >= 65535.f ? 65535 : (mm > 0.f ? (int)mm : 0));
    }
    unsigned short* drow = (unsigned short*)(p.depth + (size_t)y * p.depth_pitch);
    drow[x] = mm_out;
  }
}
}  // namespace

int launch_raycast(yakhip_ctx* ctx, const RaycastParams& p)
{
  const int tiles_x = (p.cols + 15) / 16, tiles_y = (p.rows + 15) / 16;
  const long long n_tiles_ll = (long long)tiles_x * tiles_y;
  if (n_tiles_ll > 0x7ffffff0ll / 8)
    return fail(YAKHIP_ERANGE, "raycast: image too large");
  const int n_tiles = (int)n_tiles_ll;
  const int per_xcd = (n_tiles + 7) / 8;
  const dim3 grid((unsigned)(per_xcd * 8));
  const bool idx32 = (unsigned long long)p.X * (unsigned long long)p.Y * (unsigned long long)p.Z <= (1ull << 30);

  std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
  int rc = timing_begin(ctx, &ev);
  if (rc != 0)
    return rc;
  if (idx32)
    hipLaunchKernelGGL((raycast_kernel<true>), grid, dim3(256), 0, ctx->stream, p, tiles_x, n_tiles, per_xcd);
  else
    hipLaunchKernelGGL((raycast_kernel<false>), grid, dim3(256), 0, ctx->stream, p, tiles_x, n_tiles, per_xcd);
  if ((rc = timing_end(ctx, ev)) != 0)
    return rc;
  return check(hipGetLastError(), "raycast_kernel launch");
}
}  // namespace yakhip
