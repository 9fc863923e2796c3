/*
 * yak_oracle.c -- scalar CPU restatement of the Yak TSDF hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see yak_oracle.h).  PARITY UNPINNED for integrate /
 * compute_dists: the reference has no test or golden vector for them.
 *
 * Build: gcc -std=c11 -O2 -ffp-contract=off -fno-fast-math -fopenmp (oracle/Makefile).
 * -ffp-contract=off matters: every fused multiply-add below is written as fmaf()
 * exactly where the CUDA source has __fmaf_rn, and nowhere else.
 *
 * CUDA intrinsic -> C mapping used throughout:
 *   __fmaf_rn -> fmaf          __fsqrt_rn / sqrtf -> sqrtf (correctly rounded)
 *   __fdividef -> IEEE '/'     __float2half[_rn] -> yko_float2half_rn (RNE)
 *   tex2D point fetch, unnormalised coords -> texel (floor(x), floor(y))
 *   __float2int_rd of a NaN pixel coordinate (gives 0 on the device; only a non-finite pose or intrinsics
 *   produce one) -> the voxel is skipped, see integrate_column
 *
 * Citations are relative to /root/reference/yak/.
 */
#include "yak_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * IEEE binary16 <-> binary32.  cuda_fp16's __float2half is round-to-nearest-even;
 * __half2float is exact.  (The reference's vendored half.hpp truncates on
 * float->half, half.hpp:214-215, so it is NOT used to produce voxels.)
 * ------------------------------------------------------------------------------------------ */
uint16_t yko_float2half_rn(float f)
{
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint32_t sign = (u >> 16) & 0x8000u;
  const uint32_t abs = u & 0x7fffffffu;

  if (abs >= 0x7f800000u) /* inf / nan */
    return (uint16_t)(sign | 0x7c00u | (abs > 0x7f800000u ? (0x0200u | ((abs >> 13) & 0x3ffu)) : 0u));
  if (abs >= 0x477ff000u) /* >= 65520 rounds to inf */
    return (uint16_t)(sign | 0x7c00u);
  if (abs < 0x33000001u) /* <= 2^-25 rounds to zero (tie at exactly 2^-25 goes to even = 0) */
    return (uint16_t)sign;

  const int exp = (int)(abs >> 23); /* biased fp32 exponent */
  uint32_t mant = abs & 0x7fffffu;
  uint32_t half;
  if (exp >= 113)
  { /* normal half: rebias 127 -> 15, drop 13 mantissa bits with RNE */
    half = ((uint32_t)(exp - 112) << 10) | (mant >> 13);
    const uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1u)))
      ++half; /* carry may roll into the exponent: still the right encoding */
  }
  else
  { /* subnormal half: value = mant24 * 2^(exp-150), result unit = 2^-24 */
    mant |= 0x800000u;
    const int shift = 126 - exp; /* 14..24 */
    half = mant >> shift;
    const uint32_t rem = mant & ((1u << shift) - 1u);
    const uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half & 1u)))
      ++half;
  }
  return (uint16_t)(sign | half);
}

float yko_half2float(uint16_t h)
{
  const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  const uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t mant = h & 0x3ffu;
  uint32_t u;
  if (exp == 0)
  {
    if (mant == 0)
      u = sign;
    else
    { /* subnormal: normalise */
      int e = -1;
      do
      {
        ++e;
        mant <<= 1;
      } while (!(mant & 0x400u));
      u = sign | ((uint32_t)(112 - e) << 23) | ((mant & 0x3ffu) << 13);
    }
  }
  else if (exp == 31)
    u = sign | 0x7f800000u | (mant << 13);
  else
    u = sign | ((exp + 112u) << 23) | (mant << 13);
  float f;
  memcpy(&f, &u, 4);
  return f;
}

void yko_float2half_array(const float* in, uint16_t* out, size_t n)
{
  for (size_t i = 0; i < n; ++i)
    out[i] = yko_float2half_rn(in[i]);
}

void yko_half2float_array(const uint16_t* in, float* out, size_t n)
{
  for (size_t i = 0; i < n; ++i)
    out[i] = yko_half2float(in[i]);
}

/* device.hpp:69-81 (the __CUDA_ARCH__ >= 500 branch: short2, so the weight is a SIGNED 16-bit
 * lane; with weights in [0, 32767] this is identical to the ushort2 branch :83-96).
 * Little-endian: bytes 0-1 = half tsdf, bytes 2-3 = weight, the same word layout as
 * TSDFContainer::unpack (tsdf_container.h:45-49) and the test's pack() (test/marching_cubes_tests.cpp:46-50). */
uint32_t yko_pack_tsdf(float tsdf, int weight)
{
  return (uint32_t)yko_float2half_rn(tsdf) | ((uint32_t)(uint16_t)(int16_t)weight << 16);
}

float yko_unpack_tsdf(uint32_t voxel, int* weight)
{
  if (weight)
    *weight = (int)(int16_t)(uint16_t)(voxel >> 16);
  return yko_half2float((uint16_t)(voxel & 0xffffu));
}

/* ------------------------------------------------------------------------------------------
 * imgproc.cu:259-283.  finv = 1/f is formed on the host (:281); per pixel
 *   xl = (x - cx) * finvx; yl = (y - cy) * finvy; lambda = sqrtf(xl*xl + yl*yl + 1)
 *   dists(y,x) = half_rn( depth(y,x) * lambda * 0.001f )
 * evaluated left to right in fp32.  No FMA contraction here (C semantics of the source;
 * nvcc's default -fmad could fuse xl*xl + yl*yl -- unpinned, documented in DESIGN.md).
 * The kernel's bounds test uses '||' (:264); for in-range pixels the result is the same.
 * ------------------------------------------------------------------------------------------ */
void yko_compute_dists(const uint16_t* depth, size_t depth_pitch_bytes, uint16_t* dists, size_t dists_pitch_bytes,
                       int rows, int cols, float fx, float fy, float cx, float cy)
{
  const float finvx = 1.f / fx, finvy = 1.f / fy;
  for (int y = 0; y < rows; ++y)
  {
    const uint16_t* drow = (const uint16_t*)((const char*)depth + (size_t)y * depth_pitch_bytes);
    uint16_t* orow = (uint16_t*)((char*)dists + (size_t)y * dists_pitch_bytes);
    const float yl = ((float)y - cy) * finvy;
    for (int x = 0; x < cols; ++x)
    {
      const float xl = ((float)x - cx) * finvx;
      const float lambda = sqrtf(xl * xl + yl * yl + 1.f);
      orow[x] = yko_float2half_rn((float)drow[x] * lambda * 0.001f);
    }
  }
}

/* tsdf_volume.cu:14-27: every voxel <- pack_tsdf(0.f, 0) == 0x00000000 */
void yko_clear_volume(uint32_t* volume, size_t n_voxels)
{
  const uint32_t zero = yko_pack_tsdf(0.f, 0);
  for (size_t i = 0; i < n_voxels; ++i)
    volume[i] = zero;
}

/* temp_utils.hpp:36-39 */
static inline float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
  return fmaf(ax, bx, fmaf(ay, by, az * bz));
}

/* One (x, y) column: tsdf_volume.cu:62-108. */
static void integrate_column(uint32_t* slab, int x, int y, const int dims[3], int z0, int z1, const float vs[3],
                             float trunc_dist, float trunc_dist_inv, int max_weight, const uint16_t* dists,
                             size_t pitch, int rows, int cols, const float R[9], const float t[3],
                             const float intr[4], uint64_t* visited, uint64_t* updated)
{
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];

  /* :69  zstep = (R.data[0].z, R.data[1].z, R.data[2].z) * voxel_size.z */
  const float zsx = R[2] * vs[2], zsy = R[5] * vs[2], zsz = R[8] * vs[2];

  /* :71-72  vx = (x*vsx, y*vsy, 0); vc = vol2cam * vx = R*vx + t   (device.hpp:106-111) */
  const float vxx = (float)x * vs[0], vxy = (float)y * vs[1], vxz = 0.f;
  float vcx = dot3(R[0], R[1], R[2], vxx, vxy, vxz) + t[0];
  float vcy = dot3(R[3], R[4], R[5], vxx, vxy, vxz) + t[1];
  float vcz = dot3(R[6], R[7], R[8], vxx, vxy, vxz) + t[2];

  const size_t plane = (size_t)dims[0] * (size_t)dims[1];
  uint32_t* vptr = slab + (size_t)x + (size_t)y * (size_t)dims[0]; /* beg(x,y), device.hpp:33-36 */

  /* :76  for (i = 0; i < dims.z; ++i, vc += zstep, vptr = zstep(vptr)) -- the additions run on
   * every step, including skipped ones, and from z = 0 even when only [z0,z1) is owned. */
  for (int i = 0; i < z1; ++i, vcx += zsx, vcy += zsy, vcz += zsz)
  {
    if (i < z0)
      continue;
    uint32_t* p = vptr + (size_t)(i - z0) * plane;
    ++*visited;

    /* :89 'vc.z <= 0' is tested after the fetch in the source; tested first here because with
     * vc.z <= 0 the projection may be NaN/inf.  Both orders 'continue' for exactly the same voxels. */
    if (vcz <= 0.f)
      continue;

    /* :78  coo = proj(vc)   device.hpp:46-52 */
    const float coox = fmaf(fx, vcx / vcz, cx);
    const float cooy = fmaf(fy, vcy / vcz, cy);

    /* :83 -- the reference tests the converted integers; floor(coo) lies inside the image <=> 0 <= coo < size.
     * Written so that a NaN coordinate (non-finite pose or intrinsics) is rejected: __float2int_rd turns a NaN into
     * 0 and the reference then reads row / column 0 of the image -- not reproduced, here or in the HIP kernel. */
    if (!(coox >= 0.f && cooy >= 0.f && coox < (float)cols && cooy < (float)rows))
      continue;

    /* :87  point-filtered, unnormalised texture fetch of a half texel */
    const int px = (int)floorf(coox), py = (int)floorf(cooy);
    const float Dp = yko_half2float(((const uint16_t*)((const char*)dists + (size_t)py * pitch))[px]);

    /* :89 */
    if (Dp == 0.f)
      continue;

    /* :92 */
    const float sdf = Dp - sqrtf(dot3(vcx, vcy, vcz, vcx, vcy, vcz));

    /* :94-107 */
    if (sdf >= -trunc_dist)
    {
      const float tsdf = fminf(1.f, sdf * trunc_dist_inv);
      int weight_prev;
      const float tsdf_prev = yko_unpack_tsdf(*p, &weight_prev);
      const float tsdf_new = fmaf(tsdf_prev, (float)weight_prev, tsdf) / (float)(weight_prev + 1);
      const int weight_new = weight_prev + 1 < max_weight ? weight_prev + 1 : max_weight;
      *p = yko_pack_tsdf(tsdf_new, weight_new);
      ++*updated;
    }
  }
}

void yko_integrate(uint32_t* slab, const int dims[3], int z0, int z1, const float voxel_size[3], float trunc_dist,
                   int max_weight, const uint16_t* dists, size_t dists_pitch_bytes, int rows, int cols,
                   const float R[9], const float t[3], const float intr[4], uint64_t stats[2], int nthreads)
{
  /* tsdf_volume.cu:125 */
  const float trunc_dist_inv = 1.f / trunc_dist;
  uint64_t visited = 0, updated = 0;
  if (z0 < 0)
    z0 = 0;
  if (z1 > dims[2])
    z1 = dims[2];

#ifdef _OPENMP
  if (nthreads < 1)
    nthreads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads) reduction(+ : visited, updated)
#endif
  for (int y = 0; y < dims[1]; ++y)
    for (int x = 0; x < dims[0]; ++x)
      integrate_column(slab, x, y, dims, z0, z1, voxel_size, trunc_dist, trunc_dist_inv, max_weight, dists,
                       dists_pitch_bytes, rows, cols, R, t, intr, &visited, &updated);

  if (stats)
  {
    stats[0] += visited;
    stats[1] += updated;
  }
}

/* ------------------------------------------------------------------------------------------
 * Ray casting: tsdf_volume.cu:149-425.
 * ------------------------------------------------------------------------------------------ */
typedef struct
{
  const uint32_t* vol;
  int X, Y, Z;
  float vsx, vsy, vsz;       /* voxel_size */
  float ivx, ivy, ivz;       /* voxel_size_inv = 1.f / voxel_size  (cu:402) */
  float gdx, gdy, gdz;       /* gradient_delta = voxel_size * factor  (cu:401) */
  float szx, szy, szz;       /* volume_size = voxel_size * dims  (cu:399) */
  float time_step;           /* trunc_dist * raycast_step_factor  (cu:400) */
  const float *R, *t, *Rinv; /* aff = cam2vol, Rinv */
  float finvx, finvy, cx, cy; /* Reprojector, precomp.cpp:73-76 */
} Raycaster;

static inline float voxel_tsdf(const Raycaster* rc, int x, int y, int z)
{
  return yko_half2float((uint16_t)(rc->vol[(size_t)x + (size_t)rc->X * ((size_t)y + (size_t)rc->Y * (size_t)z)] & 0xffffu));
}

static inline int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

/* cu:209-217  nearest voxel, round-half-even */
static inline float fetch_tsdf(const Raycaster* rc, float px, float py, float pz, uint64_t* fetches)
{
  const int x = clampi((int)nearbyintf(px * rc->ivx), rc->X - 1);
  const int y = clampi((int)nearbyintf(py * rc->ivy), rc->Y - 1);
  const int z = clampi((int)nearbyintf(pz * rc->ivz), rc->Z - 1);
  ++*fetches;
  return voxel_tsdf(rc, x, y, z);
}

/* cu:167-192  trilinear; the argument is already in voxel units */
static float interpolate(const Raycaster* rc, float cfx, float cfy, float cfz)
{
  const float fx = floorf(cfx), fy = floorf(cfy), fz = floorf(cfz);
  /* compare in float first so that NaN / huge values cannot reach the int conversion */
  if (!(fx >= 0.f && fx < (float)(rc->X - 1) && fy >= 0.f && fy < (float)(rc->Y - 1) && fz >= 0.f && fz < (float)(rc->Z - 1)))
    return NAN;
  const int gx = (int)fx, gy = (int)fy, gz = (int)fz;
  const float a = cfx - (float)gx, b = cfy - (float)gy, c = cfz - (float)gz;
  float tsdf = 0.f;
  tsdf += voxel_tsdf(rc, gx + 0, gy + 0, gz + 0) * (1 - a) * (1 - b) * (1 - c);
  tsdf += voxel_tsdf(rc, gx + 0, gy + 0, gz + 1) * (1 - a) * (1 - b) * c;
  tsdf += voxel_tsdf(rc, gx + 0, gy + 1, gz + 0) * (1 - a) * b * (1 - c);
  tsdf += voxel_tsdf(rc, gx + 0, gy + 1, gz + 1) * (1 - a) * b * c;
  tsdf += voxel_tsdf(rc, gx + 1, gy + 0, gz + 0) * a * (1 - b) * (1 - c);
  tsdf += voxel_tsdf(rc, gx + 1, gy + 0, gz + 1) * a * (1 - b) * c;
  tsdf += voxel_tsdf(rc, gx + 1, gy + 1, gz + 0) * a * b * (1 - c);
  tsdf += voxel_tsdf(rc, gx + 1, gy + 1, gz + 1) * a * b * c;
  return tsdf;
}

/* temp_utils.hpp:103  v * rsqrt(dot(v, v)) */
static inline void normalize3(float* x, float* y, float* z)
{
  const float s = 1.0f / sqrtf(dot3(*x, *y, *z, *x, *y, *z));
  *x *= s;
  *y *= s;
  *z *= s;
}

/* cu:354-372 */
static void compute_normal(const Raycaster* rc, float px, float py, float pz, float n[3])
{
  const float Fx1 = interpolate(rc, (px + rc->gdx) * rc->ivx, py * rc->ivy, pz * rc->ivz);
  const float Fx2 = interpolate(rc, (px - rc->gdx) * rc->ivx, py * rc->ivy, pz * rc->ivz);
  n[0] = (Fx1 - Fx2) / rc->gdx;
  const float Fy1 = interpolate(rc, px * rc->ivx, (py + rc->gdy) * rc->ivy, pz * rc->ivz);
  const float Fy2 = interpolate(rc, px * rc->ivx, (py - rc->gdy) * rc->ivy, pz * rc->ivz);
  n[1] = (Fy1 - Fy2) / rc->gdy;
  const float Fz1 = interpolate(rc, px * rc->ivx, py * rc->ivy, (pz + rc->gdz) * rc->ivz);
  const float Fz2 = interpolate(rc, px * rc->ivx, py * rc->ivy, (pz - rc->gdz) * rc->ivz);
  n[2] = (Fz1 - Fz2) / rc->gdz;
  normalize3(&n[0], &n[1], &n[2]);
}

/* One pixel: cu:288-352 (and :220-285, which differs only in what is stored).
 * Returns 1 and fills vertex (camera frame) / normal on a hit. */
static int raycast_pixel(const Raycaster* rc, int x, int y, float vertex[3], float normal[3], uint64_t* fetches,
                         int z_own0, int z_own1, int* key)
{
  /* Slab form (no counterpart in the reference; SURVEY.md 8e option 1): the same march, but a sample pair
   * is examined only if its first sample's nearest layer lies in [z_own0, z_own1).  *key = 2 * index of
   * the pair that ended the march, or INT32_MAX.  With [0, Z) this is the reference's march. */
  *key = 0x7fffffff;
  const float* R = rc->R;
  const float* t = rc->t;
  /* cu:300-301  ray_dir = normalized(aff.R * reproj(x, y, 1.f));  device.hpp:57-62 */
  const float z1 = 1.f;
  const float rx = z1 * ((float)x - rc->cx) * rc->finvx;
  const float ry = z1 * ((float)y - rc->cy) * rc->finvy;
  float dx = dot3(R[0], R[1], R[2], rx, ry, z1);
  float dy = dot3(R[3], R[4], R[5], rx, ry, z1);
  float dz = dot3(R[6], R[7], R[8], rx, ry, z1);
  normalize3(&dx, &dy, &dz);
  const float ox = t[0], oy = t[1], oz = t[2];

  /* cu:306  box_max = volume_size - voxel_size ;  cu:149-165 intersect (box_min = 0) */
  const float bx = rc->szx - rc->vsx, by = rc->szy - rc->vsy, bz = rc->szz - rc->vsz;
  const float irx = 1.f / dx, iry = 1.f / dy, irz = 1.f / dz;
  const float tbx = irx * (0.f - ox), tby = iry * (0.f - oy), tbz = irz * (0.f - oz);
  const float ttx = irx * (bx - ox), tty = iry * (by - oy), ttz = irz * (bz - oz);
  const float tmnx = fminf(ttx, tbx), tmny = fminf(tty, tby), tmnz = fminf(ttz, tbz);
  const float tmxx = fmaxf(ttx, tbx), tmxy = fmaxf(tty, tby), tmxz = fmaxf(ttz, tbz);
  float tmin = fmaxf(fmaxf(tmnx, tmny), fmaxf(tmnx, tmnz));
  float tmax = fminf(fminf(tmxx, tmxy), fminf(tmxx, tmxz));

  /* cu:311-314 */
  tmin = fmaxf(0.f, tmin);
  if (!(tmin < tmax)) /* 'if (tmin >= tmax) return', and NaN bounds never march */
    return 0;

  /* cu:316-318 */
  const float ts = rc->time_step;
  tmax -= ts;
  const float sx = dx * ts, sy = dy * ts, sz = dz * ts;
  float nx = ox + dx * tmin, ny = oy + dy * tmin, nz = oz + dz * tmin;

  /* cu:320-351 */
  float tsdf_next = fetch_tsdf(rc, nx, ny, nz, fetches);
  int pair = 0;
  for (float tcurr = tmin; tcurr < tmax; tcurr += ts, ++pair)
  {
    const float tsdf_curr = tsdf_next;
    const float cxp = nx, cyp = ny, czp = nz;
    const int zc = clampi((int)nearbyintf(czp * rc->ivz), rc->Z - 1);
    const int own = zc >= z_own0 && zc < z_own1;
    nx += sx;
    ny += sy;
    nz += sz;
    tsdf_next = fetch_tsdf(rc, nx, ny, nz, fetches);
    if (!own)
      continue;
    if (tsdf_curr < 0.f && tsdf_next > 0.f)
    {
      *key = 2 * pair;
      return 0;
    }
    if (tsdf_curr > 0.f && tsdf_next < 0.f)
    {
      *key = 2 * pair;
      const float Ft = interpolate(rc, cxp * rc->ivx, cyp * rc->ivy, czp * rc->ivz);
      const float Ftdt = interpolate(rc, nx * rc->ivx, ny * rc->ivy, nz * rc->ivz);
      const float Ts = tcurr - (ts * Ft) / (Ftdt - Ft);
      const float vx = ox + dx * Ts, vy = oy + dy * Ts, vz = oz + dz * Ts;
      float n[3];
      compute_normal(rc, vx, vy, vz, n);
      if (isnan(n[0] * n[1] * n[2]))
        return 0;
      const float* Ri = rc->Rinv;
      normal[0] = dot3(Ri[0], Ri[1], Ri[2], n[0], n[1], n[2]);
      normal[1] = dot3(Ri[3], Ri[4], Ri[5], n[0], n[1], n[2]);
      normal[2] = dot3(Ri[6], Ri[7], Ri[8], n[0], n[1], n[2]);
      const float wx = vx - t[0], wy = vy - t[1], wz = vz - t[2];
      vertex[0] = dot3(Ri[0], Ri[1], Ri[2], wx, wy, wz);
      vertex[1] = dot3(Ri[3], Ri[4], Ri[5], wx, wy, wz);
      vertex[2] = dot3(Ri[6], Ri[7], Ri[8], wx, wy, wz);
      return 1;
    }
  }
  return 0;
}

void yko_raycast(const uint32_t* volume, const int dims[3], const float voxel_size[3], float trunc_dist,
                 const float R[9], const float t[3], const float Rinv[9], const float intr[4],
                 float raycast_step_factor, float gradient_delta_factor, int rows, int cols, float* points,
                 size_t points_pitch_bytes, float* normals, size_t normals_pitch_bytes, uint16_t* depth,
                 size_t depth_pitch_bytes, uint64_t stats[2], int nthreads)
{
  yko_raycast_slab(volume, dims, voxel_size, trunc_dist, R, t, Rinv, intr, raycast_step_factor, gradient_delta_factor,
                   rows, cols, points, points_pitch_bytes, normals, normals_pitch_bytes, depth, depth_pitch_bytes, stats,
                   nthreads, 0, dims[2], NULL, 0);
}

void yko_raycast_slab(const uint32_t* volume, const int dims[3], const float voxel_size[3], float trunc_dist,
                      const float R[9], const float t[3], const float Rinv[9], const float intr[4],
                      float raycast_step_factor, float gradient_delta_factor, int rows, int cols, float* points,
                      size_t points_pitch_bytes, float* normals, size_t normals_pitch_bytes, uint16_t* depth,
                      size_t depth_pitch_bytes, uint64_t stats[2], int nthreads, int z_own0, int z_own1, int32_t* keys,
                      size_t keys_pitch_bytes)
{
  Raycaster rc;
  rc.vol = volume;
  rc.X = dims[0], rc.Y = dims[1], rc.Z = dims[2];
  rc.vsx = voxel_size[0], rc.vsy = voxel_size[1], rc.vsz = voxel_size[2];
  rc.ivx = 1.f / rc.vsx, rc.ivy = 1.f / rc.vsy, rc.ivz = 1.f / rc.vsz;
  rc.gdx = rc.vsx * gradient_delta_factor, rc.gdy = rc.vsy * gradient_delta_factor, rc.gdz = rc.vsz * gradient_delta_factor;
  rc.szx = rc.vsx * (float)rc.X, rc.szy = rc.vsy * (float)rc.Y, rc.szz = rc.vsz * (float)rc.Z;
  rc.time_step = trunc_dist * raycast_step_factor;
  rc.R = R, rc.t = t, rc.Rinv = Rinv;
  rc.finvx = 1.f / intr[0], rc.finvy = 1.f / intr[1], rc.cx = intr[2], rc.cy = intr[3];

  uint64_t fetches = 0, hits = 0;
#ifdef _OPENMP
  if (nthreads < 1)
    nthreads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads) reduction(+ : fetches, hits)
#endif
  for (int y = 0; y < rows; ++y)
  {
    float* prow = points ? (float*)((char*)points + (size_t)y * points_pitch_bytes) : NULL;
    float* nrow = (float*)((char*)normals + (size_t)y * normals_pitch_bytes);
    uint16_t* drow = depth ? (uint16_t*)((char*)depth + (size_t)y * depth_pitch_bytes) : NULL;
    for (int x = 0; x < cols; ++x)
    {
      float v[3], n[3];
      int key;
      const int hit = raycast_pixel(&rc, x, y, v, n, &fetches, z_own0, z_own1, &key);
      hits += (uint64_t)hit;
      if (keys)
        ((int32_t*)((char*)keys + (size_t)y * keys_pitch_bytes))[x] = key;
      if (hit)
      {
        nrow[4 * x + 0] = n[0], nrow[4 * x + 1] = n[1], nrow[4 * x + 2] = n[2], nrow[4 * x + 3] = 0.f;
        if (prow)
          prow[4 * x + 0] = v[0], prow[4 * x + 1] = v[1], prow[4 * x + 2] = v[2], prow[4 * x + 3] = 0.f;
        if (drow)
        { /* cu:280  static_cast<ushort>(vertex.z * 1000): truncation; saturating as the GPU conversion does */
          const float mm = v[2] * 1000.f;
          drow[x] = (uint16_t)(mm >= 65535.f ? 65535 : (mm > 0.f ? (int)mm : 0));
        }
      }
      else
      {
        nrow[4 * x + 0] = nrow[4 * x + 1] = nrow[4 * x + 2] = nrow[4 * x + 3] = NAN;
        if (prow)
          prow[4 * x + 0] = prow[4 * x + 1] = prow[4 * x + 2] = prow[4 * x + 3] = NAN;
        if (drow)
          drow[x] = 0;
      }
    }
  }
  if (stats)
  {
    stats[0] += fetches;
    stats[1] += hits;
  }
}

/* ------------------------------------------------------------------------------------------
 * Depth pre-processing: imgproc.cu.
 * ------------------------------------------------------------------------------------------ */
#define U16_AT(base, pitch, y, x) (((const uint16_t*)((const char*)(base) + (size_t)(y) * (pitch)))[x])
#define F4_AT(base, pitch, y, x) ((const float*)((const char*)(base) + (size_t)(y) * (pitch)) + 4 * (size_t)(x))
#define F4_OUT(base, pitch, y, x) ((float*)((char*)(base) + (size_t)(y) * (pitch)) + 4 * (size_t)(x))

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
/* int * int as the GPU does it (wraps; differences above 46340 mm would overflow a signed product) */
static inline int wrap_sq(int d) { return (int)((uint32_t)d * (uint32_t)d); }

void yko_bilateral_filter(const uint16_t* src, size_t src_pitch, uint16_t* dst, size_t dst_pitch, int rows, int cols,
                          int ksz, float sigma_spatial, float sigma_depth)
{
  sigma_depth *= 1000; /* :48 metres -> mm */
  const float s2 = 0.5f / (sigma_spatial * sigma_spatial), d2 = 0.5f / (sigma_depth * sigma_depth); /* :54 */
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x)
    {
      const int value = U16_AT(src, src_pitch, y, x);
      const int tx = imin(x - ksz / 2 + ksz, cols - 1); /* :20-21 (exclusive bounds, clipped to size - 1) */
      const int ty = imin(y - ksz / 2 + ksz, rows - 1);
      float sum1 = 0, sum2 = 0;
      for (int cy = imax(y - ksz / 2, 0); cy < ty; ++cy)
        for (int cx = imax(x - ksz / 2, 0); cx < tx; ++cx)
        {
          const int depth = U16_AT(src, src_pitch, cy, cx);
          const float space2 = (float)((x - cx) * (x - cx) + (y - cy) * (y - cy));
          const float color2 = (float)wrap_sq(value - depth);
          const float weight = expf(-(space2 * s2 + color2 * d2));
          sum1 += (float)depth * weight;
          sum2 += weight;
        }
      const float q = sum1 / sum2;
      ((uint16_t*)((char*)dst + (size_t)y * dst_pitch))[x] = isnan(q) ? 0 : (uint16_t)(int)nearbyintf(q); /* :41 */
    }
}

void yko_truncate_depth(uint16_t* depth, size_t pitch, int rows, int cols, float max_dist_m)
{
  const float mm = max_dist_m * 1000.f; /* :83 static_cast<ushort>(max_dist * 1000.f) */
  const uint16_t max_dist = (uint16_t)(mm >= 65535.f ? 65535 : (mm > 0.f ? (int)mm : 0));
  for (int y = 0; y < rows; ++y)
  {
    uint16_t* row = (uint16_t*)((char*)depth + (size_t)y * pitch);
    for (int x = 0; x < cols; ++x)
      if (row[x] > max_dist)
        row[x] = 0;
  }
}

void yko_depth_pyramid(const uint16_t* src, size_t src_pitch, int src_rows, int src_cols, uint16_t* dst,
                       size_t dst_pitch, float sigma_depth)
{
  const float sigma_depth_mult3 = sigma_depth * 1000 * 3; /* :129, :134 */
  const int rows = src_rows / 2, cols = src_cols / 2;
  const int D = 5;
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x)
    {
      const int center = U16_AT(src, src_pitch, 2 * y, 2 * x);
      const int tx = imin(2 * x - D / 2 + D, src_cols - 1);
      const int ty = imin(2 * y - D / 2 + D, src_rows - 1);
      int sum = 0, count = 0;
      for (int cy = imax(0, 2 * y - D / 2); cy < ty; ++cy)
        for (int cx = imax(0, 2 * x - D / 2); cx < tx; ++cx)
        {
          const int val = U16_AT(src, src_pitch, cy, cx);
          if ((float)abs(val - center) < sigma_depth_mult3)
          {
            sum += val;
            ++count;
          }
        }
      ((uint16_t*)((char*)dst + (size_t)y * dst_pitch))[x] = (uint16_t)(count == 0 ? 0 : sum / count);
    }
}

/* device.hpp:57-62 */
static inline void reproj(float finvx, float finvy, float cx, float cy, int u, int v, float z, float out[3])
{
  out[0] = z * ((float)u - cx) * finvx;
  out[1] = z * ((float)v - cy) * finvy;
  out[2] = z;
}

void yko_compute_points_normals(const uint16_t* depth, size_t depth_pitch, int rows, int cols, const float intr[4],
                                float* points, size_t points_pitch, float* normals, size_t normals_pitch)
{
  const float finvx = 1.f / intr[0], finvy = 1.f / intr[1], cx = intr[2], cy = intr[3]; /* precomp.cpp:73-76 */
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x)
    {
      float* p = F4_OUT(points, points_pitch, y, x);
      float* n = F4_OUT(normals, normals_pitch, y, x);
      p[0] = p[1] = p[2] = p[3] = n[0] = n[1] = n[2] = n[3] = NAN; /* :219 */
      if (x >= cols - 1 || y >= rows - 1)
        continue;
      const float z00 = (float)U16_AT(depth, depth_pitch, y, x) * 0.001f;
      const float z01 = (float)U16_AT(depth, depth_pitch, y, x + 1) * 0.001f;
      const float z10 = (float)U16_AT(depth, depth_pitch, y + 1, x) * 0.001f;
      if (z00 * z01 * z10 != 0)
      {
        float v00[3], v01[3], v10[3];
        reproj(finvx, finvy, cx, cy, x, y, z00, v00);
        reproj(finvx, finvy, cx, cy, x + 1, y, z01, v01);
        reproj(finvx, finvy, cx, cy, x, y + 1, z10, v10);
        const float ax = v01[0] - v00[0], ay = v01[1] - v00[1], az = v01[2] - v00[2];
        const float bx = v10[0] - v00[0], by = v10[1] - v00[1], bz = v10[2] - v00[2];
        float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx; /* temp_utils.hpp:105-108 */
        normalize3(&nx, &ny, &nz);
        n[0] = -nx, n[1] = -ny, n[2] = -nz, n[3] = 0.f;
        p[0] = v00[0], p[1] = v00[1], p[2] = v00[2], p[3] = 0.f;
      }
    }
}

void yko_resize_points_normals(const float* points, size_t points_pitch, const float* normals, size_t normals_pitch,
                               int src_rows, int src_cols, float* points_out, size_t points_out_pitch,
                               float* normals_out, size_t normals_out_pitch)
{
  const int rows = src_rows / 2, cols = src_cols / 2;
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x)
    {
      float* vo = F4_OUT(points_out, points_out_pitch, y, x);
      float* no = F4_OUT(normals_out, normals_out_pitch, y, x);
      vo[0] = vo[1] = vo[2] = no[0] = no[1] = no[2] = NAN; /* :357 */
      vo[3] = no[3] = 0.f;
      const int xs = 2 * x, ys = 2 * y;
      const float* d00 = F4_AT(points, points_pitch, ys + 0, xs + 0);
      const float* d01 = F4_AT(points, points_pitch, ys + 0, xs + 1);
      const float* d10 = F4_AT(points, points_pitch, ys + 1, xs + 0);
      const float* d11 = F4_AT(points, points_pitch, ys + 1, xs + 1);
      if (!isnan(d00[0] * d01[0] * d10[0] * d11[0]))
      {
        const float* n00 = F4_AT(normals, normals_pitch, ys + 0, xs + 0);
        const float* n01 = F4_AT(normals, normals_pitch, ys + 0, xs + 1);
        const float* n10 = F4_AT(normals, normals_pitch, ys + 1, xs + 0);
        const float* n11 = F4_AT(normals, normals_pitch, ys + 1, xs + 1);
        for (int k = 0; k < 3; ++k)
        {
          vo[k] = (d00[k] + d01[k] + d10[k] + d11[k]) * 0.25f;
          no[k] = (n00[k] + n01[k] + n10[k] + n11[k]) * 0.25f;
        }
      }
    }
}

/* ------------------------------------------------------------------------------------------
 * Debug display: imgproc.cu:400-561.  Parity unpinned (the reference holds no image for it); the
 * specular term's __powf(v, 20) (:441, the CUDA hardware's exp2(20 * log2 v) approximation, which no
 * CPU routine reproduces to the bit) is restated as v^20 by five multiplications, so a grey level may
 * differ from the reference's by one.
 * ------------------------------------------------------------------------------------------ */
static inline float saturatef(float v) { return v > 0.f ? (v < 1.f ? v : 1.f) : 0.f; } /* __saturatef: NaN -> 0 */

static inline float pow20(float v)
{
  const float v2 = v * v, v4 = v2 * v2, v5 = v4 * v, v10 = v5 * v5;
  return v10 * v10;
}

/* :414-418 / :467-471, stored b, g, r, 0 (:445-450) */
static inline void backdrop(int y, int rows, uint8_t out[4])
{
  const float w = (float)y / (float)rows;
  const float b = (4.f / 255.f) * (1 - w) + (236.f / 255.f) * w;
  const float g = (2.f / 255.f) * (1 - w) + (120.f / 255.f) * w;
  out[0] = (uint8_t)(saturatef(b) * 255.f);
  out[1] = out[2] = (uint8_t)(saturatef(g) * 255.f);
  out[3] = 0;
}

/* :422-443 / :475-496 */
static inline void phong(const float P[3], const float* N, const float light[3], uint8_t out[4])
{
  float Lx = light[0] - P[0], Ly = light[1] - P[1], Lz = light[2] - P[2];
  normalize3(&Lx, &Ly, &Lz);
  float Vx = 0.f - P[0], Vy = 0.f - P[1], Vz = 0.f - P[2];
  normalize3(&Vx, &Vy, &Vz);
  const float nl = dot3(N[0], N[1], N[2], Lx, Ly, Lz);
  float Rx = 2 * N[0] * nl - Lx, Ry = 2 * N[1] * nl - Ly, Rz = 2 * N[2] * nl - Lz;
  normalize3(&Rx, &Ry, &Rz);
  const float rv = dot3(Rx, Ry, Rz, Vx, Vy, Vz);
  const float Ix = 0.3f + 0.5f * (nl > 0.f ? nl : 0.f) + 0.2f * pow20(rv > 0.f ? rv : 0.f);
  out[0] = out[1] = out[2] = (uint8_t)(saturatef(Ix) * 255.f);
  out[3] = 0;
}

void yko_render_image_points(const float* points, size_t points_pitch, const float* normals, size_t normals_pitch,
                             int rows, int cols, const float light_pose[3], uint8_t* image, size_t image_pitch)
{
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x)
    {
      const float* p = F4_AT(points, points_pitch, y, x);
      uint8_t* out = image + (size_t)y * image_pitch + 4 * (size_t)x;
      if (isnan(p[0]))
        backdrop(y, rows, out);
      else
        phong(p, F4_AT(normals, normals_pitch, y, x), light_pose, out);
    }
}

void yko_render_image_depth(const uint16_t* depth, size_t depth_pitch, const float* normals, size_t normals_pitch,
                            int rows, int cols, const float intr[4], const float light_pose[3], uint8_t* image,
                            size_t image_pitch)
{
  const float finvx = 1.f / intr[0], finvy = 1.f / intr[1], cx = intr[2], cy = intr[3]; /* precomp.cpp:73-76 */
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x)
    {
      const int d = U16_AT(depth, depth_pitch, y, x);
      uint8_t* out = image + (size_t)y * image_pitch + 4 * (size_t)x;
      if (d == 0)
        backdrop(y, rows, out);
      else
      {
        float P[3];
        reproj(finvx, finvy, cx, cy, x, y, (float)d * 0.001f, P); /* :422 */
        phong(P, F4_AT(normals, normals_pitch, y, x), light_pose, out);
      }
    }
}

/* :545-547: float -> unsigned char truncates; saturating, NaN -> 0 (a unit normal gives 38 .. 216) */
static inline uint8_t tangent_byte(float n, float k)
{
  const float v = (5.f - n * k) * 25.5f;
  return v > 0.f ? (v < 255.f ? (uint8_t)v : 255) : 0;
}

void yko_render_tangent_colors(const float* normals, size_t normals_pitch, int rows, int cols, uint8_t* image,
                               size_t image_pitch)
{
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x)
    {
      const float* n = F4_AT(normals, normals_pitch, y, x);
      uint8_t* out = image + (size_t)y * image_pitch + 4 * (size_t)x;
      out[0] = tangent_byte(n[2], 3.5f); /* :549 make_uchar4(b, g, r, 0) */
      out[1] = tangent_byte(n[1], 2.5f);
      out[2] = tangent_byte(n[0], 3.5f);
      out[3] = 0;
    }
}

/* ------------------------------------------------------------------------------------------
 * Projective ICP, one linearisation: proj_icp.cu.
 * ------------------------------------------------------------------------------------------ */
void yko_icp_accumulate(const float R[9], const float t[3], const float intr[4], int level, float dist_thres,
                        float angle_thres, const float* vcurr, size_t vcurr_pitch, const float* ncurr,
                        size_t ncurr_pitch, const float* vprev, size_t vprev_pitch, const float* nprev,
                        size_t nprev_pitch, int rows, int cols, double sums[27], uint64_t counts[6], uint8_t* codes)
{
  /* projective_icp.cpp:10-22 */
  const float min_cosine = (float)cos((double)angle_thres);
  const float dist2_thres = dist_thres * dist_thres;
  const int div = 1 << level;
  const float fx = intr[0] / (float)div, fy = intr[1] / (float)div, cx = intr[2] / (float)div, cy = intr[3] / (float)div;
  const float frows = (float)rows, fcols = (float)cols; /* projective_icp.cpp:200-201 */

  for (int k = 0; k < 27; ++k)
    sums[k] = 0.0;
  for (int k = 0; k < 6; ++k)
    counts[k] = 0;

  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x)
    {
      int code = 0;
      float row[7] = { 0, 0, 0, 0, 0, 0, 0 };
      do
      { /* find_coresp, proj_icp.cu:74-102 */
        const float* sp = F4_AT(vcurr, vcurr_pitch, y, x);
        if (isnan(sp[0]))
        {
          code = 40;
          break;
        }
        const float sx = dot3(R[0], R[1], R[2], sp[0], sp[1], sp[2]) + t[0];
        const float sy = dot3(R[3], R[4], R[5], sp[0], sp[1], sp[2]) + t[1];
        const float sz = dot3(R[6], R[7], R[8], sp[0], sp[1], sp[2]) + t[2];
        if (sz <= 0)
        {
          code = 80;
          break;
        }
        const float coox = fmaf(fx, sx / sz, cx), cooy = fmaf(fy, sy / sz, cy); /* :24-31 */
        if (!(coox >= 0 && cooy >= 0 && coox < fcols && cooy < frows))
        {
          code = 80;
          break;
        }
        const int px = (int)floorf(coox), py = (int)floorf(cooy); /* point-filtered tex2D */
        const float* dp = F4_AT(vprev, vprev_pitch, py, px);
        if (isnan(dp[0]))
        {
          code = 120;
          break;
        }
        const float ex = sx - dp[0], ey = sy - dp[1], ez = sz - dp[2];
        if (dot3(ex, ey, ez, ex, ey, ez) > dist2_thres)
        {
          code = 160;
          break;
        }
        const float* nc = F4_AT(ncurr, ncurr_pitch, y, x);
        const float nsx = dot3(R[0], R[1], R[2], nc[0], nc[1], nc[2]);
        const float nsy = dot3(R[3], R[4], R[5], nc[0], nc[1], nc[2]);
        const float nsz = dot3(R[6], R[7], R[8], nc[0], nc[1], nc[2]);
        const float* nd = F4_AT(nprev, nprev_pitch, py, px);
        const float cosine = fabsf(dot3(nsx, nsy, nsz, nd[0], nd[1], nd[2]));
        if (cosine < min_cosine)
        { /* literal: a NaN cosine passes this test and poisons the sums, which makes the caller's
             determinant NaN and the frame a tracking failure (projective_icp.cpp:217-222) */
          code = 200;
          break;
        }
        /* :354-359  row = [cross(s, n), n, dot(n, d - s)] */
        row[0] = sy * nd[2] - sz * nd[1];
        row[1] = sz * nd[0] - sx * nd[2];
        row[2] = sx * nd[1] - sy * nd[0];
        row[3] = nd[0], row[4] = nd[1], row[5] = nd[2];
        row[6] = dot3(nd[0], nd[1], nd[2], dp[0] - sx, dp[1] - sy, dp[2] - sz);
      } while (0);

      if (codes)
        codes[(size_t)y * (size_t)cols + (size_t)x] = (uint8_t)code;
      counts[code / 40] += 1;
      if (code == 0)
      {
        int k = 0;
        for (int i = 0; i < 6; ++i)
          for (int j = i; j < 7; ++j)
            sums[k++] += (double)(row[i] * row[j]); /* the product is formed in fp32, cu:122-335 */
      }
    }
}

/* ------------------------------------------------------------------------------------------
 * Marching cubes: src/mc/marching_cubes.cpp.
 * ------------------------------------------------------------------------------------------ */
/* :22-31 */
static float mc_get_offset(float value1, float value2, float value_desired)
{
  const double delta = value2 - value1;
  if (delta == 0.0)
    return 0.5f;
  return (float)((value_desired - value1) / delta);
}

/* :33-37 */
static const int mc_cube_offsets[8][3] = {
  { 1, 1, 0 }, { 1, 0, 0 }, { 0, 0, 0 }, { 0, 1, 0 }, { 1, 1, 1 }, { 1, 0, 1 }, { 0, 0, 1 }, { 0, 1, 1 },
};

/* the cubes of one x index (the reference's OpenMP loop is over x, :163-170); triangles[] starts at this
 * slice's first triangle, capacity counts from there */
static uint64_t mc_slice(const uint32_t* volume, const int dims[3], int min_weight, double scale, const int* tri_table,
                         const int* edge_connections, int x, float* triangles, uint64_t capacity)
{
  const int X = dims[0], Y = dims[1], Z = dims[2];
  uint64_t n = 0;
  {
    for (int y = 0; y + 1 < Y; ++y)
      for (int z = 0; z + 1 < Z; ++z)
      {
        /* processCube, :39-156 */
        float val[8];
        int ok = 1;
        for (int c = 0; c < 8 && ok; ++c)
        {
          const size_t idx = (size_t)X * (size_t)Y * (size_t)(z + mc_cube_offsets[c][2]) +
                             (size_t)X * (size_t)(y + mc_cube_offsets[c][1]) + (size_t)(x + mc_cube_offsets[c][0]);
          const uint32_t word = volume[idx];
          const int w = (int)(uint16_t)(word >> 16); /* uint16_t weight, tsdf_container.h:45-49 */
          val[c] = 5.0f * yko_half2float((uint16_t)(word & 0xffffu)); /* :46 */
          if (w < min_weight)
            ok = 0; /* :51-74 */
        }
        if (!ok)
          continue;
        unsigned cubeindex = 0; /* :80-96, isolevel 0 */
        for (int c = 0; c < 8; ++c)
          if (val[c] < 0.0f)
            cubeindex |= 1u << c;
        /* :101-106 edge flags: an edge is cut iff its end corners differ in sign */
        int edge_flags = 0;
        for (int e = 0; e < 12; ++e)
          if (((cubeindex >> edge_connections[2 * e]) & 1u) != ((cubeindex >> edge_connections[2 * e + 1]) & 1u))
            edge_flags |= 1 << e;
        if (edge_flags == 0)
          continue;
        float edges[12][3];
        for (int e = 0; e < 12; ++e) /* :112-131 */
          if (edge_flags & (1 << e))
          {
            const int v0 = edge_connections[2 * e], v1 = edge_connections[2 * e + 1];
            const float offset = mc_get_offset(val[v0], val[v1], 0.0f);
            const int base[3] = { x, y, z };
            for (int k = 0; k < 3; ++k)
            {
              const float p0 = (float)(base[k] + mc_cube_offsets[v0][k]), p1 = (float)(base[k] + mc_cube_offsets[v1][k]);
              edges[e][k] = p0 + offset * (p1 - p0);
            }
          }
        const int* triangulation = tri_table + 16 * cubeindex; /* :136-153 */
        for (int i = 0; i < 5; ++i)
        {
          if (triangulation[i * 3] < 0)
            break;
          if (n < capacity)
            for (int j = 0; j < 3; ++j)
              for (int k = 0; k < 3; ++k) /* :175-183 */
                triangles[n * 9 + (uint64_t)(3 * j + k)] = (float)(scale * (double)edges[triangulation[i * 3 + j]][k]);
          ++n;
        }
      }
  }
  return n;
}

uint64_t yko_marching_cubes(const uint32_t* volume, const int dims[3], int min_weight, double scale,
                            const int* tri_table, const int* edge_connections, float* triangles, uint64_t capacity)
{
  uint64_t n = 0;
  for (int x = 0; x + 1 < dims[0]; ++x) /* :164-170 with x-1, y-1, z-1 */
    n += mc_slice(volume, dims, min_weight, scale, tri_table, edge_connections, x, triangles ? triangles + 9 * n : NULL,
                  capacity > n ? capacity - n : 0);
  return n;
}

/* The same over x in parallel, as the reference runs it (#pragma omp parallel for over x, :163): a counting pass,
 * a prefix sum over the x slices, an emitting pass into each slice's own range -- the triangles come out in the
 * serial routine's order (the reference's own order is nondeterministic: its inserts share one critical
 * section, :165-171).  For the timed CPU baseline of bench.py. */
uint64_t yko_marching_cubes_mt(const uint32_t* volume, const int dims[3], int min_weight, double scale,
                               const int* tri_table, const int* edge_connections, float* triangles, uint64_t capacity,
                               int nthreads)
{
  const int nx = dims[0] > 1 ? dims[0] - 1 : 0;
  uint64_t* start = (uint64_t*)calloc((size_t)nx + 1, sizeof(uint64_t));
  if (!start)
    return 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
  for (int x = 0; x < nx; ++x)
    start[x + 1] = mc_slice(volume, dims, min_weight, scale, tri_table, edge_connections, x, NULL, 0);
  for (int x = 0; x < nx; ++x)
    start[x + 1] += start[x];
  const uint64_t n = start[nx];
  if (triangles && n <= capacity)
  {
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int x = 0; x < nx; ++x)
      mc_slice(volume, dims, min_weight, scale, tri_table, edge_connections, x, triangles + 9 * start[x],
               start[x + 1] - start[x]);
  }
  free(start);
  return n;
}

/* ------------------------------------------------------------------------------------------
 * Cloud extraction: tsdf_volume.cu:464-800.
 * ------------------------------------------------------------------------------------------ */
static inline int cloud_voxel(const uint32_t* vol, const int dims[3], int x, int y, int z, float* F)
{ /* FullScan6::fetch, cu:485-489: unpack_tsdf; returns "W != 0 && F != 1.f" (cu:524, 532, ...) */
  const uint32_t w = vol[(size_t)x + (size_t)dims[0] * ((size_t)y + (size_t)dims[1] * (size_t)z)];
  *F = yko_half2float((uint16_t)(w & 0xffffu));
  return (w >> 16) != 0 && *F != 1.f;
}

uint64_t yko_extract_cloud(const uint32_t* volume, const int dims[3], const float voxel_size[3], const float R[9],
                           const float t[3], float* points, uint64_t capacity)
{
  uint64_t n = 0;
  const int X = dims[0], Y = dims[1], Z = dims[2];
  for (int z = 0; z < Z - 1; ++z) /* cu:514 */
    for (int y = 0; y < Y; ++y)
      for (int x = 0; x < X; ++x)
      {
        float F;
        if (!cloud_voxel(volume, dims, x, y, z, &F)) /* cu:522-524 */
          continue;
        const float V[3] = { ((float)x + 0.5f) * voxel_size[0], ((float)y + 0.5f) * voxel_size[1],
                             ((float)z + 0.5f) * voxel_size[2] }; /* cu:508-509, 526 */
        for (int axis = 0; axis < 3; ++axis) /* cu:528-591: dx, dy, dz */
        {
          const int nx = x + (axis == 0), ny = y + (axis == 1), nz = z + (axis == 2);
          if (nx >= X || ny >= Y) /* (z + 1 < Z by the loop) */
            continue;
          float Fn;
          if (!cloud_voxel(volume, dims, nx, ny, nz, &Fn))
            continue;
          if (!((F > 0 && Fn < 0) || (F < 0 && Fn > 0)))
            continue;
          float p[3] = { V[0], V[1], V[2] };
          const float Vn = V[axis] + voxel_size[axis];
          const float d_inv = 1.f / (fabsf(F) + fabsf(Fn));
          p[axis] = (V[axis] * fabsf(Fn) + Vn * fabsf(F)) * d_inv;
          if (n < capacity)
          { /* aff * p, device.hpp:106-111 */
            float* o = points + 4 * n;
            o[0] = dot3(R[0], R[1], R[2], p[0], p[1], p[2]) + t[0];
            o[1] = dot3(R[3], R[4], R[5], p[0], p[1], p[2]) + t[1];
            o[2] = dot3(R[6], R[7], R[8], p[0], p[1], p[2]) + t[2];
            o[3] = 0.f;
          }
          ++n;
        }
      }
  return n;
}

void yko_extract_normals(const uint32_t* volume, const int dims[3], const float voxel_size[3], const float* points,
                         uint64_t n_points, const float R[9], const float t[3], const float Rinv[9],
                         float gradient_delta_factor, float* normals)
{
  Raycaster rc; /* for interpolate(): cu:167-192 */
  memset(&rc, 0, sizeof(rc));
  rc.vol = volume;
  rc.X = dims[0], rc.Y = dims[1], rc.Z = dims[2];
  rc.ivx = 1.f / voxel_size[0], rc.ivy = 1.f / voxel_size[1], rc.ivz = 1.f / voxel_size[2]; /* cu:684-689 */
  const float gdx = voxel_size[0] * gradient_delta_factor, gdy = voxel_size[1] * gradient_delta_factor,
              gdz = voxel_size[2] * gradient_delta_factor; /* cu:786 */
  for (uint64_t i = 0; i < n_points; ++i)
  {
    const float* q = points + 4 * i;
    float n[3] = { NAN, NAN, NAN };
    /* cu:715  point = Rinv * (tr(p) - aff.t) */
    const float wx = q[0] - t[0], wy = q[1] - t[1], wz = q[2] - t[2];
    const float px = dot3(Rinv[0], Rinv[1], Rinv[2], wx, wy, wz), py = dot3(Rinv[3], Rinv[4], Rinv[5], wx, wy, wz),
                pz = dot3(Rinv[6], Rinv[7], Rinv[8], wx, wy, wz);
    /* cu:694-701  __float2int_rn; the comparisons below reject NaN and anything out of int range first */
    const float fx = nearbyintf(px * rc.ivx), fy = nearbyintf(py * rc.ivy), fz = nearbyintf(pz * rc.ivz);
    if (fx > 1.f && fy > 1.f && fz > 1.f && fx < (float)(rc.X - 2) && fy < (float)(rc.Y - 2) && fz < (float)(rc.Z - 2))
    { /* cu:718-751 */
      const float Fx1 = interpolate(&rc, (px + gdx) * rc.ivx, py * rc.ivy, pz * rc.ivz);
      const float Fx2 = interpolate(&rc, (px - gdx) * rc.ivx, py * rc.ivy, pz * rc.ivz);
      const float Fy1 = interpolate(&rc, px * rc.ivx, (py + gdy) * rc.ivy, pz * rc.ivz);
      const float Fy2 = interpolate(&rc, px * rc.ivx, (py - gdy) * rc.ivy, pz * rc.ivz);
      const float Fz1 = interpolate(&rc, px * rc.ivx, py * rc.ivy, (pz + gdz) * rc.ivz);
      const float Fz2 = interpolate(&rc, px * rc.ivx, py * rc.ivy, (pz - gdz) * rc.ivz);
      const float gx = (Fx1 - Fx2) / gdx, gy = (Fy1 - Fy2) / gdy, gz = (Fz1 - Fz2) / gdz;
      /* cu:753  n = normalized(aff.R * n) */
      n[0] = dot3(R[0], R[1], R[2], gx, gy, gz);
      n[1] = dot3(R[3], R[4], R[5], gx, gy, gz);
      n[2] = dot3(R[6], R[7], R[8], gx, gy, gz);
      normalize3(&n[0], &n[1], &n[2]);
    }
    float* o = normals + 4 * i;
    o[0] = n[0], o[1] = n[1], o[2] = n[2], o[3] = 0.f;
  }
}

int yko_max_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
