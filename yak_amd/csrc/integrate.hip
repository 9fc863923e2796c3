// integrate.hip -- TsdfVolume::integrate for gfx950 (MI355X), hand-written HIP.
//
// Replaces kfusion::device::integrate + TsdfIntegrator (reference yak/src/cuda/tsdf_volume.cu:51-140)
// and clear_volume (:14-40).  Same per-voxel arithmetic, different decomposition:
//
//   reference (CUDA)                         here (CDNA4)
//   ---------------------------------------  ----------------------------------------------------
//   thread = one (x,y) column, serial         lane = VX (4) consecutive x voxels = one 16-byte
//   512-step z loop, 4-byte ld/st             global_load/store_dwordx4; a wave64 covers 1 KiB of
//                                             one x-row run per instruction
//   grid = 1024 blocks over (x,y)             grid = (plane tiles) x (z chunks): thousands of
//                                             workgroups, each walking `z_chunk` layers
//   pose/intrinsics in a by-value struct      the same, but read once into SGPRs (wave-uniform)
//   dists through a texture reference         plain 2-byte gathers; the 600 KiB image stays in L2
//   ld/st.global.cs                           non-temporal load/store (the volume is streamed once)
//
// Bit parity with the scalar oracle: the reference accumulates the camera-space position with
// one float addition per z step (vc += zstep, tsdf_volume.cu:76), so vc(z) != vc0 + z*zstep in
// floating point.  A workgroup that starts at layer zs therefore REPLAYS the zs additions
// (3 VALU per voxel per step, no memory) before its first voxel.  All FMAs are explicit and this
// file is compiled with -ffp-contract=off; '/' and sqrtf are the correctly rounded forms
// (-fhip-fp32-correctly-rounded-divide-sqrt), matching the oracle's IEEE arithmetic.
#include "yakhip_internal.h"

namespace yakhip
{
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int VX>
struct Vec;
template <>
struct Vec<4>
{
  typedef u32x4 type;
};
template <>
struct Vec<2>
{
  typedef u32x2 type;
};
template <>
struct Vec<1>
{
  typedef uint32_t type;
};

template <int VX>
__device__ __forceinline__ uint32_t vget(const typename Vec<VX>::type& v, int i)
{
  return v[i];
}
template <>
__device__ __forceinline__ uint32_t vget<1>(const uint32_t& v, int)
{
  return v;
}
template <int VX>
__device__ __forceinline__ void vset(typename Vec<VX>::type& v, int i, uint32_t x)
{
  v[i] = x;
}
template <>
__device__ __forceinline__ void vset<1>(uint32_t& v, int, uint32_t x)
{
  v = x;
}

// One voxel's measurement: tsdf_volume.cu:78-96.  Returns true when the voxel takes an update.
__device__ __forceinline__ bool measure(const IntegrateParams& p, float vcx, float vcy, float vcz, float& tsdf)
{
  // :89 tests 'vc.z <= 0' after the fetch; testing it first skips exactly the same voxels and keeps
  // the projection away from x/0.
  if (!(vcz > 0.f))
    return false;

  // Projector, device.hpp:46-52
  const float coox = __builtin_fmaf(p.fx, vcx / vcz, p.cx);
  const float cooy = __builtin_fmaf(p.fy, vcy / vcz, p.cy);

  // :83
  if (coox < 0.f || cooy < 0.f || coox >= (float)p.cols || cooy >= (float)p.rows)
    return false;

  // :87 point fetch at (floor(coo.x), floor(coo.y)); coo >= 0 here so truncation == floor
  const int px = (int)coox, py = (int)cooy;
  const uint16_t hbits = *(const uint16_t*)(p.dists + (size_t)py * p.dists_pitch + (size_t)px * 2u);
  const float Dp = half_bits_to_float(hbits);
  if (Dp == 0.f)
    return false;

  // :92
  const float sdf = Dp - __builtin_sqrtf(dot3(vcx, vcy, vcz, vcx, vcy, vcz));
  if (!(sdf >= -p.trunc))
    return false;

  // :96
  tsdf = __builtin_fminf(1.f, sdf * p.trunc_inv);
  return true;
}

// Weighted running average of one packed voxel: tsdf_volume.cu:98-106, device.hpp:69-77.
__device__ __forceinline__ uint32_t fuse(uint32_t voxel, float tsdf, int max_weight)
{
  const float tsdf_prev = half_bits_to_float(voxel & 0xffffu);
  const int weight_prev = (int)(int16_t)(uint16_t)(voxel >> 16);
  const float tsdf_new = __builtin_fmaf(tsdf_prev, (float)weight_prev, tsdf) / (float)(weight_prev + 1);
  const int weight_new = min(weight_prev + 1, max_weight);
  return float_to_half_bits(tsdf_new) | ((uint32_t)(uint16_t)weight_new << 16);
}

// VX voxels per lane along x, ZU layers in flight per loop trip.
template <int VX, int ZU, bool NT>
__global__ __launch_bounds__(256) void integrate_kernel(const IntegrateParams p)
{
  typedef typename Vec<VX>::type vec_t;

  const uint32_t group = blockIdx.x * 256u + threadIdx.x;  // VX-voxel group inside one z layer
  if (group >= p.plane / VX)
    return;
  const uint32_t elem = group * VX;  // x + y*X of the first voxel; X % VX == 0 so no row straddle
  const int y = (int)(elem / (uint32_t)p.X);
  const int x0 = (int)(elem - (uint32_t)y * (uint32_t)p.X);

  const int zs = p.z0 + (int)blockIdx.y * p.z_chunk;
  const int ze = min(zs + p.z_chunk, p.z1);

  // :69
  const float zsx = p.R[2] * p.vsz, zsy = p.R[5] * p.vsz, zsz = p.R[8] * p.vsz;

  // :71-72
  float vcx[VX], vcy[VX], vcz[VX];
  const float vxy = (float)y * p.vsy;
#pragma unroll
  for (int v = 0; v < VX; ++v)
  {
    const float vxx = (float)(x0 + v) * p.vsx;
    vcx[v] = dot3(p.R[0], p.R[1], p.R[2], vxx, vxy, 0.f) + p.t[0];
    vcy[v] = dot3(p.R[3], p.R[4], p.R[5], vxx, vxy, 0.f) + p.t[1];
    vcz[v] = dot3(p.R[6], p.R[7], p.R[8], vxx, vxy, 0.f) + p.t[2];
  }

  // Replay the additions the reference performed on layers [0, zs) (tsdf_volume.cu:76).
  for (int i = 0; i < zs; ++i)
  {
#pragma unroll
    for (int v = 0; v < VX; ++v)
    {
      vcx[v] += zsx;
      vcy[v] += zsy;
      vcz[v] += zsz;
    }
  }

  uint32_t* ptr = p.vol + (size_t)(zs - p.z0) * p.plane + elem;

  for (int z = zs; z < ze; z += ZU, ptr += (size_t)ZU * p.plane)
  {
    float tsdf[ZU][VX];
    uint32_t hit[ZU];  // bit v set: voxel v of layer z+u takes an update

    // phase 1: measurements for ZU layers (dists gathers in flight together)
#pragma unroll
    for (int u = 0; u < ZU; ++u)
    {
      hit[u] = 0u;
#pragma unroll
      for (int v = 0; v < VX; ++v)
      {
        tsdf[u][v] = 0.f;
        if (z + u < ze && measure(p, vcx[v], vcy[v], vcz[v], tsdf[u][v]))
          hit[u] |= 1u << v;
        vcx[v] += zsx;
        vcy[v] += zsy;
        vcz[v] += zsz;
      }
    }

    // phase 2: fetch the voxel groups that changed (the reference reads only updated voxels, :98-100)
    vec_t vox[ZU];
#pragma unroll
    for (int u = 0; u < ZU; ++u)
    {
      if (hit[u])
      {
        const vec_t* src = (const vec_t*)(ptr + (size_t)u * p.plane);
        vox[u] = NT ? __builtin_nontemporal_load(src) : *src;
      }
    }

    // phase 3: running average + write back
#pragma unroll
    for (int u = 0; u < ZU; ++u)
    {
      if (hit[u])
      {
        vec_t out = vox[u];
#pragma unroll
        for (int v = 0; v < VX; ++v)
          if (hit[u] & (1u << v))
            vset<VX>(out, v, fuse(vget<VX>(vox[u], v), tsdf[u][v], p.max_weight));
        vec_t* dst = (vec_t*)(ptr + (size_t)u * p.plane);
        if (NT)
          __builtin_nontemporal_store(out, dst);
        else
          *dst = out;
      }
    }
  }
}

template <int VX, int ZU>
static int launch_variant(const yakhip_ctx* ctx, const IntegrateParams& p, dim3 grid)
{
  if (ctx->nontemporal)
    hipLaunchKernelGGL((integrate_kernel<VX, ZU, true>), grid, dim3(256), 0, ctx->stream, p);
  else
    hipLaunchKernelGGL((integrate_kernel<VX, ZU, false>), grid, dim3(256), 0, ctx->stream, p);
  return check(hipGetLastError(), "integrate_kernel launch");
}

int launch_integrate(const yakhip_ctx* ctx, const IntegrateParams& p_in)
{
  IntegrateParams p = p_in;
  const uintptr_t base = (uintptr_t)p.vol;

  // widest vector the layout allows: rows must not straddle a lane's group and the slab base,
  // every row start and every layer start must be aligned to the vector size
  int vx = 1;
  if (p.X % 4 == 0 && base % 16 == 0)
    vx = 4;
  else if (p.X % 2 == 0 && base % 8 == 0)
    vx = 2;

  const uint32_t groups = p.plane / (uint32_t)vx;
  const uint32_t blocks_x = (groups + 255u) / 256u;
  const int layers = p.z1 - p.z0;

  // z_chunk: enough workgroups to fill 256 CUs several times over, but long enough that the
  // replay of layers [0, zs) stays a small fraction of the chunk's own work.
  int chunk = ctx->z_chunk;
  if (chunk <= 0)
  {
    const uint32_t want_blocks = 2048u;
    uint32_t nchunks = (want_blocks + blocks_x - 1u) / blocks_x;
    if (nchunks < 1u)
      nchunks = 1u;
    chunk = (int)((layers + (int)nchunks - 1) / (int)nchunks);
    if (chunk < 16)
      chunk = 16;
  }
  if (chunk > layers)
    chunk = layers;
  chunk = (chunk + 1) & ~1;  // whole ZU=2 trips
  p.z_chunk = chunk;
  const uint32_t nchunks = (uint32_t)((layers + chunk - 1) / chunk);
  if (nchunks > 65535u)
    return fail(YAKHIP_ERANGE, "integrate: %u z-chunks exceed the grid limit; raise z_chunk", nchunks);

  const dim3 grid(blocks_x, nchunks);
  switch (vx)
  {
    case 4:
      return launch_variant<4, 2>(ctx, p, grid);
    case 2:
      return launch_variant<2, 2>(ctx, p, grid);
    default:
      return launch_variant<1, 2>(ctx, p, grid);
  }
}

// ---- clear: tsdf_volume.cu:14-40.  pack_tsdf(0.f, 0) is the all-zero word. ----------------------
__global__ __launch_bounds__(256) void clear_kernel_x4(u32x4* data, size_t n4)
{
  const u32x4 zero = { 0u, 0u, 0u, 0u };
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256u)
    __builtin_nontemporal_store(zero, data + i);
}

__global__ __launch_bounds__(256) void clear_kernel_x1(uint32_t* data, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u)
    data[i] = 0u;
}

int launch_clear(const yakhip_ctx* ctx, uint32_t* data, size_t n_voxels)
{
  // head (to 16-byte alignment) and tail with the scalar kernel, body with 16-byte stores
  const uintptr_t addr = (uintptr_t)data;
  size_t head = ((16u - (addr & 15u)) & 15u) / 4u;
  if (head > n_voxels)
    head = n_voxels;
  const size_t body4 = (n_voxels - head) / 4u;
  const size_t tail = n_voxels - head - body4 * 4u;

  if (head)
    hipLaunchKernelGGL(clear_kernel_x1, dim3(1), dim3(256), 0, ctx->stream, data, head);
  if (body4)
  {
    size_t blocks = (body4 + 255u) / 256u;
    if (blocks > 8192u)
      blocks = 8192u;
    hipLaunchKernelGGL(clear_kernel_x4, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (u32x4*)(data + head),
                       body4);
  }
  if (tail)
    hipLaunchKernelGGL(clear_kernel_x1, dim3(1), dim3(256), 0, ctx->stream, data + head + body4 * 4u, tail);
  return check(hipGetLastError(), "clear_kernel launch");
}
}  // namespace yakhip
