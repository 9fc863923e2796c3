// Exhaustive / brute-force checks of the short division sequences against IEEE '/' on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ float refined_rcp(float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r, 1.f);
  return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float div3(float n, float d, float r) {
  float q = n * r;
  float e = __builtin_fmaf(-d, q, n);
  return __builtin_fmaf(e, r, q);
}
__device__ __forceinline__ float div5(float n, float d, float r) {
  float q = n * r;
  float e = __builtin_fmaf(-d, q, n);
  q = __builtin_fmaf(e, r, q);
  e = __builtin_fmaf(-d, q, n);
  return __builtin_fmaf(e, r, q);
}
__device__ __forceinline__ uint32_t rng(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 32); }

// all 2^23 mantissas x a few exponents: refined_rcp(d) == 1.0f / d ?
__global__ void rcp_all(unsigned long long* bad, uint32_t* example) {
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;  // 0 .. 2^23-1
  const int exps[] = {107, 120, 126, 127, 128, 130, 140, 147};
  for (int k = 0; k < 8; ++k) {
    const float d = __uint_as_float(((uint32_t)exps[k] << 23) | m);
    const float a = refined_rcp(d), b = 1.0f / d;
    if (__float_as_uint(a) != __float_as_uint(b)) { atomicAdd(bad, 1ull); *example = __float_as_uint(d); }
  }
}

// random pairs: n in +-[2^-30, 2^8], d in [2^-20, 2^20]
__global__ void div_rand(unsigned long long* bad3, unsigned long long* bad5, uint32_t* ex, int iters, uint64_t seed) {
  uint64_t s = seed + (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
  unsigned long long b3 = 0, b5 = 0;
  for (int i = 0; i < iters; ++i) {
    const uint32_t a = rng(s), b = rng(s);
    const float n = __uint_as_float((a & 0x807fffffu) | ((97u + (a >> 23) % 39u) << 23));
    const float d = __uint_as_float((b & 0x007fffffu) | ((107u + (b >> 23) % 41u) << 23));
    const float r = refined_rcp(d);
    const float q = n / d;
    const float q3 = div3(n, d, r), q5 = div5(n, d, r);
    if (__float_as_uint(q3) != __float_as_uint(q)) { ++b3; ex[0] = __float_as_uint(n); ex[1] = __float_as_uint(d); }
    if (__float_as_uint(q5) != __float_as_uint(q)) { ++b5; ex[2] = __float_as_uint(n); ex[3] = __float_as_uint(d); }
  }
  if (b3) atomicAdd(bad3, b3);
  if (b5) atomicAdd(bad5, b5);
}

// integer denominators 1..32768 (the running-average divide), numerators random in +-[2^-26, 2^16]
__global__ void div_int(unsigned long long* bad3, unsigned long long* bad5, unsigned long long* badr, int iters, uint64_t seed) {
  uint64_t s = seed + (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
  unsigned long long b3 = 0, b5 = 0, br = 0;
  for (int i = 0; i < iters; ++i) {
    const uint32_t a = rng(s), b = rng(s);
    const float n = __uint_as_float((a & 0x807fffffu) | ((101u + (a >> 23) % 43u) << 23));
    const float d = (float)(1 + (b % 32768u));
    const float r = refined_rcp(d);
    if (__float_as_uint(r) != __float_as_uint(1.0f / d)) ++br;
    const float q = n / d;
    if (__float_as_uint(div3(n, d, r)) != __float_as_uint(q)) ++b3;
    if (__float_as_uint(div5(n, d, r)) != __float_as_uint(q)) ++b5;
  }
  if (b3) atomicAdd(bad3, b3);
  if (b5) atomicAdd(bad5, b5);
  if (br) atomicAdd(badr, br);
}

__global__ void flr_test(const float* in, int* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { int r; asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(in[i])); out[i] = r; }
}

int main() {
  unsigned long long* c; uint32_t* ex;
  CHECK(hipMalloc(&c, 64)); CHECK(hipMalloc(&ex, 64));
  CHECK(hipMemset(c, 0, 64)); CHECK(hipMemset(ex, 0, 64));
  hipLaunchKernelGGL(rcp_all, dim3((1 << 23) / 256), dim3(256), 0, 0, c, ex);
  hipLaunchKernelGGL(div_rand, dim3(4096), dim3(256), 0, 0, c + 1, c + 2, ex + 1, 4096, 12345ull);
  hipLaunchKernelGGL(div_int, dim3(4096), dim3(256), 0, 0, c + 3, c + 4, c + 5, 2048, 777ull);
  CHECK(hipDeviceSynchronize());
  unsigned long long h[8]; uint32_t hex[8];
  CHECK(hipMemcpy(h, c, 64, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hex, ex, 32, hipMemcpyDeviceToHost));
  printf("refined_rcp != 1/d: %llu of %llu (example d bits %08x)\n", h[0], 8ull << 23, hex[0]);
  printf("random pairs (%.3g): div3 mismatches %llu, div5 mismatches %llu  ex3 n=%08x d=%08x ex5 n=%08x d=%08x\n",
         4096.0 * 256 * 4096, h[1], h[2], hex[1], hex[2], hex[3], hex[4]);
  printf("integer denominators (%.3g): div3 mismatches %llu, div5 mismatches %llu, rcp mismatches %llu\n",
         4096.0 * 256 * 2048, h[3], h[4], h[5]);
  float hin[8] = {-0.0f, -0.5f, -1.0f, 0.999f, 639.99f, 640.f, -1e30f, 1e30f}; int hout[8];
  float* din; int* dout; CHECK(hipMalloc(&din, 32)); CHECK(hipMalloc(&dout, 32));
  CHECK(hipMemcpy(din, hin, 32, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(flr_test, dim3(1), dim3(64), 0, 0, din, dout, 8);
  CHECK(hipMemcpy(hout, dout, 32, hipMemcpyDeviceToHost));
  for (int i = 0; i < 8; ++i) printf("flr(%g) = %d\n", hin[i], hout[i]);
  return 0;
}
