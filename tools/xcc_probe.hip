// Which XCC does workgroup i of a launch run on?  (HW_REG_XCC_ID; the ICP's one-XCD launch, icp.hip, counts on
// workgroup ids going round-robin over the XCDs.)   hipcc --offload-arch=gfx950 -O2 -o xcc_probe xcc_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out)
{
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0)
    out[blockIdx.x] = x;
}
int main()
{
  for (int threads : { 64, 256, 768, 1024 })
    for (int grid : { 64, 200, 1024 })
    {
      unsigned* d = nullptr;
      if (hipMalloc(&d, grid * 4) != hipSuccess)
        return 1;
      for (int rep = 0; rep < 3; ++rep)
      {
        hipLaunchKernelGGL(probe, dim3(grid), dim3(threads), 0, 0, d);
        std::vector<unsigned> h(grid);
        if (hipMemcpy(h.data(), d, grid * 4, hipMemcpyDeviceToHost) != hipSuccess)
          return 1;
        int as_expected = 0;
        for (int i = 0; i < grid; ++i)
          as_expected += (h[i] & 15u) == (unsigned)((i + (int)(h[0] & 15u)) & 7);
        printf("threads %4d grid %4d rep %d: raw[0..15] =", threads, grid, rep);
        for (int i = 0; i < 16; ++i)
          printf(" %x", h[i]);
        printf("  | id %% 8 offset by wg 0's: %d of %d\n", as_expected, grid);
      }
      (void)hipFree(d);
    }
  return 0;
}
