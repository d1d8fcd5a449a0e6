// Dependent-issue latency of VALU ops on gfx950: CH independent chains per lane, 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH, int KIND>
__global__ void k(unsigned *out, int iters, unsigned seed) {
  unsigned a[CH]; for (int c = 0; c < CH; c++) a[c] = seed + c + threadIdx.x;
  unsigned z = seed * 3 + threadIdx.x;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 32 / CH; r++)
#pragma unroll
      for (int c = 0; c < CH; c++) {
        unsigned y;
        if (KIND == 0) asm volatile("v_add_u32 %0, %1, %2" : "=v"(y) : "v"(z), "v"(a[c]));
        else if (KIND == 1) asm volatile("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(y) : "v"(z), "v"(a[c]));
        else { unsigned t; asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "v"(z), "v"(a[c]));      /* butterfly: add then lshl_add on its result */
               asm volatile("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(y) : "v"(z), "v"(t)); }
        a[c] = y;
      }
  }
  unsigned s = 0; for (int c = 0; c < CH; c++) s += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
typedef void (*kern_t)(unsigned *, int, unsigned);
template <int KIND> void run(const char *name, unsigned *d) {
  kern_t ks[] = { k<1, KIND>, k<2, KIND>, k<4, KIND>, k<8, KIND>, k<16, KIND> };
  int chs[] = { 1, 2, 4, 8, 16 };
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int wps : { 1, 2, 4 })
    for (int q = 0; q < 5; q++) {
      const int iters = 4000, grid = 256 * wps;
      hipLaunchKernelGGL(ks[q], dim3(grid), dim3(256), 0, 0, d, 100, 1u); (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0); hipLaunchKernelGGL(ks[q], dim3(grid), dim3(256), 0, 0, d, iters, 1u); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      double n = (double)iters * 32 * wps * (KIND == 2 ? 2 : 1);
      printf("%-10s waves/SIMD %d chains %2d: %.3f ns per wave-instr per SIMD\n", name, wps, chs[q], ms * 1e6 / n);
    }
}
int main() {
  unsigned *d; (void)hipMalloc(&d, 1 << 26);
  run<0>("add", d); run<1>("lshl_add", d); run<2>("butterfly", d);
  return 0;
}
