// VALU issue-rate microbenchmark (gfx950): CH independent chains of v_lshl_add_u32 / v_sub_u32 per lane,
// run at 1, 2, 4, 8 waves per SIMD.  Answers: how many cycles does one wave64 integer VALU op cost a SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
template<int CH>
__global__ void k(unsigned *out, int iters, unsigned seed) {
  unsigned a[CH];
  for (int c=0;c<CH;c++) a[c]=seed+c+threadIdx.x;
  unsigned z = seed*3+threadIdx.x;
  for (int i=0;i<iters;i++) {
#pragma unroll
    for (int c=0;c<CH;c++) { unsigned y; asm volatile("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(y) : "v"(z), "v"(a[c])); a[c]=y; }
#pragma unroll
    for (int c=0;c<CH;c++) { unsigned y; asm volatile("v_sub_u32 %0, %1, %2" : "=v"(y) : "v"(a[c]), "v"(z)); a[c]=y; }
  }
  unsigned s=0; for (int c=0;c<CH;c++) s+=a[c];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
int main(){
  unsigned *d; (void)hipMalloc(&d, 1<<26);
  hipEvent_t e0,e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters=20000, CH=8;
  for (int wpc : {4,8,16,32}) {           // waves per CU
    int threads=256; int blocks_per_cu = wpc/4; int grid=256*blocks_per_cu;
    hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(threads), 0, 0, d, 10, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(threads), 0, 0, d, iters, 1u); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms,e0,e1);
    double instr_per_simd = (double)iters*2*CH * (wpc/4.0);   // wave-instr per SIMD
    printf("waves/CU %2d (per SIMD %d): %.3f ms, %.3f ns per VALU wave-instr per SIMD (= %.2f cyc at 2.4 GHz)\n", wpc, wpc/4, ms, ms*1e6/instr_per_simd, ms*1e6/instr_per_simd*2.4);
  }
  return 0;
}
