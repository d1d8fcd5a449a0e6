// Per-opcode VALU issue cost on gfx950: 8 independent chains per lane, 4 waves per SIMD (16 waves/CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAINS 8
#define DEFK(NAME, ASMSTR) \
__global__ void NAME(unsigned *out, int iters, unsigned seed) { \
  unsigned a[CHAINS]; for (int c=0;c<CHAINS;c++) a[c]=seed+c+threadIdx.x; \
  unsigned z = seed*3+threadIdx.x; \
  for (int i=0;i<iters;i++) { \
    _Pragma("unroll") for (int r=0;r<4;r++) \
    _Pragma("unroll") for (int c=0;c<CHAINS;c++) { unsigned y; asm volatile(ASMSTR : "=v"(y) : "v"(z), "v"(a[c])); a[c]=y; } \
  } \
  unsigned s=0; for (int c=0;c<CHAINS;c++) s+=a[c]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
DEFK(k_add,      "v_add_u32 %0, %1, %2")
DEFK(k_sub,      "v_sub_u32 %0, %1, %2")
DEFK(k_lshl_add, "v_lshl_add_u32 %0, %1, 1, %2")
DEFK(k_mad24,    "v_mad_i32_i24 %0, %1, -2, %2")
DEFK(k_add3,     "v_add3_u32 %0, %1, %2, %2")
DEFK(k_fma,      "v_fma_f32 %0, %1, %2, %2")
DEFK(k_mul24,    "v_mul_i32_i24 %0, %1, %2")
DEFK(k_perm,     "v_perm_b32 %0, %1, %2, %2")
DEFK(k_xor,      "v_xor_b32 %0, %1, %2")
DEFK(k_pkadd16,  "v_pk_add_u16 %0, %1, %2")
DEFK(k_lshl,     "v_lshlrev_b32 %0, 1, %2")
DEFK(k_addlshl,  "v_add_lshl_u32 %0, %1, %2, 1")
typedef void (*kern_t)(unsigned*, int, unsigned);
int main(){
  unsigned *d; (void)hipMalloc(&d, 1<<26);
  hipEvent_t e0,e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters=8000;
  struct { const char *n; kern_t k; } ks[] = { {"v_add_u32",k_add},{"v_sub_u32",k_sub},{"v_lshl_add_u32",k_lshl_add},{"v_mad_i32_i24",k_mad24},
    {"v_add3_u32",k_add3},{"v_fma_f32",k_fma},{"v_mul_i32_i24",k_mul24},{"v_perm_b32",k_perm},{"v_xor_b32",k_xor},{"v_pk_add_u16",k_pkadd16},
    {"v_lshlrev_b32",k_lshl},{"v_add_lshl_u32",k_addlshl} };
  for (int wps : {2, 4}) {
    for (auto &e : ks) {
      int grid=256*wps;  // blocks of 256 threads = 4 waves = 1 per SIMD each
      hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, d, 200, 1u); (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0); hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, d, iters, 1u); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms,e0,e1);
      double n = (double)iters*4*CHAINS*wps;
      printf("waves/SIMD %d  %-16s %.3f ns per wave-instr per SIMD\n", wps, e.n, ms*1e6/n);
    }
  }
  return 0;
}
