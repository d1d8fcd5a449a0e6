// Practical HBM ceiling of this box: device-to-device copy with 16 B/lane accesses (read + write bytes / time),
// grid-stride persistent kernel vs one-shot grid, plus hipMemcpyDtoD for comparison.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) copy_persist(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) copy_unroll4(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
  size_t i = (blockIdx.x * 256ull + threadIdx.x);
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i + 3 * stride < n; i += 4 * stride) {
    uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n; i += stride) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) read_only(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
  uint4 acc = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { uint4 v = src[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) dst[0] = acc;
}
__global__ void __launch_bounds__(256) write_only(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
  const uint4 v = {1, 2, 3, 4};
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = v;
}
/* the tile kernel's shape: 1024 workgroups, each streaming through its own contiguous region in 16 KB steps */
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <int NT_LOAD, int NT_STORE>
__global__ void __launch_bounds__(256) copy_runs(const uint4 *__restrict__ src_, uint4 *__restrict__ dst_, size_t n) {
  const size_t per = n / gridDim.x;
  const v4u *s = reinterpret_cast<const v4u *>(src_) + blockIdx.x * per; v4u *d = reinterpret_cast<v4u *>(dst_) + blockIdx.x * per;
  for (size_t i = threadIdx.x; i + 768 < per; i += 1024) {
    v4u v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = NT_LOAD ? __builtin_nontemporal_load(&s[i + 256 * k]) : s[i + 256 * k];
#pragma unroll
    for (int k = 0; k < 4; k++) { if (NT_STORE) __builtin_nontemporal_store(v[k], &d[i + 256 * k]); else d[i + 256 * k] = v[k]; }
  }
}
/* (round 6: what /opt/skills/guides/MI355X_MICROARCH.md quotes 6.29 TB/s for is "a float4 copy" - the plain shapes, for the record) */
__global__ void __launch_bounds__(256) copy_oneshot(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
  const size_t i = blockIdx.x * 256ull + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) copy_oneshot4(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
  const size_t i = blockIdx.x * 1024ull + threadIdx.x;
  if (i + 768 < n) {
    const uint4 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
    dst[i] = a; dst[i + 256] = b; dst[i + 512] = c; dst[i + 768] = d;
  }
}
/* one workgroup of 1024 threads per CU, sixteen 16-byte loads in flight per thread */
__global__ void __launch_bounds__(1024) copy_deep(const uint4 *__restrict__ src_, uint4 *__restrict__ dst_, size_t n) {
  const v4u *s = reinterpret_cast<const v4u *>(src_); v4u *d = reinterpret_cast<v4u *>(dst_);
  const size_t stride = (size_t)gridDim.x * 1024;
  for (size_t i = blockIdx.x * 1024ull + threadIdx.x; i + 15 * stride < n; i += 16 * stride) {
    v4u v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = __builtin_nontemporal_load(&s[i + k * stride]);
#pragma unroll
    for (int k = 0; k < 16; k++) __builtin_nontemporal_store(v[k], &d[i + k * stride]);
  }
}
#ifdef COPY_BW_LIB
/* bench.py's "practical HBM ceiling of this box": the best of the copy kernels above on two fresh buffers of `bytes` each
 * (read + written bytes / time); built into profiles/ubench/libcopybw.so by libacm_amd/_build.py, called through ctypes */
extern "C" double acm_copy_ceiling_gbs(size_t bytes, char *best_name, size_t best_cap) {
  const size_t n = bytes / 16;
  uint4 *a = nullptr, *b = nullptr;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { (void)hipFree(a); return -1.0; }
  (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 2, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  double best = 0;
  auto timeit = [&](const char *name, auto launch) {
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); for (int r = 0; r < 5; r++) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double gbs = 5 * 2.0 * bytes / (ms * 1e-3) / 1e9;
    if (getenv("ACM_COPY_BW_VERBOSE")) fprintf(stderr, "copy_bw: %-66s %7.1f GB/s\n", name, gbs);
    if (gbs > best) { best = gbs; if (best_name && best_cap) snprintf(best_name, best_cap, "%s", name); }
  };
  for (int rep = 0; rep < 2; rep++) {
    timeit("copy unroll4, 4096 WGs", [&]() { hipLaunchKernelGGL(copy_unroll4, dim3(4096), dim3(256), 0, 0, a, b, n); });
    timeit("contiguous runs, 1024 WGs", [&]() { hipLaunchKernelGGL((copy_runs<0, 0>), dim3(1024), dim3(256), 0, 0, a, b, n); });
    timeit("contiguous runs, 1024 WGs, nt stores", [&]() { hipLaunchKernelGGL((copy_runs<0, 1>), dim3(1024), dim3(256), 0, 0, a, b, n); });
    timeit("contiguous runs, 1024 WGs, nt both", [&]() { hipLaunchKernelGGL((copy_runs<1, 1>), dim3(1024), dim3(256), 0, 0, a, b, n); });
    timeit("contiguous runs, 2048 WGs, nt both", [&]() { hipLaunchKernelGGL((copy_runs<1, 1>), dim3(2048), dim3(256), 0, 0, a, b, n); });
    timeit("one 16-byte element per thread, one-shot grid", [&]() { hipLaunchKernelGGL(copy_oneshot, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, b, n); });
    timeit("four elements per thread, one-shot grid", [&]() { hipLaunchKernelGGL(copy_oneshot4, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, a, b, n); });
    timeit("one 1024-thread workgroup per CU, sixteen loads in flight, nt", [&]() { hipLaunchKernelGGL(copy_deep, dim3(256), dim3(1024), 0, 0, a, b, n); });
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(a); (void)hipFree(b);
  return best;
}
/* the same box's one-way rates: out[0] = read-only GB/s (the loaded bytes are folded into a value nobody stores), out[1] = write-only GB/s */
extern "C" int acm_one_way_gbs(size_t bytes, double *out) {
  const size_t n = bytes / 16;
  uint4 *a = nullptr, *b = nullptr;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { (void)hipFree(a); return -1; }
  (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 2, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int which = 0; which < 2; which++) {
    double best = 0;
    for (int rep = 0; rep < 3; rep++) {
      (void)hipEventRecord(e0);
      for (int r = 0; r < 5; r++) { if (which == 0) hipLaunchKernelGGL(read_only, dim3(2048), dim3(256), 0, 0, a, b, n); else hipLaunchKernelGGL(write_only, dim3(2048), dim3(256), 0, 0, a, b, n); }
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      const double gbs = 5.0 * bytes / (ms * 1e-3) / 1e9;
      if (gbs > best) best = gbs;
    }
    out[which] = best;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(a); (void)hipFree(b);
  return 0;
}
/* bench.py poisons the whole PCM buffer before it times another staged form: a launch that skipped tiles must not find the
 * previous form's (correct) samples there */
/* the best of those kernels between two buffers of the caller's (profiles/placement_k3_realloc.py: is a slow allocation slow for a copy too?) */
extern "C" double acm_copy_between_gbs(void *dst, const void *src, size_t bytes) {
  const size_t n = bytes / 16;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  double best = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL((copy_runs<1, 1>), dim3(2048), dim3(256), 0, 0, (const uint4 *)src, (uint4 *)dst, n);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((copy_runs<1, 1>), dim3(2048), dim3(256), 0, 0, (const uint4 *)src, (uint4 *)dst, n);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double gbs = 5 * 2.0 * bytes / (ms * 1e-3) / 1e9;
    if (gbs > best) best = gbs;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return best;
}
/* device memory through the virtual-memory API, in physical chunks of `chunk` bytes (0: one chunk) mapped back to back: does the way the
 * pages are obtained decide the +-5 % of profiles/r5_placement.txt?  (never freed: a probe) */
extern "C" void *acm_vmm_alloc(size_t bytes, size_t chunk, size_t *granularity) {
  int dev = 0; (void)hipGetDevice(&dev);
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran) return nullptr;
  if (granularity) *granularity = gran;
  if (!chunk) chunk = bytes;
  chunk = (chunk + gran - 1) / gran * gran;
  const size_t total = (bytes + chunk - 1) / chunk * chunk;
  void *base = nullptr;
  if (hipMemAddressReserve(&base, total, gran, nullptr, 0) != hipSuccess) return nullptr;
  for (size_t at = 0; at < total; at += chunk) {
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) return nullptr;
    if (hipMemMap((char *)base + at, chunk, 0, h, 0) != hipSuccess) return nullptr;
  }
  hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  if (hipMemSetAccess(base, total, &acc, 1) != hipSuccess) return nullptr;
  return base;
}
extern "C" int acm_poison(void *p, size_t bytes, int value) {
  if (hipMemset(p, value, bytes) != hipSuccess) return -1;
  return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}
#else
#include <cstdlib>
#include <cstring>
#include <chrono>
int main(int argc, char **argv) {
  const size_t bytes = 4ull << 30, n = bytes / 16;
  if (argc >= 4 && !strcmp(argv[1], "loop")) {
    /* copy_bw.bin loop <pattern> <seconds>: one pattern back to back (power / clock readings beside it: profiles/power_probe.sh) */
    uint4 *a, *b; (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 2, bytes);
    const int pat = atoi(argv[2]); const double secs = atof(argv[3]);
    const auto t0 = std::chrono::steady_clock::now(); size_t reps = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
      for (int k = 0; k < 20; k++, reps++) {
        if (pat == 0) hipLaunchKernelGGL((copy_runs<1, 1>), dim3(1024), dim3(256), 0, 0, a, b, n);
        else if (pat == 1) hipLaunchKernelGGL(copy_unroll4, dim3(4096), dim3(256), 0, 0, a, b, n);
        else if (pat == 2) hipLaunchKernelGGL(copy_persist, dim3(16384), dim3(256), 0, 0, a, b, n);
        else if (pat == 3) hipLaunchKernelGGL(read_only, dim3(2048), dim3(256), 0, 0, a, b, n);
        else hipLaunchKernelGGL(write_only, dim3(2048), dim3(256), 0, 0, a, b, n);
      }
      (void)hipDeviceSynchronize();
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("pattern %d: %.1f GB/s over %.1f s\n", pat, reps * (pat >= 3 ? 1.0 : 2.0) * bytes / dt / 1e9, dt);
    return 0;
  }
  uint4 *a, *b; (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 2, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto timeit = [&](const char *name, auto launch, double bytes_moved) {
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); for (int r = 0; r < 5; r++) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %8.1f GB/s\n", name, 5 * bytes_moved / (ms * 1e-3) / 1e9);
  };
  timeit("hipMemcpyDtoD (r+w)", [&]() { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, 2.0 * bytes);
  for (int g : { 512, 1024, 2048, 4096, 16384 }) {
    char nm[64]; snprintf(nm, sizeof nm, "copy grid-stride, %d WGs (r+w)", g);
    timeit(nm, [&]() { hipLaunchKernelGGL(copy_persist, dim3(g), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
    snprintf(nm, sizeof nm, "copy unroll4, %d WGs (r+w)", g);
    timeit(nm, [&]() { hipLaunchKernelGGL(copy_unroll4, dim3(g), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
  }
  for (int rep = 0; rep < 2; rep++) {
  timeit("contiguous runs, 1024 WGs (r+w)", [&]() { hipLaunchKernelGGL((copy_runs<0, 0>), dim3(1024), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
  timeit("contiguous runs, nt loads", [&]() { hipLaunchKernelGGL((copy_runs<1, 0>), dim3(1024), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
  timeit("contiguous runs, nt stores", [&]() { hipLaunchKernelGGL((copy_runs<0, 1>), dim3(1024), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
  timeit("contiguous runs, nt both", [&]() { hipLaunchKernelGGL((copy_runs<1, 1>), dim3(1024), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
  timeit("contiguous runs, 2048 WGs nt both", [&]() { hipLaunchKernelGGL((copy_runs<1, 1>), dim3(2048), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
  }
  timeit("read only, 2048 WGs", [&]() { hipLaunchKernelGGL(read_only, dim3(2048), dim3(256), 0, 0, a, b, n); }, 1.0 * bytes);
  timeit("write only, 2048 WGs", [&]() { hipLaunchKernelGGL(write_only, dim3(2048), dim3(256), 0, 0, a, b, n); }, 1.0 * bytes);
  return 0;
}
#endif
