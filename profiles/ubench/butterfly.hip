// How fast does the real pass_body (3 stages over a 16-element body, registers only) run?  Uses the product's
// own device code (acm_kernels.hip) in a loop without LDS / HBM traffic.  cycles per butterfly per SIMD.
#include "../../libacm_amd/csrc/acm_kernels.hip"
#include <cstdio>
template <int L, int K0, int G>
__global__ void bf(unsigned *out, int iters, unsigned seed)
{
	uint32_t v[2 << G], h[G][1 << G];
	for (int u = 0; u < (2 << G); u++) v[u] = seed * (u + 3) + threadIdx.x;
	for (int t = 0; t < G; t++) for (int x = 0; x < (1 << G); x++) h[t][x] = seed + x + t;
	for (int i = 0; i < iters; i++) {
		pass_body<L, K0, G>(v, h, 0u, 0u);
		// feed outputs back so that nothing is hoisted or removed
	}
	unsigned s = 0;
	for (int u = 0; u < (2 << G); u++) s += v[u];
	for (int t = 0; t < G; t++) for (int x = 0; x < (1 << G); x++) s += h[t][x];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int L, int K0, int G> void run(const char *name, unsigned *d)
{
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const int iters = 4000;
	for (int wps : {1, 2, 4, 8}) {
		int grid = 256 * wps;
		hipLaunchKernelGGL((bf<L, K0, G>), dim3(grid), dim3(256), 0, 0, d, 100, 1u); (void)hipDeviceSynchronize();
		(void)hipEventRecord(e0);
		hipLaunchKernelGGL((bf<L, K0, G>), dim3(grid), dim3(256), 0, 0, d, iters, 1u);
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1);
		double bfly = (double)iters * (2 << G) * G * wps;      // butterflies (wave-level) per SIMD
		printf("%-22s waves/SIMD %d: %.3f ms  -> %.2f ns per butterfly per SIMD (2 VALU ops each)\n", name, wps, ms, ms * 1e6 / bfly);
	}
}
int main()
{
	unsigned *d; (void)hipMalloc(&d, 1 << 26);
	run<7, 0, 3>("L7 stages 0-2 (mad24)", d);
	run<7, 3, 2>("L7 stages 3-4", d);
	run<11, 0, 3>("L11 stages 0-2 (exact)", d);
	return 0;
}
