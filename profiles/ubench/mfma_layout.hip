// Which lane holds what in v_mfma_i32_16x16x32_i8 on gfx950: D = A (16 x 32) * B (32 x 16) + C.
// Feeds one-hot A and B operands and prints where the single 1 of D lands: confirms the operand layout FirstPassM
// (libacm_amd/csrc/acm_kernels.hip) is written for:  A: lane l, byte j = A[l % 16][8 * (l / 16) + j];  B: lane l, byte j =
// B[8 * (l / 16) + j][l % 16];  D: lane l, register i = D[4 * (l / 16) + i][l % 16].
//   hipcc --offload-arch=gfx950 -O2 -o mfma_layout mfma_layout.hip && ./mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void probe(const uint64_t *a, const uint64_t *b, int *d)
{
	const int l = threadIdx.x;
	v4i c = { 0, 0, 0, 0 };
	c = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)a[l], (long)b[l], c, 0, 0, 0);
	for (int i = 0; i < 4; i++)
		d[l * 4 + i] = c[i];
}
int main()
{
	uint64_t *a, *b;
	int *d;
	hipMallocManaged(&a, 64 * 8);
	hipMallocManaged(&b, 64 * 8);
	hipMallocManaged(&d, 256 * 4);
	int bad = 0;
	for (int m = 0; m < 16; m += 5)
		for (int k = 0; k < 32; k += 7)
			for (int n = 0; n < 16; n += 3) {
				for (int l = 0; l < 64; l++)
					a[l] = b[l] = 0;
				a[(k / 8) * 16 + m] = (uint64_t)3 << (8 * (k % 8));          // A[m][k] = 3
				b[(k / 8) * 16 + n] = (uint64_t)(uint8_t)-5 << (8 * (k % 8));   // B[k][n] = -5
				hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, d);
				hipDeviceSynchronize();
				for (int l = 0; l < 64; l++)
					for (int i = 0; i < 4; i++) {
						const int want = (l % 16 == n && 4 * (l / 16) + i == m) ? -15 : 0;
						if (d[l * 4 + i] != want) {
							if (bad < 10)
								printf("A[%d][%d] B[%d][%d]: lane %d reg %d = %d, expected %d\n", m, k, k, n, l, i, d[l * 4 + i], want);
							bad++;
						}
					}
			}
	printf("mfma_i32_16x16x32_i8 operand layout: %s\n", bad ? "NOT as assumed" : "as assumed");
	return bad != 0;
}
