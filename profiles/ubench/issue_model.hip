// Issue model of gfx950 for the instruction mix of the tile kernel (round 2).
//
//   A  VALU throughput per opcode at 1/2/4/8 waves per SIMD (16 independent chains per lane)
//   B  butterfly mixes (add + lshl_add, add + mad24, three adds with DPP operands)
//   C  LDS instruction throughput per CU by width and form
//   D  does LDS traffic of one wave overlap VALU work of another (and of the same wave)?
//
// Every number is cycles of the shader clock (s_memtime) per wave-instruction per SIMD (A, B) or per CU (C),
// taken as the 99th percentile over waves of (end - start) (older waves win arbitration and finish early); waves of a CU start together because a workgroup's LDS
// reservation admits exactly `wps` workgroups of 256 threads per CU.  The clock the chip held is printed too.
//
//   hipcc --offload-arch=gfx950 -O3 -o issue_model issue_model.hip && ./issue_model
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));
struct Stamp { unsigned long long c0, c1, r0, r1; };
extern __shared__ unsigned dyn_lds[];

__device__ __forceinline__ void stamp_begin(Stamp &s)
{
	__syncthreads();
	asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(s.c0), "=s"(s.r0) :: "memory");
}
__device__ __forceinline__ void stamp_end(Stamp &s, Stamp *out)
{
	asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(s.c1), "=s"(s.r1) :: "memory");
	if ((threadIdx.x & 63) == 0)
		out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;
}

#define CH 16
/* one VALU opcode, CH independent chains, 4 x CH instructions per loop trip */
#define DEFV(NAME, ASMSTR) \
__global__ void __launch_bounds__(256) NAME(unsigned *out, Stamp *st, int iters, unsigned seed) { \
	unsigned a[CH]; for (int c = 0; c < CH; c++) a[c] = seed + c * 77 + threadIdx.x; \
	unsigned z = seed * 3 + threadIdx.x; Stamp s; stamp_begin(s); \
	for (int i = 0; i < iters; i++) { \
		_Pragma("unroll") for (int r = 0; r < 4; r++) \
		_Pragma("unroll") for (int c = 0; c < CH; c++) { unsigned y; asm volatile(ASMSTR : "=v"(y) : "v"(z), "v"(a[c])); a[c] = y; } \
	} \
	stamp_end(s, st); unsigned q = 0; for (int c = 0; c < CH; c++) q += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = q; }

DEFV(v_add,       "v_add_u32 %0, %1, %2")
DEFV(v_sub,       "v_sub_u32 %0, %1, %2")
DEFV(v_xor,       "v_xor_b32 %0, %1, %2")
DEFV(v_lshl,      "v_lshlrev_b32 %0, 1, %2")
DEFV(v_mov,       "v_mov_b32 %0, %2")
DEFV(v_lshl_add,  "v_lshl_add_u32 %0, %1, 1, %2")
DEFV(v_add_lshl,  "v_add_lshl_u32 %0, %1, %2, 1")
DEFV(v_add3,      "v_add3_u32 %0, %1, %2, %2")
DEFV(v_mad24,     "v_mad_i32_i24 %0, %1, -2, %2")
DEFV(v_madu24,    "v_mad_u32_u24 %0, %1, 2, %2")
DEFV(v_mul24,     "v_mul_i32_i24 %0, %1, %2")
DEFV(v_mul24s,    "v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
DEFV(v_perm,      "v_perm_b32 %0, %1, %2, %2")
DEFV(v_fma,       "v_fma_f32 %0, %1, %2, %2")
DEFV(v_fmac,      "v_mul_f32 %0, %1, %2")
DEFV(v_pkadd16,   "v_pk_add_u16 %0, %1, %2")
DEFV(v_and_or,    "v_and_or_b32 %0, %1, %2, %2")
DEFV(v_bfe,       "v_bfe_i32 %0, %2, 0, 16")
DEFV(v_ashr,      "v_ashrrev_i32 %0, 16, %2")
DEFV(v_add_dpp1,  "v_add_u32_dpp %0, %2, %1 wave_shr:1 row_mask:0xf bank_mask:0xf")
DEFV(v_add_dppr,  "v_add_u32_dpp %0, %2, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
DEFV(v_add_dppq,  "v_add_u32_dpp %0, %2, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEFV(v_mov_dpp1,  "v_mov_b32_dpp %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf")
DEFV(v_add_sdwa,  "v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
DEFV(v_cndmask,   "v_cndmask_b32 %0, %1, %2, vcc")
DEFV(v_lshrsdwa,  "v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD")

DEFV(v_lshr,      "v_lshrrev_b32 %0, 9, %2")
DEFV(v_and,       "v_and_b32 %0, %1, %2")
DEFV(v_or,        "v_or_b32 %0, %1, %2")
DEFV(v_min,       "v_min_i32 %0, %1, %2")
DEFV(v_lshl_or,   "v_lshl_or_b32 %0, %1, 9, %2")
DEFV(v_add_lit,   "v_add_u32 %0, 0x12345, %2")
DEFV(v_add_inl,   "v_add_u32 %0, 17, %2")
DEFV(v_sub_e64,   "v_sub_u32_e64 %0, %1, %2")
DEFV(v_add_vcc,   "v_add_co_u32 %0, vcc, %1, %2")
DEFV(v_cmp_lt,    "v_cmp_lt_i32 vcc, %1, %2\n\tv_mov_b32 %0, %2")
DEFV(v_xad,       "v_xad_u32 %0, %1, %2, %2")
DEFV(v_sub_clamp, "v_sub_u32 %0, %1, %2 clamp")
DEFV(v_mov_dpp_q, "v_mov_b32_dpp %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEFV(v_alignbit,  "v_alignbit_b32 %0, %1, %2, 16")
DEFV(v_lshl4,     "v_lshlrev_b32 %0, 4, %2")
DEFV(v_add_f32,   "v_add_f32 %0, %1, %2")
DEFV(v_cvt,       "v_cvt_f32_i32 %0, %2")
DEFV(v_pk_mad16,  "v_pk_mad_u16 %0, %1, %2, %2")

/* VALU ops with scalar operands */
#define DEFVS(NAME, ASMSTR) \
__global__ void __launch_bounds__(256) NAME(unsigned *out, Stamp *st, int iters, unsigned seed) { \
	unsigned a[CH]; for (int c = 0; c < CH; c++) a[c] = seed + c * 77 + threadIdx.x; \
	unsigned z = __builtin_amdgcn_readfirstlane(seed * 3); unsigned long long zz = __builtin_amdgcn_readfirstlane(seed) | 0xff00ull; Stamp s; stamp_begin(s); \
	for (int i = 0; i < iters; i++) { \
		_Pragma("unroll") for (int r = 0; r < 4; r++) \
		_Pragma("unroll") for (int c = 0; c < CH; c++) { unsigned y; asm volatile(ASMSTR : "=v"(y) : "s"(z), "v"(a[c]), "s"(zz)); a[c] = y; } \
	} \
	stamp_end(s, st); unsigned q = 0; for (int c = 0; c < CH; c++) q += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = q; }
DEFVS(vs_add,      "v_add_u32 %0, %1, %2")
DEFVS(vs_lshl_add, "v_lshl_add_u32 %0, %2, 1, %1")
DEFVS(vs_mul24,    "v_mul_i32_i24 %0, %1, %2")
DEFVS(vs_cnd,      "v_cndmask_b32_e64 %0, %2, %2, %3")
DEFVS(vs_mov,      "v_mov_b32 %0, %1")

/* two-register ops */
__global__ void __launch_bounds__(256) v_plswap32(unsigned *out, Stamp *st, int iters, unsigned seed) {
	unsigned a[CH]; for (int c = 0; c < CH; c++) a[c] = seed + c * 77 + threadIdx.x;
	Stamp s; stamp_begin(s);
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int r = 0; r < 8; r++)
#pragma unroll
			for (int c = 0; c < CH; c += 2) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[c]), "+v"(a[c + 1]));
	}
	stamp_end(s, st); unsigned q = 0; for (int c = 0; c < CH; c++) q += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = q; }
__global__ void __launch_bounds__(256) v_plswap16(unsigned *out, Stamp *st, int iters, unsigned seed) {
	unsigned a[CH]; for (int c = 0; c < CH; c++) a[c] = seed + c * 77 + threadIdx.x;
	Stamp s; stamp_begin(s);
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int r = 0; r < 8; r++)
#pragma unroll
			for (int c = 0; c < CH; c += 2) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[c]), "+v"(a[c + 1]));
	}
	stamp_end(s, st); unsigned q = 0; for (int c = 0; c < CH; c++) q += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = q; }

/* B: butterfly mixes; 64 wave-instructions per trip in every kernel (instruction count is the divisor) */
/* KIND 0: add + lshl_add (32 butterflies)   1: sub + mad24   2: alternate kinds 0/1   3: three plain adds (21 butterflies + 1)
 * 4: three adds with the remote operands taken through DPP wave_shr:1   5: add(dpp) + lshl_add */
template <int KIND>
__global__ void __launch_bounds__(256) bfly(unsigned *out, Stamp *st, int iters, unsigned seed) {
	unsigned a[CH]; for (int c = 0; c < CH; c++) a[c] = seed + c * 77 + threadIdx.x;
	unsigned z = seed * 3 + threadIdx.x; Stamp s; stamp_begin(s);
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int r = 0; r < ((KIND == 3 || KIND == 4) ? 1 : 2); r++)
#pragma unroll
			for (int c = 0; c < CH; c++) {
				unsigned t, y, p = a[(c + 1) % CH];
				if (KIND == 0 || (KIND == 2 && (c & 1) == 0)) {
					asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "v"(p), "v"(a[c]));
					asm volatile("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(y) : "v"(z), "v"(t));
				} else if (KIND == 1 || KIND == 2) {
					asm volatile("v_sub_u32 %0, %1, %2" : "=v"(t) : "v"(p), "v"(a[c]));
					asm volatile("v_mad_i32_i24 %0, %1, -2, %2" : "=v"(y) : "v"(z), "v"(t));
				} else if (KIND == 3) {
					unsigned t2;
					asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "v"(p), "v"(a[c]));
					asm volatile("v_add_u32 %0, %1, %2" : "=v"(t2) : "v"(z), "v"(t));
					asm volatile("v_add_u32 %0, %1, %2" : "=v"(y) : "v"(z), "v"(t2));
					asm volatile("v_add_u32 %0, %1, %2" : "=v"(y) : "v"(y), "v"(p));
				} else if (KIND == 4) {
					unsigned t2;
					asm volatile("v_add_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(p), "v"(a[c]));
					asm volatile("v_add_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(t2) : "v"(z), "v"(t));
					asm volatile("v_add_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(y) : "v"(z), "v"(t2));
					asm volatile("v_add_u32 %0, %1, %2" : "=v"(y) : "v"(y), "v"(p));
				} else {
					asm volatile("v_add_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(p), "v"(a[c]));
					asm volatile("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(y) : "v"(z), "v"(t));
				}
				a[c] = y;
			}
	}
	stamp_end(s, st); unsigned q = 0; for (int c = 0; c < CH; c++) q += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = q; }

/* same mixes with the ops of the 16 butterflies issued phase by phase (dependent ops 16 apart).
 * KIND 0: add, lshl_add (32 instr)  1: (z1 + a) + (z1 + b): three adds (48 instr)  2: (z1 - a) + (z1 - b)  */
template <int KIND>
__global__ void __launch_bounds__(256) bfly_il(unsigned *out, Stamp *st, int iters, unsigned seed) {
	unsigned a[CH]; for (int c = 0; c < CH; c++) a[c] = seed + c * 77 + threadIdx.x;
	unsigned z = seed * 3 + threadIdx.x; Stamp s; stamp_begin(s);
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int r = 0; r < 2; r++) {
			unsigned t[CH], u[CH];
#pragma unroll
			for (int c = 0; c < CH; c++) {
				if (KIND == 2) asm volatile("v_sub_u32 %0, %1, %2" : "=v"(t[c]) : "v"(z), "v"(a[c]));
				else asm volatile("v_add_u32 %0, %1, %2" : "=v"(t[c]) : "v"(a[(c + 1) % CH]), "v"(a[c]));
			}
			if (KIND != 0) {
#pragma unroll
				for (int c = 0; c < CH; c++) {
					if (KIND == 2) asm volatile("v_sub_u32 %0, %1, %2" : "=v"(u[c]) : "v"(z), "v"(a[(c + 1) % CH]));
					else asm volatile("v_add_u32 %0, %1, %2" : "=v"(u[c]) : "v"(z), "v"(t[c]));
				}
			}
#pragma unroll
			for (int c = 0; c < CH; c++) {
				if (KIND == 0) asm volatile("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(a[c]) : "v"(z), "v"(t[c]));
				else if (KIND == 1) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[c]) : "v"(z), "v"(u[c]));
				else asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[c]) : "v"(t[c]), "v"(u[c]));
			}
		}
	}
	stamp_end(s, st); unsigned q = 0; for (int c = 0; c < CH; c++) q += a[c]; out[blockIdx.x * blockDim.x + threadIdx.x] = q; }

/* C: LDS forms.  MODE 0 read b32, 1 read b64, 2 read b128, 3 write b32, 4 write b64, 5 write b128,
 * 6 read_addtid b32, 7 write_addtid b32, 8 read2_b32 (two strided dwords), 9 write2_b32, 10 read2st64 */
template <int MODE>
__global__ void __launch_bounds__(256) ldsk(unsigned *out, Stamp *st, int iters, unsigned seed) {
	const unsigned tid = threadIdx.x;
	for (unsigned k = tid; k < 4096; k += 256) dyn_lds[k] = k * seed;
	__syncthreads();
	/* conflict-free addresses: lane-contiguous units of the access width, a separate 4 KB window per wave */
	const unsigned w = (MODE == 1 || MODE == 4) ? 8 : (MODE == 2 || MODE == 5) ? 16 : 4;
	unsigned addr = (tid >> 6) * 4096 + (tid & 63) * w;
	unsigned acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
	unsigned d0 = seed + tid, d1 = seed * 3 + tid, d2 = d0 ^ d1, d3 = d0 + d1;
	asm volatile("s_mov_b32 m0, %0" :: "s"(__builtin_amdgcn_readfirstlane((tid >> 6) * 4096u)));
	Stamp s; stamp_begin(s);
	for (int i = 0; i < iters; i++) {
		if (MODE == 0) {
			unsigned v[16];
#pragma unroll
			for (int r = 0; r < 16; r++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[r]) : "v"(addr), "n"(r * 256));
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
			for (int r = 0; r < 16; r++) asm volatile("" :: "v"(v[r]));
		}
		if (MODE == 1) {
			unsigned long long v[16];
#pragma unroll
			for (int r = 0; r < 16; r++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[r]) : "v"(addr), "n"((r & 7) * 512));
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
			for (int r = 0; r < 16; r++) asm volatile("" :: "v"(v[r]));
		}
		if (MODE == 2) {
			v4u v[8];
#pragma unroll
			for (int r = 0; r < 8; r++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[r]) : "v"(addr), "n"((r & 3) * 1024));
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
			for (int r = 0; r < 8; r++) asm volatile("" :: "v"(v[r]));
#pragma unroll
			for (int r = 0; r < 8; r++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[r]) : "v"(addr), "n"((r & 3) * 1024));
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
			for (int r = 0; r < 8; r++) asm volatile("" :: "v"(v[r]));
		}
		if (MODE == 3) {
#pragma unroll
			for (int r = 0; r < 16; r++) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(addr), "v"(d0), "n"(r * 256) : "memory");
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		}
		if (MODE == 4) {
			unsigned long long dd = ((unsigned long long)d1 << 32) | d0;
#pragma unroll
			for (int r = 0; r < 16; r++) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(addr), "v"(dd), "n"((r & 7) * 512) : "memory");
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		}
		if (MODE == 5) {
			v4u dd = { d0, d1, d2, d3 };
#pragma unroll
			for (int r = 0; r < 16; r++) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(dd), "n"((r & 3) * 1024) : "memory");
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		}
		if (MODE == 6) {
			unsigned v[16];
#pragma unroll
			for (int r = 0; r < 16; r++) asm volatile("ds_read_addtid_b32 %0 offset:%1" : "=v"(v[r]) : "n"(r * 256) : "memory");
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
			for (int r = 0; r < 16; r++) asm volatile("" :: "v"(v[r]));
		}
		if (MODE == 7) {
#pragma unroll
			for (int r = 0; r < 16; r++) asm volatile("ds_write_addtid_b32 %0 offset:%1" :: "v"(d0), "n"(r * 256) : "memory");
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		}
		if (MODE == 8) {
			unsigned long long v[16];
#pragma unroll
			for (int r = 0; r < 16; r++) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v[r]) : "v"(addr), "n"(r * 8), "n"(r * 8 + 65) : "memory");
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
			for (int r = 0; r < 16; r++) asm volatile("" :: "v"(v[r]));
		}
		if (MODE == 9) {
#pragma unroll
			for (int r = 0; r < 16; r++) asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" :: "v"(addr), "v"(d0), "v"(d1), "n"(r * 8), "n"(r * 8 + 65) : "memory");
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		}
	}
	stamp_end(s, st);
	out[blockIdx.x * blockDim.x + tid] = acc0 + acc1 + acc2 + acc3 + dyn_lds[tid];
}

/* D: a pass-like loop.  Per trip: NLD ds_reads (b32 or b64), NV VALU ops on the loaded data (2-op butterflies),
 * NLD ds_writes.  PIPE = 1: the reads of trip i+1 are issued before the VALU work of trip i.
 * WIDE = 0: b32 accesses of 16 elements, WIDE = 1: b64 accesses of 16 elements (8 instructions each way). */
template <int NVPER, int WIDE, int PIPE, int DO_LDS, int DO_VALU>
__global__ void __launch_bounds__(256) passk(unsigned *out, Stamp *st, int iters, unsigned seed) {
	const unsigned tid = threadIdx.x;
	for (unsigned k = tid; k < 4096; k += 256) dyn_lds[k] = k * seed;
	__syncthreads();
	unsigned addr = (tid >> 6) * 4096 + (tid & 63) * (WIDE ? 8 : 4);
	unsigned z = seed * 3 + tid;
	unsigned cur[16], nxt[16];
	for (int c = 0; c < 16; c++) cur[c] = nxt[c] = seed + c;
	auto rd = [&](unsigned (&v)[16]) {
		if (!DO_LDS) return;
		if (WIDE) {
#pragma unroll
			for (int r = 0; r < 8; r++) { unsigned long long q; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(q) : "v"(addr), "n"(r * 512) : "memory"); v[2 * r] = (unsigned)q; v[2 * r + 1] = (unsigned)(q >> 32); }
		} else {
#pragma unroll
			for (int r = 0; r < 16; r++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[r]) : "v"(addr), "n"(r * 256) : "memory");
		}
	};
	auto wr = [&](unsigned (&v)[16]) {
		if (!DO_LDS) return;
		if (WIDE) {
#pragma unroll
			for (int r = 0; r < 8; r++) { unsigned long long q = ((unsigned long long)v[2 * r + 1] << 32) | v[2 * r]; asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(addr), "v"(q), "n"(r * 512) : "memory"); }
		} else {
#pragma unroll
			for (int r = 0; r < 16; r++) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(addr), "v"(v[r]), "n"(r * 256) : "memory");
		}
	};
	Stamp s; stamp_begin(s);
	if (PIPE) rd(nxt);
	for (int i = 0; i < iters; i++) {
		if (PIPE) {
			if (DO_LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
			for (int c = 0; c < 16; c++) cur[c] = nxt[c];
			rd(nxt);
		} else {
			rd(cur);
			if (DO_LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		}
		if (DO_VALU) {
#pragma unroll
			for (int g = 0; g < NVPER; g++)
#pragma unroll
				for (int c = 0; c < 16; c++) {
					unsigned t, y;
					asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "v"(cur[(c + 1) & 15]), "v"(cur[c]));
					asm volatile("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(y) : "v"(z), "v"(t));
					cur[c] = y;
				}
		}
		wr(cur);
	}
	stamp_end(s, st);
	unsigned q = 0; for (int c = 0; c < 16; c++) q += cur[c] + nxt[c];
	out[blockIdx.x * blockDim.x + tid] = q + dyn_lds[tid];
}

/* E: co-issue.  Per trip and wave: 64 VALU (32 x add + lshl_add on 16 independent chains), NLDS ds_read_b32 + NLDS ds_write_b32
 * sprinkled evenly between them (no waits inside the trip; the loaded values are consumed one trip later), NSALU s_add_u32. */
template <int NLDS, int NSALU, int SIMPLE>
__global__ void __launch_bounds__(256) mixk(unsigned *out, Stamp *st, int iters, unsigned seed) {
	const unsigned tid = threadIdx.x;
	for (unsigned k = tid; k < 4096; k += 256) dyn_lds[k] = k * seed;
	__syncthreads();
	unsigned addr = (tid >> 6) * 4096 + (tid & 63) * 4;
	unsigned a[CH]; for (int c = 0; c < CH; c++) a[c] = seed + c * 77 + tid;
	unsigned z = seed * 3 + tid, sacc = seed;
	unsigned ld[8]; for (int c = 0; c < 8; c++) ld[c] = c;
	Stamp s; stamp_begin(s);
	for (int i = 0; i < iters; i++) {
		if (NLDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		unsigned fold = 0;
#pragma unroll
		for (int c = 0; c < NLDS; c++) fold ^= ld[c];
		z ^= fold & 1;
#pragma unroll
		for (int r = 0; r < 2; r++)
#pragma unroll
			for (int c = 0; c < CH; c++) {
				unsigned t;
				const int slot = r * CH + c;                 /* 0..31 */
				if (SIMPLE) {
					asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "v"(a[(c + 5) % CH]), "v"(a[c]));
					asm volatile("v_sub_u32 %0, %1, %2" : "=v"(a[c]) : "v"(z), "v"(t));
				} else {
					asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "v"(a[(c + 5) % CH]), "v"(a[c]));
					asm volatile("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(a[c]) : "v"(z), "v"(t));
				}
				if (NLDS && slot % (32 / (2 * NLDS)) == 0) {
					const int k = slot / (32 / (2 * NLDS));      /* 0 .. 2*NLDS-1 */
					if (k < NLDS) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(ld[k]) : "v"(addr), "n"(k * 256) : "memory");
					else asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(addr), "v"(a[c]), "n"((k - NLDS) * 256 + 2048) : "memory");
				}
				if (NSALU && slot % (32 / NSALU) == 0)
					asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc) :: "scc");
			}
	}
	stamp_end(s, st);
	unsigned q = sacc; for (int c = 0; c < CH; c++) q += a[c]; for (int c = 0; c < 8; c++) q += ld[c];
	out[blockIdx.x * blockDim.x + tid] = q + dyn_lds[tid];
}

/* F: what one extra instruction costs a SIMD whose VALU is saturated.  Per trip and wave: 64 VALU (add + lshl_add, 16
 * independent chains) and 8 copies of instruction X spread evenly; results of X are never consumed inside the loop and
 * nothing waits inside it (counters are drained after the loop), so the difference to X = none is pure issue cost. */
template <int X>
__global__ void __launch_bounds__(256) costk(unsigned *out, Stamp *st, int iters, unsigned seed) {
	const unsigned tid = threadIdx.x;
	for (unsigned k = tid; k < 4096; k += 256) dyn_lds[k] = k * seed;
	__syncthreads();
	unsigned addr = (tid >> 6) * 4096 + (tid & 63) * (X == 2 || X == 6 ? 16 : (X == 1 || X == 5 ? 8 : 4));
	asm volatile("s_mov_b32 m0, %0" :: "s"(__builtin_amdgcn_readfirstlane((tid >> 6) * 4096u)));
	unsigned a[CH]; for (int c = 0; c < CH; c++) a[c] = seed + c * 77 + tid;
	unsigned z = seed * 3 + tid, sacc = seed;
	unsigned d0 = seed + tid, d1 = d0 * 3;
	unsigned long long dd = ((unsigned long long)d1 << 32) | d0;
	v4u d4 = { d0, d1, d0 ^ d1, d0 + d1 };
	const unsigned *gp = out + (blockIdx.x * 256 + tid) % 4096;          /* L2-resident lines */
	const v4u *gp4 = reinterpret_cast<const v4u *>(out) + (blockIdx.x * 256 + tid) % 4096;
	unsigned *sp = out + 65536 + (blockIdx.x * 256 + tid) * 4;
	unsigned r32; unsigned long long r64; v4u r128;
	Stamp s; stamp_begin(s);
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int r = 0; r < 2; r++)
#pragma unroll
			for (int c = 0; c < CH; c++) {
				unsigned t;
				const int slot = r * CH + c;
				asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "v"(a[(c + 5) % CH]), "v"(a[c]));
				asm volatile("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(a[c]) : "v"(z), "v"(t));
				if (slot % 4 == 1) {
					const int k = slot / 4;
					if (X == 0) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r32) : "v"(addr), "n"(k * 256) : "memory");
					if (X == 1) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r64) : "v"(addr), "n"((k & 7) * 512) : "memory");
					if (X == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r128) : "v"(addr), "n"((k & 3) * 1024) : "memory");
					if (X == 3) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(r64) : "v"(addr), "n"(k * 8), "n"(k * 8 + 65) : "memory");
					if (X == 4) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(addr), "v"(d0), "n"(k * 256) : "memory");
					if (X == 5) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(addr), "v"(dd), "n"((k & 7) * 512) : "memory");
					if (X == 6) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(d4), "n"((k & 3) * 1024) : "memory");
					if (X == 7) asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" :: "v"(addr), "v"(d0), "v"(d1), "n"(k * 8), "n"(k * 8 + 65) : "memory");
					if (X == 8) asm volatile("ds_write_addtid_b32 %0 offset:%1" :: "v"(d0), "n"(k * 256) : "memory");
					if (X == 9) asm volatile("ds_read_addtid_b32 %0 offset:%1" : "=v"(r32) : "n"(k * 256) : "memory");
					if (X == 10) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc) :: "scc");
					if (X == 11) asm volatile("s_nop 0");
					if (X == 12) asm volatile("s_waitcnt vmcnt(63)");
					if (X == 13) asm volatile("global_load_dword %0, %1, off" : "=v"(r32) : "v"(gp) : "memory");
					if (X == 14) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r128) : "v"(gp4) : "memory");
					if (X == 15) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(sp), "v"(d4) : "memory");
					if (X == 16) asm volatile("global_load_ushort %0, %1, off" : "=v"(r32) : "v"(gp) : "memory");
					if (X == 17) asm volatile("v_mov_b32 %0, %1" : "=v"(r32) : "v"(d0));
				}
			}
	}
	asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
	stamp_end(s, st);
	unsigned q = sacc; for (int c = 0; c < CH; c++) q += a[c];
	if (X == 99) q += r32 + (unsigned)r64 + r128.x;
	out[blockIdx.x * blockDim.x + tid] = q + dyn_lds[tid];
}

typedef void (*kern_t)(unsigned *, Stamp *, int, unsigned);

struct Result { double cyc_med, cyc_max, ghz, ms; };

static int run(kern_t k, int wps, int iters, unsigned *d_out, Stamp *d_st, Result &res)
{
	const int cus = 256, grid = cus * wps;
	/* LDS reservation: exactly wps workgroups fit a CU (160 KiB) */
	size_t lds = (160 * 1024 / wps) & ~1023u;       /* >= 20 KB; the kernels use 16 KB */
	if (lds > 64 * 1024)
		CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, d_out, d_st, 50, 1u);
	CK(hipDeviceSynchronize());
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	CK(hipEventRecord(e0));
	hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, d_out, d_st, iters, 1u);
	CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	std::vector<Stamp> st(grid * 4);
	CK(hipMemcpy(st.data(), d_st, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
	std::vector<double> cyc, ghz;
	for (auto &s : st) { cyc.push_back((double)(s.c1 - s.c0)); if (s.r1 > s.r0) ghz.push_back((double)(s.c1 - s.c0) / (double)(s.r1 - s.r0) * 0.1); }
	std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
	res.cyc_med = cyc[cyc.size() - 1 - cyc.size() / 100];   /* 99th percentile: the waves that ran to the end */ res.cyc_max = cyc.back(); res.ghz = ghz.empty() ? 0 : ghz[ghz.size() / 2]; res.ms = ms;
	CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
	return 0;
}

int main(int argc, char **argv)
{
	setvbuf(stdout, nullptr, _IOLBF, 0);
	const char *only = argc > 1 ? argv[1] : "";
	unsigned *d_out; Stamp *d_st;
	CK(hipMalloc(&d_out, 256 * 8 * 256 * 4 * 8)); CK(hipMalloc(&d_st, 256 * 8 * 4 * sizeof(Stamp)));
	const int wpss[4] = { 1, 2, 4, 8 };
	Result r;
	if (!*only || !strcmp(only, "A")) {
		struct { const char *n; kern_t k; int per_trip; } ks[] = {
			{ "v_add_u32", v_add, 64 }, { "v_sub_u32", v_sub, 64 }, { "v_xor_b32", v_xor, 64 }, { "v_lshlrev_b32", v_lshl, 64 }, { "v_mov_b32", v_mov, 64 },
			{ "v_lshl_add_u32", v_lshl_add, 64 }, { "v_add_lshl_u32", v_add_lshl, 64 }, { "v_add3_u32", v_add3, 64 },
			{ "v_mad_i32_i24", v_mad24, 64 }, { "v_mad_u32_u24", v_madu24, 64 }, { "v_mul_i32_i24", v_mul24, 64 }, { "v_mul_i32_i24_sdwa", v_mul24s, 64 },
			{ "v_perm_b32", v_perm, 64 }, { "v_fma_f32", v_fma, 64 }, { "v_mul_f32", v_fmac, 64 }, { "v_pk_add_u16", v_pkadd16, 64 },
			{ "v_and_or_b32", v_and_or, 64 }, { "v_bfe_i32", v_bfe, 64 }, { "v_ashrrev_i32", v_ashr, 64 },
			{ "v_add_u32 dpp wave_shr:1", v_add_dpp1, 64 }, { "v_add_u32 dpp row_shr:1", v_add_dppr, 64 }, { "v_add_u32 dpp quad_perm", v_add_dppq, 64 },
			{ "v_mov_b32 dpp wave_shr:1", v_mov_dpp1, 64 }, { "v_add_u32 sdwa", v_add_sdwa, 64 }, { "v_cndmask_b32", v_cndmask, 64 },
			{ "v_lshrrev_b32 sdwa W1", v_lshrsdwa, 64 },
			{ "v_lshrrev_b32", v_lshr, 64 }, { "v_and_b32", v_and, 64 }, { "v_or_b32", v_or, 64 }, { "v_min_i32", v_min, 64 }, { "v_lshl_or_b32", v_lshl_or, 64 },
			{ "v_add_u32 literal", v_add_lit, 64 }, { "v_add_u32 inline const", v_add_inl, 64 }, { "v_sub_u32_e64", v_sub_e64, 64 }, { "v_add_co_u32 (vcc out)", v_add_vcc, 64 },
			{ "v_cmp + v_mov (2 instr)", v_cmp_lt, 128 }, { "v_xad_u32", v_xad, 64 }, { "v_sub_u32 clamp", v_sub_clamp, 64 }, { "v_mov dpp quad_perm", v_mov_dpp_q, 64 },
			{ "v_alignbit_b32", v_alignbit, 64 }, { "v_lshlrev_b32 4", v_lshl4, 64 }, { "v_add_f32", v_add_f32, 64 }, { "v_cvt_f32_i32", v_cvt, 64 },
			{ "v_pk_mad_u16", v_pk_mad16, 64 }, { "v_add_u32 v,s,v", vs_add, 64 }, { "v_lshl_add_u32 v,v,1,s", vs_lshl_add, 64 }, { "v_mul_i32_i24 v,s,v", vs_mul24, 64 },
			{ "v_cndmask_b32_e64 sgpr mask", vs_cnd, 64 }, { "v_mov_b32 v,s", vs_mov, 64 }, { "v_permlane32_swap", v_plswap32, 64 }, { "v_permlane16_swap", v_plswap16, 64 },
		};
		printf("A: cycles per wave-instruction per SIMD (16 independent chains per lane)\n%-28s %8s %8s %8s %8s   GHz\n", "op", "1 w/SIMD", "2", "4", "8");
		for (auto &e : ks) {
			printf("%-28s", e.n);
			double g = 0;
			for (int wps : wpss) { if (run(e.k, wps, 2000, d_out, d_st, r)) return 1; printf(" %8.2f", r.cyc_med / (2000.0 * e.per_trip * wps)); g = r.ghz; }
			printf("   %.2f\n", g);
		}
	}
	if (!*only || !strcmp(only, "B")) {
		struct { const char *n; kern_t k; int per_trip; } ks[] = {
			{ "add + lshl_add", bfly<0>, 64 }, { "sub + mad24", bfly<1>, 64 }, { "alternating both", bfly<2>, 64 },
			{ "add,add,add,(add)", bfly<3>, 64 }, { "add dpp x3,(add)", bfly<4>, 64 }, { "add dpp + lshl_add", bfly<5>, 64 },
			{ "phased: add, lshl_add", bfly_il<0>, 64 }, { "phased: 3 adds", bfly_il<1>, 96 }, { "phased: sub,sub,add", bfly_il<2>, 96 },
		};
		printf("B: butterfly mixes, cycles per wave-instruction per SIMD\n%-28s %8s %8s %8s %8s   GHz\n", "mix", "1 w/SIMD", "2", "4", "8");
		for (auto &e : ks) {
			printf("%-28s", e.n);
			double g = 0;
			for (int wps : wpss) { if (run(e.k, wps, 2000, d_out, d_st, r)) return 1; printf(" %8.2f", r.cyc_med / (2000.0 * e.per_trip * wps)); g = r.ghz; }
			printf("   %.2f\n", g);
		}
	}
	if (!*only || !strcmp(only, "C")) {
		struct { const char *n; kern_t k; int per_trip; int bytes; } ks[] = {
			{ "ds_read_b32", ldsk<0>, 16, 256 }, { "ds_read_b64", ldsk<1>, 16, 512 }, { "ds_read_b128", ldsk<2>, 16, 1024 },
			{ "ds_write_b32", ldsk<3>, 16, 256 }, { "ds_write_b64", ldsk<4>, 16, 512 }, { "ds_write_b128", ldsk<5>, 16, 1024 },
			{ "ds_read_addtid_b32", ldsk<6>, 16, 256 }, { "ds_write_addtid_b32", ldsk<7>, 16, 256 },
			{ "ds_read2_b32", ldsk<8>, 16, 512 }, { "ds_write2_b32", ldsk<9>, 16, 512 },
		};
		printf("C: LDS, cycles per wave-instruction per CU (4 x wps waves issuing), and B/clk/CU\n%-22s %14s %14s %14s %14s\n", "op", "1 w/SIMD", "2", "4", "8");
		for (auto &e : ks) {
			printf("%-22s", e.n);
			for (int wps : wpss) { if (run(e.k, wps, 2000, d_out, d_st, r)) return 1; double c = r.cyc_med / (2000.0 * e.per_trip * wps * 4); printf(" %6.2f (%5.0f)", c, e.bytes / c); }
			printf("\n");
		}
	}
	if (!*only || !strcmp(only, "D")) {
		/* per trip and wave: 16 elements in, NVPER x 32 VALU, 16 elements out */
		struct { const char *n; kern_t k; int valu; int ldsinst; } ks[] = {
			{ "VALU only, 64/trip", passk<2, 0, 0, 0, 1>, 64, 0 },
			{ "LDS b32 only", passk<2, 0, 0, 1, 0>, 0, 32 },
			{ "LDS b64 only", passk<2, 1, 0, 1, 0>, 0, 16 },
			{ "b32 + 64 VALU", passk<2, 0, 0, 1, 1>, 64, 32 },
			{ "b32 + 64 VALU, pipelined", passk<2, 0, 1, 1, 1>, 64, 32 },
			{ "b64 + 64 VALU", passk<2, 1, 0, 1, 1>, 64, 16 },
			{ "b64 + 64 VALU, pipelined", passk<2, 1, 1, 1, 1>, 64, 16 },
			{ "VALU only, 96/trip", passk<3, 0, 0, 0, 1>, 96, 0 },
			{ "b32 + 96 VALU", passk<3, 0, 0, 1, 1>, 96, 32 },
			{ "b32 + 96 VALU, pipelined", passk<3, 0, 1, 1, 1>, 96, 32 },
			{ "b64 + 96 VALU", passk<3, 1, 0, 1, 1>, 96, 16 },
			{ "b64 + 96 VALU, pipelined", passk<3, 1, 1, 1, 1>, 96, 16 },
		};
		printf("D: pass-like loop, cycles per trip per SIMD per wave-slot (= cycles for 16 elements x 64 lanes on one SIMD)\n%-28s %8s %8s %8s %8s\n", "loop", "1 w/SIMD", "2", "4", "8");
		for (auto &e : ks) {
			printf("%-28s", e.n);
			for (int wps : wpss) { if (run(e.k, wps, 1000, d_out, d_st, r)) return 1; printf(" %8.1f", r.cyc_med / (1000.0 * wps)); }
			printf("\n");
		}
	}
	if (!*only || !strcmp(only, "E")) {
		struct { const char *n; kern_t k; int instr; } ks[] = {
			{ "64 VALU (add+lshl_add)", mixk<0, 0, 0>, 64 },
			{ "64 VALU + 4 SALU", mixk<0, 4, 0>, 68 },
			{ "64 VALU + 16 SALU", mixk<0, 16, 0>, 80 },
			{ "64 VALU + 2+2 LDS", mixk<2, 0, 0>, 68 },
			{ "64 VALU + 4+4 LDS", mixk<4, 0, 0>, 72 },
			{ "64 VALU + 8+8 LDS", mixk<8, 0, 0>, 80 },
			{ "64 VALU + 4+4 LDS + 8 SALU", mixk<4, 8, 0>, 80 },
			{ "64 simple VALU (add+sub)", mixk<0, 0, 1>, 64 },
			{ "64 simple + 4+4 LDS", mixk<4, 0, 1>, 72 },
			{ "64 simple + 4+4 LDS + 8 SALU", mixk<4, 8, 1>, 80 },
		};
		printf("E: co-issue, cycles per trip per SIMD per wave-slot (64 VALU per trip; extra instructions cost nothing if they co-issue)\n%-32s %8s %8s %8s %8s\n", "loop", "1 w/SIMD", "2", "4", "8");
		for (auto &e : ks) {
			printf("%-32s", e.n);
			for (int wps : wpss) { if (run(e.k, wps, 1000, d_out, d_st, r)) return 1; printf(" %8.1f", r.cyc_med / (1000.0 * wps)); }
			printf("\n");
		}
	}
	if (!*only || !strcmp(only, "F")) {
		struct { const char *n; kern_t k; } ks[] = {
			{ "none", costk<-1> }, { "ds_read_b32", costk<0> }, { "ds_read_b64", costk<1> }, { "ds_read_b128", costk<2> }, { "ds_read2_b32", costk<3> },
			{ "ds_write_b32", costk<4> }, { "ds_write_b64", costk<5> }, { "ds_write_b128", costk<6> }, { "ds_write2_b32", costk<7> },
			{ "ds_write_addtid_b32", costk<8> }, { "ds_read_addtid_b32", costk<9> }, { "s_add_u32", costk<10> }, { "s_nop 0", costk<11> },
			{ "s_waitcnt (satisfied)", costk<12> }, { "global_load_dword (L2)", costk<13> }, { "global_load_dwordx4 (L2)", costk<14> },
			{ "global_store_dwordx4", costk<15> }, { "global_load_ushort (L2)", costk<16> }, { "v_mov_b32", costk<17> },
		};
		printf("F: SIMD cycles added by ONE extra instruction in a VALU-saturated stream (64 VALU + 8 X per trip and wave)\n%-28s %8s %8s %8s\n", "X", "1 w/SIMD", "2", "4");
		double base[3] = { 0, 0, 0 };
		for (auto &e : ks) {
			printf("%-28s", e.n);
			for (int q = 0; q < 3; q++) {
				if (run(e.k, wpss[q], 1000, d_out, d_st, r)) return 1;
				const double per_simd = r.cyc_med / 1000.0;           /* cycles per trip for all wps waves of the SIMD */
				if (base[q] == 0) { base[q] = per_simd; printf(" %8.1f", per_simd / wpss[q]); }
				else printf(" %8.2f", (per_simd - base[q]) / (8.0 * wpss[q]));
			}
			printf("\n");
		}
	}
	return 0;
}
