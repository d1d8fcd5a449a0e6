/*
 * acm_kernels.hip - the device half of the ACM decode path, gfx950 (CDNA4).
 *
 * What is computed (reference: /root/reference/src/decode.c):
 *   unpack    value = idx * val                      (table :592-600, lookup :174-177)
 *   synthesis `level` cascaded butterfly stages      (juggle :508-526, juggle_block :528-577)
 *   write-out (value >> level) as 16-bit             (:617-677)
 *
 * Formulation (verified against the reference's own juggle_block by
 * tests/test_oracle_vs_ref.py::test_cascade_formulation_equals_reference_juggle): with m the flat sample index of a stream
 * (row*cols + col, running on across blocks) stage k, stride s = cols >> (k+1):
 *     y[m] = 2*x[m-s] + sg*(x[m-2s] + x[m]),  sg = +1 if bit log2(s) of m is 0 else -1
 *     after stage 0 only: y[m] += 1 where m % (cols/2) == 0
 *     x[<0] = 0, all arithmetic mod 2^32.
 * The reference's wrapbuf is just the two previous inputs per column per
 * stage, so an output depends on at most 2*cols-2 earlier raw samples: any
 * run of rows can be synthesised from its own staged rows plus the two rows
 * before it ("halo").  No state is carried between launches.
 *
 * Kernel families:
 *   acm_tile2 (levels 6..14, the whole tiles of streams decoded from row 0: the bulk of a batch): the lean form of the
 *     tile kernel below - 32 KB tiles at four workgroups per CU (levels 13 and 14: one 128 KB tile, one sixteen-wave
 *     workgroup per CU), one record per tile, vector memory issued and waited for by hand.  See the comment in front of it.
 *   fused tile kernel acm_fused_tile (levels 5..12; ragged tails, windows, level 5): persistent workgroups, one tile at a time.  A tile is
 *     TR rows (2 halo + T payload) of one stream held in LDS as int32.  The stages are grouped into passes of
 *     G = 2..4: each thread owns one residue class of the pass's smallest stride and walks it with the inputs
 *     of the G stages in registers (one LDS read + one write per element per PASS, not per stage).  The
 *     first pass is fed straight from HBM (4-byte loads issued one tile ahead, unpacked by an SDWA multiply),
 *     the last one emits packed 16-bit samples that leave as 16 B/lane stores.  HBM traffic: 2 B read
 *     (+2/T halo, mostly L2 hits) + 2 B written per sample.  Two VALU ops per butterfly (sign folding +
 *     v_mad_i32_i24); measured limit is instruction issue, not HBM.
 *   acm_small_level (levels 0..4): the whole cascade in one thread's registers.
 *   level 15, and what acm_tile2 leaves of levels 13..14 (ragged tails, windows, small plans): acm_sw_prefix (unpack + the
 *     first level-12 stages, a register cascade per residue mod 4096, into an int32 plane) + the plane-input build of
 *     the level-12 tile kernel (MODE_PLANE).  12 B of HBM traffic per sample.
 *   stage-wise kernels (any level 0..15; tiles that can see an H1 patch; patched streams of levels 13..15): unpack
 *     to an int32 plane, one elementwise launch per stage (ping-pong planes), emit.  8*level B of HBM traffic per
 *     sample; generic fallback and cross-check.
 *
 * Integer add/shift only: bound by HBM (and LDS/VALU at high levels), no MFMA.
 */
#include <hip/hip_runtime.h>

#include <utility>

#include "acm_device.h"

#ifndef ACM_L14_GROUPS
#define ACM_L14_GROUPS 3, 3, 3, 3, 2
#endif
#ifndef ACM_L13_GROUPS
#define ACM_L13_GROUPS 2, 3, 3, 3, 2
#endif

namespace {

// ---------------------------------------------------------------------------
// sample format (decode.c:617-655): fmt bit0 = big-endian, bit1 = unsigned
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pcm16(int32_t v, int level, unsigned fmt)
{
	uint32_t w = (uint32_t)(v >> level);          // arithmetic shift (:620)
	if (fmt & 2u)
		w += 0x8000u;                         // :640
	w &= 0xFFFFu;
	if (fmt & 1u)
		w = ((w & 0xFFu) << 8) | (w >> 8);    // :631-632
	return w;
}

// ---------------------------------------------------------------------------
// stage-wise family
// ---------------------------------------------------------------------------
constexpr int SW_THREADS = 256;

__global__ void __launch_bounds__(SW_THREADS)
acm_sw_unpack(const AcmDevStream *__restrict__ streams, const uint32_t *__restrict__ list,
	      const int16_t *__restrict__ idx, const acmhip_blkhdr *__restrict__ hdr,
	      int32_t *__restrict__ x, const uint32_t shift)
{
	const AcmDevStream s = streams[list[blockIdx.y]];
	const uint64_t first = (uint64_t)s.halo_row << s.level;
	const uint64_t n = ((uint64_t)(s.nrows - s.halo_row)) << s.level;
	const int16_t *src = idx + s.idx_off + first;
	const acmhip_blkhdr *h = hdr + s.hdr_off;
	int32_t *dst = x + s.scratch_off;
	for (uint64_t e = (uint64_t)blockIdx.x * SW_THREADS + threadIdx.x; e < n;
	     e += (uint64_t)gridDim.x * SW_THREADS) {
		const uint32_t row = s.halo_row + (uint32_t)(e >> s.level);
		const uint32_t val = h[row / s.rows].val;
		dst[e] = (int32_t)((uint32_t)(int32_t)src[e] * (val << shift));    // midbuf[idx] == idx*val (:592-600); shift: see acmk_launch_unpack
	}
}

__global__ void __launch_bounds__(SW_THREADS)
acm_sw_patch(const AcmDevPatch *__restrict__ p, uint64_t n, int32_t *__restrict__ x)
{
	const uint64_t i = (uint64_t)blockIdx.x * SW_THREADS + threadIdx.x;
	if (i < n)
		x[p[i].dst] = p[i].value;
}

/* one butterfly stage, out of place; e counts from the first halo sample of the stream */
__global__ void __launch_bounds__(SW_THREADS)
acm_sw_stage(const AcmDevStream *__restrict__ streams, const uint32_t *__restrict__ list,
	     uint32_t level, uint32_t k, const int32_t *__restrict__ in, int32_t *__restrict__ out, const uint32_t one)
{
	const AcmDevStream s = streams[list[blockIdx.y]];
	const uint64_t n = ((uint64_t)(s.nrows - s.halo_row)) << level;
	const uint32_t sh = level - 1 - k;            // log2(stride)
	const uint64_t st = 1ull << sh;
	const uint32_t *xi = (const uint32_t *)in + s.scratch_off;
	uint32_t *yo = (uint32_t *)out + s.scratch_off;
	const uint64_t halfmask = ((1ull << level) >> 1) - 1;
	for (uint64_t e = (uint64_t)blockIdx.x * SW_THREADS + threadIdx.x; e < n;
	     e += (uint64_t)gridDim.x * SW_THREADS) {
		const uint32_t x0 = xi[e];
		const uint32_t x1 = e >= st ? xi[e - st] : 0u;
		const uint32_t x2 = e >= 2 * st ? xi[e - 2 * st] : 0u;
		uint32_t y = ((e >> sh) & 1) ? 2u * x1 - (x2 + x0)      // :519
					     : 2u * x1 + (x2 + x0);     // :518
		if (k == 0 && (e & halfmask) == 0)
			y += one;                                       // :561-564 (1, or the scale of a level 13-15 prefix)
		yo[e] = y;
	}
}

/*
 * Levels 13-15 without H1 patches: unpack and the first J = level - 12 stages in one sweep, 2 B in and 4 B out per sample
 * (stage by stage it was 2+4, then 4+4 per further stage).  Those stages have strides of 4096 samples and more, so for a fixed
 * residue r = e mod 4096 they are a complete level-J cascade over the subsequence q -> x[r + 4096 q] (2^J entries per row): a
 * thread owns two adjacent residues and walks q with the stage inputs of the last 2^(J+1) - 2 steps in registers; a row of the
 * stream is one unrolled body of 2^J steps, so sign, "+1" and the block's amplitude step are fixed per body position.  The
 * walk of a stream is cut into chunks of PREFIX_CHUNK_ROWS rows; a chunk starts two rows early with zero history, which is exact
 * behind those rows (the cascade reaches back 2^(J+1) - 2 < 2^(J+1) steps), and stores nothing for them.
 */
constexpr int PREFIX_CHUNK_ROWS = 64;
constexpr int PREFIX_THREADS = 256;

template <int J>
__global__ void __launch_bounds__(PREFIX_THREADS)
acm_sw_prefix(const AcmDevStream *__restrict__ streams, const uint32_t *__restrict__ list,
	      const int16_t *__restrict__ idx, const acmhip_blkhdr *__restrict__ hdr,
	      int32_t *__restrict__ y, const uint32_t shift)
{
	constexpr int U = 1 << J;                               /* steps per row */
	constexpr uint32_t RB = 2048 / PREFIX_THREADS;          /* workgroups per chunk: 2048 residue pairs */
	const AcmDevStream s = streams[list[blockIdx.y]];
	const uint32_t nrows = s.nrows - s.halo_row;
	const uint32_t chunk = blockIdx.x / RB;
	const uint32_t row0 = chunk * PREFIX_CHUNK_ROWS;
	if (row0 >= nrows)
		return;
	const uint32_t row_end = min(nrows, row0 + PREFIX_CHUNK_ROWS);
	const uint32_t pair = (blockIdx.x % RB) * PREFIX_THREADS + threadIdx.x;         /* residues 2 pair, 2 pair + 1 */
	const uint64_t first = (uint64_t)s.halo_row << s.level;
	const uint32_t *src = reinterpret_cast<const uint32_t *>(idx + s.idx_off + first) + pair;      /* staged rows are 8 KB multiples: aligned */
	const acmhip_blkhdr *h = hdr + s.hdr_off;
	uint2 *dst = reinterpret_cast<uint2 *>(reinterpret_cast<uint32_t *>(y) + s.scratch_off) + pair;
	const uint32_t one = pair == 0 ? 1u << shift : 0u;      /* decode.c:561-564: residue 0 of stage 0 */

	uint32_t ha[J][U], hb[J][U];                            /* [t][x], x < 2d: the 2d inputs of stage t in front of the body */
#pragma unroll
	for (int t = 0; t < J; t++)
#pragma unroll
		for (int x = 0; x < U; x++)
			ha[t][x] = hb[t][x] = 0u;

	const uint32_t warm = row0 >= 2 ? 2u : row0;
	for (uint32_t row = row0 - warm; row < row_end; row++) {
		const uint32_t val = h[(s.halo_row + row) / s.rows].val << shift;
		uint32_t a[U], b[U];
#pragma unroll
		for (int u = 0; u < U; u++) {
			const uint32_t two = __builtin_nontemporal_load(src + (((uint64_t)row << J) + u) * 2048);
			a[u] = (uint32_t)((int32_t)(two << 16) >> 16) * val;            /* midbuf[idx] == idx*val (:592-600) */
			b[u] = (uint32_t)((int32_t)two >> 16) * val;
		}
#pragma unroll
		for (int t = 0; t < J; t++) {
			const int d = 1 << (J - 1 - t);
			uint32_t ia[U], ib[U];
#pragma unroll
			for (int u = 0; u < U; u++) {
				ia[u] = a[u];
				ib[u] = b[u];
			}
#pragma unroll
			for (int u = 0; u < U; u++) {
				const uint32_t a1 = u >= d ? ia[u - d] : ha[t][u + d], a2 = u >= 2 * d ? ia[u - 2 * d] : ha[t][u];
				const uint32_t b1 = u >= d ? ib[u - d] : hb[t][u + d], b2 = u >= 2 * d ? ib[u - 2 * d] : hb[t][u];
				const bool odd = (u >> (J - 1 - t)) & 1;
				a[u] = odd ? 2u * a1 - (a2 + ia[u]) : 2u * a1 + (a2 + ia[u]);  /* :518-519 */
				b[u] = odd ? 2u * b1 - (b2 + ib[u]) : 2u * b1 + (b2 + ib[u]);
				if (t == 0 && (u % ((U / 2) > 0 ? (U / 2) : 1)) == 0)
					a[u] += one;
			}
#pragma unroll
			for (int x = 0; x < 2 * d; x++) {
				ha[t][x] = ia[U - 2 * d + x];
				hb[t][x] = ib[U - 2 * d + x];
			}
		}
		if (row >= row0) {
#pragma unroll
			for (int u = 0; u < U; u++)
				dst[(((uint64_t)row << J) + u) * 2048] = make_uint2(a[u], b[u]);
		}
	}
}

__global__ void __launch_bounds__(SW_THREADS)
acm_sw_emit(const AcmDevStream *__restrict__ streams, const uint32_t *__restrict__ list,
	    const int32_t *__restrict__ x, int16_t *__restrict__ pcm, unsigned fmt)
{
	const AcmDevStream s = streams[list[blockIdx.y]];
	const int32_t *src = x + s.scratch_off + ((uint64_t)(s.row_begin - s.halo_row) << s.level);
	uint16_t *dst = (uint16_t *)pcm + s.pcm_off;
	for (uint64_t e = (uint64_t)blockIdx.x * SW_THREADS + threadIdx.x; e < s.n_emit;
	     e += (uint64_t)gridDim.x * SW_THREADS)
		dst[e] = (uint16_t)pcm16(src[e], (int)s.level, fmt);
}

// ---------------------------------------------------------------------------
// small levels (cols <= 16): the whole cascade in one thread's registers
// ---------------------------------------------------------------------------
/*
 * For level <= 4 an output depends on fewer than 2*cols - 2 <= 30 earlier samples, so one thread can own SL_K = 32
 * consecutive outputs, load them and the 8 / 16 / 32 samples in front of them that the cascade can reach (sl_halo; aligned
 * 16-byte loads), run every stage over that register array and write 64 bytes of PCM - one launch, 2 B in + 2 B out per sample, instead of the stage-wise
 * family's launch per stage over int32 planes.  Coordinates are the stage-wise family's: e counts from the first
 * staged sample the kernels may read (row halo_row), samples before it are zeros (exact: the reach is < 2 rows).
 * Because chunks start on multiples of 32 >= cols, column, sign and the "+1" of an element are compile-time
 * functions of its position in the register array.
 */
constexpr int SL_K = 32;                 /* outputs per thread */
/* samples in front of them that the thread loads and runs through the stages as well: the cascade reaches back
 * 2*cols - 2 samples, rounded up to whole 16-byte loads and whole rows */
constexpr int sl_halo(int level) { return level <= 2 ? 8 : level == 3 ? 16 : 32; }
/* history the stages 0..k need in front of a position, in samples */
constexpr int sl_reach(int cols, int k) { int r = 0; for (int i = 0; i <= k; i++) r += 2 * (cols >> (i + 1)); return r; }
template <int L, int K> struct SlStage {
	static constexpr int ST = (1 << L) >> (K + 1), LO = sl_reach(1 << L, K);
};
constexpr int SL_THREADS = 256;

template <int L, int K>
__device__ __forceinline__ void sl_stages(uint32_t (&x)[SL_K + sl_halo(L)], const int64_t e0)
{
	if constexpr (K < L) {
		constexpr int ST = SlStage<L, K>::ST, LO = SlStage<L, K>::LO, COLS = 1 << L, N = SL_K + sl_halo(L);
		static_assert(SlStage<L, L - 1>::LO <= sl_halo(L), "the halo covers the cascade's reach");
		/* positions below LO lack history inside the array: never used by a valid output */
#pragma unroll
		for (int jj = N - 1; jj >= LO; jj--) {
			const uint32_t a = x[jj - 2 * ST], z = x[jj - ST], c = x[jj];
			uint32_t y = ((jj / ST) & 1) ? 2u * z - (a + c) : 2u * z + (a + c);     /* decode.c:518-519 */
			if (K == 0 && (jj & (COLS / 2 - 1)) == 0 && e0 + jj >= 0)
				y += 1u;                                                /* :561-564, rows that exist only */
			x[jj] = y;
		}
		sl_stages<L, K + 1>(x, e0);
	}
}

template <int L>
__global__ void __launch_bounds__(SL_THREADS)
acm_small_level(const AcmDevStream *__restrict__ streams, const uint32_t *__restrict__ list,
		const int16_t *__restrict__ idx, const acmhip_blkhdr *__restrict__ hdr,
		int16_t *__restrict__ pcm, unsigned fmt)
{
	constexpr int COLS = 1 << L, H = sl_halo(L), N = SL_K + H;
	static_assert(H % COLS == 0 && N % COLS == 0 && N % 8 == 0, "whole rows, whole 16-byte loads");
	const AcmDevStream s = streams[list[blockIdx.y]];
	const int64_t n_in = (int64_t)(s.nrows - s.halo_row) << L;                  /* staged samples the stream has, from e = 0 */
	const int64_t e_emit = (int64_t)(s.row_begin - s.halo_row) << L;            /* e of the first emitted sample */
	const int16_t *src = idx + s.idx_off + ((uint64_t)s.halo_row << L);
	const acmhip_blkhdr *h = hdr + s.hdr_off;
	uint16_t *dst = reinterpret_cast<uint16_t *>(pcm) + s.pcm_off;
	const uint64_t nchunks = (s.n_emit + SL_K - 1) / SL_K;

	for (uint64_t q = (uint64_t)blockIdx.x * SL_THREADS + threadIdx.x; q < nchunks; q += (uint64_t)gridDim.x * SL_THREADS) {
		const int64_t e0 = e_emit + (int64_t)q * SL_K - H;                  /* e of x[0]; a multiple of cols */
		const uint64_t g0 = q * SL_K;                                       /* first output of the chunk */
		/* staged indices, sign-extended: N samples; zeros outside [0, n_in) */
		int32_t ix[N];
		const bool inside = e0 >= 0 && e0 + N <= n_in && (reinterpret_cast<uintptr_t>(src + e0) & 15u) == 0;
		if (inside) {
#pragma unroll
			for (int v = 0; v < N / 8; v++) {
				const uint4 w = *reinterpret_cast<const uint4 *>(src + e0 + v * 8);
				const uint32_t ww[4] = { w.x, w.y, w.z, w.w };
#pragma unroll
				for (int k = 0; k < 4; k++) {
					ix[v * 8 + 2 * k] = (int32_t)(int16_t)(ww[k] & 0xFFFFu);
					ix[v * 8 + 2 * k + 1] = (int32_t)ww[k] >> 16;
				}
			}
		} else {
#pragma unroll
			for (int k = 0; k < N; k++)
				ix[k] = (e0 + k >= 0 && e0 + k < n_in) ? (int32_t)src[e0 + k] : 0;
		}
		/* unpack: value = idx * val of the sample's block (decode.c:592-600); one header look-up per row */
		uint32_t x[N];
		{
			const int64_t row0 = (int64_t)s.halo_row + (e0 >> L);             /* stream row of x[0] (may be negative) */
			uint32_t blk = row0 > 0 ? (uint32_t)row0 / s.rows : 0u;
			uint32_t rem = row0 > 0 ? (uint32_t)row0 % s.rows : 0u;
#pragma unroll
			for (int r = 0; r < N / COLS; r++) {
				const int64_t row = row0 + r;
				int32_t val = 0;
				if (row >= 0 && row < (int64_t)s.nrows) {
					val = (int32_t)h[blk].val;
					if (++rem == s.rows) {
						rem = 0;
						blk++;
					}
				}
#pragma unroll
				for (int c = 0; c < COLS; c++)
					x[r * COLS + c] = (uint32_t)__mul24(ix[r * COLS + c], val);  /* |idx| < 2^15, val < 2^16 */
			}
		}
		/* the stages, highest position first so that every tap is still the previous stage's value */
		sl_stages<L, 0>(x, e0);
		/* write-out (decode.c:617-655): 32 samples = 64 bytes */
		if (g0 + SL_K <= s.n_emit) {
			uint4 *o = reinterpret_cast<uint4 *>(dst + g0);
#pragma unroll
			for (int v = 0; v < SL_K / 8; v++) {
				uint32_t w[4];
#pragma unroll
				for (int k = 0; k < 4; k++)
					w[k] = pcm16((int32_t)x[H + v * 8 + 2 * k], L, fmt) | pcm16((int32_t)x[H + v * 8 + 2 * k + 1], L, fmt) << 16;
				o[v] = make_uint4(w[0], w[1], w[2], w[3]);
			}
		} else {
#pragma unroll
			for (int k = 0; k < SL_K; k++)
				if (g0 + k < s.n_emit)
					dst[g0 + k] = (uint16_t)pcm16((int32_t)x[H + k], L, fmt);
		}
	}
}

// ---------------------------------------------------------------------------
// fused tile kernel
// ---------------------------------------------------------------------------
/* tile configuration: level, threads per workgroup, tile elements held in LDS */
template <int L_, int NT_, int NELEM_>
struct TileCfg {
	static constexpr int L = L_;
	static constexpr int NT = NT_;
	static constexpr int NELEM = NELEM_;
	static constexpr int COLS = 1 << L_;
	static constexpr int TR = NELEM_ / COLS;          // tile rows incl. the 2 halo rows
	static constexpr int NJ_LAST = NELEM_ / NT_;      // samples per thread in the last pass
	static constexpr int PS = NJ_LAST >= 64 ? 6 : 5;  // LDS pad: one dword per 2^PS elements (= one last-pass walk)
	static_assert(NJ_LAST == 64 || NJ_LAST == 32 || NJ_LAST == 128, "walk length of the last pass");
	static_assert(TR >= 1, "whole rows");
	static constexpr bool ROW_PAIRS = TR >= 2 && (TR % 2) == 0;     /* what the first passes fed from HBM work on (the halo flavour of acm_fused_tile needs four rows: two of them are halo) */
};

/* t - 2*z: one VALU op when 25 result bits suffice (level <= 9: the write-out
 * only looks at bits [level, level+16) and every op here is add/shift, so bit
 * i of a result depends on bits <= i of its inputs), else shift+sub. */
template <bool EXACT32>
__device__ __forceinline__ uint32_t sub_twice(uint32_t t, uint32_t z)
{
	if constexpr (EXACT32) {
		return t - (z << 1);
	} else {
		int32_t y;
		asm("v_mad_i32_i24 %0, %1, -2, %2" : "=v"(y) : "v"((int32_t)z), "v"((int32_t)t));
		return (uint32_t)y;
	}
}

/* idx * val with both operands inside 24 bits: one full-rate multiply (the compiler otherwise falls back to
 * the quarter-rate v_mul_lo_u32 when it cannot prove the operand ranges across the prefetch loop) */
template <bool PAIR>
__device__ __forceinline__ uint32_t mul_idx_val(uint32_t loaded, int32_t val, int word)
{
	/* SDWA: operand 0 = one sign-extended 16-bit word of the loaded register (low word, or either word of a
	 * 4-byte load holding two adjacent columns), so no extraction / extension op is needed */
	uint32_t y;
	if (PAIR && word == 1)
		asm("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
		    : "=v"(y) : "v"(loaded), "v"(val));
	else
		asm("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD"
		    : "=v"(y) : "v"(loaded), "v"(val));
	return y;
}

/* four loaded registers (two adjacent columns each) -> eight values, as ONE asm statement: the compiler puts an s_nop
 * between an asm statement and a VALU op that reads its result (it cannot see that the destination is a whole dword),
 * and every instruction, s_nop included, costs a SIMD ~3 cycles of issue */
__device__ __forceinline__ void mul_idx_val_x4(const uint32_t r0, const uint32_t r1, const uint32_t r2, const uint32_t r3, const int32_t val,
					       uint32_t (&lo)[4], uint32_t (&hi)[4])
{
#define ACM_SDWA_MUL(D, S, WORD) "v_mul_i32_i24_sdwa " D ", sext(" S "), %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" WORD " src1_sel:DWORD\n\t"
	asm(ACM_SDWA_MUL("%0", "%8", "WORD_0") ACM_SDWA_MUL("%1", "%8", "WORD_1")
	    ACM_SDWA_MUL("%2", "%9", "WORD_0") ACM_SDWA_MUL("%3", "%9", "WORD_1")
	    ACM_SDWA_MUL("%4", "%10", "WORD_0") ACM_SDWA_MUL("%5", "%10", "WORD_1")
	    ACM_SDWA_MUL("%6", "%11", "WORD_0") ACM_SDWA_MUL("%7", "%11", "WORD_1")
	    : "=&v"(lo[0]), "=&v"(hi[0]), "=&v"(lo[1]), "=&v"(hi[1]), "=&v"(lo[2]), "=&v"(hi[2]), "=&v"(lo[3]), "=&v"(hi[3])
	    : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(val));
#undef ACM_SDWA_MUL
}

/* t + 2*z as exactly one VALU op (kept opaque so that the compiler does not re-associate the butterfly) */
__device__ __forceinline__ uint32_t add_twice(uint32_t t, uint32_t z)
{
	uint32_t y;
	asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(y) : "v"(z), "v"(t));
	return y;
}

/*
 * Sign folding.  Odd positions of a stage need 2*z1 - (z2+z0); to keep every
 * element-stage at two VALU ops the stages alternate between two conventions:
 *   "P": inputs plain, output of an odd position is stored negated
 *   "N": inputs arrive negated where the PREVIOUS stage's parity bit is set
 *        (exactly what a P stage leaves behind), outputs plain
 * The last stage is always N, so the cascade ends with plain values; when that
 * makes stage 0 an N stage (odd level) the raw inputs must be negated on odd
 * tile rows, which the unpack does for free by multiplying with -val.
 */
template <int L, int K>
struct StageKind {
	static constexpr bool N = ((L - 1 - K) % 2) == 0;
};

/*
 * LDS layout of the tile: one pad dword after every 2^PS elements (PS = 5 or 6: one last-pass walk),
 * addr(m) = m + (m >> PS).  A wave reading residue i of stride sigma < 64 touches 64/sigma walk segments whose
 * starts are 64*sigma apart - all on the same banks without the pad, rotated by sigma banks each with it.  That
 * removes the systematic n-way conflicts; what the counters still see at level 9 (SQ_LDS_BANK_CONFLICT = 9 % of
 * SQ_LDS_IDX_ACTIVE, profiles/r2_level9_summary.txt) comes from the two-address ds_read2 / ds_write2 forms, whose
 * halves land on the same bank for part of the (segment, residue) pairs of a half-wave.
 */
template <int PS = 6>
__device__ __forceinline__ int lds_at(int m) { return m + (m >> PS); }
/*
 * Where the last pass parks its packed samples: dwords between the 16-byte pieces of one thread's parking area (the start of its own,
 * consumed, walk).  The write-out reads piece p of owner o into lane 4 o + p (32-sample walks): side by side (4 dwords apart) the
 * pieces of the four lanes that share an owner meet the next owners' on the same banks - owner o sits 33 o dwords on, bank o + 4 p -
 * a two-way conflict on every read of the write-out (SQ_LDS_BANK_CONFLICT, profiles/r5_level9_summary.txt).  Eight apart they do not
 * (bank o + 8 p: 32 lanes, 32 banks), and there is room: the samples of a body need half the dwords its inputs had.
 */
template <int NJ_LAST>
constexpr int park_piece() { return NJ_LAST == 32 ? 8 : 4; }

/*
 * Write-out without per-sample shifts where possible.  The whole cascade is linear mod 2^32 and only bits
 * [level, level+16) of the result are used, so it may run on values scaled by a power of two (the unpack
 * multiplies by val << SHIFT, the "+1" becomes 1 << SHIFT) that moves the sample onto a byte boundary:
 *   level <= 8 : SHIFT = 8 - level, sample = bytes 1..2 - bits [0,24) still inside the 25 bits that the
 *                one-op v_mad_i32_i24 butterfly keeps exact;
 *   level == 9 : no scaling (bits 9..24 are needed and 25 is all mad24 gives): two shifts per pair instead;
 *   level >= 10: these levels use exact 32-bit butterflies anyway, SHIFT = 16 - level, sample = bytes 2..3.
 * Two samples are then packed, byte-swapped if asked, by ONE v_perm_b32; unsigned output is one xor per pair.
 */
template <int L>
struct OutScale {
	static constexpr int SHIFT = (L <= 8) ? 8 - L : (L >= 10 ? 16 - L : 0);
	static constexpr int BYTE = (L <= 8) ? 1 : (L >= 10 ? 2 : 0);     /* first byte of the sample inside the value */
	static constexpr bool PRESHIFT = (L == 9);
};

struct PcmFmt {
	uint32_t sel;     /* v_perm_b32 selector */
	uint32_t flip;    /* xor mask for unsigned output */
};

template <int L>
__device__ __forceinline__ PcmFmt make_pcm_fmt(unsigned fmt)
{
	PcmFmt f;
	const bool be = fmt & 1u, uns = fmt & 2u;
	/* v_perm_b32(S0 = second value, S1 = first value): byte k of S1 is index k, of S0 index 4+k */
	constexpr uint32_t lo = OutScale<L>::BYTE, hi = OutScale<L>::BYTE + 1;
	f.sel = be ? ((4 + lo) << 24 | (4 + hi) << 16 | lo << 8 | hi)
		   : ((4 + hi) << 24 | (4 + lo) << 16 | hi << 8 | lo);
	f.flip = uns ? (be ? 0x00800080u : 0x80008000u) : 0u;
	return f;
}

template <int L, bool FLIP, bool BIGEND = true>
__device__ __forceinline__ uint32_t pack_pcm(uint32_t a, uint32_t b, const PcmFmt &f)
{
	if (OutScale<L>::PRESHIFT) {
		if (!BIGEND) {
			/* two ops per pair: a >> L, then SDWA drops the low half of b >> L into the upper word */
			uint32_t p = a >> L;
			asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
			    : "+v"(p) : "v"((uint32_t)L), "v"(b));
			return FLIP ? (p ^ f.flip) : p;
		}
		a >>= L;
		b >>= L;
	}
	const uint32_t p = __builtin_amdgcn_perm(b, a, f.sel);
	return FLIP ? (p ^ f.flip) : p;                 /* signed output (the common case) skips the xor */
}

template <class C, int K0, int G>
struct PassGeo {
	static constexpr int L = C::L;
	static constexpr int NT = C::NT;
	static constexpr int COLS = C::COLS;
	static constexpr int NELEM = C::NELEM;
	static constexpr int SIGMA = COLS >> (K0 + G);          // smallest stride of the pass
	static constexpr int U = 1 << G;
	static constexpr int BODY = 2 * U;                      // elements per unrolled body
	static constexpr int NJ_TOTAL = NELEM / SIGMA;          // walk length of one residue over the tile
	static_assert(SIGMA < NT, "LDS passes need more threads than residues (the first pass takes the wide strides)");
	static constexpr int NSEG = NT / SIGMA;                 // walk segments per residue
	static constexpr int NJ = NJ_TOTAL / NSEG;              // walk length per thread
	static_assert(SIGMA >= 1, "pass exceeds level");
	static_assert(NJ % BODY == 0 && NJ >= BODY, "segment must be whole bodies");
	/* LDS offset of walk element u relative to the body's first element (which is
	 * aligned to BODY*SIGMA, and 64 | BODY*SIGMA or BODY*SIGMA | 64) */
	static constexpr int off(int u) { return u * SIGMA + ((u * SIGMA) >> C::PS); }
};

/*
 * G butterfly stages over BODY consecutive elements of one residue class.
 * v: in = stage-K0 inputs, out = stage-(K0+G-1) outputs (conventions above).
 * h[t][x], x in [0, 2d): the 2d inputs of stage t that precede this body
 * (d = stride of stage t in walk units = 2^(G-1-t)); updated on exit.
 * bias_lo / bias_hi: the "+1" of decode.c:561-564 for the first / second half
 * of the body (only the thread owning residue 0 of a stage-0 pass passes 1,
 * and only for rows that exist: history before the stream start is all-zero).
 */
template <int L, int K0, int G>
__device__ __forceinline__ void pass_body(uint32_t (&v)[2 << G], uint32_t (&h)[G][1 << G],
					  uint32_t bias_lo, uint32_t bias_hi)
{
	constexpr int U = 1 << G, BODY = 2 * U;
	constexpr bool EXACT32 = (L > 9);
#pragma unroll
	for (int t = 0; t < G; t++) {
		const int d = 1 << (G - 1 - t);
		const int pb = G - 1 - t;                       // parity bit of this stage within u
		const bool kindN = ((L - 1 - (K0 + t)) % 2) == 0;
		uint32_t in[BODY];
#pragma unroll
		for (int u = 0; u < BODY; u++)
			in[u] = v[u];
#pragma unroll
		for (int u = 0; u < BODY; u++) {
			const uint32_t z0 = in[u];
			const uint32_t z1 = (u >= d) ? in[u - d] : h[t][u + d];
			const uint32_t z2 = (u >= 2 * d) ? in[u - 2 * d] : h[t][u];
			const int beta = (u >> pb) & 1;
			uint32_t y;
			/* the "+1" rides on the first op (an add3 at worst) */
			const bool biased = (K0 + t == 0) && (u % ((U / 2) > 0 ? (U / 2) : 1)) == 0;
			const uint32_t b = biased ? ((u < U) ? bias_lo : bias_hi) : 0u;
			if (!kindN) {
				/* even: y = (z2+z0+b) + 2*z1;  odd: -y = (z2+z0-b) - 2*z1 */
				y = beta ? sub_twice<EXACT32>(biased ? z2 + z0 - b : z2 + z0, z1)
					 : add_twice(biased ? z2 + z0 + b : z2 + z0, z1);
			} else {
				const int aneg = (u >> (pb + 1)) & 1;
				if ((aneg ^ beta) == 0)
					y = sub_twice<EXACT32>(biased ? z0 - z2 + b : z0 - z2, z1);
				else
					y = add_twice(biased ? z2 - z0 + b : z2 - z0, z1);
			}
			v[u] = y;
		}
#pragma unroll
		for (int x = 0; x < 2 * d; x++)
			h[t][x] = in[BODY - 2 * d + x];
	}
}

template <int G>
__device__ __forceinline__ void clear_hist(uint32_t (&h)[G][1 << G])
{
#pragma unroll
	for (int t = 0; t < G; t++)
#pragma unroll
		for (int x = 0; x < (1 << G); x++)
			h[t][x] = 0u;
}

/* what a workgroup needs to know about one tile (all wave-uniform) */
struct TileCtx {
	const int16_t *src;      /* staged row 0 of the stream */
	const acmhip_blkhdr *hdr;
	uint16_t *dst;           /* where sample (row_begin, 0) goes */
	uint64_t n_emit;
	int row_first;           /* stream row of tile row 0 (may be < 0) */
	int nrows;
	int row_begin;
	uint32_t rows;           /* acm_rows */
	bool fresh, discard;     /* carry mode: ACM_TILE_* of this tile */
};

/*
 * First pass (stages 0..G-1): inputs come straight from HBM (staged int16
 * indices, unpacked with the row's val), outputs go to the LDS tile.  A body is
 * exactly two tile rows of one residue (2^G elements per row).  Split in two so
 * that the loads of the NEXT tile can be in flight while this tile's LDS passes
 * run: load() only issues global loads into `raw`, compute() consumes them.
 */
/* bit of the ABL template argument that is not an ablation: the first pass reads an int32 plane (stages already applied by
 * the stage-wise kernels: levels 13-15) instead of staged indices - no unpack multiply, no "+1" */
constexpr int MODE_PLANE = 64;
/* another one: the tile belongs to ONE wavefront (acm_chunk) - the LDS operations of a wavefront are carried out in the order they were
 * issued, so what orders its lanes among themselves is the order of the instructions: no workgroup barrier, only a fence the compiler
 * may not move LDS accesses across */
constexpr int MODE_WAVE = 128;
/* and: do not read the next body of an LDS pass while this one is computed (level 11 of acm_chunk keeps three rows of staged bytes in
 * registers through its LDS passes and has none to spare) */
constexpr int MODE_LEAN = 256;
template <int ABL>
__device__ __forceinline__ void tile_barrier()
{
	if constexpr ((ABL & MODE_WAVE) != 0) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	} else if constexpr (!(ABL & 32)) {
		__syncthreads();
	}
}

template <class C, int G, int W, int ABL = 0, bool FORCE_WARM = false>
struct FirstPass {
	static constexpr int L = C::L, NT = C::NT, COLS = C::COLS;
	static constexpr int U = 1 << G, BODY = 2 * U;
	static constexpr int SIGMA = COLS >> G;                 // stride between a residue's consecutive columns
	static constexpr int TPS = SIGMA / W;                   // threads per segment, each owning W adjacent residues
	static constexpr int NSEG = NT / TPS;                   // row segments per tile
	static constexpr int RPS = C::TR / NSEG;                // rows per segment
	static constexpr int NB = RPS / 2;                      // bodies (row pairs) per segment
	static constexpr bool WARM = NSEG > 1 || FORCE_WARM;    // segments > 0 re-run the two rows in front of them (FORCE_WARM: segment 0 too)
	static constexpr int NRAW = (NB + (WARM ? 1 : 0)) * BODY;   // loaded registers: one per (row, q), W samples each
	static constexpr bool PLANE = (ABL & MODE_PLANE) != 0;
	static constexpr int NREG = NRAW * (PLANE ? W : 1);        // a plane holds int32: the second column of a lane sits NRAW further on
	static_assert(W == 1 || W == 2, "1 or 2 adjacent columns per lane");
	static_assert(C::ROW_PAIRS, "a body is two tile rows");
	static_assert(SIGMA % W == 0 && TPS <= NT && NT % TPS == 0, "segment geometry");
	static_assert(RPS >= 2 && RPS % 2 == 0 && NSEG * RPS == C::TR, "segments are whole row pairs");
	/* LDS offset of the residue's q-th column / second row relative to (first row, first column) of the body */
	static constexpr int off(int u) { return u * SIGMA + ((u * SIGMA) >> C::PS); }

	/* every staged index of this thread's walk, issued back to back (one HBM round trip): 2-byte loads for
	 * W = 1, 4-byte loads (two adjacent columns) for W = 2.  Rows that do not exist are read from a clamped
	 * address; their rowval is 0. */
	static __device__ __forceinline__ void load(uint32_t (&raw)[NREG], const TileCtx &t, const int tid)
	{
		const int seg = tid / TPS;
		const int i0 = (tid % TPS) * W;
		const int lr_seg = seg * RPS;
		const int last_row = t.nrows - 1;
		/* lowest row any lane may touch: segment 0's (zero-weighted) warm-up sits two rows above the tile */
		const int base_row = t.row_first - 2 < 0 ? 0 : (t.row_first - 2 > last_row ? last_row : t.row_first - 2);
		const uint16_t *tbase = reinterpret_cast<const uint16_t *>(t.src) + ((size_t)base_row << L);   /* wave-uniform; per-lane offsets stay 32-bit */
		const uint32_t *pbase = reinterpret_cast<const uint32_t *>(t.src) + ((size_t)base_row << L);   /* MODE_PLANE: t.src is an int32 plane */
		/* interior tile (the common case): every row from row_first-2 to row_first+TR-1 exists, so the
		 * offsets are lane-constant + compile-time constants; otherwise clamp each row into the stream */
		const bool interior = (t.row_first >= 2) && (t.row_first + C::TR <= t.nrows);
		const unsigned lane_off = ((unsigned)(lr_seg + 2) << L) + (unsigned)i0;
#pragma unroll
		for (int b = (WARM ? -1 : 0); b < NB; b++) {
#pragma unroll
			for (int half = 0; half < 2; half++) {
				const int lr = lr_seg + 2 * b + half;
				unsigned off0;
				if (interior) {
					off0 = lane_off + (unsigned)((2 * b + half) << L);
				} else {
					int rho = t.row_first + lr;
					rho = rho < 0 ? 0 : (rho > last_row ? last_row : rho);
					off0 = ((unsigned)(rho - base_row) << L) + (unsigned)i0;
				}
#pragma unroll
				for (int q = 0; q < U; q++) {
					uint32_t x;
					if (ABL & 1)
						x = off0 + q;
					else if (PLANE && W == 2) {
						const uint2 two = *reinterpret_cast<const uint2 *>(pbase + off0 + q * SIGMA);
						x = two.x;
						raw[NRAW + (b + (WARM ? 1 : 0)) * BODY + half * U + q] = two.y;
					} else if (PLANE)
						x = pbase[off0 + q * SIGMA];
					else if (W == 1)
						x = tbase[off0 + q * SIGMA];
					else
						x = *reinterpret_cast<const uint32_t *>(tbase + off0 + q * SIGMA);
					raw[(b + (WARM ? 1 : 0)) * BODY + half * U + q] = x;
				}
			}
		}
	}

	/* rowval[lr + 2] = +-val of tile row lr (pre-scaled), 0 for rows that do not exist (also lr = -2, -1) */
	template <bool CARRY = false>
	static __device__ __forceinline__ void compute(const uint32_t (&raw)[NREG], uint32_t *tile, const int32_t *rowval,
						       const int row_first, const int tid)
	{
		constexpr int LR_MIN = CARRY ? -2 : 0;          /* carry mode: the two rows above the tile carry weight */
		const int seg = tid / TPS;
		const int i0 = (tid % TPS) * W;
		const int lr_seg = seg * RPS;
		uint32_t h[W][G][U];
#pragma unroll
		for (int w = 0; w < W; w++)
			clear_hist<G>(h[w]);
		/* the loads were issued a whole tile ago: touching the YOUNGEST one first makes the compiler emit a
		 * single s_waitcnt vmcnt for all of them instead of one per consumer */
		asm volatile("" :: "v"(raw[NREG - 1]));
#pragma unroll
		for (int b = (WARM ? -1 : 0); b < NB; b++) {
			const int lr0 = lr_seg + 2 * b;
			const int32_t v0 = rowval[lr0 + 2], v1 = rowval[lr0 + 3];
			constexpr uint32_t ONE = 1u << OutScale<L>::SHIFT;
			const uint32_t b0 = (!(ABL & MODE_PLANE) && i0 == 0 && lr0 >= LR_MIN && row_first + lr0 >= 0) ? ONE : 0u;
			const uint32_t b1 = (!(ABL & MODE_PLANE) && i0 == 0 && lr0 + 1 >= LR_MIN && row_first + lr0 + 1 >= 0) ? ONE : 0u;
			uint32_t v[W][BODY];
			if constexpr (PLANE) {
				/* rowval is a mask here: all ones for rows that exist, 0 for rows in front of / behind the stream
				 * (they were read from a clamped address) */
#pragma unroll
				for (int w = 0; w < W; w++)
#pragma unroll
					for (int u = 0; u < BODY; u++)
						v[w][u] = raw[w * NRAW + (b + (WARM ? 1 : 0)) * BODY + u] & (uint32_t)(u < U ? v0 : v1);
			} else if constexpr (W == 2 && U % 4 == 0) {
#pragma unroll
				for (int u = 0; u < BODY; u += 4) {
					const uint32_t *r = &raw[(b + (WARM ? 1 : 0)) * BODY + u];
					uint32_t lo[4], hi[4];
					mul_idx_val_x4(r[0], r[1], r[2], r[3], u < U ? v0 : v1, lo, hi);
#pragma unroll
					for (int k = 0; k < 4; k++) {
						v[0][u + k] = lo[k];
						v[W - 1][u + k] = hi[k];
					}
				}
			} else {
#pragma unroll
				for (int w = 0; w < W; w++)
#pragma unroll
					for (int u = 0; u < BODY; u++)
						v[w][u] = mul_idx_val<W == 2>(raw[(b + (WARM ? 1 : 0)) * BODY + u], u < U ? v0 : v1, w);
			}
#pragma unroll
			for (int w = 0; w < W; w++)
				if (!(ABL & 4))
					pass_body<L, 0, G>(v[w], h[w], w == 0 ? b0 : 0u, w == 0 ? b1 : 0u);
			if (b >= 0) {
				/* rows of a segment start on a multiple of 64 elements: body b sits at a constant offset */
				static_assert((2 * COLS) % 64 == 0 && (RPS * COLS) % 64 == 0, "row pairs are whole 64-element groups");
				uint32_t *o = tile + lds_at<C::PS>(lr_seg * COLS + i0) + b * (2 * COLS + ((2 * COLS) >> C::PS));
#pragma unroll
				for (int u = 0; u < BODY; u++)
#pragma unroll
					for (int w = 0; w < W; w++)
						o[off(u) + w] = v[w][u];
			}
		}
	}
};

/*
 * Middle / last passes (stages K0..K0+G-1), in place on the LDS tile.
 * LAST: the outputs are final values; they are converted to 16-bit samples,
 * packed two per dword and parked at the start of the thread's own (already
 * consumed) segment: sample e of thread `tid` -> dword lds_at(tid*NJ) + e/2.
 */
/* padded size of a carry buffer holding the BS elements in front of a tile (same pad rule as P::off) */
constexpr int carry_words(int bs, int ps = 6) { return bs + (bs >> ps) + 2; }

/* bias / bias_warm (a pass that starts at stage 0 only: acm_tile2p runs ALL its stages in LDS): the "+1" of decode.c:561-564 for
 * the thread that owns residue 0 - in the bodies of its walk / in the warm-up body in front of it (0 where those rows do not exist) */
template <class C, int K0, int G, bool LAST, int ABL = 0, bool FLIP = true, bool BIGEND = true, bool CARRY = false>
__device__ __forceinline__ void lds_pass(uint32_t *tile, const int tid, const unsigned fmt, uint32_t *carry = nullptr, const uint32_t bias = 0u,
					 const uint32_t bias_warm = 0u)
{
	using P = PassGeo<C, K0, G>;
	constexpr int L = C::L;
	constexpr int U = P::U, BODY = P::BODY, SIGMA = P::SIGMA;
	static_assert(!LAST || SIGMA == 1, "last pass must end at stride 1");

	const int seg = tid / SIGMA;
	const int i = tid % SIGMA;
	const int m_seg = seg * P::NJ * SIGMA + i;              // first element of this thread's walk
	const PcmFmt pf = make_pcm_fmt<L>(fmt);
	/* one runtime address per thread; every body of the walk (and the warm-up body in front of it) sits at a
	 * compile-time offset from it: NJ*SIGMA is a multiple of 64 and bodies never straddle a pad dword */
	constexpr int PS = C::PS, PG = 1 << PS;
	uint32_t *const base = tile + lds_at<PS>(m_seg);
	constexpr int BS = BODY * SIGMA;
	constexpr bool ALIGNED = (P::NJ * SIGMA) % PG == 0;     // segments start on a pad group
	auto body_ptr = [&](int it) -> uint32_t * {
		if constexpr (ALIGNED)
			return base + (it * BS + ((it * BS) >> PS));
		else
			return tile + lds_at<PS>(m_seg + it * BS);
	};
	uint32_t h[G][U];
	clear_hist<G>(h);

	/* warm-up: the BODY elements in front of this segment belong to the previous
	 * segment's owner, who is about to overwrite them in place - read them
	 * first, then everybody may start walking.  Segment 0 reads the zeroed guard
	 * zone in front of the tile (history before the tile = zeros). */
	tile_barrier<ABL>();
	uint32_t w[BODY];
	constexpr int NTAIL = CARRY ? (BS + C::NT - 1) / C::NT : 1;     /* carried elements per thread (1 for every default geometry) */
	uint32_t tail[NTAIL];
	{
		const uint32_t *pw = ALIGNED ? base - (BS + (BS >= PG ? BS / PG : 1)) : tile + lds_at<PS>(m_seg - BS);
		if constexpr (CARRY) {
			/* segment 0's history is what the previous tile of this stream left behind: the last BS elements
			 * of its input to this pass, kept in `carry` with the same pad rule (element j at j + off-pad) */
			if (seg == 0)
				pw = carry + i;
#pragma unroll
			for (int k = 0; k < NTAIL; k++) {
				const int j = tid + k * C::NT;
				tail[k] = (j < BS) ? tile[lds_at<PS>(C::NELEM - BS + j)] : 0u;      /* this tile's bequest, read before it is overwritten */
			}
		}
#pragma unroll
		for (int u = 0; u < BODY; u++)
			w[u] = pw[P::off(u)];
	}
	tile_barrier<ABL>();
	if constexpr (CARRY) {
#pragma unroll
		for (int k = 0; k < NTAIL; k++) {
			const int j = tid + k * C::NT;
			if (j < BS)
				carry[j + (((j / SIGMA) * SIGMA) >> PS)] = tail[k];     /* next read: this pass of the next tile */
		}
	}

	/* software pipeline: the reads of body k+1 are in flight while body k is computed */
	constexpr int NBODY = P::NJ / BODY;
	uint32_t nxt[BODY];
	{
#pragma unroll
		for (int u = 0; u < BODY; u++)
			nxt[u] = base[P::off(u)];
	}
	if (!(ABL & 2))
		pass_body<L, K0, G>(w, h, K0 == 0 ? bias_warm : 0u, K0 == 0 ? bias_warm : 0u);

#pragma unroll
	for (int it = 0; it < NBODY; it++) {
		uint32_t *p = body_ptr(it);
		uint32_t v[BODY];
#pragma unroll
		for (int u = 0; u < BODY; u++)
			v[u] = nxt[u];
		if (it + 1 < NBODY && !(ABL & MODE_LEAN)) {
			const uint32_t *pn = body_ptr(it + 1);
#pragma unroll
			for (int u = 0; u < BODY; u++)
				nxt[u] = pn[P::off(u)];
		}
		if (!(ABL & 2))
			pass_body<L, K0, G>(v, h, K0 == 0 ? bias : 0u, K0 == 0 ? bias : 0u);
		if (it + 1 < NBODY && (ABL & MODE_LEAN) != 0) {
			/* short of registers: the next body is asked for behind this one's butterflies, not beside them */
			__builtin_amdgcn_sched_barrier(0);
			const uint32_t *pn = body_ptr(it + 1);
#pragma unroll
			for (int u = 0; u < BODY; u++)
				nxt[u] = pn[P::off(u)];
		}
		if constexpr (!LAST) {
#pragma unroll
			for (int u = 0; u < BODY; u++)
				p[P::off(u)] = v[u];
		} else {
#pragma unroll
			for (int u = 0; u < BODY; u += 2) {
				const int d = it * (BODY / 2) + u / 2;          /* dword of the thread's samples */
				base[(d / 4) * park_piece<C::NJ_LAST>() + d % 4] = pack_pcm<L, FLIP, BIGEND>(v[u], v[u + 1], pf);
			}
		}
	}
}

#ifdef ACM_STAMPS
/* diagnostic build only (profiles/ubench/phases.hip): per-workgroup cycle sums of the tile loop's phases */
__device__ unsigned long long g_acm_stamps[2048][8];
__device__ __forceinline__ unsigned long long stamp_now()
{
	unsigned long long t;
	__builtin_amdgcn_sched_barrier(0);
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
	__builtin_amdgcn_sched_barrier(0);
	return t;
}
#define ACM_STAMP(k) do { const unsigned long long n_ = stamp_now(); acc_[k] += n_ - last_; last_ = n_; } while (0)
#else
#define ACM_STAMP(k) do { } while (0)
#endif

/* the passes after the first: stage groups G, Rest... starting at stage K0; the last one emits PCM */
/* dwords of carry buffer the LDS passes G, Rest... (starting at stage K0) need between them: every pass its own size - the
 * first one's is the widest by far (two bodies of its smallest stride) */
template <class C, int K0, int G, int... Rest>
constexpr int carry_total()
{
	constexpr int mine = carry_words(PassGeo<C, K0, G>::BODY * PassGeo<C, K0, G>::SIGMA, C::PS);
	if constexpr (sizeof...(Rest) == 0)
		return mine;
	else
		return mine + carry_total<C, K0 + G, Rest...>();
}

template <class C, int ABL, bool CARRY, int K0, int G, int... Rest>
__device__ __forceinline__ void run_lds_passes(uint32_t *tile, int tid, unsigned fmt, uint32_t *carry = nullptr, const uint32_t bias = 0u,
					       const uint32_t bias_warm = 0u)
{
	constexpr bool last = sizeof...(Rest) == 0;
	static_assert(!last || K0 + G == C::L, "stage groups must add up to the level");
	uint32_t *cb = CARRY ? carry : nullptr;
	if constexpr (!last) {
		lds_pass<C, K0, G, false, ABL, true, true, CARRY>(tile, tid, fmt, cb, bias, bias_warm);
		run_lds_passes<C, ABL, CARRY, K0 + G, Rest...>(tile, tid, fmt,
							       CARRY ? carry + carry_words(PassGeo<C, K0, G>::BODY * PassGeo<C, K0, G>::SIGMA, C::PS) : nullptr);
	} else if (fmt == ACMHIP_FMT_S16LE) {
		lds_pass<C, K0, G, true, ABL, false, false, CARRY>(tile, tid, fmt, cb);   /* the common layout: no xor, no byte swap */
	} else {
		lds_pass<C, K0, G, true, ABL, true, true, CARRY>(tile, tid, fmt, cb);     /* any other layout through the general path */
	}
}

/*
 * Wave priorities (s_setprio).  The four workgroups of a CU are in different phases of their tiles, and a SIMD issues one
 * instruction at a time: when a wave in an LDS pass (read, wait, butterflies, write - latency-bound) competes with a wave in
 * its first pass (a long run of VALU work that can run any time), the LDS-pass wave must win, or its LDS round trips queue
 * up behind arithmetic that is not urgent.  Measured on the level-9 batch (same box, interleaved runs): 0.542 of the roofline
 * with equal priorities, 0.568-0.580 with the LDS passes above the first pass, 0.543-0.545 with the first pass at or above
 * them; the exact levels do not matter (1 against 0 gains as much as 3 against 2), nor does ranking the LDS passes among
 * themselves.  Loads and the waits stay at 0.  Level 7 +2 %, 8 +3 %, 9 +6.5 %, 10 +9 %, 11 +5 % (profiles/ab_levels.sh).
 * With one workgroup per CU (level 12) every wave of a SIMD is in the same phase and the priorities only cost (-3 %): off.
 */
template <bool ON, int P>
__device__ __forceinline__ void phase_prio()
{
	if constexpr (ON)
		__builtin_amdgcn_s_setprio(P);
}
#ifndef ACM_PRIO_FIRST
#define ACM_PRIO_FIRST 2
#endif
#ifndef ACM_PRIO_LDS
#define ACM_PRIO_LDS 3
#endif
constexpr int PRIO_IDLE = 0, PRIO_FIRST_PASS = ACM_PRIO_FIRST, PRIO_LDS_PASSES = ACM_PRIO_LDS;

/*
 * Persistent workgroups: workgroup w handles tiles w, w + gridDim.x, ...  While the LDS passes of tile n
 * run, the staged indices and block headers of tile n+1 are already on their way from HBM (registers), so
 * the only exposed memory latency is the very first tile's.
 * C: tile configuration; G0, Gs...: how the `level` stages are grouped into passes (first pass fed from
 * HBM, the others in LDS).
 */
template <class C, int WAVES_PER_SIMD, int ABL, int W0, bool CARRY, int G0, int... Gs>
__global__ void __launch_bounds__(C::NT, WAVES_PER_SIMD)
acm_fused_tile(const AcmDevStream *__restrict__ streams, const AcmTile *__restrict__ tiles, const uint32_t ntiles,
	       const int16_t *__restrict__ idx, const acmhip_blkhdr *__restrict__ hdr,
	       int16_t *__restrict__ pcm, unsigned fmt)
{
	constexpr int L = C::L, NT = C::NT, COLS = C::COLS, NELEM = C::NELEM, TR = C::TR, NJ_LAST = C::NJ_LAST;
	constexpr bool NEG_ODD_ROWS = StageKind<L, 0>::N;       // stage 0 wants odd tile rows negated
	static_assert(!(ABL & MODE_PLANE) || !NEG_ODD_ROWS, "a plane comes with plain signs");
	constexpr int NRV = (TR + 2 + NT - 1) / NT;             // rowval entries per thread
	using FP = FirstPass<C, G0, W0, ABL>;
	constexpr bool PRIO = WAVES_PER_SIMD * 256 / NT > 1;    // several workgroups per CU: see phase_prio

	constexpr int GUARD = NELEM / 32 + 64;                  // zeros in front of the tile: segment 0's warm-up reads land here
	__shared__ uint32_t tile_mem[GUARD + NELEM + (NELEM >> C::PS)];
	__shared__ int32_t rowval[2][TR + 2];                   // [buf][lr + 2]; two leading zeros for segment 0's warm-up
	uint32_t *const tile = tile_mem + GUARD;
	/* carry mode: per LDS pass, the tail of the previous tile's input to that pass (the first LDS pass has the
	 * widest: two bodies of its smallest stride) */
	constexpr int NCARRY_WORDS = CARRY ? carry_total<C, G0, Gs...>() : 1;
	__shared__ uint32_t carry_mem[NCARRY_WORDS];
	/* payload rows of a tile: all of them in carry mode, all but the two halo rows otherwise */
	constexpr int HALO = CARRY ? 0 : 2;

	const int tid = threadIdx.x;

	auto fetch_ctx = [&](uint32_t t) -> TileCtx {
		const AcmTile tl = tiles[t];
		const AcmDevStream s = streams[tl.stream];
		TileCtx c;
		c.src = (ABL & MODE_PLANE) ? reinterpret_cast<const int16_t *>(reinterpret_cast<const int32_t *>(idx) + s.idx_off) : idx + s.idx_off;
		c.hdr = hdr + s.hdr_off;
		c.dst = reinterpret_cast<uint16_t *>(pcm) + s.pcm_off;
		c.n_emit = s.n_emit;
		c.row_first = (int)tl.row0 - HALO;
		c.nrows = (int)s.nrows;
		c.row_begin = (int)s.row_begin;
		c.rows = s.rows;
		c.fresh = tl.flags & ACM_TILE_FRESH;
		c.discard = tl.flags & ACM_TILE_DISCARD;
		return c;
	};
	/* the block's val (decode.c:589) of every tile row, pre-scaled and signed per the stage-0 convention */
	auto fetch_vals = [&](int32_t (&hv)[NRV], const TileCtx &c) {
#pragma unroll
		for (int k = 0; k < NRV; k++) {
			const int lr = tid + k * NT - 2;
			const int rho = c.row_first + lr;
			hv[k] = 0;
			if (lr >= -2 + HALO && lr < TR && rho >= 0 && rho < c.nrows)
				hv[k] = (ABL & MODE_PLANE) ? -1 : (int32_t)c.hdr[(uint32_t)rho / c.rows].val;
		}
	};
	auto store_vals = [&](const int32_t (&hv)[NRV], int32_t *rv) {
#pragma unroll
		for (int k = 0; k < NRV; k++) {
			const int lr = tid + k * NT - 2;
			if (lr < TR) {
				int32_t v = (ABL & MODE_PLANE) ? hv[k] : (int32_t)((uint32_t)hv[k] << OutScale<L>::SHIFT);
				if (NEG_ODD_ROWS && (lr & 1))
					v = -v;
				rv[lr + 2] = v;
			}
		}
	};

	/* which tiles: every gridDim.x-th one, or in carry mode a contiguous run of the table (a run that starts in
	 * the middle of a stream first replays the tile in front of it as a lead-in: one tile is enough for the carries
	 * to be exact, every pass reaches back less than a tile) */
	uint32_t t = blockIdx.x, t_end = ntiles;
	bool lead_in = false;
	if constexpr (CARRY) {
		const uint32_t per = (ntiles + gridDim.x - 1) / gridDim.x;
		t = blockIdx.x * per;
		t_end = t + per < ntiles ? t + per : ntiles;
		if (t < ntiles && !(tiles[t].flags & ACM_TILE_FRESH)) {
			lead_in = true;
			t--;
		}
	}
	if (t >= t_end)
		return;
	for (int k = tid; k < GUARD; k += NT)
		tile_mem[k] = 0u;                               /* never written again */
	TileCtx cur = fetch_ctx(t);
	if (lead_in) {
		cur.fresh = true;
		cur.discard = true;
	}
	uint32_t raw[FP::NREG];
	int32_t hv[NRV];
	fetch_vals(hv, cur);
	FP::load(raw, cur, tid);
	store_vals(hv, rowval[0]);
	int buf = 0;
#ifdef ACM_STAMPS
	unsigned long long acc_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
	unsigned long long last_ = stamp_now();
#endif

	for (;;) {
		if constexpr (CARRY) {
			if (cur.fresh)                          /* nothing in front of this tile: history is zeros (util.c:241) */
				for (int k = tid; k < NCARRY_WORDS; k += NT)
					carry_mem[k] = 0u;
		}
		__syncthreads();                                /* rowval[buf] complete; previous write-out done with the tile */
		ACM_STAMP(0);
		phase_prio<PRIO, PRIO_FIRST_PASS>();
		FP::template compute<CARRY>(raw, tile, rowval[buf], cur.row_first, tid);
		phase_prio<PRIO, PRIO_IDLE>();
		ACM_STAMP(1);

		/* prefetch the next tile: context (scalar), headers and staged indices (registers) */
		const uint32_t tn = CARRY ? t + 1 : t + gridDim.x;
		const bool more = tn < t_end;
		TileCtx nxt = cur;
		if (more) {
			nxt = fetch_ctx(tn);
			fetch_vals(hv, nxt);
			FP::load(raw, nxt, tid);
		}

		ACM_STAMP(2);
		phase_prio<PRIO, PRIO_LDS_PASSES>();
		if (!(ABL & 8))
			run_lds_passes<C, ABL, CARRY, G0, Gs...>(tile, tid, fmt, carry_mem);
		phase_prio<PRIO, PRIO_IDLE>();
		ACM_STAMP(3);
		__syncthreads();
		ACM_STAMP(4);

		/* write-out of the payload rows (tile rows 2..TR-1): 8 samples (16 B) per lane per step,
		 * gathered from the per-thread parking areas of the last pass */
		if (!(CARRY && cur.discard)) {
			constexpr int NVEC = (TR - HALO) * COLS / 8;    /* 16-byte pieces of the payload */
			constexpr int PER_OWNER = NJ_LAST / 8;          /* pieces per parking area */
			const uint64_t g0 = (uint64_t)(uint32_t)(cur.row_first + HALO - cur.row_begin) << L;   /* first payload sample */
			const bool whole = (cur.row_first + TR <= cur.nrows) && (g0 + (uint64_t)(TR - HALO) * COLS <= cur.n_emit);
			if (whole) {
				/* interior tile (the common case): no per-piece checks, 32-bit offsets from a uniform base */
				uint4 *out = reinterpret_cast<uint4 *>(cur.dst + g0);
#pragma unroll
				for (int k = 0; k < (NVEC + NT - 1) / NT; k++) {
					const int vec = tid + k * NT;
					if (k < NVEC / NT || vec < NVEC) {      /* only the last round can be partial */
						const int owner = (HALO * COLS / 8 + vec) / PER_OWNER;
						const int piece = (HALO * COLS / 8 + vec) % PER_OWNER;
						const uint32_t *q = tile + lds_at<C::PS>(owner * NJ_LAST) + piece * park_piece<NJ_LAST>();
						uint4 o;
						o.x = q[0];
						o.y = q[1];
						o.z = q[2];
						o.w = q[3];
						if (!(ABL & 16) || o.x == 0x12345u)
							out[vec] = o;
					}
				}
			} else {
				for (int vec = tid; vec < NVEC; vec += NT) {
					const int ml = HALO * COLS + vec * 8;
					const int lr = ml >> L;
					const int col = ml & (COLS - 1);
					const int rho = cur.row_first + lr;
					if (rho >= cur.nrows)
						break;
					const uint64_t g = ((uint64_t)(uint32_t)(rho - cur.row_begin) << L) + (uint32_t)col;
					if (g >= cur.n_emit)
						break;
					const int owner = ml / NJ_LAST;
					const uint32_t *q = tile + lds_at<C::PS>(owner * NJ_LAST) + (ml % NJ_LAST) / 8 * park_piece<NJ_LAST>();
					const uint32_t w[4] = { q[0], q[1], q[2], q[3] };
					if (g + 8 <= cur.n_emit) {
						*reinterpret_cast<uint4 *>(cur.dst + g) = make_uint4(w[0], w[1], w[2], w[3]);
					} else {
						for (int e = 0; e < 8 && g + e < cur.n_emit; e++)
							cur.dst[g + e] = (uint16_t)(w[e >> 1] >> ((e & 1) * 16));
					}
				}
			}
		}
		ACM_STAMP(5);
		if (!more)
			break;
		store_vals(hv, rowval[buf ^ 1]);
		cur = nxt;
		t = tn;
		buf ^= 1;
	}
#ifdef ACM_STAMPS
	if ((tid & 63) == 0 && blockIdx.x < 2048 / 4) {
		for (int k = 0; k < 8; k++)
			g_acm_stamps[blockIdx.x * 4 + (tid >> 6) % 4][k] = acc_[k];
	}
#endif
}

/*
 * Kernel variants (ACM_K1_VARIANT=n picks one; tuning aid, see profiles/sweep_variants.py):
 *   0  default: per level the fastest measured geometry
 *   1  256 threads, 16K-element tiles, 2-byte first-pass loads (64 samples per thread per pass)
 *   2  512 threads (32 samples per thread per pass, 16 waves per CU)
 *   3  256 threads on half-size tiles (4 workgroups per CU)
 *   4  as 1 with 4-byte first-pass loads (two adjacent columns per lane)
 *   5  as 1 with fewer, deeper LDS passes
 *   6, 7  alternative stage groupings;  8  128-thread workgroups (128 samples per thread per pass)
 */
struct FusedEntry {
	void (*fn)(const AcmDevStream *, const AcmTile *, uint32_t, const int16_t *, const acmhip_blkhdr *, int16_t *, unsigned);
	int threads;
	int tile_rows;
	int wg_per_cu;       /* resident workgroups per CU (LDS-limited) */
	void (*fn_carry)(const AcmDevStream *, const AcmTile *, uint32_t, const int16_t *, const acmhip_blkhdr *, int16_t *, unsigned);
};

template <class C, int W, int... Gs>
constexpr FusedEntry entry() { return FusedEntry{ acm_fused_tile<C, W, 0, 1, false, Gs...>, C::NT, C::TR, W * 256 / C::NT, nullptr }; }
/* same with two adjacent columns per lane in the first pass (4-byte HBM loads) */
template <class C, int W, int... Gs>
constexpr FusedEntry entry2() { return FusedEntry{ acm_fused_tile<C, W, 0, 2, false, Gs...>, C::NT, C::TR, W * 256 / C::NT, nullptr }; }
/* ... plus the carry-mode build of the same geometry (no halo rows; see ACM_TILE_*) */
template <class C, int W, int... Gs>
constexpr FusedEntry entry2c() { return FusedEntry{ acm_fused_tile<C, W, 0, 2, false, Gs...>, C::NT, C::TR, W * 256 / C::NT,
						     acm_fused_tile<C, W, 0, 2, true, Gs...> }; }
#ifdef ACM_ABLATION
/* timing-only builds of the level-7 and level-9 kernels with parts removed (wrong output by design) */
template <class C, int W, int ABL, int... Gs>
constexpr FusedEntry abl2() { return FusedEntry{ acm_fused_tile<C, W, ABL, 2, false, Gs...>, C::NT, C::TR, W * 256 / C::NT, nullptr }; }
template <class C, int W, int ABL, int... Gs>
constexpr FusedEntry abl() { return FusedEntry{ acm_fused_tile<C, W, ABL, 1, false, Gs...>, C::NT, C::TR, W * 256 / C::NT, nullptr }; }
#endif

/* variants 1.. are tuning aids (geometry sweeps: -DACM_TUNING; timing-only ablations: -DACM_ABLATION, which implies it) */
#if defined(ACM_ABLATION) && !defined(ACM_TUNING)
#define ACM_TUNING 1
#endif
#ifdef ACM_ABLATION
constexpr int NVARIANTS = 20;
#elif defined(ACM_TUNING)
constexpr int NVARIANTS = 10;
#else
constexpr int NVARIANTS = 1;
#endif
const FusedEntry g_fused[NVARIANTS][ACM_K1_MAX_LEVEL - ACM_K1_MIN_LEVEL + 1] = {
	{	/* variant 0 (default): per level the fastest measured geometry (profiles/sweep_variants.py) */
		entry2c<TileCfg<5, 128, 8192>, 2, 2, 3>(),
		entry2c<TileCfg<6, 256, 16384>, 2, 2, 2, 2>(),
		entry2c<TileCfg<7, 256, 16384>, 2, 2, 2, 3>(),
		entry2c<TileCfg<8, 256, 16384>, 2, 3, 3, 2>(),
		entry2c<TileCfg<9, 256, 16384>, 2, 3, 3, 3>(),
		entry2c<TileCfg<10, 256, 16384>, 2, 3, 3, 4>(),
		entry2c<TileCfg<11, 512, 32768>, 2, 3, 4, 4>(),
		entry2c<TileCfg<12, 512, 32768>, 2, 3, 3, 3, 3>(),
	},
#include "acm_kernels_tuning.inc"     /* variants 1.. (-DACM_TUNING) and the timing-only ablation builds (-DACM_ABLATION) */
};


// ---------------------------------------------------------------------------
// K2: the lean tile kernel for the bulk of a batch
// ---------------------------------------------------------------------------
/*
 * Same passes as acm_fused_tile in carry mode, for the tiles that need no special handling: every tile row exists,
 * every payload sample is emitted, the stream is decoded from its row 0 (AcmTile2 records, cut by the planner;
 * whatever is left - the ragged tail of a stream, windows that start inside a stream - goes to acm_fused_tile).
 * Measured on gfx950 (profiles/ubench/issue_model.hip): a SIMD issues ONE instruction at a time whatever its kind
 * (simple VALU 2.25 cycles, other VALU 4.2, LDS 5-6.5, SALU / s_waitcnt / s_nop ~3, global_load_dword 6.6, with four
 * waves per SIMD), so the tile loop is written for instruction count: 32 KB tiles at four workgroups per CU, all
 * per-tile scalars in one 32-byte record, no clamping, no selects, one address register for all staged-index loads.
 */
/* a wave-uniform address as an SGPR pair (readfirstlane says so to the compiler): the base operand of the hand-issued loads */
__device__ __forceinline__ const uint8_t *sgpr_u64(const uint64_t a)
{
	const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
	return reinterpret_cast<const uint8_t *>(((uint64_t)hi << 32) | lo);
}

template <class C, int G, int W, int ABL = 0>
struct FirstPass2 : FirstPass<C, G, W, ABL, true> {              /* every segment, the first one included, re-runs the two rows in front of it */
	using FP = FirstPass<C, G, W, ABL, true>;
	static constexpr int L = C::L, COLS = C::COLS, U = FP::U, BODY = FP::BODY, SIGMA = FP::SIGMA, NB = FP::NB, NRAW = FP::NRAW;
	static constexpr bool WARM = FP::WARM;
	static_assert(FP::WARM, "K2 geometry: every segment re-runs the two rows in front of it");
	/* byte offset of this thread's first staged index relative to tile row -2 */
	static __device__ __forceinline__ uint32_t lane_offset(const int tid)
	{
		const int seg = tid / FP::TPS, i0 = (tid % FP::TPS) * W;
		return (uint32_t)((seg * FP::RPS * COLS + i0) * 2);
	}
	/* base = staged index of (tile row -2, column 0); voff_warm = voff except for segment 0 of a stream's first tile,
	 * whose two rows in front do not exist (they are read from rows 0..1 instead and weigh 0).
	 * The loads are issued by hand (see k2_wait): one SGPR base, one VGPR offset, compile-time immediates. */
	template <int K>
	static __device__ __forceinline__ void load_one(uint32_t (&raw)[NRAW], const uint8_t *base, const uint32_t voff, const uint32_t voff_warm)
	{
		constexpr int b = K / BODY - 1, half = (K % BODY) / U, q = K % U;
		constexpr int off = (((2 * (b + 1) + half) * COLS) + q * SIGMA) * 2;
		constexpr int imm = off % 4096, far = off - imm;        /* 12 bits in the instruction, the rest on the scalar base */
		if (ABL & 1)                            /* timing-only build: no HBM loads */
			raw[K] = voff + off;
		else
			asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(raw[K]) : "v"(b < 0 ? voff_warm : voff), "s"(base + far), "n"(imm) : "memory");
	}
	template <int... Ks>
	static __device__ __forceinline__ void load_all(uint32_t (&raw)[NRAW], const uint8_t *base, const uint32_t voff, const uint32_t voff_warm,
							std::integer_sequence<int, Ks...>)
	{
		(load_one<Ks>(raw, base, voff, voff_warm), ...);
	}
	/* rows are asked for in address order */
	static __device__ __forceinline__ void load(uint32_t (&raw)[NRAW], const uint8_t *base, const uint32_t voff, const uint32_t voff_warm)
	{
		static_assert(W == 2, "two adjacent columns per lane");
		load_all(raw, base, voff, voff_warm, std::make_integer_sequence<int, NRAW>{});
	}
	/* what acm_tile2 asks of its first pass (FirstPassM below answers the same questions for the byte-plane form) */
	struct Raw { uint32_t r[NRAW]; };
	static __device__ __forceinline__ void load(Raw &raw, const uint8_t *base, const uint32_t voff, const uint32_t voff_warm)
	{
		load(raw.r, base, voff, voff_warm);
	}
	struct Tables { };
	struct Desc { };
	static __device__ __forceinline__ Desc fetch_desc(const uint32_t *, const AcmTile2 &, const int) { return Desc{}; }
	/* base = staged index of (tile row -2, column 0) */
	static __device__ __forceinline__ void issue(Raw &raw, const int16_t *idx, const AcmTile2 &r, const Desc &, const int, const uint32_t voff, const uint32_t voff_warm)
	{
		load(raw, sgpr_u64(reinterpret_cast<uint64_t>(idx) + 2 * (r.idx_off - 2 * (uint64_t)COLS)), voff, voff_warm);
	}
	static constexpr bool KEEPS_HISTORY = false;
	static constexpr bool SIGNED_ROWVAL = true;              /* odd tile rows carry -val when stage 0 is an N stage; rows in front of a stream weigh 0 */
	static __device__ __forceinline__ bool fresh_lane(const int tid) { return tid < FP::TPS; }      /* lanes whose two rows in front are missing in a stream's first tile */
	static __device__ __forceinline__ void fill_tables(Tables &, const int) { }
	static __device__ __forceinline__ void run(const Raw &raw, uint32_t *tile, const int32_t *rowval, const bool fresh_stream, const int tid, const Tables &, const Desc &,
						   const uint32_t = 0u)
	{
		FP::template compute<true>(raw.r, tile, rowval, fresh_stream ? 0 : 2, tid);
	}
};

/*
 * The same first pass on the matrix cores, for the byte-plane staged form (acmhip_mform_rows, acm_pack.cpp).
 * Three stages over one residue class are a banded integer matrix (tools/gen_mfma_tables.py): the 16 stage-2 outputs of a row
 * pair = A (16 x 32, |coefficient| <= 8) x the 32 stage-0 inputs of that pair and the pair in front.  With the unpack multiply
 * moved behind the matrix (val is constant over a block; the rows of a unit mostly share it) one v_mfma_i32_16x16x32_i8 does
 * 16 outputs x 16 residues, and what is left for the vector ALU is one shift-add (low and high byte planes) and one multiply-add
 * (val, and what the "+1" of decode.c:561-564 has become) per sample instead of two per sample and stage plus the unpack.
 * Staged bytes are the same two per sample as the int16 form, in the order the B operand wants them: per row, per residue
 * c < cols/8, eight low bytes (columns c + q*cols/8, stored minus 128 so that they are signed) and eight high bytes.  The -128
 * comes back through the accumulator input (128 x the coefficient sums, ACM_MF_KROW).
 * A unit = (row pair, 16 residues); a wave does eight of them per tile.  Lane l feeds input row l/16 of the four (two in
 * front, the pair itself) for residue l%16 and receives outputs 4*(l/16) .. +3 of that residue.
 * Units whose four rows do not share one val (a block boundary inside) run the matrix once per row, with the other rows'
 * coefficients masked out of A (a lane only holds the coefficients of its own input row: the mask is a lane select).
 */
typedef int32_t v4i_t __attribute__((ext_vector_type(4)));
typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
#include "acm_mfma_tables.inc"

/* G = 3: v_mfma_i32_16x16x32_i8 (8 operand bytes per lane); G = 4: the first FOUR stages, 32 outputs from 64 inputs per residue class and row
 * pair: two v_mfma_i32_16x16x64_i8 (16 operand bytes per lane) per byte plane, one per row of the pair; a staged row then holds 16 low and
 * 16 high bytes per residue c < cols/16.  (Five stages would need 128 inputs per output and sums beyond the 24-bit multiplier.) */
template <int G> struct MfmaTables;
template <> struct MfmaTables<3> {
	static __device__ __forceinline__ int a(int v, int m, int k) { return ACM_MF_A3[v][m][k]; }
	static __device__ __forceinline__ int krow(int v, int r, int m) { return ACM_MF_KROW3[v][r][m]; }
	static __device__ __forceinline__ int bias(int v, int f, int m) { return ACM_MF_BIAS3[v][f][m]; }
};
template <> struct MfmaTables<4> {
	static __device__ __forceinline__ int a(int v, int m, int k) { return ACM_MF_A4[v][m][k]; }
	static __device__ __forceinline__ int krow(int v, int r, int m) { return ACM_MF_KROW4[v][r][m]; }
	static __device__ __forceinline__ int bias(int v, int f, int m) { return ACM_MF_BIAS4[v][f][m]; }
};

template <class C, int G_>
struct FirstPassM {
	static constexpr int L = C::L, NT = C::NT, COLS = C::COLS, TR = C::TR;
	static constexpr int G = G_, QN = 1 << G, SIGMA = COLS / QN;    /* QN columns of a residue class per row */
	static_assert(G == 3 || G == 4, "one matrix instruction spans the four input rows: 4 x 2^G = its K");
	static_assert(C::ROW_PAIRS, "a unit is a row pair");
	static constexpr int NM = 2 * QN / 16;                  /* matrix instructions per unit and byte plane: 16 outputs each (G = 4: one per row of the pair) */
	static constexpr int LB = QN;                           /* operand bytes per lane: lane l feeds input row l / 16, all its QN columns */
	static constexpr int NGRP = SIGMA / 16;                 /* groups of 16 residues per row pair */
	static constexpr int NW = NT / 64;
	static constexpr int NUNIT = (TR / 2) * NGRP;
	static constexpr int NU = NUNIT / NW;                   /* units per wave and tile */
	static constexpr int VPU = 2 * LB / 16;                 /* 16-byte loads per unit and lane at 16 bits per index: low bytes, high bytes (G = 3: both in one) */
	static constexpr int NRAW = NU * VPU * 4;
	static_assert(SIGMA % 16 == 0 && NU * NW == NUNIT && NU >= 1, "whole units per wave");
	/* a wave's units are consecutive in (row pair, group) order: several pairs per wave, or several waves per pair */
	static constexpr bool MANYG = NGRP > NU;
	static_assert(MANYG ? NGRP % NU == 0 : NU % NGRP == 0, "units of a wave are whole row pairs, or a whole fraction of one");
	static constexpr int WPP = MANYG ? NGRP / NU : 1;       /* waves per row pair */
	static constexpr int UPP = MANYG ? NU : NGRP;           /* units of one row pair in a wave */
	static constexpr int NPW = NU / UPP;                    /* row pairs per wave */
	static constexpr int ROWB_W = COLS * 2;                 /* staged bytes per row at 16 bits per index (class 3); class c: >> (3 - c) */
	static constexpr int RESB_W = 2 * QN;                   /* ... per row and residue */
	static constexpr int VARIANT = StageKind<L, G - 1>::N ? 0 : 1;          /* a P stage leaves odd positions negated */
	static constexpr bool KEEPS_HISTORY = false;
	static constexpr bool SIGNED_ROWVAL = false;            /* rowval = val << SHIFT for every row; rows in front of a stream repeat row 0's */
	using T = MfmaTables<G>;
	typedef std::conditional_t<G == 3, uint64_t, v4i_t> Operand;

	struct Raw { v4u_t r[NU * VPU]; };
	/* pair-table entries (include/acm_hip.h: offset in 64-byte units << 2 | width class) of the wave's row pairs and of the pair in front of them */
	struct Desc { uint32_t e[NPW + 1]; };
	/* per-lane operands that never change, parked in LDS between tiles (registers are what the LDS passes are short of) */
	struct Tables {
		Operand a[NM][64];          /* lane's coefficients for output tile mt: output 16 mt + l%16, inputs of row l/16 */
		v4i_t kc[NM][4];            /* [mt][rs] accumulator input of outputs 16 mt + 4 rs .. +3: 128 x the coefficient sums of all four rows */
		v4i_t krow[4][NM][4];       /* [input row][mt][rs]: the same for one input row */
		v4i_t khalf[2][NM][4];      /* [rows in front / the pair itself][mt][rs] */
		v4i_t bias[2][NM][5];       /* [rows in front missing][mt][rs, 4 = lanes that do not own residue 0]: the "+1" response, scaled */
	};

	/* f(width class as a compile-time constant): one wave-uniform three-way branch, straight-line code behind it */
	template <class F>
	static __device__ __forceinline__ void by_class(const uint32_t cls, F &&f)
	{
		if (cls == ACMHIP_BP_WORD)
			f(std::integral_constant<uint32_t, ACMHIP_BP_WORD>{});
		else if (cls == ACMHIP_BP_BYTE)
			f(std::integral_constant<uint32_t, ACMHIP_BP_BYTE>{});
		else
			f(std::integral_constant<uint32_t, ACMHIP_BP_NIBBLE>{});
	}

	/* what every row pair of a wave needs from the tables, fetched once per tile */
	struct Pre {
		Operand a[NM];
		v4i_t kc[NM];
	};
	static constexpr int pair_of(int k) { return MANYG ? 0 : k / NGRP; }                    /* relative to the wave's first row pair */
	static constexpr int grp_of(int k) { return MANYG ? k : k % NGRP; }                     /* relative to the wave's first group */
	static __device__ __forceinline__ int pair0(const int tid) { return MANYG ? (tid >> 6) / WPP : (tid >> 6) * (NU / (MANYG ? 1 : NGRP)); }
	static __device__ __forceinline__ int grp0(const int tid) { return MANYG ? ((tid >> 6) % WPP) * NU : 0; }
	static __device__ __forceinline__ uint32_t lane_offset(const int tid)
	{
		const int lane = tid & 63, rs = lane >> 4, n = lane & 15;
		return (uint32_t)(rs * ROWB_W + (grp0(tid) * 16 + n) * RESB_W);
	}
	static __device__ __forceinline__ bool fresh_lane(const int) { return false; }         /* the pair in front of a stream exists in this form: zeros */

	/* through the scalar cache: the tile's entry (the pair in front of it) + the wave's first pair are wave-uniform */
	static __device__ __forceinline__ Desc fetch_desc(const uint32_t *__restrict__ pairs, const AcmTile2 &r, const int tid)
	{
		Desc d;
		const uint32_t at = __builtin_amdgcn_readfirstlane((uint32_t)r.idx_off + (uint32_t)pair0(tid));    /* (a table of < 2^32 entries) */
#pragma unroll
		for (int j = 0; j <= NPW; j++)
			d.e[j] = pairs[at + j];
		return d;
	}

	/* the loads of one tile: per unit, lanes 0-31 ask for the rows of the pair in front, lanes 32-63 for the pair's own, each at the
	 * width its pair is stored at (16 bytes are asked for whatever the class: what lies behind a narrow residue's bytes is its neighbour's) */
	template <int PP, int... Js>
	static __device__ __forceinline__ void issue_pair(Raw &raw, const uint8_t *arena, const Desc &d, const int tid, const uint32_t lane16, std::integer_sequence<int, Js...>)
	{
		const int lane = tid & 63, rs = lane >> 4, n = lane & 15;
		const uint32_t ef = d.e[PP], ec = d.e[PP + 1];
		const uint8_t *base = sgpr_u64(reinterpret_cast<uint64_t>(arena) + ((uint64_t)(ef >> 2) << 6));
		/* the offsets differ, the loads do not: one sequence of them, whatever the widths (a load inside a branch would also hide from
		 * tests/test_isa_invariants.py which registers are in flight) */
		uint32_t lane_part, ustride;
		if ((ef & 3u) == (ec & 3u)) {
			/* both pairs at one width (the usual case): the pair's own rows follow the rows in front at a fixed distance, every lane's
			 * offset is the 16-bit one shifted */
			const uint32_t sh = 3u - (ec & 3u);
			lane_part = lane16 >> sh;
			ustride = (uint32_t)(16 * RESB_W) >> sh;
		} else {
			const uint32_t delta = ((ec >> 2) - (ef >> 2)) << 6;
			const bool front = rs < 2;
			const uint32_t sh = front ? 3u - (ef & 3u) : 3u - (ec & 3u);
			lane_part = ((uint32_t)((rs & 1) * ROWB_W + (grp0(tid) * 16 + n) * RESB_W) >> sh) + (front ? 0u : delta);
			ustride = (uint32_t)(16 * RESB_W) >> sh;
		}
		auto one = [&](auto jj) {
			constexpr int k = PP * UPP + decltype(jj)::value;
			const uint32_t voff = lane_part + (uint32_t)grp_of(k) * ustride;
			asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(raw.r[k * VPU]) : "v"(voff), "s"(base) : "memory");
			if constexpr (VPU == 2)
				asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(raw.r[k * VPU + 1]) : "v"(voff), "s"(base) : "memory");
		};
		(one(std::integral_constant<int, Js>{}), ...);
	}
	template <int... PPs>
	static __device__ __forceinline__ void issue_pairs(Raw &raw, const uint8_t *arena, const Desc &d, const int tid, const uint32_t lane16, std::integer_sequence<int, PPs...>)
	{
		(issue_pair<PPs>(raw, arena, d, tid, lane16, std::make_integer_sequence<int, UPP>{}), ...);
	}
	/* lane16 = lane_offset(tid): the lane's byte offset inside a unit's four rows when all of them are at 16 bits */
	static __device__ __forceinline__ void issue(Raw &raw, const int16_t *idx, const AcmTile2 &, const Desc &d, const int tid, const uint32_t lane16, const uint32_t)
	{
		issue_pairs(raw, reinterpret_cast<const uint8_t *>(idx), d, tid, lane16, std::make_integer_sequence<int, NPW>{});
	}

	static __device__ __forceinline__ void fill_tables(Tables &t, const int tid)
	{
		constexpr int32_t ONE = 1 << OutScale<L>::SHIFT;
		if (tid < 64) {
			const int rs = tid >> 4, m = tid & 15;
			for (int mt = 0; mt < NM; mt++) {
				uint32_t w[4] = { 0, 0, 0, 0 };
				for (int j = 0; j < LB; j++)
					w[j / 4] |= (uint32_t)(uint8_t)T::a(VARIANT, 16 * mt + m, LB * rs + j) << (8 * (j % 4));
				if constexpr (G == 3)
					t.a[mt][tid] = ((uint64_t)w[1] << 32) | w[0];
				else
					t.a[mt][tid] = v4i_t{ (int)w[0], (int)w[1], (int)w[2], (int)w[3] };
			}
		}
		if (tid < 4 * NM) {
			const int rs = tid & 3, mt = tid >> 2;
			v4i_t kc = { 0, 0, 0, 0 };
			for (int r = 0; r < 4; r++) {
				v4i_t kr;
				for (int i = 0; i < 4; i++)
					kr[i] = T::krow(VARIANT, r, 16 * mt + 4 * rs + i);
				t.krow[r][mt][rs] = kr;
				kc += kr;
				if (r == 1)
					t.khalf[0][mt][rs] = kc;
			}
			t.kc[mt][rs] = kc;
			t.khalf[1][mt][rs] = kc - t.khalf[0][mt][rs];
			for (int f = 0; f < 2; f++) {
				v4i_t b;
				for (int i = 0; i < 4; i++)
					b[i] = T::bias(VARIANT, f, 16 * mt + 4 * rs + i) * ONE;
				t.bias[f][mt][rs] = b;
			}
		}
		if (tid < 2 * NM)
			t.bias[tid & 1][tid >> 1][4] = v4i_t{ 0, 0, 0, 0 };
	}

	static __device__ __forceinline__ v4i_t mfma(const Operand a, const Operand b, const v4i_t c)
	{
		if constexpr (G == 3)
			return __builtin_amdgcn_mfma_i32_16x16x32_i8((long)a, (long)b, c, 0, 0, 0);
		else
			return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
	}
	/* y (+)= c * val for the lane's four outputs: |c| < 2^23 (162 x 32768 at most) and val << SHIFT < 2^23, so the 24-bit multiplier is
	 * exact mod 2^32.  One asm statement per four (see mul_idx_val_x4); its inputs are VALU results, never the matrix core's own registers
	 * (the compiler does not look inside asm for the wait states those need). */
	template <bool ACC>
	static __device__ __forceinline__ void scale4(v4i_t &y, const v4i_t c, const int32_t val)
	{
		int32_t y0 = y[0], y1 = y[1], y2 = y[2], y3 = y[3];
		if constexpr (ACC)
			asm("v_mad_i32_i24 %0, %4, %8, %0\n\tv_mad_i32_i24 %1, %5, %8, %1\n\tv_mad_i32_i24 %2, %6, %8, %2\n\tv_mad_i32_i24 %3, %7, %8, %3"
			    : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "s"(val));
		else
			asm("v_mul_i32_i24_e32 %0, %8, %4\n\tv_mul_i32_i24_e32 %1, %8, %5\n\tv_mul_i32_i24_e32 %2, %8, %6\n\tv_mul_i32_i24_e32 %3, %8, %7"
			    : "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3) : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "s"(val));
		y = v4i_t{ y0, y1, y2, y3 };
	}
	/* the matrix core's result moved through the vector ALU once: an asm statement (scale4) must not read the matrix core's registers
	 * directly - the compiler places the wait states those need in front of instructions it knows, not in front of asm.  The zero it adds
	 * is one the compiler cannot see through. */
	static __device__ __forceinline__ v4i_t settle(const v4i_t d)
	{
		uint32_t z;
		asm("v_mov_b32 %0, 0" : "=v"(z));
		v4i_t c;
#pragma unroll
		for (int i = 0; i < 4; i++)
			c[i] = (int32_t)((uint32_t)d[i] + z);
		return c;
	}
	/* low-byte and high-byte products of one unit -> index-weighted sums */
	static __device__ __forceinline__ v4i_t join(const v4i_t d, const v4i_t e)
	{
		v4i_t c;
#pragma unroll
		for (int i = 0; i < 4; i++)
			c[i] = (int32_t)(((uint32_t)e[i] << 8) + (uint32_t)d[i]);
		return c;
	}
	static __device__ __forceinline__ Operand masked(const Operand a, const bool keep)
	{
		if constexpr (G == 3)
			return keep ? a : 0ull;
		else
			return keep ? a : v4i_t{ 0, 0, 0, 0 };
	}
	/* the matrix operand of unit k for rows stored at width class CLS: the low plane (CLS 3: low bytes minus 128; 2: the indices; 1: the
	 * nibbles, i.e. indices plus 8) and, for CLS 3, the high plane */
	static __device__ __forceinline__ Operand plane_lo(const Raw &raw, const int k, const uint32_t cls)
	{
		constexpr uint32_t M = 0x0f0f0f0fu;
		if constexpr (G == 3) {
			const v4u_t r = raw.r[k];
			if (cls == ACMHIP_BP_NIBBLE)
				return ((uint64_t)((r.x >> 4) & M) << 32) | (r.x & M);
			return ((uint64_t)r.y << 32) | r.x;
		} else {
			const v4u_t r = raw.r[2 * k];
			if (cls == ACMHIP_BP_NIBBLE)
				return v4i_t{ (int)(r.x & M), (int)((r.x >> 4) & M), (int)(r.y & M), (int)((r.y >> 4) & M) };
			return (v4i_t)r;
		}
	}
	static __device__ __forceinline__ Operand plane_hi(const Raw &raw, const int k)
	{
		if constexpr (G == 3)
			return ((uint64_t)raw.r[k].w << 32) | raw.r[k].z;
		else
			return (v4i_t)raw.r[2 * k + 1];
	}
	/* what the accumulator must start from for rows of class cls, given what it starts from at 16 bits (128 x the coefficient sums: the low
	 * bytes are stored minus 128): nothing at 8 bits, -8 x the sums at 4 bits (stored plus 8) */
	static __device__ __forceinline__ v4i_t k_of(const uint32_t cls, const v4i_t k16)
	{
		if (cls == ACMHIP_BP_WORD)
			return k16;
		if (cls == ACMHIP_BP_BYTE)
			return v4i_t{ 0, 0, 0, 0 };
		return v4i_t{ -(k16[0] >> 4), -(k16[1] >> 4), -(k16[2] >> 4), -(k16[3] >> 4) };
	}

	/* the lane's four outputs of output tile mt of unit k into the LDS tile.  Output 16 mt + 4 rs + i of a unit is row (16 mt + 4 rs) / QN of
	 * the pair, column q = (16 mt + 4 rs) % QN + i of the lane's residue; o0 already names rs and the residue (run()).  G = 4: one output tile
	 * per row of the pair; G = 3: the one tile is both rows */
	static __device__ __forceinline__ void put_outputs(uint32_t *const o0, const int k, const int mt, const v4i_t y)
	{
		constexpr int PS = C::PS;
		constexpr int per_row = QN / 16 > 0 ? QN / 16 : 1;
		uint32_t *o = o0 + pair_of(k) * (2 * COLS + ((2 * COLS) >> PS)) + (16 * grp_of(k) + ((16 * grp_of(k)) >> PS));
		if (QN >= 16)
			o += (mt / per_row) * (COLS + (COLS >> PS)) + ((mt % per_row) * 16 * SIGMA + (((mt % per_row) * 16 * SIGMA) >> PS));
#pragma unroll
		for (int i = 0; i < 4; i++)
			o[i * SIGMA + ((i * SIGMA) >> PS)] = (uint32_t)y[i];
	}

	/* the units of row pair PP when its four input rows share one val and one width CLS: the multiply moves behind the matrix */
	template <int PP, uint32_t CLS, int... Js>
	static __device__ __forceinline__ void fast_pair(const Raw &raw, uint32_t *const o0, const bool nothing_in_front, const int lane, const bool owns0,
							 const Tables &t, const Pre &pre, const int32_t val, std::integer_sequence<int, Js...>)
	{
		constexpr int PS = C::PS;
		const int rs = lane >> 4;
		const v4i_t zero = { 0, 0, 0, 0 };
		auto unit = [&](auto jj) {
			constexpr int k = PP * UPP + decltype(jj)::value;
			constexpr bool with_bias = grp_of(k) == 0;
#pragma unroll
			for (int mt = 0; mt < NM; mt++) {
				const Operand a = pre.a[mt];
				const v4i_t d = mfma(a, plane_lo(raw, k, CLS), k_of(CLS, pre.kc[mt]));
				const v4i_t c = CLS == ACMHIP_BP_WORD ? join(d, mfma(a, plane_hi(raw, k), zero)) : settle(d);
				v4i_t y = with_bias ? t.bias[nothing_in_front ? 1 : 0][mt][owns0 ? rs : 4] : zero;
				scale4<with_bias>(y, c, val);
				put_outputs(o0, k, mt, y);
			}
		};
		(unit(std::integral_constant<int, Js>{}), ...);
	}

	/* rowval[lr + 2] = val << SHIFT of tile row lr (rows -2, -1 of a stream's first tile: row 0's; the stager put a pair of zeros there) */
	template <int PP, int... Js>
	static __device__ __forceinline__ void run_pair(const Raw &raw, uint32_t *const o0, const int32_t *rv, const bool nothing_in_front, const int lane,
							const bool owns0, const Tables &t, const Pre &pre, const uint32_t clsF, const uint32_t clsC,
							const bool one_val_one_width, const bool halves_uniform, std::integer_sequence<int, Js...>)
	{
		constexpr int PS = C::PS;
		const int rs = lane >> 4;
		const v4i_t zero = { 0, 0, 0, 0 };
		if (one_val_one_width) {
			/* one val over all four rows, one width (the usual case): the width is decided once per pair, its units are straight-line code */
			const int32_t vc = __builtin_amdgcn_readfirstlane(rv[2]);
			by_class(clsC, [&](auto cc) {
				fast_pair<PP, decltype(cc)::value>(raw, o0, nothing_in_front, lane, owns0, t, pre, vc, std::integer_sequence<int, Js...>{});
			});
			return;
		}
		const int32_t va = __builtin_amdgcn_readfirstlane(rv[0]), vb = __builtin_amdgcn_readfirstlane(rv[1]);      /* rows 2P-2, 2P-1 */
		const int32_t vc = __builtin_amdgcn_readfirstlane(rv[2]), vd = __builtin_amdgcn_readfirstlane(rv[3]);      /* rows 2P, 2P+1 */
		/* the "+1" only reaches the lane that owns residue 0, in the unit of group 0 (the pair's first unit, if this wave has it) */
		auto bias = [&](int mt) { return t.bias[nothing_in_front ? 1 : 0][mt][owns0 ? rs : 4]; };
		auto store = [&](int k, int mt, const v4i_t y) { put_outputs(o0, k, mt, y); };
		/* one matrix pass over the operand's low plane and, at 16 bits, its high plane */
		auto product = [&](const Operand a, const int k, const uint32_t cls, const v4i_t k0) -> v4i_t {
			const v4i_t d = mfma(a, plane_lo(raw, k, cls), k0);
			if (cls == ACMHIP_BP_WORD)
				return join(d, mfma(a, plane_hi(raw, k), zero));
			return settle(d);
		};
		if (halves_uniform) {
			/* a block boundary between the pair and the rows in front of it (another val, maybe another width): the matrix once per half,
			 * the other half's coefficients masked out of A - so each pass may read every lane's bytes at ITS half's width */
			auto unit = [&](auto jj) {
				constexpr int k = PP * UPP + decltype(jj)::value;
				constexpr bool with_bias = grp_of(k) == 0;
#pragma unroll
				for (int mt = 0; mt < NM; mt++) {
					const Operand a = pre.a[mt];
					v4i_t c1, c2;
					by_class(clsF, [&](auto cc) { constexpr uint32_t CLS = decltype(cc)::value; c1 = product(masked(a, rs < 2), k, CLS, k_of(CLS, t.khalf[0][mt][rs])); });
					by_class(clsC, [&](auto cc) { constexpr uint32_t CLS = decltype(cc)::value; c2 = product(masked(a, rs >= 2), k, CLS, k_of(CLS, t.khalf[1][mt][rs])); });
					v4i_t y = with_bias ? bias(mt) : zero;
					scale4<with_bias>(y, c1, va);
					scale4<true>(y, c2, vc);
					store(k, mt, y);
				}
			};
			(unit(std::integral_constant<int, Js>{}), ...);
		} else {
			/* anything else (odd acm_rows): once per input row */
			auto unit = [&](auto jj) {
				constexpr int k = PP * UPP + decltype(jj)::value;
				constexpr bool with_bias = grp_of(k) == 0;
#pragma unroll 1
				for (int mt = 0; mt < NM; mt++) {
					const Operand a = t.a[mt][lane];         /* (mt is a run-time value in this loop) */
					v4i_t y = with_bias ? bias(mt) : zero;
#pragma unroll 1
					for (int r = 0; r < 4; r++) {
						v4i_t c;
						by_class(r < 2 ? clsF : clsC, [&](auto cc) {
							constexpr uint32_t CLS = decltype(cc)::value;
							c = product(masked(a, rs == r), k, CLS, k_of(CLS, t.krow[r][mt][rs]));
						});
						scale4<true>(y, c, r == 0 ? va : (r == 1 ? vb : (r == 2 ? vc : vd)));
					}
					store(k, mt, y);
				}
			};
			(unit(std::integral_constant<int, Js>{}), ...);
		}
	}
	template <int... PPs>
	static __device__ __forceinline__ void run_pairs(const Raw &raw, uint32_t *const o0, const int32_t *rv0, const bool missing, const int lane, const bool owns0,
							 const Tables &t, const Pre &pre, const Desc &d, std::integer_sequence<int, PPs...>)
	{
		/* which path each row pair takes, decided for all of them at once with one lane per pair: lane j looks at the four row values of
		 * pair j and at whether its width differs from the width in front of it; two ballots later a pair's decision is one scalar bit test
		 * (instead of four readfirstlanes and a dozen scalar compare-and-select instructions per pair) */
		uint32_t width_change = 0;
#pragma unroll
		for (int j = 0; j < NPW; j++)
			width_change |= (((d.e[j] ^ d.e[j + 1]) & 3u) != 0 ? 1u : 0u) << j;
		const int j = lane < NPW ? lane : 0;
		const int32_t a = rv0[2 * j], b = rv0[2 * j + 1], c = rv0[2 * j + 2], e = rv0[2 * j + 3];
		const uint32_t within = (uint32_t)((a ^ b) | (c ^ e)), between = (uint32_t)(b ^ c) | ((width_change >> j) & 1u);
		const uint32_t fast_mask = (uint32_t)__builtin_amdgcn_ballot_w64((within | between) == 0);
		const uint32_t halves_mask = (uint32_t)__builtin_amdgcn_ballot_w64(within == 0);
		(run_pair<PPs>(raw, o0, rv0 + 2 * PPs, missing && PPs == 0, lane, owns0, t, pre, d.e[PPs] & 3u, d.e[PPs + 1] & 3u, (fast_mask >> PPs) & 1u,
			       (halves_mask >> PPs) & 1u, std::make_integer_sequence<int, UPP>{}), ...);
	}
	static __device__ __forceinline__ void run(const Raw &raw, uint32_t *tile, const int32_t *rowval, const bool fresh_stream, const int tid, const Tables &t,
						   const Desc &d, const uint32_t = 0u)
	{
		const int lane = tid & 63, rs = lane >> 4, n = lane & 15;
		const int p0 = pair0(tid), g0 = grp0(tid);
		const bool owns0 = (n == 0) && (g0 == 0);                       /* residue 0 sits in group 0 */
		/* LDS place of output 4*rs of the lane's residue in the wave's first unit: outputs 4 rs .. of a 16-output tile are columns
		 * q = 4 rs .. (+ 16 per further tile of the same row); G = 3: rs >= 2 is the pair's second row (q = 4 (rs - 2) ..) */
		constexpr int PS = C::PS;
		const int q_lane = (4 * rs) % QN, row_lane = (4 * rs) / QN;
		uint32_t *const o0 = tile + lds_at<PS>(2 * p0 * COLS + g0 * 16) + n + row_lane * (COLS + (COLS >> PS)) + (q_lane * SIGMA + ((q_lane * SIGMA) >> PS));
		const bool missing = fresh_stream && __builtin_amdgcn_readfirstlane(p0) == 0;     /* the wave's first row pair has nothing in front of it */
		Pre pre;
#pragma unroll
		for (int mt = 0; mt < NM; mt++) {
			pre.a[mt] = t.a[mt][lane];
			pre.kc[mt] = t.kc[mt][rs];
		}
		run_pairs(raw, o0, rowval + 2 * p0, missing, lane, owns0, t, pre, d, std::make_integer_sequence<int, NPW>{});
	}
};

#include "acm_toeplitz_tables.inc"

/*
 * The same six-stage first pass for rows that no longer fit a wavefront (levels 13 and 14: 128 / 256 residue classes per row), as a first
 * pass of acm_tile2: the tile is ONE ROW PAIR in the workgroup's LDS (level 13: 512 threads, two workgroups per CU; level 14: 1024), each
 * wavefront takes sixteen classes - one matrix "set" per row - of both rows, and keeps the two rows in front of them in its registers
 * from the tile before (the same wavefront had the same classes there).  Behind it acm_tile2's LDS passes with their barriers, two
 * (level 13) or three (level 14) of them instead of the three / four behind the four-stage FirstPassM.
 */
template <class C_>
struct FirstPassZW {
	using C = C_;
	static constexpr int L = C::L, NT = C::NT, COLS = C::COLS, TR = C::TR, PS = C::PS;
	static constexpr int G = 6, QN = 1 << G, SIGMA = COLS / QN, NWAVE = NT / 64;
	static constexpr int NGW = SIGMA / 16 / NWAVE;          /* groups of sixteen classes per wavefront */
	static constexpr int NSW = TR, NX = TR + 2, NM = QN / 16, NE = 2;      /* the pair in front and the tile's own pair */
	/* a tile is a row pair - or ONE row (round 6: smaller tiles, more workgroups per CU out of phase with each other), which is then the
	 * first or the second row of its pair (ACM_TILE_ODD) with one row of the stream in front (ACM_TILE_ROW1), none (ACM_TILE_FRESH) or two */
	static_assert((TR == 2 || TR == 1) && NGW >= 1 && NGW * 16 * NWAVE == SIGMA && SIGMA >= 32, "a row or a row pair per tile, whole groups per wavefront");
	static constexpr int VARIANT = StageKind<L, G - 1>::N ? 0 : 1;
	static constexpr bool SIGNED_ROWVAL = false, KEEPS_HISTORY = true;
	static constexpr uint32_t CB = 16;
	static __device__ __forceinline__ uint32_t group_at(const uint32_t g) { return 8u * (g & 1u) + 32u * (g >> 1); }
	static __device__ __forceinline__ uint32_t class_of(const uint32_t i) { return (i & 3u) + CB * ((i >> 2) & 1u) + 4u * (i >> 3); }

	struct Raw { v4u_t lo[NGW][NX], hi[NGW][NX]; };
	struct Desc { uint32_t e[NE]; };
	struct Tables {
		v4i_t coef[3][NM][64];
		int32_t bias[3][QN];            /* [rows in front that exist: 0, 1, 2 and more][q], scaled; for the lane that owns residue 0 */
	};
	static __device__ __forceinline__ void fill_tables(Tables &t, const int tid)
	{
		constexpr int32_t ONE = 1 << OutScale<L>::SHIFT;
		for (int k = tid; k < 3 * NM * 64; k += NT) {
			const int j = k / (NM * 64), mt = (k / 64) % NM, lane = k % 64;
			t.coef[j][mt][lane] = *reinterpret_cast<const v4i_t *>(&ACM_TZ6[VARIANT][j][16 * mt + (lane & 15)][16 * (lane >> 4)]);
		}
		for (int k = tid; k < 3 * QN; k += NT)
			t.bias[k / QN][k % QN] = ACM_TZ6_BIAS[VARIANT][k / QN][k % QN] * ONE;
	}
	static __device__ __forceinline__ uint32_t lane_offset(const int) { return 0u; }
	static __device__ __forceinline__ bool fresh_lane(const int) { return false; }
	static __device__ __forceinline__ Desc fetch_desc(const uint32_t *__restrict__ pairs, const AcmTile2 &r, const int)
	{
		Desc d;
		const uint32_t at = __builtin_amdgcn_readfirstlane((uint32_t)r.idx_off);
#pragma unroll
		for (int j = 0; j < NE; j++)
			d.e[j] = pairs[at + j];
		return d;
	}
	static __device__ __forceinline__ int32_t opaque_s(int32_t v)
	{
		asm("" : "+s"(v));
		return v;
	}
	static __device__ __forceinline__ int32_t opaque_v(int32_t v)
	{
		asm("" : "+v"(v));
		return v;
	}
	/* rows k = 0, 1: the pair in front (d.e[0]); 2, 3: the tile's own (d.e[1]).  A tile of one row that is the second row of its pair
	 * (odd): rows k = 0 .. 2 are rows 1, 2, 3 of those four */
	template <bool KEEP>
	static __device__ __forceinline__ void issue_rows(Raw &raw, const uint8_t *arena, const Desc &d, const int tid, const uint32_t odd)
	{
		const uint32_t lane = (uint32_t)tid & 63u, i = lane & 15u, ks = lane >> 4;
		const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6) * (uint32_t)NGW;
		const uint32_t c = class_of(i);
		const uint32_t e0 = d.e[0];
		const uint8_t *base = sgpr_u64(reinterpret_cast<uint64_t>(arena) + ((uint64_t)(e0 >> 2) << 6));
#pragma unroll
		for (int k = KEEP ? 2 : 0; k < NX; k++) {
			const uint32_t u = (TR & 1 ? (uint32_t)__builtin_amdgcn_readfirstlane(odd) : 0u) + (uint32_t)k;        /* (wave-uniform: says so to the compiler) */
			const uint32_t e = (u >> 1) ? (uint32_t)opaque_s((int32_t)d.e[1]) : (uint32_t)opaque_s((int32_t)d.e[0]);
			const uint32_t sh = (e & 3u) == ACMHIP_BP_BYTE ? 0u : 1u;       /* 0: a byte per index, 1: two (both 16-bit classes) */
			const uint32_t row_at = (((e >> 2) - (e0 >> 2)) << 6) + ((u & 1u) ? (uint32_t)COLS << sh : 0u) + 16u * ks;
#pragma unroll
			for (int g = 0; g < NGW; g++) {
				const uint32_t v = row_at + (((group_at(g0 + (uint32_t)g) + c) * (uint32_t)QN) << sh);
				asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(raw.lo[g][k]) : "v"(v), "s"(base) : "memory");
				asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=v"(raw.hi[g][k]) : "v"(v), "s"(base) : "memory");
			}
		}
	}
	/* the first tile of a run: every row, the two in front included (a stream's first tile finds the stager's pair of zeros there) */
	static __device__ __forceinline__ void issue_all(Raw &raw, const int16_t *idx, const Desc &d, const int tid, const uint32_t flags)
	{
		issue_rows<false>(raw, reinterpret_cast<const uint8_t *>(idx), d, tid, (flags & ACM_TILE_ODD) ? 1u : 0u);
	}
	/* every other tile (called behind run() of the tile before): that tile's own rows are this one's rows in front - unless this one
	 * starts a stream (zeros).  (A window's lead-in record - not the successor of the tile before - finds another stream's rows there:
	 * its output is dropped, and the planner puts a second lead-in tile behind it: acmk_tile2m_lead_in) */
	static __device__ __forceinline__ void issue(Raw &raw, const int16_t *idx, const AcmTile2 &r, const Desc &d, const int tid, const uint32_t, const uint32_t)
	{
		const v4u_t z = { 0, 0, 0, 0 };
		const bool fresh = (r.flags & ACM_TILE_FRESH) != 0;
#pragma unroll
		for (int g = 0; g < NGW; g++)
#pragma unroll
			for (int k = 0; k < 2; k++) {
				raw.lo[g][k] = fresh ? z : raw.lo[g][k + NSW];
				raw.hi[g][k] = fresh ? z : raw.hi[g][k + NSW];
			}
		issue_rows<true>(raw, reinterpret_cast<const uint8_t *>(idx), d, tid, (r.flags & ACM_TILE_ODD) ? 1u : 0u);
	}
	static __device__ __forceinline__ v4i_t mfma(const v4u_t data, const v4i_t coef, const v4i_t acc)
	{
		return __builtin_amdgcn_mfma_i32_16x16x64_i8((v4i_t)data, coef, acc, 0, 0, 0);
	}

	/* rowval[k] = val << SHIFT of tile row k - 2 (FirstPassZ::run_t, further down, has the algebra: one accumulator chain per output row, the block
	 * boundaries as multiply-adds of its partial sums) */
	/* in_front: rows of the stream in front of the tile: 0, 1 (tiles of one row only), or 2 for "two or more" */
	/* WHOLE (round 6): some row in reach is of the whole-range class - its low bytes are stored minus 128 (acm_pack.cpp).  What the
	 * matrices make of the missing 128s is 128 x their row sums, per source row and scaled by that row's val like everything else of the
	 * row: whole_rows bit k says so for row k, and the row sums come from the matrix cores themselves (the coefficients times a constant
	 * operand of 64s, twice) - no table, the workgroup's LDS is spoken for to the last byte at level 13 */
	template <bool WORDS, bool WHOLE = false>
	static __device__ __forceinline__ void run_t(const Raw &raw, uint32_t *const tile, const int32_t *rowval, const uint32_t in_front, const int tid, const Tables &t,
						     const uint32_t whole_rows = 0u)
	{
		const v4i_t zero = { 0, 0, 0, 0 };
		const uint32_t lane = (uint32_t)tid & 63u, h = lane >> 4, qd = lane & 15u;
		const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6) * (uint32_t)NGW;
		int32_t rv[NX];
#pragma unroll
		for (int k = 0; k < NX; k++)
			rv[k] = __builtin_amdgcn_readfirstlane(rowval[k]);
		int32_t cw[NX];                 /* 2 val of the rows that miss their 128s, 0 for the others (scalar) */
#pragma unroll
		for (int k = 0; k < NX; k++)
			cw[k] = WHOLE && ((whole_rows >> k) & 1u) ? 2 * rv[k] : 0;
		int32_t val[NSW], dv2[NSW], dv1[NSW];
		bool step2[NSW], step1[NSW];
#pragma unroll
		for (int s = 0; s < NSW; s++) {
			val[s] = rv[s + 2];
			dv2[s] = rv[s + 1] - rv[s + 2];
			dv1[s] = rv[s] - rv[s + 1];
			step2[s] = dv2[s] != 0;
			step1[s] = dv1[s] != 0;
		}
		const v4i_t *cf = &t.coef[0][0][lane];
#pragma unroll 1
		for (int mt = 0; mt < NM; mt++) {
			const v4i_t cf0 = cf[0], cf1 = cf[NM * 64], cf2 = cf[2 * NM * 64];
			int32_t r64[3] = { 0, 0, 0 };           /* 64 x the row sums of T0, T1, T2 for this lane's output q (every instance the same) */
			if constexpr (WHOLE) {
				const v4u_t c64 = { 0x40404040u, 0x40404040u, 0x40404040u, 0x40404040u };
				r64[0] = mfma(c64, cf0, zero)[0];
				r64[1] = mfma(c64, cf1, zero)[0];
				r64[2] = mfma(c64, cf2, zero)[0];
			}
#pragma unroll
			for (int g = 0; g < NGW; g++) {
				const uint32_t c0 = group_at(g0 + (uint32_t)g) + class_of(4u * h);     /* the lane's four outputs: classes c0 .. c0 + 3 */
#pragma unroll
				for (int s = 0; s < NSW; s++) {
					const uint32_t var = in_front + (uint32_t)s < 2u ? in_front + (uint32_t)s : 2u;        /* rows of the stream in front of this one: 0, 1, two or more */
					int32_t b = c0 == 0 ? (&t.bias[0][0])[var * QN + qd + 16u * (uint32_t)mt] : 0;
					int32_t bw = 0;
					if constexpr (WHOLE)    /* rows s, s + 1, s + 2 go through T2, T1, T0 (operands below 2^24: val << SHIFT < 2^20, 64 x 1822 < 2^17) */
						bw = __mul24(r64[2], cw[s]) + __mul24(r64[1], cw[s + 1]) + __mul24(r64[0], cw[s + 2]);
					const v4i_t l1 = mfma(raw.lo[g][s], cf2, zero);
					const v4i_t l2 = mfma(raw.lo[g][s + 1], cf1, l1);
					const v4i_t la = mfma(raw.lo[g][s + 2], cf0, l2);
					v4i_t y;
#pragma unroll
					for (int v = 0; v < 4; v++)
						y[v] = __mul24(la[v], val[s]) + (v == 0 ? b : 0) + bw;
					if (step2[s]) {
#pragma unroll
						for (int v = 0; v < 4; v++)
							y[v] += __mul24(l2[v], dv2[s]);
					}
					if (step1[s]) {
#pragma unroll
						for (int v = 0; v < 4; v++)
							y[v] += __mul24(l1[v], dv1[s]);
					}
					if constexpr (WORDS) {
						const v4i_t h1 = mfma(raw.hi[g][s], cf2, zero);
						const v4i_t h2 = mfma(raw.hi[g][s + 1], cf1, h1);
						const v4i_t ha = mfma(raw.hi[g][s + 2], cf0, h2);
						/* (level 13's sixteen wavefronts per CU have 128 registers each, and the second path costs the one that spills) */
						if (L != 13 && (uint32_t)val[s] < 65536u && !step2[s] && !step1[s]) {
							/* (one v_mad_u32_u24 per output: FirstPassZ::run_t says why it is exact) */
							const uint32_t v8 = (uint32_t)opaque_v((int32_t)((uint32_t)val[s] << 8));
#pragma unroll
							for (int v = 0; v < 4; v++)
								y[v] = (int32_t)(__umul24((uint32_t)ha[v], v8) + (uint32_t)y[v]);
						} else {
						/* (opaque copies: FirstPassZ::run_t says why) */
						const int32_t wv = opaque_v(val[s]), w2 = opaque_v(dv2[s]), w1 = opaque_v(dv1[s]);
						v4i_t yh;
#pragma unroll
						for (int v = 0; v < 4; v++)
							yh[v] = __mul24(ha[v], wv);
						if (step2[s]) {
#pragma unroll
							for (int v = 0; v < 4; v++)
								yh[v] += __mul24(h2[v], w2);
						}
						if (step1[s]) {
#pragma unroll
							for (int v = 0; v < 4; v++)
								yh[v] += __mul24(h1[v], w1);
						}
#pragma unroll
						for (int v = 0; v < 4; v++)
							y[v] = (int32_t)(((uint32_t)opaque_v(yh[v]) << 8) + (uint32_t)y[v]);
						}
					}
					const uint32_t m = (uint32_t)(s * COLS) + c0 + (uint32_t)SIGMA * (qd + 16u * (uint32_t)mt);
					uint32_t *const o = tile + (m + (m >> PS));
#pragma unroll
					for (int v = 0; v < 4; v++)
						o[v] = (uint32_t)y[v];
				}
			}
			cf += 64;
		}
	}
	static __device__ __forceinline__ void run(Raw &raw, uint32_t *const tile, const int32_t *rowval, const bool, const int tid, const Tables &t, const Desc &d,
						   const uint32_t flags)
	{
		const uint32_t in_front = (flags & ACM_TILE_FRESH) ? 0u : (flags & ACM_TILE_ROW1) ? 1u : 2u;
		const uint32_t odd = (TR & 1) && ((uint32_t)__builtin_amdgcn_readfirstlane(flags) & ACM_TILE_ODD) ? 1u : 0u;
		uint32_t any_word = 0;
#pragma unroll
		for (int j = 0; j < NE; j++)
			any_word |= (d.e[j] & 3u) != ACMHIP_BP_BYTE ? 1u : 0u;
		if (any_word) {
			/* a pair at 8 bits has no high bytes: what was loaded in their place is its neighbour's low ones (in place: the rows that stay in
			 * their registers for the next tile stay what they are) */
			uint32_t whole_rows = 0;
#pragma unroll
			for (int k = 0; k < NX; k++) {
				const uint32_t ek = ((odd + (uint32_t)k) >> 1) ? (uint32_t)opaque_s((int32_t)d.e[1]) : (uint32_t)opaque_s((int32_t)d.e[0]);
				const uint32_t mask = (ek & 3u) != ACMHIP_BP_BYTE ? 0xFFFFFFFFu : 0u;
				whole_rows |= (ek & 3u) == ACMHIP_BP_WORDU ? 1u << k : 0u;
#pragma unroll
				for (int g = 0; g < NGW; g++)
					raw.hi[g][k] &= mask;
			}
			/* (a pair in front of the stream, or rows of a lead-in that do not exist, are never of that class: zeros are written at 8 bits) */
			if (__builtin_expect(whole_rows == 0u, 1))
				run_t<true>(raw, tile, rowval, in_front, tid, t);
			else
				run_t<true, true>(raw, tile, rowval, in_front, tid, t, whole_rows);
		} else {
			run_t<false>(raw, tile, rowval, in_front, tid, t);
		}
	}
};

/*
 * Vector memory in acm_tile2 is issued and waited for by hand.  Left to the compiler, the first use of a prefetched
 * index waits with vmcnt(0) - for the PCM stores of the previous tile as well (it cannot prove how many stores are
 * behind the loads once a lead-in tile may have skipped them), and the row-value fetch is waited for right where it
 * is issued: one full memory latency per tile with every other wave parked at the next barrier.  Here the loads are
 * inline asm, so the compiler's counter bookkeeping never sees them, and ONE wait at the end of the iteration that
 * issued them names exactly the number of younger operations (that tile's PCM stores) it may leave in flight.
 * The loaded registers are first read in the NEXT iteration, behind that wait and a barrier.  What the compiler must
 * not do is copy or spill such a register between its load and the wait (it would copy the old content);
 * tests/test_isa_invariants.py checks the generated code for that.
 * The counted wait relies on what gfx9 / CDNA guarantee: the vector memory operations of a wave return (and decrement vmcnt)
 * in the order they were issued, loads and stores alike.  Targets with separate load and store counters or out-of-order
 * returns (gfx10 on) need another wait; this file is written for gfx950 and refuses anything else.
 */
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "acm_kernels.hip: gfx950 (MI355X) only - the hand-counted s_waitcnt vmcnt(N) of acm_tile2 assumes in-order vector memory returns"
#endif

template <int YOUNGER>
__device__ __forceinline__ void k2_wait()
{
	static_assert(YOUNGER >= 0 && YOUNGER < 64, "vmcnt is a 6-bit field");
	asm volatile("s_waitcnt vmcnt(%0)" :: "n"(YOUNGER) : "memory");
}

/* MFORM: idx is the byte-plane staged form and the first pass runs on the matrix cores (FirstPassM); everything else is the same */
template <class C, int WPS, int ABL, bool MFORM, int G0, int... Gs>
__global__ void __launch_bounds__(C::NT, WPS)
acm_tile2(const AcmTile2 *__restrict__ tiles, const uint32_t ntiles, const int16_t *__restrict__ idx, const uint32_t *__restrict__ pairs,
	  const acmhip_blkhdr *__restrict__ hdr, int16_t *__restrict__ pcm, int16_t *__restrict__ sink, const unsigned fmt)
{
	constexpr int L = C::L, NT = C::NT, COLS = C::COLS, NELEM = C::NELEM, TR = C::TR, NJ_LAST = C::NJ_LAST;
	using FP = std::conditional_t<MFORM, std::conditional_t<G0 == 6, FirstPassZW<C>, FirstPassM<C, (G0 == 4 ? 4 : 3)>>, FirstPass2<C, G0, 2, ABL>>;
	static_assert(!MFORM || G0 == 3 || G0 == 4 || G0 == 6, "coefficient tables exist for a first pass of three, four or six stages");
	constexpr bool NEG_ODD_ROWS = FP::SIGNED_ROWVAL && StageKind<L, 0>::N;
	static_assert(TR + 2 <= NT, "one row value per thread");
	constexpr bool PRIO = WPS * 256 / NT > 1;               /* several workgroups per CU: see phase_prio */

	__shared__ uint32_t tile_mem[8 + NELEM + (NELEM >> C::PS)];     /* no guard zone: segment 0 always reads the carry */
	__shared__ int32_t rowval[2][TR + 2];
	constexpr int NCARRY_WORDS = carry_total<C, G0, Gs...>();
	__shared__ uint32_t carry_mem[NCARRY_WORDS];
	__shared__ typename FP::Tables fp_tables;
	uint32_t *const tile = tile_mem + 8;

	const int tid = threadIdx.x;
	FP::fill_tables(fp_tables, tid);                /* read behind the first barrier of the tile loop */
	const uint32_t per = (ntiles + gridDim.x - 1) / gridDim.x;
	uint32_t t = blockIdx.x * per;
	const uint32_t t_end = t + per < ntiles ? t + per : ntiles;
	if (t >= t_end)
		return;
	/* a run that starts inside a stream first replays the tile in front of it without storing PCM - unless the table says so itself: a
	 * window into a stream (rows from row_begin on) starts with a record of the tile in front of it, marked ACM_TILE_DISCARD */
	bool discard = false;
	{
		const uint32_t f0 = tiles[__builtin_amdgcn_readfirstlane(t)].flags;
		if (f0 & ACM_TILE_DISCARD) {
			discard = true;
		} else if (!(f0 & ACM_TILE_FRESH)) {
			discard = true;
			t--;
		}
	}

	const uint32_t voff = FP::lane_offset(tid);
	const uint32_t seg0 = FP::fresh_lane(tid) ? 0xFFFFFFFFu : 0u;

	/* row values of one tile: thread lr < TR + 2 fetches the val of tile row lr - 2 (decode.c:589).  Every lane of every
	 * wave issues the load (lanes beyond the tile repeat its last row) so that all waves count the same vector-memory
	 * operations; what is loaded is only looked at in finish_val, a whole tile later */
	const uint32_t lr_fetch = (uint32_t)(tid < TR + 2 ? tid : TR + 1);
	auto fetch_val = [&](const AcmTile2 &r) -> uint32_t {
		/* rows in front of the stream do not exist (they weigh 0): the record counts from the stream's row 0 then (a tile of one row that is
		 * row 1 of its stream, ACM_TILE_ROW1, has one such row) */
		const uint32_t missing = (r.flags & ACM_TILE_FRESH) ? 2u : (r.flags & ACM_TILE_ROW1) ? 1u : 0u;
		const uint32_t q = r.rowpos + (lr_fetch < missing ? 0u : lr_fetch - missing);           /* rows counted from the record's reference row */
		const uint32_t b = r.magic ? __umulhi(q, r.magic) : q;     /* q / acm_rows (magic = ceil(2^32 / rows), 0 for rows == 1) */
		const uint32_t *p = &hdr[r.hdr_blk + b].val;
		uint32_t v;
		if (ABL & 1)
			v = b;
		else
			asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
		return v;
	};
	auto finish_val = [&](uint32_t v, const AcmTile2 &r) -> int32_t {
		if (FP::SIGNED_ROWVAL && (r.flags & ACM_TILE_FRESH) && tid < 2)
			v = 0;
		v <<= OutScale<L>::SHIFT;
		return (NEG_ODD_ROWS && (tid & 1)) ? -(int32_t)v : (int32_t)v;
	};
	auto warm_off = [&](const AcmTile2 &r) -> uint32_t {
		return voff + (seg0 & ((r.flags & ACM_TILE_FRESH) ? (uint32_t)(2 * COLS * 2) : 0u));
	};
	auto load_tile = [&](typename FP::Raw &raw, const AcmTile2 &r, const typename FP::Desc &d) {
		FP::issue(raw, idx, r, d, tid, voff, warm_off(r));
	};

	constexpr int NVEC = TR * COLS / 8, PER_OWNER = NJ_LAST / 8, NSTORE = NVEC / NT;      /* 16-byte PCM stores per thread and tile */
	static_assert(NVEC % NT == 0, "whole rounds");
	/* tile records through the scalar cache (the index is wave-uniform; readfirstlane says so to the compiler):
	 * a vector load here would be tracked by the compiler's vmcnt bookkeeping, which knows nothing of the asm loads */
	AcmTile2 cur = tiles[__builtin_amdgcn_readfirstlane(t)];
	typename FP::Desc dcur = FP::fetch_desc(pairs, cur, tid);        /* byte-plane form: where the tile's row pairs are and how wide */
	typename FP::Raw raw;
	uint32_t hv = fetch_val(cur);
	if constexpr (FP::KEEPS_HISTORY)
		FP::issue_all(raw, idx, dcur, tid, cur.flags);  /* (the rows in front of a tile are the registers of the tile before: not at a run's start) */
	else
		load_tile(raw, cur, dcur);
	k2_wait<0>();                                   /* first tile of the run: nothing to hide the latency behind yet */
	int buf = 0;
	bool fresh = true;              /* the first tile of a run starts from zero carries (stream start or lead-in) */
#ifdef ACM_STAMPS
	unsigned long long acc_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
	unsigned long long last_ = stamp_now();
#endif

	/* tile records come through the scalar cache one iteration ahead (asked for behind the PCM stores, used after the
	 * next first pass); the last tile of a run names itself as its successor */
	AcmTile2 nxt = tiles[__builtin_amdgcn_readfirstlane(t + 1 < t_end ? t + 1 : t)];
	typename FP::Desc dnxt = FP::fetch_desc(pairs, nxt, tid);
	for (;;) {
		const uint32_t tn = t + 1;
		const bool more = tn < t_end;
		if (fresh)
			for (int k = tid; k < NCARRY_WORDS; k += NT)
				carry_mem[k] = 0u;
		if (tid < TR + 2)
			rowval[buf][tid] = finish_val(hv, cur);  /* fetched (and waited for) while the previous tile was in the LDS passes */
		__syncthreads();
		ACM_STAMP(0);
		/* history in front of the stream is zeros: no "+1" there (decode.c:561-564 runs on existing rows only) */
		phase_prio<PRIO, PRIO_FIRST_PASS>();
		FP::run(raw, tile, rowval[buf], (cur.flags & ACM_TILE_FRESH) != 0, tid, fp_tables, dcur, cur.flags);
		phase_prio<PRIO, PRIO_IDLE>();
		ACM_STAMP(1);

		hv = fetch_val(nxt);                            /* the last tile of a run fetches its own again: no branch around the loads */
		ACM_STAMP(2);

		/* (spreading these loads over the LDS passes instead of issuing them in one burst was measured: no gain) */
		load_tile(raw, nxt, dnxt);
		phase_prio<PRIO, PRIO_LDS_PASSES>();            /* until the PCM stores are issued */
		if (!(ABL & 8))
			run_lds_passes<C, ABL, true, G0, Gs...>(tile, tid, fmt, carry_mem);
		ACM_STAMP(3);
		__syncthreads();
		ACM_STAMP(4);

		{
			/* a lead-in tile stores too - into a sink nobody reads - so that every iteration issues the same vector
			 * memory operations and ONE counted wait serves them all */
			typedef uint32_t v4u __attribute__((ext_vector_type(4)));
			v4u *out = discard ? reinterpret_cast<v4u *>(sink) : reinterpret_cast<v4u *>(reinterpret_cast<uint16_t *>(pcm) + cur.pcm_off);
#pragma unroll
			for (int k = 0; k < NVEC / NT; k++) {
				const int vec = tid + k * NT;
				const uint32_t *q = tile + lds_at<C::PS>((vec / PER_OWNER) * NJ_LAST) + (vec % PER_OWNER) * park_piece<NJ_LAST>();
				const v4u o = { q[0], q[1], q[2], q[3] };
				if ((ABL & 16) && o.x != 0x12345u)              /* timing-only build: no stores */
					continue;
				__builtin_nontemporal_store(o, &out[vec]);      /* PCM is written once and never read back here (+1 % over plain stores;
										 * sc1 / sc0 sc1 / sc1 nt: -0.8 ... +0.3 %, profiles/r3_level9_experiments.txt) */
			}
		}
		ACM_STAMP(5);
		/* the next tile's staged indices were requested from inside the LDS passes; behind them only this tile's PCM
		 * stores may still be on their way.  (Waiting here, inside the iteration that issued the loads, keeps the loaded
		 * registers out of any copy the compiler places on the loop's back edge.) */
		phase_prio<PRIO, PRIO_IDLE>();
		k2_wait<NSTORE>();
		ACM_STAMP(6);
		if (!more)
			break;
		fresh = (nxt.flags & (ACM_TILE_FRESH | ACM_TILE_DISCARD)) != 0;          /* (a window's lead-in starts from zero carries like a run's) */
		discard = (nxt.flags & ACM_TILE_DISCARD) != 0;
		cur = nxt;
		dcur = dnxt;
		t = tn;
		nxt = tiles[__builtin_amdgcn_readfirstlane(t + 1 < t_end ? t + 1 : t)];
		dnxt = FP::fetch_desc(pairs, nxt, tid);
		buf ^= 1;
	}
#ifdef ACM_STAMPS
	if ((tid & 63) == 0 && blockIdx.x < 2048 / 4)
		for (int k = 0; k < 8; k++)
			g_acm_stamps[blockIdx.x * 4 + (tid >> 6) % 4][k] = acc_[k];
#endif
}

// ---------------------------------------------------------------------------
// K3: the chunk kernel - six stages on the matrix cores, one wavefront per chunk, no workgroup barrier
// ---------------------------------------------------------------------------
/*
 * What bounded acm_tile2 with its first three stages on the matrix cores (profiles/r4_level9_summary.txt) was no longer instruction issue
 * and not yet memory: a third of the wave time parked at workgroup barriers and waits, the LDS pipe busy 57 % of the launch.  Both come from
 * the LDS passes.  This kernel removes one of them and every barrier:
 *
 * Six stages at once.  Over one residue class of the columns (columns c + q * cols / 64, q < 64) the first six stages of juggle_block
 * (decode.c:528-577) are a block-Toeplitz operator over ROWS: out[r] = T0 x[r] + T1 x[r - 1] + T2 x[r - 2], three 64 x 64 integer matrices
 * with |coefficient| <= 64 (tools/gen_mfma_tables.py; the stages' signs repeat with the row and their reach, 126 positions, stays inside two
 * rows).  val is constant over a block, so as in acm_tile2's matrix build the multiply moves behind the matrix and the operands are the
 * staged INDICES as signed bytes: v_mfma_i32_16x16x64_i8 with A = indices (16 instances = residue classes x row walkers; lane l: instance
 * l % 16, columns q = 16 (l / 16) ...), B = coefficients (lane l: output q = 16 mt + l % 16) and D = four consecutive instances of one
 * output per lane - four ADJACENT COLUMNS, which go to the LDS tile without a bank conflict.  A 16-bit index is idx = 256 hi + lo with
 * BOTH bytes signed (the byte-plane form of this kernel's levels, acm_pack.cpp): two matrix passes, joined behind the multiply.
 *
 * One wavefront per chunk.  What is left is ONE LDS pass (level 9: stages 6-8, stride <= 4), and with 32 consecutive elements per lane
 * its walks never leave the 2048 elements a wavefront has just produced: a chunk of 2048 samples (4 rows at level 9) is produced, finished
 * and stored by ONE wavefront.  The LDS operations of a wavefront are carried out in issue order, so nothing in the loop needs a barrier;
 * the sixteen wavefronts of a workgroup share nothing but the coefficient tables (12 KB of LDS, which is why the workgroup is the CU) and
 * drift apart by themselves, which is what the phase priorities want.  The first pass is stateless (it re-reads the two rows in front of
 * its walk, like FirstPassM); the last pass takes its 16-element history from the previous chunk through a carry buffer of the
 * wavefront's own, as every LDS pass of acm_tile2 does.
 */
/* (acm_toeplitz_tables.inc - ACM_TZ6, ACM_TZ6_BIAS - is included in front of acm_tile2, whose level-13 / 14 first pass reads it too) */

template <int L_>
struct FirstPassZ {
	/* what ONE wavefront holds in LDS: 2048 elements (32 per lane in the last pass), or one row where a row is longer (levels 12 and 13:
	 * 4096 / 8192 elements, eight / four wavefronts per workgroup with twice / four times the registers each) */
	static constexpr int NELEM = (1 << L_) > 2048 ? (1 << L_) : 2048;
	using C = TileCfg<L_, 64, NELEM>;
	static constexpr int NW = 32768 / NELEM;                /* wavefronts per workgroup = per CU */
	static constexpr int L = L_, COLS = C::COLS, TR = C::TR, PS = C::PS;
	static constexpr int G = 6, QN = 1 << G, SIGMA = COLS / QN;            /* QN columns of a residue class per row, SIGMA classes */
	static_assert(SIGMA >= 2 && TR >= 1, "sixteen instances per matrix instruction: sixteen classes of one row, or all the classes of 2 / 4 / 8 rows");
	static constexpr int RR = SIGMA >= 16 ? 1 : 16 / SIGMA; /* row walkers among the sixteen instances */
	static constexpr int NG = SIGMA >= 16 ? SIGMA / 16 : 1; /* groups of sixteen classes per row */
	static constexpr int NSW = TR / RR;                     /* rows a walker walks */
	static constexpr int NX = NSW + 2;                      /* rows a lane loads per group: its walk and the two rows in front of it */
	static constexpr int NSET = NSW * NG;                   /* matrix "sets" (sixteen instances x 64 outputs) per chunk: 2048 / 1024 */
	static constexpr int NE = (((TR & 1) + (RR - 1) * NSW + NX - 1) >> 1) + 1;      /* pair-table entries a chunk may read, from the pair in front of it on
										 * (a chunk of one row may start on the second row of a pair) */
	static constexpr int NM = QN / 16;                      /* output tiles per row */
	static_assert(NSW * RR == TR && NSET * 1024 == NELEM, "a chunk is whole sets");
	static constexpr int VARIANT = StageKind<L, G - 1>::N ? 0 : 1;          /* a P stage leaves odd positions negated */
	/*
	 * Which class instance i of a set stands for.  A lane receives four consecutive instances of one output (instances 4 h .. 4 h + 3,
	 * h = lane / 16, output q = 16 mt + lane % 16) and stores them as four adjacent columns; a store instruction serves lanes 0-31
	 * (h = 0, 1) and 32-63 in turn, and column c of output q sits at c + SIGMA q + (that >> 5): on bank c + 8 q + q / 4 (SIGMA 8),
	 * c + 16 q + q / 2 (16), c + q (32).  So that the 32 lanes land on 32 banks, h = 1 must sit 4 / 8 / 16 columns beyond h = 0:
	 * SIGMA 8: classes 4 (h & 1) + v of walker h >> 1; SIGMA >= 16: classes v + CB (h & 1) + 4 (h >> 1), CB = 8 or 16, and with
	 * 32 classes and more group g of sixteen is the same pattern 8 (g & 1) + 32 (g >> 1) further on.
	 */
	static constexpr uint32_t CB = SIGMA >= 32 ? 16 : 8;
	static constexpr uint32_t group_at(const int g) { return 8u * (uint32_t)(g & 1) + 32u * (uint32_t)(g >> 1); }
	static_assert(NG <= 8 && (NG == 1 || SIGMA >= 32), "groups of sixteen classes");
	/* the four instances a lane receives: one walker's four adjacent classes - or, with two classes per row (level 7), two walkers' */
	static constexpr int NH = SIGMA >= 4 ? 1 : 4 / SIGMA, NVH = 4 / NH;
	static __device__ __forceinline__ uint32_t class_of(const uint32_t i)
	{
		if constexpr (RR > 1)
			return i % SIGMA;
		else
			return (i & 3u) + CB * ((i >> 2) & 1u) + 4u * (i >> 3);
	}

	struct Raw { v4u_t lo[NG][NX], hi[NG][NX]; };
	struct Desc { uint32_t e[NE]; };
	struct Tables {
		v4i_t coef[3][NM][64];          /* [j][mt][lane]: T_j[16 mt + lane % 16][16 (lane / 16) .. + 15] */
		int32_t bias[3][2][QN];         /* [rows in front that exist][lane owns residue 0: 0, else 1][q], scaled */
		int32_t rsum[3][QN];            /* [j][q]: 128 x the sum of row q of T_j - what a row stored with its low bytes minus 128
						 * (ACMHIP_BP_WORDU) is short of, per unit of val */
	};

	static __device__ __forceinline__ void fill_tables(Tables &t, const int tid, const int nthreads)
	{
		constexpr int32_t ONE = 1 << OutScale<L>::SHIFT;
		for (int k = tid; k < 3 * NM * 64; k += nthreads) {
			const int j = k / (NM * 64), mt = (k / 64) % NM, lane = k % 64;
			t.coef[j][mt][lane] = *reinterpret_cast<const v4i_t *>(&ACM_TZ6[VARIANT][j][16 * mt + (lane & 15)][16 * (lane >> 4)]);
		}
		for (int k = tid; k < 3 * 2 * QN; k += nthreads) {
			const int var = k / (2 * QN), par = (k / QN) & 1, q = k % QN;
			t.bias[var][par][q] = par ? 0 : ACM_TZ6_BIAS[VARIANT][var][q] * ONE;
		}
		for (int k = tid; k < 3 * QN; k += nthreads) {
			int32_t sum = 0;
			for (int c = 0; c < 64; c++)
				sum += ACM_TZ6[VARIANT][k / QN][k % QN][c];
			t.rsum[k / QN][k % QN] = 128 * sum;
		}
	}

	static __device__ __forceinline__ Desc fetch_desc(const uint32_t *__restrict__ pairs, const AcmTile2 &r)
	{
		Desc d;
		const uint32_t at = __builtin_amdgcn_readfirstlane((uint32_t)r.idx_off);        /* the entry of the pair in front of the chunk's first pair */
#pragma unroll
		for (int j = 0; j < NE; j++)
			d.e[j] = pairs[at + j];
		return d;
	}

	/* the pair-table entry of row `u` counted from the first row of the pair in front (u = 0, 1: d.e[0]; 2, 3: d.e[1]; ...) */
	static __device__ __forceinline__ uint32_t entry_of(const Desc &d, const uint32_t u)
	{
		/* (opaque copies: a chain of selects over the members of a struct is otherwise turned into an indexed load from a copy of the
		 * struct in scratch memory) */
		uint32_t e = (uint32_t)opaque_s((int32_t)d.e[0]);
#pragma unroll
		for (int j = 1; j < NE; j++)
			e = (u >> 1) >= (uint32_t)j ? (uint32_t)opaque_s((int32_t)d.e[j]) : e;
		return e;
	}

	/* the loads of one chunk: lane l = (instance i = l % 16: class and walker; columns 16 (l / 16) .. + 15 of the class) asks for 16 low bytes
	 * and, 64 bytes on, 16 high bytes of each of its rows (a pair at 8 bits has no high bytes: what comes back instead is never
	 * used, see run_t()).  odd: the chunk starts on the second row of a pair (chunks of one row only) */
	/* KEEP (one walker only: the rows in front of a lane's walk are rows the SAME lane loaded for the chunk in front): only the walk's own
	 * rows are asked for, the two rows in front stay in their registers (shift_history) - a third (level 11) or half of the requests */
	static constexpr bool HISTORY_IN_REGISTERS = RR == 1;
	/* two walkers of two rows each (level 9): the rows in front of walker 1 are walker 0's own rows of the same chunk, the rows in front of
	 * walker 0 are walker 1's own rows of the chunk before - both sit eight lanes away in the same row of sixteen, one v_mov_b32_dpp per
	 * register instead of a second request for bytes the partner lane has just loaded (half the load instructions of a chunk) */
#ifndef ACM_K3_DPP_HISTORY
#define ACM_K3_DPP_HISTORY 1
#endif
	/* (round 6: the same with four walkers of two rows each, level 8 - walker w's rows in front are walker w - 1's own, SIGMA lanes down
	 * the row of sixteen; walker 0's are the last walker's of the chunk before, 16 - SIGMA lanes up) */
	static constexpr bool HISTORY_BY_DPP = ACM_K3_DPP_HISTORY && RR >= 2 && NSW == 2 && RR * SIGMA == 16 && SIGMA >= 4;
	static constexpr bool KEEPS_ROWS = HISTORY_IN_REGISTERS || HISTORY_BY_DPP;
	template <int CTRL, int BANKS>
	static __device__ __forceinline__ v4u_t dpp4(const v4u_t old, const v4u_t src)
	{
		v4u_t r;
#pragma unroll
		for (int v = 0; v < 4; v++)
			r[v] = (uint32_t)__builtin_amdgcn_update_dpp((int)old[v], (int)src[v], CTRL, 0xF, BANKS, false);
		return r;
	}
	/* DPP controls: row_shl:N = 0x100 + N, row_shr:N = 0x110 + N; a bank is four lanes of a row of sixteen */
	static constexpr int DPP_W0_BANKS = (1 << (SIGMA / 4)) - 1;            /* the lanes of walker 0: instances 0 .. SIGMA - 1 */
	/* behind run(): the lanes of walker 0 take the LAST walker's own rows - the chunk's last two - as the rows in front of the next chunk */
	static __device__ __forceinline__ void hand_history_on(Raw &raw)
	{
#pragma unroll
		for (int k = 0; k < 2; k++) {
			raw.lo[0][k] = dpp4<0x100 + (16 - SIGMA), DPP_W0_BANKS>(raw.lo[0][k], raw.lo[0][k + NSW]);    /* row_shl:(16 - SIGMA) into walker 0's lanes of every row of sixteen */
			raw.hi[0][k] = dpp4<0x100 + (16 - SIGMA), DPP_W0_BANKS>(raw.hi[0][k], raw.hi[0][k + NSW]);
		}
	}
	/* in front of run(): the lanes of walkers 1 .. take the own rows of the walker below them in this chunk */
	static __device__ __forceinline__ void take_history_from_partner(Raw &raw)
	{
#pragma unroll
		for (int k = 0; k < 2; k++) {
			raw.lo[0][k] = dpp4<0x110 + SIGMA, 0xF & ~DPP_W0_BANKS>(raw.lo[0][k], raw.lo[0][k + NSW]);     /* row_shr:SIGMA into the other walkers' lanes */
			raw.hi[0][k] = dpp4<0x110 + SIGMA, 0xF & ~DPP_W0_BANKS>(raw.hi[0][k], raw.hi[0][k + NSW]);
		}
	}
	/*
	 * The fast path (round 6).  profiles/ubench/issue_model.hip: a SIMD of this chip issues ONE instruction at a time whatever its kind, and
	 * the chunk loop is bound by that (r6 notes in DESIGN.md: ~860 instructions per chunk, 190 of them scalar) - so what a chunk can know
	 * as a SCALAR it should not work out per lane.  Two facts make most chunks simple:
	 *   - every row in reach lies in one block (ACM_TILE_ONEBLOCK, set by the planner from the geometry alone), so one val - a scalar
	 *     load of the block header, asked for two chunks ahead with the chunk's pair-table entries - scales everything: no per-row
	 *     values, no differences, no per-set branches;
	 *   - the row pairs in reach are stored at ONE width and back to back (what every stager writes inside a block; checked here on the
	 *     entries themselves, which are scalars already), so a row's place is a scalar base plus ONE lane offset that does not change
	 *     from chunk to chunk, nothing is masked, and a chunk of 8-bit pairs does not ask for high bytes at all.
	 * mode: 0 = the general path, 1 = fast, pairs at 8 bits, 2 = fast, pairs at 16 bits.
	 */
	static constexpr int JM = (TR & 1) ? 1 : TR / 2;         /* pair-table entries 0 .. JM hold the rows in reach of a chunk */
	static_assert(JM + 1 <= NE, "entries in reach are entries fetched");
	/* HALF-bytes per index of a width class: ACMHIP_BP_WORDU (0) 4, _NIB12 (1) 3, _BYTE (2) 2, _WORD (3) 4; a row of a pair takes COLS / 2 times that many
	 * bytes, the QN indices of a residue class QN / 2 times (their low bytes first, the high bytes - or nibbles - 64 bytes on) */
	static __device__ __forceinline__ uint32_t half_bytes(const uint32_t cls) { return (0x4234u >> (4u * cls)) & 15u; }
	/* J0: the first entry whose rows the chunk's loads ask for (1 where the rows in front stay in registers).
	 * Returns the mode: 0 general; else fast with every pair in reach at 8 (1), 16 (2) or 12 bits (3) */
	template <int J0>
	static __device__ __forceinline__ uint32_t mode_of(const uint32_t flags, const Desc &d, const uint32_t sval)
	{
		const uint32_t cls = d.e[0] & 3u;
		const uint32_t pair_units = (uint32_t)(COLS / 64) * half_bytes(cls);   /* 64-byte units a pair of this width takes */
		bool ok = (flags & ACM_TILE_ONEBLOCK) != 0 && cls != 0u;
		if constexpr (OutScale<L>::SHIFT != 0)
			ok = ok && sval < 65536u;               /* (the one-instruction join of the high plane, see run_t) */
#pragma unroll
		for (int j = 1; j <= JM; j++)
			ok = ok && (d.e[j] & 3u) == cls;
#pragma unroll
		for (int j = J0; j < JM; j++)
			ok = ok && (d.e[j + 1] >> 2) - (d.e[j] >> 2) == pair_units;
		return ok ? (cls == ACMHIP_BP_NIB12 ? 3u : cls - (ACMHIP_BP_BYTE - 1u)) : 0u;
	}

#ifndef ACM_K3_LD_MOD
#define ACM_K3_LD_MOD " nt"     /* the staged bytes are read once: non-temporal.  A/B in one process, every build on its own staging (profiles/ab_kernels.py
				 * --own-form): +1.8 % on a box whose allocation runs the launch in its slow mode (1.5221 -> 1.4957 ms), +0.3 % on a fast one;
				 * sc1 / sc0 sc1: -0.7 ... -0.8 %, sc0: 0, sc1 nt / sc0 sc1 nt: as nt (profiles/r6_level9_notes.txt) */
#endif
	template <bool KEEP>
	static __device__ __forceinline__ void issue(Raw &raw, const uint8_t *arena, const Desc &d, const int lane, const uint32_t odd, const uint32_t mode)
	{
		static_assert(!KEEP || KEEPS_ROWS, "the rows in front belong to another lane");
		constexpr int K0 = KEEP ? 2 : 0, J0 = KEEP ? 1 : 0;
		const uint32_t i = (uint32_t)lane & 15u, ks = (uint32_t)lane >> 4;
		const uint32_t c = class_of(i), rr = RR == 1 ? 0u : i / SIGMA;
		const uint32_t smode = __builtin_amdgcn_readfirstlane(mode);   /* (wave-uniform by construction; says so to the compiler) */
		/* where a lane's sixteen bytes of (row k, group g) are: a lane offset - one for the low bytes, one for what is 64 bytes behind them:
		 * sixteen high bytes, or eight bytes of high nibbles, which sit 8 ks and not 16 ks into their part (and are asked for as sixteen
		 * bytes like everything else: what comes with them is dropped, see expand_nib) - from a base the wavefront shares */
		uint32_t va[NX][NG], vh[NX][NG];
		const uint8_t *sb[NX][NG];
		if (mode) {
			/* one width, back to back: row u (counted from the first row of entry J0's pair) starts u rows of this width on */
			const uint32_t hb = smode == 1u ? 2u : smode == 2u ? 4u : 3u;
			const uint64_t base = reinterpret_cast<uint64_t>(arena) + ((uint64_t)(d.e[J0] >> 2) << 6);
			const uint32_t vl = (rr * (uint32_t)(NSW * COLS / 2) + c * (uint32_t)(QN / 2)) * hb + 16u * ks;
			const uint32_t vlh = vl - (smode == 3u ? 8u * ks : 0u);
#pragma unroll
			for (int k = K0; k < NX; k++)
#pragma unroll
				for (int g = 0; g < NG; g++) {
					va[k][g] = vl;
					vh[k][g] = vlh;
					sb[k][g] = sgpr_u64(base + (uint64_t)((((TR & 1 ? odd : 0u) + (uint32_t)(k - K0)) * (uint32_t)(COLS / 2) + group_at(g) * (uint32_t)(QN / 2)) * hb));
				}
		} else {
			const uint32_t e0 = d.e[0];
			const uint8_t *base = sgpr_u64(reinterpret_cast<uint64_t>(arena) + ((uint64_t)(e0 >> 2) << 6));
#pragma unroll
			for (int k = K0; k < NX; k++) {
				const uint32_t u = (TR & 1 ? odd : 0u) + rr * NSW + k;          /* row 0 = the first row of the pair in front */
				const uint32_t e = entry_of(d, u);
				const uint32_t hb = half_bytes(e & 3u);
				const uint32_t row_at = (((e >> 2) - (e0 >> 2)) << 6) + ((u & 1u) ? (uint32_t)(COLS / 2) * hb : 0u) + 16u * ks;
#pragma unroll
				for (int g = 0; g < NG; g++) {
					va[k][g] = row_at + (group_at(g) + c) * (uint32_t)(QN / 2) * hb;
					vh[k][g] = va[k][g] - ((e & 3u) == ACMHIP_BP_NIB12 ? 8u * ks : 0u);
					sb[k][g] = base;
				}
			}
		}
		/* ONE place that asks for a register, whatever the path: two load sequences joined by the compiler would be joined with copies of
		 * registers whose loads are still in flight (tests/test_isa_invariants.py).  The high bytes of a chunk of 8-bit pairs (mode 1)
		 * are not asked for: the skip is a scalar branch inside the statement, the register keeps what it had (never looked at: the
		 * fast path at 8 bits runs no high plane, the general path masks the rows that have none) */
#pragma unroll
		for (int k = K0; k < NX; k++)
#pragma unroll
			for (int g = 0; g < NG; g++) {
				/* (non-temporal only where every staged byte is read ONCE: where the rows in front of a walk are re-read by the next
				 * walker - level 8 - the second reader wants them in the cache: 0.70 -> 0.60 with nt there) */
				if constexpr (KEEPS_ROWS) {
					asm volatile("global_load_dwordx4 %0, %1, %2" ACM_K3_LD_MOD : "=v"(raw.lo[g][k]) : "v"(va[k][g]), "s"(sb[k][g]) : "memory");
					asm volatile("s_cmp_eq_u32 %3, 1\n\ts_cbranch_scc1 .Lacm_z8_%=\n\t"
						     "global_load_dwordx4 %0, %1, %2 offset:64" ACM_K3_LD_MOD "\n"
						     ".Lacm_z8_%=:"
						     : "+v"(raw.hi[g][k]) : "v"(vh[k][g]), "s"(sb[k][g]), "s"(smode) : "memory", "scc");
				} else {
					asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(raw.lo[g][k]) : "v"(va[k][g]), "s"(sb[k][g]) : "memory");
					asm volatile("s_cmp_eq_u32 %3, 1\n\ts_cbranch_scc1 .Lacm_z8_%=\n\t"
						     "global_load_dwordx4 %0, %1, %2 offset:64\n"
						     ".Lacm_z8_%=:"
						     : "+v"(raw.hi[g][k]) : "v"(vh[k][g]), "s"(sb[k][g]), "s"(smode) : "memory", "scc");
				}
			}
	}
	/*
	 * The 12-bit class in registers.  What the load of a row's "high bytes" brings for a 12-bit pair is eight bytes of high NIBBLES (dwords
	 * 0 and 1; dwords 2 and 3 are the next lane's): element 8 d + b of the lane's sixteen in the high nibble of byte b of dword d, element
	 * 8 d + 4 + b in the low one (acm_pack.cpp put_row_nib12).  expand: the sixteen operand bytes hi << 4 - signed bytes as they stand,
	 * the matrix pass of such rows is scaled by val << 4 instead of val << 8.  This "N form" is what a 12-bit row's registers hold from
	 * then on (history handed from chunk to chunk included); the general path, which runs ONE high plane over rows of any width, takes
	 * such rows to true bytes first (sra4) and the rows that stay for the next chunk back afterwards (shl4).
	 */
	static __device__ __forceinline__ v4u_t expand_nib(const v4u_t x)
	{
		const uint32_t M = 0xF0F0F0F0u;
		return v4u_t{ x[0] & M, (x[0] << 4) & M, x[1] & M, (x[1] << 4) & M };
	}
	static __device__ __forceinline__ v4u_t sra4(const v4u_t x)    /* every byte arithmetically shifted right by four */
	{
		v4u_t r;
#pragma unroll
		for (int v = 0; v < 4; v++) {
			const uint32_t m = x[v] & 0x80808080u;          /* sign bits: 0xF0 per negative byte = (m << 1) - (m >> 3), whole-word arithmetic */
			r[v] = ((x[v] >> 4) & 0x0F0F0F0Fu) | ((m << 1) - (m >> 3));
		}
		return r;
	}
	static __device__ __forceinline__ v4u_t shl4(const v4u_t x)
	{
		return v4u_t{ (x[0] << 4) & 0xF0F0F0F0u, (x[1] << 4) & 0xF0F0F0F0u, (x[2] << 4) & 0xF0F0F0F0u, (x[3] << 4) & 0xF0F0F0F0u };
	}
	/* what the chunk behind this one finds in front of its walk: the last two rows of this one's (the registers of rows 2 .. are about to
	 * be asked for again).  A stream's first chunk has nothing in front of it: zeros, as the stager's pair of zeros would have been */
	static __device__ __forceinline__ void shift_history(Raw &raw)
	{
#pragma unroll
		for (int g = 0; g < NG; g++)
#pragma unroll
			for (int k = 0; k < 2; k++) {
				raw.lo[g][k] = raw.lo[g][k + NSW];
				raw.hi[g][k] = raw.hi[g][k + NSW];
			}
	}
	static __device__ __forceinline__ void zero_history(Raw &raw)
	{
		const v4u_t z = { 0, 0, 0, 0 };
#pragma unroll
		for (int g = 0; g < NG; g++)
#pragma unroll
			for (int k = 0; k < 2; k++)
				raw.lo[g][k] = raw.hi[g][k] = z;
	}

	static __device__ __forceinline__ v4i_t mfma(const v4u_t data, const v4i_t coef, const v4i_t acc)
	{
		return __builtin_amdgcn_mfma_i32_16x16x64_i8((v4i_t)data, coef, acc, 0, 0, 0);
	}

	/* v * c with both inside 24 bits.  (The library's __mul24 is plain arithmetic to the optimiser, which then turns val * hi * 256 + val * lo
	 * into ONE multiply of val with a sum that no longer fits 24 bits - the quarter-rate v_mul_lo_u32; the opaque copy of val keeps
	 * the two products apart.) */
	static __device__ __forceinline__ int32_t opaque_s(int32_t v)
	{
		asm("" : "+s"(v));
		return v;
	}
	static __device__ __forceinline__ int32_t opaque_v(int32_t v)
	{
		asm("" : "+v"(v));
		return v;
	}

	/*
	 * rowval[k] = val << SHIFT of chunk row k - 2 (k < TR + 2), as scalars.  in_front: rows of the stream in front of the chunk (0, 1, or 2
	 * for "two or more").
	 * WORDS: some pair of the chunk is stored at 16 bits (run() has put zeros where a row of this lane is not: it has no high bytes, and what
	 *        was loaded in their place is its neighbour's low ones).
	 * The three terms of an output row r are one accumulator chain, oldest row first: P1 = T2 x[r-2], P2 = P1 + T1 x[r-1], A = P2 + T0 x[r].
	 * With one val over the rows in reach (the usual case) the output is val A.  Across a block boundary
	 *     val[r] T0 x[r] + val[r-1] T1 x[r-1] + val[r-2] T2 x[r-2] = val[r] A + (val[r-1] - val[r]) P2 + (val[r-2] - val[r-1]) P1,
	 * so a chunk that sees a change of val keeps the partial sums of the chain and pays one multiply-add per plane for each of the two
	 * differences that is not zero in SOME lane of the set (step2 / step1, decided per set from the scalar row values) - no matrix
	 * instruction more, and nothing for the chunks in the middle of a block.
	 */
	/* hvs: lane k holds val << SHIFT of chunk row k - 2 (k < TR + 2).
	 * FAST (mode_of): one val, below 2^16 as scaled, over every row in reach - the scalar sval; hvs is not looked at.  No differences,
	 * no branches per set, the multipliers are scalar operands: per output ONE instruction per plane. */
	/* NIB (FAST only): the high plane's rows are 12-bit rows in N form (expand_nib): scaled by val << 4, a signed multiply */
	/* nform (general path): every row in reach that has a high plane is a 12-bit row, in N form (hi << 4): the plane joins shifted by 4, not 8.
	 * wordu (general path): some row in reach is stored with its low bytes minus 128 (ACMHIP_BP_WORDU: idx = 256 hi + (lo + 128)): such a row's
	 * share of an output is short of 128 x val x the row sum of its matrix.  Which rows: from the pair-table entries *dd, for the walker
	 * whose OUTPUTS the lane receives */
	template <bool WORDS, bool FAST, bool NIB = false>
	static __device__ __forceinline__ void run_t(const Raw &raw, uint32_t *const tile, const Tables &t, const int lane, const uint32_t hvs,
						     const uint32_t in_front, const uint32_t sval, const bool wordu = false, const Desc *dd = nullptr,
						     const uint32_t odd = 0u, const bool nform = false)
	{
		static_assert(!NIB || (WORDS && FAST), "rows in N form are a fast-path matter; the general path takes them to bytes first");
		const v4i_t zero = { 0, 0, 0, 0 };
		/* output side of the lane: instances 4 (lane / 16) .. + 3 = four adjacent classes of one walker (two classes each of two walkers
		 * where a row has only two), output q = 16 mt + lane % 16 */
		const uint32_t h = (uint32_t)lane >> 4, qd = (uint32_t)lane & 15u;
		uint32_t c0[NH], rrd[NH];
		uint32_t o_lane[NH];            /* (dword offsets into the tile, not pointers: see bias_at) */
#pragma unroll
		for (int hf = 0; hf < NH; hf++) {
			const uint32_t i = 4u * h + (uint32_t)(hf * NVH);
			c0[hf] = RR == 1 ? class_of(i) : i % SIGMA;
			rrd[hf] = RR == 1 ? 0u : i / SIGMA;
			const uint32_t m_lane = rrd[hf] * (uint32_t)(NSW * COLS) + c0[hf] + (uint32_t)SIGMA * qd;
			o_lane[hf] = m_lane + (m_lane >> PS);
		}
		/* the "+1" of decode.c:561-564, six stages on: the lane that owns residue 0 adds it to that output; rows in front of a stream do
		 * not exist and add nothing */
		uint32_t bias_at[NSW][NH];      /* (an index, not a pointer: a pointer picked at run time loses its address space and the load becomes a flat one) */
		/* per walk row s (and walker of the lane): val of the row, the two differences, and whether any lane of the set has one */
		int32_t val[NSW][NH], dv2[NSW][NH], dv1[NSW][NH];
		bool step2[NSW], step1[NSW];
		bool small[NSW];                /* the row's (scaled) val is below 2^16 in every lane: see the one-instruction join of the high plane */
#pragma unroll
		for (int s = 0; s < NSW; s++)
			small[s] = OutScale<L>::SHIFT == 0;
		if constexpr (FAST) {
#pragma unroll
			for (int s = 0; s < NSW; s++) {
				step2[s] = step1[s] = false;
				small[s] = true;
#pragma unroll
				for (int hf = 0; hf < NH; hf++)
					val[s][hf] = dv2[s][hf] = dv1[s][hf] = 0;
			}
		} else if constexpr (RR <= 2) {
			int32_t rowval[TR + 2];
#pragma unroll
			for (int k = 0; k < TR + 2; k++)
				rowval[k] = (int32_t)__builtin_amdgcn_readlane(hvs, k);
#pragma unroll
			for (int s = 0; s < NSW; s++) {
				/* chunk row of walker w's output: w * NSW + s; rowval is indexed from row -2 */
				const int32_t a0 = rowval[s + 2], a1 = rowval[s + 1], a2 = rowval[s];
				if constexpr (RR == 1) {
					val[s][0] = a0;
					dv2[s][0] = a1 - a0;
					dv1[s][0] = a2 - a1;
					step2[s] = a1 != a0;
					step1[s] = a2 != a1;
					small[s] = (uint32_t)a0 < 65536u;
				} else {
					const int32_t b0 = rowval[NSW + s + 2], b1 = rowval[NSW + s + 1], b2 = rowval[NSW + s];
					val[s][0] = rrd[0] ? b0 : a0;
					dv2[s][0] = rrd[0] ? b1 - b0 : a1 - a0;
					dv1[s][0] = rrd[0] ? b2 - b1 : a2 - a1;
					step2[s] = a1 != a0 || b1 != b0;
					step1[s] = a2 != a1 || b2 != b1;
					small[s] = (uint32_t)(a0 | b0) < 65536u;
				}
			}
		} else {
			/* four or eight walkers: a lane fetches the values of its walker's rows from the lanes that hold them */
			const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane(hvs);
			const bool one_val = __builtin_amdgcn_ballot_w64(lane < TR + 2 && hvs != first) == 0;
#pragma unroll
			for (int s = 0; s < NSW; s++)
				step2[s] = step1[s] = false;
#pragma unroll
			for (int hf = 0; hf < NH; hf++) {
				int32_t g[NSW + 2];
#pragma unroll
				for (int k = 0; k < NSW + 2; k++)
					g[k] = one_val ? (int32_t)first : __builtin_amdgcn_ds_bpermute((int)(4u * (rrd[hf] * NSW + (uint32_t)k)), (int32_t)hvs);
#pragma unroll
				for (int s = 0; s < NSW; s++) {
					val[s][hf] = g[s + 2];
					dv2[s][hf] = g[s + 1] - g[s + 2];
					dv1[s][hf] = g[s] - g[s + 1];
					if (!one_val) {
						step2[s] = step2[s] || __builtin_amdgcn_ballot_w64(dv2[s][hf] != 0) != 0;
						step1[s] = step1[s] || __builtin_amdgcn_ballot_w64(dv1[s][hf] != 0) != 0;
					}
				}
			}
		}
#pragma unroll
		for (int s = 0; s < NSW; s++)
#pragma unroll
			for (int hf = 0; hf < NH; hf++) {
				const uint32_t row = in_front + rrd[hf] * NSW + s;
				bias_at[s][hf] = ((row < 2 ? row : 2u) * 2u + (c0[hf] == 0 ? 0u : 1u)) * (uint32_t)QN + qd;
			}
		const v4i_t *cf = &t.coef[0][0][lane];
		const auto &hi = raw.hi;
#ifndef ACM_K3_MT_UNROLL
#define ACM_K3_MT_UNROLL 1              /* (unrolled further the loop wants more registers than a wave of four per SIMD has) */
#endif
#define ACM_PRAGMA_(x) _Pragma(#x)
#define ACM_UNROLL(n) ACM_PRAGMA_(unroll n)
		ACM_UNROLL(ACM_K3_MT_UNROLL)
		for (int mt = 0; mt < NM; mt++) {
			const v4i_t cf0 = cf[0], cf1 = cf[NM * 64], cf2 = cf[2 * NM * 64];
#pragma unroll
			for (int e = 0; e < NSET; e++) {
				const int g = e / NSW, s = e % NSW;
				/* rows of this set: x[r] = raw[g][s + 2], x[r - 1] = raw[g][s + 1], x[r - 2] = raw[g][s] */
				int32_t b[NH];
#pragma unroll
				for (int hf = 0; hf < NH; hf++)
					b[hf] = g == 0 ? (&t.bias[0][0][0])[bias_at[s][hf] + 16u * (uint32_t)mt] : 0;   /* (residue 0 is in group 0) */
				const v4i_t l1 = mfma(raw.lo[g][s], cf2, zero);
				const v4i_t l2 = mfma(raw.lo[g][s + 1], cf1, l1);
				const v4i_t la = mfma(raw.lo[g][s + 2], cf0, l2);
				v4i_t y;
				if constexpr (FAST) {
					/* (opaque copies of the scalar multipliers: or the optimiser folds the two planes into one 32-bit multiply, see below) */
					const int32_t sv = opaque_s((int32_t)__builtin_amdgcn_readfirstlane(sval));
#pragma unroll
					for (int v = 0; v < 4; v++)
						y[v] = __mul24(la[v], sv) + (v % NVH == 0 ? b[v / NVH] : 0);
					if constexpr (WORDS) {
						const v4i_t h1 = mfma(hi[g][s], cf2, zero);
						const v4i_t h2 = mfma(hi[g][s + 1], cf1, h1);
						const v4i_t ha = mfma(hi[g][s + 2], cf0, h2);
						if constexpr (NIB) {
							const int32_t sv4 = opaque_s((int32_t)__builtin_amdgcn_readfirstlane(sval << 4));      /* below 2^20: a signed 24-bit operand */
#pragma unroll
							for (int v = 0; v < 4; v++)
								y[v] = __mul24(ha[v], sv4) + y[v];
						} else {
							const uint32_t sv8 = (uint32_t)opaque_s((int32_t)__builtin_amdgcn_readfirstlane(sval << 8));
#pragma unroll
							for (int v = 0; v < 4; v++)
								y[v] = (int32_t)(__umul24((uint32_t)ha[v], sv8) + (uint32_t)y[v]);
						}
					}
				} else {
#pragma unroll
				for (int v = 0; v < 4; v++)
					y[v] = __mul24(la[v], val[s][v / NVH]) + (v % NVH == 0 ? b[v / NVH] : 0);
				if (step2[s]) {
#pragma unroll
					for (int v = 0; v < 4; v++)
						y[v] += __mul24(l2[v], dv2[s][v / NVH]);
				}
				if (step1[s]) {
#pragma unroll
					for (int v = 0; v < 4; v++)
						y[v] += __mul24(l1[v], dv1[s][v / NVH]);
				}
				if constexpr (WORDS && NH == 1) {
					if (wordu) {
						/* rows of the whole-range 16-bit class in reach: 128 x val of the row x the row sum of its matrix, per output q */
						const int32_t *rs = &t.rsum[0][0] + 16 * mt + (int)qd;
						bool urow[3];
#pragma unroll
						for (int j = 0; j < 3; j++)
							urow[j] = (entry_of(*dd, (TR & 1 ? odd : 0u) + rrd[0] * NSW + (uint32_t)(s + j)) & 3u) == ACMHIP_BP_WORDU;
						/* (row values are below 2^22 as scaled, the sums below 2^18: 24-bit multiplies, exact mod 2^32) */
						const int32_t v0 = val[s][0], v1 = v0 + dv2[s][0], v2 = v1 + dv1[s][0];
						const int32_t add = __mul24(urow[2] ? v0 : 0, rs[0]) + __mul24(urow[1] ? v1 : 0, rs[QN]) + __mul24(urow[0] ? v2 : 0, rs[2 * QN]);
#pragma unroll
						for (int v = 0; v < 4; v++)
							y[v] += add;
					}
				}
				if constexpr (WORDS) {
					const v4i_t h1 = mfma(hi[g][s], cf2, zero);
					const v4i_t h2 = mfma(hi[g][s + 1], cf1, h1);
					const v4i_t ha = mfma(hi[g][s + 2], cf0, h2);
					/* With the row's val (as scaled: val << SHIFT) below 2^16 - always at levels 8 and 9, which do not scale; at the others
					 * for val below 2^(level - 4 ... level), which is most material - and no change of val in reach, the high plane
					 * joins in ONE instruction per output: y += ha * (val << 8) as v_mad_u32_u24.  Its operands are the low 24 bits of
					 * each register taken as unsigned: val << 8 is below 2^24, and a negative ha reads as ha + 2^24, which adds
					 * 2^24 * 256 * val = 0 (mod 2^32).  (val << 8 through an opaque copy: or the optimiser folds both planes into one
					 * 32-bit multiply) */
					{
						if (small[s] && !step2[s] && !step1[s] && !nform) {
							uint32_t v8[NH];
#pragma unroll
							for (int hf = 0; hf < NH; hf++)
								v8[hf] = (uint32_t)opaque_v((int32_t)((uint32_t)val[s][hf] << 8));
#pragma unroll
							for (int v = 0; v < 4; v++)
								y[v] = (int32_t)(__umul24((uint32_t)ha[v], v8[v / NVH]) + (uint32_t)y[v]);
							goto joined;
						}
					}
					{
					/* (opaque copies of the multipliers: or the optimiser adds the planes first and multiplies a sum beyond 24 bits at
					 * a quarter of the rate; an opaque sum: or it moves the shift into the multipliers) */
					v4i_t yh;
					int32_t wv[NH], w2[NH], w1[NH];
#pragma unroll
					for (int hf = 0; hf < NH; hf++) {
						wv[hf] = opaque_v(val[s][hf]);
						w2[hf] = opaque_v(dv2[s][hf]);
						w1[hf] = opaque_v(dv1[s][hf]);
					}
#pragma unroll
					for (int v = 0; v < 4; v++)
						yh[v] = __mul24(ha[v], wv[v / NVH]);
					if (step2[s]) {
#pragma unroll
						for (int v = 0; v < 4; v++)
							yh[v] += __mul24(h2[v], w2[v / NVH]);
					}
					if (step1[s]) {
#pragma unroll
						for (int v = 0; v < 4; v++)
							yh[v] += __mul24(h1[v], w1[v / NVH]);
					}
					const uint32_t hs = nform ? 4u : 8u;                    /* (a scalar: one v_lshl_add_u32 either way) */
#pragma unroll
					for (int v = 0; v < 4; v++)
						y[v] = (int32_t)(((uint32_t)opaque_v(yh[v]) << hs) + (uint32_t)y[v]);
					}
				joined:;
				}
				}       /* !FAST */
				/* the constant parts of the address: multiples of 32, or (16 g) small enough to stay inside the lane's group of 32 - the pad rule splits */
				static_assert(COLS % (1 << PS) == 0 && (SIGMA * 16) % (1 << PS) == 0 &&
					      (NG == 1 || (group_at(NG - 1) % (1 << PS) + CB + 7 < (1 << PS) && SIGMA % (1 << PS) == 0)), "address split");
#pragma unroll
				for (int hf = 0; hf < NH; hf++) {
					uint32_t *const o = tile + (o_lane[hf] + (uint32_t)((s * COLS + ((s * COLS) >> PS)) + (group_at(g) + (group_at(g) >> PS))));
#pragma unroll
					for (int v = 0; v < NVH; v++)
						o[v] = (uint32_t)y[hf * NVH + v];
				}
			}
			cf += 64;
#pragma unroll
			for (int hf = 0; hf < NH; hf++)
				o_lane[hf] += SIGMA * 16 + ((SIGMA * 16) >> PS);
		}
	}

	/* first: the chunk's loads asked for every row, the two in front included (a run's first chunk; every chunk where the rows in front are
	 * not kept in registers) - else rows 0 and 1 are history in the form the chunk before left them in */
	static __device__ __forceinline__ void run(Raw &raw, uint32_t *const tile, const Tables &t, const int lane_, const uint32_t hvs,
						   const Desc &d, const uint32_t in_front, const uint32_t odd, const uint32_t mode, const uint32_t sval,
						   const bool first)
	{
		/* what a lane derives from its number (LDS places, table offsets) is worked out again per chunk - a handful of instructions - instead
		 * of living in registers through the LDS passes, which are what the kernel is short of */
#ifdef ACM_K3_OPAQUE_LANE
		const int lane = opaque_v(lane_);
#else
		const int lane = lane_;
#endif
		const bool all_new = !KEEPS_ROWS || first;
		if (mode == 3u) {
			/* twelve bits throughout: the rows just loaded go to N form, that is all */
#pragma unroll
			for (int k = 0; k < NX; k++)
				if (k >= 2 || all_new) {
#pragma unroll
					for (int g = 0; g < NG; g++)
						raw.hi[g][k] = expand_nib(raw.hi[g][k]);
				}
			if constexpr (HISTORY_BY_DPP)
				take_history_from_partner(raw);
			run_t<true, true, true>(raw, tile, t, lane, 0u, in_front, sval);
			return;
		}
		if (mode) {
			if constexpr (HISTORY_BY_DPP)
				take_history_from_partner(raw);         /* (a run's first chunk has loaded these rows itself: the same bytes again) */
			if (mode == 2u)
				run_t<true, true>(raw, tile, t, lane, 0u, in_front, sval);
			else
				run_t<false, true>(raw, tile, t, lane, 0u, in_front, sval);
			return;
		}
		uint32_t any_word = 0, any_nib = 0, any_wordu = 0;
#pragma unroll
		for (int j = 0; j < NE; j++) {
			any_word |= (d.e[j] & 3u) == ACMHIP_BP_WORD ? 1u : 0u;
			any_nib |= (d.e[j] & 3u) == ACMHIP_BP_NIB12 ? 1u : 0u;
			any_wordu |= (d.e[j] & 3u) == ACMHIP_BP_WORDU ? 1u : 0u;
		}
		any_word |= any_wordu;
		const uint32_t rr = RR == 1 ? 0u : ((uint32_t)lane & 15u) / SIGMA;
		if (any_nib) {
			/* the rows just loaded that are 12-bit rows: to N form (before a partner lane takes them as its history) */
#pragma unroll
			for (int k = 0; k < NX; k++)
				if (k >= 2 || all_new) {
					if ((entry_of(d, (TR & 1 ? odd : 0u) + rr * NSW + k) & 3u) == ACMHIP_BP_NIB12) {
#pragma unroll
						for (int g = 0; g < NG; g++)
							raw.hi[g][k] = expand_nib(raw.hi[g][k]);
					}
				}
		}
		if constexpr (HISTORY_BY_DPP)
			take_history_from_partner(raw);                 /* (a run's first chunk has loaded these rows itself: the same bytes again) */
		/* 12-bit rows and no 16-bit row in reach (a block boundary between blocks of pwr 8-10, 8-bit pairs beside 12-bit ones): the high plane
		 * runs on the N form as it stands and joins shifted by 4; only where 12-bit rows meet 16-bit ones are they taken to bytes */
		const bool nform = any_nib && !any_word;
		if (any_word | any_nib) {
			/* in place: the rows that stay in their registers for the next chunk (HISTORY_IN_REGISTERS) stay what they are - 8-bit rows have
			 * no high bytes (zeros); 12-bit rows go from N form to bytes for ONE high plane over rows of any width, and the rows the next
			 * chunk inherits go back behind it */
#pragma unroll
			for (int k = 0; k < NX; k++) {
				const uint32_t cls = entry_of(d, (TR & 1 ? odd : 0u) + rr * NSW + k) & 3u;
				const uint32_t mask = cls == ACMHIP_BP_BYTE ? 0u : 0xFFFFFFFFu;
#pragma unroll
				for (int g = 0; g < NG; g++)
					raw.hi[g][k] &= mask;
				if (any_nib && !nform && cls == ACMHIP_BP_NIB12) {
#pragma unroll
					for (int g = 0; g < NG; g++)
						raw.hi[g][k] = sra4(raw.hi[g][k]);
				}
			}
			run_t<true, false>(raw, tile, t, lane, hvs, in_front, 0u, any_wordu != 0, &d, odd, nform);
			if (any_nib && !nform && KEEPS_ROWS) {
#pragma unroll
				for (int k = NSW; k < NX; k++)
					if ((entry_of(d, (TR & 1 ? odd : 0u) + rr * NSW + k) & 3u) == ACMHIP_BP_NIB12) {
#pragma unroll
						for (int g = 0; g < NG; g++)
							raw.hi[g][k] = shl4(raw.hi[g][k]);
					}
			}
		} else {
			run_t<false, false>(raw, tile, t, lane, hvs, in_front, 0u);
		}
	}
};

/* Gs: the LDS passes behind the six matrix-core stages (they add up to level - 6) */
template <int L_, int ABL, int... Gs>
__global__ void __launch_bounds__(64 * FirstPassZ<L_>::NW, 1)
acm_chunk(const AcmTile2 *__restrict__ tiles, const uint32_t ntiles, const int16_t *__restrict__ idx, const uint32_t *__restrict__ pairs,
	  const acmhip_blkhdr *__restrict__ hdr, int16_t *__restrict__ pcm, int16_t *__restrict__ sink, const unsigned fmt)
{
	using FP = FirstPassZ<L_>;
	using C = typename FP::C;
	constexpr int L = L_, COLS = C::COLS, NELEM = C::NELEM, TR = C::TR, NJ_LAST = C::NJ_LAST, NW = FP::NW;
	constexpr int WTILE = 8 + NELEM + (NELEM >> C::PS);
	constexpr int NCARRY_WORDS = carry_total<C, FP::G, Gs...>();
	constexpr int PASS_ABL = ABL | MODE_WAVE | (FP::NG > 1 && NW == 16 ? MODE_LEAN : 0);

	__shared__ uint32_t tile_mem[NW][WTILE];
	__shared__ uint32_t carry_all[NW][NCARRY_WORDS];
	__shared__ typename FP::Tables tables;

	const int lane0 = threadIdx.x & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	FP::fill_tables(tables, threadIdx.x, NW * 64);
	__syncthreads();                                /* the only one: the coefficient tables are shared */

	uint32_t *const tile = tile_mem[wave] + 8;
	uint32_t *const carry_mem = carry_all[wave];

	/* a wavefront is a workgroup of its own from here on: a contiguous run of the chunk table */
	const uint32_t nvw = gridDim.x * NW, vw = blockIdx.x * NW + wave;
	const uint32_t per = (ntiles + nvw - 1) / nvw;
	uint32_t t = vw * per;
	const uint32_t t_end = t + per < ntiles ? t + per : ntiles;
	if (t >= t_end)
		return;
	bool discard = false;
	{
		const uint32_t f0 = tiles[__builtin_amdgcn_readfirstlane(t)].flags;
		if (f0 & ACM_TILE_DISCARD) {
			discard = true;                 /* the lead-in record of a window into a stream: the table names the chunk in front itself */
		} else if (!(f0 & ACM_TILE_FRESH)) {
			discard = true;                 /* a run that starts inside a stream replays the chunk in front of it without storing PCM */
			t--;
		}
	}

	/* rows of the stream in front of a chunk: 0, 1 (chunks of one row only), or 2 for "two or more" */
	auto rows_in_front = [](const AcmTile2 &r) -> uint32_t { return (r.flags & ACM_TILE_FRESH) ? 0u : (r.flags & ACM_TILE_ROW1) ? 1u : 2u; };
	/* row values: lane lr < TR + 2 fetches the val of chunk row lr - 2 (decode.c:589; the record counts from that row, or from row 0 of the
	 * stream where it does not exist); every lane issues the load.  A chunk on the fast path (mode != 0) has ONE val, a scalar (val_of):
	 * it asks for nothing here - the skip is a scalar branch inside the load statement, so that whichever way it goes there is ONE
	 * statement that writes the register (see FirstPassZ::issue) */
	auto fetch_val = [&](const AcmTile2 &r, const int lane, uint32_t v, const uint32_t mode) -> uint32_t {
		const uint32_t *p = &hdr[0].val;
		if (!mode) {
			const uint32_t lr_fetch = (uint32_t)(lane < TR + 2 ? lane : TR + 1);
			const uint32_t missing = 2u - rows_in_front(r);
			const uint32_t q = r.rowpos + (lr_fetch < missing ? 0u : lr_fetch - missing);
			const uint32_t b = r.magic ? __umulhi(q, r.magic) : q;
			p = &hdr[r.hdr_blk + b].val;
		}
		asm volatile("s_cmp_lg_u32 %2, 0\n\ts_cbranch_scc1 .Lacm_zv_%=\n\t"
			     "global_load_dword %0, %1, off\n"
			     ".Lacm_zv_%=:"
			     : "+v"(v) : "v"(p), "s"(__builtin_amdgcn_readfirstlane(mode)) : "memory", "scc");
		return v;
	};
	/* the val of the block that holds the first row in reach, scaled like every row value: THE val of an ACM_TILE_ONEBLOCK chunk.  Through the
	 * scalar cache, asked for two chunks ahead with the chunk's pair-table entries */
	/* (read through the CONSTANT address space - the block headers are never written while the kernel runs: hdr itself is an operand of
	 * the hand-issued row-value loads, and what a "memory"-clobbering statement has seen the compiler reloads with a VECTOR load,
	 * whose wait it would count without knowing of the hand-issued ones) */
	typedef const uint32_t __attribute__((address_space(4))) *const_u32_ptr;
	auto val_of = [&](const AcmTile2 &r) -> uint32_t {
		const uint64_t a = reinterpret_cast<uint64_t>(&hdr[__builtin_amdgcn_readfirstlane(r.hdr_blk)].val);
		return *(const_u32_ptr)a << OutScale<L>::SHIFT;
	};

	constexpr int NVEC = TR * COLS / 8, PER_OWNER = NJ_LAST / 8, NSTORE = NVEC / 64;
	static_assert(NVEC % 64 == 0, "whole rounds");
	const uint8_t *const arena = reinterpret_cast<const uint8_t *>(idx);
	/* chunk records and pair-table entries come through the scalar cache, TWO chunks ahead: a scalar load is waited for with every LDS
	 * wait behind it (one counter, and scalar loads return in any order), so a record asked for at the end of an iteration and the
	 * entries it names were a chain of two memory latencies in front of every first pass (profiles/ubench/phases_k3.hip: 13 % of a
	 * wavefront's time).  Asked for a whole iteration before they are looked at, they cost nothing.  The last chunks of a run name
	 * themselves as their successors */
	auto record_at = [&](const uint32_t k) -> AcmTile2 { return tiles[__builtin_amdgcn_readfirstlane(k < t_end ? k : t_end - 1)]; };
	/* the path a chunk takes (FirstPassZ::mode_of).  Where the rows in front of a walk stay in registers a chunk's loads start at its own
	 * first pair; elsewhere (level 8: four walkers) at the pair in front */
	constexpr int MODE_J0 = FP::KEEPS_ROWS ? 1 : 0;
#ifdef ACM_K3_NO_FAST
	auto mode_of = [&](const AcmTile2 &, const typename FP::Desc &, const uint32_t) -> uint32_t { return 0u; };      /* (A/B builds: the round-5 path throughout) */
#else
	auto mode_of = [&](const AcmTile2 &r, const typename FP::Desc &d, const uint32_t sval) -> uint32_t {
		return FP::template mode_of<MODE_J0>(r.flags, d, sval);
	};
#endif
	AcmTile2 cur = record_at(t);
	typename FP::Desc dcur = FP::fetch_desc(pairs, cur);
	typename FP::Raw raw;
	{
		const v4u_t z = { 0, 0, 0, 0 };
#pragma unroll
		for (int g = 0; g < FP::NG; g++)
#pragma unroll
			for (int k = 0; k < FP::NX; k++)
				raw.hi[g][k] = z;               /* (the load statement of the high bytes reads the register it may leave alone) */
	}
	/* a run's first chunk asks for all its rows, the ones in front included, and for its row values: the general path */
	uint32_t mode_cur = 0u, sv_cur = val_of(cur);
	uint32_t hv = fetch_val(cur, lane0, 0u, 0u);
	FP::template issue<false>(raw, arena, dcur, lane0, (cur.flags & ACM_TILE_ODD) ? 1u : 0u, 0u);
	k2_wait<0>();
	bool fresh = true, first_of_run = true;
	AcmTile2 nxt = record_at(t + 1), nx2 = record_at(t + 2);
	typename FP::Desc dnxt = FP::fetch_desc(pairs, nxt);
	uint32_t sv_nxt = val_of(nxt);
	uint32_t mode_nxt = mode_of(nxt, dnxt, sv_nxt);
#ifdef ACM_STAMPS
	unsigned long long acc_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
	unsigned long long last_ = stamp_now();
#endif
	for (;;) {
		const uint32_t tn = t + 1;
		const bool more = tn < t_end;
		/* whatever a lane works out from its number is worked out per chunk: registers are what the LDS passes are short of, and a
		 * loop-invariant value would sit in one through all of them */
		int lane = lane0;
#ifdef ACM_K3_OPAQUE_LANE
		asm volatile("" : "+v"(lane));
#endif
		if (fresh)
			for (int k = lane; k < NCARRY_WORDS; k += 64)
				carry_mem[k] = 0u;
		const uint32_t hvs = hv << OutScale<L>::SHIFT;
		ACM_STAMP(0);
		phase_prio<true, PRIO_FIRST_PASS>();
		FP::run(raw, tile, tables, lane, hvs, dcur, rows_in_front(cur), (cur.flags & ACM_TILE_ODD) ? 1u : 0u, mode_cur, sv_cur, first_of_run);
		first_of_run = false;
		phase_prio<true, PRIO_IDLE>();
		ACM_STAMP(1);

		hv = fetch_val(nxt, lane, hv, mode_nxt);              /* the last chunk of a run fetches its own again: no branch around the loads */
		if constexpr (FP::HISTORY_IN_REGISTERS) {
			/* the rows in front of the next chunk's walk are in this lane's registers already - unless that chunk starts a stream
			 * (nothing in front of it: zeros).  (The last chunk of a run, which names itself, is never looked at again.) */
			/* (the first lead-in chunk of a window - ACM_TILE_DISCARD, not the successor of this chunk - finds another stream's rows there:
			 * its output is wrong and dropped, and the planner puts enough lead-in chunks in front of a window that the last of them has
			 * its own stream's rows in front of it, acmk_tile2m_lead_in.  A second load sequence for that case, in a branch, makes the
			 * compiler join the two with copies of registers whose loads are still in flight: tests/test_isa_invariants.py) */
			FP::shift_history(raw);
			if (nxt.flags & ACM_TILE_FRESH)
				FP::zero_history(raw);
			FP::template issue<true>(raw, arena, dnxt, lane, (nxt.flags & ACM_TILE_ODD) ? 1u : 0u, mode_nxt);
		} else if constexpr (FP::HISTORY_BY_DPP) {
			/* (the same with two walkers: walker 0's rows in front come from walker 1's lanes now, walker 1's from walker 0's in front of
			 * the next first pass - FirstPassZ::run.  A window's first lead-in chunk: as above) */
			FP::hand_history_on(raw);
			if (nxt.flags & ACM_TILE_FRESH)
				FP::zero_history(raw);
			FP::template issue<true>(raw, arena, dnxt, lane, (nxt.flags & ACM_TILE_ODD) ? 1u : 0u, mode_nxt);
		} else {
			FP::template issue<false>(raw, arena, dnxt, lane, (nxt.flags & ACM_TILE_ODD) ? 1u : 0u, mode_nxt);
		}
		/* (asked for here, looked at from the end of this iteration on: behind the LDS passes' own waits) */
		const AcmTile2 nx3 = record_at(t + 3);
		const typename FP::Desc dnx2 = FP::fetch_desc(pairs, nx2);
		const uint32_t sv_nx2 = val_of(nx2);
		ACM_STAMP(2);
		phase_prio<true, PRIO_LDS_PASSES>();
		run_lds_passes<C, PASS_ABL, true, FP::G, Gs...>(tile, lane, fmt, carry_mem);
		tile_barrier<MODE_WAVE>();
		ACM_STAMP(3);
		{
			typedef uint32_t v4u __attribute__((ext_vector_type(4)));
			v4u *out = discard ? reinterpret_cast<v4u *>(sink) : reinterpret_cast<v4u *>(reinterpret_cast<uint16_t *>(pcm) + cur.pcm_off);
#pragma unroll
			for (int k = 0; k < NSTORE; k++) {
				const int vec = lane + k * 64;
				const uint32_t *q = tile + lds_at<C::PS>((vec / PER_OWNER) * NJ_LAST) + (vec % PER_OWNER) * park_piece<NJ_LAST>();
				const v4u o = { q[0], q[1], q[2], q[3] };
#ifdef ACM_K3_ST_MOD            /* (A/B builds: another cache policy for the PCM stores - "", " sc1", " sc0 sc1", " sc1 nt" ...; the shipped one is nt) */
				asm volatile("global_store_dwordx4 %0, %1, %2" ACM_K3_ST_MOD :: "v"((uint32_t)(vec * 16)), "v"(o), "s"(sgpr_u64(reinterpret_cast<uint64_t>(out))) : "memory");
#else
				__builtin_nontemporal_store(o, &out[vec]);
#endif
			}
		}
		ACM_STAMP(5);
		phase_prio<true, PRIO_IDLE>();
		k2_wait<NSTORE>();                              /* the next chunk's indices are here; only this chunk's PCM stores may still be on their way */
		ACM_STAMP(6);
		tile_barrier<MODE_WAVE>();                      /* (the next first pass overwrites what the stores have just read) */
		if (!more)
			break;
		fresh = (nxt.flags & (ACM_TILE_FRESH | ACM_TILE_DISCARD)) != 0;
		discard = (nxt.flags & ACM_TILE_DISCARD) != 0;
		cur = nxt;
		dcur = dnxt;
		sv_cur = sv_nxt;
		mode_cur = mode_nxt;
		t = tn;
		nxt = nx2;
		dnxt = dnx2;
		sv_nxt = sv_nx2;
		mode_nxt = mode_of(nxt, dnxt, sv_nxt);
		nx2 = nx3;
	}
#ifdef ACM_STAMPS
	if (lane0 == 0 && vw < 2048)
		for (int k = 0; k < 8; k++)
			g_acm_stamps[vw][k] = acc_[k];
#endif
}

struct Tile2Entry {
	typedef void (*Fn)(const AcmTile2 *, uint32_t, const int16_t *, const uint32_t *, const acmhip_blkhdr *, int16_t *, int16_t *, unsigned);
	Fn fn;
	int threads, tile_rows, wg_per_cu;
};
template <class C, int... Gs>
constexpr Tile2Entry entry_k2()
{
	return Tile2Entry{ acm_tile2<C, 4, 0, false, Gs...>, C::NT, C::TR, 1024 / C::NT };
}
/* bigger tiles: WPC workgroups per CU */
template <class C, int WPC, int... Gs>
constexpr Tile2Entry entry_k2w()
{
	return Tile2Entry{ acm_tile2<C, WPC * C::NT / 256, 0, false, Gs...>, C::NT, C::TR, WPC };
}
#ifdef ACM_ABLATION
/* timing-only builds of the level-9 kernel with parts removed (wrong output by design): ACM_K2_ABL=<mask> */
template <int ABL>
constexpr Tile2Entry abl_k2() { return Tile2Entry{ acm_tile2<TileCfg<9, 256, 8192>, 4, ABL, false, 3, 3, 3>, 256, 16, 4 }; }
const struct { int mask; Tile2Entry e; } g_tile2_abl[] = {
	{ 1, abl_k2<1>() }, { 2, abl_k2<2>() }, { 4, abl_k2<4>() }, { 6, abl_k2<6>() }, { 8, abl_k2<8>() }, { 16, abl_k2<16>() },
	{ 17, abl_k2<17>() }, { 23, abl_k2<23>() }, { 32, abl_k2<32>() }, { 25, abl_k2<25>() }, { 31, abl_k2<31>() }, { 12, abl_k2<12>() },
};
#endif
/* per level the fastest measured geometry and stage grouping (profiles/r2_sweep_levels.txt): 32 KB tiles at four workgroups
 * per CU up to level 10, 64 KB tiles of 512 threads (two workgroups, still four waves per SIMD) above; passes of at most
 * three stages, because with 32-element walks the warm-up of a four-stage pass costs as much as a pass */
/* per level the fastest measured geometry and stage grouping with the phase priorities on (profiles/r2_sweep_levels.txt;
 * alternatives built and timed on one box, all CRC-verified: level 7 (2,2,3) and (2,3,2) -2 %; level 8 (2,3,3), (3,2,3)
 * equal; level 9 as 64 KB tiles of 512 threads -0.4 %; level 10 (2,3,3,2) -1.5 %, (2,2,3,3) -2.5 %, 64 KB tiles -1.5 %;
 * level 11 as 64 KB tiles of 512 threads, two per CU, (2,3,3,3): -2.7 %, which had been the best without priorities) */
const Tile2Entry g_tile2[ACM_K2_MAX_LEVEL - ACM_K2_MIN_LEVEL + 1] = {
	entry_k2<TileCfg<6, 256, 8192>, 2, 2, 2>(),
	entry_k2<TileCfg<7, 256, 8192>, 3, 2, 2>(),
	entry_k2<TileCfg<8, 256, 8192>, 3, 3, 2>(),
	entry_k2<TileCfg<9, 256, 8192>, 3, 3, 3>(),
	entry_k2<TileCfg<10, 256, 8192>, 3, 3, 2, 2>(),
	entry_k2<TileCfg<11, 256, 8192>, 3, 3, 3, 2>(),
	entry_k2w<TileCfg<12, 512, 16384>, 2, 3, 3, 3, 3>(),   /* two 64 KB tiles per CU (127 registers): +19 % over one 128 KB tile, whose waves are all in the same phase */
	/* level 13: four rows are 128 KB - one workgroup of sixteen waves per CU (still four per SIMD), no plane, no prefix sweep:
	 * 4 B of HBM traffic per sample instead of the 12 B of the prefix + plane pair.  A two-stage first pass makes the
	 * whole tile ONE segment (1024 threads x two adjacent columns = the 2048 residues of stride 2048): two warm-up rows per
	 * four rows instead of per two, 24 instead of 32 prefetch registers: +8 % over (3,2,3,3,2) (2.94 against 3.18 ms for
	 * 2.1 Gsamples; at levels 10 and 11 the same trade loses 3 %: an LDS-pass stage costs more than a first-pass stage) */
	entry_k2w<TileCfg<13, 1024, 32768>, 1, ACM_L13_GROUPS>(),
	/* level 14: one row pair is the 128 KB tile (the first pass's body is two rows: the least a tile can be), 155 KB of LDS
	 * with the carries of the first LDS pass (two bodies of stride 256 = 4096 elements) */
	entry_k2w<TileCfg<14, 1024, 32768>, 1, ACM_L14_GROUPS>(),
};
inline const Tile2Entry &tile2_entry(uint32_t level)
{
	return g_tile2[level - ACM_K2_MIN_LEVEL];
}
/* the same geometries with the first pass on the matrix cores (levels whose first pass has three stages) */
template <class C, int... Gs>
constexpr Tile2Entry entry_k2m()
{
	return Tile2Entry{ acm_tile2<C, 4, 0, true, Gs...>, C::NT, C::TR, 1024 / C::NT };
}
template <class C, int WPC, int... Gs>
constexpr Tile2Entry entry_k2mw()
{
	return Tile2Entry{ acm_tile2<C, WPC * C::NT / 256, 0, true, Gs...>, C::NT, C::TR, WPC };
}
#ifndef ACM_K2M_L11
#define ACM_K2M_L11 entry_k2m<TileCfg<11, 256, 8192>, 4, 3, 2, 2>(), 4
#endif
#ifndef ACM_K2M_L12
#define ACM_K2M_L12 entry_k2mw<TileCfg<12, 512, 16384>, 2, 4, 3, 3, 2>(), 4
#endif
#ifndef ACM_K2M_L13
#define ACM_K2M_L13 entry_k2mw<TileCfg<13, 512, 16384>, 2, 4, 3, 3, 3>(), 4
#endif
/* [level][first pass of three / four stages]; the staged form differs between the two (8 or 16 columns of a residue class side by side).
 * The library ships ONE of them per level, the measured better one (profiles/r4_mfma_first_pass.txt, 2.1 Gsamples per level, one box:
 * level 8 equal, level 9 three stages +1.6 %, levels 10 / 11 / 12 four stages +5.5 / +6.6 / +5 %: one LDS pass, or one of its stages, less);
 * the other depth is a tuning build (-DACM_TUNING, chosen per process with ACM_K2M_G0=3|4) */
struct Tile2MEntry { Tile2Entry e; int g0; };
#ifdef ACM_TUNING
#define ACM_K2M_ALT(...) __VA_ARGS__
#else
#define ACM_K2M_ALT(...) Tile2Entry{ nullptr, 0, 0, 0 }, 0
#endif
const Tile2MEntry g_tile2m[ACM_K2M_MAX_LEVEL - ACM_K2M_MIN_LEVEL + 1][2] = {
	{ { entry_k2m<TileCfg<7, 256, 8192>, 3, 2, 2>(), 3 }, { Tile2Entry{ nullptr, 0, 0, 0 }, 0 } },        /* 8 residue classes of stride 8: less than one operand tile */
	{ { entry_k2m<TileCfg<8, 256, 8192>, 3, 3, 2>(), 3 }, { ACM_K2M_ALT(entry_k2m<TileCfg<8, 256, 8192>, 4, 2, 2>(), 4) } },
	{ { entry_k2m<TileCfg<9, 256, 8192>, 3, 3, 3>(), 3 }, { ACM_K2M_ALT(entry_k2m<TileCfg<9, 256, 8192>, 4, 3, 2>(), 4) } },
	{ { ACM_K2M_ALT(entry_k2m<TileCfg<10, 256, 8192>, 3, 3, 2, 2>(), 3) }, { entry_k2m<TileCfg<10, 256, 8192>, 4, 3, 3>(), 4 } },
	{ { ACM_K2M_ALT(entry_k2m<TileCfg<11, 256, 8192>, 3, 3, 3, 2>(), 3) }, { ACM_K2M_L11 } },
	{ { ACM_K2M_ALT(entry_k2mw<TileCfg<12, 512, 16384>, 2, 3, 3, 3, 3>(), 3) }, { ACM_K2M_L12 } },
	/* level 13: the vector-ALU build needs 128 KB tiles (its first pass re-runs two rows per segment); here the rows in front cost a second
	 * read through L2 and nothing else, so a tile may be one row pair.  Level 14: a row pair IS 128 KB, sixteen waves of 128 registers;
	 * (4,3,3,2,2) and (4,3,2,3,2) spill seven of them, (4,2,3,3,2) none */
	{ { Tile2Entry{ nullptr, 0, 0, 0 }, 0 }, { ACM_K2M_L13 } },
	{ { Tile2Entry{ nullptr, 0, 0, 0 }, 0 }, { entry_k2mw<TileCfg<14, 1024, 32768>, 1, 4, 2, 3, 3, 2>(), 4 } },
};
constexpr int g_tile2m_default[ACM_K2M_MAX_LEVEL - ACM_K2M_MIN_LEVEL + 1] = { 3, 3, 3, 4, 4, 4, 4, 4 };
/* levels whose byte-plane tiles go to the chunk kernel (acm_chunk: six stages on the matrix cores, a wavefront per chunk of 2048 samples)
 * unless ACM_K3=0 asks for acm_tile2's matrix build; the staged form follows the choice (64 columns of a residue class side by side) */
/* ... and the six-stage first pass inside acm_tile2 (FirstPassZW): a row pair per tile, LDS passes with barriers behind it */
#ifndef ACM_K3_L13
#define ACM_K3_L13 entry_k2mw<TileCfg<13, 512, 16384>, 2, 6, 4, 3>(), 6
#endif
#ifndef ACM_K3_L14
#define ACM_K3_L14 entry_k2mw<TileCfg<14, 1024, 32768>, 1, 6, 3, 3, 2>(), 6
#endif
template <int L, int... Gs>
constexpr Tile2MEntry entry_k3()
{
	return Tile2MEntry{ Tile2Entry{ acm_chunk<L, 0, Gs...>, 64 * FirstPassZ<L>::NW, FirstPassZ<L>::TR, 1 }, 6 };
}
const Tile2MEntry g_chunk[ACM_K2M_MAX_LEVEL - ACM_K2M_MIN_LEVEL + 1] = {
	/* level 7 (two classes per row: eight row walkers per matrix set, a lane's four outputs belong to two of them) was built, is bit-exact
	 * (tests/test_gpu_byteplane.py at the time) and SLOWER than acm_tile2's matrix build, 0.616 against 0.680: the matrix work per sample
	 * is the same at every level while the LDS passes it replaces are few at level 7.  Level 8 (four walkers): 0.680 against 0.668. */
	{ Tile2Entry{ nullptr, 0, 0, 0 }, 0 },
	entry_k3<8, 2>(),
	entry_k3<9, 3>(),
	entry_k3<10, 2, 2>(),
	entry_k3<11, 3, 2>(),
	entry_k3<12, 3, 3>(),
	/* (level 13 as a chunk kernel: a row of 8192 per wavefront is four wavefronts of 350 registers - the staged rows of a chunk alone are
	 * 192 - and past 256 the compiler parks what the hand-issued loads have just asked for in accumulation registers BEFORE the wait,
	 * i.e. copies what is not there yet: tests/test_isa_invariants.py refuses the build.)  Levels 13 and 14: the same first pass shared
	 * by the wavefronts of a workgroup, in acm_tile2 */
	{ ACM_K3_L13 },
	{ ACM_K3_L14 },
};
inline const Tile2MEntry &tile2m_entry(uint32_t level)
{
	/* (tuning builds: ACM_K3=0 puts levels 8-12 back on acm_tile2's three / four-stage matrix build - and on ITS form: process-wide,
	 * read once, because the staged form follows the kernel, acmhip_mform_group) */
	static const bool k3 = !(ACM_TUNING_ENV("ACM_K3") && atoi(ACM_TUNING_ENV("ACM_K3")) == 0);
	if (k3 && g_chunk[level - ACM_K2M_MIN_LEVEL].g0)
		return g_chunk[level - ACM_K2M_MIN_LEVEL];
	const Tile2MEntry *row = g_tile2m[level - ACM_K2M_MIN_LEVEL];
#ifdef ACM_TUNING
	static const int forced = getenv("ACM_K2M_G0") ? atoi(getenv("ACM_K2M_G0")) : 0;
#else
	constexpr int forced = 0;
#endif
	const int want = forced ? forced : g_tile2m_default[level - ACM_K2M_MIN_LEVEL];
	return ((want == 4 || row[0].g0 == 0) && row[1].g0 == 4) ? row[1] : row[0];
}

// ---------------------------------------------------------------------------
// K2P: the lean tile kernel on the packed staged form
// ---------------------------------------------------------------------------
/*
 * Same tiles, same LDS passes, same write-out as acm_tile2; what differs is where the stage-0 inputs come from.  acm_tile2
 * owns a residue class per thread in its first pass, so its lanes stand for COLUMNS, and a column's index width is what varies
 * in the packed form (include/acm_hip.h): per lane the field width, the register a field sits in and the address would all
 * differ.  Here the unpack is a phase of its own in STORAGE order: the host stager has sorted the column pairs of a row group
 * by width class, a wavefront takes a chunk of 64 units (2 columns x 4 rows each) of ONE class - one scalar branch, then
 * SDWA multiplies (words, bytes) or v_bfe_i32 + multiply (nibbles) with compile-time field positions - and scatters
 * value = idx * val (decode.c:592-600, :174-177) into the LDS tile at (row, column pair) of the chunk's column-pair list.
 * All `level` stages then run as LDS passes in place (the pass that starts at stage 0 applies the "+1" of decode.c:561-564),
 * their histories carried from tile to tile like every other pass's: no warm-up rows are re-read or re-computed.
 * Per tile and thread: ~6 coalesced dword loads instead of 33, 0.6-0.9 B per index instead of 2 (4 on the rows acm_tile2 reads
 * twice), one more LDS round trip per element.
 * The chunk loads of the next tile are issued by hand before the LDS passes of this one and waited for with the tile's PCM
 * stores in flight (k2_wait); which slot holds what is scalar state (the chunk descriptors' second word).
 */
template <class C, int GR_>
struct PackGeo {
	static constexpr int GR = GR_;                          /* rows per group: one width class per column pair */
	static constexpr int NQ = GR / 4;                       /* row quads per group */
	static constexpr int RPC = 64 / NQ;                     /* column pairs (ranks) per chunk */
	static constexpr int PERMB = RPC * 2;                   /* bytes of a chunk's column-pair list */
	static constexpr int NG = C::TR / GR;
	static constexpr int P = C::COLS / 2;
	static constexpr int MAXCHUNKS = NG * (P * NQ / 64 + 3);   /* four classes per group, each rounded up to whole chunks */
	static constexpr int NW = C::NT / 64;
	static constexpr int NSLOT = (MAXCHUNKS + NW - 1) / NW; /* chunk descriptors per wave and tile */
	static constexpr int RS = C::COLS + (C::COLS >> C::PS); /* dwords between (row, c) and (row + 1, c) in the padded tile */
	static_assert(GR % 4 == 0 && 64 % NQ == 0 && C::TR % GR == 0 && (P * NQ) % 64 == 0 && PERMB % 16 == 0, "packed geometry");
	static_assert(C::COLS % (1 << C::PS) == 0, "rows are whole pad groups");
	/* where the row value of tile row `row` sits in LDS: a unit's four rows (q, q + NQ, q + 2 NQ, q + 3 NQ of their group)
	 * side by side, one 16-byte read per lane */
	static __device__ __forceinline__ int rowval_pos(const int row)
	{
		const int g = row / GR, rr = row % GR;
		return g * GR + 4 * (rr % NQ) + rr / NQ;
	}
};

template <class C, class PG>
struct PhaseU {
	static constexpr int L = C::L, NQ = PG::NQ, RPC = PG::RPC, PERMB = PG::PERMB, RS = PG::RS;
	struct Lane {
		uint32_t unit;          /* index of this lane's unit among the chunk's 64: rank * NQ + quad */
		uint32_t rank;          /* column pair of the chunk this lane works on */
		uint32_t perm_voff;     /* byte offset of the dword that holds its column-pair entry ... */
		uint32_t perm_sh;       /* ... and which half */
		uint32_t row_dw;        /* dwords from the group's first row to this lane's first row */
		uint32_t quad4;         /* 4 * quad: where its four row values start */
	};
	static __device__ __forceinline__ Lane lane_consts(const int tid)
	{
		/* adjacent lanes take adjacent column pairs of the same quad; the quads of a chunk start one row apart (their rows
		 * interleave), which puts the stores of one instruction on different banks */
		const uint32_t l = (uint32_t)tid & 63u, q = l / RPC, r = l % RPC;
		return Lane{ r * NQ + q, r, (r >> 1) * 4u, (r & 1u) * 16u, q * (uint32_t)RS, q * 4u };
	}
	/* one chunk: its column-pair entries and 1 / 2 / 4 dwords of units per lane, by kind.  ONE asm statement with the
	 * branches inside: every register has one definition on every path (tests/test_isa_invariants.py).  A chunk of zeros
	 * (kind 1) loads a dword of whatever follows its list: cheaper than a branch around it for all the others */
	static __device__ __forceinline__ void load(uint32_t (&d)[4], uint32_t &pm, const uint8_t *base, const uint32_t kind, const Lane &ln)
	{
		const uint32_t vd = ln.unit << kind;            /* 4, 8 or 16 bytes per unit */
		asm volatile("s_cmp_eq_u32 %[k], 0\n\ts_cbranch_scc1 .Lacm_pk%=\n\t"
			     "global_load_dword %[pm], %[vp], %[b]\n\t"
			     "global_load_dword %[d0], %[vd], %[b] offset:%[o0]\n\t"
			     "s_cmp_lt_u32 %[k], 3\n\ts_cbranch_scc1 .Lacm_pk%=\n\t"
			     "global_load_dword %[d1], %[vd], %[b] offset:%[o1]\n\t"
			     "s_cmp_lt_u32 %[k], 4\n\ts_cbranch_scc1 .Lacm_pk%=\n\t"
			     "global_load_dword %[d2], %[vd], %[b] offset:%[o2]\n\t"
			     "global_load_dword %[d3], %[vd], %[b] offset:%[o3]\n\t"
			     ".Lacm_pk%=:"
			     : [pm] "=&v"(pm), [d0] "=&v"(d[0]), [d1] "=&v"(d[1]), [d2] "=&v"(d[2]), [d3] "=&v"(d[3])
			     : [vp] "v"(ln.perm_voff), [vd] "v"(vd), [b] "s"(base), [k] "s"(kind), [o0] "n"(PERMB), [o1] "n"(PERMB + 4),
			       [o2] "n"(PERMB + 8), [o3] "n"(PERMB + 12)
			     : "memory", "scc");
	}
	/* a unit's eight indices times the values of its four rows (decode.c:592-600: midbuf[idx] == idx * val), by kind - one
	 * statement with scalar branches inside, the most frequent kind first: left to the compiler the three forms come out as a
	 * web of flag registers and zero fills on every path */
	static __device__ __forceinline__ void unpack(const uint32_t (&d)[4], const int32_t (&rv)[4], const uint32_t kind, uint32_t (&x)[4][2])
	{
#define ACM_PK_NIB(D, POS, V) "v_bfe_i32 " D ", %[d0], " POS ", 4\n\tv_mul_i32_i24 " D ", " D ", " V "\n\t"
#define ACM_PK_SDWA(D, S, SEL, V) "v_mul_i32_i24_sdwa " D ", sext(" S "), " V " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" SEL " src1_sel:DWORD\n\t"
		asm("s_cmp_eq_u32 %[k], 2\n\ts_cbranch_scc0 .Lacm_un_b%=\n\t"
		    ACM_PK_NIB("%[x0]", "0", "%[r0]") ACM_PK_NIB("%[x1]", "4", "%[r0]") ACM_PK_NIB("%[x2]", "8", "%[r1]") ACM_PK_NIB("%[x3]", "12", "%[r1]")
		    ACM_PK_NIB("%[x4]", "16", "%[r2]") ACM_PK_NIB("%[x5]", "20", "%[r2]") ACM_PK_NIB("%[x6]", "24", "%[r3]") ACM_PK_NIB("%[x7]", "28", "%[r3]")
		    "s_branch .Lacm_un_e%=\n"
		    ".Lacm_un_b%=:\n\ts_cmp_eq_u32 %[k], 3\n\ts_cbranch_scc0 .Lacm_un_w%=\n\t"
		    ACM_PK_SDWA("%[x0]", "%[d0]", "BYTE_0", "%[r0]") ACM_PK_SDWA("%[x1]", "%[d0]", "BYTE_1", "%[r0]")
		    ACM_PK_SDWA("%[x2]", "%[d0]", "BYTE_2", "%[r1]") ACM_PK_SDWA("%[x3]", "%[d0]", "BYTE_3", "%[r1]")
		    ACM_PK_SDWA("%[x4]", "%[d1]", "BYTE_0", "%[r2]") ACM_PK_SDWA("%[x5]", "%[d1]", "BYTE_1", "%[r2]")
		    ACM_PK_SDWA("%[x6]", "%[d1]", "BYTE_2", "%[r3]") ACM_PK_SDWA("%[x7]", "%[d1]", "BYTE_3", "%[r3]")
		    "s_branch .Lacm_un_e%=\n"
		    ".Lacm_un_w%=:\n\ts_cmp_eq_u32 %[k], 4\n\ts_cbranch_scc0 .Lacm_un_z%=\n\t"
		    ACM_PK_SDWA("%[x0]", "%[d0]", "WORD_0", "%[r0]") ACM_PK_SDWA("%[x1]", "%[d0]", "WORD_1", "%[r0]")
		    ACM_PK_SDWA("%[x2]", "%[d1]", "WORD_0", "%[r1]") ACM_PK_SDWA("%[x3]", "%[d1]", "WORD_1", "%[r1]")
		    ACM_PK_SDWA("%[x4]", "%[d2]", "WORD_0", "%[r2]") ACM_PK_SDWA("%[x5]", "%[d2]", "WORD_1", "%[r2]")
		    ACM_PK_SDWA("%[x6]", "%[d3]", "WORD_0", "%[r3]") ACM_PK_SDWA("%[x7]", "%[d3]", "WORD_1", "%[r3]")
		    "s_branch .Lacm_un_e%=\n"
		    ".Lacm_un_z%=:\n\t"
		    "v_mov_b32 %[x0], 0\n\tv_mov_b32 %[x1], 0\n\tv_mov_b32 %[x2], 0\n\tv_mov_b32 %[x3], 0\n\t"
		    "v_mov_b32 %[x4], 0\n\tv_mov_b32 %[x5], 0\n\tv_mov_b32 %[x6], 0\n\tv_mov_b32 %[x7], 0\n"
		    ".Lacm_un_e%=:"
		    : [x0] "=&v"(x[0][0]), [x1] "=&v"(x[0][1]), [x2] "=&v"(x[1][0]), [x3] "=&v"(x[1][1]), [x4] "=&v"(x[2][0]), [x5] "=&v"(x[2][1]),
		      [x6] "=&v"(x[3][0]), [x7] "=&v"(x[3][1])
		    : [d0] "v"(d[0]), [d1] "v"(d[1]), [d2] "v"(d[2]), [d3] "v"(d[3]), [r0] "v"(rv[0]), [r1] "v"(rv[1]), [r2] "v"(rv[2]), [r3] "v"(rv[3]),
		      [k] "s"(kind)
		    : "scc");
#undef ACM_PK_NIB
#undef ACM_PK_SDWA
	}
	/* meta = the chunk descriptor's second word: count | kind << 16 | row0 << 24 (0: the slot is empty) */
	typedef int32_t v4i __attribute__((ext_vector_type(4)));
	/* +-val of a unit's four rows (decode.c:589): one 16-byte read */
	static __device__ __forceinline__ v4i row_values(const int32_t *rowval, const uint32_t row0, const Lane &ln)
	{
		return *reinterpret_cast<const v4i *>(rowval + row0 + ln.quad4);
	}
	/* r4: the lane's row values when the tile is a single group (they are the same for every chunk then), else read per chunk */
	static __device__ __forceinline__ void compute(const uint32_t (&d)[4], const uint32_t pm, const uint32_t meta, uint32_t *tile,
						       const int32_t *rowval, const Lane &ln, const v4i r4_tile)
	{
		const uint32_t kind = (meta >> 16) & 0xFFu, count = meta & 0xFFFFu, row0 = meta >> 24;
		if (kind == 0u)
			return;
		if (ln.rank < count) {
			const uint32_t at = (pm >> ln.perm_sh) & 0xFFFFu;                       /* 2 p + p / 16: the pair's place in a padded row */
			uint32_t *o = tile + row0 * (uint32_t)RS + ln.row_dw + at;
			const v4i r4 = PG::NG == 1 ? r4_tile : row_values(rowval, row0, ln);
			const int32_t rv[4] = { r4.x, r4.y, r4.z, r4.w };
			uint32_t x[4][2];
			unpack(d, rv, kind, x);
#pragma unroll
			for (int k = 0; k < 4; k++) {
				o[k * NQ * RS] = x[k][0];
				o[k * NQ * RS + 1] = x[k][1];
			}
		}
	}
};

template <class C, int WPS, int GR, int G0, int... Gs>
__global__ void __launch_bounds__(C::NT, WPS)
acm_tile2p(const AcmTile2 *__restrict__ tiles, const uint32_t ntiles, const uint2 *__restrict__ chunks /* acmhip_packed_chunk as two words */,
	   const uint8_t *__restrict__ blob, const acmhip_blkhdr *__restrict__ hdr, int16_t *__restrict__ pcm, int16_t *__restrict__ sink,
	   const unsigned fmt)
{
	constexpr int L = C::L, NT = C::NT, COLS = C::COLS, NELEM = C::NELEM, TR = C::TR, NJ_LAST = C::NJ_LAST;
	constexpr bool NEG_ODD_ROWS = StageKind<L, 0>::N;
	using PG = PackGeo<C, GR>;
	using PU = PhaseU<C, PG>;
	constexpr int NSLOT = PG::NSLOT;
	static_assert(TR <= NT, "one row value per thread");
	constexpr bool PRIO = WPS * 256 / NT > 1;

	__shared__ uint32_t tile_mem[8 + NELEM + (NELEM >> C::PS)];
	__shared__ __attribute__((aligned(16))) int32_t rowval[2][TR];
	constexpr int NCARRY_WORDS = carry_total<C, 0, G0, Gs...>();
	__shared__ uint32_t carry_mem[NCARRY_WORDS];
	uint32_t *const tile = tile_mem + 8;

	const int tid = threadIdx.x;
	const uint32_t per = (ntiles + gridDim.x - 1) / gridDim.x;
	uint32_t t = blockIdx.x * per;
	const uint32_t t_end = t + per < ntiles ? t + per : ntiles;
	if (t >= t_end)
		return;
	bool discard = false;
	if (!(tiles[__builtin_amdgcn_readfirstlane(t)].flags & ACM_TILE_FRESH)) {
		discard = true;                 /* a run that starts inside a stream replays the tile in front of it without storing PCM */
		t--;
	}

	const typename PU::Lane ln = PU::lane_consts(tid);
	const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane(tid) >> 6;

	/* row values: thread lr < TR fetches the val of tile row lr (decode.c:589); looked at in finish_val, a whole tile later */
	const uint32_t lr_fetch = (uint32_t)(tid < TR ? tid : TR - 1);
	const int rv_pos = PG::rowval_pos(tid < TR ? tid : 0);
	auto fetch_val = [&](const AcmTile2 &r) -> uint32_t {
		const uint32_t q = r.rowpos + lr_fetch;
		const uint32_t b = r.magic ? __umulhi(q, r.magic) : q;     /* q / acm_rows */
		const uint32_t *p = &hdr[r.hdr_blk + b].val;
		uint32_t v;
		asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
		return v;
	};
	auto finish_val = [&](uint32_t v) -> int32_t {
		v <<= OutScale<L>::SHIFT;
		return (NEG_ODD_ROWS && (tid & 1)) ? -(int32_t)v : (int32_t)v;
	};
	auto sgpr_ptr = [&](const uint64_t a) -> const uint8_t * {
		const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
		return reinterpret_cast<const uint8_t *>(((uint64_t)hi << 32) | lo);
	};
	/* this wave's chunk descriptors of a tile: NSLOT consecutive entries of the table (the packer deals a tile's chunks out to
	 * the waves).  They come through the scalar cache - the index is wave-uniform; a vector load here would be tracked by the
	 * compiler's vmcnt bookkeeping, which knows nothing of the asm loads - an iteration before the chunk loads they describe */
	struct Descs { uint2 w[NSLOT]; };
	auto fetch_descs = [&](const AcmTile2 &r) -> Descs {
		const uint64_t at = r.idx_off + wv * (uint32_t)NSLOT;           /* idx_off of a packed tile's record: its first entry in the chunk table */
		const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)at), hi = __builtin_amdgcn_readfirstlane((uint32_t)(at >> 32));
		const uint2 *run = chunks + (((uint64_t)hi << 32) | lo);
		Descs ds;
#pragma unroll
		for (int s = 0; s < NSLOT; s++)
			ds.w[s] = run[s];               /* { blob_off16, count | kind << 16 | row0 << 24 } */
		return ds;
	};
	auto load_tile = [&](uint32_t (&d)[NSLOT][4], uint32_t (&pm)[NSLOT], uint32_t (&meta)[NSLOT], const Descs &ds) {
#pragma unroll
		for (int s = 0; s < NSLOT; s++) {
			meta[s] = ds.w[s].y;
			PU::load(d[s], pm[s], sgpr_ptr(reinterpret_cast<uint64_t>(blob) + (uint64_t)ds.w[s].x * 16u), (meta[s] >> 16) & 0xFFu, ln);
		}
	};
	auto tile_at = [&](const uint32_t k) -> AcmTile2 { return tiles[__builtin_amdgcn_readfirstlane(k < t_end ? k : t_end - 1u)]; };

	constexpr int NVEC = TR * COLS / 8, PER_OWNER = NJ_LAST / 8, NSTORE = NVEC / NT;
	static_assert(NVEC % NT == 0, "whole rounds");
	AcmTile2 cur = tile_at(t);
	uint32_t d[NSLOT][4], pm[NSLOT], meta[NSLOT];
	uint32_t hv = fetch_val(cur);
	load_tile(d, pm, meta, fetch_descs(cur));
	k2_wait<0>();
	int buf = 0;
	bool fresh = true;              /* the first tile of a run starts from zero carries (stream start or lead-in) */
	bool first_rows = (cur.flags & ACM_TILE_FRESH) != 0;    /* nothing in front of this tile: no "+1" in front of it either */
	constexpr uint32_t ONE = 1u << OutScale<L>::SHIFT;
	const uint32_t bias = (tid % PassGeo<C, 0, G0>::SIGMA) == 0 ? ONE : 0u;

	/* tile records come through the scalar cache one iteration ahead, the chunk descriptors of that tile at the top of the
	 * iteration that issues its chunk loads; the last tile of a run names itself as its successor */
	AcmTile2 nxt = tile_at(t + 1);
	for (;;) {
		const uint32_t tn = t + 1;
		const bool more = tn < t_end;
		if (fresh)
			for (int k = tid; k < NCARRY_WORDS; k += NT)
				carry_mem[k] = 0u;
		if (tid < TR)
			rowval[buf][rv_pos] = finish_val(hv);
		const Descs dn = fetch_descs(nxt);
		__syncthreads();                /* row values complete; the previous write-out is done with the tile */
#ifndef ACM_PK_PRIO_U
#define ACM_PK_PRIO_U PRIO_FIRST_PASS
#endif
		phase_prio<PRIO, ACM_PK_PRIO_U>();
		{
			typename PU::v4i r4 = { 0, 0, 0, 0 };
			if constexpr (PG::NG == 1)
				r4 = PU::row_values(rowval[buf], 0u, ln);
#pragma unroll
			for (int s = 0; s < NSLOT; s++)
				PU::compute(d[s], pm[s], meta[s], tile, rowval[buf], ln, r4);
		}
		phase_prio<PRIO, PRIO_IDLE>();

		hv = fetch_val(nxt);            /* the last tile of a run fetches its own again: no branch around the loads */
		load_tile(d, pm, meta, dn);
		phase_prio<PRIO, PRIO_LDS_PASSES>();
		/* history in front of the stream is zeros: no "+1" there (decode.c:561-564 runs on existing rows only) */
		run_lds_passes<C, 0, true, 0, G0, Gs...>(tile, tid, fmt, carry_mem, bias, (first_rows && tid < PassGeo<C, 0, G0>::SIGMA) ? 0u : bias);
		__syncthreads();
		{
			typedef uint32_t v4u __attribute__((ext_vector_type(4)));
			v4u *out = discard ? reinterpret_cast<v4u *>(sink) : reinterpret_cast<v4u *>(reinterpret_cast<uint16_t *>(pcm) + cur.pcm_off);
#pragma unroll
			for (int k = 0; k < NVEC / NT; k++) {
				const int vec = tid + k * NT;
				const uint32_t *q = tile + lds_at<C::PS>((vec / PER_OWNER) * NJ_LAST) + (vec % PER_OWNER) * park_piece<NJ_LAST>();
				const v4u o = { q[0], q[1], q[2], q[3] };
				__builtin_nontemporal_store(o, &out[vec]);
			}
		}
		phase_prio<PRIO, PRIO_IDLE>();
		k2_wait<NSTORE>();              /* the next tile's chunks are here; only this tile's PCM stores may still be on their way */
		if (!more)
			break;
		fresh = first_rows = (nxt.flags & ACM_TILE_FRESH) != 0;
		discard = false;
		cur = nxt;
		t = tn;
		nxt = tile_at(t + 1);
		buf ^= 1;
	}
}

struct Tile2PEntry {
	typedef void (*Fn)(const AcmTile2 *, uint32_t, const uint2 *, const uint8_t *, const acmhip_blkhdr *, int16_t *, int16_t *, unsigned);
	Fn fn;
	int threads, tile_rows, wg_per_cu, group_rows, slots, pad_shift;
};
template <class C, int GR, int... Gs>
constexpr Tile2PEntry entry_k2p()
{
	return Tile2PEntry{ acm_tile2p<C, 4, GR, Gs...>, C::NT, C::TR, 1024 / C::NT, GR, PackGeo<C, GR>::NW * PackGeo<C, GR>::NSLOT, C::PS };
}
/* the tile geometries and stage groupings of acm_tile2 (g_tile2); groups of 16 rows = one block of the usual acm_rows = 16
 * (level 6: 32, or a wave would hold ten chunks) */
const Tile2PEntry g_tile2p[ACM_K2P_MAX_LEVEL - ACM_K2P_MIN_LEVEL + 1] = {
	entry_k2p<TileCfg<6, 256, 8192>, 32, 2, 2, 2>(),
	entry_k2p<TileCfg<7, 256, 8192>, 16, 3, 2, 2>(),
	entry_k2p<TileCfg<8, 256, 8192>, 16, 3, 3, 2>(),
	entry_k2p<TileCfg<9, 256, 8192>, 16, 3, 3, 3>(),
};

/* levels 13-15: the stage-wise kernels apply the first level-12 stages into an int32 plane, this level-12 build of the tile
 * kernel (one 128 KB tile per CU - two 64 KB tiles spill with the 64 prefetch registers of a plane; halo or carry flavour like
 * every other group) reads the plane and does the other twelve */
const FusedEntry g_fused_plane = { acm_fused_tile<TileCfg<12, 512, 32768>, 2, MODE_PLANE, 2, false, 3, 3, 3, 3>, 512, 8, 1,
				   acm_fused_tile<TileCfg<12, 512, 32768>, 2, MODE_PLANE, 2, true, 3, 3, 3, 3> };

inline dim3 sw_grid(uint64_t max_elems, uint32_t nlist)
{
	uint64_t gx = (max_elems + (uint64_t)SW_THREADS * 4 - 1) / ((uint64_t)SW_THREADS * 4);
	if (gx < 1)
		gx = 1;
	if (gx > 2048)
		gx = 2048;
	return dim3((unsigned)gx, nlist, 1);
}

} // namespace

#define ACMK_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

namespace {
__global__ void acm_warmup_kernel() {}
}

extern "C" int acmk_tuning_build(void)
{
#ifdef ACM_TUNING
	return 1;
#else
	return 0;
#endif
}

extern "C" int acmk_warmup(void *stream)
{
	hipLaunchKernelGGL(acm_warmup_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream);
	ACMK_CHECK_LAUNCH();
	return 0;
}

/* gridDim.y carries the stream index and is limited to 65535: walk long lists in slices */
constexpr uint32_t SW_MAX_Y = 65535;

extern "C" int acmk_fused_variants(void)
{
	return NVARIANTS;
}

extern "C" int acmk_fused_tile_rows(uint32_t level, int variant)
{
	if (level < ACM_K1_MIN_LEVEL || level > ACM_K1_MAX_LEVEL || variant < 0 || variant >= NVARIANTS)
		return 0;
	return g_fused[variant][level - ACM_K1_MIN_LEVEL].tile_rows;
}

extern "C" int acmk_fused_has_carry(uint32_t level, int variant)
{
	if (level < ACM_K1_MIN_LEVEL || level > ACM_K1_MAX_LEVEL || variant < 0 || variant >= NVARIANTS)
		return 0;
	return g_fused[variant][level - ACM_K1_MIN_LEVEL].fn_carry != nullptr;
}

extern "C" int acmk_fused_grid(uint32_t level, int variant, int cus)
{
	if (level < ACM_K1_MIN_LEVEL || level > ACM_K1_MAX_LEVEL || variant < 0 || variant >= NVARIANTS)
		return 0;
	return (cus > 0 ? cus : 256) * g_fused[variant][level - ACM_K1_MIN_LEVEL].wg_per_cu;
}

extern "C" int acmk_launch_fused(uint32_t level, int variant, int cus, int carry, const AcmDevStream *d_streams, const AcmTile *d_tiles,
				 uint32_t ntiles, const int16_t *d_idx, const acmhip_blkhdr *d_hdr,
				 int16_t *d_pcm, unsigned fmt, void *stream)
{
	if (ntiles == 0)
		return 0;
	if (level < ACM_K1_MIN_LEVEL || level > ACM_K1_MAX_LEVEL || variant < 0 || variant >= NVARIANTS)
		return -1;
	const FusedEntry &e = g_fused[variant][level - ACM_K1_MIN_LEVEL];
	/* persistent grid: as many workgroups as the chip holds at once, never more than tiles.  `cus` comes from
	 * the device handle: nothing here queries or synchronises, so the launch can be captured into a hipGraph */
	uint32_t grid = (uint32_t)((cus > 0 ? cus : 256) * e.wg_per_cu);
	if (grid > ntiles)
		grid = ntiles;
	if (carry && !e.fn_carry)
		return -1;
	hipLaunchKernelGGL(carry ? e.fn_carry : e.fn, dim3(grid), dim3(e.threads), 0, (hipStream_t)stream,
			   d_streams, d_tiles, ntiles, d_idx, d_hdr, d_pcm, fmt);
	ACMK_CHECK_LAUNCH();
	return 0;
}

/* shift: the planes of a level 13-15 prefix carry values scaled by 2^shift (16 - level) so that the level-12 tile kernel
 * that finishes them finds a sample in bytes 2..3, as it does for its own levels; 0 everywhere else */
extern "C" int acmk_launch_unpack(const AcmDevStream *d_streams, const uint32_t *d_list, uint32_t nlist,
				  uint64_t max_elems, const int16_t *d_idx, const acmhip_blkhdr *d_hdr,
				  int32_t *d_x, uint32_t shift, void *stream)
{
	for (uint32_t at = 0; at < nlist; at += SW_MAX_Y) {
		const uint32_t n = nlist - at < SW_MAX_Y ? nlist - at : SW_MAX_Y;
		hipLaunchKernelGGL(acm_sw_unpack, sw_grid(max_elems, n), dim3(SW_THREADS), 0, (hipStream_t)stream,
				   d_streams, d_list + at, d_idx, d_hdr, d_x, shift);
		ACMK_CHECK_LAUNCH();
	}
	return 0;
}

extern "C" int acmk_launch_patch(const AcmDevPatch *d_patches, uint64_t n, int32_t *d_x, void *stream)
{
	if (n == 0)
		return 0;
	hipLaunchKernelGGL(acm_sw_patch, dim3((unsigned)((n + SW_THREADS - 1) / SW_THREADS)), dim3(SW_THREADS), 0,
			   (hipStream_t)stream, d_patches, n, d_x);
	ACMK_CHECK_LAUNCH();
	return 0;
}

extern "C" int acmk_launch_stage(const AcmDevStream *d_streams, const uint32_t *d_list, uint32_t nlist,
				 uint64_t max_elems, uint32_t level, uint32_t k, const int32_t *d_in,
				 int32_t *d_out, uint32_t shift, void *stream)
{
	for (uint32_t at = 0; at < nlist; at += SW_MAX_Y) {
		const uint32_t n = nlist - at < SW_MAX_Y ? nlist - at : SW_MAX_Y;
		hipLaunchKernelGGL(acm_sw_stage, sw_grid(max_elems, n), dim3(SW_THREADS), 0, (hipStream_t)stream,
				   d_streams, d_list + at, level, k, d_in, d_out, 1u << shift);
		ACMK_CHECK_LAUNCH();
	}
	return 0;
}

extern "C" int acmk_launch_small(uint32_t level, const AcmDevStream *d_streams, const uint32_t *d_list, uint32_t nlist,
				 uint64_t max_emit, const int16_t *d_idx, const acmhip_blkhdr *d_hdr, int16_t *d_pcm,
				 unsigned fmt, void *stream)
{
	if (level > ACM_SMALL_MAX_LEVEL)
		return -1;
	uint64_t gx = (max_emit + (uint64_t)SL_THREADS * SL_K - 1) / ((uint64_t)SL_THREADS * SL_K);
	gx = gx < 1 ? 1 : gx > 4096 ? 4096 : gx;
	for (uint32_t at = 0; at < nlist; at += SW_MAX_Y) {
		const uint32_t n = nlist - at < SW_MAX_Y ? nlist - at : SW_MAX_Y;
		const dim3 grid((unsigned)gx, n, 1);
		switch (level) {
		case 0: hipLaunchKernelGGL(acm_small_level<0>, grid, dim3(SL_THREADS), 0, (hipStream_t)stream, d_streams, d_list + at, d_idx, d_hdr, d_pcm, fmt); break;
		case 1: hipLaunchKernelGGL(acm_small_level<1>, grid, dim3(SL_THREADS), 0, (hipStream_t)stream, d_streams, d_list + at, d_idx, d_hdr, d_pcm, fmt); break;
		case 2: hipLaunchKernelGGL(acm_small_level<2>, grid, dim3(SL_THREADS), 0, (hipStream_t)stream, d_streams, d_list + at, d_idx, d_hdr, d_pcm, fmt); break;
		case 3: hipLaunchKernelGGL(acm_small_level<3>, grid, dim3(SL_THREADS), 0, (hipStream_t)stream, d_streams, d_list + at, d_idx, d_hdr, d_pcm, fmt); break;
		default: hipLaunchKernelGGL(acm_small_level<4>, grid, dim3(SL_THREADS), 0, (hipStream_t)stream, d_streams, d_list + at, d_idx, d_hdr, d_pcm, fmt); break;
		}
		ACMK_CHECK_LAUNCH();
	}
	return 0;
}

extern "C" int acmk_launch_emit(const AcmDevStream *d_streams, const uint32_t *d_list, uint32_t nlist,
				uint64_t max_emit, const int32_t *d_x, int16_t *d_pcm, unsigned fmt, void *stream)
{
	for (uint32_t at = 0; at < nlist; at += SW_MAX_Y) {
		const uint32_t n = nlist - at < SW_MAX_Y ? nlist - at : SW_MAX_Y;
		hipLaunchKernelGGL(acm_sw_emit, sw_grid(max_emit, n), dim3(SW_THREADS), 0, (hipStream_t)stream,
				   d_streams, d_list + at, d_x, d_pcm, fmt);
		ACMK_CHECK_LAUNCH();
	}
	return 0;
}

extern "C" int acmk_tile2_rows(uint32_t level)
{
	if (level < ACM_K2_MIN_LEVEL || level > ACM_K2_MAX_LEVEL)
		return 0;
	return tile2_entry(level).tile_rows;
}

extern "C" int acmk_tile2_grid(uint32_t level, int cus)
{
	if (level < ACM_K2_MIN_LEVEL || level > ACM_K2_MAX_LEVEL)
		return 0;
	return (cus > 0 ? cus : 256) * tile2_entry(level).wg_per_cu;
}

extern "C" int acmk_launch_tile2(uint32_t level, int cus, const AcmTile2 *d_tiles, uint32_t ntiles, const int16_t *d_idx, const acmhip_blkhdr *d_hdr,
				 int16_t *d_pcm, int16_t *d_sink, unsigned fmt, void *stream)
{
	if (ntiles == 0)
		return 0;
	if (!d_sink)
		return -1;
	if (level < ACM_K2_MIN_LEVEL || level > ACM_K2_MAX_LEVEL)
		return -1;
	Tile2Entry e = tile2_entry(level);
#ifdef ACM_ABLATION
	if (const char *a = getenv("ACM_K2_ABL"))
		for (const auto &x : g_tile2_abl)
			if (level == 9 && x.mask == atoi(a))
				e = x.e;
#endif
	uint32_t grid = (uint32_t)((cus > 0 ? cus : 256) * e.wg_per_cu);
	if (grid > ntiles)
		grid = ntiles;
	hipLaunchKernelGGL(e.fn, dim3(grid), dim3(e.threads), 0, (hipStream_t)stream, d_tiles, ntiles, d_idx, (const uint32_t *)nullptr, d_hdr, d_pcm, d_sink, fmt);
	ACMK_CHECK_LAUNCH();
	return 0;
}

extern "C" int acmk_tile2m_rows(uint32_t level)
{
	if (level < ACM_K2M_MIN_LEVEL || level > ACM_K2M_MAX_LEVEL)
		return 0;
	return tile2m_entry(level).e.tile_rows;
}

/* tiles of this build the planner puts in front of a window into a stream (ACM_TILE_DISCARD records): one - every pass reaches back less
 * than a tile - except where the chunk kernel keeps the two rows in front of a walk in registers from the chunk before: there the
 * last lead-in chunk needs lead-in chunks of its own for those rows (levels 10: one more; 11, 12 - chunks of one row -: two more) */
extern "C" int acmk_tile2m_lead_in(uint32_t level)
{
	if (level < ACM_K2M_MIN_LEVEL || level > ACM_K2M_MAX_LEVEL || tile2m_entry(level).g0 != 6)
		return 1;
	switch (level) {
	case 8: return FirstPassZ<8>::KEEPS_ROWS ? 1 + (2 + FirstPassZ<8>::NSW - 1) / FirstPassZ<8>::NSW : 1;
	case 9: return FirstPassZ<9>::KEEPS_ROWS ? 1 + (2 + FirstPassZ<9>::NSW - 1) / FirstPassZ<9>::NSW : 1;
	case 10: return FirstPassZ<10>::HISTORY_IN_REGISTERS ? 1 + (2 + FirstPassZ<10>::NSW - 1) / FirstPassZ<10>::NSW : 1;
	case 11: return FirstPassZ<11>::HISTORY_IN_REGISTERS ? 1 + (2 + FirstPassZ<11>::NSW - 1) / FirstPassZ<11>::NSW : 1;
	case 12: return FirstPassZ<12>::HISTORY_IN_REGISTERS ? 1 + (2 + FirstPassZ<12>::NSW - 1) / FirstPassZ<12>::NSW : 1;
	case 13: case 14: return 2;             /* FirstPassZW: a row pair per tile, the pair in front of it in registers */
	default: return 1;
	}
}

/* wavefronts that share a launch of the chunk kernel's table among them (each takes one contiguous run of it); 0: not the chunk kernel */
extern "C" int acmk_tile2m_run_waves(uint32_t level, int cus)
{
	if (level < ACM_K2M_MIN_LEVEL || level > 12 || tile2m_entry(level).g0 != 6)
		return 0;
	const Tile2Entry &e = tile2m_entry(level).e;
	return (cus > 0 ? cus : 256) * e.wg_per_cu * (e.threads / 64);
}

extern "C" int acmk_tile2m_stages(uint32_t level)
{
	if (level < ACM_K2M_MIN_LEVEL || level > ACM_K2M_MAX_LEVEL)
		return 0;
	return tile2m_entry(level).g0;
}

extern "C" int acmk_launch_tile2m(uint32_t level, int cus, const AcmTile2 *d_tiles, uint32_t ntiles, const uint8_t *d_mform, const acmhip_mform_pair *d_pairs,
				  const acmhip_blkhdr *d_hdr, int16_t *d_pcm, int16_t *d_sink, unsigned fmt, void *stream)
{
	if (ntiles == 0)
		return 0;
	if (!d_sink || !d_mform || !d_pairs)
		return -1;
	if (level < ACM_K2M_MIN_LEVEL || level > ACM_K2M_MAX_LEVEL)
		return -1;
	const Tile2Entry &e = tile2m_entry(level).e;
	uint32_t grid = (uint32_t)((cus > 0 ? cus : 256) * e.wg_per_cu);
	if (grid > ntiles)
		grid = ntiles;
	hipLaunchKernelGGL(e.fn, dim3(grid), dim3(e.threads), 0, (hipStream_t)stream, d_tiles, ntiles, reinterpret_cast<const int16_t *>(d_mform),
			   reinterpret_cast<const uint32_t *>(d_pairs), d_hdr, d_pcm, d_sink, fmt);
	ACMK_CHECK_LAUNCH();
	return 0;
}

extern "C" int acmk_tile2p_rows(uint32_t level)
{
	if (level < ACM_K2P_MIN_LEVEL || level > ACM_K2P_MAX_LEVEL)
		return 0;
	return g_tile2p[level - ACM_K2P_MIN_LEVEL].tile_rows;
}

extern "C" int acmk_tile2p_group_rows(uint32_t level)
{
	if (level < ACM_K2P_MIN_LEVEL || level > ACM_K2P_MAX_LEVEL)
		return 0;
	return g_tile2p[level - ACM_K2P_MIN_LEVEL].group_rows;
}

extern "C" int acmk_tile2p_slots(uint32_t level)
{
	if (level < ACM_K2P_MIN_LEVEL || level > ACM_K2P_MAX_LEVEL)
		return 0;
	return g_tile2p[level - ACM_K2P_MIN_LEVEL].slots;
}

extern "C" int acmk_tile2p_waves(uint32_t level)
{
	if (level < ACM_K2P_MIN_LEVEL || level > ACM_K2P_MAX_LEVEL)
		return 0;
	return g_tile2p[level - ACM_K2P_MIN_LEVEL].threads / 64;
}

extern "C" int acmk_tile2p_pad_shift(uint32_t level)
{
	if (level < ACM_K2P_MIN_LEVEL || level > ACM_K2P_MAX_LEVEL)
		return 0;
	return g_tile2p[level - ACM_K2P_MIN_LEVEL].pad_shift;
}

extern "C" int acmk_launch_tile2p(uint32_t level, int cus, const AcmTile2 *d_tiles, uint32_t ntiles, const acmhip_packed_chunk *d_chunks, const uint8_t *d_blob,
				  const acmhip_blkhdr *d_hdr, int16_t *d_pcm, int16_t *d_sink, unsigned fmt, void *stream)
{
	if (ntiles == 0)
		return 0;
	if (!d_sink || !d_chunks || !d_blob)
		return -1;
	if (level < ACM_K2P_MIN_LEVEL || level > ACM_K2P_MAX_LEVEL)
		return -1;
	const Tile2PEntry &e = g_tile2p[level - ACM_K2P_MIN_LEVEL];
	uint32_t grid = (uint32_t)((cus > 0 ? cus : 256) * e.wg_per_cu);
	if (grid > ntiles)
		grid = ntiles;
	static_assert(sizeof(acmhip_packed_chunk) == sizeof(uint2), "a chunk descriptor is two words");
	hipLaunchKernelGGL(e.fn, dim3(grid), dim3(e.threads), 0, (hipStream_t)stream, d_tiles, ntiles, reinterpret_cast<const uint2 *>(d_chunks), d_blob, d_hdr,
			   d_pcm, d_sink, fmt);
	ACMK_CHECK_LAUNCH();
	return 0;
}

extern "C" int acmk_plane_tile_rows(void)
{
	return g_fused_plane.tile_rows;
}

/* d_plane stands where the staged indices stand in acmk_launch_fused; the streams' idx_off count int32 units into it */
extern "C" int acmk_plane_grid(int cus)
{
	return (cus > 0 ? cus : 256) * g_fused_plane.wg_per_cu;
}

extern "C" int acmk_launch_fused_plane(int cus, int carry, const AcmDevStream *d_streams, const AcmTile *d_tiles, uint32_t ntiles,
				       const int32_t *d_plane, int16_t *d_pcm, unsigned fmt, void *stream)
{
	if (ntiles == 0)
		return 0;
	uint32_t grid = (uint32_t)((cus > 0 ? cus : 256) * g_fused_plane.wg_per_cu);
	if (grid > ntiles)
		grid = ntiles;
	hipLaunchKernelGGL(carry ? g_fused_plane.fn_carry : g_fused_plane.fn, dim3(grid), dim3(g_fused_plane.threads), 0, (hipStream_t)stream,
			   d_streams, d_tiles, ntiles, reinterpret_cast<const int16_t *>(d_plane), nullptr, d_pcm, fmt);
	ACMK_CHECK_LAUNCH();
	return 0;
}

extern "C" int acmk_launch_prefix(const AcmDevStream *d_streams, const uint32_t *d_list, uint32_t nlist, uint64_t max_elems,
				 uint32_t level, const int16_t *d_idx, const acmhip_blkhdr *d_hdr, int32_t *d_y, void *stream)
{
	if (level < 13 || level > 15)
		return (int)hipErrorInvalidValue;
	const uint64_t rows = max_elems >> level;
	const uint32_t chunks = (uint32_t)((rows + PREFIX_CHUNK_ROWS - 1) / PREFIX_CHUNK_ROWS);
	const uint32_t gx = (chunks ? chunks : 1) * (2048 / PREFIX_THREADS);
	for (uint32_t at = 0; at < nlist; at += SW_MAX_Y) {
		const uint32_t n = nlist - at < SW_MAX_Y ? nlist - at : SW_MAX_Y;
		auto k = level == 13 ? acm_sw_prefix<1> : level == 14 ? acm_sw_prefix<2> : acm_sw_prefix<3>;
		hipLaunchKernelGGL(k, dim3(gx, n), dim3(PREFIX_THREADS), 0, (hipStream_t)stream,
				   d_streams, d_list + at, d_idx, d_hdr, d_y, 16 - level);
		ACMK_CHECK_LAUNCH();
	}
	return 0;
}
