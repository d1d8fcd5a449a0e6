/*
 * acm_kernels.hip - the device half of the ACM decode path, gfx950 (CDNA4).
 *
 * What is computed (reference: /root/reference/src/decode.c):
 *   unpack    value = idx * val                      (table :592-600, lookup :174-177)
 *   synthesis `level` cascaded butterfly stages      (juggle :508-526, juggle_block :528-577)
 *   write-out (value >> level) as 16-bit             (:617-677)
 *
 * Formulation (verified against the reference by
 * tests/test_cascade_equiv.py): with m the flat sample index of a stream
 * (row*cols + col, running on across blocks) stage k, stride s = cols >> (k+1):
 *     y[m] = 2*x[m-s] + sg*(x[m-2s] + x[m]),  sg = +1 if bit log2(s) of m is 0 else -1
 *     after stage 0 only: y[m] += 1 where m % (cols/2) == 0
 *     x[<0] = 0, all arithmetic mod 2^32.
 * The reference's wrapbuf is just the two previous inputs per column per
 * stage, so an output depends on at most 2*cols-2 earlier raw samples: any
 * run of rows can be synthesised from its own staged rows plus the two rows
 * before it ("halo").  No state is carried between launches.
 *
 * Two kernel families:
 *   fused tile kernel (levels 5..11, no H1 patches): one workgroup = one tile of
 *     TR rows (2 halo + T payload) held in LDS as int32; load+unpack, then the
 *     stages in groups of G=2..3 per LDS round trip ("passes": each thread owns
 *     one residue class of the pass's smallest stride and walks it with the
 *     inputs of the G stages in registers), then convert+store.  HBM traffic:
 *     2 B read (+2/T halo) + 2 B written per sample.
 *   stage-wise kernels (any level 0..15, H1 patches): unpack to an int32 plane,
 *     one elementwise launch per stage (ping-pong planes), emit.  8*level B of
 *     HBM traffic per sample; generic fallback and cross-check.
 *
 * Integer add/shift only: bound by HBM (and LDS/VALU at high levels), no MFMA.
 */
#include <hip/hip_runtime.h>

#include "acm_device.h"

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
	      int32_t *__restrict__ x)
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
		dst[e] = (int32_t)((uint32_t)(int32_t)src[e] * val);    // midbuf[idx] == idx*val (:592-600)
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
	     uint32_t level, uint32_t k, const int32_t *__restrict__ in, int32_t *__restrict__ out)
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
			y += 1u;                                        // :561-564
		yo[e] = y;
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
// fused tile kernel
// ---------------------------------------------------------------------------
constexpr int NT = ACM_K1_THREADS;

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
 * LDS layout of the tile: one pad dword after every 64 elements,
 * addr(m) = m + (m >> 6).  With it every access pattern of every pass is
 * bank-conflict free for 4-byte accesses: a wave reading residue i of stride
 * sigma < 64 touches 64/sigma walk segments whose starts are 64*sigma apart -
 * all on the same banks without the pad, rotated by sigma banks each with it.
 */
__device__ __forceinline__ int lds_at(int m) { return m + (m >> 6); }

/*
 * Write-out without per-sample shifts where possible.  For level <= 8 the whole
 * cascade runs on values scaled by 2^(8-level) (the unpack multiplies by
 * val << (8-level), the "+1" becomes 1 << (8-level); everything is linear mod
 * 2^32 and only bits [level, level+16) of the result are used, which the
 * scaling moves to bits [8, 24) - still inside the 25 bits v_mad_i32_i24 keeps
 * exact).  The 16-bit sample is then bytes 1..2 of the value and two samples
 * are packed, byte-swapped if asked, by ONE v_perm_b32; unsigned output is one
 * xor per pair.  Levels 9..11 shift first.
 */
template <int L>
struct OutScale {
	static constexpr int SHIFT = (L <= 8) ? 8 - L : 0;
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
	if (L <= 8)      /* sample = bytes 1,2 of each value: S1 = first value (bytes 0-3), S0 = second (bytes 4-7) */
		f.sel = be ? 0x05060102u : 0x06050201u;
	else             /* values already shifted down: sample = bytes 0,1 */
		f.sel = be ? 0x04050001u : 0x05040100u;
	f.flip = uns ? (be ? 0x00800080u : 0x80008000u) : 0u;
	return f;
}

template <int L>
__device__ __forceinline__ uint32_t pack_pcm(uint32_t a, uint32_t b, const PcmFmt &f)
{
	if (L > 8) {
		a >>= L;
		b >>= L;
	}
	return __builtin_amdgcn_perm(b, a, f.sel) ^ f.flip;
}

template <int L, int K0, int G>
struct PassGeo {
	static constexpr int COLS = 1 << L;
	static constexpr int NELEM = (L >= 11) ? 32768 : 16384;
	static constexpr int SIGMA = COLS >> (K0 + G);          // smallest stride of the pass
	static constexpr int U = 1 << G;
	static constexpr int BODY = 2 * U;                      // elements per unrolled body
	static constexpr int NJ_TOTAL = NELEM / SIGMA;          // walk length of one residue over the tile
	static constexpr bool MULTI_RES = SIGMA >= NT;          // at least as many residues as threads
	static constexpr int RPT = MULTI_RES ? SIGMA / NT : 1;  // residues per thread
	static constexpr int NSEG = MULTI_RES ? 1 : NT / SIGMA; // walk segments per residue
	static constexpr int NJ = NJ_TOTAL / NSEG;              // walk length per thread
	static_assert(SIGMA >= 1, "pass exceeds level");
	static_assert(NJ % BODY == 0 && NJ >= BODY, "segment must be whole bodies");
	/* LDS offset of walk element u relative to the body's first element (which is
	 * aligned to BODY*SIGMA, and 64 | BODY*SIGMA or BODY*SIGMA | 64) */
	static constexpr int off(int u) { return u * SIGMA + ((u * SIGMA) >> 6); }
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

/*
 * First pass (stages 0..G-1): inputs come straight from HBM (staged int16
 * indices, unpacked with the row's val), outputs go to the LDS tile.  A body is
 * exactly two tile rows of one residue (2^G elements per row).
 */
template <int L, int G>
__device__ __forceinline__ void first_pass(uint32_t *tile, const int32_t *rowval, const int16_t *src,
					   const int row_first, const int nrows, const int tid)
{
	using P = PassGeo<L, 0, G>;
	constexpr int U = P::U, BODY = P::BODY, SIGMA = P::SIGMA, COLS = P::COLS;
	constexpr int ROWS_PER_SEG = P::NJ / U;
	constexpr int NB = P::NJ / BODY;                        // bodies per walk
	constexpr bool WARM = P::NSEG > 1;                      // segments > 0 re-run the two rows in front of them

	/* rowval[lr + 2] = +-val of tile row lr, 0 for rows that do not exist (also lr = -2, -1):
	 * a missing row is loaded from a clamped address and multiplied by 0, no predication */
	const int last_row = nrows - 1;
	/* lowest row any lane may touch: segment 0's (zero-weighted) warm-up sits two rows above the tile */
	const int base_row = row_first - 2 < 0 ? 0 : (row_first - 2 > last_row ? last_row : row_first - 2);
	const int16_t *tbase = src + ((size_t)base_row << L);   /* wave-uniform; per-lane offsets stay 32-bit */

#pragma unroll 1
	for (int r = 0; r < P::RPT; r++) {
		const int seg = P::MULTI_RES ? 0 : tid / SIGMA;
		const int i = P::MULTI_RES ? tid + r * NT : tid % SIGMA;
		const int lr_seg = seg * ROWS_PER_SEG;

		/* every staged index of this walk, issued back to back (one HBM round trip) */
		int32_t raw[(NB + (WARM ? 1 : 0)) * BODY];
#pragma unroll
		for (int b = (WARM ? -1 : 0); b < NB; b++) {
#pragma unroll
			for (int half = 0; half < 2; half++) {
				const int lr = lr_seg + 2 * b + half;
				int rho = row_first + lr;
				rho = rho < 0 ? 0 : (rho > last_row ? last_row : rho);
				const unsigned off = ((unsigned)(rho - base_row) << L) + (unsigned)i;
#pragma unroll
				for (int q = 0; q < U; q++)
					raw[(b + (WARM ? 1 : 0)) * BODY + half * U + q] = (int32_t)tbase[off + q * SIGMA];
			}
		}

		uint32_t h[G][U];
		clear_hist<G>(h);
#pragma unroll
		for (int b = (WARM ? -1 : 0); b < NB; b++) {
			const int lr0 = lr_seg + 2 * b;
			const int32_t v0 = rowval[lr0 + 2], v1 = rowval[lr0 + 3];
			uint32_t v[BODY];
#pragma unroll
			for (int u = 0; u < BODY; u++)
				v[u] = (uint32_t)__mul24(raw[(b + (WARM ? 1 : 0)) * BODY + u], u < U ? v0 : v1);
			constexpr uint32_t ONE = 1u << OutScale<L>::SHIFT;
			const uint32_t b0 = (i == 0 && lr0 >= 0 && row_first + lr0 >= 0) ? ONE : 0u;
			const uint32_t b1 = (i == 0 && lr0 + 1 >= 0 && row_first + lr0 + 1 >= 0) ? ONE : 0u;
			pass_body<L, 0, G>(v, h, b0, b1);
			if (b >= 0) {
				uint32_t *o = tile + lds_at(lr0 * COLS + i);
#pragma unroll
				for (int u = 0; u < BODY; u++)
					o[P::off(u)] = v[u];
			}
		}
	}
}

/*
 * Middle / last passes (stages K0..K0+G-1), in place on the LDS tile.
 * LAST: the outputs are final values; they are converted to 16-bit samples,
 * packed two per dword and parked at the start of the thread's own (already
 * consumed) segment: sample e of thread `tid` -> dword lds_at(tid*NJ) + e/2.
 */
template <int L, int K0, int G, bool LAST>
__device__ __forceinline__ void lds_pass(uint32_t *tile, const int tid, const unsigned fmt)
{
	using P = PassGeo<L, K0, G>;
	constexpr int U = P::U, BODY = P::BODY, SIGMA = P::SIGMA;
	static_assert(!P::MULTI_RES, "only the first pass may own several residues");
	static_assert(!LAST || SIGMA == 1, "last pass must end at stride 1");

	const int seg = tid / SIGMA;
	const int i = tid % SIGMA;
	const int m_seg = seg * P::NJ * SIGMA + i;              // first element of this thread's walk
	const PcmFmt pf = make_pcm_fmt<L>(fmt);
	uint32_t h[G][U];
	clear_hist<G>(h);

	/* warm-up: the BODY elements in front of this segment belong to the previous
	 * segment's owner, who is about to overwrite them in place - read them
	 * first, then everybody may start walking */
	__syncthreads();
	uint32_t w[BODY];
	{
		const uint32_t *pw = tile + lds_at(seg ? m_seg - BODY * SIGMA : m_seg);
#pragma unroll
		for (int u = 0; u < BODY; u++)
			w[u] = seg ? pw[P::off(u)] : 0u;
	}
	__syncthreads();
	pass_body<L, K0, G>(w, h, 0u, 0u);

#pragma unroll
	for (int it = 0; it < P::NJ / BODY; it++) {
		uint32_t *p = tile + lds_at(m_seg + it * BODY * SIGMA);
		uint32_t v[BODY];
#pragma unroll
		for (int u = 0; u < BODY; u++)
			v[u] = p[P::off(u)];
		pass_body<L, K0, G>(v, h, 0u, 0u);
		if constexpr (!LAST) {
#pragma unroll
			for (int u = 0; u < BODY; u++)
				p[P::off(u)] = v[u];
		} else {
			uint32_t *o = tile + lds_at(m_seg) + it * (BODY / 2);
#pragma unroll
			for (int u = 0; u < BODY; u += 2)
				o[u / 2] = pack_pcm<L>(v[u], v[u + 1], pf);
		}
	}
}

/* stage grouping per level: G <= 3 keeps a body at 16 elements */
template <int L> struct Plan;
#define ACM_PLAN(LV, FIRST_G, ...) \
	template <> struct Plan<LV> { \
		static constexpr int G0 = FIRST_G; \
		static __device__ __forceinline__ void rest(uint32_t *t, int tid, unsigned fmt) { __VA_ARGS__ } \
	};
ACM_PLAN(5, 3, lds_pass<5, 3, 2, true>(t, tid, fmt);)
ACM_PLAN(6, 3, lds_pass<6, 3, 3, true>(t, tid, fmt);)
ACM_PLAN(7, 3, lds_pass<7, 3, 2, false>(t, tid, fmt); lds_pass<7, 5, 2, true>(t, tid, fmt);)
ACM_PLAN(8, 3, lds_pass<8, 3, 3, false>(t, tid, fmt); lds_pass<8, 6, 2, true>(t, tid, fmt);)
ACM_PLAN(9, 3, lds_pass<9, 3, 3, false>(t, tid, fmt); lds_pass<9, 6, 3, true>(t, tid, fmt);)
ACM_PLAN(10, 3, lds_pass<10, 3, 3, false>(t, tid, fmt); lds_pass<10, 6, 2, false>(t, tid, fmt); lds_pass<10, 8, 2, true>(t, tid, fmt);)
ACM_PLAN(11, 3, lds_pass<11, 3, 3, false>(t, tid, fmt); lds_pass<11, 6, 3, false>(t, tid, fmt); lds_pass<11, 9, 2, true>(t, tid, fmt);)
#undef ACM_PLAN

template <int L>
__global__ void __launch_bounds__(NT, 2)
acm_fused_tile(const AcmDevStream *__restrict__ streams, const AcmTile *__restrict__ tiles,
	       const int16_t *__restrict__ idx, const acmhip_blkhdr *__restrict__ hdr,
	       int16_t *__restrict__ pcm, unsigned fmt)
{
	constexpr int COLS = 1 << L;
	constexpr int NELEM = (L >= 11) ? 32768 : 16384;
	constexpr int TR = NELEM / COLS;                        // tile rows incl. 2 halo rows
	constexpr int NJ_LAST = NELEM / NT;                     // samples per thread in the last pass
	constexpr bool NEG_ODD_ROWS = StageKind<L, 0>::N;       // stage 0 wants odd tile rows negated

	__shared__ uint32_t tile[NELEM + NELEM / 64];
	__shared__ int32_t rowval[TR + 2];                      // [lr + 2]; two leading zeros for the warm-up of segment 0

	const int tid = threadIdx.x;
	const AcmTile tl = tiles[blockIdx.x];
	const AcmDevStream s = streams[tl.stream];
	const int row_first = (int)tl.row0 - 2;                 // stream row of tile row 0 (may be < 0)
	const int nrows = (int)s.nrows;

	/* per tile row: the block's val (decode.c:589), signed per the stage-0 convention */
	for (int lr = tid - 2; lr < TR; lr += NT) {
		const int rho = row_first + lr;
		int32_t v = 0;
		if (lr >= 0 && rho >= 0 && rho < nrows) {
			v = (int32_t)(hdr[s.hdr_off + (uint32_t)rho / s.rows].val << OutScale<L>::SHIFT);
			if (NEG_ODD_ROWS && (lr & 1))
				v = -v;
		}
		rowval[lr + 2] = v;
	}
	__syncthreads();

	first_pass<L, Plan<L>::G0>(tile, rowval, idx + s.idx_off, row_first, nrows, tid);
	Plan<L>::rest(tile, tid, fmt);
	__syncthreads();

	/* write-out of the payload rows (tile rows 2..TR-1): 8 samples (16 B) per lane per step,
	 * gathered from the per-thread parking areas of the last pass */
	uint16_t *dst = reinterpret_cast<uint16_t *>(pcm) + s.pcm_off;
	for (int vec = tid; vec < (TR - 2) * COLS / 8; vec += NT) {
		const int ml = 2 * COLS + vec * 8;
		const int lr = ml >> L;
		const int col = ml & (COLS - 1);
		const int rho = row_first + lr;
		if (rho >= nrows)
			break;
		const uint64_t g = ((uint64_t)(uint32_t)(rho - (int)s.row_begin) << L) + (uint32_t)col;
		if (g >= s.n_emit)
			break;
		const int owner = ml / NJ_LAST;
		const uint32_t *q = tile + lds_at(owner * NJ_LAST) + (ml % NJ_LAST) / 2;
		uint4 o;
		o.x = q[0];
		o.y = q[1];
		o.z = q[2];
		o.w = q[3];
		if (g + 8 <= s.n_emit) {
			*reinterpret_cast<uint4 *>(dst + g) = o;
		} else {
			const uint32_t w[4] = { o.x, o.y, o.z, o.w };
			for (int e = 0; e < 8 && g + e < s.n_emit; e++)
				dst[g + e] = (uint16_t)(w[e >> 1] >> ((e & 1) * 16));
		}
	}
}

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

extern "C" int acmk_launch_fused(uint32_t level, const AcmDevStream *d_streams, const AcmTile *d_tiles,
				 uint32_t ntiles, const int16_t *d_idx, const acmhip_blkhdr *d_hdr,
				 int16_t *d_pcm, unsigned fmt, void *stream)
{
	hipStream_t st = (hipStream_t)stream;
	if (ntiles == 0)
		return 0;
	switch (level) {
#define ACMK_CASE(LV) case LV: hipLaunchKernelGGL(acm_fused_tile<LV>, dim3(ntiles), dim3(NT), 0, st, d_streams, d_tiles, d_idx, d_hdr, d_pcm, fmt); break;
	ACMK_CASE(5) ACMK_CASE(6) ACMK_CASE(7) ACMK_CASE(8) ACMK_CASE(9) ACMK_CASE(10) ACMK_CASE(11)
#undef ACMK_CASE
	default:
		return -1;
	}
	ACMK_CHECK_LAUNCH();
	return 0;
}

extern "C" int acmk_launch_unpack(const AcmDevStream *d_streams, const uint32_t *d_list, uint32_t nlist,
				  uint64_t max_elems, const int16_t *d_idx, const acmhip_blkhdr *d_hdr,
				  int32_t *d_x, void *stream)
{
	if (nlist == 0)
		return 0;
	hipLaunchKernelGGL(acm_sw_unpack, sw_grid(max_elems, nlist), dim3(SW_THREADS), 0, (hipStream_t)stream,
			   d_streams, d_list, d_idx, d_hdr, d_x);
	ACMK_CHECK_LAUNCH();
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
				 int32_t *d_out, void *stream)
{
	if (nlist == 0)
		return 0;
	hipLaunchKernelGGL(acm_sw_stage, sw_grid(max_elems, nlist), dim3(SW_THREADS), 0, (hipStream_t)stream,
			   d_streams, d_list, level, k, d_in, d_out);
	ACMK_CHECK_LAUNCH();
	return 0;
}

extern "C" int acmk_launch_emit(const AcmDevStream *d_streams, const uint32_t *d_list, uint32_t nlist,
				uint64_t max_emit, const int32_t *d_x, int16_t *d_pcm, unsigned fmt, void *stream)
{
	if (nlist == 0)
		return 0;
	hipLaunchKernelGGL(acm_sw_emit, sw_grid(max_emit, nlist), dim3(SW_THREADS), 0, (hipStream_t)stream,
			   d_streams, d_list, d_x, d_pcm, fmt);
	ACMK_CHECK_LAUNCH();
	return 0;
}
