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

/* sign-extended low half of a dword holding two staged int16 indices */
__device__ __forceinline__ int32_t lo16(int32_t w) { return (int32_t)(int16_t)(uint16_t)((uint32_t)w & 0xFFFFu); }

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

template <int L, int K0, int G>
struct PassGeo {
	static constexpr int COLS = 1 << L;
	static constexpr int NELEM = (L >= 11) ? 32768 : 16384;
	static constexpr int SIGMA = COLS >> (K0 + G);          // smallest stride of the pass
	static constexpr int U = 1 << G;
	static constexpr int BODY = 2 * U;                      // elements per unrolled body
	static constexpr int NJ_TOTAL = NELEM / SIGMA;          // walk length of one residue over the tile
	static constexpr bool MULTI_RES = SIGMA >= NT;          // more residues than threads
	static constexpr int RPT = MULTI_RES ? SIGMA / NT : 1;  // residues per thread
	static constexpr int NSEG = MULTI_RES ? 1 : NT / SIGMA; // walk segments per residue
	static constexpr int NJ = NJ_TOTAL / NSEG;              // walk length per thread
	static_assert(SIGMA >= 1, "pass exceeds level");
	static_assert(NJ % BODY == 0 && NJ >= BODY, "segment must be whole bodies");
};

/*
 * G butterfly stages over BODY consecutive elements of one residue class.
 * v: in = stage-K0 inputs, out = stage-(K0+G-1) outputs (conventions above).
 * h[t][x], x in [0, 2d): the 2d inputs of stage t that precede this body
 * (d = stride of stage t in walk units = 2^(G-1-t)); updated on exit.
 * bias: 1 for the thread owning residue 0 (the "+1" of decode.c:561-564), else 0.
 */
template <int L, int K0, int G>
__device__ __forceinline__ void pass_body(uint32_t (&v)[2 << G], uint32_t (&h)[G][1 << G], uint32_t bias)
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
			if (!kindN) {
				const uint32_t s = z2 + z0;
				y = beta ? sub_twice<EXACT32>(s, z1) : s + (z1 << 1);
			} else {
				const int aneg = (u >> (pb + 1)) & 1;
				if ((aneg ^ beta) == 0)
					y = sub_twice<EXACT32>(z0 - z2, z1);
				else
					y = (z2 - z0) + (z1 << 1);
			}
			if (K0 + t == 0 && (u % ((U / 2) > 0 ? (U / 2) : 1)) == 0)
				y += (kindN || !beta) ? bias : (0u - bias);
			v[u] = y;
		}
#pragma unroll
		for (int x = 0; x < 2 * d; x++)
			h[t][x] = in[BODY - 2 * d + x];
	}
}

template <int L, int K0, int G>
__device__ __forceinline__ void fused_pass(uint32_t *tile, const int tid)
{
	using P = PassGeo<L, K0, G>;
	constexpr int U = P::U, BODY = P::BODY, SIGMA = P::SIGMA;

	if constexpr (P::MULTI_RES) {
		__syncthreads();
#pragma unroll 1
		for (int r = 0; r < P::RPT; r++) {
			const int i = tid + r * NT;
			const uint32_t bias = (K0 == 0 && i == 0) ? 1u : 0u;
			uint32_t h[G][U];
#pragma unroll
			for (int t = 0; t < G; t++)
#pragma unroll
				for (int x = 0; x < U; x++)
					h[t][x] = 0u;
			uint32_t *p = tile + i;
#pragma unroll 1
			for (int it = 0; it < P::NJ / BODY; it++, p += BODY * SIGMA) {
				uint32_t v[BODY];
#pragma unroll
				for (int u = 0; u < BODY; u++)
					v[u] = p[u * SIGMA];
				pass_body<L, K0, G>(v, h, bias);
#pragma unroll
				for (int u = 0; u < BODY; u++)
					p[u * SIGMA] = v[u];
			}
		}
	} else {
		const int seg = tid / SIGMA;
		const int i = tid % SIGMA;
		const uint32_t bias = (K0 == 0 && i == 0) ? 1u : 0u;
		uint32_t *p = tile + (size_t)seg * P::NJ * SIGMA + i;
		uint32_t h[G][U];
#pragma unroll
		for (int t = 0; t < G; t++)
#pragma unroll
			for (int x = 0; x < U; x++)
				h[t][x] = 0u;

		/* warm-up: the BODY elements in front of this segment belong to the
		 * previous segment's owner, who is about to overwrite them in place -
		 * read them first, then everybody may start walking */
		__syncthreads();
		uint32_t w[BODY];
#pragma unroll
		for (int u = 0; u < BODY; u++)
			w[u] = seg ? p[(u - BODY) * SIGMA] : 0u;
		__syncthreads();
		pass_body<L, K0, G>(w, h, seg ? bias : 0u);

#pragma unroll 1
		for (int it = 0; it < P::NJ / BODY; it++, p += BODY * SIGMA) {
			uint32_t v[BODY];
#pragma unroll
			for (int u = 0; u < BODY; u++)
				v[u] = p[u * SIGMA];
			pass_body<L, K0, G>(v, h, bias);
#pragma unroll
			for (int u = 0; u < BODY; u++)
				p[u * SIGMA] = v[u];
		}
	}
}

/* stage grouping per level: G <= 3 keeps a body at 16 elements */
template <int L> __device__ __forceinline__ void run_passes(uint32_t *tile, int tid);
template <> __device__ __forceinline__ void run_passes<5>(uint32_t *t, int tid)  { fused_pass<5, 0, 3>(t, tid); fused_pass<5, 3, 2>(t, tid); }
template <> __device__ __forceinline__ void run_passes<6>(uint32_t *t, int tid)  { fused_pass<6, 0, 3>(t, tid); fused_pass<6, 3, 3>(t, tid); }
template <> __device__ __forceinline__ void run_passes<7>(uint32_t *t, int tid)  { fused_pass<7, 0, 3>(t, tid); fused_pass<7, 3, 2>(t, tid); fused_pass<7, 5, 2>(t, tid); }
template <> __device__ __forceinline__ void run_passes<8>(uint32_t *t, int tid)  { fused_pass<8, 0, 3>(t, tid); fused_pass<8, 3, 3>(t, tid); fused_pass<8, 6, 2>(t, tid); }
template <> __device__ __forceinline__ void run_passes<9>(uint32_t *t, int tid)  { fused_pass<9, 0, 3>(t, tid); fused_pass<9, 3, 3>(t, tid); fused_pass<9, 6, 3>(t, tid); }
template <> __device__ __forceinline__ void run_passes<10>(uint32_t *t, int tid) { fused_pass<10, 0, 3>(t, tid); fused_pass<10, 3, 3>(t, tid); fused_pass<10, 6, 2>(t, tid); fused_pass<10, 8, 2>(t, tid); }
template <> __device__ __forceinline__ void run_passes<11>(uint32_t *t, int tid) { fused_pass<11, 0, 3>(t, tid); fused_pass<11, 3, 3>(t, tid); fused_pass<11, 6, 3>(t, tid); fused_pass<11, 9, 2>(t, tid); }

template <int L>
__global__ void __launch_bounds__(NT)
acm_fused_tile(const AcmDevStream *__restrict__ streams, const AcmTile *__restrict__ tiles,
	       const int16_t *__restrict__ idx, const acmhip_blkhdr *__restrict__ hdr,
	       int16_t *__restrict__ pcm, unsigned fmt)
{
	constexpr int COLS = 1 << L;
	constexpr int NELEM = (L >= 11) ? 32768 : 16384;
	constexpr int TR = NELEM / COLS;                        // tile rows incl. 2 halo rows
	constexpr bool NEG_ODD_ROWS = StageKind<L, 0>::N;       // stage 0 wants odd tile rows negated

	__shared__ __attribute__((aligned(16))) uint32_t tile[NELEM];
	__shared__ int32_t rowval[TR];

	const int tid = threadIdx.x;
	const AcmTile tl = tiles[blockIdx.x];
	const AcmDevStream s = streams[tl.stream];
	const int row_first = (int)tl.row0 - 2;                 // stream row of tile row 0 (may be < 0)
	const int nrows = (int)s.nrows;

	/* per tile row: the block's val (decode.c:589), signed per the stage-0 convention */
	for (int lr = tid; lr < TR; lr += NT) {
		const int rho = row_first + lr;
		int32_t v = 0;
		if (rho >= 0 && rho < nrows) {
			v = (int32_t)hdr[s.hdr_off + (uint32_t)rho / s.rows].val;
			if (NEG_ODD_ROWS && (lr & 1))
				v = -v;
		}
		rowval[lr] = v;
	}
	__syncthreads();

	/* load + unpack: 8 staged indices (16 B) per lane per step */
	const int16_t *src = idx + s.idx_off;
	for (int vec = tid; vec < NELEM / 8; vec += NT) {
		const int ml = vec * 8;
		const int lr = ml >> L;
		const int col = ml & (COLS - 1);
		const int rho = row_first + lr;
		int4 raw = make_int4(0, 0, 0, 0);
		if (rho >= 0 && rho < nrows)
			raw = *reinterpret_cast<const int4 *>(src + ((size_t)rho << L) + col);
		const int32_t val = rowval[lr];
		uint4 lo, hi;
		lo.x = (uint32_t)__mul24(lo16(raw.x), val);
		lo.y = (uint32_t)__mul24(raw.x >> 16, val);
		lo.z = (uint32_t)__mul24(lo16(raw.y), val);
		lo.w = (uint32_t)__mul24(raw.y >> 16, val);
		hi.x = (uint32_t)__mul24(lo16(raw.z), val);
		hi.y = (uint32_t)__mul24(raw.z >> 16, val);
		hi.z = (uint32_t)__mul24(lo16(raw.w), val);
		hi.w = (uint32_t)__mul24(raw.w >> 16, val);
		*reinterpret_cast<uint4 *>(&tile[ml]) = lo;
		*reinterpret_cast<uint4 *>(&tile[ml + 4]) = hi;
	}

	run_passes<L>(tile, tid);
	__syncthreads();

	/* write-out of the payload rows (tile rows 2..TR-1), 8 samples (16 B) per lane per step */
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
		const uint4 a = *reinterpret_cast<const uint4 *>(&tile[ml]);
		const uint4 b = *reinterpret_cast<const uint4 *>(&tile[ml + 4]);
		uint32_t w[8];
		w[0] = pcm16((int32_t)a.x, L, fmt); w[1] = pcm16((int32_t)a.y, L, fmt);
		w[2] = pcm16((int32_t)a.z, L, fmt); w[3] = pcm16((int32_t)a.w, L, fmt);
		w[4] = pcm16((int32_t)b.x, L, fmt); w[5] = pcm16((int32_t)b.y, L, fmt);
		w[6] = pcm16((int32_t)b.z, L, fmt); w[7] = pcm16((int32_t)b.w, L, fmt);
		if (g + 8 <= s.n_emit) {
			uint4 o;
			o.x = w[0] | (w[1] << 16);
			o.y = w[2] | (w[3] << 16);
			o.z = w[4] | (w[5] << 16);
			o.w = w[6] | (w[7] << 16);
			*reinterpret_cast<uint4 *>(dst + g) = o;
		} else {
			for (int e = 0; e < 8 && g + e < s.n_emit; e++)
				dst[g + e] = (uint16_t)w[e];
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
