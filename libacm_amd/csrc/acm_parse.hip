/*
 * acm_parse.hip - optional DEVICE-side bit parsing for batches.
 *
 * An ACM bitstream is sequential: where a column starts is only known once every column before it has
 * been walked (variable-length fillers, no index; /root/reference/src/decode.c:478-502).  But WALKING is
 * much cheaper than DECODING: a linear or ternary column has a length known from its 5-bit code alone
 * (decode.c:196-215, :400-476), only the eight k-fillers (:217-398) have to be stepped symbol by symbol,
 * and nothing is stored except one bit offset per column.  So the work is split in two kernels:
 *
 *   acm_parse_scan     one LANE per stream walks its stream and writes colpos[block][column] (the bit offset
 *                      of every column's 5-bit code) and the block headers.  Sequential per stream, light.
 *                      Up to 2048 streams: acm_parse_scan_lone, one stream per WAVEFRONT - then the walk is
 *                      wave-uniform and runs on the scalar unit (s_load + SALU), ~1.7x faster per stream.
 *   acm_parse_columns  one LANE per COLUMN decodes `rows` indices from its bit offset.  64 adjacent columns
 *                      per wavefront, all lanes produce row r in the same iteration, so every store is a
 *                      contiguous run of the row-major staged form the synthesis kernels read.  Symbols of
 *                      the k/t fillers come out of a 7-bit look-up table in LDS: one code path for all of
 *                      them, so lanes holding different fillers do not serialise.
 *
 * The device handles the CLEAN path only.  Anything it is not sure to reproduce bit for bit - data
 * running out, an invalid filler code (decode.c:190-194), a ternary symbol out of range (:412, :438, :464),
 * an index outside the block's amplitude range (hazard H1) - flags the stream, and the host re-parses that
 * stream with the exact reader (acm_fill.cpp).
 */
#include <hip/hip_runtime.h>

#include "acm_device.h"

namespace {

constexpr int SCAN_THREADS = 64;
constexpr uint32_t K_WALK = 0xFFFFFFFEu, BAD_CODE = 0xFFFFFFFFu;
constexpr int COL_THREADS = 256;

/* ---- bit window over a file image (arena slots are 16-byte aligned with >= 16 zero bytes behind the file) ---- */
struct DevBits {
	const uint32_t *w;
	uint32_t bit;            /* next unread bit (files are < 256 MiB) */
	uint32_t have;
	uint64_t win;

	__device__ __forceinline__ void seek(const uint32_t *base, uint32_t to)
	{
		w = base;
		bit = to;
		have = 0;
		win = 0;
	}
	__device__ __forceinline__ void refill()
	{
		const uint32_t i = bit >> 5, sh = bit & 31;
		const uint64_t two = ((uint64_t)w[i + 1] << 32) | w[i];
		win = two >> sh;
		have = 64 - sh;                 /* >= 33 */
	}
	__device__ __forceinline__ void need(uint32_t n)        /* n <= 32 */
	{
		if (have < n)
			refill();
	}
	__device__ __forceinline__ void drop(uint32_t n)        /* n <= have */
	{
		win >>= n;
		have -= n;
		bit += n;
	}
	__device__ __forceinline__ uint32_t get(uint32_t n)     /* n <= 16 */
	{
		need(n);
		const uint32_t v = (uint32_t)win & ((1u << n) - 1);
		drop(n);
		return v;
	}
	__device__ __forceinline__ void skip(uint32_t n)
	{
		bit += n;
		have = 0;
	}
};

/* ---- filler classes ---- */
enum { CLS_ZERO = 0, CLS_LINEAR, CLS_TERN, CLS_K, CLS_BAD };

constexpr uint64_t class_word(int first)        /* 3 bits per code, 16 codes per word */
{
	uint64_t v = 0;
	for (int c = first; c < first + 16; c++) {
		uint64_t k = CLS_BAD;
		if (c == 0)
			k = CLS_ZERO;
		else if (c >= 3 && c <= 16)
			k = CLS_LINEAR;
		else if (c == 19 || c == 22 || c == 29)
			k = CLS_TERN;
		else if (c == 17 || c == 18 || c == 20 || c == 21 || c == 23 || c == 24 || c == 26 || c == 27)
			k = CLS_K;
		v |= k << (3 * (c - first));
	}
	return v;
}

__device__ __forceinline__ uint32_t code_class(uint32_t code)
{
	constexpr uint64_t lo = class_word(0), hi = class_word(16);
	return (uint32_t)(((code & 16) ? hi : lo) >> (3 * (code & 15))) & 7u;
}

/* k-fillers: the first three bits of a symbol fix its length and whether it stands for two rows ("0" of the
 * x3/x4/x5 family, decode.c:217-398).  One nibble per 3-bit prefix: len | two << 3. */
constexpr uint32_t k_prefix_table(int code)
{
	uint32_t t = 0;
	for (int p = 0; p < 8; p++) {
		const bool b0 = p & 1, b1 = p & 2, b2 = p & 4;
		uint32_t len = 1, two = 0;
		switch (code) {
		case 17: if (!b0) two = 1; else len = b1 ? 3 : 2; break;                 /* k13: 0 | 10 | 11s */
		case 18: len = b0 ? 2 : 1; break;                                        /* k12: 0 | 1s */
		case 20: if (!b0) two = 1; else len = b1 ? 4 : 2; break;                 /* k24: 0 | 10 | 11nn */
		case 21: len = b0 ? 3 : 1; break;                                        /* k23: 0 | 1nn */
		case 23: if (!b0) two = 1; else len = !b1 ? 2 : !b2 ? 4 : 5; break;      /* k35: 0 | 10 | 110s | 111ff */
		case 24: len = !b0 ? 1 : !b1 ? 3 : 4; break;                             /* k34: 0 | 10s | 11ff */
		case 26: if (!b0) two = 1; else len = b1 ? 5 : 2; break;                 /* k45: 0 | 10 | 11www */
		case 27: len = b0 ? 4 : 1; break;                                        /* k44: 0 | 1www */
		default: break;
		}
		t |= (len | two << 3) << (4 * p);
	}
	return t;
}

__device__ __forceinline__ uint32_t k_table_for(uint32_t code)
{
	switch (code) {
	case 17: return k_prefix_table(17);
	case 18: return k_prefix_table(18);
	case 20: return k_prefix_table(20);
	case 21: return k_prefix_table(21);
	case 23: return k_prefix_table(23);
	case 24: return k_prefix_table(24);
	case 26: return k_prefix_table(26);
	default: return k_prefix_table(27);
	}
}

/* ---- kernel 1: walk the streams ---- */

/* payload bits of a column by filler code for `rows` rows; K_WALK = step through it, BAD_CODE = stop */
__device__ __forceinline__ uint32_t column_bits(uint32_t code, uint32_t rows)
{
	const uint32_t cls = code_class(code);
	return cls == CLS_ZERO ? 0u : cls == CLS_LINEAR ? rows * code : cls == CLS_K ? K_WALK : cls == CLS_BAD ? BAD_CODE :
	       code == 19 ? (rows + 2) / 3 * 5 : code == 22 ? (rows + 2) / 3 * 7 : (rows + 1) / 2 * 7;
}

/*
 * One stream.  LONE = this wavefront has no other stream: everything about the walk is then wave-uniform, the
 * compiler keeps it on the scalar unit (s_load through the scalar cache for the window, SALU for the bit
 * arithmetic) and a step costs a few cycles instead of the ~8 per dependent VALU instruction plus an exposed vector
 * memory round trip per window refill.  Otherwise `collen` is this lane's 32-entry table in LDS.
 */
template <bool LONE>
__device__ __forceinline__ void scan_stream(const AcmParseJob &job, const uint8_t *__restrict__ files, uint32_t *__restrict__ colpos,
					    acmhip_blkhdr *__restrict__ hdr, AcmParseResult *__restrict__ out, const uint32_t *collen)
{
	const uint32_t rows = job.rows, cols = 1u << job.level;
	const uint32_t safe = job.file_len * 8u;                /* bits that really belong to the file */

	DevBits bs;
	bs.seek(reinterpret_cast<const uint32_t *>(files + job.file_off), job.data_start * 8u);

	uint32_t done = 0, status = 0;
	uint32_t *cp = colpos + job.col_off;
	for (uint32_t b = 0; b < job.blocks; b++) {
		if (bs.bit + 20 > safe) {
			status = 1;
			break;
		}
		const uint32_t pwr = bs.get(4);
		const uint32_t val = bs.get(16);
		for (uint32_t c = 0; c < cols; c++) {
			if (bs.bit + 5 > safe) {
				status = 1;
				break;
			}
			cp[c] = bs.bit;
			const uint32_t code = bs.get(5);
			const uint32_t len = LONE ? column_bits(code, rows) : collen[code];
			if (len < K_WALK) {
				bs.skip(len);
			} else if (len == K_WALK) {
				const uint32_t tab = k_table_for(code);
				uint32_t r = 0;
				/* four symbols per trip (<= 5 bits each), predicated on the rows left: the loop control and the
				 * window check are paid once per trip */
				while (r < rows && bs.bit < safe) {
					bs.need(20);
#pragma unroll
					for (int u = 0; u < 4; u++) {
						const bool go = r < rows;
						const uint32_t e = (tab >> (((uint32_t)bs.win & 7u) * 4)) & 15u;
						const uint32_t len1 = go ? (e & 7u) : 0u;
						bs.drop(len1);
						r += go ? 1 + (e >> 3) : 0u;
					}
				}
				if (r < rows)
					status = 1;                     /* ran out of data inside the column */
			} else {
				status = 1;
			}
			if (status || bs.bit > safe) {                  /* the column must end inside the file */
				status = 1;
				break;
			}
		}
		if (status)
			break;
		hdr[job.hdr_off + b] = acmhip_blkhdr{ val, pwr };
		cp += cols;
		done++;
	}
	*out = AcmParseResult{ done, status };
}

__global__ void __launch_bounds__(SCAN_THREADS)
acm_parse_scan(const AcmParseJob *__restrict__ jobs, uint32_t njobs, const uint8_t *__restrict__ files,
	       uint32_t *__restrict__ colpos, acmhip_blkhdr *__restrict__ hdr, AcmParseResult *__restrict__ res)
{
	/* Streams are dealt out across wavefronts first, lanes second: a walk is a chain of dependent steps and
	 * lanes of one wavefront serialise each other's branches. */
	const uint32_t j = blockIdx.x + threadIdx.x * gridDim.x;
	if (j >= njobs)
		return;
	const AcmParseJob job = jobs[j];
	extern __shared__ uint32_t scan_lds[];
	uint32_t *collen = scan_lds + threadIdx.x * 33;
	for (uint32_t code = 0; code < 32; code++)
		collen[code] = column_bits(code, job.rows);
	scan_stream<false>(job, files, colpos, hdr, res + j, collen);
}

/* few streams: one per wavefront, walked on the scalar unit */
__global__ void __launch_bounds__(SCAN_THREADS)
acm_parse_scan_lone(const AcmParseJob *__restrict__ jobs, uint32_t njobs, const uint8_t *__restrict__ files,
		    uint32_t *__restrict__ colpos, acmhip_blkhdr *__restrict__ hdr, AcmParseResult *__restrict__ res)
{
	if (threadIdx.x != 0 || blockIdx.x >= njobs)
		return;
	const AcmParseJob job = jobs[blockIdx.x];
	scan_stream<true>(job, files, colpos, hdr, res + blockIdx.x, nullptr);
}

/* ---- kernel 2: decode the columns ---- */

/* One look-up entry per (k/t filler, next 7 bits): the symbol at the head of those bits.
 *   bits 0-2 length, 3-4 values produced (1..3), 5-8 / 9-12 / 13-16 the values + 8, bit 17 invalid symbol. */
constexpr int LUT_CLASSES = 11;
__device__ __forceinline__ int lut_class(uint32_t code)    /* 17..24, 26, 27, 29 -> 0..10 */
{
	return code <= 24 ? (int)code - 17 : code <= 27 ? (int)code - 18 : 10;
}

__device__ uint32_t lut_entry(uint32_t code, uint32_t bits)
{
	const bool b0 = bits & 1, b1 = bits & 2, b2 = bits & 4;
	uint32_t len = 1, cnt = 1, bad = 0;
	int v0 = 0, v1 = 0, v2 = 0;
	auto sign1 = [](uint32_t b) { return b ? 1 : -1; };
	auto near2 = [](uint32_t b) { return b < 2 ? (int)b - 2 : (int)b - 1; };        /* -2 -1 +1 +2 */
	auto far2 = [](uint32_t b) { return b < 2 ? (int)b - 3 : (int)b; };             /* -3 -2 +2 +3 */
	auto wide3 = [](uint32_t b) { return b < 4 ? (int)b - 4 : (int)b - 3; };        /* -4..-1 +1..+4 */
	switch (code) {
	case 17:
		if (!b0) cnt = 2; else if (!b1) len = 2; else { len = 3; v0 = sign1(b2); }
		break;
	case 18:
		if (b0) { len = 2; v0 = sign1(b1); }
		break;
	case 19: {
		const uint32_t b = bits & 31;
		len = 5; cnt = 3; bad = b >= 27;
		v0 = (int)(b % 3) - 1; v1 = (int)(b / 3 % 3) - 1; v2 = (int)(b / 9 % 3) - 1;
		break;
	}
	case 20:
		if (!b0) cnt = 2; else if (!b1) len = 2; else { len = 4; v0 = near2((bits >> 2) & 3); }
		break;
	case 21:
		if (b0) { len = 3; v0 = near2((bits >> 1) & 3); }
		break;
	case 22: {
		const uint32_t b = bits & 127;
		len = 7; cnt = 3; bad = b >= 125;
		v0 = (int)(b % 5) - 2; v1 = (int)(b / 5 % 5) - 2; v2 = (int)(b / 25 % 5) - 2;
		break;
	}
	case 23:
		if (!b0) cnt = 2; else if (!b1) len = 2;
		else if (!b2) { len = 4; v0 = sign1(bits & 8); }
		else { len = 5; v0 = far2((bits >> 3) & 3); }
		break;
	case 24:
		if (b0) {
			if (!b1) { len = 3; v0 = sign1(b2); }
			else { len = 4; v0 = far2((bits >> 2) & 3); }
		}
		break;
	case 26:
		if (!b0) cnt = 2; else if (!b1) len = 2; else { len = 5; v0 = wide3((bits >> 2) & 7); }
		break;
	case 27:
		if (b0) { len = 4; v0 = wide3((bits >> 1) & 7); }
		break;
	default: {      /* 29 */
		const uint32_t b = bits & 127;
		len = 7; cnt = 2; bad = b >= 121;
		v0 = (int)(b % 11) - 5; v1 = (int)(b / 11 % 11) - 5;
		break;
	}
	}
	return len | cnt << 3 | (uint32_t)(v0 + 8) << 5 | (uint32_t)(v1 + 8) << 9 | (uint32_t)(v2 + 8) << 13 | bad << 17;
}

__global__ void __launch_bounds__(COL_THREADS)
acm_parse_columns(const AcmParseJob *__restrict__ jobs, const AcmParseResult *__restrict__ res,
		  const uint8_t *__restrict__ files, const uint32_t *__restrict__ colpos,
		  const acmhip_blkhdr *__restrict__ hdr, int16_t *__restrict__ idx, uint32_t *__restrict__ flags)
{
	__shared__ uint32_t lut[LUT_CLASSES * 128];
	{
		constexpr uint32_t codes[LUT_CLASSES] = { 17, 18, 19, 20, 21, 22, 23, 24, 26, 27, 29 };
		for (uint32_t e = threadIdx.x; e < LUT_CLASSES * 128; e += COL_THREADS)
			lut[e] = lut_entry(codes[e >> 7], e & 127);
		__syncthreads();
	}
	const AcmParseJob job = jobs[blockIdx.y];
	const AcmParseResult rs = res[blockIdx.y];
	if (rs.status != 0)
		return;                                         /* the host redoes the whole stream */
	const uint32_t rows = job.rows, level = job.level, cols = 1u << level;
	const uint32_t ncol = rs.blocks_done << level;          /* blocks * cols < 2^32 (acmk_parse_supported) */
	const uint64_t bl = (uint64_t)rows << level;
	const uint32_t *base = reinterpret_cast<const uint32_t *>(files + job.file_off);
	uint32_t bad = 0;

	for (uint32_t g = blockIdx.x * COL_THREADS + threadIdx.x; g < ncol; g += gridDim.x * COL_THREADS) {
		const uint32_t b = g >> level, c = g & (cols - 1);
		const int lim = 1 << hdr[job.hdr_off + b].pwr;
		int16_t *out = idx + job.idx_off + (uint64_t)b * bl + c;
		DevBits bs;
		bs.seek(base, colpos[job.col_off + g]);
		const uint32_t code = bs.get(5);
		const uint32_t cls = code_class(code);
		const bool table = cls >= CLS_TERN;             /* CLS_BAD cannot occur: the scan flagged the stream */
		const uint32_t lbase = table ? (uint32_t)lut_class(code) * 128u : 0u;
		const uint32_t width = table ? 0u : code;       /* zero filler = linear with no bits */
		const uint32_t mask = (1u << width) - 1u;
		const int mid = width ? 1 << (width - 1) : 0;
		uint32_t pend = 0, npend = 0;
		for (uint32_t r = 0; r < rows; r++) {
			int v;
			if (npend) {
				v = (int)(pend & 15u) - 8;
				pend >>= 4;
				npend--;
			} else {
				bs.need(16);
				const uint32_t raw = (uint32_t)bs.win;
				if (table) {
					const uint32_t e = lut[lbase + (raw & 127u)];
					v = (int)((e >> 5) & 15u) - 8;
					pend = e >> 9;
					npend = ((e >> 3) & 3u) - 1;
					bad |= (e >> 17) & 1u;
					bs.drop(e & 7u);
				} else {
					v = (int)(raw & mask) - mid;
					bs.drop(width);
				}
			}
			bad |= (v >= lim) | (v < -lim);                 /* hazard H1: the host resolves stale-table reads */
			out[(uint64_t)r << level] = (int16_t)v;
		}
	}
	if (bad)
		atomicOr(&flags[blockIdx.y], 1u);
}

} // namespace

#define ACMP_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

extern "C" int acmk_parse_supported(uint32_t level, uint32_t rows, uint64_t file_len, uint64_t blocks)
{
	/* 32-bit bit offsets with headroom (a corrupt stream may step one column past the end of its file before
	 * the walk notices, and must not wrap), 32-bit column counts */
	return rows >= 1 && blocks >= 1 && file_len < 0x10000000ull && (blocks << level) < 0xFFFFFFFFull;
}

/*
 * d_flags[njobs] must be zero on entry; after the kernels a stream is clean iff
 * d_res[j].status == 0 && d_res[j].blocks_done == jobs[j].blocks && d_flags[j] == 0.
 */
extern "C" int acmk_launch_parse(const AcmParseJob *d_jobs, uint32_t njobs, const uint8_t *d_files,
				 uint32_t *d_colpos, int16_t *d_idx, acmhip_blkhdr *d_hdr,
				 AcmParseResult *d_res, uint32_t *d_flags, uint64_t max_columns, void *stream)
{
	if (njobs == 0)
		return 0;
	hipStream_t st = (hipStream_t)stream;
	const uint32_t full = (njobs + SCAN_THREADS - 1) / SCAN_THREADS;
	const uint32_t scan_waves = njobs < 8192u ? njobs : full < 8192u ? 8192u : full;
	const uint32_t scan_lanes = (njobs + scan_waves - 1) / scan_waves;
	/* one scalar unit serves the four SIMDs of a CU: the scalar walk wins while there are at most ~8 streams per CU
	 * (17 ms against 30 ms for 1024 streams of 512 K samples; level at ~3000 streams; behind at 8192) */
	if (njobs <= 2048)
		hipLaunchKernelGGL(acm_parse_scan_lone, dim3(njobs), dim3(SCAN_THREADS), 0, st,
				   d_jobs, njobs, d_files, d_colpos, d_hdr, d_res);
	else
		hipLaunchKernelGGL(acm_parse_scan, dim3(scan_waves), dim3(SCAN_THREADS), scan_lanes * 33 * sizeof(uint32_t), st,
				   d_jobs, njobs, d_files, d_colpos, d_hdr, d_res);
	ACMP_CHECK();
	uint64_t gx = (max_columns + COL_THREADS - 1) / COL_THREADS;
	if (gx < 1)
		gx = 1;
	if (gx > 2048)
		gx = 2048;
	for (uint32_t at = 0; at < njobs; at += 65535) {
		const uint32_t n = njobs - at < 65535 ? njobs - at : 65535;
		hipLaunchKernelGGL(acm_parse_columns, dim3((unsigned)gx, n), dim3(COL_THREADS), 0, st,
				   d_jobs + at, d_res + at, d_files, d_colpos, d_hdr, d_idx, d_flags + at);
		ACMP_CHECK();
	}
	return 0;
}
