/*
 * acm_parse.hip - optional DEVICE-side bit parsing for big batches.
 *
 * The bitstream of one ACM stream is strictly sequential (variable-length fillers, no block index), so
 * the only parallelism is across streams: one LANE per stream.  That only beats a many-core host when
 * there are thousands of streams (BASELINE configs[4]: 65 536 streams), which is when the batch front
 * end selects it; the host parser (acm_fill.cpp) stays the reference implementation and the fallback.
 *
 * The device parser handles the CLEAN path only.  Anything it is not sure to reproduce bit for bit -
 * running out of data, an invalid filler code (/root/reference/src/decode.c:190-194), a ternary symbol out
 * of range (:412, :438, :464), an index outside the block's amplitude range (hazard H1) - makes it stop and
 * flag the stream; the host then re-parses that stream with the exact reader.
 *
 * Output of the parse kernel is column-major per block (each lane writes sequentially); a second kernel
 * transposes blocks through LDS into the row-major staged form the synthesis kernels read.
 */
#include <hip/hip_runtime.h>

#include "acm_device.h"

namespace {

constexpr int PARSE_THREADS = 64;

struct DevBits {
	const uint32_t *w;       /* file image as dwords (arena slots are 8-byte aligned, zero padded behind the file) */
	uint64_t bit;            /* next unread bit */
	uint64_t limit;          /* bits that really belong to the file */
	uint64_t win;
	uint32_t have;
	bool over;               /* CAREFUL mode: a read went past `limit` */

	__device__ __forceinline__ void refill()
	{
		const uint64_t i = bit >> 5;
		const uint64_t two = ((uint64_t)w[i + 1] << 32) | w[i];
		const uint32_t sh = (uint32_t)(bit & 31);
		win = two >> sh;
		have = 64 - sh;                 /* >= 33 */
	}
	template <bool CAREFUL>
	__device__ __forceinline__ uint32_t get(uint32_t n)     /* n <= 16 */
	{
		if (CAREFUL && (over || bit + n > limit)) {
			over = true;                    /* stop consuming: the caller discards the block */
			return 0;
		}
		if (have < n)
			refill();
		const uint32_t v = (uint32_t)win & ((1u << n) - 1);
		win >>= n;
		have -= n;
		bit += n;
		return v;
	}
};

__device__ __forceinline__ uint32_t code_reach(uint32_t code)
{
	/* largest |index| a code can produce; linear code c spans [-2^(c-1), 2^(c-1)) */
	if (code >= 3 && code <= 16)
		return 1u << (code - 1);
	switch (code) {
	case 17: case 18: case 19: return 1;
	case 20: case 21: case 22: return 2;
	case 23: case 24: return 3;
	case 26: case 27: return 4;
	case 29: return 5;
	default: return 0;
	}
}

/* one column; returns false on a symbol the reference rejects (corrupt) */
template <bool CF>
__device__ __forceinline__ bool parse_column(DevBits &bs, uint32_t code, uint32_t rows, int16_t *col)
{
	uint32_t r = 0, b;
	if (code == 0) {
		for (; r < rows; r++)
			col[r] = 0;
		return true;
	}
	if (code >= 3 && code <= 16) {
		const int mid = 1 << (code - 1);
		for (; r < rows; r++)
			col[r] = (int16_t)((int)bs.get<CF>(code) - mid);
		return true;
	}
	switch (code) {
	case 17: case 20: case 23: case 26:                     /* "0" = two zeros, "10" = zero, "11.." = value */
		while (r < rows) {
			if (!bs.get<CF>(1)) {
				col[r++] = 0;
				if (r >= rows)
					break;
				col[r++] = 0;
				continue;
			}
			int v = 0;
			if (bs.get<CF>(1)) {
				if (code == 17) {
					v = bs.get<CF>(1) ? 1 : -1;
				} else if (code == 20) {
					b = bs.get<CF>(2);
					v = (b < 2) ? (int)b - 2 : (int)b - 1;          /* -2 -1 +1 +2 */
				} else if (code == 23) {
					if (!bs.get<CF>(1)) {
						v = bs.get<CF>(1) ? 1 : -1;
					} else {
						b = bs.get<CF>(2);
						v = (b < 2) ? (int)b - 3 : (int)b;      /* -3 -2 +2 +3 */
					}
				} else {
					b = bs.get<CF>(3);
					v = (b < 4) ? (int)b - 4 : (int)b - 3;          /* -4..-1 +1..+4 */
				}
			}
			col[r++] = (int16_t)v;
		}
		return true;
	case 18: case 21: case 24: case 27:                     /* "0" = zero, "1.." = value */
		for (; r < rows; r++) {
			int v = 0;
			if (bs.get<CF>(1)) {
				if (code == 18) {
					v = bs.get<CF>(1) ? 1 : -1;
				} else if (code == 21) {
					b = bs.get<CF>(2);
					v = (b < 2) ? (int)b - 2 : (int)b - 1;
				} else if (code == 24) {
					if (!bs.get<CF>(1)) {
						v = bs.get<CF>(1) ? 1 : -1;
					} else {
						b = bs.get<CF>(2);
						v = (b < 2) ? (int)b - 3 : (int)b;
					}
				} else {
					b = bs.get<CF>(3);
					v = (b < 4) ? (int)b - 4 : (int)b - 3;
				}
			}
			col[r] = (int16_t)v;
		}
		return true;
	case 19: case 22: {                                     /* three base-3 / base-5 digits per 5 / 7 bits */
		const uint32_t base = (code == 19) ? 3 : 5, width = (code == 19) ? 5 : 7;
		while (r < rows) {
			b = bs.get<CF>(width);
			if (b >= base * base * base)
				return false;
			for (int k = 0; k < 3 && r < rows; k++, r++) {
				col[r] = (int16_t)((int)(b % base) - (int)(base / 2));
				b /= base;
			}
		}
		return true;
	}
	case 29:                                                /* two base-11 digits per 7 bits */
		while (r < rows) {
			b = bs.get<CF>(7);
			if (b >= 121)
				return false;
			col[r++] = (int16_t)((int)(b % 11) - 5);
			if (r >= rows)
				break;
			col[r++] = (int16_t)((int)(b / 11) - 5);
		}
		return true;
	default:
		return false;                                   /* 1, 2, 25, 28, 30, 31 */
	}
}

__global__ void __launch_bounds__(PARSE_THREADS)
acm_parse_streams(const AcmParseJob *__restrict__ jobs, uint32_t njobs, const uint8_t *__restrict__ files,
		  int16_t *__restrict__ idx_cm, acmhip_blkhdr *__restrict__ hdr, AcmParseResult *__restrict__ res)
{
	const uint32_t j = blockIdx.x * PARSE_THREADS + threadIdx.x;
	if (j >= njobs)
		return;
	const AcmParseJob job = jobs[j];
	const uint32_t rows = job.rows, cols = 1u << job.level;
	const uint64_t bl = (uint64_t)rows * cols;
	const uint64_t safe_bits = (uint64_t)job.file_len * 8;          /* reads beyond this are the host's business */
	const uint64_t col_worst = 5 + (uint64_t)rows * 16;

	DevBits bs;
	bs.w = reinterpret_cast<const uint32_t *>(files + job.file_off);
	bs.bit = (uint64_t)job.data_start * 8;
	bs.limit = safe_bits;
	bs.over = false;
	bs.refill();

	uint32_t done = 0, status = 0;
	for (uint32_t b = 0; b < job.blocks && !status; b++) {
		if (bs.bit + 20 > safe_bits) {
			status = 1;
			break;
		}
		const uint32_t pwr = bs.get<false>(4);
		const uint32_t val = bs.get<false>(16);
		const int lim = 1 << pwr;
		int16_t *blk = idx_cm + job.idx_off + (uint64_t)b * bl;
		for (uint32_t c = 0; c < cols; c++) {
			int16_t *col = blk + (uint64_t)c * rows;
			uint32_t code;
			bool good;
			if (bs.bit + col_worst <= safe_bits) {
				code = bs.get<false>(5);
				good = parse_column<false>(bs, code, rows, col);
			} else {                                        /* near the end of the data: check every read */
				code = bs.get<true>(5);
				good = parse_column<true>(bs, code, rows, col) && !bs.over;
			}
			if (!good) {
				status = 1;
				break;
			}
			const int reach = (int)code_reach(code);
			if ((code >= 3 && code <= 16) ? (reach > lim) : (reach >= lim)) {
				for (uint32_t r = 0; r < rows; r++)             /* hazard H1: host resolves stale-table reads */
					if (col[r] >= lim || col[r] < -lim)
						status = 1;
				if (status)
					break;
			}
		}
		if (status)
			break;
		hdr[job.hdr_off + b] = acmhip_blkhdr{ val, pwr };
		done++;
	}
	res[j] = AcmParseResult{ done, status };
}

/* column-major blocks -> row-major staged form, CC columns of one block per workgroup pass */
template <int CC>
__global__ void __launch_bounds__(256)
acm_parse_transpose(const AcmParseJob *__restrict__ jobs, const AcmParseResult *__restrict__ res,
		    const int16_t *__restrict__ idx_cm, int16_t *__restrict__ idx_rm)
{
	extern __shared__ int16_t tile[];
	const AcmParseJob job = jobs[blockIdx.y];
	const uint32_t rows = job.rows, cols = 1u << job.level;
	const uint32_t cc = cols < (uint32_t)CC ? cols : (uint32_t)CC;
	const uint32_t chunks = cols / cc;
	const uint32_t pitch = rows | 1;                        /* odd pitch: conflict-free column reads */
	const uint64_t bl = (uint64_t)rows * cols;
	const uint32_t nwork = res[blockIdx.y].blocks_done * chunks;
	for (uint32_t wk = blockIdx.x; wk < nwork; wk += gridDim.x) {
		const uint32_t b = wk / chunks, c0 = (wk % chunks) * cc;
		const int16_t *src = idx_cm + job.idx_off + (uint64_t)b * bl + (uint64_t)c0 * rows;
		int16_t *dst = idx_rm + job.idx_off + (uint64_t)b * bl + c0;
		const uint32_t n = cc * rows;
		__syncthreads();
		for (uint32_t e = threadIdx.x; e < n; e += 256) {
			const uint32_t c = e / rows, r = e - c * rows;
			tile[c * pitch + r] = src[e];
		}
		__syncthreads();
		for (uint32_t e = threadIdx.x; e < n; e += 256) {
			const uint32_t r = e / cc, c = e - r * cc;
			dst[(uint64_t)r * cols + c] = tile[c * pitch + r];
		}
	}
}

} // namespace

#define ACMP_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

extern "C" int acmk_parse_supported(uint32_t level, uint32_t rows)
{
	(void)level;
	return rows <= 512;      /* the transpose tile holds 32 columns x rows in LDS */
}

extern "C" int acmk_launch_parse(const AcmParseJob *d_jobs, uint32_t njobs, const uint8_t *d_files,
				 int16_t *d_idx_cm, int16_t *d_idx_rm, acmhip_blkhdr *d_hdr,
				 AcmParseResult *d_res, uint32_t max_blocks, uint32_t max_cols, void *stream)
{
	if (njobs == 0)
		return 0;
	hipStream_t st = (hipStream_t)stream;
	hipLaunchKernelGGL(acm_parse_streams, dim3((njobs + PARSE_THREADS - 1) / PARSE_THREADS), dim3(PARSE_THREADS), 0, st,
			   d_jobs, njobs, d_files, d_idx_cm, d_hdr, d_res);
	ACMP_CHECK();
	constexpr int CC = 32;
	const size_t lds = (size_t)CC * 513 * sizeof(int16_t);
	uint64_t gx = (uint64_t)max_blocks * ((max_cols + CC - 1) / CC);
	if (gx < 1)
		gx = 1;
	if (gx > 4096)
		gx = 4096;
	for (uint32_t at = 0; at < njobs; at += 65535) {
		const uint32_t n = njobs - at < 65535 ? njobs - at : 65535;
		hipLaunchKernelGGL(acm_parse_transpose<CC>, dim3((unsigned)gx, n), dim3(256), lds, st,
				   d_jobs + at, d_res + at, d_idx_cm, d_idx_rm);
		ACMP_CHECK();
	}
	return 0;
}
