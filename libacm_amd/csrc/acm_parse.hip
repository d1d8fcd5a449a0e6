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
 *                      Up to 32 K streams: acm_parse_scan_wave, one stream per WAVEFRONT - the walk is wave-uniform
 *                      on the scalar unit, the lanes are its register file (bitstream window, code tables, column
 *                      offsets) and resolve a k-column together; 2.2x round 1's scalar walk on 1024 long streams.
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

#include <cstdlib>
#include <type_traits>

#include "acm_device.h"

#pragma clang diagnostic ignored "-Winline-asm"         /* m0 on a clobber list: see acm_parse_scan_wave */

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

/* ---- byte-plane staging: the width class of a block from its pwr alone ----
 * An index of a block lies in [-2^pwr, 2^pwr) (decode.c:592-600; anything else is hazard H1 and goes back to the host), so a class of
 * include/acm_hip.h that holds the block is known before a single column is decoded - which is what lets the walk place every block:
 * 8 bits up to pwr 7, two signed bytes up to pwr 14 (they end at 32 639), the whole-range class for pwr 15.  The host stager
 * (acm_pack.cpp) looks at the indices themselves and also has the 12-bit class; here that class was measured and left out
 * (profiles/r6_level9_notes.txt section 17): its nibble exchange and third store cost the column kernel 21 us per range launch, the
 * synthesis of a range gains nothing measurable from the bytes it saves. */
__device__ __forceinline__ uint32_t bp_class(uint32_t pwr)
{
	return pwr < 8u ? ACMHIP_BP_BYTE : pwr == 15u ? ACMHIP_BP_WORDU : ACMHIP_BP_WORD;
}
/* bytes per index, times two (WORDU 4, NIB12 3 - not written here -, BYTE 2, WORD 4) */
__device__ __forceinline__ uint32_t bp_half_bytes(uint32_t cls)
{
	return (0x4234u >> (4u * cls)) & 15u;
}
/* the class that holds both (a row pair that straddles two blocks takes the wider of theirs): BYTE < NIB12 < WORD < WORDU */
__device__ __forceinline__ uint32_t bp_wider(uint32_t a, uint32_t b)
{
	const uint32_t ra = (0x2013u >> (4u * a)) & 15u, rb = (0x2013u >> (4u * b)) & 15u;
	return ra >= rb ? a : b;
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

/* striped upload: stripe s of S of a file's arena slot (the file, padded to 16 bytes, + 16 zero bytes) starts here */
__host__ __device__ __forceinline__ uint32_t acm_stripe_bound(uint32_t file_len, uint32_t s, uint32_t S)
{
	const uint32_t slot = ((file_len + 15u) & ~15u) + 16u;
	return s >= S ? slot : (uint32_t)(((uint64_t)slot * s / S) & ~15ull);
}

/* first block of block range r of R of a stream of `blocks` blocks (r == R: its end).  Ranges are cut at multiples of `unit` blocks (0 / 1:
 * anywhere; acm_batch.cpp asks for whole tiles of the lean kernel where a stream has the byte-plane form, which also keeps every row pair
 * of an odd block height inside one range); a short stream may have empty ranges */
__host__ __device__ __forceinline__ uint32_t acm_range_bound(uint32_t blocks, uint32_t r, uint32_t R, uint32_t unit)
{
	if (r >= R)
		return blocks;
	const uint32_t b = (uint32_t)((uint64_t)blocks * r / R);
	return unit > 1u ? b / unit * unit : b;
}

/* payload bits of a column by filler code for `rows` rows; K_WALK = step through it, BAD_CODE = stop */
__device__ __forceinline__ uint32_t column_bits(uint32_t code, uint32_t rows)
{
	const uint32_t cls = code_class(code);
	return cls == CLS_ZERO ? 0u : cls == CLS_LINEAR ? rows * code : cls == CLS_K ? K_WALK : cls == CLS_BAD ? BAD_CODE :
	       code == 19 ? (rows + 2) / 3 * 5 : code == 22 ? (rows + 2) / 3 * 7 : (rows + 1) / 2 * 7;
}

/*
 * One stream.  LONE (tuning builds) = this wavefront has no other stream: everything about the walk is then wave-uniform, the
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

#ifdef ACM_TUNING
/* round 1's kernel for few streams, kept for A/B runs (ACM_PARSE_SCAN=1): one stream per wavefront, lane 0 only, on the scalar unit */
__global__ void __launch_bounds__(SCAN_THREADS)
acm_parse_scan_lone(const AcmParseJob *__restrict__ jobs, uint32_t njobs, const uint8_t *__restrict__ files,
		    uint32_t *__restrict__ colpos, acmhip_blkhdr *__restrict__ hdr, AcmParseResult *__restrict__ res)
{
	if (threadIdx.x != 0 || blockIdx.x >= njobs)
		return;
	const AcmParseJob job = jobs[blockIdx.x];
	scan_stream<true>(job, files, colpos, hdr, res + blockIdx.x, nullptr);
}
#endif

/*
 * One stream per wavefront, the wavefront as the scalar walk's register file.
 *
 * Round 1's scalar walk (scan_stream<true>, still built with ACM_TUNING for A/B runs) spends ~600 cycles on a fixed-length
 * column and ~2700 on a 16-row k-column (measured, one filler per stream: profiles/r2_parse_probe.txt): a dependent scalar
 * load per column, the code -> length decision as a tree of branches, and ~12 dependent scalar instructions per k-symbol.  Here
 *   - the bitstream window is 65 consecutive dwords held one (dword, next dword) pair per lane, loaded coalesced once
 *     per ~1900 bits; any 64 bits of it are two v_readlane away;
 *   - code -> column length and code -> k-prefix table are v_readlane lookups in per-lane tables;
 *   - column offsets collect in a VGPR (v_writelane) and go out 64 at a time, coalesced;
 *   - a k-column is resolved by the whole wavefront: lane p assumes "a symbol starts at bit p of the next 64" and looks up
 *     where it would end and how many rows it stands for; pointer doubling (one ds_bpermute per round) turns that into
 *     "where do 2, 4, 8, 16 symbols from p end", and a binary descent on the scalar side finds the end of the column from
 *     the real start, p = 0.
 */
struct WaveWindow {
	const uint32_t *w;
	uint32_t maxdw;         /* last dword that may be read (inside the zero padding behind the file) */
	uint32_t base;          /* dword index lane 0 holds */
	uint32_t lo, hi;        /* this lane: w[base + lane], w[base + lane + 1] */
	uint32_t lane;

	__device__ __forceinline__ void load(uint32_t dw)
	{
		base = dw;
		const uint32_t i0 = min(dw + lane, maxdw), i1 = min(dw + lane + 1, maxdw);
		lo = __builtin_nontemporal_load(w + i0);
		hi = __builtin_nontemporal_load(w + i1);
	}
	/* make dwords bit/32 .. bit/32 + 3 readable; returns the lane that holds the first of them */
	__device__ __forceinline__ uint32_t at(uint32_t bit)
	{
		uint32_t r = (bit >> 5) - base;
		if (r > 60) {
			load(bit >> 5);
			r = 0;
		}
		return r;
	}
	__device__ __forceinline__ uint64_t peek64(uint32_t bit)     /* 64 bits from dword bit/32, shifted down to `bit`: >= 33 valid */
	{
		const uint32_t r = at(bit);
		const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)lo, (int)r);
		const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)hi, (int)r);
		return (((uint64_t)b << 32) | a) >> (bit & 31);
	}
};

/* returns the bit behind the column's last symbol, or ~0 when the data ends inside the column; rows_left >= 1 */
template <int LEVELS>
__device__ __forceinline__ uint32_t walk_k_column(WaveWindow &ww, uint32_t pos, const uint32_t tab, uint32_t rows_left, const uint32_t safe)
{
	while (rows_left > 0 && pos < safe) {
		const uint32_t r = ww.at(pos), sh = pos & 31;
		const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)ww.lo, (int)r);
		const uint32_t w1 = (uint32_t)__builtin_amdgcn_readlane((int)ww.hi, (int)r);
		const uint32_t w2 = (uint32_t)__builtin_amdgcn_readlane((int)ww.lo, (int)(r + 2));
		const uint32_t w3 = (uint32_t)__builtin_amdgcn_readlane((int)ww.hi, (int)(r + 2));
		const uint32_t o = sh + ww.lane, d = o >> 5;
		const uint32_t a = d == 0 ? w0 : d == 1 ? w1 : w2, b = d == 0 ? w1 : d == 1 ? w2 : w3;
		const uint32_t x = (uint32_t)((((uint64_t)b << 32) | a) >> (o & 31));
		/* jump[k] = (rows << 8 | end) of 2^k symbols from this lane's bit.  Four symbols are at most 20 bits: the first
		 * three levels come out of this lane's own 32 bits; from there on pointer doubling, where a chain that has left
		 * the window (end >= 64) stays as it is - a short jump, still a whole number of symbols. */
		uint32_t jump[LEVELS];
		{
			uint32_t used = 0, got = 0;
#pragma unroll
			for (int sy = 0; sy < 4; sy++) {
				const uint32_t e = (tab >> (((x >> used) & 7u) * 4)) & 15u;
				used += e & 7u;
				got += 1 + (e >> 3);
				if (sy == 0 || sy == 1 || sy == 3)
					jump[sy == 3 ? 2 : sy] = (ww.lane + used) | got << 8;
			}
		}
#pragma unroll
		for (int k = 2; k + 1 < LEVELS; k++) {
			const uint32_t end = jump[k] & 255u;
			const uint32_t far = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(end << 2), (int)jump[k]);
			jump[k + 1] = end < 64 ? far + (jump[k] & ~255u) : jump[k];
		}
		/* descent from the real start: the largest jumps that stay short of the column's last row ... */
		uint32_t cur = 0, acc = 0;
#pragma unroll
		for (int k = LEVELS - 1; k >= 0; k--) {
			const uint32_t j = (uint32_t)__builtin_amdgcn_readlane((int)jump[k], (int)(cur & 63u));
			const bool take = cur < 64 && acc + (j >> 8) < rows_left;
			cur = take ? (j & 255u) : cur;
			acc = take ? acc + (j >> 8) : acc;
		}
		/* ... then one symbol more: it holds the last row, unless 2^LEVELS - 1 symbols were not enough (next round) */
		if (cur < 64) {
			const uint32_t j = (uint32_t)__builtin_amdgcn_readlane((int)jump[0], (int)cur);
			cur = j & 255u;
			acc += j >> 8;
		}
		pos += cur;
		rows_left = acc >= rows_left ? 0u : rows_left - acc;
	}
	return rows_left ? 0xFFFFFFFFu : pos;
}

constexpr int WAVE_SCAN_WAVES = 4;      /* streams per workgroup: one per SIMD of the CU */

__global__ void __launch_bounds__(64 * WAVE_SCAN_WAVES)
acm_parse_scan_wave(const AcmParseJob *__restrict__ jobs, uint32_t njobs, const uint8_t *__restrict__ files,
		    uint32_t *__restrict__ colpos, acmhip_blkhdr *__restrict__ hdr, AcmParseResult *__restrict__ res,
		    const uint32_t range, const uint32_t nranges, const uint32_t stripes_up, uint32_t *__restrict__ blkoff,
		    uint32_t *__restrict__ mf_pairs)
{
	const uint32_t jobno = blockIdx.x * WAVE_SCAN_WAVES + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	if (jobno >= njobs)
		return;
	const uint32_t lane = threadIdx.x & 63u;
	const AcmParseJob job = jobs[jobno];
	const uint32_t rows = job.rows, cols = 1u << job.level;
	/* striped upload (acm_batch.cpp): only the first stripes_up of nranges stripes of every file are on the device yet; a
	 * stream whose range reaches beyond them stops as if its data had run out, and the host reader takes it */
	const uint32_t safe = 8u * (stripes_up && stripes_up < nranges ? min(job.file_len, acm_stripe_bound(job.file_len, stripes_up, nranges))
								      : job.file_len);
	/* per-lane tables, looked up with v_readlane: lane = filler code */
	const uint32_t code_len = column_bits(lane & 31u, rows);
	const uint32_t code_tab = k_table_for(lane & 31u);
	WaveWindow ww;
	ww.w = reinterpret_cast<const uint32_t *>(files + job.file_off);
	ww.maxdw = (job.file_len + 15u) / 4u;
	ww.lane = lane;
	/* block range `range` of `nranges` (1 of 1: the whole stream): a later range resumes where the one before stopped */
	const uint32_t b_lo = acm_range_bound(job.blocks, range, nranges, job.range_unit), b_hi = acm_range_bound(job.blocks, range + 1u, nranges, job.range_unit);
	uint32_t bit = job.data_start * 8u;
	/* byte-plane staging (job.mf_rows != 0): where block b starts in the stream's region - behind the pair of zeros, every block at
	 * 1, 1.5 or 2 bytes per index by its pwr (bp_class) - is a running sum only this walk knows */
	uint32_t mf_at = (2u * cols) >> 6;
	/* a block height that is odd: every other block starts with the second row of a pair that began in the block before it.  Such a pair
	 * takes the wider class of the two blocks, so its size is known only here, at the second one; str_at is where it lies, str_cls the
	 * class the first block asked for.  (Block ranges never cut such a pair: job.range_unit makes them end on whole tiles.) */
	uint32_t str_at = 0, str_cls = ACMHIP_BP_BYTE;
	/* Byte-plane staging and block ranges: the synthesis of a range is queued behind its walk without the host looking at the result, on
	 * a plan cut from what the headers promised.  A stream whose walk stops early leaves the pair-table entries of the blocks it did not
	 * reach unwritten - and an entry is a PLACE: the chunk kernel would load from wherever the garbage points.  Those entries are
	 * therefore all made to name the place right behind the last block that was staged (inside the stream's region, at or behind
	 * every entry in front of them: the kernel's offsets stay small and positive).  What is decoded from there is garbage the host
	 * reader's redo replaces; it is only not allowed to fault. */
	auto park_entries = [&](const uint32_t first_block, const uint32_t at64) {
		if (!mf_pairs || !job.mf_rows)
			return;
		const uint32_t safe = (uint32_t)(((job.mf_off >> 6) + at64) << 2) | ACMHIP_BP_BYTE;
		const uint32_t p_end = min(b_hi * rows, job.mf_rows) / 2u;
		for (uint32_t p = first_block * rows / 2u + lane; p < p_end; p += 64u)
			mf_pairs[job.mf_pair_off + 1u + p] = safe;
		if (lane == 0 && range == 0)
			mf_pairs[job.mf_pair_off] = (uint32_t)((job.mf_off >> 6) << 2) | ACMHIP_BP_BYTE;     /* (the column kernel writes it again with block 0) */
	};
	if (range > 0) {
		const AcmParseResult prev = res[jobno];
		if (prev.status != 0 || prev.blocks_done != b_lo) {
			park_entries(b_lo, prev.mf_at);
			return;                                 /* the stream failed earlier: its record stays as it is */
		}
		bit = prev.end_bit;
		mf_at = prev.mf_at;
	}
	const uint32_t mf_at_lo = mf_at;                /* where this range's first block would be staged */
	ww.load(bit >> 5);
	uint32_t done = b_lo, status = 1;
	uint32_t *cp = colpos + job.col_off + (uint64_t)b_lo * cols;
	for (uint32_t b = b_lo; b < b_hi; b++) {
		if (bit + 20 > safe)
			goto out;
		const uint32_t h20 = (uint32_t)ww.peek64(bit) & 0xFFFFFu;
		bit += 20;
		for (uint32_t c0 = 0; c0 < cols; c0 += 64) {
			const uint32_t n = __builtin_amdgcn_readfirstlane(min(64u, cols - c0));
			uint32_t cpv = 0, k = 0;
			for (;;) {
				/* the fast loop: fixed-length columns, one exit; why it ended is sorted out behind it.  A column
				 * that starts in the last 5 bits of the file ends behind it: one bounds check, at the end. */
				uint32_t code, len;
				bool fixed;
				do {
					/* cpv[lane k] = bit.  Two scalar operands do not fit one VALU instruction: the lane select goes
					 * through m0, which compiler-generated gfx950 code does not use here (tests/test_isa_invariants.py) */
					asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(cpv) : "s"(bit), "s"(k) : "m0");
					code = (uint32_t)ww.peek64(bit) & 31u;
					len = (uint32_t)__builtin_amdgcn_readlane((int)code_len, (int)code);
					fixed = len < K_WALK;
					bit += 5 + (fixed ? len : 0u);
					k++;
				} while (fixed & (bit <= safe) & (k < n));
				if (!fixed) {
					if (len != K_WALK)
						goto out;
					const uint32_t tab = (uint32_t)__builtin_amdgcn_readlane((int)code_tab, (int)code);
					bit = rows <= 16 ? walk_k_column<4>(ww, bit, tab, rows, safe) : walk_k_column<5>(ww, bit, tab, rows, safe);
				}
				if (bit > safe)                                 /* the column must end inside the file */
					goto out;
				if (k >= n)
					break;
			}
			if (lane < n)
				cp[c0 + lane] = cpv;
		}
		{
			/* blkoff[b]: where the row pair that holds the block's first row lies */
			const uint32_t cls = bp_class(h20 & 15u);
			uint32_t inner = rows, first_at = mf_at;
			if ((b * rows) & 1u) {
				first_at = str_at;
				mf_at = str_at + ((cols * bp_half_bytes(bp_wider(str_cls, cls))) >> 6);
				inner = rows - 1u;
			}
			if (lane == 0) {
				hdr[job.hdr_off + b] = acmhip_blkhdr{ h20 >> 4, h20 & 15u };
				if (blkoff)
					blkoff[job.hdr_off + b] = first_at;
			}
			mf_at += (inner >> 1) * ((cols * bp_half_bytes(cls)) >> 6);
			if (inner & 1u) {
				str_at = mf_at;
				str_cls = cls;
			}
		}
		cp += cols;
		done++;
	}
	status = 0;
out:
	/* (a walk that stops inside its range: the column kernel skips the stream altogether - status != 0 -, the blocks of this range the walk
	 * did get through included; none of the range's entries is written by anybody else) */
	if (status != 0)
		park_entries(b_lo, mf_at_lo);
	else if (range == 0)
		park_entries(done, mf_at);
	if (lane == 0)
		res[jobno] = AcmParseResult{ done, status, bit, mf_at };
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
		  const acmhip_blkhdr *__restrict__ hdr, int16_t *__restrict__ idx, uint32_t *__restrict__ flags,
		  const uint32_t range, const uint32_t nranges, uint8_t *__restrict__ mf, uint32_t *__restrict__ mf_pairs,
		  const uint32_t *__restrict__ blkoff)
{
	__shared__ uint32_t lut[LUT_CLASSES * 128];
	const AcmParseJob job = jobs[blockIdx.y];
	const AcmParseResult rs = res[blockIdx.y];
	if (rs.status != 0)
		return;                                         /* the host redoes the whole stream */
	const uint32_t rows = job.rows, level = job.level, cols = 1u << level;
	/* the columns of the blocks the walk has just covered (block range `range` of `nranges`) */
	const uint32_t b_lo = acm_range_bound(job.blocks, range, nranges, job.range_unit);
	const uint32_t b_hi = min(rs.blocks_done, acm_range_bound(job.blocks, range + 1u, nranges, job.range_unit));
	const uint32_t ncol = b_hi << level;                    /* blocks * cols < 2^32 (acmk_parse_supported) */
	/* the grid is sized for the longest stream of the batch: in a batch of ragged streams most workgroups have nothing to do,
	 * and should find that out before they build the table (a corpus of 4000 files: 1.3 M workgroups, 5.3 ms per launch) */
	if ((uint64_t)(b_lo << level) + (uint64_t)blockIdx.x * COL_THREADS >= ncol)
		return;
	{
		constexpr uint32_t codes[LUT_CLASSES] = { 17, 18, 19, 20, 21, 22, 23, 24, 26, 27, 29 };
		for (uint32_t e = threadIdx.x; e < LUT_CLASSES * 128; e += COL_THREADS)
			lut[e] = lut_entry(codes[e >> 7], e & 127);
		__syncthreads();
	}
	const uint64_t bl = (uint64_t)rows << level;
	const uint32_t *base = reinterpret_cast<const uint32_t *>(files + job.file_off);
	uint32_t bad = 0;
	/* byte-plane staging (the chunk kernel's form, include/acm_hip.h: 64 columns of a residue class side by side, 8, 12 or 16 bits per
	 * index by the block's pwr - bp_class): rows [0, mf_rows) go there, only the rows from mf_rows - 2 on to the int16 arena.
	 * A wavefront takes 64 columns of ONE class - thread t of it column class + SIGMA t - so that its stores of a row are 64
	 * consecutive bytes */
	const uint32_t mf_rows = mf && mf_pairs && blkoff ? job.mf_rows : 0u;
	const uint32_t sigma = cols >> 6;
	uint8_t *const region = mf + job.mf_off;

	for (uint32_t g = (b_lo << level) + blockIdx.x * COL_THREADS + threadIdx.x; g < ncol; g += gridDim.x * COL_THREADS) {
		const uint32_t b = g >> level, cg = g & (cols - 1);
		const uint32_t c = mf_rows ? (cg >> 6) + sigma * (cg & 63u) : cg;
		/* the three loads a column starts with travel together: one round trip, then the bit window's */
		const uint32_t pwr = hdr[job.hdr_off + b].pwr;
		const uint32_t cpos = colpos[job.col_off + ((uint64_t)b << level) + c];  /* (c: the column this thread decodes - not its place in the grid, see above) */
		const uint32_t boff = mf_rows ? blkoff[job.hdr_off + b] : 0u;
		const int lim = 1 << pwr;
		int16_t *out = idx + job.idx_off + (uint64_t)b * bl + c;
		const uint32_t row0 = b * rows;                 /* stream row of the block's first row (blocks * rows < 2^32: acmk_parse_supported) */
		/* The block's rows in the byte-plane form.  Local pair j = the j-th row pair that has a row in this block; all of them have the
		 * block's own class (bp_class) except a first one that began in the block before (row0 odd) and a last one that ends in the block
		 * behind (the block's end odd): those take the wider class of the two blocks.  Places: blkoff[b] (from the walk) is local pair 0,
		 * the others follow back to back.
		 * A stream with the form has >= 256 columns, so the 64 columns of a wavefront are ONE block's and one residue class's: block,
		 * classes and row places are said to be wave-uniform (readfirstlane) - scalar registers, scalar branches in the row loop - and
		 * a thread adds only its lane, q.  rp_*: byte offsets of the wavefront's class in the block's rows, from local pair 0 on (a
		 * block is < 2^26 bytes): rows of the block's own class follow each other rp_half apart; the first pair and a last pair of
		 * another class have their own geometry */
		uint32_t odd0 = 0, mf_cnt = 0;                  /* mf_cnt: the block's first rows that are staged in the form */
		uint32_t cls_b = ACMHIP_BP_BYTE, cls_head = ACMHIP_BP_BYTE, cls_tail = ACMHIP_BP_BYTE;    /* the block's class, local pair 0's, the last local pair's */
		uint32_t rp_half = 0, rp_half_head = 0, rp_head = 0, rp_body = 0, rp_tail_t = 0xFFFFFFFFu, rp_tail = 0;
		const uint32_t q = threadIdx.x & 63u;
		uint8_t *mo = nullptr;                          /* local pair 0, + q */
		const uint32_t bu = (uint32_t)__builtin_amdgcn_readfirstlane((int)b), cgu = (uint32_t)__builtin_amdgcn_readfirstlane((int)cg);
		const uint32_t row0u = bu * rows;               /* (the condition below is a scalar one: what it guards stays in scalar registers) */
		if (mf_rows && row0u < mf_rows) {
			const uint32_t cidx = cgu >> 6;
			odd0 = row0u & 1u;
			mf_cnt = min(rows, mf_rows - row0u);
			const uint32_t npair = (odd0 + rows + 1u) >> 1;
			/* (pwr and blkoff come by the vector loads that travel with the column's bit offset - one wait for the three - and are
			 * made scalar afterwards: a scalar load here would be a second round trip in front of every column) */
			const uint32_t at64 = (uint32_t)__builtin_amdgcn_readfirstlane((int)boff);
			cls_b = cls_head = cls_tail = bp_class((uint32_t)__builtin_amdgcn_readfirstlane((int)pwr));
			if (odd0)
				cls_head = bp_wider(cls_b, bp_class(hdr[job.hdr_off + bu - 1u].pwr));
			if (((row0u + rows) & 1u) && row0u + rows < mf_rows) {      /* (an even mf_rows: the partner row is staged too, its block was walked) */
				cls_tail = bp_wider(cls_b, bp_class(hdr[job.hdr_off + bu + 1u].pwr));
				if (npair == 1u)
					cls_head = cls_tail;
			}
			if (npair == 1u)
				cls_tail = cls_head;
			const uint64_t at = (uint64_t)at64 << 6;
			mo = region + at + q;
			const uint32_t hb_b = bp_half_bytes(cls_b), hb_h = bp_half_bytes(cls_head), hb_t = bp_half_bytes(cls_tail);
			rp_half = (cols * hb_b) >> 1;
			rp_half_head = (cols * hb_h) >> 1;
			rp_head = cidx * 32u * hb_h;
			rp_body = cols * hb_h + cidx * 32u * hb_b - 2u * rp_half;       /* + t * rp_half, t >= 2 (mod 2^32) */
			if (npair >= 2u && cls_tail != cls_b) {
				rp_tail_t = 2u * (npair - 1u);
				rp_tail = cols * hb_h + (npair - 2u) * cols * hb_b + cidx * 32u * hb_t;
			}
			/* the pair-table entries (entry k of a stream = where row pair k - 1 starts; entry 0: the pair of zeros in front) of the pairs
			 * that BEGIN in this block, written by the block's first threads */
			for (uint32_t j = odd0 + cg; j < npair; j += cols) {
				const uint32_t first_row = row0u + 2u * j - odd0;
				if (first_row < mf_rows) {
					const uint32_t cj = j == 0u ? cls_head : j + 1u == npair ? cls_tail : cls_b;
					const uint64_t pj = j == 0u ? 0u : (uint64_t)cols * hb_h + (uint64_t)(j - 1u) * cols * hb_b;
					mf_pairs[job.mf_pair_off + 1u + first_row / 2u] = (uint32_t)(((job.mf_off + at + pj) >> 6) << 2) | cj;
				}
			}
			if (b == 0 && cg == 0)
				mf_pairs[job.mf_pair_off] = (uint32_t)((job.mf_off >> 6) << 2) | ACMHIP_BP_BYTE;
			if (b == 0 && cg < cols / 8u)
				reinterpret_cast<uint4 *>(region)[cg] = make_uint4(0u, 0u, 0u, 0u);       /* the pair of zeros: 2 * cols bytes */
		}
		DevBits bs;
		bs.seek(base, cpos);
		const uint32_t code = bs.get(5);
		const uint32_t cls = code_class(code);
		const bool table = cls >= CLS_TERN;             /* CLS_BAD cannot occur: the scan flagged the stream */
		const uint32_t lbase = table ? (uint32_t)lut_class(code) * 128u : 0u;
		const uint32_t width = table ? 0u : code;       /* zero filler = linear with no bits */
		const uint32_t mask = (1u << width) - 1u;
		const int mid = width ? 1 << (width - 1) : 0;
		uint32_t pend = 0, npend = 0;
		/* plain: every row pair of the block in the form has the block's own class (no first pair that began in the block before, no last
		 * pair of another class: every block of an even height) - a row is then placed with one multiply-add and its class is the block's.
		 * Two copies of the row loop: with the per-row selects of the general one in it, the plain blocks paid a fifth more
		 * (profiles/r6_level9_notes.txt section 17).  (Also plain: every block of a stream without the form, mf_cnt = 0) */
#define ACM_COLUMN_ROWS(PLAIN)                                                                                                          \
		for (uint32_t r = 0; r < rows; r++) {                                                                                   \
			int v;                                                                                                          \
			if (npend) {                                                                                                    \
				v = (int)(pend & 15u) - 8;                                                                              \
				pend >>= 4;                                                                                             \
				npend--;                                                                                                \
			} else {                                                                                                        \
				bs.need(16);                                                                                            \
				const uint32_t raw = (uint32_t)bs.win;                                                                  \
				if (table) {                                                                                            \
					const uint32_t e = lut[lbase + (raw & 127u)];                                                   \
					v = (int)((e >> 5) & 15u) - 8;                                                                  \
					pend = e >> 9;                                                                                  \
					npend = ((e >> 3) & 3u) - 1;                                                                    \
					bad |= (e >> 17) & 1u;                                                                          \
					bs.drop(e & 7u);                                                                                \
				} else {                                                                                                \
					v = (int)(raw & mask) - mid;                                                                    \
					bs.drop(width);                                                                                 \
				}                                                                                                       \
			}                                                                                                               \
			bad |= (v >= lim) | (v < -lim);                 /* hazard H1: the host resolves stale-table reads */             \
			const uint32_t row = row0 + r;                                                                                  \
			if (r < mf_cnt) {                                                                                               \
				/* the row's place (rp_*, above): 64 columns of a residue class side by side - this wavefront's are     \
				 * class cg / 64, thread t of it column t -, 64 or 128 bytes per class and row.  Everything here but    \
				 * the thread's lane is wave-uniform */                                                                  \
				const uint32_t t = r + odd0;            /* row t & 1 of local pair t / 2 */                              \
				uint32_t cj = cls_b, off = rp_body + t * rp_half;                                                       \
				if (!(PLAIN)) {                                                                                         \
					const bool in_head = t < 2u, in_tail = t >= rp_tail_t;                                          \
					cj = in_head ? cls_head : in_tail ? cls_tail : cls_b;                                           \
					off = in_head ? rp_head + (t & 1u) * rp_half_head : in_tail ? rp_tail : off;                    \
				}                                                                                                       \
				const bool whole = cj == ACMHIP_BP_WORDU;                                                               \
				uint8_t *const chunk = mo + off;                                                                        \
				const int lo = (int)(int8_t)(uint8_t)v;         /* two signed bytes: idx = 256 hi + lo */                \
				/* (the whole-range class: the low byte unsigned, stored minus 128 - the kernel adds 128 val x the      \
				 * matrices' row sums back) */                                                                           \
				chunk[0] = (uint8_t)(lo ^ (whole ? 0x80 : 0));                                                          \
				if (cj != ACMHIP_BP_BYTE)                                                                               \
					chunk[64] = (uint8_t)((v - (whole ? (v & 255) : lo)) >> 8);                                     \
			}                                                                                                               \
			if (row + 2u >= mf_rows)                                                                                        \
				out[(uint64_t)r << level] = (int16_t)v;                                                                 \
		}
		if (odd0 == 0u && rp_tail_t == 0xFFFFFFFFu && cls_head == cls_b) {
			ACM_COLUMN_ROWS(true)
		} else {
			ACM_COLUMN_ROWS(false)
		}
#undef ACM_COLUMN_ROWS
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
/* striped upload: stripe s of every file, uploaded back to back into the staging arena, goes to its place in the file arena */
namespace {
__global__ void __launch_bounds__(256)
acm_scatter_stripe(const AcmParseJob *__restrict__ jobs, const uint32_t njobs, const uint64_t *__restrict__ stripe_at,
		   const uint8_t *__restrict__ stage, uint8_t *__restrict__ files, const uint32_t s, const uint32_t S)
{
	const uint32_t k = blockIdx.x;
	if (k >= njobs)
		return;
	const AcmParseJob job = jobs[k];
	const uint32_t lo = acm_stripe_bound(job.file_len, s, S), hi = acm_stripe_bound(job.file_len, s + 1, S);
	const uint4 *src = reinterpret_cast<const uint4 *>(stage + stripe_at[(uint64_t)s * njobs + k]);
	uint4 *dst = reinterpret_cast<uint4 *>(files + job.file_off + lo);
	for (uint32_t v = threadIdx.x; v < (hi - lo) / 16u; v += 256u)
		dst[v] = src[v];
}
}

extern "C" uint32_t acmk_stripe_bound(uint32_t file_len, uint32_t s, uint32_t S)
{
	return acm_stripe_bound(file_len, s, S);
}

extern "C" uint32_t acmk_range_bound(uint32_t blocks, uint32_t r, uint32_t R, uint32_t unit)
{
	return acm_range_bound(blocks, r, R, unit);
}

extern "C" int acmk_launch_scatter_stripe(const AcmParseJob *d_jobs, uint32_t njobs, const uint64_t *d_stripe_at, const uint8_t *d_stage,
					  uint8_t *d_files, uint32_t s, uint32_t S, void *stream)
{
	if (njobs == 0)
		return 0;
	hipLaunchKernelGGL(acm_scatter_stripe, dim3(njobs), dim3(256), 0, (hipStream_t)stream, d_jobs, njobs, d_stripe_at, d_stage, d_files, s, S);
	ACMP_CHECK();
	return 0;
}

extern "C" int acmk_launch_parse_range(const AcmParseJob *d_jobs, uint32_t njobs, const uint8_t *d_files,
				       uint32_t *d_colpos, int16_t *d_idx, acmhip_blkhdr *d_hdr,
				       AcmParseResult *d_res, uint32_t *d_flags, uint64_t max_columns, uint32_t range, uint32_t nranges,
				       uint32_t stripes_up, void *stream)
{
	return acmk_launch_parse_range_mf(d_jobs, njobs, d_files, d_colpos, d_idx, d_hdr, d_res, d_flags, max_columns, range, nranges, stripes_up,
					  nullptr, nullptr, nullptr, stream);
}

extern "C" int acmk_launch_parse_range_mf(const AcmParseJob *d_jobs, uint32_t njobs, const uint8_t *d_files,
					  uint32_t *d_colpos, int16_t *d_idx, acmhip_blkhdr *d_hdr,
					  AcmParseResult *d_res, uint32_t *d_flags, uint64_t max_columns, uint32_t range, uint32_t nranges,
					  uint32_t stripes_up, uint8_t *d_mf, uint32_t *d_pairs, uint32_t *d_blkoff, void *stream)
{
	if (njobs == 0)
		return 0;
	if (nranges == 0 || range >= nranges || (nranges > 1 && njobs > ACM_PARSE_RANGE_MAX_STREAMS))
		return (int)hipErrorInvalidValue;
	const bool mf = d_mf && d_pairs && d_blkoff;
	if (mf && njobs > ACM_PARSE_RANGE_MAX_STREAMS)
		return (int)hipErrorInvalidValue;               /* the block offsets come from the wave-per-stream walk */
	hipStream_t st = (hipStream_t)stream;
	const uint32_t full = (njobs + SCAN_THREADS - 1) / SCAN_THREADS;
	const uint32_t scan_waves = njobs < 8192u ? njobs : full < 8192u ? 8192u : full;
	const uint32_t scan_lanes = (njobs + scan_waves - 1) / scan_waves;
	/* up to ~32 K streams the wave-per-stream walk wins (profiles/r2_parse_probe.txt: 3.0 against 3.4 ms at 32768 streams of
	 * 8 blocks, 3.9 against 8.5 ms at 4096 of 64); beyond, one stream per lane keeps more streams in flight than waves fit */
	uint32_t wave_max = 32768;
	if (mf)
		wave_max = ACM_PARSE_RANGE_MAX_STREAMS;
#ifdef ACM_TUNING
	static const int scan_mode = ACM_TUNING_ENV("ACM_PARSE_SCAN") ? atoi(ACM_TUNING_ENV("ACM_PARSE_SCAN")) : 2;     /* 0 lanes, 1 scalar (r1), 2 wave */
	if (ACM_TUNING_ENV("ACM_PARSE_WAVE_MAX"))
		wave_max = (uint32_t)atoi(ACM_TUNING_ENV("ACM_PARSE_WAVE_MAX"));
	if (scan_mode == 0)
		wave_max = 0;
	if (nranges > 1)                                        /* block ranges exist in the wave-per-stream walk only */
		wave_max = ACM_PARSE_RANGE_MAX_STREAMS;
	else if (njobs <= wave_max && scan_mode == 1)
		hipLaunchKernelGGL(acm_parse_scan_lone, dim3(njobs), dim3(SCAN_THREADS), 0, st,
				   d_jobs, njobs, d_files, d_colpos, d_hdr, d_res);
	if (nranges == 1 && njobs <= wave_max && scan_mode == 1)
		;
	else
#endif
	if (njobs <= wave_max)
		hipLaunchKernelGGL(acm_parse_scan_wave, dim3((njobs + WAVE_SCAN_WAVES - 1) / WAVE_SCAN_WAVES), dim3(64 * WAVE_SCAN_WAVES), 0, st,
				   d_jobs, njobs, d_files, d_colpos, d_hdr, d_res, range, nranges, stripes_up, mf ? d_blkoff : nullptr, mf ? d_pairs : nullptr);
	else
		hipLaunchKernelGGL(acm_parse_scan, dim3(scan_waves), dim3(SCAN_THREADS), scan_lanes * 33 * sizeof(uint32_t), st,
				   d_jobs, njobs, d_files, d_colpos, d_hdr, d_res);
	ACMP_CHECK();
	/* a block range of the longest stream: its share of the blocks, rounded up, + one for where the cut falls */
	const uint64_t range_columns = max_columns / nranges + 32768;           /* (the kernel strides over the grid: a bound, not a contract) */
	uint64_t gx = ((range_columns < max_columns ? range_columns : max_columns) + COL_THREADS - 1) / COL_THREADS;
	if (gx < 1)
		gx = 1;
	if (gx > 2048)
		gx = 2048;
	for (uint32_t at = 0; at < njobs; at += 65535) {
		const uint32_t n = njobs - at < 65535 ? njobs - at : 65535;
		hipLaunchKernelGGL(acm_parse_columns, dim3((unsigned)gx, n), dim3(COL_THREADS), 0, st,
				   d_jobs + at, d_res + at, d_files, d_colpos, d_hdr, d_idx, d_flags + at, range, nranges, mf ? d_mf : nullptr,
				   mf ? d_pairs : nullptr, mf ? d_blkoff : nullptr);
		ACMP_CHECK();
	}
	return 0;
}

extern "C" int acmk_launch_parse(const AcmParseJob *d_jobs, uint32_t njobs, const uint8_t *d_files,
				 uint32_t *d_colpos, int16_t *d_idx, acmhip_blkhdr *d_hdr,
				 AcmParseResult *d_res, uint32_t *d_flags, uint64_t max_columns, void *stream)
{
	return acmk_launch_parse_range(d_jobs, njobs, d_files, d_colpos, d_idx, d_hdr, d_res, d_flags, max_columns, 0, 1, 0, stream);
}
