/*
 * acm_host_synth.cpp - the synthesis half of the decode path on the HOST (product code: nothing here comes from, links to or calls the test
 * oracle under oracle/; parity is checked against it like every kernel's, tests/test_host_synth.py).
 *
 * Why it exists (Missing 2): the reference decodes anywhere (decode.c:826-876); this library's synthesis lived on the GPU only,
 * so acm_read() into a buffer failed on a box without one, and a small job paid 0.2 s for the HIP runtime to come up before its first
 * sample (one 2-Msample file: 0.13 s against the reference's 0.012 s).  With this file acm_read() / acmtool -d take the host path when no
 * usable device exists, and for streams below acmhip_host_synth_limit() samples while no device handle is open in the process yet.
 * The batch API and the plan API never come here: a caller that asks for the device gets the device or an error.
 *
 * What it computes - decode.c:586-600 (unpack: value = idx * val), :508-577 (juggle / juggle_block), :617-677 (the four writers) - in the
 * same formulation as the kernels (checked against the reference's own juggle_block by tests/test_oracle_vs_ref.py):
 * the synthesis history is nothing but the two previous INPUT rows, so over the flat sample index m = row * cols + col, running on across
 * blocks,
 *     stage k (0 .. level-1), stride s = cols >> (k + 1):   y[m] = 2 x[m - s] + sg (x[m - 2 s] + x[m]),   sg = -1 where (m / s) is odd
 *     after stage 0:  y[m] += 1  where m % (cols / 2) == 0                x[< 0] = 0, arithmetic mod 2^32
 * and any run of rows can be produced from its own staged rows plus the two in front of it.  Here: tiles of rows that fit the level-2
 * cache, each stage one pass over the tile IN PLACE from the top index down (an output only looks at lower indices), eight samples per
 * step with AVX2 where the CPU has it.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "acm_hip.h"

namespace {

std::atomic<uint64_t> g_host_limit{ 1ull << 27 };    /* where the two paths meet for ONE stream through acm_read(): include/acm_hip.h */

/* one stage over n samples in place; w[-2 s .. -1] must be readable (the rows in front, or zeros) */
void stage_scalar(uint32_t *w, size_t n, size_t s)
{
	for (size_t m = n; m-- > 0;) {
		const uint32_t a = w[(ptrdiff_t)m - (ptrdiff_t)s], b = w[(ptrdiff_t)m - 2 * (ptrdiff_t)s], c = w[m];
		w[m] = ((m / s) & 1) ? 2 * a - (b + c) : 2 * a + (b + c);
	}
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) void stage_avx2(uint32_t *w, size_t n, size_t s)
{
	/* n is a multiple of 8 (cols >= 8 here) and w is the first sample of a row: m % (2 s) is known from m alone */
	if (s >= 8) {
		for (size_t m = n; m >= 8;) {
			m -= 8;
			const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(w + m - s));
			const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(w + m - 2 * s));
			const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(w + m));
			const __m256i t = _mm256_add_epi32(b, c), a2 = _mm256_add_epi32(a, a);
			_mm256_storeu_si256(reinterpret_cast<__m256i *>(w + m), ((m / s) & 1) ? _mm256_sub_epi32(a2, t) : _mm256_add_epi32(a2, t));
		}
		return;
	}
	/* strides 4, 2, 1: the sign pattern of eight consecutive samples is the same in every group of eight */
	int32_t pat[8];
	for (int i = 0; i < 8; i++)
		pat[i] = (((size_t)i / s) & 1) ? -1 : 1;
	const __m256i sg = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(pat));
	for (size_t m = n; m >= 8;) {
		m -= 8;
		const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(w + m - s));
		const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(w + m - 2 * s));
		const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(w + m));
		const __m256i t = _mm256_sign_epi32(_mm256_add_epi32(b, c), sg);        /* (-x of INT_MIN stays INT_MIN: the same value mod 2^32) */
		_mm256_storeu_si256(reinterpret_cast<__m256i *>(w + m), _mm256_add_epi32(_mm256_add_epi32(a, a), t));
	}
}
bool have_avx2()
{
	static const bool yes = __builtin_cpu_supports("avx2");
	return yes;
}
#endif

inline void run_stage(uint32_t *w, size_t n, size_t s)
{
#if defined(__x86_64__)
	if ((n & 7) == 0 && have_avx2()) {
		stage_avx2(w, n, s);
		return;
	}
#endif
	stage_scalar(w, n, s);
}

/* out_s16le / _s16be / _u16le / _u16be of decode.c:617-655: (v >> level), no saturation, low 16 bits */
inline void emit(const uint32_t *w, size_t n, unsigned level, unsigned fmt, int16_t *dst)
{
	const uint32_t flip = (fmt & ACMHIP_FMT_U16LE) ? 0x8000u : 0u;
	uint16_t *o = reinterpret_cast<uint16_t *>(dst);
	if (fmt & ACMHIP_FMT_S16BE) {
		for (size_t k = 0; k < n; k++) {
			const uint16_t v = (uint16_t)(((uint32_t)((int32_t)w[k] >> level)) + flip);
			o[k] = (uint16_t)((v >> 8) | (v << 8));
		}
	} else {
		for (size_t k = 0; k < n; k++)
			o[k] = (uint16_t)(((uint32_t)((int32_t)w[k] >> level)) + flip);
	}
}

} // namespace

extern "C" void acmhip_set_host_synth_limit(uint64_t samples)
{
	g_host_limit.store(samples);
}

extern "C" uint64_t acmhip_host_synth_limit(void)
{
	return g_host_limit.load();
}

static int host_synth(const acmhip_stream_desc *s, const int16_t *idx, const acmhip_blkhdr *hdr,
		      const acmhip_patch *patches, size_t npatches, unsigned fmt, int16_t *pcm);

extern "C" int acmhip_host_synth(const acmhip_stream_desc *s, const int16_t *idx, const acmhip_blkhdr *hdr,
				 const acmhip_patch *patches, size_t npatches, unsigned fmt, int16_t *pcm)
{
	try {
		return host_synth(s, idx, hdr, patches, npatches, fmt, pcm);
	} catch (...) {
		return ACMHIP_ERR_NOMEM;        /* (tile buffers, the patch list, threads: nothing else in there throws) */
	}
}

static int host_synth(const acmhip_stream_desc *s, const int16_t *idx, const acmhip_blkhdr *hdr,
		      const acmhip_patch *patches, size_t npatches, unsigned fmt, int16_t *pcm)
{
	if (!s || (!idx && s->nrows) || !hdr || (!pcm && s->n_emit) || (!patches && npatches) || fmt > ACMHIP_FMT_U16BE ||
	    s->level > 15 || s->rows == 0 || s->rows > 4095)
		return ACMHIP_ERR_ARG;
	const unsigned level = s->level;
	const size_t cols = (size_t)1 << level;
	const uint64_t rows_out = (s->n_emit + cols - 1) >> level;
	if ((uint64_t)s->row_begin + rows_out > s->nrows)
		return ACMHIP_ERR_ARG;
	if (s->n_emit == 0)
		return ACMHIP_OK;
	const int16_t *src = idx + s->idx_off;
	const acmhip_blkhdr *h = hdr + s->hdr_off;
	int16_t *dst = pcm + s->pcm_off;

	/* H1 patches (include/acm_hip.h): sorted by sample so that a tile finds its own with one search */
	std::vector<acmhip_patch> ps(patches, patches + npatches);
	std::sort(ps.begin(), ps.end(), [](const acmhip_patch &a, const acmhip_patch &b) { return a.sample < b.sample; });

	/* tile = T rows + the two in front of them, about a megabyte of int32 (the level-2 cache); `cols` zeros in front of the buffer are what
	 * the first two rows of a stream see where their inputs would be.  Tiles are independent (that is the point of the formulation): a
	 * long window is shared out among a few threads, tile by tile */
	const size_t T = std::max<size_t>(2, ((size_t)256 << 10) / cols);
	const uint64_t row_end = s->row_begin + rows_out;
	const uint64_t ntiles = (rows_out + T - 1) / T;
	std::atomic<uint64_t> next{ 0 };
	std::atomic<bool> failed{ false };
	auto work = [&]() {
		std::vector<uint32_t> buf;
		try {
			buf.resize(cols + (T + 2) * cols);
		} catch (...) {
			failed.store(true);             /* (the other workers take the tiles; if none could, the caller hears of it) */
			return;
		}
		uint32_t *const base = buf.data() + cols;
		for (uint64_t t = next.fetch_add(1); t < ntiles; t = next.fetch_add(1)) {
			const uint64_t r0 = s->row_begin + t * T;
			const uint64_t r1 = std::min<uint64_t>(r0 + T, row_end);
			const uint64_t rh = r0 >= 2 ? r0 - 2 : 0;               /* first row of the tile's input */
			const size_t n = (size_t)(r1 - rh) * cols;
			/* unpack: value = idx * val of the row's block (decode.c:592-600, :174-177) */
			for (uint64_t r = rh; r < r1; r++) {
				const uint32_t val = h[r / s->rows].val;
				const int16_t *x = src + (r << level);
				uint32_t *w = base + (size_t)(r - rh) * cols;
				for (size_t c = 0; c < cols; c++)
					w[c] = (uint32_t)((int32_t)x[c] * (int32_t)val);
			}
			if (!ps.empty()) {
				const uint64_t lo = rh << level, hi = r1 << level;
				auto it = std::lower_bound(ps.begin(), ps.end(), lo, [](const acmhip_patch &p, uint64_t v) { return p.sample < v; });
				for (; it != ps.end() && it->sample < hi; ++it)
					base[it->sample - lo] = (uint32_t)it->value;
			}
			memset(buf.data(), 0, cols * sizeof(uint32_t));
			size_t st = cols >> 1;
			for (unsigned k = 0; k < level; k++, st >>= 1) {
				run_stage(base, n, st);
				if (k == 0)
					for (size_t m = 0; m < n; m += cols / 2)       /* decode.c:561-564 */
						base[m] += 1u;
			}
			const size_t skip = (size_t)(r0 - rh) * cols;
			const uint64_t first = (r0 - s->row_begin) << level;
			const size_t want = (size_t)std::min<uint64_t>((r1 - r0) << level, s->n_emit - first);
			emit(base + skip, want, level, fmt, dst + first);
		}
	};
	/* (a thread per 2 Msamples, eight at most, never more than the machine has; short windows - the first ones of every stream - stay on
	 * the caller's thread) */
	unsigned nthreads = (unsigned)std::min<uint64_t>({ 8, s->n_emit >> 21, ntiles, std::max(1u, std::thread::hardware_concurrency()) });
	if (nthreads <= 1) {
		work();
	} else {
		std::vector<std::thread> pool;
		pool.reserve(nthreads);
		try {
			for (unsigned k = 1; k < nthreads; k++)
				pool.emplace_back(work);
		} catch (...) {
			/* (no more threads to be had: the ones that started and this one do the job) */
		}
		work();
		for (std::thread &th : pool)
			th.join();
	}
	if (next.load() < ntiles || (failed.load() && next.load() == 0))
		return ACMHIP_ERR_NOMEM;
	return ACMHIP_OK;
}
