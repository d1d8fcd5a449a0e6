/*
 * acm_synth.c - synthetic ACM v1 bitstream writer (see include/acm_synth.h).
 *
 * Bit order: fields are emitted LSB-first into a little-endian byte stream,
 * which is the order /root/reference/src/decode.c:84-88,120,131-133 reads them.
 */
#include <string.h>

#include "acm_synth.h"

/* ---- PRNG: splitmix64 seeding, xorshift64* stream ---- */
typedef struct { uint64_t s; } rng_t;

static uint64_t splitmix64(uint64_t *x)
{
	uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

static void rng_seed(rng_t *r, uint64_t seed)
{
	uint64_t x = seed;
	r->s = splitmix64(&x);
	if (r->s == 0)
		r->s = 0x1234567887654321ull;
}

static uint64_t rng_next(rng_t *r)
{
	uint64_t x = r->s;
	x ^= x >> 12;
	x ^= x << 25;
	x ^= x >> 27;
	r->s = x;
	return x * 0x2545F4914F6CDD1Dull;
}

/* uniform in [lo, hi] (inclusive); modulo bias is irrelevant for test data */
static uint32_t rng_range(rng_t *r, uint32_t lo, uint32_t hi)
{
	if (hi <= lo)
		return lo;
	return lo + (uint32_t)((rng_next(r) >> 16) % ((uint64_t)hi - lo + 1));
}

/* ---- bit writer ---- */
typedef struct {
	uint8_t *out;
	size_t cap, len;
	uint64_t acc;
	unsigned nbits;
	int overflow;
} bitw_t;

static void bw_put(bitw_t *w, uint32_t v, unsigned n)
{
	w->acc |= (uint64_t)(v & (n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1))) << w->nbits;
	w->nbits += n;
	while (w->nbits >= 8) {
		if (w->len < w->cap)
			w->out[w->len++] = (uint8_t)w->acc;
		else
			w->overflow = 1;
		w->acc >>= 8;
		w->nbits -= 8;
	}
}

static void bw_flush(bitw_t *w)
{
	if (w->nbits)
		bw_put(w, 0, 8 - w->nbits);
}

/* ---- filler payload writers (reader side: decode.c:181-476) ---- */

/* the 11 k/t codes */
static const uint8_t kt_codes[11] = { 17, 18, 19, 20, 21, 22, 23, 24, 26, 27, 29 };

static void put_linear(bitw_t *w, rng_t *r, unsigned code, unsigned rows)
{
	unsigned i;
	for (i = 0; i < rows; i++)
		bw_put(w, (uint32_t)rng_next(r), code);       /* any `code`-bit field is a valid index */
}

/* families with / without the "0 = two zeros" symbol */
static void put_k(bitw_t *w, rng_t *r, unsigned code, unsigned rows)
{
	const int pair = (code == 17 || code == 20 || code == 23 || code == 26);
	unsigned i = 0;
	while (i < rows) {
		unsigned pick = rng_range(r, 0, 9);
		if (pick < 4) {                       /* "0" */
			bw_put(w, 0, 1);
			i += pair ? 2 : 1;
			continue;
		}
		if (pair && pick < 6) {               /* "1 0" -> single zero */
			bw_put(w, 1, 1);
			bw_put(w, 0, 1);
			i++;
			continue;
		}
		bw_put(w, 1, 1);
		if (pair)
			bw_put(w, 1, 1);
		switch (code) {
		case 17: case 18:                     /* b */
			bw_put(w, (uint32_t)rng_next(r), 1);
			break;
		case 20: case 21:                     /* bb */
			bw_put(w, (uint32_t)rng_next(r), 2);
			break;
		case 23: case 24:                     /* 0 b | 1 bb */
			if (rng_next(r) & 1) {
				bw_put(w, 0, 1);
				bw_put(w, (uint32_t)rng_next(r), 1);
			} else {
				bw_put(w, 1, 1);
				bw_put(w, (uint32_t)rng_next(r), 2);
			}
			break;
		default:                              /* 26, 27: bbb */
			bw_put(w, (uint32_t)rng_next(r), 3);
			break;
		}
		i++;
	}
}

static void put_t(bitw_t *w, rng_t *r, unsigned code, unsigned rows)
{
	unsigned i = 0;
	while (i < rows) {
		if (code == 19) {
			bw_put(w, rng_range(r, 0, 26), 5);
			i += 3;
		} else if (code == 22) {
			bw_put(w, rng_range(r, 0, 124), 7);
			i += 3;
		} else {
			bw_put(w, rng_range(r, 0, 120), 7);
			i += 2;
		}
	}
}

static void put_column(bitw_t *w, rng_t *r, unsigned code, unsigned rows)
{
	bw_put(w, code, 5);
	if (code == 0)
		return;
	if (code >= 3 && code <= 16)
		put_linear(w, r, code, rows);
	else if (code == 19 || code == 22 || code == 29)
		put_t(w, r, code, rows);
	else
		put_k(w, r, code, rows);
	/* invalid codes (1,2,25,28,30,31) carry no payload: the reader bails out on them */
}

static unsigned pick_code(const acmsynth_params *p, rng_t *r, unsigned pwr)
{
	unsigned lin_hi = pwr + 1;
	if (p->allow_out_of_range)
		lin_hi = 16;
	if (lin_hi > 16)
		lin_hi = 16;
	if (p->mix == ACMSYNTH_MIX_SINGLE)
		return p->single_code & 31;
	if (p->mix == ACMSYNTH_MIX_UNIFORM) {
		/* 26 valid codes: 0, 3..16, and the 11 k/t codes; linear capped by pwr */
		unsigned k = rng_range(r, 0, 25);
		if (k == 0)
			return 0;
		if (k <= 14) {
			unsigned c = 2 + k;
			if (c > lin_hi)
				c = (lin_hi >= 3) ? rng_range(r, 3, lin_hi) : 0;
			return c;
		}
		return kt_codes[k - 15];
	}
	/* speech-like */
	{
		unsigned k = rng_range(r, 0, 9);
		if (k < 4)
			return (lin_hi >= 3) ? rng_range(r, 3, lin_hi) : 0;
		if (k < 6)
			return 0;
		return kt_codes[rng_range(r, 0, 10)];
	}
}

void acmsynth_defaults(acmsynth_params *p)
{
	memset(p, 0, sizeof(*p));
	p->seed = 0xAC3D0000ull;
	p->level = 7;
	p->rows = 16;
	p->nblocks = 1;
	p->channels = 1;
	p->rate = 22050;
	p->pwr_min = 4;
	p->pwr_max = 12;
	p->val_min = 1;
	p->val_max = 255;
	p->mix = ACMSYNTH_MIX_SPEECH;
}

size_t acmsynth_bound(const acmsynth_params *p)
{
	size_t cols = (size_t)1 << (p->level & 15);
	size_t per_col_bits = 5 + (size_t)p->rows * 16;
	size_t per_block_bits = 20 + cols * per_col_bits;
	return 14 + 28 + ((size_t)p->nblocks * per_block_bits + 7) / 8 + 16;
}

size_t acmsynth_generate(const acmsynth_params *p, uint8_t *out, size_t cap)
{
	bitw_t w;
	rng_t r;
	uint32_t cols, total, b, c;

	if (p->level > 15 || p->rows == 0 || p->rows > 4095 || p->pwr_max > 15 || p->val_max > 65535)
		return 0;
	memset(&w, 0, sizeof(w));
	w.out = out;
	w.cap = cap;
	rng_seed(&r, p->seed);
	cols = 1u << p->level;
	total = p->total_values ? p->total_values : p->nblocks * p->rows * cols;

	if (p->wavc) {
		/* 'WAVC' + 12 words; the reader checks words 0,1 ("V1.0") and word 6 (28), decode.c:689-706 */
		bw_put(&w, 0x564157, 24);
		bw_put(&w, 'C', 8);
		bw_put(&w, 0x3156, 16);
		bw_put(&w, 0x302E, 16);
		bw_put(&w, (total * 2) & 0xFFFF, 16);        /* raw size lo/hi (unchecked) */
		bw_put(&w, (total * 2) >> 16, 16);
		bw_put(&w, 0, 16);                           /* acm size lo/hi (unchecked) */
		bw_put(&w, 0, 16);
		bw_put(&w, 28, 16);
		bw_put(&w, 0, 16);
		bw_put(&w, p->channels, 16);
		bw_put(&w, 16, 16);
		bw_put(&w, p->rate & 0xFFFF, 16);
		bw_put(&w, 0, 16);
	}
	/* 14-byte stream header, decode.c:718-750 */
	bw_put(&w, 0x032897, 24);
	bw_put(&w, 1, 8);
	bw_put(&w, total & 0xFFFF, 16);
	bw_put(&w, total >> 16, 16);
	bw_put(&w, p->channels, 16);
	bw_put(&w, p->rate, 16);
	bw_put(&w, p->level, 4);
	bw_put(&w, p->rows, 12);

	for (b = 0; b < p->nblocks; b++) {
		unsigned pwr = rng_range(&r, p->pwr_min, p->pwr_max);
		if (b == 0 && p->prime_table)
			pwr = 15;
		unsigned val = rng_range(&r, p->val_min, p->val_max);
		bw_put(&w, pwr, 4);
		bw_put(&w, val, 16);
		for (c = 0; c < cols; c++)
			put_column(&w, &r, pick_code(p, &r, pwr), p->rows);
	}
	bw_flush(&w);
	if (w.overflow)
		return 0;
	return w.len;
}
