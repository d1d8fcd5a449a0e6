/*
 * acm_oracle.c - plain-C restatement of the reference ACM decode path
 * (markokr/libacm v1.3: src/decode.c, src/util.c).
 *
 * TEST INFRASTRUCTURE ONLY - see acm_oracle.h.  The shipped decode path lives
 * in libacm_amd/csrc and never touches this file.
 *
 * Parity status: PINNED against the compiled reference (oracle/_ref, built by
 * `make -C oracle ref`) and against tests/golden/.
 *
 * The restatement keeps the reference's observable state machine (what a
 * caller of acm_read / acm_seek_pcm can see, including the odd corners: the
 * single virtual zero byte at EOF, bits that get dropped when a refill comes
 * up short, the amplitude table that is never cleared between blocks) but is
 * organised differently: one bit-source, one column decoder driven by a
 * switch, one synthesis routine.
 */
#include <stdlib.h>
#include <string.h>

#include "acm_oracle.h"

#define CHUNK_BYTES   65536u          /* src/decode.c:29 ACM_BUFLEN */
#define AMP_ENTRIES   0x10000         /* src/decode.c:809 */
#define AMP_ZERO      0x8000          /* src/decode.c:810 midbuf = ampbuf + 0x8000 */
#define STREAM_MAGIC  0x032897u       /* src/libacm.h:28 */
#define WAVC_MAGIC    0x564157u       /* src/decode.c:685 */
#define PLAIN_HDR     14              /* src/util.c:29 */
#define WAVC_HDR      28              /* src/util.c:28 */

struct acmo_stream {
	acmo_info info;
	unsigned total_values;

	/* simulated file + read_func */
	const uint8_t *file;
	size_t file_len;
	size_t file_off;
	unsigned max_read;
	int has_seek;

	/* chunk buffer and bit accumulator (src/libacm.h:80-84) */
	uint8_t *chunk;
	unsigned chunk_len, chunk_pos;
	unsigned acc, acc_bits;
	unsigned chunk_base;        /* buf_start_ofs */
	int at_eof;
	int is_wavc;

	unsigned block_len, wrap_len;
	int32_t *block, *wrap, *amp;

	int block_ready;
	unsigned stream_pos, block_pos;
};

/* ------------------------------------------------------------------ */
/* bit source                                                          */
/* ------------------------------------------------------------------ */

/* src/decode.c:41-67 load_buf: one read_func call per refill; the first
 * zero-length read turns into exactly one virtual 0x00 byte. */
static void next_chunk(acmo_stream *s)
{
	size_t n;

	if (s->at_eof)
		return;                       /* :45-46, nothing changes */
	s->chunk_base += s->chunk_len;        /* :48 */
	n = s->file_len - s->file_off;
	if (n > CHUNK_BYTES)
		n = CHUNK_BYTES;
	if (s->max_read && n > s->max_read)
		n = s->max_read;
	if (n == 0) {                         /* :57-61 */
		s->at_eof = 1;
		s->chunk[0] = 0;
		s->chunk_len = 1;
	} else {
		memcpy(s->chunk, s->file + s->file_off, n);
		s->file_off += n;
		s->chunk_len = (unsigned)n;
	}
	s->chunk_pos = 0;
}

/* GET_BITS_NOERR + get_bits_reload + load_bits, src/decode.c:69-144.
 * n <= 31.  Returns the field or a negative error. */
static int take(acmo_stream *s, unsigned n)
{
	unsigned lo, have, need, word, wbits, left, i;
	int v;

	if (s->acc_bits >= n) {               /* :138-141 */
		v = (int)(s->acc & ((1u << n) - 1));
		s->acc >>= n;
		s->acc_bits -= n;
		return v;
	}

	lo = s->acc;
	have = s->acc_bits;
	need = n - have;                      /* :113-115 */

	left = s->chunk_len - s->chunk_pos;
	if (left >= 4) {                      /* :117-121 */
		const uint8_t *p = s->chunk + s->chunk_pos;
		word = (unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) | ((unsigned)p[3] << 24);
		wbits = 32;
		s->chunk_pos += 4;
	} else {
		/* :69-106 - pick up the 0..3 byte tail WITHOUT advancing, refill,
		 * then top up to 32 bits from wherever chunk_pos now points */
		word = 0;
		wbits = 0;
		for (i = 0; i < left; i++) {
			word |= (unsigned)s->chunk[s->chunk_pos + i] << wbits;
			wbits += 8;
		}
		next_chunk(s);
		while (wbits < 32 && s->chunk_pos != s->chunk_len) {
			word |= (unsigned)s->chunk[s->chunk_pos++] << wbits;
			wbits += 8;
		}
		s->acc = word;                /* :103-104: state is replaced even on failure */
		s->acc_bits = wbits;
		if (wbits < need)             /* :125-126; the `have` old bits are gone */
			return ACMO_ERR_UNEXPECTED_EOF;
	}
	v = (int)(lo | ((word & ((1u << need) - 1)) << have));   /* :131 */
	s->acc = word >> need;
	s->acc_bits = wbits - need;
	return v;
}

/* GET_BITS_EXPECT_EOF, src/decode.c:154-163 */
static int take_or_clean_eof(acmo_stream *s, unsigned n)
{
	int v = take(s, n);
	if (v == ACMO_ERR_UNEXPECTED_EOF)
		return ACMO_CLEAN_EOF;
	return v;
}

/* ------------------------------------------------------------------ */
/* header                                                              */
/* ------------------------------------------------------------------ */

#define NEED(var, s, n) do { int t_ = take((s), (n)); if (t_ < 0) return t_; (var) = (unsigned)t_; } while (0)

/* src/decode.c:687-710.  Twelve 16-bit words follow "WAVC"; only "V1.0"
 * (words 0,1) and the 28 in word 6 are actually checked. */
static int parse_wavc_tail(acmo_stream *s)
{
	unsigned w[12], i;
	for (i = 0; i < 12; i++)
		NEED(w[i], s, 16);
	if (w[0] != 0x3156 || w[1] != 0x302E)
		return -1;
	if (w[6] != 28)
		return -1;
	s->is_wavc = 1;
	return 0;
}

/* src/decode.c:712-752 */
static int parse_header(acmo_stream *s)
{
	unsigned v, hi;

	NEED(v, s, 24);
	if (v == WAVC_MAGIC) {
		NEED(v, s, 8);
		if (v != 'C')
			return ACMO_ERR_NOT_ACM;
		if (parse_wavc_tail(s) < 0)
			return ACMO_ERR_NOT_ACM;
		NEED(v, s, 24);
	}
	if (v != STREAM_MAGIC)
		return ACMO_ERR_NOT_ACM;
	s->info.acm_id = v;
	NEED(s->info.acm_version, s, 8);
	if (s->info.acm_version != 1)
		return ACMO_ERR_NOT_ACM;
	NEED(s->total_values, s, 16);
	NEED(hi, s, 16);
	s->total_values += hi << 16;
	if (s->total_values == 0)
		return ACMO_ERR_NOT_ACM;
	NEED(s->info.channels, s, 16);
	if (s->info.channels < 1 || s->info.channels > 2)
		return ACMO_ERR_NOT_ACM;
	s->info.acm_channels = s->info.channels;
	NEED(s->info.rate, s, 16);
	if (s->info.rate < 4096)
		return ACMO_ERR_NOT_ACM;
	NEED(s->info.acm_level, s, 4);
	NEED(s->info.acm_rows, s, 12);
	if (s->info.acm_rows == 0)
		return ACMO_ERR_NOT_ACM;
	return 0;
}

/* src/decode.c:758-824 */
int acmo_open_mem(acmo_stream **out, const uint8_t *data, size_t len,
		  int force_chans, unsigned max_read, int seekable)
{
	acmo_stream *s = calloc(1, sizeof(*s));
	if (!s)
		return ACMO_ERR_OTHER;
	s->file = data;
	s->file_len = len;
	s->max_read = max_read;
	s->has_seek = seekable;
	s->chunk = malloc(CHUNK_BYTES);
	if (!s->chunk) {
		free(s);
		return ACMO_ERR_OTHER;
	}
	if (parse_header(s) < 0) {            /* :783-785: every header failure is NOT_ACM */
		acmo_close(s);
		return ACMO_ERR_NOT_ACM;
	}
	if (force_chans > 0)                  /* :795-798 */
		s->info.channels = (unsigned)force_chans;
	else if (force_chans == -1 && !s->is_wavc && s->info.channels < 2)
		s->info.channels = 2;

	s->info.acm_cols = 1u << s->info.acm_level;          /* :802-804 */
	s->wrap_len = 2 * s->info.acm_cols - 2;
	s->block_len = s->info.acm_rows * s->info.acm_cols;

	s->block = malloc((size_t)s->block_len * sizeof(int32_t));
	s->wrap = calloc(s->wrap_len ? s->wrap_len : 1, sizeof(int32_t));   /* :812 zeroed */
	/* The reference leaves this table uninitialised (:809); a stream that
	 * indexes an entry no block has written reads heap garbage there.  We use
	 * zeros, which is also what a fresh glibc mmap'ed allocation holds. */
	s->amp = calloc(AMP_ENTRIES, sizeof(int32_t));
	if (!s->block || !s->wrap || !s->amp) {
		acmo_close(s);
		return ACMO_ERR_OTHER;
	}
	*out = s;
	return ACMO_OK;
}

void acmo_close(acmo_stream *s)
{
	if (!s)
		return;
	free(s->chunk);
	free(s->block);
	free(s->wrap);
	free(s->amp);
	free(s);
}

/* ------------------------------------------------------------------ */
/* fillers (src/decode.c:168-502)                                      */
/* ------------------------------------------------------------------ */

static const int pm1[2]  = { -1, +1 };                          /* :168 */
static const int near2[4] = { -2, -1, +1, +2 };                 /* :169 */
static const int far2[4]  = { -3, -2, +2, +3 };                 /* :170 */
static const int wide3[8] = { -4, -3, -2, -1, +1, +2, +3, +4 }; /* :171 */

#define BITS(var, n) do { int t_ = take(s, (n)); if (t_ < 0) return t_; (var) = (unsigned)t_; } while (0)
/* set_pos, :174-177 */
#define PUT(r, idx) (s->block[((size_t)(r) << s->info.acm_level) + col] = mid[(idx)])

/* Decode one column with filler code `code` (0..31).  Returns 1 or <0. */
static int decode_column(acmo_stream *s, unsigned code, unsigned col)
{
	const unsigned rows = s->info.acm_rows;
	const int32_t *mid = s->amp + AMP_ZERO;
	unsigned r = 0, b;

	switch (code) {
	case 0:                                   /* f_zero :181-188 */
		for (r = 0; r < rows; r++)
			PUT(r, 0);
		return 1;

	case 3: case 4: case 5: case 6: case 7: case 8: case 9: case 10:
	case 11: case 12: case 13: case 14: case 15: case 16: {   /* f_linear :196-206 */
		int centre = 1 << (code - 1);
		for (r = 0; r < rows; r++) {
			BITS(b, code);
			PUT(r, (int)b - centre);
		}
		return 1;
	}

	case 17: case 20: case 23: case 26:       /* k13 :208, k24 :252, k35 :297, k45 :359 */
		/* family with a "0 = two zeros" symbol */
		while (r < rows) {
			BITS(b, 1);
			if (b == 0) {
				PUT(r, 0);
				r++;
				if (r >= rows)
					break;            /* second zero falls off the column */
				PUT(r, 0);
				r++;
				continue;
			}
			BITS(b, 1);
			if (b == 0) {
				PUT(r, 0);
				r++;
				continue;
			}
			if (code == 17) {             /* 1 1 b */
				BITS(b, 1);
				PUT(r, pm1[b]);
			} else if (code == 20) {      /* 1 1 bb */
				BITS(b, 2);
				PUT(r, near2[b]);
			} else if (code == 23) {      /* 1 1 0 b | 1 1 1 bb */
				BITS(b, 1);
				if (b == 0) {
					BITS(b, 1);
					PUT(r, pm1[b]);
				} else {
					BITS(b, 2);
					PUT(r, far2[b]);
				}
			} else {                      /* 26: 1 1 bbb */
				BITS(b, 3);
				PUT(r, wide3[b]);
			}
			r++;
		}
		return 1;

	case 18: case 21: case 24: case 27:       /* k12 :234, k23 :279, k34 :333, k44 :387 */
		for (r = 0; r < rows; r++) {
			BITS(b, 1);
			if (b == 0) {
				PUT(r, 0);
				continue;
			}
			if (code == 18) {             /* 1 b */
				BITS(b, 1);
				PUT(r, pm1[b]);
			} else if (code == 21) {      /* 1 bb */
				BITS(b, 2);
				PUT(r, near2[b]);
			} else if (code == 24) {      /* 1 0 b | 1 1 bb */
				BITS(b, 1);
				if (b == 0) {
					BITS(b, 1);
					PUT(r, pm1[b]);
				} else {
					BITS(b, 2);
					PUT(r, far2[b]);
				}
			} else {                      /* 27: 1 bbb */
				BITS(b, 3);
				PUT(r, wide3[b]);
			}
		}
		return 1;

	case 19:                                  /* f_t15 :405-429, three base-3 digits in 5 bits */
	case 22: {                                /* f_t27 :431-455, three base-5 digits in 7 bits */
		const unsigned base = (code == 19) ? 3 : 5;
		const unsigned width = (code == 19) ? 5 : 7;
		const int off = (int)(base / 2);
		while (r < rows) {
			unsigned k;
			BITS(b, width);
			if (b >= base * base * base)
				return ACMO_ERR_CORRUPT;
			for (k = 0; k < 3 && r < rows; k++, r++) {
				PUT(r, (int)(b % base) - off);
				b /= base;
			}
		}
		return 1;
	}

	case 29:                                  /* f_t37 :457-476, two base-11 digits in 7 bits */
		while (r < rows) {
			BITS(b, 7);
			if (b >= 121)
				return ACMO_ERR_CORRUPT;
			PUT(r, (int)(b % 11) - 5);
			r++;
			if (r >= rows)
				break;
			PUT(r, (int)(b / 11) - 5);
			r++;
		}
		return 1;

	default:                                  /* 1,2,25,28,30,31: f_bad :190-194 */
		return ACMO_ERR_CORRUPT;
	}
}

/* block header + amplitude table + fill_block: src/decode.c:586-604, 491-502 */
static int fill_block(acmo_stream *s, int *pwr_out, int *val_out)
{
	int pwr, val, v;
	unsigned n, i, col;
	int32_t *mid = s->amp + AMP_ZERO;
	uint32_t x;

	s->block_ready = 0;
	s->block_pos = 0;

	pwr = take_or_clean_eof(s, 4);
	if (pwr < 0)
		return pwr;
	val = take_or_clean_eof(s, 16);
	if (val < 0)
		return val;

	/* :592-600 - entries [-2^pwr, 2^pwr) are (re)written, everything else
	 * keeps whatever an earlier block left there.  Unsigned arithmetic: the
	 * reference's signed accumulation wraps the same way on every target it
	 * is built for. */
	n = 1u << pwr;
	for (i = 0, x = 0; i < n; i++, x += (uint32_t)val)
		mid[i] = (int32_t)x;
	for (i = 1, x = (uint32_t)-val; i <= n; i++, x -= (uint32_t)val)
		mid[-(int)i] = (int32_t)x;

	for (col = 0; col < s->info.acm_cols; col++) {
		v = take_or_clean_eof(s, 5);  /* :496 */
		if (v < 0)
			return v;
		v = decode_column(s, (unsigned)v, col);
		if (v < 0)
			return v;
	}
	if (pwr_out)
		*pwr_out = pwr;
	if (val_out)
		*val_out = val;
	return 1;
}

/* ------------------------------------------------------------------ */
/* synthesis (src/decode.c:508-577)                                    */
/* ------------------------------------------------------------------ */

/* juggle, :508-526: one butterfly stage over a view of `count` rows by
 * `width` columns; hist holds the two previous inputs of every column. */
static void stage(int32_t *hist, int32_t *base, unsigned width, unsigned count)
{
	unsigned c, k;
	for (c = 0; c < width; c++) {
		int32_t *p = base + c;
		uint32_t a = (uint32_t)hist[2 * c], b = (uint32_t)hist[2 * c + 1];
		for (k = 0; k < count / 2; k++) {
			uint32_t x = (uint32_t)p[0];
			uint32_t y = (uint32_t)p[width];
			p[0] = (int32_t)(2 * b + (a + x));         /* :518 */
			p[width] = (int32_t)(2 * x - (b + y));     /* :519 */
			p += 2 * width;
			a = x;
			b = y;
		}
		hist[2 * c] = (int32_t)a;
		hist[2 * c + 1] = (int32_t)b;
	}
}

void acmo_juggle_block(unsigned level, unsigned rows, int32_t *block, int32_t *wrap)
{
	const unsigned cols = 1u << level;
	unsigned slab, left, take_rows, width, count, k;
	int32_t *at = block;

	if (level == 0)                           /* :534-535 */
		return;
	slab = (level > 9) ? 1 : (2048u >> level) - 2;   /* :538-541 */

	for (left = rows; ; left -= slab, at += (size_t)slab << level) {
		int32_t *h = wrap;                /* :550 */
		take_rows = left < slab ? left : slab;
		width = cols / 2;
		count = take_rows * 2;
		stage(h, at, width, count);       /* :558 */
		h += 2 * width;
		for (k = 0; k < count; k++)       /* :561-564 */
			at[(size_t)k * width] += 1;
		while (width > 1) {               /* :566-571 */
			width /= 2;
			count *= 2;
			stage(h, at, width, count);
			h += 2 * width;
		}
		if (left <= slab)                 /* :572-573 */
			break;
	}
}

/* ------------------------------------------------------------------ */
/* write-out (src/decode.c:617-677)                                    */
/* ------------------------------------------------------------------ */

int acmo_output(const int32_t *src, unsigned char *dst, int n, int level,
		int bigendianp, int wordlen, int sgned)
{
	int i;
	if (wordlen != 2)
		return ACMO_ERR_BADFMT;           /* :661, :676 */
	for (i = 0; i < n; i++) {
		int v = src[i] >> level;          /* arithmetic shift, :620 */
		if (!sgned)
			v += 0x8000;              /* :640 */
		if (bigendianp) {
			dst[2 * i] = (unsigned char)((v >> 8) & 0xFF);
			dst[2 * i + 1] = (unsigned char)(v & 0xFF);
		} else {
			dst[2 * i] = (unsigned char)(v & 0xFF);
			dst[2 * i + 1] = (unsigned char)((v >> 8) & 0xFF);
		}
	}
	return 2 * n;
}

/* ------------------------------------------------------------------ */
/* stream control (src/decode.c:580-611, 826-876)                      */
/* ------------------------------------------------------------------ */

static int next_block(acmo_stream *s)
{
	int rc = fill_block(s, NULL, NULL);
	if (rc <= 0)
		return rc;                        /* :603-604 */
	acmo_juggle_block(s->info.acm_level, s->info.acm_rows, s->block, s->wrap);
	s->block_ready = 1;
	return 1;
}

int acmo_fill_next_block(acmo_stream *s, int32_t *raw_out, int *pwr, int *val)
{
	int rc = fill_block(s, pwr, val);
	if (rc <= 0)
		return rc;
	if (raw_out)
		memcpy(raw_out, s->block, (size_t)s->block_len * sizeof(int32_t));
	return 1;
}

int acmo_read(acmo_stream *s, void *dst, unsigned nbytes, int bigendianp, int wordlen, int sgned)
{
	int words, room, got, rc;

	if (wordlen != 2)
		return ACMO_ERR_BADFMT;           /* :832-835 */
	words = (int)(nbytes / 2);
	if (s->stream_pos >= s->total_values)
		return 0;                         /* :837-838 */
	if (!s->block_ready) {                    /* :840-846 */
		rc = next_block(s);
		if (rc == ACMO_CLEAN_EOF)
			return 0;
		if (rc < 0)
			return rc;
	}
	room = (int)(s->block_len - s->block_pos);        /* :849-851 */
	if (room < words)
		words = room;
	if (s->stream_pos + (unsigned)words > s->total_values)   /* :853-854 */
		words = (int)(s->total_values - s->stream_pos);
	if (s->info.channels > 1)                 /* :856-857 */
		words -= words % (int)s->info.channels;

	if (dst)
		got = acmo_output(s->block + s->block_pos, dst, words, (int)s->info.acm_level,
				  bigendianp, wordlen, sgned);
	else
		got = words * wordlen;            /* :865-866 */
	if (got >= 0) {                           /* :868-873 */
		s->stream_pos += (unsigned)words;
		s->block_pos += (unsigned)words;
		if (s->block_pos == s->block_len)
			s->block_ready = 0;
	}
	return got;
}

/* src/util.c:258-277 */
int acmo_read_loop(acmo_stream *s, void *dst, unsigned nbytes, int bigendianp, int wordlen, int sgned)
{
	unsigned char *p = dst;
	int got = 0, rc;
	while (nbytes > 0) {
		rc = acmo_read(s, p, nbytes, bigendianp, wordlen, sgned);
		if (rc > 0) {
			if (p)
				p += rc;
			got += rc;
			nbytes -= (unsigned)rc;
			continue;
		}
		if (rc < 0 && got == 0)
			return rc;
		break;
	}
	return got;
}

/* src/util.c:214-253 */
int acmo_seek_pcm(acmo_stream *s, unsigned pcm_pos)
{
	unsigned target = pcm_pos * s->info.channels;

	if (target < s->stream_pos) {
		if (!s->has_seek)
			return ACMO_ERR_NOT_SEEKABLE;
		/* seek_func(io_arg, 14 [+28], SEEK_SET), :223-228 */
		s->file_off = PLAIN_HDR + (s->is_wavc ? WAVC_HDR : 0);
		if (s->file_off > s->file_len)
			s->file_off = s->file_len;
		s->at_eof = 0;
		s->chunk_pos = 0;
		s->chunk_len = 0;
		s->acc_bits = 0;
		s->acc = 0;
		s->stream_pos = 0;
		s->block_pos = 0;
		s->block_ready = 0;
		s->chunk_base = PLAIN_HDR;        /* :239 (yes, 14 even for WAVC) */
		memset(s->wrap, 0, (size_t)s->wrap_len * sizeof(int32_t));
	}
	while (s->stream_pos < target) {          /* :243-251 */
		unsigned step = 2048;
		int rc;
		if (s->stream_pos + step > target)
			step = target - s->stream_pos;
		rc = acmo_read(s, NULL, step * 2, 0, 2, 1);
		if (rc < 1)
			break;
	}
	return (int)(s->stream_pos / s->info.channels);
}

/* src/util.c:121-131 */
static unsigned to_ms(acmo_stream *s, unsigned long long pcm)
{
	return (unsigned)(pcm * 1000 / s->info.rate);
}

int acmo_seek_time(acmo_stream *s, unsigned pos_ms)
{
	unsigned long long pcm = (unsigned long long)pos_ms * s->info.rate / 1000;
	int rc = acmo_seek_pcm(s, (unsigned)pcm);     /* util.c:206-212 */
	if (rc <= 0)
		return rc;
	return (int)to_ms(s, (unsigned)rc);
}

/* getters, src/util.c:137-200 */
const acmo_info *acmo_get_info(acmo_stream *s) { return &s->info; }
int acmo_seekable(acmo_stream *s) { return s->file_len > 0; }
unsigned acmo_rate(acmo_stream *s) { return s->info.rate; }
unsigned acmo_channels(acmo_stream *s) { return s->info.channels; }
unsigned acmo_raw_total(acmo_stream *s) { return (unsigned)s->file_len; }
unsigned acmo_raw_tell(acmo_stream *s) { return s->chunk_base + s->chunk_pos; }
unsigned acmo_pcm_total(acmo_stream *s) { return s->total_values / s->info.channels; }
unsigned acmo_pcm_tell(acmo_stream *s) { return s->stream_pos / s->info.channels; }
unsigned acmo_time_total(acmo_stream *s) { return to_ms(s, acmo_pcm_total(s)); }
unsigned acmo_time_tell(acmo_stream *s) { return to_ms(s, acmo_pcm_tell(s)); }
unsigned acmo_total_values(acmo_stream *s) { return s->total_values; }
unsigned acmo_block_len(acmo_stream *s) { return s->block_len; }

unsigned acmo_bitrate(acmo_stream *s)             /* util.c:157-170 */
{
	unsigned long long ms, bits;
	if (acmo_raw_total(s) == 0)
		return 13000;
	ms = acmo_time_total(s);
	if (ms == 0)
		return 0;
	bits = (unsigned)(8u * acmo_raw_total(s));    /* 32-bit product, as in util.c:166 */
	return (unsigned)(1000 * bits / ms);
}

const char *acmo_strerror(int err)                /* util.c:34-52, typo included */
{
	static const char *const text[] = {
		"No error", "ACM error", "Cannot open file", "Not an ACM file", "Read error",
		"Bad format", "Corrupt file", "Unexcpected EOF", "Stream not seekable"
	};
	if (err > 0 || -err >= (int)(sizeof(text) / sizeof(text[0])))
		return "Unknown error";
	return text[-err];
}

/* ------------------------------------------------------------------ */
/* whole-file helpers                                                  */
/* ------------------------------------------------------------------ */

long acmo_decode_all(const uint8_t *data, size_t len, int force_chans,
		     int16_t *pcm, size_t cap_words, unsigned step_bytes,
		     int bigendianp, int sgned, int *status)
{
	acmo_stream *s;
	size_t done = 0;
	int rc = acmo_open_mem(&s, data, len, force_chans, 0, 1);
	if (rc < 0) {
		if (status)
			*status = rc;
		return 0;
	}
	if (step_bytes == 0)
		step_bytes = 8192;
	for (;;) {
		size_t room = (cap_words - done) * 2;
		unsigned ask = room < step_bytes ? (unsigned)room : step_bytes;
		if (ask < 2) {
			rc = 0;
			break;
		}
		rc = acmo_read_loop(s, (unsigned char *)pcm + done * 2, ask, bigendianp, 2, sgned);
		if (rc <= 0)
			break;
		done += (size_t)rc / 2;
	}
	if (status)
		*status = rc;
	acmo_close(s);
	return (long)done;
}

long acmo_decode_discard(const uint8_t *data, size_t len, int force_chans, int *status)
{
	acmo_stream *s;
	long done = 0;
	static unsigned char sink[16384];
	int rc = acmo_open_mem(&s, data, len, force_chans, 0, 1);
	if (rc < 0) {
		if (status)
			*status = rc;
		return 0;
	}
	for (;;) {
		rc = acmo_read_loop(s, sink, 8192, 0, 2, 1);   /* acmtool.c:269-275 */
		if (rc <= 0)
			break;
		done += rc / 2;
	}
	if (status)
		*status = rc;
	acmo_close(s);
	return done;
}
