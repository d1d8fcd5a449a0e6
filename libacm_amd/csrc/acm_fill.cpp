/*
 * acm_fill.cpp - sequential bitstream reader + filler parsers -> staged form.
 * See acm_fill.h.  Reference behaviour being reproduced: /root/reference/src/decode.c
 * (line numbers below refer to it).
 */
#include "acm_fill.h"

#include <string.h>

#include <vector>

namespace acmfill {

namespace {

/* One read_func call per refill; the first empty read becomes a single
 * virtual zero byte (:41-67).  Returns 0 or ACM_ERR_READ_ERR. */
int pull_chunk(ACMStream *s)
{
	if (s->file_eof)
		return 0;
	s->buf_start_ofs += s->buf_size;
	int got = 0;
	if (s->io.read_func)
		got = s->io.read_func(s->buf, 1, (int)s->buf_max, s->io_arg);
	if (got < 0)
		return ACM_ERR_READ_ERR;
	if (got == 0) {
		s->file_eof = 1;
		s->buf[0] = 0;
		s->buf_size = 1;
	} else {
		s->buf_size = (unsigned)got;
	}
	s->buf_pos = 0;
	return 0;
}

/* LSB-first bit cursor; the accumulator lives in registers while a block is
 * being parsed and is written back to the public struct on exit. */
struct BitCursor {
	static constexpr bool kTableDriven = false;     /* exact reader: every field through get() */
	ACMStream *s;
	uint32_t acc;
	unsigned avail;

	explicit BitCursor(ACMStream *st) : s(st), acc(st->bit_data), avail(st->bit_avail) {}
	void commit() const
	{
		s->bit_data = acc;
		s->bit_avail = avail;
	}

	/* n <= 31; value or negative error */
	inline int get(unsigned n)
	{
		if (avail >= n) {
			const int v = (int)(acc & ((1u << n) - 1));
			acc >>= n;
			avail -= n;
			return v;
		}
		return get_across(n);
	}

	/* the accumulator does not hold n bits: take what is there and top up from
	 * the buffer 32 bits at a time (:108-135), refilling the buffer through
	 * read_func when fewer than 4 bytes remain (:69-106) */
	__attribute__((noinline)) int get_across(unsigned n)
	{
		const uint32_t low = acc;
		const unsigned have = avail;
		const unsigned need = n - have;
		uint32_t word;
		unsigned wbits;
		const unsigned left = s->buf_size - s->buf_pos;

		if (left >= 4) {
			const unsigned char *p = s->buf + s->buf_pos;
			word = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
			wbits = 32;
			s->buf_pos += 4;
		} else {
			word = 0;
			wbits = 0;
			for (unsigned i = 0; i < left; i++, wbits += 8)
				word |= (uint32_t)s->buf[s->buf_pos + i] << wbits;
			const int rc = pull_chunk(s);
			if (rc < 0)
				return rc;              /* accumulator untouched, as in the reference */
			while (wbits < 32 && s->buf_pos != s->buf_size) {
				word |= (uint32_t)s->buf[s->buf_pos++] << wbits;
				wbits += 8;
			}
			acc = word;                     /* the old partial bits are dropped even on failure */
			avail = wbits;
			if (wbits < need)
				return ACM_ERR_UNEXPECTED_EOF;
		}
		const int v = (int)(low | ((word & ((1u << need) - 1)) << have));
		acc = word >> need;
		avail = wbits - need;
		return v;
	}

	/* reads where running dry is a legal end of stream (:154-163) */
	inline int get_or_end(unsigned n)
	{
		const int v = get(n);
		return v == ACM_ERR_UNEXPECTED_EOF ? kCleanEof : v;
	}
};

/*
 * Fast cursor: plain bit offset into the current refill buffer, fields fetched with one unaligned 8-byte
 * load.  Only used while the column being parsed lies well inside the buffer, where the reference reader's
 * state is a pure function of the bit offset (it tops up exactly 32 bits whenever a field does not fit, so
 * buf_pos stays congruent to its phase mod 4 and bit_avail = 8*buf_pos - offset in [0, 31]); enter()/leave()
 * convert between the two views.  Near the end of the buffer, across refills, at EOF and after short reads
 * the exact BitCursor above takes over.
 */
struct FastCursor {
	static constexpr bool kTableDriven = true;      /* the window can be peeked: k/t symbols come from a table */
	const unsigned char *buf;
	uint64_t bit;           /* offset of the next unread bit in buf */
	uint64_t win;           /* the next `have` bits, LSB first */
	unsigned have;

	inline void refill()
	{
		memcpy(&win, buf + (bit >> 3), sizeof(win));
		win >>= (bit & 7);
		have = 64 - (unsigned)(bit & 7);        /* >= 57 */
	}
	inline int get(unsigned n)                      /* n <= 31 */
	{
		if (have < n)
			refill();
		const int v = (int)((uint32_t)win & ((1u << n) - 1));
		win >>= n;
		have -= n;
		bit += n;
		return v;
	}
};

/* bits a column can take at most, code included: linear code 16 */
inline uint64_t worst_column_bits(unsigned rows) { return 5 + (uint64_t)rows * 16; }

inline bool fast_enter(const BitCursor &bc, FastCursor &fc, uint64_t need_bits)
{
	const ACMStream *s = bc.s;
	const int64_t at = (int64_t)8 * s->buf_pos - (int64_t)bc.avail;     /* next unread bit, relative to the buffer */
	if (at < 0 || s->buf_pos < 4 || (uint64_t)at + need_bits + 64 > (uint64_t)8 * s->buf_size)
		return false;
	fc.buf = s->buf;
	fc.bit = (uint64_t)at;
	fc.refill();
	return true;
}

inline void fast_leave(BitCursor &bc, const FastCursor &fc, unsigned phase)
{
	ACMStream *s = bc.s;
	/* the reference sits on the first 4-byte boundary (of its phase) at or behind the cursor */
	unsigned pos = (unsigned)((fc.bit + 7) >> 3);
	pos += (phase - pos) & 3u;
	const unsigned avail = 8 * pos - (unsigned)fc.bit;
	uint64_t w;
	memcpy(&w, s->buf + (fc.bit >> 3), sizeof(w));
	s->buf_pos = pos;
	bc.avail = avail;
	bc.acc = avail ? (uint32_t)(w >> (fc.bit & 7)) & (0xFFFFFFFFu >> (32 - avail)) : 0u;
}

/* largest |index| a filler code can produce; 0xFFFF marks the invalid codes (:480-489) */
inline unsigned code_reach(unsigned code)
{
	static const uint16_t reach[32] = {
		0, 0xFFFF, 0xFFFF, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384,
		32768, 1, 1, 1, 2, 2, 2, 3, 3, 0xFFFF, 4, 4, 0xFFFF, 5, 0xFFFF, 0xFFFF
	};
	return reach[code];
}

/* symbol tables of the k-fillers (:168-171) */
const int8_t kSign1[2] = { -1, 1 };
const int8_t kNear2[4] = { -2, -1, 1, 2 };
const int8_t kFar2[4] = { -3, -2, 2, 3 };
const int8_t kWide3[8] = { -4, -3, -2, -1, 1, 2, 3, 4 };

#define TAKE(var, n) do { const int t_ = bc.get(n); if (t_ < 0) return t_; (var) = (unsigned)t_; } while (0)

/*
 * Table-driven k/t fillers for the fast cursor.  One entry per (filler, next 7 bits) = the symbol at the head of
 * those bits (:217-476): bits 0-2 length, 3-4 rows produced (1..3), 5-8 / 9-12 / 13-16 the indices + 8, bit 17 =
 * the reference rejects this symbol (ternary value out of range).  Decoding a symbol is then one look-up and three
 * unconditional stores - no data-dependent branch, where the bit grammar costs one mispredict per symbol.
 */
struct SymbolTable {
	uint32_t e[11][128];
	static int slot(unsigned code) { return code <= 24 ? (int)code - 17 : code <= 27 ? (int)code - 18 : 10; }
	SymbolTable()
	{
		static const unsigned codes[11] = { 17, 18, 19, 20, 21, 22, 23, 24, 26, 27, 29 };
		for (int k = 0; k < 11; k++)
			for (unsigned bits = 0; bits < 128; bits++)
				e[k][bits] = entry(codes[k], bits);
	}
	static uint32_t entry(unsigned code, unsigned bits)
	{
		const bool b0 = bits & 1, b1 = bits & 2, b2 = bits & 4;
		unsigned len = 1, cnt = 1, bad = 0;
		int v0 = 0, v1 = 0, v2 = 0;
		switch (code) {
		case 17: if (!b0) cnt = 2; else if (!b1) len = 2; else { len = 3; v0 = kSign1[b2]; } break;
		case 18: if (b0) { len = 2; v0 = kSign1[b1]; } break;
		case 19: { const unsigned b = bits & 31; len = 5; cnt = 3; bad = b >= 27;
			   v0 = (int)(b % 3) - 1; v1 = (int)(b / 3 % 3) - 1; v2 = (int)(b / 9 % 3) - 1; break; }
		case 20: if (!b0) cnt = 2; else if (!b1) len = 2; else { len = 4; v0 = kNear2[(bits >> 2) & 3]; } break;
		case 21: if (b0) { len = 3; v0 = kNear2[(bits >> 1) & 3]; } break;
		case 22: { const unsigned b = bits & 127; len = 7; cnt = 3; bad = b >= 125;
			   v0 = (int)(b % 5) - 2; v1 = (int)(b / 5 % 5) - 2; v2 = (int)(b / 25 % 5) - 2; break; }
		case 23: if (!b0) cnt = 2; else if (!b1) len = 2;
			 else if (!b2) { len = 4; v0 = kSign1[(bits >> 3) & 1]; } else { len = 5; v0 = kFar2[(bits >> 3) & 3]; } break;
		case 24: if (b0) { if (!b1) { len = 3; v0 = kSign1[b2]; } else { len = 4; v0 = kFar2[(bits >> 2) & 3]; } } break;
		case 26: if (!b0) cnt = 2; else if (!b1) len = 2; else { len = 5; v0 = kWide3[(bits >> 2) & 7]; } break;
		case 27: if (b0) { len = 4; v0 = kWide3[(bits >> 1) & 7]; } break;
		default: { const unsigned b = bits & 127; len = 7; cnt = 2; bad = b >= 121;
			   v0 = (int)(b % 11) - 5; v1 = (int)(b / 11 % 11) - 5; break; }
		}
		return len | cnt << 3 | (uint32_t)(v0 + 8) << 5 | (uint32_t)(v1 + 8) << 9 | (uint32_t)(v2 + 8) << 13 | bad << 17;
	}
};

/*
 * The k-fillers' symbols are one to five bits long, so the next 8 bits usually hold several of them: one entry per
 * (k-filler, next 8 bits) = every symbol that lies wholly inside those bits, up to 8 rows' worth - bits 0-3 the bits they
 * take, 4-7 the rows they produce, then eight nibbles (index + 8; rows not produced hold 8 = index 0).  A look-up and eight
 * unconditional stores replace three to eight single-symbol steps; used while at least eight rows of the column are left
 * (the stores stay inside the column, and the "two zeros" symbol cannot reach past its end), the rest goes symbol by symbol.
 */
struct MultiSymbolTable {
	uint64_t e[8][256];
	static int slot(unsigned code) { return code <= 18 ? (int)code - 17 : code <= 21 ? (int)code - 18 : code <= 24 ? (int)code - 19 : (int)code - 20; }
	static bool has(unsigned code) { return code == 17 || code == 18 || code == 20 || code == 21 || code == 23 || code == 24 || code == 26 || code == 27; }
	MultiSymbolTable()
	{
		static const unsigned codes[8] = { 17, 18, 20, 21, 23, 24, 26, 27 };
		for (int k = 0; k < 8; k++)
			for (unsigned bits = 0; bits < 256; bits++) {
				unsigned used = 0, rows = 0;
				uint64_t vals = 0x88888888u;
				for (;;) {
					const uint32_t one = SymbolTable::entry(codes[k], (bits >> used) & 127u);
					const unsigned len = one & 7u, cnt = (one >> 3) & 3u;
					if (used + len > 8 || rows + cnt > 8)
						break;
					vals = (vals & ~((uint64_t)0xF << (4 * rows))) | (uint64_t)((one >> 5) & 15u) << (4 * rows);
					if (cnt == 2)           /* "0" of the x3 / x4 / x5 families: two zeros */
						vals = (vals & ~((uint64_t)0xF << (4 * (rows + 1)))) | (uint64_t)((one >> 9) & 15u) << (4 * (rows + 1));
					used += len;
					rows += cnt;
				}
				e[k][bits] = used | rows << 4 | vals << 8;
			}
	}
};

/*
 * One column.  `col` points at idx[0*cols + c]; consecutive rows are `pitch`
 * apart.  Mirrors the per-filler bit grammar of :181-476; returns 1 or <0.
 */
template <class Cursor>
__attribute__((always_inline)) inline int parse_column(Cursor &bc, unsigned code, unsigned rows, int16_t *col, size_t pitch)
{
	unsigned r = 0, b;

	if (code == 0) {                                        /* all zero, no payload */
		for (; r < rows; r++, col += pitch)
			*col = 0;
		return 1;
	}
	if (code >= 3 && code <= 16) {                          /* fixed-width offset binary */
		const int mid = 1 << (code - 1);
		for (; r < rows; r++, col += pitch) {
			TAKE(b, code);
			*col = (int16_t)((int)b - mid);
		}
		return 1;
	}

	if constexpr (Cursor::kTableDriven) {
		if (code < 17 || code > 29 || code == 25 || code == 28)
			return ACM_ERR_CORRUPT;                         /* 1, 2, 25, 28, 30, 31 */
		static const SymbolTable table;
		const uint32_t *const tab = table.e[SymbolTable::slot(code)];
		int16_t sink;                                           /* rows past the column's end land here */
		if (MultiSymbolTable::has(code)) {
			static const MultiSymbolTable multi;
			const uint64_t *const mt = multi.e[MultiSymbolTable::slot(code)];
			while (rows - r >= 8) {
				if (bc.have < 8)
					bc.refill();
				const uint64_t e = mt[(uint32_t)bc.win & 255u];
				const unsigned len = (unsigned)e & 15u, cnt = (unsigned)(e >> 4) & 15u;
				const uint32_t v = (uint32_t)(e >> 8);
				bc.win >>= len;
				bc.have -= len;
				bc.bit += len;
				for (unsigned k = 0; k < 8; k++)
					col[k * pitch] = (int16_t)((int)((v >> (4 * k)) & 15u) - 8);
				col += cnt * pitch;
				r += cnt;
			}
		}
		while (r < rows) {
			if (bc.have < 7)
				bc.refill();
			const uint32_t e = tab[(uint32_t)bc.win & 127u];
			if (e & (1u << 17))
				return ACM_ERR_CORRUPT;
			const unsigned len = e & 7u, cnt = (e >> 3) & 3u;
			bc.win >>= len;
			bc.have -= len;
			bc.bit += len;
			/* all three stores always: a surplus one writes the 0 a later symbol overwrites, or the sink */
			col[0] = (int16_t)((int)((e >> 5) & 15u) - 8);
			*(r + 1 < rows ? col + pitch : &sink) = (int16_t)((int)((e >> 9) & 15u) - 8);
			*(r + 2 < rows ? col + 2 * pitch : &sink) = (int16_t)((int)((e >> 13) & 15u) - 8);
			col += cnt * pitch;
			r += cnt;
		}
		return 1;
	}

	switch (code) {
	case 17: case 20: case 23: case 26: {                   /* "0" = two zeros, "10" = zero, "11.." = value */
		const int tail = (code == 17) ? 1 : (code == 26) ? 3 : 2;
		while (r < rows) {
			TAKE(b, 1);
			if (!b) {
				col[0] = 0;
				if (++r >= rows)
					break;                  /* the second zero would be row `rows` */
				col[pitch] = 0;
				++r;
				col += 2 * pitch;
				continue;
			}
			TAKE(b, 1);
			int v = 0;
			if (b) {
				if (code == 23) {               /* 110s | 111ff */
					TAKE(b, 1);
					if (!b) {
						TAKE(b, 1);
						v = kSign1[b];
					} else {
						TAKE(b, 2);
						v = kFar2[b];
					}
				} else {
					TAKE(b, (unsigned)tail);
					v = (code == 17) ? kSign1[b] : (code == 20) ? kNear2[b] : kWide3[b];
				}
			}
			*col = (int16_t)v;
			col += pitch;
			++r;
		}
		return 1;
	}
	case 18: case 21: case 24: case 27: {                   /* "0" = zero, "1.." = value */
		for (; r < rows; r++, col += pitch) {
			TAKE(b, 1);
			int v = 0;
			if (b) {
				if (code == 24) {               /* 10s | 11ff */
					TAKE(b, 1);
					if (!b) {
						TAKE(b, 1);
						v = kSign1[b];
					} else {
						TAKE(b, 2);
						v = kFar2[b];
					}
				} else if (code == 18) {
					TAKE(b, 1);
					v = kSign1[b];
				} else if (code == 21) {
					TAKE(b, 2);
					v = kNear2[b];
				} else {
					TAKE(b, 3);
					v = kWide3[b];
				}
			}
			*col = (int16_t)v;
		}
		return 1;
	}
	case 19: case 22: {                                     /* three digits base 3 / base 5 per 5 / 7 bits */
		const unsigned base = (code == 19) ? 3 : 5, width = (code == 19) ? 5 : 7;
		const int bias = (int)base / 2;
		while (r < rows) {
			TAKE(b, width);
			if (b >= base * base * base)
				return ACM_ERR_CORRUPT;
			for (int k = 0; k < 3 && r < rows; k++, r++, col += pitch) {
				*col = (int16_t)((int)(b % base) - bias);
				b /= base;
			}
		}
		return 1;
	}
	case 29:                                                /* two digits base 11 per 7 bits */
		while (r < rows) {
			TAKE(b, 7);
			if (b >= 121)
				return ACM_ERR_CORRUPT;
			*col = (int16_t)((int)(b % 11) - 5);
			col += pitch;
			if (++r >= rows)
				break;
			*col = (int16_t)((int)(b / 11) - 5);
			col += pitch;
			++r;
		}
		return 1;
	default:                                                /* 1, 2, 25, 28, 30, 31 */
		return ACM_ERR_CORRUPT;
	}
}

#undef TAKE

} // namespace

int32_t TableHistory::stale_value(int idx) const
{
	unsigned p = 0;
	if (idx >= 0) {
		while (p < 16 && (1 << p) <= idx)       /* smallest p with 2^p > idx */
			p++;
	} else {
		while (p < 16 && (1 << p) < -idx)       /* smallest p with 2^p >= -idx */
			p++;
	}
	const uint32_t v = p < 16 ? val_ge[p] : 0;
	return (int32_t)((uint32_t)idx * v);
}

#define HDR(var, n) do { const int t_ = bc.get(n); if (t_ < 0) { bc.commit(); return t_; } (var) = (unsigned)t_; } while (0)

int read_headers(ACMStream *s)
{
	BitCursor bc(s);
	unsigned v, hi;

	HDR(v, 24);
	if (v == 0x564157) {                                    /* "WAV" (:685) */
		HDR(v, 8);
		if (v != 'C') {
			bc.commit();
			return ACM_ERR_NOT_ACM;
		}
		unsigned w[12];
		for (int i = 0; i < 12; i++)
			HDR(w[i], 16);
		/* only "V1.0" and the 28 in word 6 are checked (:699-706) */
		if (w[0] != 0x3156 || w[1] != 0x302E || w[6] != 28) {
			bc.commit();
			return ACM_ERR_NOT_ACM;
		}
		s->wavc_file = 1;
		HDR(v, 24);
	}
	bc.commit();
	if (v != ACM_ID)
		return ACM_ERR_NOT_ACM;
	s->info.acm_id = v;
	HDR(s->info.acm_version, 8);
	bc.commit();
	if (s->info.acm_version != 1)
		return ACM_ERR_NOT_ACM;
	HDR(s->total_values, 16);
	HDR(hi, 16);
	bc.commit();
	s->total_values += hi << 16;
	if (s->total_values == 0)
		return ACM_ERR_NOT_ACM;
	HDR(s->info.channels, 16);
	bc.commit();
	if (s->info.channels < 1 || s->info.channels > 2)
		return ACM_ERR_NOT_ACM;
	s->info.acm_channels = s->info.channels;
	HDR(s->info.rate, 16);
	bc.commit();
	if (s->info.rate < 4096)
		return ACM_ERR_NOT_ACM;
	HDR(s->info.acm_level, 4);
	HDR(s->info.acm_rows, 12);
	bc.commit();
	if (s->info.acm_rows == 0)
		return ACM_ERR_NOT_ACM;
	return 0;
}

#undef HDR

int parse_block(ACMStream *s, TableHistory *tab, int16_t *const out, acmhip_blkhdr *hdr, PatchSink *sink)
{
	BitCursor bc(s);
	const unsigned rows = s->info.acm_rows;
	const unsigned cols = s->info.acm_cols;
	/* A column is written row by row, `cols` elements apart: from 2048 columns on that is a multiple of 4 KB, every row of a column in the
	 * SAME set of a 32-48 KB level-1 data cache, and with more rows than the cache has ways every store of the block misses (EPYC
	 * 9575F, level 11: 463 Msamples/s per thread with 16 rows, 127 with 64; profiles/host_parse_rate.py).  Such a block is parsed into a
	 * scratch of the thread's whose rows are a cache line further apart, and copied out row by row. */
	const size_t row_bytes = (size_t)cols * sizeof(int16_t);
	const unsigned pitch_padded = cols + 32;
	/* (from 1024 columns on - rows 2 KB apart, every row of a column in one of two cache sets - and with more rows than those sets have
	 * ways: the shapes the measurement covers.  Narrower blocks spread over enough sets and keep the direct write: ADVICE r5) */
	const bool padded = row_bytes >= 2048 && (uint64_t)rows * row_bytes / 4096 > 6 && (uint64_t)rows * pitch_padded * sizeof(int16_t) <= (8u << 20);
	/* the scratch stays with the thread up to 1 MB (level 12 x 64 rows takes 528 KB); what a giant block needed beyond that goes back
	 * to the allocator when the block is done */
	static thread_local std::vector<int16_t> scratch;
	struct ScratchTrim {
		std::vector<int16_t> &v;
		~ScratchTrim()
		{
			if (v.capacity() * sizeof(int16_t) > (1u << 20))
				std::vector<int16_t>().swap(v);
		}
	} trim{ scratch };
	if (padded && scratch.size() < (size_t)rows * pitch_padded)
		scratch.resize((size_t)rows * pitch_padded);
	int16_t *const idx = padded ? scratch.data() : out;
	const unsigned pitch = padded ? pitch_padded : cols;
	const size_t mark = (sink && sink->out) ? sink->out->size() : 0;
	const uint64_t mark_count = sink ? sink->count : 0;
	int rc;

	const int pwr = bc.get_or_end(4);                       /* :588 */
	if (pwr < 0) {
		bc.commit();
		return pwr;
	}
	const int val = bc.get_or_end(16);                      /* :589 */
	if (val < 0) {
		bc.commit();
		return val;
	}
	tab->note_block((unsigned)pwr, (uint32_t)val);          /* what the table build of :592-600 leaves behind */
	hdr->val = (uint32_t)val;
	hdr->pwr = (uint32_t)pwr;
	const int lim = 1 << pwr;                               /* valid indices: [-lim, lim) */

	const uint64_t col_bits = worst_column_bits(rows);
	/* whole rest of the block inside the buffer (the normal case: a block is a few KB, the buffer 64 KB):
	 * stay on the fast cursor for all columns and convert back once */
	FastCursor blk;
	const bool block_fast = fast_enter(bc, blk, (uint64_t)cols * col_bits);
	const unsigned block_phase = s->buf_pos & 3u;
	for (unsigned c = 0; c < cols; c++) {
		int code;
		FastCursor fc;
		if (block_fast) {
			FastCursor cur = blk;                   /* a local copy keeps the cursor in registers */
			code = cur.get(5);
			rc = parse_column(cur, (unsigned)code, rows, idx + c, pitch);
			blk = cur;
			if (rc < 0) {
				fast_leave(bc, blk, block_phase);
				goto fail;
			}
		} else if (fast_enter(bc, fc, col_bits)) {
			const unsigned phase = s->buf_pos & 3u;
			code = fc.get(5);
			rc = parse_column(fc, (unsigned)code, rows, idx + c, pitch);
			fast_leave(bc, fc, phase);
			if (rc < 0)
				goto fail;
		} else {
			code = bc.get_or_end(5);                /* :496 */
			if (code < 0) {
				rc = code;
				goto fail;
			}
			rc = parse_column(bc, (unsigned)code, rows, idx + c, pitch);
			if (rc < 0)
				goto fail;
		}
		/* can this code produce an index outside [-lim, lim)?  linear codes span
		 * [-reach, reach-1], the others are symmetric +-reach */
		const int reach = (int)code_reach((unsigned)code);
		if ((code >= 3 && code <= 16) ? (reach > lim) : (reach >= lim)) {
			/* hazard H1: the column may hold indices the current table does not cover */
			const int16_t *p = idx + c;
			for (unsigned r = 0; r < rows; r++, p += pitch) {
				const int v = *p;
				if (v >= lim || v < -lim) {
					if (sink) {
						if (sink->out)
							sink->out->push_back(acmhip_patch{
								sink->base_sample + (uint64_t)r * cols + c,
								tab->stale_value(v), sink->stream });
						sink->count++;
					}
				}
			}
		}
	}
	if (block_fast)
		fast_leave(bc, blk, block_phase);
	bc.commit();
	if (padded)
		for (unsigned r = 0; r < rows; r++)
			memcpy(out + (size_t)r * cols, idx + (size_t)r * pitch, row_bytes);
	return 1;

fail:
	bc.commit();
	if (sink) {
		if (sink->out)
			sink->out->resize(mark);
		sink->count = mark_count;
	}
	return rc;
}

int skip_bits(ACMStream *s, unsigned n)
{
	if (n == 0)
		return 0;
	BitCursor bc(s);
	const int v = bc.get(n);
	bc.commit();
	return v < 0 ? v : 0;
}

void reset_reader(ACMStream *s)
{
	s->file_eof = 0;
	s->buf_pos = 0;
	s->buf_size = 0;
	s->bit_avail = 0;
	s->bit_data = 0;
	s->buf_start_ofs = 14;                                  /* util.c:239 uses the plain header length even for WAVC */
}

} // namespace acmfill
