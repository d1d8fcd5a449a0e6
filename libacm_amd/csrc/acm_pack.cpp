/*
 * acm_pack.cpp - the host stager's packed half (include/acm_hip.h, "Packed staged form").
 *
 * The reference stores every filler index through set_pos() as a table look-up into an int32 block matrix
 * (/root/reference/src/decode.c:174-177); what range an index can have is fixed by its column's filler (zero filler: 0;
 * k / t fillers: |idx| <= 5; linear filler of `ind` bits: [-2^(ind-1), 2^(ind-1)), decode.c:181-476).  Here the staged
 * indices of a tile leave the host as "width class per column pair and row group + the indices at that width", sorted
 * by class so that the device unpacks a whole wavefront's worth with one code path (acm_kernels.hip: acm_tile2p).
 * The class is taken from the values themselves (the narrowest of 0 / 4 / 8 / 16 bits that holds them), which is never
 * wider than what the filler codes promise and needs no knowledge of them.
 *
 * Pure host code: no HIP call in this file.
 */
#include <string.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <algorithm>
#include <vector>

#include "acm_device.h"
#include "acm_mform.h"
#include "acm_hip.h"

namespace {

struct Geo {
	uint32_t level, cols, pairs, tile_rows, group_rows, nq, rpc, permb, ngroups, slots, waves, per_wave, pad_shift;
};

bool geo_of(uint32_t level, Geo *g)
{
	const int tr = acmk_tile2p_rows(level), gr = acmk_tile2p_group_rows(level);
	if (tr <= 0 || gr <= 0)
		return false;
	g->level = level;
	g->cols = 1u << level;
	g->pairs = g->cols / 2;
	g->tile_rows = (uint32_t)tr;
	g->group_rows = (uint32_t)gr;
	g->nq = g->group_rows / 4;
	g->rpc = 64 / g->nq;
	g->permb = g->rpc * 2;
	g->ngroups = g->tile_rows / g->group_rows;
	g->slots = (uint32_t)acmk_tile2p_slots(level);
	g->waves = (uint32_t)acmk_tile2p_waves(level);
	g->per_wave = g->slots / g->waves;
	g->pad_shift = (uint32_t)acmk_tile2p_pad_shift(level);
	return true;
}

inline uint32_t unit_bytes(uint32_t kind) { return kind >= ACMHIP_PK_NIBBLE ? 1u << kind : 0u; }     /* 4 / 8 / 16 */

inline uint64_t round16(uint64_t v) { return (v + 15) & ~15ull; }

/* bytes past a chunk's last unit that a wavefront may still read: all 64 lanes load their unit's dwords, count or not */
constexpr uint64_t kReadSlack = 64 * 16 + 128;

/* where the first column of pair p sits in the kernel's padded LDS row, in dwords (acm_kernels.hip: lds_at) */
inline uint32_t lds_place(const Geo &g, uint32_t p) { return 2 * p + ((2 * p) >> g.pad_shift); }

} // namespace

extern "C" int acmhip_packed_tile_rows(uint32_t level)
{
	return acmk_tile2p_rows(level);
}

extern "C" int acmhip_packed_group_rows(uint32_t level)
{
	return acmk_tile2p_group_rows(level);
}

extern "C" int acmhip_packed_slots(uint32_t level)
{
	return acmk_tile2p_slots(level);
}

extern "C" int acmhip_pack_bound(uint32_t level, uint64_t ntiles, uint64_t *max_blob_bytes)
{
	Geo g;
	if (!geo_of(level, &g))
		return ACMHIP_ERR_ARG;
	if (max_blob_bytes)     /* every index as a word + every chunk's column-pair list */
		*max_blob_bytes = ntiles * ((uint64_t)g.tile_rows * g.cols * 2 + (uint64_t)g.slots * (g.permb + 16)) + kReadSlack;
	return ACMHIP_OK;
}

extern "C" int acmhip_pack_tiles(uint32_t level, const int16_t *idx, uint64_t ntiles, acmhip_packed_chunk *chunks, uint8_t *blob,
				 uint64_t blob_base, uint64_t *blob_bytes_out)
{
	Geo g;
	if (!geo_of(level, &g) || (ntiles && (!idx || !chunks || !blob)) || (blob_base & 15))
		return ACMHIP_ERR_ARG;
	std::vector<uint16_t> need(g.pairs);            /* per column pair: OR of the magnitudes' bits in this group */
	std::vector<uint16_t> order[5];                 /* [kind]: the column pairs of that class, ascending */
	for (auto &o : order)
		o.reserve(g.pairs);
	uint64_t nb = 0;
	for (uint64_t t = 0; t < ntiles; t++) {
		acmhip_packed_chunk *tc = chunks + t * g.slots;
		memset(tc, 0, g.slots * sizeof(acmhip_packed_chunk));
		uint32_t dealt = 0;             /* chunks of this tile so far: chunk k goes to wave k % waves, its slot k / waves */
		for (uint32_t grp = 0; grp < g.ngroups; grp++) {
			const int16_t *rows = idx + ((t * g.tile_rows + (uint64_t)grp * g.group_rows) << level);
			memset(need.data(), 0, g.pairs * sizeof(uint16_t));
			for (uint32_t r = 0; r < g.group_rows; r++) {
				const int16_t *row = rows + ((uint64_t)r << level);
				for (uint32_t p = 0; p < g.pairs; p++) {
					const int16_t a = row[2 * p], b = row[2 * p + 1];
					/* v and ~v have the same width in two's complement; bit 15 of the OR marks "not zero" for a lone -1 */
					need[p] |= (uint16_t)((a ^ (a >> 15)) | (b ^ (b >> 15)) | ((a | b) ? 0x8000 : 0));
				}
			}
			for (auto &o : order)
				o.clear();
			for (uint32_t p = 0; p < g.pairs; p++) {
				const uint16_t m = need[p] & 0x7FFF;
				const uint32_t kind = !need[p] ? ACMHIP_PK_ZERO : m < 8 ? ACMHIP_PK_NIBBLE : m < 128 ? ACMHIP_PK_BYTE : ACMHIP_PK_WORD;
				order[kind].push_back((uint16_t)p);
			}
			for (uint32_t kind = ACMHIP_PK_WORD; kind >= ACMHIP_PK_ZERO; kind--) {
				const std::vector<uint16_t> &o = order[kind];
				const uint32_t ub = unit_bytes(kind);
				for (size_t at = 0; at < o.size(); at += g.rpc) {
					const uint32_t count = (uint32_t)std::min<size_t>(g.rpc, o.size() - at);
					if ((blob_base + nb) / 16 > 0xFFFFFFFFull || dealt >= g.slots)
						return ACMHIP_ERR_ARG;          /* more than 64 GB of blobs in one arena (the slots cannot run out: acm_kernels.hip MAXCHUNKS) */
					acmhip_packed_chunk &c = tc[(dealt % g.waves) * g.per_wave + dealt / g.waves];
					dealt++;
					c.blob_off16 = (uint32_t)((blob_base + nb) / 16);
					c.count = (uint16_t)count;
					c.kind = (uint8_t)kind;
					c.row0 = (uint8_t)(grp * g.group_rows);
					uint8_t *out = blob + nb;
					uint16_t *perm = reinterpret_cast<uint16_t *>(out);
					for (uint32_t k = 0; k < g.rpc; k++)
						perm[k] = k < count ? (uint16_t)lds_place(g, o[at + k]) : 0;
					uint8_t *units = out + g.permb;
					for (uint32_t k = 0; k < count && ub; k++) {
						const uint32_t p = o[at + k];
						for (uint32_t q = 0; q < g.nq; q++) {
							uint32_t *u = reinterpret_cast<uint32_t *>(units + ((size_t)k * g.nq + q) * ub);
							/* the unit's rows: q, q + nq, q + 2 nq, q + 3 nq of the group */
							const int16_t *r0 = rows + ((uint64_t)q << level) + 2 * p;
							int16_t v[8];
							for (uint32_t i = 0; i < 4; i++) {
								v[2 * i] = r0[(uint64_t)(i * g.nq) << level];
								v[2 * i + 1] = r0[((uint64_t)(i * g.nq) << level) + 1];
							}
							if (kind == ACMHIP_PK_WORD) {
								for (uint32_t i = 0; i < 4; i++)
									u[i] = (uint32_t)(uint16_t)v[2 * i] | (uint32_t)(uint16_t)v[2 * i + 1] << 16;
							} else if (kind == ACMHIP_PK_BYTE) {
								for (uint32_t i = 0; i < 2; i++)
									u[i] = (uint32_t)(uint8_t)v[4 * i] | (uint32_t)(uint8_t)v[4 * i + 1] << 8 |
									       (uint32_t)(uint8_t)v[4 * i + 2] << 16 | (uint32_t)(uint8_t)v[4 * i + 3] << 24;
							} else {
								uint32_t w = 0;
								for (uint32_t i = 0; i < 8; i++)
									w |= ((uint32_t)v[i] & 15u) << (4 * i);
								u[0] = w;
							}
						}
					}
					const uint64_t used = g.permb + (uint64_t)count * g.nq * ub;
					const uint64_t padded = round16(used);
					memset(out + used, 0, padded - used);
					nb += padded;
				}
			}
		}
	}
	memset(blob + nb, 0, kReadSlack);
	nb += kReadSlack;
	if (blob_bytes_out)
		*blob_bytes_out = nb;
	return ACMHIP_OK;
}

extern "C" int acmhip_unpack_tile(uint32_t level, const acmhip_packed_chunk *tile_chunks, const uint8_t *blob, int16_t *idx)
{
	Geo g;
	if (!geo_of(level, &g) || !tile_chunks || !blob || !idx)
		return ACMHIP_ERR_ARG;
	/* every (row, column) must be written exactly once */
	const size_t n = (size_t)g.tile_rows << level;
	std::vector<uint8_t> seen(n, 0);
	std::vector<int32_t> pair_of(lds_place(g, g.pairs - 1) + 1, -1);
	for (uint32_t p = 0; p < g.pairs; p++)
		pair_of[lds_place(g, p)] = (int32_t)p;
	for (uint32_t c = 0; c < g.slots; c++) {
		const acmhip_packed_chunk &ch = tile_chunks[c];
		if (ch.kind == 0)
			continue;
		if (ch.kind > ACMHIP_PK_WORD || ch.count == 0 || ch.count > g.rpc || ch.row0 % g.group_rows || ch.row0 + g.group_rows > g.tile_rows)
			return ACMHIP_ERR_ARG;
		const uint8_t *in = blob + (uint64_t)ch.blob_off16 * 16;
		const uint16_t *perm = reinterpret_cast<const uint16_t *>(in);
		const uint32_t ub = unit_bytes(ch.kind);
		for (uint32_t k = 0; k < ch.count; k++) {
			if (perm[k] >= pair_of.size() || pair_of[perm[k]] < 0)
				return ACMHIP_ERR_ARG;
			const uint32_t p = (uint32_t)pair_of[perm[k]];
			for (uint32_t q = 0; q < g.nq; q++) {
				const uint32_t *u = reinterpret_cast<const uint32_t *>(in + g.permb + ((size_t)k * g.nq + q) * ub);
				int16_t v[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
				if (ch.kind == ACMHIP_PK_WORD) {
					for (uint32_t i = 0; i < 4; i++) {
						v[2 * i] = (int16_t)(u[i] & 0xFFFF);
						v[2 * i + 1] = (int16_t)(u[i] >> 16);
					}
				} else if (ch.kind == ACMHIP_PK_BYTE) {
					for (uint32_t i = 0; i < 8; i++)
						v[i] = (int8_t)(u[i / 4] >> (8 * (i % 4)));
				} else if (ch.kind == ACMHIP_PK_NIBBLE) {
					for (uint32_t i = 0; i < 8; i++)
						v[i] = (int16_t)((int32_t)(u[0] << (28 - 4 * i)) >> 28);
				}
				for (uint32_t i = 0; i < 8; i++) {
					const size_t at = ((size_t)(ch.row0 + q + (i / 2) * g.nq) << level) + 2 * p + (i & 1);
					if (seen[at])
						return ACMHIP_ERR_ARG;
					seen[at] = 1;
					idx[at] = v[i];
				}
			}
		}
	}
	for (size_t k = 0; k < n; k++)
		if (!seen[k])
			return ACMHIP_ERR_ARG;
	return ACMHIP_OK;
}

/* ---------------------------------------------------------------------------
 * byte-plane staged form (include/acm_hip.h): the staged indices in the order acm_tile2's matrix-core build reads them, every row pair at
 * the narrowest of 4 / 8 / 16 bits per index that holds its indices (a block's indices lie in [-2^pwr, 2^pwr): decode.c:592-600)
 * --------------------------------------------------------------------------- */
extern "C" int acmhip_mform_tile_rows(uint32_t level)
{
	return acmk_tile2m_rows(level);
}

extern "C" int acmhip_mform_group(uint32_t level)
{
	const int g = acmk_tile2m_stages(level);
	return g ? 1 << g : 0;
}

/* the kernel reads 16 bytes (32 where a residue's columns are 16) per lane whatever the class: room behind the last pair */
static const uint64_t kMformSlack = 64;

extern "C" uint64_t acmhip_mform_bytes(uint32_t level, uint64_t nrows)
{
	const uint64_t cols = 1ull << level;
	return nrows * cols * 2 + 2 * cols + kMformSlack;       /* every pair at 16 bits, the pair of zeros in front at 4 bits or at 8 */
}

extern "C" uint64_t acmhip_mform_pairs(uint64_t nrows)
{
	return nrows / 2 + 1;
}

namespace {
/* the chunk kernel's form (64 columns of a residue class side by side): classes 8 and 16 bits - the 16-bit one as two SIGNED bytes, or
 * with an unsigned low byte for the pairs those cannot hold - and, at the levels of acm_chunk itself (8-12; levels 13 / 14 read the same
 * form inside acm_tile2, which knows no 12-bit class), 12 bits */
inline bool split_form(size_t qn) { return qn == 64; }
inline bool nib12_level(uint32_t level) { return acmhip_mform_group(level) == 64 && level <= 12; }
inline uint32_t pair_bytes(uint32_t level, uint32_t cls)
{
	if (cls == ACMHIP_BP_NIB12 && split_form((size_t)acmhip_mform_group(level)))
		return 3u << level;                     /* two rows of 1.5 bytes per index */
	if (cls == ACMHIP_BP_WORDU)
		return 4u << level;                     /* (16 bits, the low byte unsigned) */
	return (4u << level) >> (3 - cls);             /* two rows of 2 / 1 / 0.5 bytes per index */
}
#if defined(__SSE2__)
/* eight residues at once: the rows q of an 8 x 8 block of indices (row q = columns c0 .. c0 + 7 of the q-th eighth) transposed, so that
 * t[c] holds the eight indices of residue c0 + c.  The scalar loops below are what this computes; it is here because the re-order is a
 * second pass of the host pool over everything it has parsed (245 -> ~900 Msamples/s per thread at 16 bits). */
inline void transpose8x8(__m128i (&v)[8])
{
	const __m128i a0 = _mm_unpacklo_epi16(v[0], v[1]), a1 = _mm_unpackhi_epi16(v[0], v[1]);
	const __m128i a2 = _mm_unpacklo_epi16(v[2], v[3]), a3 = _mm_unpackhi_epi16(v[2], v[3]);
	const __m128i a4 = _mm_unpacklo_epi16(v[4], v[5]), a5 = _mm_unpackhi_epi16(v[4], v[5]);
	const __m128i a6 = _mm_unpacklo_epi16(v[6], v[7]), a7 = _mm_unpackhi_epi16(v[6], v[7]);
	const __m128i b0 = _mm_unpacklo_epi32(a0, a2), b1 = _mm_unpackhi_epi32(a0, a2);
	const __m128i b2 = _mm_unpacklo_epi32(a1, a3), b3 = _mm_unpackhi_epi32(a1, a3);
	const __m128i b4 = _mm_unpacklo_epi32(a4, a6), b5 = _mm_unpackhi_epi32(a4, a6);
	const __m128i b6 = _mm_unpacklo_epi32(a5, a7), b7 = _mm_unpackhi_epi32(a5, a7);
	v[0] = _mm_unpacklo_epi64(b0, b4); v[1] = _mm_unpackhi_epi64(b0, b4);
	v[2] = _mm_unpacklo_epi64(b1, b5); v[3] = _mm_unpackhi_epi64(b1, b5);
	v[4] = _mm_unpacklo_epi64(b2, b6); v[5] = _mm_unpackhi_epi64(b2, b6);
	v[6] = _mm_unpacklo_epi64(b3, b7); v[7] = _mm_unpackhi_epi64(b3, b7);
}
/* classes 2 and 3 of put_row for qn = 8, 16 or 64 (sigma is a multiple of 8).  split: the 16-bit class stores idx = 256 hi + lo with BOTH bytes
 * signed (the chunk kernel's form, qn = 64); else the low byte minus 128 and the arithmetic high byte */
bool put_row_sse(const int16_t *src, size_t sigma, size_t qn, uint32_t cls, bool split, uint8_t *dst)
{
	const bool nib12 = split && qn == 64 && cls == ACMHIP_BP_NIB12;
	if ((qn != 8 && qn != 16 && qn != 64) || (sigma & 7) || (cls != ACMHIP_BP_WORD && cls != ACMHIP_BP_BYTE && !nib12))
		return false;
	const __m128i flip = _mm_set1_epi16(0x0080), low = _mm_set1_epi16(0x00ff);
	for (size_t c0 = 0; c0 < sigma; c0 += 8) {
		__m128i t[8][8];
		for (size_t h = 0; h < qn / 8; h++) {
			for (size_t q = 0; q < 8; q++)
				t[h][q] = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + c0 + (8 * h + q) * sigma));
			transpose8x8(t[h]);
		}
		for (size_t c = 0; c < 8; c++) {
			uint8_t *d = dst + (c0 + c) * (cls == ACMHIP_BP_WORD ? 2 * qn : nib12 ? 96 : qn);
			for (size_t h = 0; h < qn / 8; h += 2) {
				const __m128i xa = t[h][c], xb = h + 1 < qn / 8 ? t[h + 1][c] : _mm_setzero_si128();
				__m128i lo, hi;
				if (nib12) {
					/* (put_row_nib12 is what this computes) the sixteen elements q = 8 h .. 8 h + 15 are one lane's: 16 low bytes, 8 bytes of nibbles */
					const __m128i la = _mm_srai_epi16(_mm_slli_epi16(xa, 8), 8), lb = _mm_srai_epi16(_mm_slli_epi16(xb, 8), 8);
					lo = _mm_packs_epi16(la, lb);
					hi = _mm_packs_epi16(_mm_srai_epi16(_mm_sub_epi16(xa, la), 8), _mm_srai_epi16(_mm_sub_epi16(xb, lb), 8));       /* [-8, 7] each */
					const __m128i up = _mm_and_si128(_mm_slli_epi32(hi, 4), _mm_set1_epi8((char)0xF0));      /* elements 0-3 | 4-7 | 8-11 | 12-15 in the high nibbles */
					const __m128i dn = _mm_and_si128(hi, _mm_set1_epi8(0x0F));
					const __m128i r = _mm_or_si128(up, _mm_srli_epi64(dn, 32));             /* dword 0: elements 0-3 over 4-7, dword 2: 8-11 over 12-15 */
					_mm_storeu_si128(reinterpret_cast<__m128i *>(d + 8 * h), lo);
					_mm_storel_epi64(reinterpret_cast<__m128i *>(d + 64 + 4 * h), _mm_shuffle_epi32(r, _MM_SHUFFLE(3, 1, 2, 0)));
					continue;
				}
				if (cls == ACMHIP_BP_BYTE) {
					lo = _mm_packs_epi16(xa, xb);                   /* every index in [-128, 127]: no saturation */
					hi = lo;
				} else if (split) {
					const __m128i la = _mm_srai_epi16(_mm_slli_epi16(xa, 8), 8), lb = _mm_srai_epi16(_mm_slli_epi16(xb, 8), 8);       /* the low byte, signed */
					lo = _mm_packs_epi16(la, lb);
					hi = _mm_packs_epi16(_mm_srai_epi16(_mm_sub_epi16(xa, la), 8), _mm_srai_epi16(_mm_sub_epi16(xb, lb), 8));     /* (idx < 32640: no overflow, no saturation) */
				} else {
					lo = _mm_packus_epi16(_mm_and_si128(_mm_xor_si128(xa, flip), low), _mm_and_si128(_mm_xor_si128(xb, flip), low));
					hi = _mm_packs_epi16(_mm_srai_epi16(xa, 8), _mm_srai_epi16(xb, 8));
				}
				if (qn == 8) {
					if (cls == ACMHIP_BP_BYTE)
						_mm_storel_epi64(reinterpret_cast<__m128i *>(d), lo);
					else
						_mm_storeu_si128(reinterpret_cast<__m128i *>(d), _mm_unpacklo_epi64(lo, hi));
				} else {
					_mm_storeu_si128(reinterpret_cast<__m128i *>(d + 8 * h), lo);
					if (cls == ACMHIP_BP_WORD)
						_mm_storeu_si128(reinterpret_cast<__m128i *>(d + qn + 8 * h), hi);
				}
			}
		}
	}
	return true;
}
#endif

/* the 12-bit class of the chunk kernel's form: idx = 256 hi + lo, lo a signed byte, hi a signed NIBBLE (idx in [-2176, 1919]).  Per residue
 * c: 64 low bytes, then 32 bytes of high nibbles in the order the kernel's lanes take them - the lane that feeds columns q = 16 ks ..
 * 16 ks + 15 of the class to the matrix instruction reads 8 bytes at 8 ks: dword d < 2 holds its elements 8 d .. 8 d + 7, element 8 d + b
 * (b < 4) in the HIGH nibble of byte b and element 8 d + 4 + b in the low one, so that x & 0xf0f0f0f0 and (x << 4) & 0xf0f0f0f0 are the
 * operand bytes (hi << 4: signed bytes as they stand) of elements 8 d .. + 3 and 8 d + 4 .. + 7 */
void put_row_nib12(const int16_t *src, size_t sigma, uint8_t *dst)
{
	for (size_t c = 0; c < sigma; c++) {
		uint8_t *d = dst + c * 96;
		int hi[64];
		for (size_t q = 0; q < 64; q++) {
			const int x = src[c + q * sigma];
			const int lo = (int8_t)(uint8_t)x;
			d[q] = (uint8_t)lo;
			hi[q] = (x - lo) >> 8;                  /* [-8, 7] */
		}
		for (size_t ks = 0; ks < 4; ks++)
			for (size_t dw = 0; dw < 2; dw++)
				for (size_t b = 0; b < 4; b++) {
					const size_t q0 = 16 * ks + 8 * dw;
					d[64 + 8 * ks + 4 * dw + b] = (uint8_t)(((hi[q0 + b] & 15) << 4) | (hi[q0 + 4 + b] & 15));
				}
	}
}
void get_row_nib12(const uint8_t *src, size_t sigma, int16_t *dst)
{
	for (size_t c = 0; c < sigma; c++) {
		const uint8_t *d = src + c * 96;
		for (size_t ks = 0; ks < 4; ks++)
			for (size_t dw = 0; dw < 2; dw++)
				for (size_t b = 0; b < 4; b++) {
					const size_t q0 = 16 * ks + 8 * dw;
					const uint8_t n = d[64 + 8 * ks + 4 * dw + b];
					const int h0 = (int)(int8_t)(n & 0xF0) >> 4, h1 = (int)(int8_t)(uint8_t)(n << 4) >> 4;
					dst[c + (q0 + b) * sigma] = (int16_t)(256 * h0 + (int)(int8_t)d[q0 + b]);
					dst[c + (q0 + 4 + b) * sigma] = (int16_t)(256 * h1 + (int)(int8_t)d[q0 + 4 + b]);
				}
	}
}

/* one row at width class cls: per residue c < sigma the qn indices of columns c + q * sigma */
void put_row(const int16_t *src, size_t sigma, size_t qn, uint32_t cls, bool split, uint8_t *dst)
{
	if (split && cls == ACMHIP_BP_WORDU) {
		/* the whole int16 range: the low byte unsigned, stored minus 128 - byte for byte what the 8 / 16-column forms write as their
		 * 16-bit class */
		cls = ACMHIP_BP_WORD;
		split = false;
	}
#if defined(__SSE2__)
	if (put_row_sse(src, sigma, qn, cls, split, dst))
		return;
#endif
	if (split && cls == ACMHIP_BP_NIB12) {
		put_row_nib12(src, sigma, dst);
		return;
	}
	for (size_t c = 0; c < sigma; c++) {
		if (cls == ACMHIP_BP_WORD) {
			uint8_t *d = dst + c * 2 * qn;
			for (size_t q = 0; q < qn; q++) {
				const int x = src[c + q * sigma];
				if (split) {
					const int lo = (int8_t)(uint8_t)x;      /* idx = 256 hi + lo, both signed bytes */
					d[q] = (uint8_t)lo;
					d[qn + q] = (uint8_t)((x - lo) >> 8);
				} else {
					d[q] = (uint8_t)((uint16_t)x ^ 0x80u);  /* low byte minus 128: a signed byte for the matrix instruction */
					d[qn + q] = (uint8_t)((uint16_t)x >> 8);
				}
			}
		} else if (cls == ACMHIP_BP_BYTE) {
			uint8_t *d = dst + c * qn;
			for (size_t q = 0; q < qn; q++)
				d[q] = (uint8_t)(int8_t)src[c + q * sigma];
		} else {
			/* eight indices per dword, plus 8 each: index q = 8 j + i in nibble 2 i of dword j, index 8 j + 4 + i in nibble 2 i + 1
			 * (x & 0x0f0f0f0f and (x >> 4) & 0x0f0f0f0f are then the bytes of columns 8 j .. 8 j + 3 and 8 j + 4 .. 8 j + 7) */
			uint8_t *d = dst + c * (qn / 2);
			for (size_t j = 0; j < qn / 8; j++) {
				uint32_t w = 0;
				for (size_t i = 0; i < 4; i++) {
					w |= (uint32_t)((src[c + (8 * j + i) * sigma] + 8) & 15) << (8 * i);
					w |= (uint32_t)((src[c + (8 * j + 4 + i) * sigma] + 8) & 15) << (8 * i + 4);
				}
				memcpy(d + 4 * j, &w, 4);
			}
		}
	}
}
void get_row(const uint8_t *src, size_t sigma, size_t qn, uint32_t cls, bool split, int16_t *dst)
{
	if (split && cls == ACMHIP_BP_WORDU) {
		cls = ACMHIP_BP_WORD;
		split = false;
	}
	if (split && cls == ACMHIP_BP_NIB12) {
		get_row_nib12(src, sigma, dst);
		return;
	}
	for (size_t c = 0; c < sigma; c++) {
		if (cls == ACMHIP_BP_WORD) {
			const uint8_t *d = src + c * 2 * qn;
			for (size_t q = 0; q < qn; q++)
				dst[c + q * sigma] = split ? (int16_t)(256 * (int)(int8_t)d[qn + q] + (int)(int8_t)d[q])
							   : (int16_t)(uint16_t)((d[q] ^ 0x80u) | ((unsigned)d[qn + q] << 8));
		} else if (cls == ACMHIP_BP_BYTE) {
			for (size_t q = 0; q < qn; q++)
				dst[c + q * sigma] = (int16_t)(int8_t)src[c * qn + q];
		} else {
			for (size_t j = 0; j < qn / 8; j++) {
				uint32_t w;
				memcpy(&w, src + c * (qn / 2) + 4 * j, 4);
				for (size_t i = 0; i < 4; i++) {
					dst[c + (8 * j + i) * sigma] = (int16_t)((int)((w >> (8 * i)) & 15) - 8);
					dst[c + (8 * j + 4 + i) * sigma] = (int16_t)((int)((w >> (8 * i + 4)) & 15) - 8);
				}
			}
		}
	}
}
} // namespace

int acm_mform_begin(AcmMformWriter *w, uint32_t level, uint8_t *out, uint64_t blob_base, acmhip_mform_pair *pairs)
{
	const size_t qn = (size_t)acmhip_mform_group(level);
	if (!w || !qn || !out || !pairs || (blob_base & 63))
		return ACMHIP_ERR_ARG;
	w->level = level;
	w->qn = qn;
	w->cols = (size_t)1 << level;
	w->sigma = w->cols / qn;
	w->split = split_form(qn);
	w->nib12 = nib12_level(level);
	w->out = out;
	w->blob_base = blob_base;
	w->pairs = pairs;
	w->at = 0;
	/* the pair in front of the stream: index 0 everywhere, at 4 bits (8 in the chunk kernel's form, which has no narrower class) */
	if ((blob_base >> 6) >= (1ull << 30))
		return ACMHIP_ERR_ARG;                  /* more than 64 GB in front of this block: the pair table counts 64-byte units in 30 bits */
	const uint32_t cls_front = w->split ? ACMHIP_BP_BYTE : ACMHIP_BP_NIBBLE;
	pairs[0] = (acmhip_mform_pair)((blob_base >> 6) << 2 | cls_front);
	memset(out, w->split ? 0 : 0x88, pair_bytes(level, cls_front));
	w->at = pair_bytes(level, cls_front);
	w->npairs = 1;
	return ACMHIP_OK;
}

int acm_mform_put_pair(AcmMformWriter *w, const int16_t *src)
{
	int lo = 0, hi = 0;
#if defined(__SSE2__)
	{
		__m128i vlo = _mm_setzero_si128(), vhi = _mm_setzero_si128();
		for (size_t m = 0; m < 2 * w->cols; m += 8) {
			const __m128i x = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + m));
			vlo = _mm_min_epi16(vlo, x);
			vhi = _mm_max_epi16(vhi, x);
		}
		int16_t a[8], b[8];
		_mm_storeu_si128(reinterpret_cast<__m128i *>(a), vlo);
		_mm_storeu_si128(reinterpret_cast<__m128i *>(b), vhi);
		for (int k = 0; k < 8; k++) {
			lo = a[k] < lo ? a[k] : lo;
			hi = b[k] > hi ? b[k] : hi;
		}
	}
#else
	for (size_t m = 0; m < 2 * w->cols; m++) {
		lo = src[m] < lo ? src[m] : lo;
		hi = src[m] > hi ? src[m] : hi;
	}
#endif
	/* 256 hi + lo with two signed bytes ends at 32639: a pair beyond that is written with the unsigned low byte (class 0: the kernels -
	 * the chunk kernel's general path, FirstPassZW at levels 13 / 14 - put the 128 back) */
	const uint32_t cls = (!w->split && lo >= -8 && hi <= 7) ? ACMHIP_BP_NIBBLE : (lo >= -128 && hi <= 127) ? ACMHIP_BP_BYTE :
			     (w->nib12 && lo >= -2176 && hi <= 1919) ? ACMHIP_BP_NIB12 : (w->split && hi >= 32640) ? ACMHIP_BP_WORDU : ACMHIP_BP_WORD;
	if (((w->blob_base + w->at) >> 6) >= (1ull << 30))
		return ACMHIP_ERR_ARG;
	w->pairs[w->npairs++] = (acmhip_mform_pair)(((w->blob_base + w->at) >> 6) << 2 | cls);
	const size_t rowb = pair_bytes(w->level, cls) / 2;
	put_row(src, w->sigma, w->qn, cls, w->split, w->out + w->at);
	put_row(src + w->cols, w->sigma, w->qn, cls, w->split, w->out + w->at + rowb);
	w->at += 2 * rowb;
	return ACMHIP_OK;
}

uint64_t acm_mform_end(AcmMformWriter *w)
{
	memset(w->out + w->at, 0, kMformSlack);
	return w->at + kMformSlack;
}

int acm_mform_get_pair(uint32_t level, const uint8_t *blob, acmhip_mform_pair entry, int16_t *two_rows)
{
	const size_t qn = (size_t)acmhip_mform_group(level);
	if (!qn || !blob || !two_rows)
		return ACMHIP_ERR_ARG;
	const size_t cols = (size_t)1 << level, sigma = cols / qn;
	const uint32_t cls = entry & 3;
	if (cls > ACMHIP_BP_WORD || (split_form(qn) ? (cls == ACMHIP_BP_NIB12 && !nib12_level(level)) : cls < ACMHIP_BP_NIBBLE))
		return ACMHIP_ERR_ARG;
	const uint8_t *src = blob + ((uint64_t)(entry >> 2) << 6);
	const size_t rowb = pair_bytes(level, cls) / 2;
	get_row(src, sigma, qn, cls, split_form(qn), two_rows);
	get_row(src + rowb, sigma, qn, cls, split_form(qn), two_rows + cols);
	return ACMHIP_OK;
}

extern "C" int acmhip_mform_rows(uint32_t level, const int16_t *idx, uint64_t nrows, uint8_t *out, uint64_t blob_base, acmhip_mform_pair *pairs,
				 uint64_t *bytes_used)
{
	if ((!idx && nrows) || (nrows & 1))
		return ACMHIP_ERR_ARG;
	AcmMformWriter w;
	int rc = acm_mform_begin(&w, level, out, blob_base, pairs);
	for (uint64_t p = 0; rc == ACMHIP_OK && p < nrows / 2; p++)
		rc = acm_mform_put_pair(&w, idx + 2 * p * w.cols);
	if (rc != ACMHIP_OK)
		return rc;
	const uint64_t used = acm_mform_end(&w);
	if (bytes_used)
		*bytes_used = used;
	return ACMHIP_OK;
}

extern "C" int acmhip_mform_unrows(uint32_t level, const uint8_t *blob, const acmhip_mform_pair *pairs, uint64_t nrows, int16_t *idx)
{
	const size_t qn = (size_t)acmhip_mform_group(level);
	if (!qn || !blob || !pairs || (!idx && nrows) || (nrows & 1))
		return ACMHIP_ERR_ARG;
	const size_t cols = (size_t)1 << level;
	std::vector<int16_t> front(2 * cols);
	for (uint64_t p = 0; p <= nrows / 2; p++) {
		int16_t *dst = p ? idx + 2 * (p - 1) * cols : front.data();
		const int rc = acm_mform_get_pair(level, blob, pairs[p], dst);
		if (rc != ACMHIP_OK)
			return rc;
		if (!p)
			for (size_t m = 0; m < 2 * cols; m++)
				if (front[m] != 0)
					return ACMHIP_ERR_ARG;          /* the pair in front of a stream is zeros */
	}
	return ACMHIP_OK;
}
