/*
 * acm_fill.h - host half of the decode path: the sequential bitstream reader
 * and the filler parsers, producing the staged form the device consumes
 * (include/acm_hip.h).  Internal to libacm_hip.so.
 *
 * Replaces /root/reference/src/decode.c:41-163 (bit reader), :181-502 (fillers
 * and fill_block), :586-589 (block header), :687-752 (file headers).  The
 * reader works directly on the public ACMStream fields (buf, buf_size,
 * buf_pos, bit_data, bit_avail, buf_start_ofs, file_eof) so that their meaning
 * - and acm_raw_tell() - stay what callers of the reference expect.
 */
#ifndef ACM_FILL_H
#define ACM_FILL_H

#include <stdint.h>

#include <vector>

#include "acm_hip.h"
#include "libacm.h"

namespace acmfill {

constexpr int kCleanEof = -99;          /* end of data at a block/column boundary (decode.c:31) */
constexpr unsigned kChunkBytes = 64 * 1024;   /* refill granularity asked of read_func (decode.c:29) */

/*
 * What is left of the reference's never-cleared amplitude table
 * (decode.c:809-810): entry i (|i| < 2^p or i == -2^p ...) was last written by
 * the most recent block whose pwr covered it, so all the table's history
 * reduces to "val of the latest block with pwr >= p" for p = 0..15.
 * Entries nobody has written read as 0 here (heap garbage in the reference).
 */
struct TableHistory {
	uint32_t val_ge[16];
	void reset() { for (auto &v : val_ge) v = 0; }
	void note_block(unsigned pwr, uint32_t val) { for (unsigned p = 0; p <= pwr && p < 16; p++) val_ge[p] = val; }
	/* value the reference would fetch for an index outside the current block's range */
	int32_t stale_value(int idx) const;
};

struct PatchSink {
	std::vector<acmhip_patch> *out;     /* may be null: count only */
	uint64_t base_sample;               /* staged-sample index of this block's sample 0 */
	uint32_t stream;
	uint64_t count;
};

/* 14-byte header (+WAVC), decode.c:712-752.  0 or ACM_ERR_NOT_ACM / a read error. */
int read_headers(ACMStream *s);

/*
 * Parse one block: header, then `cols` filler columns, scattering the indices
 * into idx[row*cols + col].  Returns 1, kCleanEof, or an ACM_ERR_* code with
 * the reader left exactly where the reference's would be.  On anything but 1
 * the contents of idx/hdr are unspecified and no patches are kept.
 */
int parse_block(ACMStream *s, TableHistory *tab, int16_t *idx, acmhip_blkhdr *hdr, PatchSink *sink);

/* parser position as acm_raw_tell() reports it (util.c:192-195) */
inline unsigned raw_position(const ACMStream *s) { return s->buf_start_ofs + s->buf_pos; }

/* rewind bookkeeping of acm_seek_pcm (util.c:230-239); the caller has already repositioned the data source */
void reset_reader(ACMStream *s);

/* consume n (< 32) bits through the exact reader (used to re-enter the stream at a remembered bit offset); 0 or error */
int skip_bits(ACMStream *s, unsigned n);

} // namespace acmfill

#endif
