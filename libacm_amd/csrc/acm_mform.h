/*
 * acm_mform.h - the byte-plane stager as a writer that takes a stream pair by pair (acm_pack.cpp).  Internal to libacm_hip.so:
 * acmhip_mform_rows (include/acm_hip.h) is begin + one put_pair per row pair + end; the host parser's fused staging
 * (acm_stage_file_mform, acm_stream.cpp) feeds it block by block while the block it has just parsed is still in the cache.
 */
#ifndef ACM_MFORM_H
#define ACM_MFORM_H

#include <stddef.h>
#include <stdint.h>

#include "acm_hip.h"

struct AcmMformWriter {
	uint32_t level;
	size_t qn, sigma, cols;
	bool split;                     /* the six-stage form (64 columns of a class side by side): a 16-bit index as two signed bytes, no 4-bit class */
	bool nib12;                     /* ... of a level of the chunk kernel: + the 12-bit class (a signed low byte and a signed high NIBBLE) */
	uint8_t *out;
	uint64_t blob_base, at;
	acmhip_mform_pair *pairs;
	uint64_t npairs;                /* entries written, the pair in front included */
};

int acm_mform_begin(AcmMformWriter *w, uint32_t level, uint8_t *out, uint64_t blob_base, acmhip_mform_pair *pairs);   /* writes the pair of zeros in front */
int acm_mform_put_pair(AcmMformWriter *w, const int16_t *two_rows);     /* ACMHIP_OK, ACMHIP_ERR_ARG */
uint64_t acm_mform_end(AcmMformWriter *w);                               /* read slack behind the last pair; bytes used */
/* rows [2 * pair - 2 .. ] back into int16: the two rows of pair-table entry `entry` (entry 0 = the pair in front) */
int acm_mform_get_pair(uint32_t level, const uint8_t *blob, acmhip_mform_pair entry, int16_t *two_rows);

#endif
