/*
 * acm_synth.h - synthetic ACM v1 bitstream writer.
 *
 * The reference has no encoder and ships no sample files, so
 * every test fixture and every benchmark input is produced here, from the
 * format exactly as the reference parses it (reader:
 * /root/reference/src/decode.c:586-589 block header, :491-502 column loop,
 * :181-476 the filler payloads, :712-752 the 14-byte header, :687-710 WAVC).
 *
 * Deterministic: the only randomness is a documented xorshift64* generator
 * seeded from `seed` through splitmix64, so a (params) tuple names one file.
 * All produced filler indices stay inside [-2^pwr, 2^pwr) unless
 * `allow_out_of_range` is set (hazard H1 fixtures).
 */
#ifndef ACM_SYNTH_H
#define ACM_SYNTH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
	ACMSYNTH_MIX_SPEECH  = 0,  /* 40 % linear (ind in [3,pwr+1]), 20 % zero, 40 % uniform over the 11 k/t codes */
	ACMSYNTH_MIX_UNIFORM = 1,  /* uniform over the 26 valid codes (linear capped at pwr+1) */
	ACMSYNTH_MIX_SINGLE  = 2   /* every column uses `single_code` */
};

typedef struct acmsynth_params {
	uint64_t seed;
	uint32_t level;          /* 0..15 */
	uint32_t rows;           /* 1..4095 */
	uint32_t nblocks;        /* blocks actually written */
	uint32_t channels;       /* header field, 1 or 2 */
	uint32_t rate;           /* header field, >= 4096 */
	uint32_t total_values;   /* header field; 0 = nblocks*rows*cols */
	uint32_t pwr_min, pwr_max;   /* per-block pwr ~ U[pwr_min,pwr_max], 0..15 */
	uint32_t val_min, val_max;   /* per-block val ~ U[val_min,val_max], 0..65535 */
	uint32_t mix;            /* ACMSYNTH_MIX_* */
	uint32_t single_code;    /* for MIX_SINGLE */
	uint32_t wavc;           /* 1 = prepend the 28-byte WAVC header */
	uint32_t allow_out_of_range; /* 1 = let linear widths exceed pwr+1 (H1) */
	uint32_t prime_table;    /* 1 = block 0 gets pwr 15, so every amplitude-table entry has been written
	                            before any later block can read a stale one (keeps H1 cases deterministic
	                            in the reference, whose table starts as uninitialised heap) */
} acmsynth_params;

/* fill *p with the BASELINE.md section 5 defaults (pwr U[4,12], val U[1,255], speech mix, mono 22050) */
void acmsynth_defaults(acmsynth_params *p);

/* worst-case output size in bytes for these parameters */
size_t acmsynth_bound(const acmsynth_params *p);

/* write the file image; returns bytes written, or 0 if cap is too small / params invalid */
size_t acmsynth_generate(const acmsynth_params *p, uint8_t *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
