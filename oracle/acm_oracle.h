/*
 * acm_oracle.h - CPU restatement of the reference ACM decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped decode path may include,
 * link or dlopen this.  Allowed users: tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py (as the checker / the timed CPU baseline).
 *
 * Parity status: PINNED.  The restatement is checked bit-for-bit against the
 * real reference (markokr/libacm v1.3 compiled by oracle/Makefile `ref` into
 * oracle/_ref/) by tests/test_oracle_vs_ref.py in the authoring container and
 * against the committed vectors in tests/golden/ everywhere else.  The
 * reference ships no golden vectors of its own.
 *
 * Every function names the reference lines it restates (paths relative to
 * /root/reference).
 */
#ifndef ACM_ORACLE_H
#define ACM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* same numeric values as src/libacm.h:31-39 */
#define ACMO_OK                  0
#define ACMO_ERR_OTHER          -1
#define ACMO_ERR_OPEN           -2
#define ACMO_ERR_NOT_ACM        -3
#define ACMO_ERR_READ_ERR       -4
#define ACMO_ERR_BADFMT         -5
#define ACMO_ERR_CORRUPT        -6
#define ACMO_ERR_UNEXPECTED_EOF -7
#define ACMO_ERR_NOT_SEEKABLE   -8
/* internal sentinel, src/decode.c:31 */
#define ACMO_CLEAN_EOF          -99

/* field-for-field mirror of ACMInfo, src/libacm.h:41-50 */
typedef struct acmo_info {
	unsigned channels;
	unsigned rate;
	unsigned acm_id;
	unsigned acm_version;
	unsigned acm_channels;
	unsigned acm_level;
	unsigned acm_cols;
	unsigned acm_rows;
} acmo_info;

typedef struct acmo_stream acmo_stream;

/*
 * Open a decoder over an in-memory file image.  `max_read` caps how many bytes
 * one simulated read_func call hands back (0 = no cap, i.e. the 65536 the
 * reference asks for, src/decode.c:51); it exists to reproduce short-read
 * behaviour.  `seekable` = whether a seek_func is present (src/util.c:220).
 */
int acmo_open_mem(acmo_stream **out, const uint8_t *data, size_t len,
		  int force_chans, unsigned max_read, int seekable);
void acmo_close(acmo_stream *s);

/* restated public API (src/libacm.h:134-170) */
int acmo_read(acmo_stream *s, void *dst, unsigned nbytes, int bigendianp, int wordlen, int sgned);
int acmo_read_loop(acmo_stream *s, void *dst, unsigned nbytes, int bigendianp, int wordlen, int sgned);
int acmo_seek_pcm(acmo_stream *s, unsigned pcm_pos);
int acmo_seek_time(acmo_stream *s, unsigned pos_ms);
const acmo_info *acmo_get_info(acmo_stream *s);
int acmo_seekable(acmo_stream *s);
unsigned acmo_bitrate(acmo_stream *s);
unsigned acmo_rate(acmo_stream *s);
unsigned acmo_channels(acmo_stream *s);
unsigned acmo_raw_total(acmo_stream *s);
unsigned acmo_raw_tell(acmo_stream *s);
unsigned acmo_pcm_total(acmo_stream *s);
unsigned acmo_pcm_tell(acmo_stream *s);
unsigned acmo_time_total(acmo_stream *s);
unsigned acmo_time_tell(acmo_stream *s);
unsigned acmo_total_values(acmo_stream *s);
unsigned acmo_block_len(acmo_stream *s);
const char *acmo_strerror(int err);

/*
 * Hot-path probes.
 *
 * acmo_fill_next_block: the first half of decode_block (src/decode.c:580-604):
 * block header, amplitude table, fill_block - WITHOUT juggle.  Copies the
 * rows*cols int32 block matrix (row-major) to raw_out and reports pwr/val.
 * Returns 1, ACMO_CLEAN_EOF, or a negative error exactly like decode_block.
 * Do not mix with acmo_read on the same stream.
 */
int acmo_fill_next_block(acmo_stream *s, int32_t *raw_out, int *pwr, int *val);

/* juggle_block (src/decode.c:528-577) on a caller-owned block + wrap state */
void acmo_juggle_block(unsigned level, unsigned rows, int32_t *block, int32_t *wrap);

/* output_values (src/decode.c:657-677) */
int acmo_output(const int32_t *src, unsigned char *dst, int n, int level,
		int bigendianp, int wordlen, int sgned);

/*
 * Whole-file convenience used by tests and the CPU baseline: decode through
 * acmo_read_loop in `step_bytes` requests (acmtool uses 8192, src/acmtool.c:275)
 * until EOF / error / cap.  Returns number of 16-bit words written; *status
 * receives the last acmo_read_loop return value (0 = EOF, <0 = error).
 */
long acmo_decode_all(const uint8_t *data, size_t len, int force_chans,
		     int16_t *pcm, size_t cap_words, unsigned step_bytes,
		     int bigendianp, int sgned, int *status);

/* decode and discard (the `acmtool -d -n` shape); returns words decoded */
long acmo_decode_discard(const uint8_t *data, size_t len, int force_chans, int *status);

#ifdef __cplusplus
}
#endif
#endif
