/*
 * libacm.h - public C API of the MI355X-native ACM decoder.
 *
 * The declarations below (function prototypes, struct layouts, constants) reproduce the public interface of
 * libacm 1.3 - they ARE the ABI this library is a drop-in for - and that interface is
 *
 *   Copyright (c) 2004-2010, Marko Kreen
 *
 *   Permission to use, copy, modify, and/or distribute this software for any
 *   purpose with or without fee is hereby granted, provided that the above
 *   copyright notice and this permission notice appear in all copies.
 *
 *   THE SOFTWARE IS PROVIDED "AS IS" AND THE AUTHOR DISCLAIMS ALL WARRANTIES
 *   WITH REGARD TO THIS SOFTWARE INCLUDING ALL IMPLIED WARRANTIES OF
 *   MERCHANTABILITY AND FITNESS. IN NO EVENT SHALL THE AUTHOR BE LIABLE FOR
 *   ANY SPECIAL, DIRECT, INDIRECT, OR CONSEQUENTIAL DAMAGES OR ANY DAMAGES
 *   WHATSOEVER RESULTING FROM LOSS OF USE, DATA OR PROFITS, WHETHER IN AN
 *   ACTION OF CONTRACT, NEGLIGENCE OR OTHER TORTIOUS ACTION, ARISING OUT OF
 *   OR IN CONNECTION WITH THE USE OR PERFORMANCE OF THIS SOFTWARE.
 *
 * (ISC licence, kept here as it asks; the implementation behind the interface is this repository's own.)
 *
 * Source- and ABI-compatible with markokr/libacm v1.3 (reference:
 * /root/reference/src/libacm.h): same 19 entry points, same error codes, same
 * public struct layouts (x86-64: sizeof(ACMStream) == 176, offsets listed in
 * SURVEY.md 8b and asserted in csrc/acm_stream.cpp), so a program written
 * against the reference header links against libacm_hip.so unchanged.
 *
 * What differs is behind the API: acm_read() serves PCM out of a read-ahead
 * window whose blocks were bit-parsed on the host and synthesised
 * (amplitude-table unpack, subband synthesis, 16-bit write-out) on the GPU -
 * or, where no usable HIP device exists, and for streams shorter than
 * acmhip_host_synth_limit() samples (include/acm_hip.h; 128 M by default) while
 * no device is open in the process, by the library's own host synthesis
 * (csrc/acm_host_synth.cpp): like the reference, this API decodes anywhere.
 */
#ifndef __LIBACM_H
#define __LIBACM_H

#ifdef __cplusplus
extern "C" {
#endif

#define LIBACM_VERSION "1.3"        /* API level implemented (reference libacm.h:26) */

#define ACM_ID    0x032897          /* stream magic, bytes 97 28 03 (reference libacm.h:28) */
#define ACM_WORD  2                 /* bytes per output sample (reference libacm.h:29) */

/* return codes (reference libacm.h:31-39) */
#define ACM_OK                   0
#define ACM_ERR_OTHER           -1
#define ACM_ERR_OPEN            -2
#define ACM_ERR_NOT_ACM         -3
#define ACM_ERR_READ_ERR        -4
#define ACM_ERR_BADFMT          -5
#define ACM_ERR_CORRUPT         -6
#define ACM_ERR_UNEXPECTED_EOF  -7
#define ACM_ERR_NOT_SEEKABLE    -8

/* stream parameters (reference libacm.h:41-50; 8 x unsigned) */
typedef struct ACMInfo {
	unsigned channels;       /* channel count in effect (after force_chans) */
	unsigned rate;           /* sample rate, Hz */
	unsigned acm_id;
	unsigned acm_version;
	unsigned acm_channels;   /* channel count as written in the file header */
	unsigned acm_level;      /* log2 of the subband count */
	unsigned acm_cols;       /* 1 << acm_level */
	unsigned acm_rows;       /* rows per block */
} ACMInfo;

/*
 * I/O callbacks (reference libacm.h:52-69).
 *   read_func       fread()-like: fill ptr with up to n items of `size` bytes,
 *                   return items read, 0 at end of data, <0 on error.  The
 *                   decoder always calls it with size 1, n 65536.
 *   seek_func       optional; only SEEK_SET to the first data byte is needed.
 *   close_func      optional; called once from acm_close().
 *   get_length_func optional; total length in bytes.
 */
typedef struct {
	int (*read_func)(void *ptr, int size, int n, void *datasrc);
	int (*seek_func)(void *datasrc, int offset, int whence);
	int (*close_func)(void *datasrc);
	int (*get_length_func)(void *datasrc);
} acm_io_callbacks;

/*
 * Public stream object (reference libacm.h:71-100).  Callers of the reference
 * read some of these fields directly (info, total_values, data_len,
 * block_len), so the layout is part of the ABI and is kept.  The bit-reader
 * fields describe the HOST parser, which runs ahead of what acm_read() has
 * handed out; block/wrapbuf/ampbuf/midbuf are unused by this implementation
 * (the block matrix lives in HBM) and stay NULL.
 */
struct ACMStream {
	ACMInfo info;
	unsigned total_values;           /* 16-bit words in the stream, all channels */

	void *io_arg;
	acm_io_callbacks io;
	unsigned data_len;

	unsigned char *buf;              /* host parser: refill buffer */
	unsigned buf_max, buf_size, buf_pos, bit_avail;
	unsigned bit_data;
	unsigned buf_start_ofs;

	unsigned block_len;              /* words per block = acm_rows * acm_cols */
	unsigned wrapbuf_len;            /* 2 * acm_cols - 2: synthesis history depth */
	int *block;
	int *wrapbuf;
	int *ampbuf;
	int *midbuf;

	unsigned block_ready:1;
	unsigned file_eof:1;
	unsigned wavc_file:1;
	unsigned stream_pos;             /* words handed out so far */
	unsigned block_pos;              /* words handed out of the current block */
};
typedef struct ACMStream ACMStream;

/* ---- core (reference decode.c) ---- */

/*
 * Open over caller-supplied callbacks.  force_chans > 0 overrides the channel
 * count; 0 trusts the header; -1 treats plain (non-WAVC) mono files as stereo.
 * On success the stream owns io_arg (acm_close calls close_func); on failure
 * the caller still owns it and *res is untouched.
 */
int acm_open_decoder(ACMStream **res, void *io_arg, acm_io_callbacks io, int force_chans);

/*
 * Read up to nbytes of PCM.  wordlen must be 2; bigendianp / sgned pick the
 * sample layout.  buf == NULL decodes and discards.  Returns bytes produced
 * (never more than the rest of the current block), 0 at end of stream, or an
 * ACM_ERR_* code.
 */
int acm_read(ACMStream *acm, void *buf, unsigned nbytes, int bigendianp, int wordlen, int sgned);
void acm_close(ACMStream *acm);

/* ---- convenience (reference util.c) ---- */

int acm_open_file(ACMStream **acm, const char *filename, int force_chans);
const ACMInfo *acm_info(ACMStream *acm);
int acm_seekable(ACMStream *acm);
unsigned acm_bitrate(ACMStream *acm);
unsigned acm_rate(ACMStream *acm);
unsigned acm_channels(ACMStream *acm);
unsigned acm_raw_total(ACMStream *acm);
unsigned acm_raw_tell(ACMStream *acm);
unsigned acm_pcm_total(ACMStream *acm);
unsigned acm_pcm_tell(ACMStream *acm);
unsigned acm_time_total(ACMStream *acm);
unsigned acm_time_tell(ACMStream *acm);
/* keep calling acm_read until nbytes are delivered, the stream ends, or it fails */
int acm_read_loop(ACMStream *acm, void *dst, unsigned nbytes, int bigendianp, int wordlen, int sgned);
int acm_seek_pcm(ACMStream *acm, unsigned pcm_pos);
int acm_seek_time(ACMStream *acm, unsigned pos_ms);
const char *acm_strerror(int err);

#ifdef __cplusplus
}
#endif

#endif
