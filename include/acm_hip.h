/*
 * acm_hip.h - C ABI of the MI355X-native ACM block-synthesis hot path.
 *
 * This is the boundary a maintainer of the reference would bind to replace the
 * data-parallel half of decode_block()/acm_read() (INTEGRATION.md shows the
 * patch).  Plain C: pointers, sizes, PODs; no C++ or torch types.  Every entry
 * point names the reference code it replaces (paths relative to
 * /root/reference/src).
 *
 * Division of labour (BASELINE.json north_star):
 *   host   - sequential bitstream reader + the 15 filler parsers
 *            (decode.c:41-163, 181-502) -> "staged" form: one int16 filler
 *            index per sample in PCM order + one acmhip_blkhdr per block.
 *   device - amplitude-table unpack (decode.c:592-600 + set_pos :174-177, i.e.
 *            value = idx * val), juggle_block subband synthesis (:508-577) and
 *            16-bit write-out (:617-677) as HIP kernels on gfx950.
 *
 * There is no CPU implementation of the device half in this library; every
 * call fails with ACMHIP_ERR_NO_DEVICE when no HIP device is usable.
 */
#ifndef ACM_HIP_H
#define ACM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACMHIP_OK             0
#define ACMHIP_ERR_NO_DEVICE -101   /* no usable HIP device / runtime */
#define ACMHIP_ERR_HIP       -102   /* a HIP call failed; see acmhip_last_error() */
#define ACMHIP_ERR_ARG       -103   /* invalid argument */
#define ACMHIP_ERR_NOMEM     -104
#define ACMHIP_ERR_RANGE     -105   /* (rounds 5 / 6: an index a staged form could not hold, acmhip_mform_rows.  Every form holds the whole int16 range now;
                                       the code is kept for callers that test for it and is no longer returned) */

/* output sample layouts = the four writers of decode.c:617-655 */
#define ACMHIP_FMT_S16LE 0u
#define ACMHIP_FMT_S16BE 1u         /* bit 0: big-endian   (bigendianp, decode.c:662) */
#define ACMHIP_FMT_U16LE 2u         /* bit 1: unsigned     (!sgned,     decode.c:663) */
#define ACMHIP_FMT_U16BE 3u

/*
 * Staged block header: the (pwr,val) pair of decode.c:586-589.  On the device
 * the amplitude table midbuf[i] = i*val (decode.c:592-600) is never built;
 * value = (int32)((uint32)idx * val).  pwr is kept for the host (range check
 * of hazard H1) and for diagnostics.
 */
typedef struct acmhip_blkhdr {
	uint32_t val;            /* 0..65535 */
	uint32_t pwr;            /* 0..15 */
} acmhip_blkhdr;

/*
 * Hazard H1: a filler index outside [-2^pwr, 2^pwr) makes the
 * reference read an amplitude-table entry left behind by an EARLIER block
 * (decode.c:809-810: the table is never cleared).  The host parser resolves
 * those samples itself and ships them as patches: "sample `sample` of this
 * stream's staged array has the unpacked value `value`, whatever idx*val says".
 */
typedef struct acmhip_patch {
	uint64_t sample;         /* index into the stream's staged samples (0 = first staged sample) */
	int32_t  value;
	uint32_t stream;         /* index into the acmhip_stream_desc array */
} acmhip_patch;

/*
 * One stream (or one window of a stream) to synthesise.  The staged samples of
 * a stream are contiguous in PCM order: sample m = row*cols + col, rows
 * running on across block boundaries (block b owns rows [b*rows, (b+1)*rows)).
 * The synthesis history of decode.c:803-812 (wrapbuf) is not stored anywhere:
 * it is a pure function of the two staged rows that precede a row
 * (SURVEY.md 7.1), so a window that starts at row_begin > 0 only needs the
 * staged rows row_begin-2.. to be present.  Rows before row 0 are zeros
 * (decode.c:812, util.c:241).
 */
typedef struct acmhip_stream_desc {
	uint64_t idx_off;        /* int16 units from d_idx to staged row 0; multiple of 8 */
	uint64_t hdr_off;        /* acmhip_blkhdr units from d_hdr to the header of block 0 */
	uint64_t pcm_off;        /* int16 units from d_pcm to where sample (row_begin, 0) goes; multiple of 8 */
	uint64_t n_emit;         /* samples to write, starting at row_begin*cols (<= (nrows-row_begin)*cols) */
	uint32_t level;          /* acm_level 0..15 */
	uint32_t rows;           /* acm_rows  1..4095 */
	uint32_t nrows;          /* staged rows present = staged blocks * rows */
	uint32_t row_begin;      /* first row to emit */
} acmhip_stream_desc;

typedef struct acmhip_device acmhip_device;   /* one HIP device + the stream work is queued on */
typedef struct acmhip_plan acmhip_plan;       /* device-resident launch tables for a fixed set of stream descs */

/* which kernel family a plan may use */
#define ACMHIP_PLAN_AUTO      0u    /* one-launch kernels where they apply (levels 0-12 without H1 patches), stage-wise kernels elsewhere */
#define ACMHIP_PLAN_STAGEWISE 1u    /* force the generic stage-wise kernels (bring-up / cross-check) */
/* modifiers (or-ed in) */
#define ACMHIP_PLAN_FORM_ONLY    0x100u  /* streams that come with a byte-plane form are never launched without it: their records over the
                                          * int16 rows are not cut (half the table bytes); a launch with no form bound fails */
#define ACMHIP_PLAN_LEAN_ALWAYS  0x400u  /* whole tiles go to the lean kernels (acm_tile2 / acm_chunk) however few they are - by default a level's
                                          * tiles must fill the chip a few times over to be worth the lean kernels' lead-in tiles.  Small plans
                                          * that want the byte-plane / packed forms read; tests */
#define ACMHIP_PLAN_NO_LEAN      0x800u  /* never the lean kernels: everything on acm_fused_tile / the stage-wise kernels (cross-checks) */
#define ACMHIP_PLAN_FORCE_HALO   0x1000u /* acm_fused_tile: every tile recomputes its two halo rows (default: by tile count) */
#define ACMHIP_PLAN_FORCE_CARRY  0x2000u /* acm_fused_tile: histories carried from tile to tile through LDS (default: by tile count) */
#define ACMHIP_PLAN_UPLOAD_ASYNC 0x200u  /* acmhip_plan_create* returns with the table uploads queued, not done: every launch of the plan
                                          * waits for them on the device (an event wait: do not capture such a launch into a graph) */

const char *acmhip_last_error(void);          /* thread-local text of the last failure */
int  acmhip_device_count(void);               /* usable HIP devices, 0 if none (never fails) */

/* hip_stream: a hipStream_t to queue on (e.g. the caller's framework stream), or NULL for a private one */
int  acmhip_device_open(int ordinal, void *hip_stream, acmhip_device **out);
void acmhip_device_close(acmhip_device *dev);
int  acmhip_device_sync(acmhip_device *dev);
void *acmhip_device_stream(acmhip_device *dev);   /* the hipStream_t in use */

/* memory plumbing so that plain C programs need no HIP headers */
int  acmhip_malloc(acmhip_device *dev, size_t bytes, void **dptr);
int  acmhip_free(acmhip_device *dev, void *dptr);
int  acmhip_host_alloc(size_t bytes, void **hptr);            /* pinned */
int  acmhip_host_free(void *hptr);
int  acmhip_upload(acmhip_device *dev, void *dptr, const void *hptr, size_t bytes);    /* async on the device stream */
int  acmhip_download(acmhip_device *dev, void *hptr, const void *dptr, size_t bytes);  /* async on the device stream */
int  acmhip_memset(acmhip_device *dev, void *dptr, int byte, size_t bytes);            /* async on the device stream: a caller that wants to
                                                                                         * see every sample written (tests, bench) poisons the PCM arena first */

/*
 * Host synthesis (csrc/acm_host_synth.cpp): the SAME contract as a plan over one stream descriptor, on host pointers, without a device -
 * value = idx * val, the cascade of decode.c:508-577, the four writers of :617-655, bit-exact.  It is what acm_read() (include/libacm.h)
 * runs on a box without a usable HIP device, and for streams shorter than acmhip_host_synth_limit() samples while no device handle is
 * open in the process (bringing the HIP runtime up costs more than such a stream's whole decode).  The plan and batch calls never use it.
 * limit: default 128 Msamples (2^27) - measured, not guessed: through acmtool -d on an MI355X box a 41-Msample stream takes 0.16 s on the
 * host path (a few threads over independent tiles), 0.37 - 0.43 s through the GPU (the runtime comes up, the window crosses PCIe twice)
 * and 0.27 s in the reference; at 164 Msamples the two paths meet (0.56 s; the reference: 1.0 s).  0 = the host path only where no device exists; UINT64_MAX = never the device from acm_read().
 */
int  acmhip_host_synth(const acmhip_stream_desc *stream, const int16_t *idx, const acmhip_blkhdr *hdr,
		       const acmhip_patch *patches, size_t npatches, unsigned fmt, int16_t *pcm);
void acmhip_set_host_synth_limit(uint64_t samples);
/* acm_seek_pcm() re-enters a stream at the block in front of its target through the block index the parser keeps (default); off = the
 * reference's way, re-parsing from the first block (util.c:219-242).  Same positions, same PCM; for cross-checks and measurements. */
void acmhip_set_seek_index(int on);
uint64_t acmhip_host_synth_limit(void);

/*
 * Build the launch tables for `n` streams (+ optional H1 patches).  Streams may
 * mix levels/rows/lengths freely.  Replaces nothing in the reference (it has
 * no batching); it is what lets one launch cover thousands of decode_block()s.
 */
int  acmhip_plan_create(acmhip_device *dev, const acmhip_stream_desc *streams, size_t n,
			const acmhip_patch *patches, size_t npatches, unsigned flags,
			acmhip_plan **out);
void acmhip_plan_destroy(acmhip_plan *plan);

/*
 * THE hot path.  Queues unpack + juggle_block + write-out for every stream of
 * the plan: replaces decode.c:592-600 (table), :174-177 (lookup), :528-577
 * (juggle_block), :657-677 (output_values).  d_idx / d_hdr / d_pcm are device
 * pointers; fmt is ACMHIP_FMT_*.  Asynchronous on the device's stream.
 */
int  acmhip_plan_launch(acmhip_plan *plan, const int16_t *d_idx, const acmhip_blkhdr *d_hdr,
			int16_t *d_pcm, unsigned fmt);

/* ------------------------------------------------------------------------
 * Packed staged form: filler class per column pair + fixed-width packed residuals
 * (BASELINE.json north_star: "per-block filler indices plus packed residuals").
 *
 * What set_pos() would store (decode.c:174-177) is an index whose range the column's filler fixes: 0 for the zero filler,
 * |idx| <= 5 for every k / t filler, `ind` bits for a linear one (decode.c:181-476).  Instead of one int16 per sample the
 * host stager ships, for the whole tiles of a stream (acmhip_packed_tile_rows(level) rows x cols, the unit the lean tile
 * kernel works on), per GROUP of rows (acmhip_packed_group_rows) and per PAIR of adjacent columns a width class - 0, 4, 8
 * or 16 bits per index, the narrowest that holds every index of the pair in those rows - and the indices at that width.
 * Column pairs of one class are stored together, so that a wavefront of the kernel unpacks with one code path:
 *
 *   tile  -> acmhip_packed_slots(level) consecutive chunk descriptors, wave-major: the kernel's wave
 *            w owns entries [w * S, (w + 1) * S) of them (S = slots / waves); unused entries have kind 0;
 *   chunk -> 64 UNITS of one group and one class: 64 / NQ column pairs x NQ row quads (NQ = group rows / 4), a unit being
 *            2 columns x the 4 rows row0 + q, row0 + q + NQ, row0 + q + 2 NQ, row0 + q + 3 NQ of its group.  Its blob, at
 *            blob_off16 * 16 bytes: 64 / NQ uint16 (entries >= count unused) that name the column pairs by where their
 *            first column sits in the kernel's padded LDS row (dwords: 2 p + p / 16), then the units, pair-major (pair r,
 *            quad q at index r * NQ + q): kind 4: four dwords, dword k = the unit's k-th row, (int16) column 2p |
 *            (int16) column 2p + 1 << 16;  kind 3: two dwords of int8 (row 0, col 2p), (row 0, 2p + 1), (row 1, 2p),
 *            (row 1, 2p + 1) | rows 2, 3;  kind 2: one dword of eight int4 in the same order;  kind 1: nothing (all zeros).
 *
 * The int16 form stays what every other kernel reads (ragged tails, windows, streams with H1 patches, levels the lean
 * kernel does not take), and what a plan falls back to while no packed arenas are bound.
 * ---------------------------------------------------------------------- */
typedef struct acmhip_packed_chunk {
	uint32_t blob_off16;     /* 16-byte units from the blob arena's base */
	uint16_t count;          /* column pairs in this chunk, 1 .. 64 / NQ */
	uint8_t  kind;           /* ACMHIP_PK_*; 0 = unused entry */
	uint8_t  row0;           /* first row of the chunk's group inside its tile */
} acmhip_packed_chunk;
#define ACMHIP_PK_ZERO   1u
#define ACMHIP_PK_NIBBLE 2u
#define ACMHIP_PK_BYTE   3u
#define ACMHIP_PK_WORD   4u

/* beside acmhip_stream_desc i: rows [0, ntiles * tile_rows) of the stream are staged in packed form as well */
typedef struct acmhip_packed_stream {
	uint64_t chunk_off;      /* ACMHIP_FORM_PACKED: the chunk-table entry the stream's first tile starts at (tile k: chunk_off + k * slots
				  * of its level);  ACMHIP_FORM_BYTEPLANE: the pair-table entry of the pair of zeros in front of the stream
				  * (acmhip_mform_rows) */
	uint32_t ntiles;         /* 0: this stream has no second staged form */
	uint32_t form;           /* ACMHIP_FORM_* */
} acmhip_packed_stream;
#define ACMHIP_FORM_PACKED    0u
#define ACMHIP_FORM_BYTEPLANE 1u

int  acmhip_packed_tile_rows(uint32_t level);     /* rows per packed tile, 0 if the level has no packed form */
int  acmhip_packed_group_rows(uint32_t level);    /* rows that share one width class per column pair */
int  acmhip_packed_slots(uint32_t level);         /* chunk descriptors per tile */
/* upper bound of the blob bytes `ntiles` tiles of `level` can take (incl. the slack the kernel may read past the last unit) */
int  acmhip_pack_bound(uint32_t level, uint64_t ntiles, uint64_t *max_blob_bytes);
/*
 * Host stager, packed half (no device involved): packs `ntiles` consecutive tiles of ONE stream from staged indices
 * idx[row * cols + col] (as acm_stage_file writes them; row 0 = the first row of the first tile) into
 * chunks[0 .. ntiles * slots) and blob[0 .. *blob_bytes).  blob_base: where blob[0] will sit in the batch's blob arena
 * (bytes, multiple of 16); the offsets written are absolute.  Replaces nothing in the reference by itself: it is the
 * storage side of set_pos (decode.c:174-177).
 */
int  acmhip_pack_tiles(uint32_t level, const int16_t *idx, uint64_t ntiles, acmhip_packed_chunk *chunks, uint8_t *blob,
		       uint64_t blob_base, uint64_t *blob_bytes);
/* the inverse, for tests and tools: one tile's descriptors -> idx[tile_rows * cols] (blob = the arena's base) */
int  acmhip_unpack_tile(uint32_t level, const acmhip_packed_chunk *tile_chunks, const uint8_t *blob, int16_t *idx);

/* acmhip_plan_create with the packed form of (some of) the streams: packed[i].ntiles whole tiles of stream i, from its
 * row 0 on, go to the packed build of the lean tile kernel once arenas are bound.  packed may be NULL. */
int  acmhip_plan_create_packed(acmhip_device *dev, const acmhip_stream_desc *streams, size_t n, const acmhip_packed_stream *packed,
			       const acmhip_patch *patches, size_t npatches, unsigned flags, acmhip_plan **out);
/* device tables the packed tiles of this plan are read from by every later launch (both NULL: back to the int16 form) */
int  acmhip_plan_bind_packed(acmhip_plan *plan, const acmhip_packed_chunk *d_chunks, const uint8_t *d_blob);

/* ------------------------------------------------------------------------
 * Byte-plane staged form: the staged indices in the order the matrix cores take them, every row pair at the narrowest
 * width class that holds it.
 *
 * The first stages of the cascade (decode.c:527-590) are linear; over one residue class of the columns (columns
 * c + q * cols/G, q < G, G = acmhip_mform_group(level)) the first log2(G) of them are one banded integer matrix over
 * rows, and a level that has such a build (acmhip_mform_tile_rows(level) != 0: levels 7-14) runs it on
 * v_mfma_i32_16x16x32_i8 / 16x16x64_i8 instead of the vector ALU: its operands are the staged indices themselves (the
 * multiply by the block's val moves behind the matrix), as signed bytes.  A block's indices lie in [-2^pwr, 2^pwr)
 * (decode.c:592-600: the amplitude table has 2^(pwr+1) entries), so quiet blocks need fewer bits.
 *
 *   stream -> a pair of zero rows (what the cascade sees in front of row 0), then its row pairs (rows 2P, 2P + 1), each at
 *             its own width class, back to back (a pair takes 64 bytes or a multiple); acmhip_mform_pair k of the stream =
 *             where pair k - 1 starts (64-byte units from the arena's base) << 2 | class code;
 *   pair   -> its two rows, each row = cols/G residues c, each residue = the G indices of columns c, c + cols/G, ...
 *
 * G = 64, levels 8-14 (six stages on the matrix cores: acm_chunk at levels 8-12, acm_tile2 at 13 / 14).  The pair in front is
 * at 8 bits.  Per residue:
 *             ACMHIP_BP_BYTE  (2)  64 bytes: idx itself (every index of the pair in [-128, 127])
 *             ACMHIP_BP_NIB12 (1)  levels 8-12 only: idx = 256 hi + lo with lo a signed byte and hi a signed NIBBLE (every index of
 *                                  the pair in [-2176, 1919]): 64 low bytes, then 32 bytes of high nibbles in the order the kernel's
 *                                  lanes read them - lane ks < 4 (columns q = 16 ks .. 16 ks + 15 of the class) takes the 8 bytes
 *                                  at 64 + 8 ks: dword d < 2 holds its elements 8 d .. 8 d + 7, element 8 d + b (b < 4) in the HIGH
 *                                  nibble of byte b, element 8 d + 4 + b in the low one (x & 0xf0f0f0f0 and (x << 4) & 0xf0f0f0f0
 *                                  are then the matrix operand bytes hi << 4 of elements 8 d .. + 3 and 8 d + 4 .. + 7)
 *             ACMHIP_BP_WORD  (3)  idx = 256 hi + lo with BOTH bytes signed: 64 low bytes, then 64 high bytes.  That ends at
 *                                  32 639; a pair with a larger index (it takes pwr 15) is written as
 *             ACMHIP_BP_WORDU (0)  idx = 256 hi + lo with hi = idx >> 8 (signed) and lo the UNSIGNED low byte,
 *                                  stored minus 128 ((idx & 0xff) ^ 0x80: a signed byte for the matrix instruction; the kernels
 *                                  add 128 x val x the coefficient row sums of such rows back): the whole int16 range, same
 *                                  64 + 64 bytes.  (Levels 8-14; no stream is refused for its indices any more)
 * G = 8, level 7 (three stages, acm_tile2; also levels 8-9, and G = 16 at 10-14, in a -DACM_TUNING build run with ACM_K3=0).  The
 * pair in front is at 4 bits.  Per residue:
 *             ACMHIP_BP_WORD    G low bytes ((idx & 0xff) ^ 0x80: signed bytes, the kernel adds the 128 back through the
 *                               accumulator), then G high bytes (idx >> 8)
 *             ACMHIP_BP_BYTE    G bytes (idx itself; every index of the pair in [-128, 127])
 *             ACMHIP_BP_NIBBLE  G / 2 bytes (every index in [-8, 7]): per dword eight indices plus 8 each, index 8 j + i in
 *                               nibble 2 i, index 8 j + 4 + i in nibble 2 i + 1 (i < 4)
 *
 * Only whole tiles are staged this way, the ragged tail of a stream stays int16 (as with the packed form).  Whoever writes the
 * form (acmhip_mform_rows, acm_stage_file_mform, the device parser) and the kernel that reads it are the same library build: the
 * form of a level follows the kernel the build ships for it.
 * ---------------------------------------------------------------------- */
typedef uint32_t acmhip_mform_pair;
#define ACMHIP_BP_WORDU  0u     /* the six-stage form of levels 8-12: 16 bits over the WHOLE int16 range (see above) */
#define ACMHIP_BP_NIBBLE 1u
#define ACMHIP_BP_NIB12  1u     /* the same code in the six-stage form of levels 8-12 (which has no 4-bit class): 12 bits, see above */
#define ACMHIP_BP_BYTE   2u
#define ACMHIP_BP_WORD   3u
int  acmhip_mform_tile_rows(uint32_t level);     /* rows per tile of the matrix-core build, 0 if the level has none */
int  acmhip_mform_group(uint32_t level);         /* columns of a residue class kept side by side: 64 (levels 8-14, a six-stage first pass), 8 (level 7: three); 0 if none */
uint64_t acmhip_mform_bytes(uint32_t level, uint64_t nrows);     /* upper bound of the bytes nrows rows take (every pair at 16 bits, + the pair in front and 64 bytes of read slack) */
uint64_t acmhip_mform_pairs(uint64_t nrows);                     /* pair-table entries of nrows rows: nrows / 2 + 1 */
/*
 * Host stager, byte-plane half: idx[row * cols + col] (as acm_stage_file writes them; nrows even) -> out[0 .. *bytes_used) and
 * pairs[0 .. nrows / 2].  blob_base: where out[0] will sit in the batch's arena (bytes, multiple of 64, < 64 GB); the
 * offsets written are absolute.
 */
int  acmhip_mform_rows(uint32_t level, const int16_t *idx, uint64_t nrows, uint8_t *out, uint64_t blob_base, acmhip_mform_pair *pairs,
		       uint64_t *bytes_used);
/* the inverse, for tests and tools (blob = the arena's base) */
int  acmhip_mform_unrows(uint32_t level, const uint8_t *blob, const acmhip_mform_pair *pairs, uint64_t nrows, int16_t *idx);
/* device arena and pair table the byte-plane tiles of this plan are read from by every later launch (both NULL: back to the int16 form).
 * The table must be readable 32 entries past its last one (the kernel fetches entries in groups through the scalar cache), the arena
 * 64 bytes past the last pair (acmhip_mform_rows leaves that slack behind every block it writes). */
int  acmhip_plan_bind_mform(acmhip_plan *plan, const uint8_t *d_mform, const acmhip_mform_pair *d_pairs);

/* introspection for benchmarks/tests */
typedef struct acmhip_plan_stats {
	uint64_t samples;        /* total n_emit */
	uint64_t tiles;          /* fused-kernel workgroups */
	uint32_t fused_streams;  /* streams handled by a one-launch kernel (fused tile kernel, levels 5-12; register kernel, levels 0-4) */
	uint32_t stagewise_streams;
	uint32_t launches;       /* kernel launches per acmhip_plan_launch */
	uint32_t reserved;
	uint32_t mform_tiles;    /* tiles of streams that came with a byte-plane form (read from it while an arena is bound: acmhip_plan_bind_mform) */
	uint32_t packed_tiles;   /* tiles that have records of the packed build too (read in packed form while arenas are bound: acmhip_plan_bind_packed) */
} acmhip_plan_stats;
int  acmhip_plan_get_stats(const acmhip_plan *plan, acmhip_plan_stats *out);
/* rows [row_begin, row_begin + *rows) of stream `stream` (its position among the descriptors the plan was created from) are read from the
 * stream's second staged form by this plan's launches while one is bound - every other row it decodes, and the two rows in front of them,
 * from the int16 arena.  (row_begin > 0: a window that starts on a tile boundary of a stream with a byte-plane form - the form must hold
 * the stream's rows from row 0 to the window's last whole tile.)  0 for a stream without a second form, and for one whose form the plan does not use (a level-13 / 14 stream of a small plan, a
 * stage-wise plan): a caller that stages both forms uploads the int16 rows from max(*rows - 2, 0) on. */
int  acmhip_plan_form_rows(const acmhip_plan *plan, size_t stream, uint64_t *rows);

/*
 * Time `reps` back-to-back acmhip_plan_launch calls with HIP events recorded on
 * the launch stream; *ms_total receives the elapsed milliseconds of the whole
 * bracket (device time, includes launch gaps).  Blocks until done.
 */
int  acmhip_plan_time(acmhip_plan *plan, const int16_t *d_idx, const acmhip_blkhdr *d_hdr,
		      int16_t *d_pcm, unsigned fmt, int reps, float *ms_total);

/* ------------------------------------------------------------------------
 * Host half of the path: bit parsing into staged form.
 * Replaces decode.c:41-163 (bit reader), :181-502 (fillers, fill_block),
 * :586-589 (block header), :687-752 (headers) for whole in-memory files.
 * ---------------------------------------------------------------------- */
typedef struct acm_stage_info {
	uint32_t level, rows, cols;
	uint32_t channels;       /* in effect (after force_chans) */
	uint32_t hdr_channels;   /* as written in the header */
	uint32_t rate;
	uint32_t total_values;
	uint32_t wavc;
	uint32_t blocks;         /* blocks completely parsed into the staging arrays */
	int32_t  end_status;     /* 0 = clean end (EOF at a block/column boundary or total_values reached);
	                            ACM_ERR_* = what acm_read would have returned after those blocks */
	uint64_t npatches;       /* H1 patches produced */
	uint64_t header_bytes;   /* 14 or 42 */
} acm_stage_info;

/* header only (decode.c:712-752 + channel forcing :795-798); returns ACM_OK or ACM_ERR_NOT_ACM */
int  acm_stage_probe(const uint8_t *data, size_t len, int force_chans, acm_stage_info *info);

/*
 * Parse every block that acm_read() would decode (stops after the block that
 * covers total_values, at a clean EOF, or at the first error) into
 *   idx[b*block_len + row*cols + col]  (row-major, PCM order) and hdr[b].
 * max_blocks bounds the arrays; patches (may be NULL when max_patches == 0)
 * receive H1 fix-ups with .stream = 0.  Returns ACM_OK (details in *info) or
 * ACM_ERR_NOT_ACM / ACMHIP_ERR_ARG.
 */
int  acm_stage_file(const uint8_t *data, size_t len, int force_chans,
		    int16_t *idx, acmhip_blkhdr *hdr, size_t max_blocks,
		    acmhip_patch *patches, size_t max_patches, acm_stage_info *info);

/*
 * The same, with the whole tiles of the stream written in the byte-plane form as they are parsed (block by block, out of the cache)
 * instead of by a second pass over idx: rows [0, *mf_rows) - whole tiles of the lean kernel - go to mf_out (pairs / mf_base /
 * *mf_bytes as acmhip_mform_rows writes them; room for acmhip_mform_bytes() / acmhip_mform_pairs() of every row the header
 * promises) and idx receives only the rows from *mf_rows - 2 on, which is all the int16 kernels read of such a stream.
 * *mf_rows = 0: the stream has no form (its level, H1 patches - then info->npatches says so and the caller stages again
 * with room for them -, at levels 13 / 14 an index the form cannot hold, a file that ends early) and idx holds every row.  Any block
 * height: a row pair may lie across two blocks.
 */
int  acm_stage_file_mform(const uint8_t *data, size_t len, int force_chans, int16_t *idx, acmhip_blkhdr *hdr, size_t max_blocks,
			  acm_stage_info *info, uint8_t *mf_out, uint64_t mf_base, acmhip_mform_pair *pairs, uint64_t *mf_rows,
			  uint64_t *mf_bytes);

/* ------------------------------------------------------------------------
 * Batch front end (no reference counterpart; BASELINE.json "batch-of-files").
 * Decodes n in-memory ACM files on ONE device: threaded host staging ->
 * pinned buffers -> H2D -> acmhip_plan_launch -> D2H, pipelined in chunks of
 * whole streams so that the five steps overlap.
 * ---------------------------------------------------------------------- */
typedef struct acm_batch_item {
	const uint8_t *data;     /* in:  file image */
	size_t   len;            /* in */
	int16_t *pcm;            /* in:  caller buffer for total_values words (may be NULL: decode and discard) */
	size_t   pcm_cap;        /* in:  capacity of pcm in 16-bit words */
	uint64_t words;          /* out: words actually decoded (<= total_values) */
	int32_t  status;         /* out: ACM_OK, or the ACM_ERR_* that ended the stream / rejected the file */
	uint32_t level, rows, channels, rate, total_values;   /* out */
	uint32_t reserved;
	uint64_t dev_off;        /* out: 16-bit word offset of this stream's PCM inside opts->d_pcm (device-resident output) */
} acm_batch_item;

typedef struct acm_batch_opts {
	int      force_chans;
	unsigned fmt;            /* ACMHIP_FMT_* */
	int      threads;        /* host staging threads, 0 = hardware concurrency */
	unsigned plan_flags;     /* ACMHIP_PLAN_* */
	unsigned parse;          /* ACM_BATCH_PARSE_* */
	unsigned flags;          /* ACM_BATCH_* below (0: none) */
	void    *d_pcm;          /* NULL: PCM goes to items[i].pcm in host memory.  Otherwise a device buffer of
	                            d_pcm_words 16-bit words (>= acm_batch_pcm_words()): PCM stays in HBM, stream i at
	                            d_pcm + items[i].dev_off, nothing is copied back (items[i].pcm is ignored) */
	uint64_t d_pcm_words;
	const struct acm_batch_prestaged *prestaged;    /* NULL, or what acm_batch_prestage made of these very items: the bit
	                                                   parsing is done already (forces ACM_BATCH_PARSE_HOST semantics) */
} acm_batch_opts;

/* opts->flags */
#define ACM_BATCH_STAGE_PACKED 2u   /* host parsing only: the pool also packs the whole tiles of every clean stream (acmhip_pack_tiles) and the
                                       upload carries the packed form + the int16 rows the other kernels still read (ragged tails) instead
                                       of the whole int16 arena - about half the bytes over PCIe and through HBM, for ~30 % more host
                                       work per stream and a synthesis launch that is 10-18 % slower (acm_tile2p).  Off by default. */
#define ACM_BATCH_STAGE_BYTEPLANE 4u /* the default, spelled out: whoever parses the bits - the host pool (acm_stage_file_mform) or the device parser's
                                       column kernel - writes the whole tiles of every clean stream of levels 7-14 in the byte-plane form (same bytes as
                                       int16 rows) and only the rows the other kernels still read as int16; the synthesis launch runs its first pass on
                                       the matrix cores (+3 ... +24 % by level).  Wins over ACM_BATCH_STAGE_PACKED when both are set. */
#define ACM_BATCH_STAGE_INT16  8u   /* every row is staged as int16, no second form (the round-1 ... round-4 default; measurements, cross-checks) */
#define ACM_BATCH_RANGES(n)    (((unsigned)(n) & 0xFFu) << 8)  /* device parsing: walk and decode the batch in n block ranges (1 = in one piece;
                                       0 = the library decides: by the length of the longest stream).  Bits 8-15 of opts->flags */
#define ACM_BATCH_PCM_PINNED   1u   /* every items[i].pcm is pinned host memory (acmhip_host_alloc): the read-back engine writes
                                       the PCM straight into it, stream by stream, instead of through the library's own pinned
                                       arena and a host copy (taken for streams of 64 KB of PCM and more on average) */

/* where the bit parsing of a batch runs */
#define ACM_BATCH_PARSE_HOST   0u   /* host thread pool (default; the exact reader, any stream) */
#define ACM_BATCH_PARSE_DEVICE 1u   /* one GPU lane per stream for clean streams; streams the device parser is not
                                       sure about (data running out, corrupt symbols, hazard H1, files >= 256 MiB) are
                                       re-parsed by the host reader.  Pays off for thousands of streams per batch. */
#define ACM_BATCH_PARSE_AUTO   2u   /* DEVICE when the batch is worth at least 5 x threads streams of its longest stream's
                                       size (9 x from 1024 streams, 16 x from 32 K streams on), HOST below that: walking a
                                       stream is sequential, and a wavefront walks 5-16x slower than one host core parses */

typedef struct acm_batch_timing {
	double stage_s;          /* wall clock until the last stream was bit-parsed (headers included, allocation not) */
	double h2d_s, kernel_s, d2h_s;   /* device-side durations summed over the pipeline's chunks; they overlap
	                                    each other and the parsing, so they do not add up to total_s */
	double total_s;          /* wall clock of the whole call */
	uint64_t samples;
	double alloc_s;          /* pinned + device arena allocation */
	uint64_t device_parsed;  /* ACM_BATCH_PARSE_DEVICE: streams staged by the device parser ... */
	uint64_t host_parsed;    /* ... and streams (re)parsed by the host reader */
	uint64_t packed_streams; /* ACM_BATCH_STAGE_PACKED: streams whose whole tiles travelled in packed form */
	uint64_t h2d_bytes;      /* staged bytes (indices in either form, block headers, file images for the device parser) sent to the device */
} acm_batch_timing;

int  acm_batch_decode(acmhip_device *dev, acm_batch_item *items, size_t n,
		      const acm_batch_opts *opts, acm_batch_timing *timing);

/*
 * The host half of acm_batch_decode ahead of time, WITHOUT a device: bit-parses the items' files into staged form
 * (host thread pool, opts->threads / force_chans as above) kept in memory this call allocates.  A later
 * acm_batch_decode on the same items with opts->prestaged = *out copies that into its upload arenas instead of
 * parsing - a front end can parse the next groups of files while the device works on the current one, or while the
 * HIP runtime is still coming up (libacm_amd/csrc/acmtool.c -B).  *seconds (may be NULL): wall clock of the parsing.
 */
typedef struct acm_batch_prestaged acm_batch_prestaged;
int  acm_batch_prestage(const acm_batch_item *items, size_t n, const acm_batch_opts *opts, acm_batch_prestaged **out, double *seconds);
void acm_batch_prestage_free(acm_batch_prestaged *p);

/* Optional, for applications that know they are about to decode through the libacm.h calls (acmtool -d): start opening
 * the process-wide default device - HIP runtime, device handle, the kernels' code object - on a thread of the library's
 * own and return at once.  The first acm_read() that needs the GPU waits for it; until then the stream keeps parsing
 * ahead on the calling thread.  Without this call the device is opened by that first acm_read() itself (a process that
 * only opens, seeks or inspects streams never touches the GPU).  No reference counterpart. */
void acmhip_prewarm(void);

/* 16-bit words of device memory acm_batch_decode needs for the PCM of these files (headers only are read;
 * every stream is padded to a multiple of 64 words) - the size of opts->d_pcm for device-resident output */
uint64_t acm_batch_pcm_words(const acm_batch_item *items, size_t n, int force_chans);

#ifdef __cplusplus
}
#endif
#endif
