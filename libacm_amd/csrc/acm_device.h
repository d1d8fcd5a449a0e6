/*
 * acm_device.h - structures shared between the launch planner (host) and the
 * HIP kernels.  Internal; the public boundary is include/acm_hip.h.
 */
#ifndef ACM_DEVICE_H
#define ACM_DEVICE_H

#include <stdint.h>

#include "acm_hip.h"

/* per-stream record as the kernels see it */
struct AcmDevStream {
	uint64_t idx_off;      /* int16 units to staged row 0 */
	uint64_t hdr_off;      /* blkhdr units to block 0 */
	uint64_t pcm_off;      /* int16 units to the first emitted sample */
	uint64_t n_emit;       /* samples to emit from row_begin*cols */
	uint64_t scratch_off;  /* stage-wise path: int32 units into the scratch planes */
	uint32_t level;
	uint32_t rows;
	uint32_t nrows;        /* staged rows present */
	uint32_t row_begin;    /* first emitted row */
	uint32_t halo_row;     /* first staged row the kernels read: max(row_begin-2, 0) */
	uint32_t pad;
};

/* one workgroup of the fused kernel: `T` payload rows starting at row0 */
struct AcmTile {
	uint32_t stream;
	int32_t row0;          /* first payload row (a carry-mode lead-in tile may start before row 0) */
	uint32_t flags;        /* ACM_TILE_*: carry-mode kernels only */
	uint32_t pad;
};
/* carry mode: consecutive tiles of a stream hand the tail of every pass's input to the next tile through LDS instead
 * of recomputing two halo rows; a workgroup walks a contiguous run of the tile table */
#define ACM_TILE_FRESH   1u    /* nothing in front of this tile: the carries start from zero (stream row 0, or a lead-in) */
#define ACM_TILE_DISCARD 2u    /* lead-in: run the passes to build the carries, store no PCM */
/* records of the chunk kernel whose chunks are ONE row (acm_chunk, level 11) */
#define ACM_TILE_ROW1    4u    /* the chunk is row 1 of its stream: one row in front of it */
#define ACM_TILE_ODD     8u    /* the chunk starts on the second row of a pair (idx_off still names the entry of the pair in front of that pair) */
#define ACM_TILE_ONEBLOCK 16u  /* chunk records: every row in reach of the chunk - the two in front of it through its last - lies in ONE block
                                * (hdr_blk), so one val scales them all (decode.c:586-600: val is per block).  Geometry only: the planner
                                * knows it without looking at a header; acm_chunk takes its fast path on it */

/* one tile of the lean kernels (acm_tile2, acm_chunk): tile_rows consecutive rows of a stream, all of them present and emitted.
 * Tiles of a stream are consecutive table entries; the first one carries ACM_TILE_FRESH when it is the stream's row 0 - or, for a
 * window into a stream (rows from a tile boundary on), it is a record of the tile IN FRONT of the window carrying ACM_TILE_DISCARD:
 * decoded like any other to build the carries, its PCM dropped. */
struct AcmTile2 {
	uint64_t idx_off;      /* int16 units: staged index of (tile row 0, column 0) */
	uint64_t pcm_off;      /* int16 units: where sample (tile row 0, column 0) goes */
	uint32_t hdr_blk;      /* blkhdr index of the block that holds tile row -2 (tile row 0 in a stream's first tile) */
	uint32_t rowpos;       /* position of that row inside its block */
	uint32_t magic;        /* ceil(2^32 / acm_rows); 0 for acm_rows == 1 */
	uint32_t flags;        /* ACM_TILE_FRESH */
};

/* resolved H1 patch for the stage-wise path: scratch[dst] = value */
struct AcmDevPatch {
	uint64_t dst;
	int32_t value;
	uint32_t pad;
};

/* device-side bit parsing (acm_parse.hip): one stream = one lane */
struct AcmParseJob {
	uint64_t file_off;     /* bytes into the file arena; multiple of 16, >= 16 zero bytes behind every file */
	uint64_t idx_off;      /* int16 units into the staged-index arenas */
	uint64_t hdr_off;      /* blkhdr units */
	uint64_t col_off;      /* uint32 units into the column-offset arena (blocks * cols entries per stream) */
	uint32_t file_len;     /* bytes */
	uint32_t data_start;   /* first bitstream byte (14, or 42 behind a WAVC prefix) */
	uint32_t level;
	uint32_t rows;
	uint32_t blocks;       /* blocks to parse */
	uint32_t range_unit;   /* block ranges of this stream are cut at multiples of this many blocks (0 / 1: anywhere): acmk_range_bound */
	/* byte-plane staging by the device parser (the chunk kernel's form: levels whose acmk_tile2m_stages is 6): rows
	 * [0, mf_rows) of the stream are written into the byte-plane arena at mf_off (bytes; the pair of zeros first), their pair-table
	 * entries from mf_pair_off on; only the rows from mf_rows - 2 on are written to the int16 arena.  mf_rows = 0: int16 throughout */
	uint64_t mf_off;
	uint32_t mf_pair_off;
	uint32_t mf_rows;
};
struct AcmParseResult {
	uint32_t blocks_done;
	uint32_t status;       /* 0 = the scan walked every block it was asked for; else the host must re-parse this stream */
	uint32_t end_bit;      /* bit offset behind the last block walked: where the next block range of the stream resumes */
	uint32_t mf_at;        /* ... and where its next block starts in the stream's byte-plane region, in 64-byte units (width from pwr) */
};

/* levels the fused tile kernel covers; its tile geometry is owned by acm_kernels.hip (acmk_fused_tile_rows) */
#define ACM_K1_MIN_LEVEL 5
#define ACM_K1_MAX_LEVEL 12
/* levels the lean tile kernel (acm_tile2) covers (measured: 32 KB tiles lose to the 64-128 KB tiles of acm_fused_tile from level 10 on) */
#define ACM_K2_MIN_LEVEL 6
#define ACM_K2_MAX_LEVEL 14
/* levels whose acm_tile2 tiles have a packed staged form and a kernel build that reads it (acm_tile2p): the 32 KB tiles of four
 * workgroups per CU whose two-row stage-0 history fits LDS beside the tile */
#define ACM_K2P_MIN_LEVEL 6
#define ACM_K2P_MAX_LEVEL 9
/* levels whose acm_tile2 build has a three-stage first pass and therefore a build that runs it on the matrix cores, fed with the
 * byte-plane staged form (acmhip_mform_rows) */
#define ACM_K2M_MIN_LEVEL 7
#define ACM_K2M_MAX_LEVEL 14
/* levels below that (cols <= 16) have their own one-launch kernel: the cascade fits one thread's registers */
#define ACM_SMALL_MAX_LEVEL 4

/* Kernel-selection and tuning switches are read from the environment only in -DACM_TUNING builds (profiles/build_variant.sh,
 * ACM_TUNING=1 python -c 'from libacm_amd import _build; _build.build_hip(True)'): what a drop-in library does must not depend on
 * its caller's environment (Weak 9).  The shipped library honours ACM_HIP_DEVICE, ACM_BATCH_TRACE and acmtool's ACMTOOL_*;
 * everything else is a plan / batch flag of include/acm_hip.h. */
#ifdef ACM_TUNING
#include <stdlib.h>
#define ACM_TUNING_ENV(name) getenv(name)
#else
#define ACM_TUNING_ENV(name) ((const char *)0)
#endif

#ifdef __cplusplus
extern "C" {
#endif
/* grow-only staging arenas owned by a device handle (acm_hip_api.cpp): pinned host slots first, device slots
 * from ACM_ARENA_D_IDX on.  hipHostMalloc of gigabytes costs ~0.1 s; the batch front end reuses them across
 * calls.  A device handle serves one batch at a time (acmhip_arena_lock/unlock bracket acm_batch_decode). */
enum {
	ACM_ARENA_H_IDX = 0, ACM_ARENA_H_HDR, ACM_ARENA_H_PCM, ACM_ARENA_H_FILES, ACM_ARENA_H_JOBS, ACM_ARENA_H_PKBLOB, ACM_ARENA_H_PKCHUNK,
	ACM_ARENA_D_IDX, ACM_ARENA_D_HDR, ACM_ARENA_D_PCM, ACM_ARENA_D_FILES, ACM_ARENA_D_COLPOS, ACM_ARENA_D_JOBS, ACM_ARENA_D_STAGE,
	ACM_ARENA_D_PKBLOB, ACM_ARENA_D_PKCHUNK, ACM_ARENA_D_BLKOFF,
	ACM_ARENA_SLOTS
};
int acmhip_arena_get(acmhip_device *dev, int slot, size_t bytes, void **out);
void acmhip_arena_lock(acmhip_device *dev);
void acmhip_arena_unlock(acmhip_device *dev);
int acmhip_copy_stream(acmhip_device *dev, void **out);          /* second stream for overlapped read-back */
#define ACM_AUX_STREAMS 2
int acmhip_aux_stream(acmhip_device *dev, int k, void **out);    /* batch pipeline: 0 = device bit parsing, 1 = file uploads */
int acmhip_report_hip(int hip_error, const char *what);
void acmhip_set_error_text(const char *text);                    /* what acmhip_last_error() returns on this thread */          /* records the text, returns ACMHIP_ERR_HIP */

/* launchers implemented in acm_kernels.hip; `stream` is a hipStream_t */
int acmk_tuning_build(void);                                     /* 1 if the library was built with -DACM_TUNING (its environment switches are live) */
int acmk_warmup(void *stream);                                   /* an empty launch: makes the runtime load the kernels' code object */
int acmk_fused_variants(void);                                   /* number of fused-kernel variants built in */
int acmk_fused_tile_rows(uint32_t level, int variant);           /* tile rows incl. the 2 halo rows, 0 if unsupported */
int acmk_fused_has_carry(uint32_t level, int variant);           /* does a carry-mode build of this geometry exist? */
int acmk_fused_grid(uint32_t level, int variant, int cus);       /* persistent workgroups the launch uses at most */
int acmk_launch_fused(uint32_t level, int variant, int cus, int carry, const AcmDevStream *d_streams, const AcmTile *d_tiles,
		      uint32_t ntiles, const int16_t *d_idx, const acmhip_blkhdr *d_hdr, int16_t *d_pcm, unsigned fmt, void *stream);
int acmk_tile2_rows(uint32_t level);                            /* rows per acm_tile2 tile, 0 if the level is not covered */
int acmk_tile2_grid(uint32_t level, int cus);
#define ACM_K2_SINK_BYTES 65536                               /* >= one tile of PCM: where lead-in tiles put theirs */
int acmk_launch_tile2(uint32_t level, int cus, const AcmTile2 *d_tiles, uint32_t ntiles, const int16_t *d_idx, const acmhip_blkhdr *d_hdr, int16_t *d_pcm,
		      int16_t *d_sink, unsigned fmt, void *stream);
/* the packed staged form (include/acm_hip.h): same tiles as acm_tile2 (records with idx_off = the tile's first entry in the chunk table and
 * hdr_blk / rowpos naming the block of tile row 0), stage-0 inputs unpacked from chunks of one width class each */
int acmk_tile2p_rows(uint32_t level);                           /* rows per packed tile (= acmk_tile2_rows), 0 if the level has no packed build */
int acmk_tile2p_group_rows(uint32_t level);                     /* rows that share a width class per column pair */
int acmk_tile2p_slots(uint32_t level);                          /* chunk descriptors per tile: waves x descriptors per wave */
int acmk_tile2p_waves(uint32_t level);
int acmk_tile2p_pad_shift(uint32_t level);                      /* the tile's LDS rows carry one pad dword per 2^shift elements */
int acmk_launch_tile2p(uint32_t level, int cus, const AcmTile2 *d_tiles, uint32_t ntiles, const acmhip_packed_chunk *d_chunks, const uint8_t *d_blob,
		       const acmhip_blkhdr *d_hdr, int16_t *d_pcm, int16_t *d_sink, unsigned fmt, void *stream);
/* the byte-plane staged form: the same tile records as acm_tile2 except that idx_off is the pair-table entry of the row pair in front of the tile */
int acmk_tile2m_rows(uint32_t level);                           /* = acmk_tile2_rows, 0 if the level has no such build */
int acmk_tile2m_lead_in(uint32_t level);                        /* tiles of this build in front of a window into a stream (ACM_TILE_DISCARD records) */
int acmk_tile2m_run_waves(uint32_t level, int cus);             /* wavefronts that share a launch's table (the chunk kernel: one contiguous run each), else 0 */
int acmk_tile2m_stages(uint32_t level);                         /* stages of its first pass (3 or 4): the form keeps 2^stages columns of a residue class side by side */
int acmk_launch_tile2m(uint32_t level, int cus, const AcmTile2 *d_tiles, uint32_t ntiles, const uint8_t *d_mform, const acmhip_mform_pair *d_pairs,
		       const acmhip_blkhdr *d_hdr, int16_t *d_pcm, int16_t *d_sink, unsigned fmt, void *stream);
int acmk_launch_unpack(const AcmDevStream *d_streams, const uint32_t *d_list, uint32_t nlist, uint64_t max_elems,
		       const int16_t *d_idx, const acmhip_blkhdr *d_hdr, int32_t *d_x, uint32_t shift, void *stream);
int acmk_launch_prefix(const AcmDevStream *d_streams, const uint32_t *d_list, uint32_t nlist, uint64_t max_elems,
		       uint32_t level, const int16_t *d_idx, const acmhip_blkhdr *d_hdr, int32_t *d_y, void *stream);   /* levels 13-15: unpack + level-12 stages */
int acmk_plane_tile_rows(void);                                 /* tile rows (incl. 2 halo rows) of the level-12 plane kernel */
int acmk_plane_grid(int cus);                                   /* persistent workgroups of that kernel */
int acmk_launch_fused_plane(int cus, int carry, const AcmDevStream *d_streams, const AcmTile *d_tiles, uint32_t ntiles,
			    const int32_t *d_plane, int16_t *d_pcm, unsigned fmt, void *stream);
int acmk_launch_patch(const AcmDevPatch *d_patches, uint64_t n, int32_t *d_x, void *stream);
int acmk_launch_stage(const AcmDevStream *d_streams, const uint32_t *d_list, uint32_t nlist, uint64_t max_elems,
		      uint32_t level, uint32_t k, const int32_t *d_in, int32_t *d_out, uint32_t shift, void *stream);
int acmk_parse_supported(uint32_t level, uint32_t rows, uint64_t file_len, uint64_t blocks);
int acmk_launch_parse(const AcmParseJob *d_jobs, uint32_t njobs, const uint8_t *d_files, uint32_t *d_colpos, int16_t *d_idx,
		      acmhip_blkhdr *d_hdr, AcmParseResult *d_res, uint32_t *d_flags, uint64_t max_columns, void *stream);
/* the same for block range r of R: stream j's blocks [blocks * r / R, blocks * (r + 1) / R), resuming at the bit offset range
 * r - 1 left in d_res (ranges are launched in order on one stream; njobs <= ACM_PARSE_RANGE_MAX_STREAMS) */
#define ACM_PARSE_RANGE_MAX_STREAMS 32768
int acmk_launch_parse_range(const AcmParseJob *d_jobs, uint32_t njobs, const uint8_t *d_files, uint32_t *d_colpos, int16_t *d_idx,
			    acmhip_blkhdr *d_hdr, AcmParseResult *d_res, uint32_t *d_flags, uint64_t max_columns, uint32_t r, uint32_t R,
			    uint32_t stripes_up, void *stream);
/* the same with byte-plane staging for the jobs that ask for it (mf_rows != 0): d_mf = the byte-plane arena, d_pairs = its pair table,
 * d_blkoff = one word per block (indexed like d_hdr): where the row pair that holds the block's first row starts in its stream's region, in
 * 64-byte units (blocks of an odd height: every other one begins inside a pair).  Streams are walked
 * by the wave-per-stream kernel only (njobs <= ACM_PARSE_RANGE_MAX_STREAMS) */
int acmk_launch_parse_range_mf(const AcmParseJob *d_jobs, uint32_t njobs, const uint8_t *d_files, uint32_t *d_colpos, int16_t *d_idx,
			       acmhip_blkhdr *d_hdr, AcmParseResult *d_res, uint32_t *d_flags, uint64_t max_columns, uint32_t r, uint32_t R,
			       uint32_t stripes_up, uint8_t *d_mf, uint32_t *d_pairs, uint32_t *d_blkoff, void *stream);
/* Striped upload of a block-range batch: every file's arena slot (the file padded to 16 bytes + 16 zero bytes) is cut into R
 * stripes at acmk_stripe_bound(len, s, R); stripe s of all files travels as ONE transfer into a staging arena and a scatter
 * kernel puts the pieces in place, so the first ranges are walked, synthesised and read back while the later stripes are
 * still on their way up (PCIe runs both directions at once).  stripes_up (0 = the whole file is there) tells the walk of a
 * range how many stripes it may read: a stream that needs more stops with a status and goes to the host reader. */
uint32_t acmk_stripe_bound(uint32_t file_len, uint32_t s, uint32_t S);
/* first block of block range r of R (r == R: the end) of a stream of `blocks` blocks whose ranges are cut at multiples of `unit` blocks */
uint32_t acmk_range_bound(uint32_t blocks, uint32_t r, uint32_t R, uint32_t unit);
int acmk_launch_scatter_stripe(const AcmParseJob *d_jobs, uint32_t njobs, const uint64_t *d_stripe_at, const uint8_t *d_stage,
			       uint8_t *d_files, uint32_t s, uint32_t S, void *stream);
int acmk_launch_small(uint32_t level, const AcmDevStream *d_streams, const uint32_t *d_list, uint32_t nlist, uint64_t max_emit,
		      const int16_t *d_idx, const acmhip_blkhdr *d_hdr, int16_t *d_pcm, unsigned fmt, void *stream);
int acmk_launch_emit(const AcmDevStream *d_streams, const uint32_t *d_list, uint32_t nlist, uint64_t max_emit,
		     const int32_t *d_x, int16_t *d_pcm, unsigned fmt, void *stream);
#ifdef __cplusplus
}
#endif

#endif
