/*
 * acm_stream.cpp - the libacm.h API (drop-in for /root/reference/src/decode.c
 * :758-893 and src/util.c) on top of the host parser (acm_fill) and the device
 * synthesis (acm_hip.h), plus the whole-file staging entry points.
 *
 * acm_read() keeps the reference's call-by-call contract (decode.c:826-876)
 * but a "decoded block" is a slice of a read-ahead window: the host parses a
 * run of blocks into staged form, one launch synthesises the whole run on the
 * GPU, and the PCM comes back in one copy.  The synthesis history the
 * reference keeps in wrapbuf (decode.c:803-812) is not state here - it is
 * recomputed from the last two staged rows of the previous window ("carry").
 * Synthesis is lazy: decode-and-discard reads (dst == NULL, i.e. seeking,
 * util.c:243-251) only parse.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <atomic>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "acm_device.h"
#include "acm_fill.h"
#include "acm_hip.h"
#include "libacm.h"

#include <stddef.h>

/* the public struct is ABI (callers of the reference poke at it) */
static_assert(sizeof(ACMInfo) == 32, "ACMInfo layout");
static_assert(sizeof(acm_io_callbacks) == 32, "acm_io_callbacks layout");
static_assert(sizeof(ACMStream) == 176, "ACMStream layout");
static_assert(offsetof(ACMStream, total_values) == 32 && offsetof(ACMStream, io_arg) == 40 &&
	      offsetof(ACMStream, io) == 48 && offsetof(ACMStream, data_len) == 80 &&
	      offsetof(ACMStream, buf) == 88 && offsetof(ACMStream, buf_max) == 96 &&
	      offsetof(ACMStream, bit_data) == 112 && offsetof(ACMStream, buf_start_ofs) == 116 &&
	      offsetof(ACMStream, block_len) == 120 && offsetof(ACMStream, wrapbuf_len) == 124 &&
	      offsetof(ACMStream, block) == 128 && offsetof(ACMStream, midbuf) == 152 &&
	      offsetof(ACMStream, stream_pos) == 164 && offsetof(ACMStream, block_pos) == 168,
	      "ACMStream field offsets");

namespace {

using acmfill::kCleanEof;

constexpr uint64_t kWindowSamplesMax = 64u << 20;       /* read-ahead ceiling per window (buffers are sized by min(this, the stream): a
                                                           whole 40-Msample file fits, so that parsing can run ahead for as long
                                                           as a prewarmed device takes to come up) */
constexpr uint64_t kWindowSamplesUnknownLength = 4u << 20;      /* ... for a data source that does not say how long it is */
constexpr uint64_t kWindowSamplesFirst = 64u << 10;     /* first window: keep time-to-first-sample short */

/* ---- process-wide default device for the single-stream API ----
 * Opened on first use - or ahead of it, on a thread of its own, when the application says it is going to decode
 * (acmhip_prewarm, include/acm_hip.h): bringing the HIP runtime up takes ~0.2 s on these boxes, which is as long as the
 * host parser needs for 40 Msamples.  While the device is still coming up the stream keeps parsing ahead (fill_window). */
std::atomic<bool> g_seek_index{ true };        /* acmhip_set_seek_index (include/acm_hip.h) */
std::mutex g_dev_mutex;
std::condition_variable g_dev_cv;
acmhip_device *g_dev = nullptr;
int g_dev_state = 0;                    /* 0 untouched, 1 being opened, 2 settled (g_dev valid or NULL) */
bool g_dev_reported = false;
char g_dev_error[256];

void open_default_device()
{
	const char *env = getenv("ACM_HIP_DEVICE");
	const int ord = env ? atoi(env) : 0;
	acmhip_device *d = nullptr;
	const int rc = acmhip_device_open(ord, nullptr, &d);
	if (rc == ACMHIP_OK)
		(void)acmk_warmup(acmhip_device_stream(d));      /* loads the kernels' code object now, not inside the first decode */
	std::lock_guard<std::mutex> lock(g_dev_mutex);
	if (rc != ACMHIP_OK) {
		snprintf(g_dev_error, sizeof(g_dev_error), "%s", acmhip_last_error());
		d = nullptr;
	}
	g_dev = d;
	g_dev_state = 2;
	g_dev_cv.notify_all();
}

/* a prewarm thread still inside the HIP runtime when the process leaves main() must be waited for */
struct PrewarmJoiner {
	std::thread t;
	~PrewarmJoiner()
	{
		if (t.joinable())
			t.join();
	}
} g_prewarm;

void start_prewarm()
{
	std::lock_guard<std::mutex> lock(g_dev_mutex);
	if (g_dev_state != 0)
		return;
	g_dev_state = 1;
	g_prewarm.t = std::thread(open_default_device);
}

bool default_device_pending()
{
	std::lock_guard<std::mutex> lock(g_dev_mutex);
	return g_dev_state == 1;
}

acmhip_device *default_device(bool quiet = false)
{
	std::unique_lock<std::mutex> lock(g_dev_mutex);
	if (g_dev_state == 0) {
		g_dev_state = 1;
		lock.unlock();
		open_default_device();
		lock.lock();
	}
	g_dev_cv.wait(lock, []() { return g_dev_state == 2; });
	if (!g_dev && !g_dev_reported && !quiet) {
		g_dev_reported = true;
		fprintf(stderr, "libacm_hip: cannot decode: %s\n", g_dev_error);
	}
	return g_dev;
}

struct HipStream {
	ACMStream pub;                          /* must stay first: ACMStream* == HipStream* */
	acmfill::TableHistory tab;              /* stale-table history as the PARSER has seen it (it reads ahead of the caller) */
	acmfill::TableHistory tab_served;       /* ... as the reference's table would be: blocks handed to the caller so far.
	                                           A backward seek continues from this one (the reference never clears its
	                                           table, decode.c:809-810, and has decoded nothing beyond the served block) */

	uint32_t carry_max = 1;                 /* staged blocks kept in front of a window for its 2-row halo */
	uint32_t carry = 0;                     /* ... how many are there now */
	uint32_t win_cap = 0;                   /* window capacity, blocks */
	uint32_t win_blocks = 0;                /* blocks parsed into the current window */
	uint32_t win_next = 0;                  /* next window block to hand out */
	uint32_t cur = 0;                       /* window block being served while block_ready */
	uint32_t grow = 0;                      /* size of the next window, blocks */
	int pending = 0;                        /* status that stopped the parser behind the window (0 = none) */
	unsigned tell_now = 0;                  /* what acm_raw_tell() reports */

	int16_t *h_idx = nullptr;
	acmhip_blkhdr *h_hdr = nullptr;
	int16_t *h_pcm = nullptr;
	std::vector<unsigned> tell_after;
	std::vector<acmhip_patch> patches;

	/* block index, built while parsing: where every block seen so far starts in the file
	 * and what its header was.  A backward seek re-enters the stream just in front of its target instead of
	 * re-parsing from the first block (util.c:219-242 has no index: block sizes are data dependent). */
	std::vector<uint64_t> mark_bit;         /* true file bit offset of block b's first bit */
	std::vector<acmhip_blkhdr> hdr_log;     /* (val, pwr) of block b: all the stale-table history needs */
	uint64_t next_block_no = 0;             /* number of the block the parser reads next */
	int64_t ofs_delta = 0;                  /* true file offset of the refill buffer minus buf_start_ofs
	                                           (after a seek the reference counts from 14 even behind a WAVC prefix) */

	acmhip_device *dev = nullptr;
	int16_t *d_idx = nullptr;
	acmhip_blkhdr *d_hdr = nullptr;
	int16_t *d_pcm = nullptr;
	acmhip_plan *plan = nullptr;
	acmhip_stream_desc plan_desc{};
	bool plan_valid = false;
	bool on_host = false;                   /* this stream's windows are synthesised by acm_host_synth.cpp */
	bool pcm_valid = false;
	unsigned pcm_fmt = 0;
};

inline HipStream *priv(ACMStream *a) { return reinterpret_cast<HipStream *>(a); }

void drop_window(HipStream *hs)
{
	hs->carry = 0;
	hs->win_blocks = 0;
	hs->win_next = 0;
	hs->pending = 0;
	hs->pcm_valid = false;
	hs->patches.clear();
}

bool alloc_window(HipStream *hs)
{
	if (hs->h_idx)
		return true;
	const ACMStream *a = &hs->pub;
	const uint64_t bl = a->block_len;
	const uint64_t total_blocks = ((uint64_t)a->total_values + bl - 1) / bl;
	uint64_t cap = std::max<uint64_t>(1, kWindowSamplesMax / bl);
	cap = std::min(cap, std::max<uint64_t>(1, total_blocks));
	/* the header's total_values is a promise, not a fact: the buffers - host and device - are sized by what the
	 * data source can hold when it says how long it is (a block costs at least its 20-bit header and a 5-bit filler code per
	 * column, decode.c:491-502, 586-589; the reader appends one zero byte, :57-61), and by the old 4-Msample window when it does not */
	if (a->data_len > 0) {
		const uint64_t bits = (uint64_t)a->data_len * 8 + 8;
		cap = std::min(cap, bits / (20 + 5 * (uint64_t)a->info.acm_cols) + 1);
	} else {
		cap = std::min(cap, std::max<uint64_t>(1, kWindowSamplesUnknownLength / bl));
	}
	hs->win_cap = (uint32_t)cap;
	hs->carry_max = a->info.acm_rows >= 2 ? 1 : 2;
	hs->grow = (uint32_t)std::min<uint64_t>(cap, std::max<uint64_t>(1, kWindowSamplesFirst / bl));
	const size_t blocks = (size_t)cap + hs->carry_max;
	hs->h_idx = (int16_t *)malloc(blocks * bl * sizeof(int16_t) + 16);
	hs->h_hdr = (acmhip_blkhdr *)malloc(blocks * sizeof(acmhip_blkhdr));
	hs->h_pcm = (int16_t *)malloc((size_t)cap * bl * sizeof(int16_t) + 16);
	hs->tell_after.resize(cap);
	return hs->h_idx && hs->h_hdr && hs->h_pcm;
}

/* parse the next run of blocks behind the current window; 1, kCleanEof or ACM_ERR_* */
int fill_window(HipStream *hs)
{
	ACMStream *a = &hs->pub;
	if (!alloc_window(hs))
		return ACM_ERR_OTHER;
	const size_t bl = a->block_len;

	/* keep the tail of what we had as halo for the new window */
	const uint32_t have = hs->carry + hs->win_blocks;
	const uint32_t keep = std::min(hs->carry_max, have);
	const uint32_t shift = have - keep;
	if (shift) {
		memmove(hs->h_idx, hs->h_idx + (size_t)shift * bl, (size_t)keep * bl * sizeof(int16_t));
		memmove(hs->h_hdr, hs->h_hdr + shift, keep * sizeof(acmhip_blkhdr));
		const uint64_t cut = (uint64_t)shift * bl;
		size_t w = 0;
		for (const acmhip_patch &p : hs->patches)
			if (p.sample >= cut) {
				hs->patches[w] = p;
				hs->patches[w++].sample -= cut;
			}
		hs->patches.resize(w);
	}
	hs->carry = keep;
	hs->win_blocks = 0;
	hs->win_next = 0;
	hs->pcm_valid = false;

	const uint64_t remaining = a->total_values - a->stream_pos;     /* > 0, checked by the caller */
	const uint64_t need = (remaining + bl - 1) / bl;
	const uint32_t want = (uint32_t)std::min<uint64_t>(std::min(hs->win_cap, hs->grow), need);

	/* while a prewarmed device is still coming up there is nothing to hand the window to: keep parsing */
	const uint32_t hard = (uint32_t)std::min<uint64_t>(hs->win_cap, need);
	acmfill::PatchSink sink{ &hs->patches, 0, 0, 0 };
	for (uint32_t i = 0; i < hard; i++) {
		if (i >= want && (hs->dev || !default_device_pending()))
			break;
		const size_t slot = hs->carry + i;
		sink.base_sample = (uint64_t)slot * bl;
		if (hs->next_block_no == hs->mark_bit.size())
			hs->mark_bit.push_back(8ull * (uint64_t)((int64_t)a->buf_start_ofs + hs->ofs_delta + a->buf_pos) - a->bit_avail);
		const int rc = acmfill::parse_block(a, &hs->tab, hs->h_idx + slot * bl, hs->h_hdr + slot, &sink);
		if (rc != 1) {
			hs->pending = rc;
			break;
		}
		if (hs->next_block_no == hs->hdr_log.size())
			hs->hdr_log.push_back(hs->h_hdr[slot]);
		hs->next_block_no++;
		hs->tell_after[i] = acmfill::raw_position(a);
		hs->win_blocks++;
	}
	if (hs->win_blocks == 0) {
		const int rc = hs->pending;
		hs->pending = 0;
		return rc;
	}
	hs->grow = (uint32_t)std::min<uint64_t>(hs->win_cap, (uint64_t)hs->grow * 2);
	return 1;
}

/* is a device handle open in this process already (without bringing the runtime up to find out)? */
bool default_device_open()
{
	std::lock_guard<std::mutex> lock(g_dev_mutex);
	return g_dev_state == 2 && g_dev != nullptr;
}

/* the current window on the host (acm_host_synth.cpp): same staged arrays, same descriptor, same bytes out */
int synth_window_host(HipStream *hs, unsigned fmt)
{
	ACMStream *a = &hs->pub;
	acmhip_stream_desc d{};
	d.level = a->info.acm_level;
	d.rows = a->info.acm_rows;
	d.nrows = (hs->carry + hs->win_blocks) * a->info.acm_rows;
	d.row_begin = hs->carry * a->info.acm_rows;
	d.n_emit = (uint64_t)hs->win_blocks * a->block_len;
	if (acmhip_host_synth(&d, hs->h_idx, hs->h_hdr, hs->patches.data(), hs->patches.size(), fmt, hs->h_pcm) != ACMHIP_OK)
		return ACM_ERR_OTHER;
	hs->pcm_valid = true;
	hs->pcm_fmt = fmt;
	return 0;
}

/* synthesise the current window in sample layout `fmt`; 0 or ACM_ERR_OTHER.  On the GPU - unless there is none, or the stream is short and
 * no device handle is open yet (include/acm_hip.h, acmhip_host_synth_limit): a stream stays on the side its first window took */
int synth_window(HipStream *hs, unsigned fmt)
{
	ACMStream *a = &hs->pub;
	const size_t bl = a->block_len;
	if (hs->on_host)
		return synth_window_host(hs, fmt);
	if (!hs->dev) {
		if ((uint64_t)a->total_values < acmhip_host_synth_limit() && !default_device_open() && !default_device_pending()) {
			hs->on_host = true;
			return synth_window_host(hs, fmt);
		}
		hs->dev = default_device(/*quiet=*/true);
		if (!hs->dev) {
			hs->on_host = true;             /* no usable device on this box: the reference decodes anywhere, so does this */
			return synth_window_host(hs, fmt);
		}
	}
	if (!hs->d_idx) {
		const size_t blocks = (size_t)hs->win_cap + hs->carry_max;
		if (acmhip_malloc(hs->dev, blocks * bl * sizeof(int16_t) + 16, (void **)&hs->d_idx) ||
		    acmhip_malloc(hs->dev, blocks * sizeof(acmhip_blkhdr), (void **)&hs->d_hdr) ||
		    acmhip_malloc(hs->dev, (size_t)hs->win_cap * bl * sizeof(int16_t) + 16, (void **)&hs->d_pcm)) {
			fprintf(stderr, "libacm_hip: %s\n", acmhip_last_error());
			return ACM_ERR_OTHER;
		}
	}
	const uint32_t nblk = hs->carry + hs->win_blocks;
	acmhip_stream_desc d{};
	d.level = a->info.acm_level;
	d.rows = a->info.acm_rows;
	d.nrows = nblk * a->info.acm_rows;
	d.row_begin = hs->carry * a->info.acm_rows;
	d.n_emit = (uint64_t)hs->win_blocks * bl;

	int rc = acmhip_upload(hs->dev, hs->d_idx, hs->h_idx, (size_t)nblk * bl * sizeof(int16_t));
	if (!rc)
		rc = acmhip_upload(hs->dev, hs->d_hdr, hs->h_hdr, nblk * sizeof(acmhip_blkhdr));
	if (!rc && !(hs->plan_valid && hs->patches.empty() && !memcmp(&d, &hs->plan_desc, sizeof(d)))) {
		acmhip_plan_destroy(hs->plan);
		hs->plan = nullptr;
		hs->plan_valid = false;
		rc = acmhip_plan_create(hs->dev, &d, 1, hs->patches.data(), hs->patches.size(), ACMHIP_PLAN_AUTO, &hs->plan);
		if (!rc) {
			hs->plan_desc = d;
			hs->plan_valid = hs->patches.empty();
		}
	}
	if (!rc)
		rc = acmhip_plan_launch(hs->plan, hs->d_idx, hs->d_hdr, hs->d_pcm, fmt);
	if (!rc)
		rc = acmhip_download(hs->dev, hs->h_pcm, hs->d_pcm, (size_t)d.n_emit * sizeof(int16_t));
	if (!rc)
		rc = acmhip_device_sync(hs->dev);
	if (rc) {
		fprintf(stderr, "libacm_hip: %s\n", acmhip_last_error());
		return ACM_ERR_OTHER;
	}
	hs->pcm_valid = true;
	hs->pcm_fmt = fmt;
	return 0;
}

/* decode_block() as seen from acm_read (decode.c:580-611): make the next block current */
int next_block(HipStream *hs)
{
	ACMStream *a = &hs->pub;
	a->block_ready = 0;
	a->block_pos = 0;
	if (hs->win_next >= hs->win_blocks) {
		if (hs->pending) {                      /* the parser already hit the end / an error right here */
			const int rc = hs->pending;
			hs->pending = 0;
			hs->tell_now = acmfill::raw_position(a);
			return rc;
		}
		const int rc = fill_window(hs);
		if (rc != 1) {
			hs->tell_now = acmfill::raw_position(a);
			return rc;
		}
	}
	hs->cur = hs->win_next++;
	hs->tell_now = hs->tell_after[hs->cur];
	hs->tab_served.note_block(hs->h_hdr[hs->carry + hs->cur].pwr, hs->h_hdr[hs->carry + hs->cur].val);
	a->block_ready = 1;
	return 1;
}

/* ---- in-memory data source for the whole-file staging calls ---- */
struct MemSource {
	const uint8_t *p;
	size_t len, pos;
};

int mem_read(void *ptr, int size, int n, void *arg)
{
	MemSource *m = (MemSource *)arg;
	size_t want = (size_t)size * (size_t)n;
	if (want > m->len - m->pos)
		want = m->len - m->pos;
	memcpy(ptr, m->p + m->pos, want);
	m->pos += want;
	return size ? (int)(want / (size_t)size) : 0;
}

/* header parse + channel forcing + derived sizes, shared by open and staging (decode.c:783-804) */
int open_common(ACMStream *a, int force_chans)
{
	if (acmfill::read_headers(a) < 0)
		return ACM_ERR_NOT_ACM;                 /* every header-stage failure reads as "not ACM" (:783-785) */
	if (force_chans > 0)
		a->info.channels = (unsigned)force_chans;
	else if (force_chans == -1 && !a->wavc_file && a->info.channels < 2)
		a->info.channels = 2;
	a->info.acm_cols = 1u << a->info.acm_level;
	a->wrapbuf_len = 2 * a->info.acm_cols - 2;
	a->block_len = a->info.acm_rows * a->info.acm_cols;
	return ACM_OK;
}

void fill_stage_info(const ACMStream *a, acm_stage_info *info)
{
	info->level = a->info.acm_level;
	info->rows = a->info.acm_rows;
	info->cols = a->info.acm_cols;
	info->channels = a->info.channels;
	info->hdr_channels = a->info.acm_channels;
	info->rate = a->info.rate;
	info->total_values = a->total_values;
	info->wavc = a->wavc_file;
	info->header_bytes = a->wavc_file ? 42 : 14;
}

} // namespace

/* ======================================================================== */
/* core API                                                                  */
/* ======================================================================== */

extern "C" void acmhip_set_seek_index(int on)
{
	g_seek_index.store(on != 0);
}

extern "C" void acmhip_prewarm(void)
{
	start_prewarm();
}

extern "C" int acm_open_decoder(ACMStream **res, void *arg, acm_io_callbacks io_cb, int force_chans)
{
	HipStream *hs = new (std::nothrow) HipStream();
	if (!hs)
		return ACM_ERR_OTHER;
	ACMStream *a = &hs->pub;
	memset(a, 0, sizeof(*a));
	hs->tab.reset();
	hs->tab_served.reset();
	a->io_arg = arg;
	a->io = io_cb;
	a->data_len = a->io.get_length_func ? (unsigned)a->io.get_length_func(a->io_arg) : 0;
	a->buf_max = acmfill::kChunkBytes;
	a->buf = (unsigned char *)malloc(a->buf_max);
	int err = ACM_ERR_OTHER;
	if (a->buf) {
		err = open_common(a, force_chans);
		if (err == ACM_OK) {
			hs->tell_now = acmfill::raw_position(a);
			*res = a;
			return ACM_OK;
		}
	}
	/* the caller keeps ownership of its handle on failure (:817-823) */
	memset(&a->io, 0, sizeof(a->io));
	a->io_arg = NULL;
	acm_close(a);
	return err;
}

extern "C" int acm_read(ACMStream *acm, void *dst, unsigned numbytes, int bigendianp, int wordlen, int sgned)
{
	HipStream *hs = priv(acm);
	if (wordlen != 2)
		return ACM_ERR_BADFMT;
	int numwords = (int)(numbytes / 2);
	if (acm->stream_pos >= acm->total_values)
		return 0;
	if (!acm->block_ready) {
		const int rc = next_block(hs);
		if (rc == kCleanEof)
			return 0;
		if (rc < 0)
			return rc;
	}
	/* how much of the current block may go out (decode.c:849-857) */
	const int avail = (int)(acm->block_len - acm->block_pos);
	if (avail < numwords)
		numwords = avail;
	if (acm->stream_pos + (unsigned)numwords > acm->total_values)
		numwords = (int)(acm->total_values - acm->stream_pos);
	if (acm->info.channels > 1)
		numwords -= numwords % (int)acm->info.channels;

	if (dst != NULL && numwords > 0) {
		const unsigned fmt = (bigendianp ? ACMHIP_FMT_S16BE : 0u) | (sgned ? 0u : ACMHIP_FMT_U16LE);
		if (!hs->pcm_valid || hs->pcm_fmt != fmt) {
			const int rc = synth_window(hs, fmt);
			if (rc < 0)
				return rc;
		}
		memcpy(dst, hs->h_pcm + (size_t)hs->cur * acm->block_len + acm->block_pos, (size_t)numwords * 2);
	}
	acm->stream_pos += (unsigned)numwords;
	acm->block_pos += (unsigned)numwords;
	if (acm->block_pos == acm->block_len)
		acm->block_ready = 0;
	return numwords * 2;
}

extern "C" void acm_close(ACMStream *acm)
{
	if (acm == NULL)
		return;
	HipStream *hs = priv(acm);
	if (acm->io.close_func)
		acm->io.close_func(acm->io_arg);
	if (hs->dev) {
		acmhip_plan_destroy(hs->plan);
		acmhip_free(hs->dev, hs->d_idx);
		acmhip_free(hs->dev, hs->d_hdr);
		acmhip_free(hs->dev, hs->d_pcm);
	}
	free(hs->h_idx);
	free(hs->h_hdr);
	free(hs->h_pcm);
	free(acm->buf);
	delete hs;
}

/* ======================================================================== */
/* util API (reference src/util.c)                                           */
/* ======================================================================== */

namespace {

int file_read(void *ptr, int size, int n, void *arg) { return (int)fread(ptr, (size_t)size, (size_t)n, (FILE *)arg); }
int file_close(void *arg) { return fclose((FILE *)arg); }
int file_seek(void *arg, int offset, int whence) { return fseek((FILE *)arg, offset, whence); }

int file_length(void *arg)
{
	FILE *f = (FILE *)arg;
	const long here = ftell(f);
	long len = -1;
	if (here < 0)
		return -1;
	if (fseek(f, 0, SEEK_END) >= 0) {
		len = ftell(f);
		fseek(f, here, SEEK_SET);
	}
	return (int)len;
}

unsigned words_to_ms(const ACMStream *a, unsigned long long pcm) { return (unsigned)(pcm * 1000 / a->info.rate); }

} // namespace

extern "C" const char *acm_strerror(int err)
{
	/* same texts as util.c:34-44, including its spelling of "Unexcpected" */
	static const char *const msg[] = {
		"No error", "ACM error", "Cannot open file", "Not an ACM file", "Read error",
		"Bad format", "Corrupt file", "Unexcpected EOF", "Stream not seekable"
	};
	const int n = (int)(sizeof(msg) / sizeof(msg[0]));
	if (err > 0 || err <= -n)
		return "Unknown error";
	return msg[-err];
}

extern "C" int acm_open_file(ACMStream **res, const char *filename, int force_chans)
{
	FILE *f = fopen(filename, "rb");
	if (!f)
		return ACM_ERR_OPEN;
	acm_io_callbacks io;
	memset(&io, 0, sizeof(io));
	io.read_func = file_read;
	io.seek_func = file_seek;
	io.close_func = file_close;
	io.get_length_func = file_length;
	ACMStream *a = NULL;
	const int err = acm_open_decoder(&a, f, io, force_chans);
	if (err < 0) {
		fclose(f);
		return err;
	}
	*res = a;
	return ACM_OK;
}

extern "C" const ACMInfo *acm_info(ACMStream *acm) { return &acm->info; }
extern "C" unsigned acm_rate(ACMStream *acm) { return acm->info.rate; }
extern "C" unsigned acm_channels(ACMStream *acm) { return acm->info.channels; }
extern "C" int acm_seekable(ACMStream *acm) { return acm->data_len > 0; }
extern "C" unsigned acm_pcm_tell(ACMStream *acm) { return acm->stream_pos / acm->info.channels; }
extern "C" unsigned acm_pcm_total(ACMStream *acm) { return acm->total_values / acm->info.channels; }
extern "C" unsigned acm_time_tell(ACMStream *acm) { return words_to_ms(acm, acm_pcm_tell(acm)); }
extern "C" unsigned acm_time_total(ACMStream *acm) { return words_to_ms(acm, acm_pcm_total(acm)); }
extern "C" unsigned acm_raw_total(ACMStream *acm) { return acm->data_len; }

/* Position of the host parser when the block now being served was finished -
 * i.e. what the reference reports (util.c:192-195) - not how far the
 * read-ahead has run. */
extern "C" unsigned acm_raw_tell(ACMStream *acm) { return priv(acm)->tell_now; }

extern "C" unsigned acm_bitrate(ACMStream *acm)
{
	if (acm_raw_total(acm) == 0)
		return 13000;                           /* util.c:161-162 */
	const unsigned long long ms = acm_time_total(acm);
	if (ms == 0)
		return 0;
	const unsigned long long bits = (unsigned)(8u * acm_raw_total(acm));   /* 32-bit product as in util.c:166 */
	return (unsigned)(1000 * bits / ms);
}

extern "C" int acm_read_loop(ACMStream *acm, void *dst, unsigned bytes, int bigendianp, int wordlen, int sgned)
{
	unsigned char *out = (unsigned char *)dst;
	int got = 0;
	while (bytes > 0) {
		const int rc = acm_read(acm, out, bytes, bigendianp, wordlen, sgned);
		if (rc > 0) {
			if (out)
				out += rc;
			got += rc;
			bytes -= (unsigned)rc;
			continue;
		}
		if (rc < 0 && got == 0)
			return rc;                      /* an error after some output is swallowed (util.c:271-273) */
		break;
	}
	return got;
}

extern "C" int acm_seek_pcm(ACMStream *acm, unsigned pcm_pos)
{
	HipStream *hs = priv(acm);
	const unsigned word_pos = pcm_pos * acm->info.channels;

	if (word_pos < acm->stream_pos) {
		/* the reference rewinds and re-parses from the first block (util.c:219-242).  Same observable result,
		 * without the re-parse: re-enter at the remembered start of the block `halo` blocks in front of the
		 * target (those blocks are the synthesis history of the target block; they are parsed, their PCM is
		 * never delivered), on the 4-byte grid the reference's reader would be on after its rewind, and bring
		 * the stale-table history to where re-parsing the skipped blocks would have left it. */
		if (acm->io.seek_func == NULL)
			return ACM_ERR_NOT_SEEKABLE;
		const unsigned start = 14 + (acm->wavc_file ? 28 : 0);
		const uint64_t bl = acm->block_len;
		const uint64_t halo = acm->info.acm_rows >= 2 ? 1 : 2;
		const uint64_t target = word_pos / bl;
		const uint64_t enter = target > halo ? target - halo : 0;
		const bool indexed = enter >= 1 && enter < hs->mark_bit.size() && enter <= hs->hdr_log.size() &&
				     bl % acm->info.channels == 0 && g_seek_index.load();
		uint64_t at = start;
		if (indexed) {
			at = hs->mark_bit[enter] >> 3;
			at -= (at - start) & 3u;
		}
		if (acm->io.seek_func(acm->io_arg, (int)at, SEEK_SET) < 0)
			return ACM_ERR_NOT_SEEKABLE;
		acmfill::reset_reader(acm);
		hs->ofs_delta = (int64_t)start - 14;
		acm->stream_pos = 0;
		acm->block_pos = 0;
		acm->block_ready = 0;
		drop_window(hs);                        /* history = zeros again (util.c:241) */
		hs->next_block_no = 0;
		hs->tab = hs->tab_served;               /* forget what only the read-ahead had seen */
		if (indexed) {
			acm->buf_start_ofs = 14 + (unsigned)(at - start);
			if (acmfill::skip_bits(acm, (unsigned)(hs->mark_bit[enter] - 8 * at)) < 0)
				return ACM_ERR_OTHER;           /* the data source no longer holds what was indexed */
			for (uint64_t b = 0; b < enter; b++) {          /* the blocks the reference would decode again on its way */
				hs->tab.note_block(hs->hdr_log[b].pwr, hs->hdr_log[b].val);
				hs->tab_served.note_block(hs->hdr_log[b].pwr, hs->hdr_log[b].val);
			}
			acm->stream_pos = (unsigned)(enter * bl);
			hs->next_block_no = enter;
		}
		hs->tell_now = acmfill::raw_position(acm);
	}
	while (acm->stream_pos < word_pos) {
		unsigned step = 2048;
		if (acm->stream_pos + step > word_pos)
			step = word_pos - acm->stream_pos;
		if (acm_read(acm, NULL, step * 2, 0, 2, 1) < 1)
			break;
	}
	return (int)(acm->stream_pos / acm->info.channels);
}

extern "C" int acm_seek_time(ACMStream *acm, unsigned time_ms)
{
	const unsigned long long pcm = (unsigned long long)time_ms * acm->info.rate / 1000;
	const int res = acm_seek_pcm(acm, (unsigned)pcm);
	if (res <= 0)
		return res;
	return (int)words_to_ms(acm, (unsigned)res);
}

/* ======================================================================== */
/* whole-file staging (include/acm_hip.h)                                    */
/* ======================================================================== */

namespace {

struct StageCtx {
	ACMStream a;
	MemSource src;
	acmfill::TableHistory tab;
	StageCtx() { memset(&a, 0, sizeof(a)); }
	~StageCtx() { free(a.buf); }
	int open(const uint8_t *data, size_t len, int force_chans)
	{
		src = MemSource{ data, len, 0 };
		tab.reset();
		a.io.read_func = mem_read;
		a.io_arg = &src;
		a.data_len = (unsigned)len;
		a.buf_max = acmfill::kChunkBytes;
		a.buf = (unsigned char *)malloc(a.buf_max);
		if (!a.buf)
			return ACM_ERR_OTHER;
		return open_common(&a, force_chans);
	}
};

} // namespace

extern "C" int acm_stage_probe(const uint8_t *data, size_t len, int force_chans, acm_stage_info *info)
{
	if (!data || !info)
		return ACMHIP_ERR_ARG;
	memset(info, 0, sizeof(*info));
	StageCtx c;
	const int rc = c.open(data, len, force_chans);
	if (rc < 0)
		return rc;
	fill_stage_info(&c.a, info);
	return ACM_OK;
}

extern "C" int acm_stage_file(const uint8_t *data, size_t len, int force_chans,
			      int16_t *idx, acmhip_blkhdr *hdr, size_t max_blocks,
			      acmhip_patch *patches, size_t max_patches, acm_stage_info *info)
{
	if (!data || !info || (max_blocks && (!idx || !hdr)) || (max_patches && !patches))
		return ACMHIP_ERR_ARG;
	memset(info, 0, sizeof(*info));
	StageCtx c;
	int rc = c.open(data, len, force_chans);
	if (rc < 0)
		return rc;
	fill_stage_info(&c.a, info);

	const size_t bl = c.a.block_len;
	const uint64_t need = ((uint64_t)c.a.total_values + bl - 1) / bl;
	const uint64_t want = std::min<uint64_t>(need, max_blocks);
	std::vector<acmhip_patch> found;
	acmfill::PatchSink sink{ &found, 0, 0, 0 };
	uint64_t b = 0;
	int status = 0;
	for (; b < want; b++) {
		sink.base_sample = b * bl;
		rc = acmfill::parse_block(&c.a, &c.tab, idx + b * bl, hdr + b, &sink);
		if (rc != 1) {
			status = (rc == kCleanEof) ? 0 : rc;
			break;
		}
	}
	info->blocks = (uint32_t)b;
	info->end_status = status;
	info->npatches = found.size();
	const size_t ncopy = std::min(found.size(), max_patches);
	if (ncopy)
		memcpy(patches, found.data(), ncopy * sizeof(acmhip_patch));
	return ACM_OK;
}

/*
 * The same with the byte-plane form written while the parsed block is still in the cache: a block is parsed into a buffer of its own
 * (16 KB at level 9: the first-level cache, where the column scatter of the parser costs nothing), its row pairs go to the byte-plane
 * writer from there, and only the rows the int16 kernels still read - from two rows in front of the ragged tail on - are copied to
 * idx.  Against acm_stage_file + acmhip_mform_rows this drops the 2 B per sample written to and read back from the int16 arena.
 * *mf_rows = rows [0, *mf_rows) are in the form (whole tiles of the lean kernel); 0: the stream has none (a level without the form,
 * H1 patches, an index beyond the form's range, a file that ends early) and idx holds every row as acm_stage_file
 * leaves it - except that with patches (info->npatches != 0) the caller stages once more with room for them.
 */
#include "acm_device.h"
#include "acm_mform.h"

extern "C" int acm_stage_file_mform(const uint8_t *data, size_t len, int force_chans, int16_t *idx, acmhip_blkhdr *hdr, size_t max_blocks,
				    acm_stage_info *info, uint8_t *mf_out, uint64_t mf_base, acmhip_mform_pair *pairs, uint64_t *mf_rows,
				    uint64_t *mf_bytes)
{
	if (!data || !info || !mf_rows || !mf_bytes || (max_blocks && (!idx || !hdr)))
		return ACMHIP_ERR_ARG;
	*mf_rows = *mf_bytes = 0;
	auto plain = [&]() { return acm_stage_file(data, len, force_chans, idx, hdr, max_blocks, nullptr, 0, info); };
	memset(info, 0, sizeof(*info));
	StageCtx c;
	int rc = c.open(data, len, force_chans);
	if (rc < 0)
		return rc;
	fill_stage_info(&c.a, info);
	const uint32_t level = info->level, rows = info->rows;
	const int T2 = acmk_tile2_rows(level), TM = acmhip_mform_tile_rows(level);
	/* (levels 13 / 14: whether a plan takes such a stream's form is known only from the whole plan - acmhip_plan_form_rows - so its
	 * int16 rows may all be needed: the plain way) */
	if (!mf_out || !pairs || T2 <= 0 || TM <= 0 || T2 % TM || level > ACM_K1_MAX_LEVEL)
		return plain();
	const size_t bl = c.a.block_len, cols = (size_t)1 << level;
	const uint64_t need = ((uint64_t)c.a.total_values + bl - 1) / bl;
	const uint64_t want = std::min<uint64_t>(need, max_blocks);
	/* what a complete file delivers (decode.c:853-857: whole blocks, the last one cut at total_values, rounded to whole frames) */
	auto deliverable = [&](uint64_t blocks) {
		uint64_t pos = 0;
		for (uint64_t b = 0; b < blocks && pos < c.a.total_values; b++) {
			uint64_t take = std::min<uint64_t>(bl, c.a.total_values - pos);
			if (info->channels > 1)
				take -= take % info->channels;
			pos += take;
			if (take != bl)
				break;
		}
		return pos;
	};
	const uint64_t rows2 = std::min<uint64_t>(want * rows, deliverable(want) >> level) / (uint64_t)T2 * (uint64_t)T2;
	if (rows2 == 0)
		return plain();
	const uint64_t tail_from = rows2 >= 2 ? rows2 - 2 : 0;
	std::vector<int16_t> block(bl), straddle((rows & 1) ? 2 * cols : 0);
	AcmMformWriter w;
	if (acm_mform_begin(&w, level, mf_out, mf_base, pairs) != ACMHIP_OK)
		return plain();
	std::vector<acmhip_patch> found;
	acmfill::PatchSink sink{ &found, 0, 0, 0 };
	uint64_t b = 0;
	int status = 0;
	for (; b < want; b++) {
		sink.base_sample = b * bl;
		rc = acmfill::parse_block(&c.a, &c.tab, block.data(), hdr + b, &sink);
		if (rc != 1) {
			status = (rc == kCleanEof) ? 0 : rc;
			break;
		}
		if (!found.empty())
			return plain();                 /* H1: the stream keeps the int16 form (and the caller stages again, for the patches) */
		const uint64_t r0 = b * rows;
		/* row pairs count from the stream's row 0: with an odd acm_rows every other block starts on the second row of a pair, whose
		 * first row is the last one of the block before (kept in `straddle`) */
		for (uint32_t r = 0; r < rows && r0 + r < rows2;) {
			const int16_t *two = block.data() + (size_t)r * cols;
			if ((r0 + r) & 1) {
				memcpy(straddle.data() + cols, two, cols * sizeof(int16_t));
				two = straddle.data();
				r += 1;
			} else if (r + 1 < rows) {
				r += 2;
			} else {
				if (straddle.empty())
					straddle.resize(2 * cols);
				memcpy(straddle.data(), two, cols * sizeof(int16_t));
				break;
			}
			if (acm_mform_put_pair(&w, two) != ACMHIP_OK)
				return plain();         /* an index beyond the form's range */
		}
		if (r0 + rows > tail_from) {
			const uint32_t from = r0 >= tail_from ? 0u : (uint32_t)(tail_from - r0);
			memcpy(idx + (r0 + from) * cols, block.data() + (size_t)from * cols, (size_t)(rows - from) * cols * sizeof(int16_t));
		}
	}
	if (b != want || deliverable(b) != deliverable(want))
		return plain();                         /* the file ends early: fewer whole tiles than its header promised - the plain way */
	info->blocks = (uint32_t)b;
	info->end_status = status;
	info->npatches = 0;
	*mf_rows = rows2;
	*mf_bytes = acm_mform_end(&w);
	return ACM_OK;
}
