/*
 * acm_batch.cpp - batch-of-files front end on one device (include/acm_hip.h).
 *
 * No counterpart in the reference (it decodes one stream at a time on one
 * thread); this is the piece BASELINE.json's north_star adds.
 * Independent streams are bit-parsed by a pool of host threads straight into one
 * pinned staging arena, in arena order.  The arena is cut into chunks of whole
 * streams; as soon as a chunk is parsed it goes H2D, is synthesised by one
 * acmhip_plan_launch and read back on a second HIP stream, while the pool parses
 * the next chunks and then hands finished PCM out to the callers' buffers.  So
 * parsing, H2D, synthesis, D2H and the final copies all overlap.
 */
#include <hip/hip_runtime.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "acm_device.h"
#include "acm_hip.h"
#include "libacm.h"

namespace {

using clk = std::chrono::steady_clock;
inline double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }

/* How many words a caller looping over acm_read_loop() (acmtool.c:274-291)
 * gets out of `blocks` decodable blocks: blocks are drained whole except where
 * the per-call rounding to a multiple of `channels` (decode.c:856-857) or the
 * total_values cut (decode.c:853-854) stops the stream for good. */
uint64_t deliverable_words(uint64_t total_values, uint64_t block_len, unsigned channels, uint64_t blocks)
{
	uint64_t pos = 0;
	for (uint64_t b = 0; b < blocks && pos < total_values; b++) {
		uint64_t take = std::min(block_len, total_values - pos);
		if (channels > 1)
			take -= take % channels;
		pos += take;
		if (take != block_len)
			break;
	}
	return pos;
}

/* Default size of the parser pool.  Streams are independent and the parser is compute bound, so more threads
 * help up to the core count; past ~64 the returns vanish (and boxes with a cgroup CPU quota time-slice the
 * surplus), measured with profiles/e2e_probe.py. */
int default_threads()
{
	int n = (int)std::max(1u, std::thread::hardware_concurrency());
	cpu_set_t set;
	if (sched_getaffinity(0, sizeof(set), &set) == 0)
		n = std::min(n, std::max(1, CPU_COUNT(&set)));
	/* a cgroup-v2 CPU quota is invisible to the two calls above; threads beyond it only time-slice */
	if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
		long long quota = 0, period = 0;
		if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
			n = std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
		fclose(f);
	}
	return std::min(n, 64);
}

/* A fixed set of worker threads that lives for one acm_batch_decode call.  run() is a blocking parallel-for
 * (the caller works too); start()/wait() leave the caller free to drive the device meanwhile. */
class Pool {
public:
	explicit Pool(int threads)
	{
		for (int t = 0; t < threads; t++)
			workers_.emplace_back([this]() { loop(); });
	}
	~Pool()
	{
		{
			std::lock_guard<std::mutex> g(m_);
			quit_ = true;
		}
		cv_.notify_all();
		for (auto &t : workers_)
			t.join();
	}
	void start(size_t n, std::function<void(size_t)> fn)
	{
		std::lock_guard<std::mutex> g(m_);
		fn_ = std::move(fn);
		n_ = n;
		next_.store(0);
		active_ = workers_.size();
		gen_++;
		cv_.notify_all();
	}
	void wait()
	{
		std::unique_lock<std::mutex> g(m_);
		done_.wait(g, [this]() { return active_ == 0; });
	}
	void run(size_t n, const std::function<void(size_t)> &fn)
	{
		if (workers_.empty() || n <= 1) {
			for (size_t i = 0; i < n; i++)
				fn(i);
			return;
		}
		start(n, fn);
		for (size_t i; (i = next_.fetch_add(1)) < n;)
			fn(i);
		wait();
	}

private:
	void loop()
	{
		uint64_t seen = 0;
		for (;;) {
			{
				std::unique_lock<std::mutex> g(m_);
				cv_.wait(g, [&]() { return quit_ || gen_ != seen; });
				if (quit_)
					return;
				seen = gen_;
			}
			for (size_t i; (i = next_.fetch_add(1)) < n_;)
				fn_(i);
			std::lock_guard<std::mutex> g(m_);
			if (--active_ == 0)
				done_.notify_all();
		}
	}
	std::vector<std::thread> workers_;
	std::mutex m_;
	std::condition_variable cv_, done_;
	std::function<void(size_t)> fn_;
	std::atomic<size_t> next_{ 0 };
	size_t n_ = 0, active_ = 0;
	uint64_t gen_ = 0;
	bool quit_ = false;
};

inline uint64_t round_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

struct Slot {
	acm_stage_info info{};
	uint64_t need_blocks = 0;
	uint64_t idx_off = 0, hdr_off = 0, pcm_off = 0;
	uint64_t idx_len = 0;           /* arena words reserved (multiple of 64) */
	std::vector<acmhip_patch> patches;
	uint32_t chunk = 0;
	bool ok = false;
	bool host_staged = false;       /* staged by the host reader (its arena slice has to go H2D) */
	/* ACM_BATCH_STAGE_PACKED: the stream's chunk-table entries (reserved from the headers), how many whole tiles the pool packed
	 * (0: the stream travels as int16) and where its blobs landed in the blob arena */
	uint64_t pk_chunk_off = 0, pk_chunk_cap = 0;
	uint32_t pk_ntiles = 0;
	/* ACM_BATCH_STAGE_BYTEPLANE: where the stream's byte-plane block sits in the blob arena (bytes) and how many rows it may hold;
	 * pk_ntiles = the whole tiles the pool staged that way */
	uint64_t mf_off = 0, mf_rows_cap = 0, mf_used = 0;
	bool mf_fused = false;                          /* the form was written by the parsing pass itself (acm_stage_file_mform) */
	uint64_t mf_pair_off = 0;       /* its first entry in the pair table */
	uint32_t range_unit = 1;        /* device parsing in block ranges: the stream's ranges are cut at multiples of this many blocks (acmk_range_bound) */
};

/* Blocks a file can possibly hold: the header promises total_values, but arenas are sized by this - a block costs at
 * least its 20-bit header and a 5-bit filler code per column (decode.c:491-502, 586-589), and the reader appends one
 * virtual zero byte (decode.c:57-61).  A 19-byte file that claims 2^32-1 samples gets one block, not 8 GB. */
inline uint64_t blocks_possible(const acm_stage_info &info, size_t len)
{
	const uint64_t bl = (uint64_t)info.rows * info.cols;
	const uint64_t promised = ((uint64_t)info.total_values + bl - 1) / bl;
	const uint64_t bits = (len > info.header_bytes ? (uint64_t)(len - info.header_bytes) * 8 : 0) + 8;
	return std::min<uint64_t>(promised, bits / (20 + 5 * (uint64_t)info.cols) + 1);
}

/* a run of whole streams that travels through the device as one unit */
struct Chunk {
	size_t first = 0, last = 0;             /* stream index range [first, last) */
	uint64_t idx_begin = 0, idx_end = 0;    /* arena ranges (int16 units; the PCM arena has the same layout) */
	uint64_t hdr_begin = 0, hdr_end = 0;
	std::atomic<int> unparsed{ 0 };
	std::atomic<uint64_t> pk_used{ 0 };     /* ACM_BATCH_STAGE_PACKED: bytes of this chunk's blob region handed out so far */
	uint64_t pk_chunk_begin = 0, pk_chunk_end = 0;  /* its range of the chunk table (entries) */
	uint64_t mf_begin = 0, mf_end = 0;              /* ACM_BATCH_STAGE_BYTEPLANE: its range of the blob arena (bytes) */
	uint64_t mf_pair_begin = 0, mf_pair_end = 0;    /* ... and of the pair table (entries) */
	std::atomic<int> back{ 0 };             /* 1 = PCM is in the pinned arena, -1 = the read-back failed */
	acmhip_plan *plan = nullptr;
	hipEvent_t ev[5] = {};                  /* h2d begin, h2d end, kernel end (device stream); d2h begin, d2h end (copy stream) */
};

/* device parsing: the files of one chunk as an upload piece (copied into the pinned file arena by the pool, sent up as
 * soon as the piece is complete, while the pool copies the next ones).  The walk itself is ONE launch over all streams:
 * a stream's walk is one wavefront's sequential job (~30 ms for two megasamples however many streams walk beside it),
 * and walks launched group by group on streams of their own were measured to slow each other down (last group done
 * after 65 / 86 / 103 ms for 1 / 2 / 4 groups of the 1024-stream level-9 batch: the waves of a later launch land on the
 * compute units - and scalar units - the earlier ones already occupy). */
struct ParseGroup {
	size_t k_first = 0, k_last = 0;         /* range of the device-parsed streams (indices into dev_ids) */
	uint64_t file_begin = 0, file_end = 0;  /* bytes of the file arenas */
	uint64_t max_columns = 0;
	std::atomic<int> uncopied{ 0 };         /* files not yet in the pinned arena */
	hipEvent_t ev[3] = {};                  /* upload begin, upload end; the last piece's third one: parse results on the host */
};

} // namespace

/* what acm_batch_prestage keeps: per item the header, the status, the staged blocks (one allocation for all items, laid out
 * like the batch's own arenas) and the H1 patches */
struct acm_batch_prestaged {
	struct Item {
		acm_stage_info info{};
		uint64_t need_blocks = 0, idx_off = 0, hdr_off = 0;
		std::vector<acmhip_patch> patches;
		int status = 0;
		bool ok = false;
	};
	std::vector<Item> items;
	std::vector<const uint8_t *> data;      /* the file images the items pointed at, and their lengths (identity check in acm_batch_decode) */
	std::vector<size_t> len;
	int16_t *idx = nullptr;
	acmhip_blkhdr *hdr = nullptr;
	size_t idx_cap = 0, hdr_cap = 0;        /* bytes */
	int force_chans = 0;
};

namespace {
/* Staging buffers of prestaged groups are a hundred megabytes each and live for milliseconds: handing them back to the C
 * library means an munmap (10 ms for 200 MB of touched pages, on the thread that feeds the device) and a fresh set of page
 * faults for the next group.  A few of them are kept here instead and handed out again, best fit. */
struct StageCache {
	struct Buf {
		void *mem;
		size_t cap;
	};
	std::mutex m;
	std::vector<Buf> bufs;
	size_t bytes = 0;
	static constexpr size_t MAX_BUFS = 8, MAX_BYTES = (size_t)1 << 30;

	void *get(size_t want, size_t *cap)
	{
		{
			std::lock_guard<std::mutex> g(m);
			size_t best = bufs.size();
			for (size_t k = 0; k < bufs.size(); k++)
				if (bufs[k].cap >= want && (best == bufs.size() || bufs[k].cap < bufs[best].cap))
					best = k;
			if (best != bufs.size() && bufs[best].cap <= 2 * want + (1u << 20)) {
				const Buf b = bufs[best];
				bufs.erase(bufs.begin() + (long)best);
				bytes -= b.cap;
				*cap = b.cap;
				return b.mem;
			}
		}
		*cap = want;
		return malloc(want);
	}
	void put(void *mem, size_t cap)
	{
		if (!mem)
			return;
		{
			std::lock_guard<std::mutex> g(m);
			if (bufs.size() < MAX_BUFS && bytes + cap <= MAX_BYTES) {
				bufs.push_back(Buf{ mem, cap });
				bytes += cap;
				return;
			}
		}
		free(mem);
	}
};
StageCache g_stage_cache;
}

extern "C" void acm_batch_prestage_free(acm_batch_prestaged *p)
{
	if (!p)
		return;
	if (p->idx && p->idx_cap)
		g_stage_cache.put(p->idx, p->idx_cap);
	if (p->hdr && p->hdr_cap)
		g_stage_cache.put(p->hdr, p->hdr_cap);
	delete p;
}

extern "C" int acm_batch_prestage(const acm_batch_item *items, size_t n, const acm_batch_opts *opts_in, acm_batch_prestaged **out, double *seconds)
{
	if (!out || (n && !items))
		return ACMHIP_ERR_ARG;
	*out = nullptr;
	const auto t0 = clk::now();
	acm_batch_opts opts{};
	if (opts_in)
		opts = *opts_in;
	acm_batch_prestaged *p = new (std::nothrow) acm_batch_prestaged;
	if (!p)
		return ACMHIP_ERR_NOMEM;
	p->items.resize(n);
	p->data.resize(n);
	p->len.resize(n);
	p->force_chans = opts.force_chans;
	const int threads_wanted = opts.threads > 0 ? opts.threads : default_threads();
	Pool pool((int)std::min<size_t>((size_t)threads_wanted, std::max<size_t>(1, n)));
	pool.run(n, [&](size_t i) {
		acm_batch_prestaged::Item &s = p->items[i];
		p->data[i] = items[i].data;
		p->len[i] = items[i].len;
		s.status = acm_stage_probe(items[i].data, items[i].len, opts.force_chans, &s.info);
		s.ok = s.status == ACM_OK;
		if (s.ok)
			s.need_blocks = blocks_possible(s.info, items[i].len);
	});
	uint64_t idx_total = 0, hdr_total = 0;
	for (acm_batch_prestaged::Item &s : p->items) {
		if (!s.ok)
			continue;
		s.idx_off = idx_total;
		s.hdr_off = hdr_total;
		idx_total += round_up(s.need_blocks * (uint64_t)s.info.rows * s.info.cols, 64);
		hdr_total += s.need_blocks;
	}
	p->idx = static_cast<int16_t *>(g_stage_cache.get(std::max<uint64_t>(idx_total, 1) * sizeof(int16_t), &p->idx_cap));
	p->hdr = static_cast<acmhip_blkhdr *>(g_stage_cache.get(std::max<uint64_t>(hdr_total, 1) * sizeof(acmhip_blkhdr), &p->hdr_cap));
	if (!p->idx || !p->hdr) {
		acm_batch_prestage_free(p);
		return ACMHIP_ERR_NOMEM;
	}
	pool.run(n, [&](size_t i) {
		acm_batch_prestaged::Item &s = p->items[i];
		if (!s.ok)
			return;
		acm_stage_info info{};
		/* first pass counts patches (normally zero), second only if there are any */
		int r = acm_stage_file(items[i].data, items[i].len, opts.force_chans, p->idx + s.idx_off, p->hdr + s.hdr_off, s.need_blocks, nullptr, 0, &info);
		if (r == ACM_OK && info.npatches) {
			s.patches.resize(info.npatches);
			r = acm_stage_file(items[i].data, items[i].len, opts.force_chans, p->idx + s.idx_off, p->hdr + s.hdr_off, s.need_blocks,
					   s.patches.data(), s.patches.size(), &info);
		}
		if (r != ACM_OK) {
			s.status = r;
			s.ok = false;
			return;
		}
		s.info = info;
		s.status = info.end_status;
	});
	if (seconds)
		*seconds = secs(t0, clk::now());
	*out = p;
	return ACMHIP_OK;
}

extern "C" uint64_t acm_batch_pcm_words(const acm_batch_item *items, size_t n, int force_chans)
{
	uint64_t total = 0;
	for (size_t i = 0; items && i < n; i++) {
		acm_stage_info info;
		if (acm_stage_probe(items[i].data, items[i].len, force_chans, &info) != ACM_OK)
			continue;
		const uint64_t bl = (uint64_t)info.rows * info.cols;
		total += round_up(blocks_possible(info, items[i].len) * bl, 64);
	}
	return total;
}

extern "C" int acm_batch_decode(acmhip_device *dev, acm_batch_item *items, size_t n,
				const acm_batch_opts *opts_in, acm_batch_timing *timing)
{
	if (!dev || (n && !items))
		return ACMHIP_ERR_ARG;
	acm_batch_opts opts{};
	if (opts_in)
		opts = *opts_in;
	if (opts.fmt > 3 || opts.parse > ACM_BATCH_PARSE_AUTO)
		return ACMHIP_ERR_ARG;
	/* parsed ahead of time (acm_batch_prestage): must be these very items */
	const acm_batch_prestaged *pre = opts.prestaged;
	if (pre) {
		if (pre->items.size() != n || pre->force_chans != opts.force_chans)
			return ACMHIP_ERR_ARG;
		for (size_t i = 0; i < n; i++)
			if (pre->data[i] != items[i].data || pre->len[i] != items[i].len)
				return ACMHIP_ERR_ARG;          /* the staged blocks are copied into arenas laid out from THESE items' lengths */
		opts.parse = ACM_BATCH_PARSE_HOST;
	}
	acm_batch_timing tm{};
	const auto t0 = clk::now();
	static const bool trace = getenv("ACM_BATCH_TRACE") != nullptr;         /* host-side timeline on stderr */
#define BNOTE(...) do { if (trace) { fprintf(stderr, "[batch %8.3f ms] ", secs(t0, clk::now()) * 1e3); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); } } while (0)
	const int threads_wanted = opts.threads > 0 ? opts.threads : default_threads();
	const int threads = (int)std::min<size_t>((size_t)threads_wanted, std::max<size_t>(1, n));
	Pool pool(threads);

	/* 1. headers -> arena layout */
	std::vector<Slot> slots(n);
	pool.run(n, [&](size_t i) {
		Slot &s = slots[i];
		acm_batch_item &it = items[i];
		it.words = 0;
		it.dev_off = 0;
		it.status = acm_stage_probe(it.data, it.len, opts.force_chans, &s.info);
		it.level = s.info.level;
		it.rows = s.info.rows;
		it.channels = s.info.channels;
		it.rate = s.info.rate;
		it.total_values = s.info.total_values;
		s.ok = (it.status == ACM_OK);
		if (s.ok) {
			const uint64_t bl = (uint64_t)s.info.rows * s.info.cols;
			s.need_blocks = blocks_possible(s.info, it.len);
			s.idx_len = round_up(s.need_blocks * bl, 64);
		}
	});
	uint64_t idx_total = 0, hdr_total = 0;
	for (Slot &s : slots) {
		if (!s.ok)
			continue;
		s.idx_off = s.pcm_off = idx_total;
		s.hdr_off = hdr_total;
		idx_total += s.idx_len;
		hdr_total += s.need_blocks;
	}
	const uint64_t pcm_total = idx_total;
	const bool keep_on_device = opts.d_pcm != nullptr;
	/* pinned caller buffers: the copy engine writes every stream's PCM where the caller wants it (one transfer per
	 * stream, so only for streams big enough that the per-transfer cost disappears) */
	bool direct_out = !keep_on_device && (opts.flags & ACM_BATCH_PCM_PINNED) && n > 0 && pcm_total / n >= 32768;
	if (keep_on_device && opts.d_pcm_words < pcm_total)
		return ACMHIP_ERR_ARG;
	for (size_t i = 0; i < n; i++)
		items[i].dev_off = slots[i].pcm_off;

	/* chunks of whole streams: about 1/16 of the batch each, but not below 8 MiB of staged indices */
	const uint64_t chunk_target = std::max<uint64_t>(4u << 20, idx_total / 16);
	std::vector<size_t> starts{ 0 };
	{
		uint64_t cut_at = 0;
		for (size_t i = 0; i < n; i++)
			if (slots[i].ok && slots[i].idx_off - cut_at >= chunk_target) {
				starts.push_back(i);
				cut_at = slots[i].idx_off;
			}
	}
	std::vector<Chunk> chunks(starts.size());
	for (size_t c = 0; c < chunks.size(); c++) {
		Chunk &ch = chunks[c];
		ch.first = starts[c];
		ch.last = (c + 1 < chunks.size()) ? starts[c + 1] : n;
		bool any = false;
		for (size_t i = ch.first; i < ch.last; i++) {
			Slot &s = slots[i];
			if (!s.ok)
				continue;
			s.chunk = (uint32_t)c;
			if (!any) {
				ch.idx_begin = s.idx_off;
				ch.hdr_begin = s.hdr_off;
				any = true;
			}
			ch.idx_end = s.idx_off + s.idx_len;
			ch.hdr_end = s.hdr_off + s.need_blocks;
		}
	}

	/* the packed staged form (opt-in, host parsing): chunk-table entries are reserved per stream from what the headers promise;
	 * a pipeline chunk's blobs share the region of the blob arena that mirrors its slice of the int16 arena (a stream whose
	 * packed form does not fit there - indices that need 16 bits throughout - simply travels as int16) */
	/* (no second form where the plans may not use the lean kernels that read it) */
	const bool lean_off = (opts.plan_flags & ACMHIP_PLAN_NO_LEAN) || (ACM_TUNING_ENV("ACM_K2") && atoi(ACM_TUNING_ENV("ACM_K2")) == 0);
	bool stage_packed = (opts.flags & ACM_BATCH_STAGE_PACKED) && !lean_off && !(opts.plan_flags & ACMHIP_PLAN_STAGEWISE);
	uint64_t pk_chunks_total = 0;
	if (stage_packed)
		for (size_t c = 0; c < chunks.size(); c++) {
			chunks[c].pk_chunk_begin = pk_chunks_total;
			for (size_t i = chunks[c].first; i < chunks[c].last; i++) {
				Slot &s = slots[i];
				const int tr = s.ok ? acmhip_packed_tile_rows(s.info.level) : 0;
				if (tr <= 0)
					continue;
				s.pk_chunk_off = pk_chunks_total;
				s.pk_chunk_cap = s.need_blocks * s.info.rows / (uint64_t)tr * (uint64_t)acmhip_packed_slots(s.info.level);
				pk_chunks_total += s.pk_chunk_cap;
			}
			chunks[c].pk_chunk_end = pk_chunks_total;
		}

	/* the byte-plane staged form (the default; ACM_BATCH_STAGE_INT16 turns it off): every stream of a level the matrix-core build covers gets room for its
	 * rows plus the two rows of zeros in front */
	/* (blocks parsed ahead of time are int16 rows already: re-ordering them is a pass of its own, taken only when asked for) */
	const bool mform_default = !(opts.flags & (ACM_BATCH_STAGE_INT16 | ACM_BATCH_STAGE_PACKED)) && !pre;
	bool stage_mform = ((opts.flags & ACM_BATCH_STAGE_BYTEPLANE) || mform_default) && !lean_off && !(opts.plan_flags & ACMHIP_PLAN_STAGEWISE);
	uint64_t mf_total = 0, mf_pairs_total = 0;
	if (stage_mform) {
		stage_packed = false;           /* one second form per batch */
		for (size_t c = 0; c < chunks.size(); c++) {
			chunks[c].mf_begin = mf_total;
			chunks[c].mf_pair_begin = mf_pairs_total;
			for (size_t i = chunks[c].first; i < chunks[c].last; i++) {
				Slot &s = slots[i];
				if (!s.ok || acmhip_mform_tile_rows(s.info.level) <= 0)
					continue;
				s.mf_off = mf_total;
				s.mf_rows_cap = (s.need_blocks * s.info.rows) & ~1ull;
				s.mf_pair_off = mf_pairs_total;
				mf_total += (acmhip_mform_bytes(s.info.level, s.mf_rows_cap) + 255) & ~255ull;
				mf_pairs_total += acmhip_mform_pairs(s.mf_rows_cap);
			}
			chunks[c].mf_end = mf_total;
			chunks[c].mf_pair_end = mf_pairs_total;
		}
	}

	/* AUTO: the device walk takes as long as the longest stream takes one wavefront, the host pool takes total / threads.
	 * Measured (profiles/r2_parse_probe.txt): a wavefront alone on its SIMD walks at ~1/5 of a host core's parsing rate
	 * (up to 1024 streams), with four per SIMD at ~1/9 (up to the 32 K streams acm_parse_scan_wave takes), a lane of
	 * acm_parse_scan at ~1/16: device when the batch is worth more than that many x threads streams of the longest one */
	bool dev_parse = opts.parse == ACM_BATCH_PARSE_DEVICE;
	if (opts.parse == ACM_BATCH_PARSE_AUTO) {
		uint64_t longest = 0;
		for (const Slot &s : slots)
			if (s.ok)
				longest = std::max(longest, s.idx_len);
		const uint64_t per_thread = n <= 1024 ? 5 : n <= 32768 ? 9 : 16;
		dev_parse = longest > 0 && idx_total / longest >= per_thread * (uint64_t)threads_wanted;
	}
	/* the device parser writes the byte-plane form itself where a stream can have it (the chunk kernel's levels, even acm_rows): rows the
	 * lean kernels take never exist as int16 then (acm_parse.hip: acm_parse_columns); the host pool's second forms are host-parsing only */
	bool dev_mform = false;
	if (dev_parse) {
		dev_mform = stage_mform && !(ACM_TUNING_ENV("ACM_BATCH_DEV_MFORM") && atoi(ACM_TUNING_ENV("ACM_BATCH_DEV_MFORM")) == 0);
		stage_packed = stage_mform = false;
	}
	uint64_t files_total = 0, cols_total = 0;
	std::vector<uint64_t> file_off;
	std::vector<size_t> dev_ids;                    /* streams handed to the device parser */
	if (dev_parse) {
		file_off.resize(n);
		for (size_t i = 0; i < n; i++) {
			const Slot &s = slots[i];
			if (!s.ok || !acmk_parse_supported(s.info.level, s.info.rows, items[i].len, s.need_blocks))
				continue;
			file_off[i] = files_total;
			files_total += round_up(items[i].len, 16) + 16; /* zero tail: the device readers load whole dwords */
			cols_total += s.need_blocks << s.info.level;
			dev_ids.push_back(i);
		}
	}

	/* Block ranges (device parsing of a big batch into host buffers): the walk of a stream is one wavefront's sequential job -
	 * 33 ms for two megasamples - and nothing can be synthesised or read back before it ends.  So the walk is cut into R
	 * launches, each taking every stream R-th of its blocks further (the bit offset it stopped at stays on the device), and
	 * the synthesis and the read-back of range r run while range r + 1 is walked.  The PCM arenas are range-major for that:
	 * range r of every stream back to back, one transfer per range; the copy-out puts the pieces where the caller wants them. */
	size_t R = 1;
	{
		size_t ok_streams = 0;
		for (const Slot &s : slots)
			ok_streams += s.ok;
		/* a range of ~128 Msamples is walked in ~2 ms and read back in ~5: 2 ... 16 ranges from 256 Msamples on (measured on
		 * the 2.1-Gsample batch: 0.136 / 0.114 / 0.106 / 0.104 s with 1 / 4 / 8 / 16 ranges, profiles/r3_batch_timeline.txt) */
		size_t want = idx_total >= (256u << 20) ? (size_t)std::min<uint64_t>(16, idx_total >> 27) : 1;
		if ((opts.flags >> 8) & 0xFFu)                          /* ACM_BATCH_RANGES(n): the caller's count (1 = in one piece) */
			want = (opts.flags >> 8) & 0xFFu;
		else if (const char *e = ACM_TUNING_ENV("ACM_BATCH_RANGES"))
			want = (size_t)std::max(1, atoi(e));
		if (want > 1 && want <= 64 && dev_parse && !keep_on_device && !dev_ids.empty() && dev_ids.size() == ok_streams &&
		    dev_ids.size() <= ACM_PARSE_RANGE_MAX_STREAMS)
			R = want;
		if (R > 1)
			direct_out = false;     /* one transfer per range into the library's arena; pinned caller buffers make the copy-out fault-free */
	}
	std::vector<uint64_t> piece_off, piece_len, rbase;      /* [r * n + i]: where range r of stream i sits in the PCM arenas, words */
	uint64_t pcm_arena_words = pcm_total;
	if (R > 1) {
		/* where a stream may travel in the byte-plane form a range has to end on a whole tile of the lean kernel (the plan of a range is a
		 * window; its ragged end would need int16 rows nobody writes) - which also keeps the row pairs of an odd block height, every
		 * other one of which lies across two blocks, inside one range.  Blocks that are whole tiles: unit 1, ranges as ever */
		for (size_t i = 0; i < n; i++) {
			Slot &s = slots[i];
			if (!s.ok)
				continue;
			const int T2 = acmk_tile2_rows(s.info.level), TM = acmk_tile2m_rows(s.info.level);
			if (acmk_tile2m_stages(s.info.level) == 6 && T2 > 0 && TM > 0 && T2 % TM == 0 && s.info.level <= ACM_K1_MAX_LEVEL) {
				uint32_t a = s.info.rows, b = (uint32_t)T2;
				while (b) {
					const uint32_t t = a % b;
					a = b;
					b = t;
				}
				s.range_unit = (uint32_t)T2 / a;
			}
		}
		piece_off.assign(R * n, 0);
		piece_len.assign(R * n, 0);
		rbase.assign(R + 1, 0);
		uint64_t at = 0;
		for (size_t r = 0; r < R; r++) {
			rbase[r] = at;
			for (size_t i = 0; i < n; i++) {
				const Slot &s = slots[i];
				if (!s.ok)
					continue;
				const uint64_t bl = (uint64_t)s.info.rows * s.info.cols;
				const uint64_t words = deliverable_words(s.info.total_values, bl, s.info.channels, s.need_blocks);
				const uint64_t lo = std::min(words, (uint64_t)acmk_range_bound((uint32_t)s.need_blocks, (uint32_t)r, (uint32_t)R, s.range_unit) * bl);
				const uint64_t hi = std::min(words, (uint64_t)acmk_range_bound((uint32_t)s.need_blocks, (uint32_t)r + 1u, (uint32_t)R, s.range_unit) * bl);
				piece_off[r * n + i] = at;
				piece_len[r * n + i] = hi - lo;
				at += round_up(hi - lo, 64);
			}
		}
		rbase[R] = at;
		pcm_arena_words = std::max(at, pcm_total);
	}

	const auto t_hdr = clk::now();
	int16_t *h_idx = nullptr, *h_pcm = nullptr, *d_idx = nullptr, *d_pcm = nullptr;
	uint32_t *d_colpos = nullptr;
	acmhip_blkhdr *h_hdr = nullptr, *d_hdr = nullptr;
	uint8_t *h_files = nullptr, *d_files = nullptr, *h_jobs = nullptr, *d_jobs = nullptr;
	uint8_t *h_pkblob = nullptr, *d_pkblob = nullptr;
	acmhip_packed_chunk *h_pkchunk = nullptr, *d_pkchunk = nullptr;
	const uint64_t pk_blob_bytes = idx_total * sizeof(int16_t) + 4096;     /* region of chunk c = bytes [2 idx_begin, 2 idx_end) */
	hipStream_t st_main = (hipStream_t)acmhip_device_stream(dev), st_copy = nullptr;
	int rc = ACMHIP_OK;
	std::mutex m;
	std::condition_variable cv;
	std::atomic<size_t> issued{ 0 };                /* chunks whose read-back has been queued */
	bool aborted = false;
	bool pool_busy = false;
	const size_t npg = dev_parse && !dev_ids.empty() ? chunks.size() : 0;
	hipStream_t st_up = nullptr, st_parse = nullptr;        /* file uploads piece by piece; the walk + column kernels */
	std::vector<ParseGroup> groups(npg);
	std::vector<Chunk> rchunks(R > 1 ? R : 0);              /* block ranges: plan, events and read-back state of each */
	std::vector<hipEvent_t> ev_stripe(R > 1 ? R + 1 : 0, nullptr);  /* [0] = first stripe's upload begins, [s + 1] = stripe s is in place */
	std::unique_ptr<std::atomic<int>[]> stripe_uncopied(R > 1 ? new std::atomic<int>[R] : nullptr);

	auto cleanup = [&]() {
		if (pool_busy) {
			{
				std::lock_guard<std::mutex> g(m);
				aborted = true;
			}
			cv.notify_all();
			pool.wait();
		}
		if (st_up)
			(void)hipStreamSynchronize(st_up);
		if (st_parse)
			(void)hipStreamSynchronize(st_parse);
		(void)hipStreamSynchronize(st_main);
		if (st_copy)
			(void)hipStreamSynchronize(st_copy);
		for (ParseGroup &g : groups)
			for (hipEvent_t e : g.ev)
				if (e)
					(void)hipEventDestroy(e);
		for (hipEvent_t e : ev_stripe)
			if (e)
				(void)hipEventDestroy(e);
		for (std::vector<Chunk> *v : { &chunks, &rchunks })
			for (Chunk &ch : *v) {
				acmhip_plan_destroy(ch.plan);
				for (hipEvent_t e : ch.ev)
					if (e)
						(void)hipEventDestroy(e);
			}
		acmhip_arena_unlock(dev);
	};
#define BTRY(call) do { rc = (call); if (rc != ACMHIP_OK) { cleanup(); return rc; } } while (0)
#define HTRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = acmhip_report_hip((int)e_, #call); cleanup(); return rc; } } while (0)
	/* arenas live in the device handle and are reused by the next batch */
	acmhip_arena_lock(dev);
	BTRY(acmhip_copy_stream(dev, (void **)&st_copy));
	if (!dev_parse) {                       /* with device parsing the host staging arenas come later, and only if a stream needs them */
		BTRY(acmhip_arena_get(dev, ACM_ARENA_H_IDX, idx_total * sizeof(int16_t), (void **)&h_idx));
		BTRY(acmhip_arena_get(dev, ACM_ARENA_H_HDR, hdr_total * sizeof(acmhip_blkhdr), (void **)&h_hdr));
	}
	if (stage_packed && pk_chunks_total) {
		BTRY(acmhip_arena_get(dev, ACM_ARENA_H_PKBLOB, pk_blob_bytes, (void **)&h_pkblob));
		BTRY(acmhip_arena_get(dev, ACM_ARENA_D_PKBLOB, pk_blob_bytes, (void **)&d_pkblob));
		BTRY(acmhip_arena_get(dev, ACM_ARENA_H_PKCHUNK, pk_chunks_total * sizeof(acmhip_packed_chunk), (void **)&h_pkchunk));
		BTRY(acmhip_arena_get(dev, ACM_ARENA_D_PKCHUNK, pk_chunks_total * sizeof(acmhip_packed_chunk), (void **)&d_pkchunk));
	} else {
		stage_packed = false;
	}
	if (stage_mform && mf_total && (mf_total >> 6) < (1ull << 30)) {         /* (the pair table counts 64-byte units in 30 bits) */
		BTRY(acmhip_arena_get(dev, ACM_ARENA_H_PKBLOB, mf_total, (void **)&h_pkblob));
		BTRY(acmhip_arena_get(dev, ACM_ARENA_D_PKBLOB, mf_total, (void **)&d_pkblob));
		/* (+ 32 entries: the kernel's scalar loads fetch whole groups) */
		BTRY(acmhip_arena_get(dev, ACM_ARENA_H_PKCHUNK, (mf_pairs_total + 32) * sizeof(acmhip_mform_pair), (void **)&h_pkchunk));
		BTRY(acmhip_arena_get(dev, ACM_ARENA_D_PKCHUNK, (mf_pairs_total + 32) * sizeof(acmhip_mform_pair), (void **)&d_pkchunk));
	} else {
		stage_mform = false;
	}
	uint32_t *d_blkoff = nullptr;
	if (dev_mform && mf_total && (mf_total >> 6) < (1ull << 30) && dev_ids.size() <= ACM_PARSE_RANGE_MAX_STREAMS) {
		/* (+ 128 KB: a stream whose walk stops early has its unwritten pair-table entries parked right behind its last staged block,
		 * acm_parse.hip, and a chunk read from there may reach two pairs on - behind the last stream that is behind the arena's contents) */
		BTRY(acmhip_arena_get(dev, ACM_ARENA_D_PKBLOB, mf_total + (128u << 10), (void **)&d_pkblob));
		BTRY(acmhip_arena_get(dev, ACM_ARENA_D_PKCHUNK, (mf_pairs_total + 32) * sizeof(acmhip_mform_pair), (void **)&d_pkchunk));
		BTRY(acmhip_arena_get(dev, ACM_ARENA_D_BLKOFF, hdr_total * sizeof(uint32_t), (void **)&d_blkoff));
	} else {
		dev_mform = false;
	}
	if (!keep_on_device && !direct_out)
		BTRY(acmhip_arena_get(dev, ACM_ARENA_H_PCM, pcm_arena_words * sizeof(int16_t), (void **)&h_pcm));
	BTRY(acmhip_arena_get(dev, ACM_ARENA_D_IDX, idx_total * sizeof(int16_t), (void **)&d_idx));
	BTRY(acmhip_arena_get(dev, ACM_ARENA_D_HDR, hdr_total * sizeof(acmhip_blkhdr), (void **)&d_hdr));
	if (keep_on_device)
		d_pcm = static_cast<int16_t *>(opts.d_pcm);
	else
		BTRY(acmhip_arena_get(dev, ACM_ARENA_D_PCM, pcm_arena_words * sizeof(int16_t), (void **)&d_pcm));
	const size_t jobs_bytes = round_up(dev_ids.size() * sizeof(AcmParseJob), 64);
	const size_t res_bytes = dev_ids.size() * (sizeof(AcmParseResult) + sizeof(uint32_t));  /* results, then flags */
	/* block ranges: the files go up in R stripes (stripe s of every file back to back: one transfer, then a scatter kernel),
	 * so that range 0 is walked, synthesised and on its way back while the later stripes are still going up */
	const size_t nd = dev_ids.size();
	const size_t stripe_tab_off = jobs_bytes + round_up(res_bytes, 64);
	const size_t stripe_tab_bytes = R > 1 ? R * nd * sizeof(uint64_t) : 0;
	uint8_t *d_stage = nullptr;
	if (!dev_ids.empty()) {
		BTRY(acmhip_arena_get(dev, ACM_ARENA_H_FILES, files_total, (void **)&h_files));
		BTRY(acmhip_arena_get(dev, ACM_ARENA_D_FILES, files_total, (void **)&d_files));
		BTRY(acmhip_arena_get(dev, ACM_ARENA_D_COLPOS, cols_total * sizeof(uint32_t), (void **)&d_colpos));
		BTRY(acmhip_arena_get(dev, ACM_ARENA_H_JOBS, stripe_tab_off + stripe_tab_bytes, (void **)&h_jobs));
		BTRY(acmhip_arena_get(dev, ACM_ARENA_D_JOBS, stripe_tab_off + stripe_tab_bytes, (void **)&d_jobs));
		if (R > 1)
			BTRY(acmhip_arena_get(dev, ACM_ARENA_D_STAGE, files_total, (void **)&d_stage));
	}
	for (std::vector<Chunk> *v : { &chunks, &rchunks })
		for (Chunk &ch : *v)
			for (hipEvent_t &e : ch.ev)
				HTRY(hipEventCreateWithFlags(&e, hipEventBlockingSync));
	if (!groups.empty()) {
		BTRY(acmhip_aux_stream(dev, ACM_AUX_STREAMS - 1, (void **)&st_up));
		/* block ranges: walk r, its columns and its synthesis (0.2 ms) follow each other anyway - one stream; what must never
		 * share a hardware queue is the upload, the read-back and the kernels (the runtime maps streams onto a handful of
		 * queues: an upload queued behind a 5 ms read-back on the same one arrives 5 ms late, every stripe of it) */
		if (R > 1)
			st_parse = st_main;
		else
			BTRY(acmhip_aux_stream(dev, 0, (void **)&st_parse));
	}
	for (size_t g = 0; g < groups.size(); g++)
		for (hipEvent_t &e : groups[g].ev)
			HTRY(hipEventCreateWithFlags(&e, hipEventBlockingSync));
	for (hipEvent_t &e : ev_stripe)
		HTRY(hipEventCreateWithFlags(&e, hipEventBlockingSync));
	const auto t_alloc = clk::now();
	tm.alloc_s = secs(t_hdr, t_alloc);
	BNOTE("headers %.3f ms, arenas + events %.3f ms; %zu chunks, %zu parse groups", secs(t0, t_hdr) * 1e3, tm.alloc_s * 1e3, chunks.size(), groups.size());

	/* the packed half of the host stager: the whole tiles of a clean stream, from the int16 rows the reader has just written
	 * (they are still in the cache), into the chunk's region of the blob arena */
	auto host_pack = [&](size_t i) {
		Slot &s = slots[i];
		if (!stage_packed || !s.ok || !s.pk_chunk_cap || !s.patches.empty() || items[i].words == 0)
			return;
		const int tr = acmhip_packed_tile_rows(s.info.level);
		const uint64_t full_rows = std::min<uint64_t>((uint64_t)s.info.blocks * s.info.rows, items[i].words >> s.info.level);
		const uint64_t ntiles = full_rows / (uint64_t)tr, slots_per = (uint64_t)acmhip_packed_slots(s.info.level);
		if (ntiles == 0 || ntiles * slots_per > s.pk_chunk_cap)
			return;
		uint64_t bound = 0;
		if (acmhip_pack_bound(s.info.level, ntiles, &bound) != ACMHIP_OK)
			return;
		static thread_local std::vector<uint8_t> scratch;
		if (scratch.size() < bound)
			scratch.resize(bound);
		uint64_t bytes = 0;
		acmhip_packed_chunk *tc = h_pkchunk + s.pk_chunk_off;
		if (acmhip_pack_tiles(s.info.level, h_idx + s.idx_off, ntiles, tc, scratch.data(), 0, &bytes) != ACMHIP_OK)
			return;
		Chunk &ch = chunks[s.chunk];
		const uint64_t region = ch.idx_begin * sizeof(int16_t), region_len = (ch.idx_end - ch.idx_begin) * sizeof(int16_t);
		const uint64_t at = ch.pk_used.fetch_add((bytes + 15) & ~15ull);
		if (at + bytes > region_len)
			return;                 /* does not fit beside the others: this stream travels as int16 (its entries stay unused) */
		memcpy(h_pkblob + region + at, scratch.data(), bytes);
		for (uint64_t k = 0; k < ntiles * slots_per; k++)
			if (tc[k].kind)
				tc[k].blob_off16 += (uint32_t)((region + at) / 16);
		s.pk_ntiles = (uint32_t)ntiles;
	};
	/* the byte-plane half: the whole tiles of a clean stream, re-ordered from the int16 rows the reader has just written */
	auto host_mform = [&](size_t i) {
		Slot &s = slots[i];
		if (!stage_mform || !s.ok || !s.mf_rows_cap || !s.patches.empty() || items[i].words == 0 || s.mf_fused)
			return;
		const int tr = acmhip_mform_tile_rows(s.info.level);
		const uint64_t full_rows = std::min<uint64_t>((uint64_t)s.info.blocks * s.info.rows, items[i].words >> s.info.level);
		uint64_t ntiles = full_rows / (uint64_t)tr;
		if ((ntiles * (uint64_t)tr) & 1)
			ntiles--;                       /* the form is written pair by pair (tiles of one row: an even number of them) */
		if (ntiles == 0 || ntiles * (uint64_t)tr > s.mf_rows_cap)
			return;
		if (acmhip_mform_rows(s.info.level, h_idx + s.idx_off, ntiles * (uint64_t)tr, h_pkblob + s.mf_off, s.mf_off,
				      reinterpret_cast<acmhip_mform_pair *>(h_pkchunk) + s.mf_pair_off, &s.mf_used) != ACMHIP_OK)
			return;
		s.pk_ntiles = (uint32_t)ntiles;
	};
	/* the exact host reader, one stream */
	auto host_stage_int16 = [&](size_t i) {
		Slot &s = slots[i];
		acm_batch_item &it = items[i];
		acm_stage_info info{};
		if (pre) {
			/* parsed already: the staged blocks only have to move into the upload arenas */
			const acm_batch_prestaged::Item &ps = pre->items[i];
			if (!ps.ok) {
				it.status = ps.status;
				s.ok = false;
				return;
			}
			const uint64_t bl = (uint64_t)ps.info.rows * ps.info.cols;
			if (ps.need_blocks != s.need_blocks || ps.info.blocks > s.need_blocks) {      /* cannot happen for the same bytes */
				it.status = ACM_ERR_OTHER;
				s.ok = false;
				return;
			}
			memcpy(h_idx + s.idx_off, pre->idx + ps.idx_off, ps.info.blocks * bl * sizeof(int16_t));
			memcpy(h_hdr + s.hdr_off, pre->hdr + ps.hdr_off, ps.info.blocks * sizeof(acmhip_blkhdr));
			s.patches = ps.patches;
			s.info = ps.info;
			s.host_staged = true;
			it.status = ps.info.end_status;
			it.words = deliverable_words(ps.info.total_values, bl, ps.info.channels, ps.info.blocks);
			return;
		}
		/* first pass counts patches (normally zero), second only if there are any.  With ACM_BATCH_STAGE_BYTEPLANE the first pass also
		 * writes the byte-plane form, block by block out of the cache (acm_stage_file_mform), where the stream can have it */
		int r;
		if (stage_mform && s.mf_rows_cap) {
			uint64_t mf_rows = 0, mf_bytes = 0;
			r = acm_stage_file_mform(it.data, it.len, opts.force_chans, h_idx + s.idx_off, h_hdr + s.hdr_off, s.need_blocks, &info,
						 h_pkblob + s.mf_off, s.mf_off, reinterpret_cast<acmhip_mform_pair *>(h_pkchunk) + s.mf_pair_off, &mf_rows,
						 &mf_bytes);
			if (r == ACM_OK && mf_rows && mf_rows <= s.mf_rows_cap) {
				s.pk_ntiles = (uint32_t)(mf_rows / (uint64_t)acmhip_mform_tile_rows(info.level));
				s.mf_used = mf_bytes;
				s.mf_fused = true;
			}
		} else {
			r = acm_stage_file(it.data, it.len, opts.force_chans, h_idx + s.idx_off, h_hdr + s.hdr_off, s.need_blocks, nullptr, 0, &info);
		}
		if (r == ACM_OK && info.npatches) {
			s.patches.resize(info.npatches);
			r = acm_stage_file(it.data, it.len, opts.force_chans, h_idx + s.idx_off, h_hdr + s.hdr_off,
					   s.need_blocks, s.patches.data(), s.patches.size(), &info);
		}
		if (r != ACM_OK) {
			it.status = r;
			s.ok = false;
			return;
		}
		s.info = info;
		s.host_staged = true;
		it.status = info.end_status;
		it.words = deliverable_words(info.total_values, (uint64_t)info.rows * info.cols, info.channels, info.blocks);
	};
	auto host_stage = [&](size_t i) {
		host_stage_int16(i);
		host_pack(i);
		host_mform(i);
	};

	/* 2a. device parsing (optional): the streams the device parser takes are copied into the pinned file arena by the pool,
	 * group by group; whatever the device flags later is re-parsed by the exact host reader on this thread (rare) */
	std::vector<size_t> host_ids;                   /* streams the host pool parses (known up front) */
	std::vector<char> on_dev(n, 0);
	AcmParseJob *jobs = reinterpret_cast<AcmParseJob *>(h_jobs);
	AcmParseResult *results = reinterpret_cast<AcmParseResult *>(h_jobs ? h_jobs + jobs_bytes : nullptr);
	const uint32_t *flags = reinterpret_cast<const uint32_t *>(results + dev_ids.size());
	std::vector<size_t> group_of_chunk(chunks.size(), 0);
	if (!groups.empty()) {
		for (size_t c = 0; c < chunks.size(); c++)
			group_of_chunk[c] = c;
		uint64_t col_off = 0;
		for (ParseGroup &g : groups)
			g.k_first = g.k_last = dev_ids.size();
		for (size_t k = 0; k < dev_ids.size(); k++) {
			const size_t i = dev_ids[k];
			const Slot &s = slots[i];
			on_dev[i] = 1;
			AcmParseJob &j = jobs[k];
			j.file_off = file_off[i];
			j.idx_off = s.idx_off;
			j.hdr_off = s.hdr_off;
			j.col_off = col_off;
			j.file_len = (uint32_t)items[i].len;
			j.data_start = (uint32_t)s.info.header_bytes;
			j.level = s.info.level;
			j.rows = s.info.rows;
			j.blocks = (uint32_t)s.need_blocks;
			j.range_unit = s.range_unit;
			j.mf_off = j.mf_pair_off = j.mf_rows = 0;
			if (dev_mform && s.mf_rows_cap) {
				/* rows [0, mf_rows) - the whole tiles of the lean kernel, as the plan will cut them - are staged in the byte-plane form.
				 * With block ranges every range ends on a tile boundary (Slot::range_unit, above) */
				const int T2 = acmk_tile2_rows(s.info.level), TM = acmk_tile2m_rows(s.info.level);
				if (acmk_tile2m_stages(s.info.level) == 6 && T2 > 0 && TM > 0 && T2 % TM == 0 && s.info.level <= ACM_K1_MAX_LEVEL) {
					const uint64_t bl = (uint64_t)s.info.rows * s.info.cols;
					const uint64_t words = deliverable_words(s.info.total_values, bl, s.info.channels, s.need_blocks);
					const uint64_t rows2 = std::min<uint64_t>(s.need_blocks * s.info.rows, words >> s.info.level) / (uint64_t)T2 * (uint64_t)T2;
					if (rows2 && rows2 <= s.mf_rows_cap && rows2 < (1ull << 32)) {
						j.mf_off = s.mf_off;
						j.mf_pair_off = (uint32_t)s.mf_pair_off;
						j.mf_rows = (uint32_t)rows2;
						slots[i].pk_ntiles = (uint32_t)(rows2 / (uint64_t)TM);
					}
				}
			}
			col_off += s.need_blocks << s.info.level;
			ParseGroup &g = groups[group_of_chunk[s.chunk]];
			if (g.k_first == dev_ids.size()) {
				g.k_first = k;
				g.file_begin = file_off[i];
			}
			g.k_last = k + 1;
			g.file_end = file_off[i] + round_up(items[i].len, 16) + 16;
			g.max_columns = std::max<uint64_t>(g.max_columns, s.need_blocks << s.info.level);
			g.uncopied.fetch_add(1);
		}
	}
	uint64_t *stripe_at = reinterpret_cast<uint64_t *>(h_jobs ? h_jobs + stripe_tab_off : nullptr);  /* [s * nd + k]: stripe s of stream k in the striped arenas */
	std::vector<uint64_t> stripe_base(R > 1 ? R + 1 : 0, 0);
	constexpr size_t STRIPE_BATCH = 32;             /* files per copy task */
	const size_t nbat = (nd + STRIPE_BATCH - 1) / STRIPE_BATCH;
	if (R > 1) {
		uint64_t at = 0;
		for (size_t s = 0; s < R; s++) {
			stripe_base[s] = at;
			for (size_t k = 0; k < nd; k++) {
				const uint32_t len = (uint32_t)items[dev_ids[k]].len;
				stripe_at[s * nd + k] = at;
				at += acmk_stripe_bound(len, (uint32_t)s + 1, (uint32_t)R) - acmk_stripe_bound(len, (uint32_t)s, (uint32_t)R);
			}
			stripe_uncopied[s].store((int)nbat);
		}
		stripe_base[R] = at;            /* == files_total: the stripes tile every slot */
	}
	for (size_t i = 0; i < n; i++)
		if (slots[i].ok && !on_dev[i])
			host_ids.push_back(i);                  /* ascending, i.e. arena order */
	if (dev_parse && !host_ids.empty()) {
		const auto ta = clk::now();
		BTRY(acmhip_arena_get(dev, ACM_ARENA_H_IDX, idx_total * sizeof(int16_t), (void **)&h_idx));
		BTRY(acmhip_arena_get(dev, ACM_ARENA_H_HDR, hdr_total * sizeof(acmhip_blkhdr), (void **)&h_hdr));
		tm.alloc_s += secs(ta, clk::now());
	}
	tm.host_parsed = host_ids.size();
	for (size_t i : host_ids)
		chunks[slots[i].chunk].unparsed.fetch_add(1);

	/* 2b. the pool: files of the device-parsed streams into the pinned arena (group order), host parsing in arena order,
	 * then finished PCM out to the callers' buffers as the chunks come back */
	std::vector<size_t> out_ids;
	for (size_t i = 0; i < n; i++)
		if (slots[i].ok && items[i].pcm && !keep_on_device && !direct_out)
			out_ids.push_back(i);
	std::atomic<size_t> parsed{ 0 };
	clk::time_point t_parsed = clk::now();
	const size_t ncopy = groups.empty() ? 0 : R > 1 ? R * nbat : dev_ids.size();
	const size_t nparse = host_ids.size();
	pool_busy = true;
	pool.start(ncopy + nparse + out_ids.size() * R, [&](size_t task) {
		if (task < ncopy && R > 1) {
			/* stripe-major: stripe 0 of every file first (a task = one stripe of STRIPE_BATCH files) */
			const size_t s = task / nbat, k0 = task % nbat * STRIPE_BATCH;
			for (size_t k = k0; k < std::min(nd, k0 + STRIPE_BATCH); k++) {
				const acm_batch_item &it = items[dev_ids[k]];
				const uint64_t lo = acmk_stripe_bound((uint32_t)it.len, (uint32_t)s, (uint32_t)R),
					       hi = acmk_stripe_bound((uint32_t)it.len, (uint32_t)s + 1, (uint32_t)R);
				uint8_t *dst = h_files + stripe_at[s * nd + k];
				const uint64_t have = it.len > lo ? std::min<uint64_t>(it.len, hi) - lo : 0;       /* file bytes in this stripe */
				if (have)
					memcpy(dst, static_cast<const uint8_t *>(it.data) + lo, have);
				memset(dst + have, 0, hi - lo - have);                                        /* the slot's zero tail */
			}
			if (stripe_uncopied[s].fetch_sub(1) == 1) {
				std::lock_guard<std::mutex> g(m);
				cv.notify_all();
			}
			return;
		}
		if (task < ncopy) {
			const size_t i = dev_ids[task];
			const acm_batch_item &it = items[i];
			uint8_t *dst = h_files + file_off[i];
			memcpy(dst, it.data, it.len);
			memset(dst + it.len, 0, round_up(it.len, 16) + 16 - it.len);     /* zero tail: the device readers load whole dwords */
			if (groups[group_of_chunk[slots[i].chunk]].uncopied.fetch_sub(1) == 1) {
				std::lock_guard<std::mutex> g(m);
				cv.notify_all();
			}
			return;
		}
		task -= ncopy;
		if (task < nparse) {
			const size_t i = host_ids[task];
			const uint32_t c = slots[i].chunk;
			host_stage(i);
			if (parsed.fetch_add(1) + 1 == nparse)
				t_parsed = clk::now();
			if (chunks[c].unparsed.fetch_sub(1) == 1) {
				std::lock_guard<std::mutex> g(m);
				cv.notify_all();
			}
			return;
		}
		/* copy-out: a whole stream of its chunk - or, with block ranges, one range's piece of a stream */
		const size_t piece = (task - nparse) / out_ids.size();          /* 0 without ranges */
		const size_t i = out_ids[(task - nparse) % out_ids.size()];
		const Slot &s = slots[i];
		const size_t unit = R > 1 ? piece : s.chunk;
		Chunk &ch = R > 1 ? rchunks[piece] : chunks[s.chunk];
		if (ch.back.load(std::memory_order_acquire) == 0) {     /* first one here waits for the chunk */
			if (issued.load(std::memory_order_acquire) <= unit) {
				std::unique_lock<std::mutex> g(m);
				cv.wait(g, [&]() { return aborted || issued.load() > unit; });
				if (aborted)
					return;
			}
			const bool good = hipEventSynchronize(ch.ev[4]) == hipSuccess;
			ch.back.store(good ? 1 : -1, std::memory_order_release);
			BNOTE("%s %zu is back", R > 1 ? "range" : "chunk", unit);
		}
		if (ch.back.load(std::memory_order_acquire) < 0 || !s.ok)
			return;
		if (R > 1) {
			/* the piece starts where the stream's earlier ranges end (they are whole blocks); a stream the device flags
			 * later is decoded again from the host reader's staging and copied out whole (fix-up, below) */
			const uint64_t bl = (uint64_t)s.info.rows * s.info.cols;
			const uint64_t lo = (uint64_t)acmk_range_bound((uint32_t)s.need_blocks, (uint32_t)piece, (uint32_t)R, s.range_unit) * bl, len = piece_len[piece * n + i];
			if (lo >= items[i].pcm_cap || len == 0)
				return;
			memcpy(items[i].pcm + lo, h_pcm + piece_off[piece * n + i], std::min<uint64_t>(len, items[i].pcm_cap - lo) * sizeof(int16_t));
			return;
		}
		if (items[i].words == 0)
			return;
		const uint64_t w = std::min<uint64_t>(items[i].words, items[i].pcm_cap);
		memcpy(items[i].pcm, h_pcm + s.pcm_off, w * sizeof(int16_t));
	});

	/* the launch plan of one chunk from what is known about its streams right now */
	auto build_plan = [&](size_t c, uint64_t *samples) -> int {
		Chunk &ch = chunks[c];
		std::vector<acmhip_stream_desc> descs;
		std::vector<acmhip_patch> patches;
		std::vector<acmhip_packed_stream> packed;
		for (size_t i = ch.first; i < ch.last; i++) {
			Slot &s = slots[i];
			if (!s.ok || items[i].words == 0)
				continue;
			if (stage_mform || dev_mform)
				packed.push_back(acmhip_packed_stream{ s.mf_pair_off, s.pk_ntiles, ACMHIP_FORM_BYTEPLANE });
			else
				packed.push_back(acmhip_packed_stream{ s.pk_chunk_off, s.pk_ntiles, ACMHIP_FORM_PACKED });
			acmhip_stream_desc d{};
			d.idx_off = s.idx_off;
			d.hdr_off = s.hdr_off;
			d.pcm_off = s.pcm_off;
			d.level = s.info.level;
			d.rows = s.info.rows;
			d.nrows = s.info.blocks * s.info.rows;
			d.row_begin = 0;
			d.n_emit = items[i].words;
			for (acmhip_patch p : s.patches) {
				p.stream = (uint32_t)descs.size();
				patches.push_back(p);
			}
			descs.push_back(d);
			if (samples)
				*samples += d.n_emit;
		}
		if (ch.plan) {
			acmhip_plan_destroy(ch.plan);
			ch.plan = nullptr;
		}
		if (descs.empty())
			return ACMHIP_OK;
		if (!stage_packed && !stage_mform && !dev_mform)
			return acmhip_plan_create(dev, descs.data(), descs.size(), patches.data(), patches.size(), opts.plan_flags, &ch.plan);
		/* (a chunk's byte-plane streams are never launched without their form: no records over int16 rows they may not have) */
		int r = acmhip_plan_create_packed(dev, descs.data(), descs.size(), packed.data(), patches.data(), patches.size(),
						  opts.plan_flags | (stage_mform || dev_mform ? ACMHIP_PLAN_FORM_ONLY : 0u), &ch.plan);
		if (r == ACMHIP_OK)
			r = stage_mform || dev_mform ? acmhip_plan_bind_mform(ch.plan, d_pkblob, reinterpret_cast<const acmhip_mform_pair *>(d_pkchunk))
					: acmhip_plan_bind_packed(ch.plan, d_pkchunk, d_pkblob);
		return r;
	};
	/* chunks made of device-parsed streams only: their plans are cut NOW, from what the headers promise (a stream the
	 * device parser flags later gets its chunk's plan rebuilt), while the queues are still empty - a plan's small table
	 * uploads otherwise wait behind whatever long walk kernel shares their hardware queue */
	std::vector<char> planned(chunks.size(), 0), replan(chunks.size(), 0);
	/* block ranges: one plan per range over every stream's piece (a window of the stream that starts at the range's first row;
	 * the staged rows in front of it are on the device by then: earlier ranges).  Cut right behind the launch of the range's walk,
	 * which takes the device longer than the tables take the host; their upload is only queued (ACMHIP_PLAN_UPLOAD_ASYNC), so a
	 * table copy that shares a hardware queue with a walk kernel holds up nothing but the synthesis that waits for that walk anyway */
	auto cut_range_plan = [&](size_t r) -> int {
		std::vector<acmhip_stream_desc> descs;
		std::vector<acmhip_packed_stream> packed;
		for (size_t i = 0; i < n; i++) {
			const Slot &s = slots[i];
			if (!s.ok || piece_len[r * n + i] == 0)
				continue;
			packed.push_back(acmhip_packed_stream{ s.mf_pair_off, s.pk_ntiles, ACMHIP_FORM_BYTEPLANE });
			acmhip_stream_desc d{};
			d.idx_off = s.idx_off;
			d.hdr_off = s.hdr_off;
			d.pcm_off = piece_off[r * n + i];
			d.level = s.info.level;
			d.rows = s.info.rows;
			d.nrows = acmk_range_bound((uint32_t)s.need_blocks, (uint32_t)r + 1u, (uint32_t)R, s.range_unit) * s.info.rows;
			d.row_begin = acmk_range_bound((uint32_t)s.need_blocks, (uint32_t)r, (uint32_t)R, s.range_unit) * s.info.rows;
			d.n_emit = piece_len[r * n + i];
			descs.push_back(d);
		}
		if (descs.empty())
			return ACMHIP_OK;
		const unsigned pf = opts.plan_flags | ACMHIP_PLAN_UPLOAD_ASYNC;
		if (!dev_mform)
			return acmhip_plan_create(dev, descs.data(), descs.size(), nullptr, 0, pf, &rchunks[r].plan);
		/* every range is a plan of windows; a window of a stream the device stages in the byte-plane form goes to the lean kernels */
		const int pr = acmhip_plan_create_packed(dev, descs.data(), descs.size(), packed.data(), nullptr, 0, pf | ACMHIP_PLAN_FORM_ONLY, &rchunks[r].plan);
		return pr != ACMHIP_OK ? pr : acmhip_plan_bind_mform(rchunks[r].plan, d_pkblob, reinterpret_cast<const acmhip_mform_pair *>(d_pkchunk));
	};
	if (!groups.empty() && R == 1) {
		for (size_t c = 0; c < chunks.size(); c++) {
			bool all_dev = true;
			for (size_t i = chunks[c].first; i < chunks[c].last; i++)
				all_dev = all_dev && (!slots[i].ok || on_dev[i]);
			if (!all_dev)
				continue;
			for (size_t i = chunks[c].first; i < chunks[c].last; i++) {
				Slot &s = slots[i];
				if (!s.ok)
					continue;
				s.info.blocks = (uint32_t)s.need_blocks;
				items[i].words = deliverable_words(s.info.total_values, (uint64_t)s.info.rows * s.info.cols, s.info.channels, s.need_blocks);
			}
			BTRY(build_plan(c, nullptr));
			planned[c] = 1;
		}
		BNOTE("plans of the all-device chunks built");
	}

	/* 2c. every piece goes up as soon as its files are in the pinned arena; behind the last one, one walk over all streams */
	uint64_t max_columns = 0;
	hipEvent_t ev_parsed = nullptr;
	for (ParseGroup &g : groups) {
		if (g.k_first == g.k_last)
			continue;
		max_columns = std::max(max_columns, g.max_columns);
		ev_parsed = g.ev[2];
		if (R > 1)
			continue;               /* striped upload below */
		{
			std::unique_lock<std::mutex> lk(m);
			cv.wait(lk, [&]() { return g.uncopied.load() == 0; });
		}
		const size_t nj = g.k_last - g.k_first;
		HTRY(hipEventRecord(g.ev[0], st_up));
		HTRY(hipMemcpyAsync(d_files + g.file_begin, h_files + g.file_begin, g.file_end - g.file_begin, hipMemcpyHostToDevice, st_up));
		HTRY(hipMemcpyAsync(d_jobs + g.k_first * sizeof(AcmParseJob), h_jobs + g.k_first * sizeof(AcmParseJob), nj * sizeof(AcmParseJob),
				    hipMemcpyHostToDevice, st_up));
		HTRY(hipEventRecord(g.ev[1], st_up));
		if (&g == &groups.back() || (&g)[1].k_first == (&g)[1].k_last)
			BNOTE("files up to piece %zu queued for upload", (size_t)(&g - groups.data()));
	}
	if (ev_parsed) {
		AcmParseResult *d_res = reinterpret_cast<AcmParseResult *>(d_jobs + jobs_bytes);
		uint32_t *d_flags = reinterpret_cast<uint32_t *>(d_res + dev_ids.size());
		/* one block range: the walk + column kernels, and with R > 1 the synthesis and the read-back of what they staged */
		auto launch_range = [&](size_t r, size_t stripes_up) -> int {
			const int e = acmk_launch_parse_range_mf(reinterpret_cast<const AcmParseJob *>(d_jobs), (uint32_t)dev_ids.size(), d_files, d_colpos, d_idx,
								 d_hdr, d_res, d_flags, max_columns, (uint32_t)r, (uint32_t)R, (uint32_t)stripes_up,
								 dev_mform ? d_pkblob : nullptr, dev_mform ? reinterpret_cast<uint32_t *>(d_pkchunk) : nullptr,
								 dev_mform ? d_blkoff : nullptr, st_parse);
			if (e != 0)
				return acmhip_report_hip(e, "acmk_launch_parse_range");
			BNOTE("range %zu: walk queued", r);
			if (R == 1)
				return ACMHIP_OK;
			/* range r is staged: synthesise it and read it back while the walk goes on */
			Chunk &rg = rchunks[r];
			{
				const int cr = cut_range_plan(r);
				if (cr != ACMHIP_OK)
					return cr;
			}
			BNOTE("range %zu: plan cut", r);
#define RTRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return acmhip_report_hip((int)e_, #call); } while (0)
			RTRY(hipEventRecord(rg.ev[1], st_parse));
			RTRY(hipStreamWaitEvent(st_main, rg.ev[1], 0));
			RTRY(hipEventRecord(rg.ev[0], st_main));
			if (rg.plan) {
				const int pr = acmhip_plan_launch(rg.plan, d_idx, d_hdr, d_pcm, opts.fmt);
				if (pr != ACMHIP_OK)
					return pr;
			}
			BNOTE("range %zu: synthesis queued", r);
			RTRY(hipEventRecord(rg.ev[2], st_main));
			RTRY(hipStreamWaitEvent(st_copy, rg.ev[2], 0));
			RTRY(hipEventRecord(rg.ev[3], st_copy));
			if (rbase[r + 1] > rbase[r])
				RTRY(hipMemcpyAsync(h_pcm + rbase[r], d_pcm + rbase[r], (rbase[r + 1] - rbase[r]) * sizeof(int16_t), hipMemcpyDeviceToHost, st_copy));
			RTRY(hipEventRecord(rg.ev[4], st_copy));
#undef RTRY
			BNOTE("range %zu: walk, synthesis and read-back queued", r);
			{
				std::lock_guard<std::mutex> g(m);
				issued.store(r + 1, std::memory_order_release);
			}
			cv.notify_all();
			return ACMHIP_OK;
		};
		HTRY(hipMemsetAsync(d_flags, 0, dev_ids.size() * sizeof(uint32_t), st_parse));
		if (R == 1) {
			HTRY(hipEventRecord(ev_parsed, st_up));         /* re-recorded below: here it only orders the walk behind the last upload */
			HTRY(hipStreamWaitEvent(st_parse, ev_parsed, 0));
			BTRY(launch_range(0, 0));
		} else {
			/* striped upload: stripe s of every file as one transfer + a scatter kernel; the walk of range r may start once
			 * stripe r + MARGIN is in place (a range of blocks ends near the end of its stripe of the bits; a stream whose bit
			 * rate is so uneven that it needs more than the margin is stopped there and taken by the host reader) */
			HTRY(hipMemcpyAsync(d_jobs, h_jobs, jobs_bytes, hipMemcpyHostToDevice, st_up));
			HTRY(hipMemcpyAsync(d_jobs + stripe_tab_off, h_jobs + stripe_tab_off, stripe_tab_bytes, hipMemcpyHostToDevice, st_up));
			HTRY(hipEventRecord(ev_stripe[0], st_up));
			for (size_t sp = 0; sp < R; sp++) {
				{
					std::unique_lock<std::mutex> lk(m);
					cv.wait(lk, [&]() { return stripe_uncopied[sp].load() == 0; });
				}
				BNOTE("stripe %zu is in the pinned arena", sp);
				if (stripe_base[sp + 1] > stripe_base[sp])
					HTRY(hipMemcpyAsync(d_stage + stripe_base[sp], h_files + stripe_base[sp], stripe_base[sp + 1] - stripe_base[sp],
							    hipMemcpyHostToDevice, st_up));
				HTRY((hipError_t)acmk_launch_scatter_stripe(reinterpret_cast<const AcmParseJob *>(d_jobs), (uint32_t)nd,
									   reinterpret_cast<const uint64_t *>(d_jobs + stripe_tab_off), d_stage, d_files,
									   (uint32_t)sp, (uint32_t)R, st_up));
				HTRY(hipEventRecord(ev_stripe[sp + 1], st_up));
				/* range r is walked behind stripe r + MARGIN: its blocks end near the end of stripe r of the bits, and how
				 * near depends on how evenly the stream spends them (a corpus of short files: a range is a block or two) */
				constexpr size_t MARGIN = 2;
				if (sp + 1 < R) {
					if (sp >= MARGIN) {
						HTRY(hipStreamWaitEvent(st_parse, ev_stripe[sp + 1], 0));
						BTRY(launch_range(sp - MARGIN, sp + 1));
					}
				} else {
					BNOTE("last stripe of the files queued for upload");
					HTRY(hipStreamWaitEvent(st_parse, ev_stripe[R], 0));
					for (size_t r = R > MARGIN ? R - MARGIN - 1 : 0; r < R; r++)
						if (r + MARGIN + 1 >= R)            /* the ranges no earlier stripe has released */
							BTRY(launch_range(r, 0));
				}
			}
		}
		HTRY(hipMemcpyAsync(results, d_jobs + jobs_bytes, res_bytes, hipMemcpyDeviceToHost, st_parse));
		HTRY(hipEventRecord(ev_parsed, st_parse));
	}
	clk::time_point t_dev_parsed = t_alloc;
	/* what the device parser said; the streams it flags go through the exact host reader here */
	bool settled = false;
	std::vector<size_t> redo;               /* block ranges: streams the device flagged - decoded again from the host reader's staging */
	auto settle = [&]() -> int {
		if (settled || !ev_parsed)
			return ACMHIP_OK;
		settled = true;
		if (hipEventSynchronize(ev_parsed) != hipSuccess)
			return acmhip_report_hip((int)hipGetLastError(), "device parsing");
		t_dev_parsed = clk::now();
		BNOTE("device parsing done");
		for (size_t k = 0; k < dev_ids.size(); k++) {
			const size_t i = dev_ids[k];
			Slot &s = slots[i];
			if (R > 1 && (results[k].status != 0 || results[k].blocks_done != s.need_blocks || flags[k] != 0)) {
				s.pk_ntiles = 0;
				redo.push_back(i);
				continue;
			}
			if (results[k].status != 0 || results[k].blocks_done != s.need_blocks || flags[k] != 0) {
				if (!h_idx) {
					int r = acmhip_arena_get(dev, ACM_ARENA_H_IDX, idx_total * sizeof(int16_t), (void **)&h_idx);
					if (r == ACMHIP_OK)
						r = acmhip_arena_get(dev, ACM_ARENA_H_HDR, hdr_total * sizeof(acmhip_blkhdr), (void **)&h_hdr);
					if (r != ACMHIP_OK)
						return r;
				}
				s.pk_ntiles = 0;                /* whatever the device wrote of its byte-plane form is not to be read: the host stages int16 */
				host_stage(i);
				tm.host_parsed++;
				replan[s.chunk] = 1;
				continue;
			}
			s.info.blocks = (uint32_t)s.need_blocks;
			s.info.end_status = ACM_OK;
			items[i].status = ACM_OK;
			items[i].words = deliverable_words(s.info.total_values, (uint64_t)s.info.rows * s.info.cols,
							   s.info.channels, s.need_blocks);
			tm.device_parsed++;
		}
		return ACMHIP_OK;
	};

	/* uploads of many small pieces of one arena (a stream's byte-plane block, its int16 tail): a transfer call costs what ~256 KB cost
	 * on the wire, so pieces closer together than that travel as one, the bytes between them with them (unread on the device); a
	 * batch of 65 536 small streams is a few transfers per chunk instead of two per stream */
	struct Span { uint64_t at, len; };
	auto upload_spans = [&](uint8_t *d_base, const uint8_t *h_base, std::vector<Span> &spans) -> hipError_t {
		constexpr uint64_t JOIN = 256u << 10;
		for (size_t k = 0; k < spans.size();) {
			uint64_t at = spans[k].at, end = spans[k].at + spans[k].len, payload = spans[k].len;
			size_t j = k + 1;
			while (j < spans.size() && spans[j].at >= at && spans[j].at <= end + JOIN) {
				end = std::max(end, spans[j].at + spans[j].len);
				payload += spans[j].len;
				j++;
			}
			const hipError_t e = hipMemcpyAsync(d_base + at, h_base + at, end - at, hipMemcpyHostToDevice, st_main);
			if (e != hipSuccess)
				return e;
			tm.h2d_bytes += std::min(payload, end - at);      /* what the plans read; the gaps a joined transfer carries along are nobody's */
			k = j;
		}
		spans.clear();
		return hipSuccess;
	};
	std::vector<Span> spans;

	/* 3. this thread feeds the device, chunk by chunk (with block ranges the loop above has queued everything already) */
	for (size_t c = 0; c < chunks.size() && R == 1; c++) {
		Chunk &ch = chunks[c];
		if (ev_parsed) {
			BTRY(settle());
			if (c == 0)
				HTRY(hipStreamWaitEvent(st_main, ev_parsed, 0));
		}
		{
			std::unique_lock<std::mutex> g(m);
			cv.wait(g, [&]() { return ch.unparsed.load() == 0; });
		}
		if (!planned[c] || replan[c])
			BTRY(build_plan(c, nullptr));
		for (size_t i = ch.first; i < ch.last; i++)
			if (slots[i].ok)
				tm.samples += items[i].words;
		if (ch.plan) {
			HTRY(hipEventRecord(ch.ev[0], st_main));
			if (!dev_parse && (stage_packed || stage_mform)) {
				/* the second form of the chunk (packed: the used front of its blob region and its chunk-table entries; byte planes:
				 * the blocks of the streams that have one); of the int16 arena only what the other kernels read - streams without
				 * a second form, and behind a stream's whole tiles its ragged tail with the two rows in front of it */
				if (stage_mform) {
					size_t in_plan_mf = 0;
					for (size_t i = ch.first; i < ch.last; i++) {
						const Slot &s = slots[i];
						if (!s.ok || items[i].words == 0)
							continue;
						const size_t at = in_plan_mf++;
						if (!s.pk_ntiles)
							continue;
						uint64_t rows2 = 0;
						BTRY(acmhip_plan_form_rows(ch.plan, at, &rows2));
						if (!rows2)
							continue;               /* the plan reads this stream's int16 rows (see below): its byte-plane block stays here */
						spans.push_back(Span{ s.mf_off, s.mf_used });
					}
					HTRY(upload_spans(d_pkblob, h_pkblob, spans));
					/* the chunk's pair-table entries in one piece (a few bytes per thousand samples; entries of streams without the form
					 * travel with them unread) instead of a transfer per stream */
					if (ch.mf_pair_end > ch.mf_pair_begin) {
						HTRY(hipMemcpyAsync(reinterpret_cast<acmhip_mform_pair *>(d_pkchunk) + ch.mf_pair_begin,
								    reinterpret_cast<acmhip_mform_pair *>(h_pkchunk) + ch.mf_pair_begin,
								    (ch.mf_pair_end - ch.mf_pair_begin) * sizeof(acmhip_mform_pair), hipMemcpyHostToDevice, st_main));
						tm.h2d_bytes += (ch.mf_pair_end - ch.mf_pair_begin) * sizeof(acmhip_mform_pair);
					}
					tm.h2d_bytes += (ch.hdr_end - ch.hdr_begin) * sizeof(acmhip_blkhdr);
				} else {
					const uint64_t region = ch.idx_begin * sizeof(int16_t), region_len = (ch.idx_end - ch.idx_begin) * sizeof(int16_t);
					const uint64_t used = std::min<uint64_t>(ch.pk_used.load(), region_len);
					if (used)
						HTRY(hipMemcpyAsync(d_pkblob + region, h_pkblob + region, used, hipMemcpyHostToDevice, st_main));
					tm.h2d_bytes += used + (ch.pk_chunk_end - ch.pk_chunk_begin) * sizeof(acmhip_packed_chunk) + (ch.hdr_end - ch.hdr_begin) * sizeof(acmhip_blkhdr);
					if (ch.pk_chunk_end > ch.pk_chunk_begin)
						HTRY(hipMemcpyAsync(d_pkchunk + ch.pk_chunk_begin, h_pkchunk + ch.pk_chunk_begin,
								    (ch.pk_chunk_end - ch.pk_chunk_begin) * sizeof(acmhip_packed_chunk), hipMemcpyHostToDevice, st_main));
				}
				size_t in_plan = 0;            /* position among the descriptors build_plan() made the chunk's plan from */
				for (size_t i = ch.first; i < ch.last; i++) {
					const Slot &s = slots[i];
					if (!s.ok || items[i].words == 0)
						continue;
					const size_t at = in_plan++;
					if (s.info.blocks == 0)
						continue;
					const uint64_t cols = s.info.cols, nrows = (uint64_t)s.info.blocks * s.info.rows;
					uint64_t from_row = 0;
					if (s.pk_ntiles) {
						/* what the plan really takes from the second form - asked of the plan, not assumed: a level-13 / 14 stream of a
						 * small plan (or with ACM_PREFIX=0) goes to kernels that read the int16 rows from row 0 on */
						uint64_t rows2 = 0;
						BTRY(acmhip_plan_form_rows(ch.plan, at, &rows2));
						if (s.mf_fused && rows2 != (uint64_t)s.pk_ntiles * (uint64_t)acmhip_mform_tile_rows(s.info.level)) {
							/* the parsing pass wrote only the int16 rows behind the tiles it put into the form: a plan that reads fewer of
							 * them from the form would decode rows that were never staged.  Stager and planner cut by the same tile
							 * geometry of the same library build, and the shipped build reads no kernel-selection switch from the
							 * environment any more (ACM_K1_VARIANT made the two disagree), so this is an internal error,
							 * reported as text, not a condition to recover from */
							char msg[200];
							snprintf(msg, sizeof(msg), "acm_batch_decode: stream %zu: the plan reads %llu rows from the byte-plane form, %llu were staged that way",
								 i, (unsigned long long)rows2, (unsigned long long)s.pk_ntiles * (unsigned long long)acmhip_mform_tile_rows(s.info.level));
							acmhip_set_error_text(msg);
							rc = ACMHIP_ERR_ARG;
							cleanup();
							return rc;
						}
						if (rows2 && rows2 * cols >= items[i].words)
							continue;               /* nothing behind the whole tiles is emitted */
						from_row = rows2 >= 2 ? rows2 - 2 : 0;
					}
					spans.push_back(Span{ (s.idx_off + from_row * cols) * sizeof(int16_t), (nrows - from_row) * cols * sizeof(int16_t) });
				}
				HTRY(upload_spans(reinterpret_cast<uint8_t *>(d_idx), reinterpret_cast<const uint8_t *>(h_idx), spans));
				HTRY(hipMemcpyAsync(d_hdr + ch.hdr_begin, h_hdr + ch.hdr_begin, (ch.hdr_end - ch.hdr_begin) * sizeof(acmhip_blkhdr),
						    hipMemcpyHostToDevice, st_main));
			} else if (!dev_parse) {
				tm.h2d_bytes += (ch.idx_end - ch.idx_begin) * sizeof(int16_t) + (ch.hdr_end - ch.hdr_begin) * sizeof(acmhip_blkhdr);
				HTRY(hipMemcpyAsync(d_idx + ch.idx_begin, h_idx + ch.idx_begin, (ch.idx_end - ch.idx_begin) * sizeof(int16_t),
						    hipMemcpyHostToDevice, st_main));
				HTRY(hipMemcpyAsync(d_hdr + ch.hdr_begin, h_hdr + ch.hdr_begin, (ch.hdr_end - ch.hdr_begin) * sizeof(acmhip_blkhdr),
						    hipMemcpyHostToDevice, st_main));
			} else {
				for (size_t i = ch.first; i < ch.last; i++) {   /* only what the host reader staged itself */
					const Slot &s = slots[i];
					if (!s.ok || !s.host_staged || s.info.blocks == 0)
						continue;
					const uint64_t bl = (uint64_t)s.info.rows * s.info.cols;
					HTRY(hipMemcpyAsync(d_idx + s.idx_off, h_idx + s.idx_off, s.info.blocks * bl * sizeof(int16_t),
							    hipMemcpyHostToDevice, st_main));
					HTRY(hipMemcpyAsync(d_hdr + s.hdr_off, h_hdr + s.hdr_off, s.info.blocks * sizeof(acmhip_blkhdr),
							    hipMemcpyHostToDevice, st_main));
				}
			}
			HTRY(hipEventRecord(ch.ev[1], st_main));
			BTRY(acmhip_plan_launch(ch.plan, d_idx, d_hdr, d_pcm, opts.fmt));
			HTRY(hipEventRecord(ch.ev[2], st_main));
			HTRY(hipStreamWaitEvent(st_copy, ch.ev[2], 0));
			HTRY(hipEventRecord(ch.ev[3], st_copy));
			if (direct_out) {
				for (size_t i = ch.first; i < ch.last; i++) {
					const Slot &s = slots[i];
					if (!s.ok || !items[i].pcm || items[i].words == 0)
						continue;
					const uint64_t w = std::min<uint64_t>(items[i].words, items[i].pcm_cap);
					HTRY(hipMemcpyAsync(items[i].pcm, d_pcm + s.pcm_off, w * sizeof(int16_t), hipMemcpyDeviceToHost, st_copy));
				}
			} else if (!keep_on_device) {
				HTRY(hipMemcpyAsync(h_pcm + ch.idx_begin, d_pcm + ch.idx_begin, (ch.idx_end - ch.idx_begin) * sizeof(int16_t),
						    hipMemcpyDeviceToHost, st_copy));
			}
		}
		HTRY(hipEventRecord(ch.ev[4], st_copy));
		BNOTE("chunk %zu: synthesis and read-back queued", c);
		{
			std::lock_guard<std::mutex> g(m);
			issued.store(c + 1, std::memory_order_release);
		}
		cv.notify_all();
	}
	if (R > 1)
		BTRY(settle());
	pool.wait();
	pool_busy = false;
	BNOTE("pool done");
	HTRY(hipStreamSynchronize(st_copy));
	HTRY(hipStreamSynchronize(st_main));
	if (R > 1) {
		/* fix-up (rare): what the device parser flagged goes through the exact host reader and one more plan, stream by
		 * stream in the now idle arenas; its PCM replaces whatever the ranges copied out for these streams */
		if (!redo.empty()) {
			BTRY(acmhip_arena_get(dev, ACM_ARENA_H_IDX, idx_total * sizeof(int16_t), (void **)&h_idx));
			BTRY(acmhip_arena_get(dev, ACM_ARENA_H_HDR, hdr_total * sizeof(acmhip_blkhdr), (void **)&h_hdr));
			std::vector<acmhip_stream_desc> descs;
			std::vector<acmhip_patch> patches;
			std::vector<size_t> live;
			for (size_t i : redo) {
				host_stage(i);
				tm.host_parsed++;
				Slot &s = slots[i];
				if (!s.ok || items[i].words == 0 || s.info.blocks == 0)
					continue;
				const uint64_t bl = (uint64_t)s.info.rows * s.info.cols;
				HTRY(hipMemcpyAsync(d_idx + s.idx_off, h_idx + s.idx_off, s.info.blocks * bl * sizeof(int16_t), hipMemcpyHostToDevice, st_main));
				HTRY(hipMemcpyAsync(d_hdr + s.hdr_off, h_hdr + s.hdr_off, s.info.blocks * sizeof(acmhip_blkhdr), hipMemcpyHostToDevice, st_main));
				acmhip_stream_desc d{};
				d.idx_off = s.idx_off;
				d.hdr_off = s.hdr_off;
				d.pcm_off = s.pcm_off;
				d.level = s.info.level;
				d.rows = s.info.rows;
				d.nrows = s.info.blocks * s.info.rows;
				d.n_emit = items[i].words;
				for (acmhip_patch p : s.patches) {
					p.stream = (uint32_t)descs.size();
					patches.push_back(p);
				}
				descs.push_back(d);
				live.push_back(i);
			}
			if (!descs.empty()) {
				acmhip_plan *fix = nullptr;
				BTRY(acmhip_plan_create(dev, descs.data(), descs.size(), patches.data(), patches.size(), opts.plan_flags, &fix));
				rc = acmhip_plan_launch(fix, d_idx, d_hdr, d_pcm, opts.fmt);
				for (size_t i : live)
					if (rc == ACMHIP_OK && items[i].pcm) {
						const uint64_t w = std::min<uint64_t>(items[i].words, items[i].pcm_cap);
						if (hipMemcpyAsync(h_pcm + slots[i].pcm_off, d_pcm + slots[i].pcm_off, w * sizeof(int16_t), hipMemcpyDeviceToHost, st_main) != hipSuccess)
							rc = acmhip_report_hip((int)hipGetLastError(), "fix-up read-back");
					}
				if (rc == ACMHIP_OK && hipStreamSynchronize(st_main) != hipSuccess)
					rc = acmhip_report_hip((int)hipGetLastError(), "fix-up");
				acmhip_plan_destroy(fix);
				if (rc != ACMHIP_OK) {
					cleanup();
					return rc;
				}
				for (size_t i : live)
					if (items[i].pcm)
						memcpy(items[i].pcm, h_pcm + slots[i].pcm_off, std::min<uint64_t>(items[i].words, items[i].pcm_cap) * sizeof(int16_t));
			}
			BNOTE("fix-up of %zu flagged streams done", redo.size());
		}
		for (size_t i = 0; i < n; i++)
			if (slots[i].ok)
				tm.samples += items[i].words;
	}
#undef BTRY
#undef HTRY

	/* the phases overlap; report the device-side time each one took (summed over chunks) and the wall clock */
	for (Chunk &ch : chunks) {
		if (!ch.plan)
			continue;
		float ms = 0;
		if (hipEventElapsedTime(&ms, ch.ev[0], ch.ev[1]) == hipSuccess)
			tm.h2d_s += ms * 1e-3;
		if (hipEventElapsedTime(&ms, ch.ev[1], ch.ev[2]) == hipSuccess)
			tm.kernel_s += ms * 1e-3;
		if (hipEventElapsedTime(&ms, ch.ev[3], ch.ev[4]) == hipSuccess)
			tm.d2h_s += ms * 1e-3;
	}
	for (Chunk &ch : rchunks) {             /* block ranges: ev[0] .. ev[2] bracket the synthesis on the device stream */
		float ms = 0;
		if (ch.plan && hipEventElapsedTime(&ms, ch.ev[0], ch.ev[2]) == hipSuccess)
			tm.kernel_s += ms * 1e-3;
		if (hipEventElapsedTime(&ms, ch.ev[3], ch.ev[4]) == hipSuccess)
			tm.d2h_s += ms * 1e-3;
	}
	double h2d_files_s = 0;
	for (ParseGroup &g : groups) {
		float ms = 0;
		if (R == 1 && g.k_first != g.k_last && hipEventElapsedTime(&ms, g.ev[0], g.ev[1]) == hipSuccess)
			h2d_files_s += ms * 1e-3;
	}
	if (R > 1) {            /* striped upload: first stripe queued .. last stripe in place (the stripes wait for the pool's copies in between) */
		float ms = 0;
		if (hipEventElapsedTime(&ms, ev_stripe[0], ev_stripe[R]) == hipSuccess)
			h2d_files_s = ms * 1e-3;
	}
	tm.h2d_s += h2d_files_s;
	if (dev_parse)
		tm.h2d_bytes += files_total;
	for (const Slot &s : slots)
		tm.packed_streams += s.pk_ntiles != 0;
	/* host-side staging: headers, then until the last stream was parsed - by the pool, or by the device (whose walks
	 * overlap the uploads of the later groups and the synthesis / read-back of the earlier ones) */
	tm.stage_s = secs(t0, t_hdr) + std::max(secs(t_alloc, t_parsed), secs(t_alloc, t_dev_parsed));
	tm.total_s = secs(t0, clk::now());
	if (timing)
		*timing = tm;
	cleanup();
	return ACMHIP_OK;
#undef BNOTE
}
