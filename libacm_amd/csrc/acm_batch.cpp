/*
 * acm_batch.cpp - batch-of-files front end on one device (include/acm_hip.h).
 *
 * No counterpart in the reference (it decodes one stream at a time on one
 * thread); this is the piece BASELINE.json's north_star adds:
 * independent streams are bit-parsed by a pool of host threads straight into
 * one pinned staging arena, shipped to HBM in one copy, synthesised by one
 * acmhip_plan_launch, and the PCM comes back in one copy.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
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
	const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
	return std::min(hw, 64);
}

template <typename F>
void parallel_for(size_t n, int threads, F fn)
{
	if (threads <= 0)
		threads = default_threads();
	threads = (int)std::min<size_t>((size_t)threads, std::max<size_t>(1, n));
	std::atomic<size_t> next{ 0 };
	auto work = [&]() {
		for (size_t i; (i = next.fetch_add(1)) < n;)
			fn(i);
	};
	std::vector<std::thread> pool;
	for (int t = 1; t < threads; t++)
		pool.emplace_back(work);
	work();
	for (auto &t : pool)
		t.join();
}

inline uint64_t round_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

struct Slot {
	acm_stage_info info{};
	uint64_t need_blocks = 0;
	uint64_t idx_off = 0, hdr_off = 0, pcm_off = 0;
	std::vector<acmhip_patch> patches;
	bool ok = false;
};

} // namespace

extern "C" int acm_batch_decode(acmhip_device *dev, acm_batch_item *items, size_t n,
				const acm_batch_opts *opts_in, acm_batch_timing *timing)
{
	if (!dev || (n && !items))
		return ACMHIP_ERR_ARG;
	acm_batch_opts opts{};
	if (opts_in)
		opts = *opts_in;
	if (opts.fmt > 3 || opts.parse > ACM_BATCH_PARSE_DEVICE)
		return ACMHIP_ERR_ARG;
	acm_batch_timing tm{};
	const auto t0 = clk::now();

	/* 1. headers -> arena layout */
	std::vector<Slot> slots(n);
	parallel_for(n, opts.threads, [&](size_t i) {
		Slot &s = slots[i];
		acm_batch_item &it = items[i];
		it.words = 0;
		it.status = acm_stage_probe(it.data, it.len, opts.force_chans, &s.info);
		it.level = s.info.level;
		it.rows = s.info.rows;
		it.channels = s.info.channels;
		it.rate = s.info.rate;
		it.total_values = s.info.total_values;
		s.ok = (it.status == ACM_OK);
		if (s.ok) {
			const uint64_t bl = (uint64_t)s.info.rows * s.info.cols;
			s.need_blocks = ((uint64_t)s.info.total_values + bl - 1) / bl;
		}
	});
	uint64_t idx_total = 0, hdr_total = 0, pcm_total = 0;
	for (Slot &s : slots) {
		if (!s.ok)
			continue;
		const uint64_t bl = (uint64_t)s.info.rows * s.info.cols;
		s.idx_off = idx_total;
		s.hdr_off = hdr_total;
		s.pcm_off = pcm_total;
		idx_total += round_up(s.need_blocks * bl, 64);
		hdr_total += s.need_blocks;
		pcm_total += round_up(s.need_blocks * bl, 64);
	}

	const bool dev_parse = (opts.parse == ACM_BATCH_PARSE_DEVICE);
	uint64_t files_total = 0;
	std::vector<uint64_t> file_off;
	std::vector<size_t> dev_ids;                    /* streams handed to the device parser */
	if (dev_parse) {
		file_off.resize(n);
		for (size_t i = 0; i < n; i++) {
			const Slot &s = slots[i];
			if (!s.ok || s.need_blocks == 0 || items[i].len >= 0xFFFFFFF0u || !acmk_parse_supported(s.info.level, s.info.rows))
				continue;
			file_off[i] = files_total;
			files_total += round_up(items[i].len, 8) + 16;  /* zero tail: the device reader loads whole dwords */
			dev_ids.push_back(i);
		}
	}

	const auto t_hdr = clk::now();
	int16_t *h_idx = nullptr, *h_pcm = nullptr, *d_idx = nullptr, *d_pcm = nullptr, *d_idx_cm = nullptr;
	acmhip_blkhdr *h_hdr = nullptr, *d_hdr = nullptr;
	uint8_t *h_files = nullptr, *d_files = nullptr, *h_jobs = nullptr, *d_jobs = nullptr;
	acmhip_plan *plan = nullptr;
	int rc = ACMHIP_OK;
	auto cleanup = [&]() {
		acmhip_plan_destroy(plan);
		acmhip_arena_unlock(dev);
	};
#define BTRY(call) do { rc = (call); if (rc != ACMHIP_OK) { cleanup(); return rc; } } while (0)
	/* arenas live in the device handle and are reused by the next batch */
	acmhip_arena_lock(dev);
	BTRY(acmhip_arena_get(dev, ACM_ARENA_H_IDX, idx_total * sizeof(int16_t), (void **)&h_idx));
	BTRY(acmhip_arena_get(dev, ACM_ARENA_H_HDR, hdr_total * sizeof(acmhip_blkhdr), (void **)&h_hdr));
	BTRY(acmhip_arena_get(dev, ACM_ARENA_H_PCM, pcm_total * sizeof(int16_t), (void **)&h_pcm));
	BTRY(acmhip_arena_get(dev, ACM_ARENA_D_IDX, idx_total * sizeof(int16_t), (void **)&d_idx));
	BTRY(acmhip_arena_get(dev, ACM_ARENA_D_HDR, hdr_total * sizeof(acmhip_blkhdr), (void **)&d_hdr));
	BTRY(acmhip_arena_get(dev, ACM_ARENA_D_PCM, pcm_total * sizeof(int16_t), (void **)&d_pcm));
	const size_t jobs_bytes = round_up(dev_ids.size() * sizeof(AcmParseJob), 64);
	const size_t res_bytes = dev_ids.size() * sizeof(AcmParseResult);
	if (!dev_ids.empty()) {
		BTRY(acmhip_arena_get(dev, ACM_ARENA_H_FILES, files_total, (void **)&h_files));
		BTRY(acmhip_arena_get(dev, ACM_ARENA_D_FILES, files_total, (void **)&d_files));
		BTRY(acmhip_arena_get(dev, ACM_ARENA_D_IDX_CM, idx_total * sizeof(int16_t), (void **)&d_idx_cm));
		BTRY(acmhip_arena_get(dev, ACM_ARENA_H_JOBS, jobs_bytes + res_bytes, (void **)&h_jobs));
		BTRY(acmhip_arena_get(dev, ACM_ARENA_D_JOBS, jobs_bytes + res_bytes, (void **)&d_jobs));
	}
	const auto t_alloc = clk::now();
	tm.alloc_s = secs(t_hdr, t_alloc);

	/* 2. bit parsing, one stream per task: the exact host reader ... */
	auto host_stage = [&](size_t i) {
		Slot &s = slots[i];
		acm_batch_item &it = items[i];
		acm_stage_info info{};
		/* first pass counts patches (normally zero), second only if there are any */
		int r = acm_stage_file(it.data, it.len, opts.force_chans, h_idx + s.idx_off, h_hdr + s.hdr_off,
				       s.need_blocks, nullptr, 0, &info);
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
		it.status = info.end_status;
		it.words = deliverable_words(info.total_values, (uint64_t)info.rows * info.cols, info.channels, info.blocks);
	};
	std::vector<size_t> host_ids;                   /* streams the host reader stages */
	double h2d_files_s = 0;
	if (!dev_parse) {
		for (size_t i = 0; i < n; i++)
			if (slots[i].ok)
				host_ids.push_back(i);
	} else {
		/* ... or one device lane per stream, with the host reader behind it for everything unusual */
		std::vector<char> on_dev(n, 0);
		AcmParseJob *jobs = reinterpret_cast<AcmParseJob *>(h_jobs);
		AcmParseResult *results = reinterpret_cast<AcmParseResult *>(h_jobs + jobs_bytes);
		uint32_t max_blocks = 0, max_cols = 0;
		for (size_t k = 0; k < dev_ids.size(); k++) {
			const Slot &s = slots[dev_ids[k]];
			on_dev[dev_ids[k]] = 1;
			AcmParseJob &j = jobs[k];
			j.file_off = file_off[dev_ids[k]];
			j.idx_off = s.idx_off;
			j.hdr_off = s.hdr_off;
			j.file_len = (uint32_t)items[dev_ids[k]].len;
			j.data_start = (uint32_t)s.info.header_bytes;
			j.level = s.info.level;
			j.rows = s.info.rows;
			j.blocks = (uint32_t)s.need_blocks;
			j.pad = 0;
			max_blocks = std::max(max_blocks, j.blocks);
			max_cols = std::max(max_cols, s.info.cols);
		}
		parallel_for(dev_ids.size(), opts.threads, [&](size_t k) {
			const acm_batch_item &it = items[dev_ids[k]];
			uint8_t *dst = h_files + file_off[dev_ids[k]];
			memcpy(dst, it.data, it.len);
			memset(dst + it.len, 0, round_up(it.len, 8) + 16 - it.len);
		});
		for (size_t i = 0; i < n; i++)
			if (slots[i].ok && !on_dev[i])
				host_ids.push_back(i);
		if (!dev_ids.empty()) {
			const auto tu0 = clk::now();
			BTRY(acmhip_upload(dev, d_files, h_files, files_total));
			BTRY(acmhip_upload(dev, d_jobs, h_jobs, jobs_bytes));
			BTRY(acmhip_device_sync(dev));
			h2d_files_s = secs(tu0, clk::now());
			rc = acmk_launch_parse(reinterpret_cast<const AcmParseJob *>(d_jobs), (uint32_t)dev_ids.size(), d_files, d_idx_cm,
					       d_idx, d_hdr, reinterpret_cast<AcmParseResult *>(d_jobs + jobs_bytes), max_blocks,
					       max_cols, acmhip_device_stream(dev));
			if (rc != 0) {
				cleanup();
				return ACMHIP_ERR_HIP;
			}
			BTRY(acmhip_download(dev, results, d_jobs + jobs_bytes, res_bytes));
			BTRY(acmhip_device_sync(dev));
			for (size_t k = 0; k < dev_ids.size(); k++) {
				const size_t i = dev_ids[k];
				Slot &s = slots[i];
				if (results[k].status != 0 || results[k].blocks_done != s.need_blocks) {
					host_ids.push_back(i);
					continue;
				}
				s.info.blocks = (uint32_t)s.need_blocks;
				s.info.end_status = ACM_OK;
				items[i].status = ACM_OK;
				items[i].words = deliverable_words(s.info.total_values, (uint64_t)s.info.rows * s.info.cols,
								   s.info.channels, s.need_blocks);
				tm.device_parsed++;
			}
		}
	}
	tm.host_parsed = host_ids.size();
	parallel_for(host_ids.size(), opts.threads, [&](size_t k) { host_stage(host_ids[k]); });
	const auto t1 = clk::now();
	tm.stage_s = secs(t0, t_hdr) + secs(t_alloc, t1) - h2d_files_s;

	/* 3. descriptors */
	std::vector<acmhip_stream_desc> descs;
	std::vector<acmhip_patch> patches;
	std::vector<size_t> owner;
	for (size_t i = 0; i < n; i++) {
		Slot &s = slots[i];
		if (!s.ok || items[i].words == 0)
			continue;
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
		owner.push_back(i);
		tm.samples += d.n_emit;
	}

	/* 4. device round trip */
	if (!dev_parse) {
		BTRY(acmhip_upload(dev, d_idx, h_idx, idx_total * sizeof(int16_t)));
		BTRY(acmhip_upload(dev, d_hdr, h_hdr, hdr_total * sizeof(acmhip_blkhdr)));
	} else {
		for (size_t i : host_ids) {             /* only what the host reader had to stage itself */
			const Slot &s = slots[i];
			if (!s.ok || s.info.blocks == 0)
				continue;
			const uint64_t bl = (uint64_t)s.info.rows * s.info.cols;
			BTRY(acmhip_upload(dev, d_idx + s.idx_off, h_idx + s.idx_off, s.info.blocks * bl * sizeof(int16_t)));
			BTRY(acmhip_upload(dev, d_hdr + s.hdr_off, h_hdr + s.hdr_off, s.info.blocks * sizeof(acmhip_blkhdr)));
		}
	}
	BTRY(acmhip_device_sync(dev));
	const auto t2 = clk::now();
	tm.h2d_s = secs(t1, t2) + h2d_files_s;
	BTRY(acmhip_plan_create(dev, descs.data(), descs.size(), patches.data(), patches.size(), opts.plan_flags, &plan));
	BTRY(acmhip_plan_launch(plan, d_idx, d_hdr, d_pcm, opts.fmt));
	BTRY(acmhip_device_sync(dev));
	const auto t3 = clk::now();
	tm.kernel_s = secs(t2, t3);
	BTRY(acmhip_download(dev, h_pcm, d_pcm, pcm_total * sizeof(int16_t)));
	BTRY(acmhip_device_sync(dev));
	const auto t4 = clk::now();
	tm.d2h_s = secs(t3, t4);
#undef BTRY

	/* 5. hand the PCM out */
	parallel_for(owner.size(), opts.threads, [&](size_t k) {
		acm_batch_item &it = items[owner[k]];
		if (!it.pcm)
			return;
		const uint64_t w = std::min<uint64_t>(it.words, it.pcm_cap);
		memcpy(it.pcm, h_pcm + slots[owner[k]].pcm_off, w * sizeof(int16_t));
	});
	tm.total_s = secs(t0, clk::now());
	if (timing)
		*timing = tm;
	cleanup();
	return ACMHIP_OK;
}
