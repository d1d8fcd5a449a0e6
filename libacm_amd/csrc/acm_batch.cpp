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

/* CPUs this process may actually use: the hardware count, capped by a cgroup-v2 CPU quota if there is one
 * (a 256-thread box with a 16-CPU quota parses slower with 256 threads than with 16) */
int usable_cpus()
{
	int n = (int)std::max(1u, std::thread::hardware_concurrency());
	if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
		long long quota = 0, period = 0;
		char q[32] = "";
		if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
			quota = atoll(q);
			if (quota > 0)
				n = std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
		}
		fclose(f);
	}
	return n;
}

template <typename F>
void parallel_for(size_t n, int threads, F fn)
{
	if (threads <= 0)
		threads = usable_cpus();
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
	if (opts.fmt > 3)
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

	int16_t *h_idx = nullptr, *h_pcm = nullptr, *d_idx = nullptr, *d_pcm = nullptr;
	acmhip_blkhdr *h_hdr = nullptr, *d_hdr = nullptr;
	acmhip_plan *plan = nullptr;
	int rc = ACMHIP_OK;
	auto cleanup = [&]() {
		acmhip_plan_destroy(plan);
		acmhip_free(dev, d_idx);
		acmhip_free(dev, d_hdr);
		acmhip_free(dev, d_pcm);
		acmhip_host_free(h_idx);
		acmhip_host_free(h_hdr);
		acmhip_host_free(h_pcm);
	};
#define BTRY(call) do { rc = (call); if (rc != ACMHIP_OK) { cleanup(); return rc; } } while (0)
	BTRY(acmhip_host_alloc(idx_total * sizeof(int16_t), (void **)&h_idx));
	BTRY(acmhip_host_alloc(hdr_total * sizeof(acmhip_blkhdr), (void **)&h_hdr));
	BTRY(acmhip_host_alloc(pcm_total * sizeof(int16_t), (void **)&h_pcm));
	BTRY(acmhip_malloc(dev, idx_total * sizeof(int16_t), (void **)&d_idx));
	BTRY(acmhip_malloc(dev, hdr_total * sizeof(acmhip_blkhdr), (void **)&d_hdr));
	BTRY(acmhip_malloc(dev, pcm_total * sizeof(int16_t), (void **)&d_pcm));

	/* 2. bit parsing, one stream per task */
	parallel_for(n, opts.threads, [&](size_t i) {
		Slot &s = slots[i];
		if (!s.ok)
			return;
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
	});
	const auto t1 = clk::now();
	tm.stage_s = secs(t0, t1);

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
	BTRY(acmhip_upload(dev, d_idx, h_idx, idx_total * sizeof(int16_t)));
	BTRY(acmhip_upload(dev, d_hdr, h_hdr, hdr_total * sizeof(acmhip_blkhdr)));
	BTRY(acmhip_device_sync(dev));
	const auto t2 = clk::now();
	tm.h2d_s = secs(t1, t2);
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
