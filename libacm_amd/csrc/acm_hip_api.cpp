/*
 * acm_hip_api.cpp - device handle, memory plumbing and the launch planner
 * behind include/acm_hip.h.  The kernels are in acm_kernels.hip.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "acm_device.h"
#include "acm_hip.h"

#define ACM_K1_DEFAULT_VARIANT 0

namespace {

thread_local char g_err[512] = "";

void set_err(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

} // namespace
/* (for the other translation units of the library: the text acmhip_last_error() returns on this thread) */
extern "C" void acmhip_set_error_text(const char *text)
{
	set_err("%s", text);
}
namespace {

int hip_fail(hipError_t e, const char *what)
{
	set_err("%s: %s", what, hipGetErrorString(e));
	return ACMHIP_ERR_HIP;
}

#define HIPTRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail(e_, #call); } while (0)

} // namespace

struct acmhip_device {
	int ordinal;
	hipStream_t stream;
	bool own_stream;
	hipStream_t copy_stream = nullptr;      /* read-back stream of the batch pipeline, created on first use */
	hipStream_t side[2] = { nullptr, nullptr };     /* plans with several level groups spread them over these too */
	hipStream_t upload = nullptr;           /* plan tables go up on a non-blocking stream of their own (to_device) */
	hipStream_t aux[ACM_AUX_STREAMS] = {};  /* the batch pipeline's per-group upload + bit-parsing streams, created on first use */
	std::mutex upload_mutex;
	/* pinned staging ring of the table uploads: a copy out of pinned memory returns when it is queued, so a plan waits once for all its
	 * tables (or not at all: ACMHIP_PLAN_UPLOAD_ASYNC) instead of once per table */
	uint8_t *up_ring = nullptr;
	bool up_ring_failed = false;            /* pinning the ring failed once: tables go up straight from where they are */
	size_t up_at = 0;
	static constexpr size_t UP_RING_BYTES = (size_t)32 << 20;
	int cus = 0;                            /* compute units, sizes the persistent grids */
	void *arena[ACM_ARENA_SLOTS] = {};
	size_t arena_cap[ACM_ARENA_SLOTS] = {};
	std::mutex arena_mutex;
	/* device blocks of destroyed plans (tile tables, planes, the lead-in sink), kept for the next plan: hipMalloc / hipFree
	 * synchronise the whole device - every stream of the process, an RCCL transfer in flight included - and a batch builds and
	 * drops a plan per chunk */
	struct Block {
		void *ptr;
		size_t bytes;
	};
	std::vector<Block> spare;
	size_t spare_bytes = 0;
	std::mutex spare_mutex;
	static constexpr size_t SPARE_MAX_BYTES = (size_t)768 << 20, SPARE_MAX_BLOCKS = 256;
};

/* per level: the fused-kernel tile table, or the stage-wise stream list */
struct LevelGroup {
	uint32_t level = 0;
	AcmTile *d_tiles = nullptr;
	uint32_t ntiles = 0;
	bool carry = false;         /* tile table cut for the carry-mode kernel (no halo rows, ACM_TILE_* flags) */
	AcmTile2 *d_tiles2 = nullptr;   /* whole tiles of streams decoded from row 0: the lean kernel (acm_tile2) */
	uint32_t ntiles2 = 0;
	/* the whole tiles of the streams that came with a packed form: as records of the packed build (acm_tile2p; idx_off = packed tile
	 * number) and as plain acm_tile2 records over the int16 arena, which a launch uses while no packed arenas are bound */
	AcmTile2 *d_tiles2p = nullptr, *d_tiles2p_plain = nullptr;
	uint32_t ntiles2p = 0;
	/* the same for streams that came with a byte-plane form (acm_tile2's matrix-core build) */
	AcmTile2 *d_tiles2m = nullptr, *d_tiles2m_plain = nullptr;
	uint32_t ntiles2m = 0, ntiles2m_plain = 0;      /* the matrix-core build may cut the same rows into smaller tiles */
	AcmTile *d_tiles_extra = nullptr;   /* halo-flavour tiles that must not join a carry run (clean tiles of patched streams) */
	uint32_t ntiles_extra = 0;
	uint32_t *d_list = nullptr;
	uint32_t nlist = 0;
	uint64_t max_elems = 0;     /* stage-wise: longest plane run in the group */
	uint64_t max_emit = 0;
	bool prefix_patched = false;    /* a stream of the group has H1 patches: unpack, patch and stage 0 stay separate launches */
	uint32_t prefix_stages = 0; /* levels 13-15: the stage-wise kernels do level - 12 stages, the level-12 plane kernel the rest (d_tiles) */
};

struct acmhip_plan {
	acmhip_device *dev = nullptr;
	std::vector<acmhip_device::Block> blocks;       /* every device allocation of this plan (handed back to the handle's spare list) */
	AcmDevStream *d_streams = nullptr;
	std::vector<LevelGroup> fused, stagewise, small, prefix;   /* small: levels 0-4, one register-cascade launch; prefix: levels 13-15 */
	int16_t *d_sink = nullptr;             /* acm_tile2: where lead-in tiles store the PCM nobody wants */
	std::vector<uint64_t> form_rows;       /* per stream: rows the plan reads from the stream's second staged form (acmhip_plan_form_rows) */
	uint32_t *d_sw_all = nullptr;          /* every stage-wise stream, for the unpack launch */
	uint32_t n_sw_all = 0;
	uint64_t sw_max_elems = 0;
	AcmDevPatch *d_patches = nullptr;
	uint64_t npatches = 0;
	int32_t *d_plane[2] = { nullptr, nullptr };
	uint64_t plane_elems = 0;
	acmhip_plan_stats stats{};
	const acmhip_packed_chunk *pk_chunks = nullptr; /* acmhip_plan_bind_packed: device tables of the packed staged form */
	const uint8_t *pk_blob = nullptr;
	const uint8_t *mform = nullptr;                 /* acmhip_plan_bind_mform */
	const acmhip_mform_pair *mform_pairs = nullptr;
	int variant = 0;                        /* fused-kernel variant the tile tables were cut for */
	/* several tile-kernel groups (a corpus of mixed levels): their launches are independent, so they go round robin
	 * over the device stream and two side streams - the ramp-up and the tail of one launch overlap the next one's
	 * instead of leaving the chip half empty three times (fork / join by events on the device stream) */
	hipEvent_t ev_fork = nullptr, ev_join[2] = { nullptr, nullptr };
	hipEvent_t ev_upload = nullptr;         /* ACMHIP_PLAN_UPLOAD_ASYNC: the tables are on the device once this has happened */
	bool tables_settled = false;            /* upload_done() has run: waited for, or covered by ev_upload */
	bool form_only = false;                 /* ACMHIP_PLAN_FORM_ONLY: no int16 twin of the byte-plane records */
};

extern "C" const char *acmhip_last_error(void)
{
	return g_err;
}

extern "C" int acmhip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) {
		(void)hipGetLastError();
		return 0;
	}
	return n;
}

extern "C" int acmhip_device_open(int ordinal, void *hip_stream, acmhip_device **out)
{
	if (!out)
		return ACMHIP_ERR_ARG;
	int n = acmhip_device_count();
	if (n <= 0) {
		set_err("no usable HIP device (hipGetDeviceCount found none); this library has no CPU synthesis path");
		return ACMHIP_ERR_NO_DEVICE;
	}
	if (ordinal < 0 || ordinal >= n) {
		set_err("device ordinal %d out of range (have %d)", ordinal, n);
		return ACMHIP_ERR_ARG;
	}
	HIPTRY(hipSetDevice(ordinal));
	acmhip_device *d = new (std::nothrow) acmhip_device;
	if (!d)
		return ACMHIP_ERR_NOMEM;
	d->ordinal = ordinal;
	{
		hipDeviceProp_t prop;
		if (hipGetDeviceProperties(&prop, ordinal) == hipSuccess)
			d->cus = prop.multiProcessorCount;
	}
	d->own_stream = (hip_stream == nullptr);
	d->stream = (hipStream_t)hip_stream;
	if (d->own_stream) {
		hipError_t e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
		if (e != hipSuccess) {
			delete d;
			return hip_fail(e, "hipStreamCreateWithFlags");
		}
	}
	/* the streams of the batch pipeline that run concurrently - kernels (above), read-back, upload - are created here, one
	 * after the other: the runtime hands out its few hardware queues in creation order, so these three get one each as long
	 * as queues are left (created later, between whatever else the process creates, two of them may share one, and an
	 * upload then waits behind every read-back).  Failing here is not fatal: they are created on first use otherwise */
	if (hipStreamCreateWithFlags(&d->copy_stream, hipStreamNonBlocking) != hipSuccess)
		d->copy_stream = nullptr;
	if (hipStreamCreateWithFlags(&d->aux[ACM_AUX_STREAMS - 1], hipStreamNonBlocking) != hipSuccess)
		d->aux[ACM_AUX_STREAMS - 1] = nullptr;
	(void)hipGetLastError();
	*out = d;
	return ACMHIP_OK;
}

extern "C" void acmhip_device_close(acmhip_device *dev)
{
	if (!dev)
		return;
	(void)hipSetDevice(dev->ordinal);
	(void)hipStreamSynchronize(dev->stream);
	for (const acmhip_device::Block &b : dev->spare)
		(void)hipFree(b.ptr);
	for (int k = 0; k < ACM_ARENA_SLOTS; k++) {
		if (!dev->arena[k])
			continue;
		if (k < ACM_ARENA_D_IDX)
			(void)hipHostFree(dev->arena[k]);
		else
			(void)hipFree(dev->arena[k]);
	}
	if (dev->copy_stream)
		(void)hipStreamDestroy(dev->copy_stream);
	if (dev->upload)
		(void)hipStreamDestroy(dev->upload);
	if (dev->up_ring)
		(void)hipHostFree(dev->up_ring);
	for (hipStream_t s : dev->aux)
		if (s)
			(void)hipStreamDestroy(s);
	for (hipStream_t s : dev->side)
		if (s)
			(void)hipStreamDestroy(s);
	if (dev->own_stream)
		(void)hipStreamDestroy(dev->stream);
	delete dev;
}

extern "C" int acmhip_arena_get(acmhip_device *dev, int slot, size_t bytes, void **out)
{
	if (!dev || slot < 0 || slot >= ACM_ARENA_SLOTS || !out)
		return ACMHIP_ERR_ARG;
	if (bytes < 16)
		bytes = 16;
	if (dev->arena_cap[slot] < bytes) {
		HIPTRY(hipSetDevice(dev->ordinal));
		if (dev->arena[slot]) {
			HIPTRY(hipStreamSynchronize(dev->stream));
			if (slot < ACM_ARENA_D_IDX)
				(void)hipHostFree(dev->arena[slot]);
			else
				(void)hipFree(dev->arena[slot]);
			dev->arena[slot] = nullptr;
			dev->arena_cap[slot] = 0;
		}
		const size_t want = bytes + bytes / 8;          /* a little headroom against creeping batches */
		if (slot < ACM_ARENA_D_IDX)
			HIPTRY(hipHostMalloc(&dev->arena[slot], want, hipHostMallocDefault));
		else
			HIPTRY(hipMalloc(&dev->arena[slot], want));
		dev->arena_cap[slot] = want;
	}
	*out = dev->arena[slot];
	return ACMHIP_OK;
}

extern "C" int acmhip_copy_stream(acmhip_device *dev, void **out)
{
	if (!dev || !out)
		return ACMHIP_ERR_ARG;
	HIPTRY(hipSetDevice(dev->ordinal));             /* also makes the device current for the calling thread */
	if (!dev->copy_stream)
		HIPTRY(hipStreamCreateWithFlags(&dev->copy_stream, hipStreamNonBlocking));
	*out = (void *)dev->copy_stream;
	return ACMHIP_OK;
}

extern "C" int acmhip_aux_stream(acmhip_device *dev, int k, void **out)
{
	if (!dev || !out || k < 0 || k >= ACM_AUX_STREAMS)
		return ACMHIP_ERR_ARG;
	HIPTRY(hipSetDevice(dev->ordinal));
	if (!dev->aux[k])
		HIPTRY(hipStreamCreateWithFlags(&dev->aux[k], hipStreamNonBlocking));
	*out = (void *)dev->aux[k];
	return ACMHIP_OK;
}

extern "C" int acmhip_report_hip(int hip_error, const char *what)
{
	return hip_fail((hipError_t)hip_error, what);
}

extern "C" void acmhip_arena_lock(acmhip_device *dev)
{
	if (dev)
		dev->arena_mutex.lock();
}

extern "C" void acmhip_arena_unlock(acmhip_device *dev)
{
	if (dev)
		dev->arena_mutex.unlock();
}

extern "C" int acmhip_device_sync(acmhip_device *dev)
{
	if (!dev)
		return ACMHIP_ERR_ARG;
	HIPTRY(hipStreamSynchronize(dev->stream));
	return ACMHIP_OK;
}

extern "C" void *acmhip_device_stream(acmhip_device *dev)
{
	return dev ? (void *)dev->stream : nullptr;
}

extern "C" int acmhip_malloc(acmhip_device *dev, size_t bytes, void **dptr)
{
	if (!dev || !dptr)
		return ACMHIP_ERR_ARG;
	HIPTRY(hipSetDevice(dev->ordinal));
	HIPTRY(hipMalloc(dptr, bytes ? bytes : 16));
	return ACMHIP_OK;
}

extern "C" int acmhip_free(acmhip_device *dev, void *dptr)
{
	if (!dev)
		return ACMHIP_ERR_ARG;
	if (dptr)
		HIPTRY(hipFree(dptr));
	return ACMHIP_OK;
}

extern "C" int acmhip_host_alloc(size_t bytes, void **hptr)
{
	if (!hptr)
		return ACMHIP_ERR_ARG;
	if (acmhip_device_count() <= 0) {
		set_err("no usable HIP device: cannot allocate pinned host memory");
		return ACMHIP_ERR_NO_DEVICE;
	}
	HIPTRY(hipHostMalloc(hptr, bytes ? bytes : 16, hipHostMallocDefault));
	return ACMHIP_OK;
}

extern "C" int acmhip_host_free(void *hptr)
{
	if (hptr)
		HIPTRY(hipHostFree(hptr));
	return ACMHIP_OK;
}

extern "C" int acmhip_upload(acmhip_device *dev, void *dptr, const void *hptr, size_t bytes)
{
	if (!dev)
		return ACMHIP_ERR_ARG;
	if (bytes)
		HIPTRY(hipMemcpyAsync(dptr, hptr, bytes, hipMemcpyHostToDevice, dev->stream));
	return ACMHIP_OK;
}

extern "C" int acmhip_download(acmhip_device *dev, void *hptr, const void *dptr, size_t bytes)
{
	if (!dev)
		return ACMHIP_ERR_ARG;
	if (bytes)
		HIPTRY(hipMemcpyAsync(hptr, dptr, bytes, hipMemcpyDeviceToHost, dev->stream));
	return ACMHIP_OK;
}

extern "C" int acmhip_memset(acmhip_device *dev, void *dptr, int byte, size_t bytes)
{
	if (!dev)
		return ACMHIP_ERR_ARG;
	if (bytes)
		HIPTRY(hipMemsetAsync(dptr, byte, bytes, dev->stream));
	return ACMHIP_OK;
}

/* ------------------------------------------------------------------------ */

namespace {

/* device memory for a plan: a spare block of a destroyed plan if one fits (no more than twice the size), else hipMalloc */
int plan_malloc(acmhip_plan *pl, void **out, size_t bytes)
{
	acmhip_device *dev = pl->dev;
	*out = nullptr;
	if (bytes < 256)
		bytes = 256;
	{
		std::lock_guard<std::mutex> g(dev->spare_mutex);
		size_t best = dev->spare.size();
		for (size_t k = 0; k < dev->spare.size(); k++)
			if (dev->spare[k].bytes >= bytes && dev->spare[k].bytes <= 2 * bytes + 4096 &&
			    (best == dev->spare.size() || dev->spare[k].bytes < dev->spare[best].bytes))
				best = k;
		if (best != dev->spare.size()) {
			const acmhip_device::Block b = dev->spare[best];
			dev->spare.erase(dev->spare.begin() + (long)best);
			dev->spare_bytes -= b.bytes;
			pl->blocks.push_back(b);
			*out = b.ptr;
			return ACMHIP_OK;
		}
	}
	void *p = nullptr;
	HIPTRY(hipMalloc(&p, bytes));
	pl->blocks.push_back(acmhip_device::Block{ p, bytes });
	*out = p;
	return ACMHIP_OK;
}

template <typename T>
int to_device(acmhip_plan *pl, const std::vector<T> &v, T **out)
{
	acmhip_device *dev = pl->dev;
	*out = nullptr;
	if (v.empty())
		return ACMHIP_OK;
	{
		const int rc = plan_malloc(pl, (void **)out, v.size() * sizeof(T));
		if (rc != ACMHIP_OK)
			return rc;
	}
	/* on a non-blocking stream of the handle's own: neither the device stream - which may be the caller's and busy with the chunks of a
	 * batch still in flight (acm_batch_decode builds the plan of chunk k+1 while chunk k runs) - nor the legacy null stream, whose copies
	 * join every blocking stream of the process and are refused while another thread captures a graph, is involved.  The table goes
	 * through the pinned ring (the host vector dies with the caller) and is on the device when upload_done() has returned */
	const size_t bytes = (v.size() * sizeof(T) + 255) & ~(size_t)255;
	std::lock_guard<std::mutex> g(dev->upload_mutex);
	if (!dev->upload)
		HIPTRY(hipStreamCreateWithFlags(&dev->upload, hipStreamNonBlocking));
	/* (the ring is pinned when the first table of 256 KB or more comes by: a handle that only ever decodes a file or two - acmtool -d -
	 * does not pay for 32 MB of pinned memory, its few small tables go up straight from where they are) */
	if (!dev->up_ring && !dev->up_ring_failed && bytes >= ((size_t)256 << 10) && bytes <= acmhip_device::UP_RING_BYTES) {
		/* (no pinned memory to be had is not a reason to fail the plan: "no ring", the direct copy below does the same job - ADVICE r5) */
		if (hipHostMalloc((void **)&dev->up_ring, acmhip_device::UP_RING_BYTES, hipHostMallocDefault) != hipSuccess) {
			(void)hipGetLastError();
			dev->up_ring = nullptr;
			dev->up_ring_failed = true;
		}
	}
	if (!dev->up_ring || bytes > acmhip_device::UP_RING_BYTES) {
		HIPTRY(hipMemcpyAsync(*out, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, dev->upload));
		HIPTRY(hipStreamSynchronize(dev->upload));
		return ACMHIP_OK;
	}
	if (dev->up_at + bytes > acmhip_device::UP_RING_BYTES) {
		HIPTRY(hipStreamSynchronize(dev->upload));     /* the ring is free again once everything queued from it has gone */
		dev->up_at = 0;
	}
	memcpy(dev->up_ring + dev->up_at, v.data(), v.size() * sizeof(T));
	HIPTRY(hipMemcpyAsync(*out, dev->up_ring + dev->up_at, v.size() * sizeof(T), hipMemcpyHostToDevice, dev->upload));
	dev->up_at += bytes;
	return ACMHIP_OK;
}

/* the tables of a new plan are on the device: waited for here, or (async) left to an event the plan's launches wait for */
int upload_done(acmhip_plan *pl, bool async)
{
	acmhip_device *dev = pl->dev;
	std::lock_guard<std::mutex> g(dev->upload_mutex);
	if (!dev->upload) {
		pl->tables_settled = true;
		return ACMHIP_OK;
	}
	if (!async) {
		HIPTRY(hipStreamSynchronize(dev->upload));
		pl->tables_settled = true;
		return ACMHIP_OK;
	}
	HIPTRY(hipEventCreateWithFlags(&pl->ev_upload, hipEventDisableTiming));
	HIPTRY(hipEventRecord(pl->ev_upload, dev->upload));
	pl->tables_settled = true;
	return ACMHIP_OK;
}

bool fused_ok(const acmhip_stream_desc &s, int variant)
{
	return s.level >= ACM_K1_MIN_LEVEL && s.level <= ACM_K1_MAX_LEVEL && acmk_fused_tile_rows(s.level, variant) > 2;
}

/* Default: carry mode saves 2 of every tile_rows rows and costs one lead-in tile per workgroup, so it pays from tile_rows/2 tiles per
 * workgroup on; it is taken from tile_rows tiles per workgroup.  ACMHIP_PLAN_FORCE_HALO / _CARRY force a flavour (tests, measurements) */
bool carry_wanted(size_t ntiles, size_t grid, size_t tile_rows, unsigned flags)
{
	if (flags & (ACMHIP_PLAN_FORCE_HALO | ACMHIP_PLAN_FORCE_CARRY))
		return (flags & ACMHIP_PLAN_FORCE_CARRY) != 0;
	if (const char *e = ACM_TUNING_ENV("ACM_K1_CARRY"))
		return atoi(e) != 0;
	return grid > 0 && ntiles >= tile_rows * grid;
}

/* the lean kernels: -1 by tile count (default), 0 never (ACMHIP_PLAN_NO_LEAN), 1 whenever there is a whole tile (ACMHIP_PLAN_LEAN_ALWAYS) */
int lean_policy(unsigned flags)
{
	if (flags & ACMHIP_PLAN_NO_LEAN)
		return 0;
	if (flags & ACMHIP_PLAN_LEAN_ALWAYS)
		return 1;
	if (const char *e = ACM_TUNING_ENV("ACM_K2"))
		return atoi(e) != 0;
	return -1;
}

/* tuning builds: ACM_K1_VARIANT=n picks another built-in tile geometry (default: the measured-best one) */
int pick_variant()
{
	int v = ACM_K1_DEFAULT_VARIANT;
	if (const char *e = ACM_TUNING_ENV("ACM_K1_VARIANT"))
		v = atoi(e);
	if (v < 0 || v >= acmk_fused_variants())
		v = ACM_K1_DEFAULT_VARIANT;
	return v;
}

} // namespace

extern "C" void acmhip_plan_destroy(acmhip_plan *plan)
{
	if (!plan)
		return;
	if (plan->dev) {
		(void)hipSetDevice(plan->dev->ordinal);
		(void)hipStreamSynchronize(plan->dev->stream);   /* the launches of this plan are done with its tables (side streams join the device stream) */
		if (!plan->tables_settled && plan->dev->upload) {
			/* a plan whose creation failed half way has tables queued on the upload stream and no event that says when they have gone:
			 * wait for the stream itself before its blocks are handed on or freed */
			std::lock_guard<std::mutex> g(plan->dev->upload_mutex);
			(void)hipStreamSynchronize(plan->dev->upload);
		}
	}
	if (plan->ev_upload) {
		(void)hipEventSynchronize(plan->ev_upload);      /* (a plan dropped before it was launched: its blocks may be handed on) */
		(void)hipEventDestroy(plan->ev_upload);
	}
	if (plan->ev_fork)
		(void)hipEventDestroy(plan->ev_fork);
	for (hipEvent_t e : plan->ev_join)
		if (e)
			(void)hipEventDestroy(e);
	if (plan->dev) {
		acmhip_device *dev = plan->dev;
		std::vector<acmhip_device::Block> drop;
		{
			std::lock_guard<std::mutex> g(dev->spare_mutex);
			for (const acmhip_device::Block &b : plan->blocks) {
				if (dev->spare.size() < acmhip_device::SPARE_MAX_BLOCKS && dev->spare_bytes + b.bytes <= acmhip_device::SPARE_MAX_BYTES &&
				    b.bytes <= acmhip_device::SPARE_MAX_BYTES / 4) {
					dev->spare.push_back(b);
					dev->spare_bytes += b.bytes;
				} else {
					drop.push_back(b);
				}
			}
		}
		for (const acmhip_device::Block &b : drop)
			(void)hipFree(b.ptr);
	}
	delete plan;
}

extern "C" int acmhip_plan_create(acmhip_device *dev, const acmhip_stream_desc *streams, size_t n,
				  const acmhip_patch *patches, size_t npatches, unsigned flags,
				  acmhip_plan **out)
{
	return acmhip_plan_create_packed(dev, streams, n, nullptr, patches, npatches, flags, out);
}

extern "C" int acmhip_plan_create_packed(acmhip_device *dev, const acmhip_stream_desc *streams, size_t n, const acmhip_packed_stream *packed,
					 const acmhip_patch *patches, size_t npatches, unsigned flags, acmhip_plan **out)
{
	if (!dev || !out || (n && !streams) || (npatches && !patches) || n > 0xFFFFFFFFull)
		return ACMHIP_ERR_ARG;
	HIPTRY(hipSetDevice(dev->ordinal));

	std::vector<AcmDevStream> ds(n);
	std::vector<uint8_t> has_patch(n, 0);
	for (size_t p = 0; p < npatches; p++) {
		if (patches[p].stream >= n) {
			set_err("patch %zu names stream %u of %zu", p, patches[p].stream, n);
			return ACMHIP_ERR_ARG;
		}
		has_patch[patches[p].stream] = 1;
	}

	std::vector<std::vector<AcmTile>> tiles(16), tiles_carry(16), tiles_rest(16), tiles_extra(16);
	/* H1 patches (stale amplitude table): only the tiles that can see a patched sample leave the tile kernel - each of them
	 * becomes a window of its stream for the stage-wise kernels (a pseudo stream behind the n real ones) */
	struct PatchWindow { uint32_t stream; uint64_t lo_row, hi_row, scratch_off; };
	std::vector<PatchWindow> windows;
	std::vector<std::vector<uint64_t>> patch_rows(npatches ? n : 0);
	for (size_t p = 0; p < npatches; p++)
		patch_rows[patches[p].stream].push_back(patches[p].sample >> streams[patches[p].stream].level);
	for (auto &v : patch_rows)
		std::sort(v.begin(), v.end());
	std::vector<std::vector<AcmTile2>> tiles2(16), tiles2p(16), tiles2p_plain(16), tiles2m(16), tiles2m_plain(16);
	std::vector<uint64_t> form_rows(n, 0);         /* rows of every stream the plan reads from its second staged form */
	/* rows [0, rows2) of stream i as records of the lean tile kernel (T2 rows each), and, where the stream came with a second staged form,
	 * once more as records of the build that reads it */
	/* rows [row0, row0 + rows2) of stream i as records of the lean tile kernel (T2 rows each), and, where the stream came with a second
	 * staged form, once more as records of the build that reads it.  row0 = 0: the stream from its first row (the first record is
	 * ACM_TILE_FRESH); row0 > 0 (a multiple of T2; byte-plane streams only): a window - the records start with one for the tile in front
	 * of row0, marked ACM_TILE_DISCARD, which builds the carries and stores nothing */
	auto cut_lean = [&](size_t i, uint64_t row0, uint64_t rows2, uint32_t T2) -> int {
		const acmhip_stream_desc &s = streams[i];
		const uint32_t magic = s.rows == 1 ? 0u : (uint32_t)(((1ull << 32) + s.rows - 1) / s.rows);
		if (packed && packed[i].ntiles && packed[i].form > ACMHIP_FORM_BYTEPLANE) {
			set_err("stream %zu: staged form %u", i, packed[i].form);
			return ACMHIP_ERR_ARG;
		}
		const bool pk = packed && packed[i].ntiles && packed[i].form == ACMHIP_FORM_PACKED && acmk_tile2p_rows(s.level) == (int)T2 && row0 == 0;
		/* the matrix-core build may cut the same rows into smaller tiles (its rows in front cost nothing): T2M divides T2 */
		const uint32_t T2M = (uint32_t)acmk_tile2m_rows(s.level);
		const bool mf = packed && packed[i].ntiles && packed[i].form == ACMHIP_FORM_BYTEPLANE && T2M && T2 % T2M == 0;
		if ((pk && packed[i].ntiles < rows2 / T2) || (mf && packed[i].ntiles < (row0 + rows2) / T2M)) {
			set_err("stream %zu: %u tiles in its second staged form, %llu whole tiles to decode", i, packed[i].ntiles,
				(unsigned long long)((row0 + rows2) / (mf ? T2M : T2)));
			return ACMHIP_ERR_ARG;
		}
		if (row0 && (!mf || row0 % T2 || row0 < T2)) {
			set_err("stream %zu: a window on the lean kernels starts on a tile boundary of a stream with a byte-plane form", i);
			return ACMHIP_ERR_ARG;
		}
		if (pk || mf)
			form_rows[i] = rows2;
		std::vector<AcmTile2> &plain = pk ? tiles2p_plain[s.level] : mf ? tiles2m_plain[s.level] : tiles2[s.level];
		auto flags_of = [&](uint64_t r, bool lead_in) -> uint32_t {
			/* (a lead-in that is the stream's first tile has nothing in front of it either: both flags) */
			return (lead_in ? ACM_TILE_DISCARD : 0u) | (r == 0 ? ACM_TILE_FRESH : 0u);
		};
		/* where sample (row r, column 0) goes: rows count from the window's first row (a lead-in stores into the sink) */
		auto pcm_of = [&](uint64_t r) -> uint64_t { return s.pcm_off + ((r >= row0 ? r - row0 : 0) << s.level); };
		/* (a batch of small streams cuts millions of records, on the thread that feeds the device: block and row of a record by
		 * 32-bit division - rows < 2^32 - and room in the vectors made per stream, not per record) */
		const uint64_t first = row0 ? row0 - T2 : 0, ntile = (row0 + rows2 - first) / T2;
		const uint64_t lead_rows = mf ? (uint64_t)T2M * (uint64_t)acmk_tile2m_lead_in(s.level) : 0;      /* (<= T2: two rows more at most) */
		if (row0 && lead_rows > T2) {
			/* the records of a window's lead-in are cut from ONE tile in front of it; a build whose first pass needs more rows than
			 * that in front (level 14: two tiles of two rows) cannot take a window - its streams only ever come here from row 0
			 * (refused loudly instead of decoded wrong, should a caller ever ask) */
			set_err("stream %zu: a window on the lean kernel of level %u needs %llu rows in front, a tile has %u", i, s.level,
				(unsigned long long)lead_rows, T2);
			return ACMHIP_ERR_ARG;
		}
		const uint32_t pk_slots = pk ? (uint32_t)acmk_tile2p_slots(s.level) : 0;
		const bool twin = !(mf && (flags & ACMHIP_PLAN_FORM_ONLY));
		/* (room for this stream's records in one step - doubling, or every stream would move the whole table) */
		auto room = [](std::vector<AcmTile2> &v, uint64_t more) {
			if (v.capacity() < v.size() + more)
				v.reserve(std::max<size_t>(2 * v.capacity(), v.size() + more));
		};
		if (twin)
			room(plain, ntile);
		if (pk)
			room(tiles2p[s.level], ntile);
		std::vector<AcmTile2> &mtab = tiles2m[s.level];
		if (mf)
			room(mtab, ntile * (T2 / T2M));
		const uint32_t srows = s.rows;
		for (uint64_t r = first; r < row0 + rows2; r += T2) {
			const bool lead_in = r < row0;
			const uint32_t rh = (uint32_t)(r >= 2 ? r - 2 : 0);         /* the row the row-value fetch counts from */
			if (twin)
				plain.push_back(AcmTile2{ s.idx_off + (r << s.level), pcm_of(r),
							  (uint32_t)(s.hdr_off + rh / srows), rh % srows, magic, flags_of(r, lead_in) });
			if (pk)
				tiles2p[s.level].push_back(AcmTile2{ packed[i].chunk_off + r / T2 * (uint64_t)pk_slots, s.pcm_off + (r << s.level),
								     (uint32_t)(s.hdr_off + (uint32_t)r / srows), (uint32_t)r % srows, magic,
								     r == 0 ? ACM_TILE_FRESH : 0u });
			/* a byte-plane tile is named by the pair-table entry of the row pair in front of it (entry 0 of a stream: the pair of zeros).
			 * (chunks of one row - T2M == 1, the chunk kernel at levels 11 and 12 - also say whether they start a pair and whether they
			 * are row 1; of a lead-in tile only the last chunks are needed, acmk_tile2m_lead_in of them) */
			if (!mf)
				continue;
			uint64_t rm = lead_in ? r + T2 - std::min<uint64_t>(T2, lead_rows) : r;
			const uint32_t rhm0 = (uint32_t)(rm >= 2 ? rm - 2 : 0);
			uint32_t blk = rhm0 / srows, pos = rhm0 % srows;        /* of row rm - 2, carried along from chunk to chunk */
			for (; rm < r + T2; rm += T2M) {
				/* rows in reach: max(rm - 2, 0) - the row (blk, pos) names - through rm + T2M - 1 */
				const uint64_t span = rm + T2M - 1 - (rm >= 2 ? rm - 2 : 0);
				mtab.push_back(AcmTile2{ packed[i].chunk_off + rm / 2, pcm_of(rm), (uint32_t)s.hdr_off + blk, pos, magic,
							 flags_of(rm, lead_in) | (rm == 1 ? ACM_TILE_ROW1 : 0u) | ((rm & 1) ? ACM_TILE_ODD : 0u) |
							 (pos + span < srows ? ACM_TILE_ONEBLOCK : 0u) });
				/* the next chunk's row rm + T2M - 2 (rows 0 and 1 of a stream both count from row 0) */
				const uint32_t step = rm >= 2 ? T2M : rm + T2M >= 2 ? (uint32_t)(rm + T2M - 2) : 0u;
				pos += step;
				while (pos >= srows) {
					pos -= srows;
					blk++;
				}
			}
		}
		return ACMHIP_OK;
	};
	const int lean = lean_policy(flags);
	const bool k2_allowed = lean != 0;
	std::vector<std::vector<uint32_t>> lists(16), small_lists(ACM_SMALL_MAX_LEVEL + 1), prefix_lists(16);
	std::vector<std::vector<AcmTile>> prefix_tiles(16), prefix_tiles_carry(16);
	std::vector<uint8_t> plane_shift(n, 0);                /* levels 13-15: planes carry values scaled by 2^(16 - level) */
	std::vector<uint8_t> on_tile_kernel(n, 0);             /* patched stream that stays on the tile kernel: its patches live in windows only */
	const bool prefix_allowed = !(flags & ACMHIP_PLAN_STAGEWISE) && !(ACM_TUNING_ENV("ACM_PREFIX") && atoi(ACM_TUNING_ENV("ACM_PREFIX")) == 0);
	std::vector<uint64_t> grp_max_elems(16, 0), grp_max_emit(16, 0);
	std::vector<uint32_t> sw_all;
	uint64_t plane = 0, sw_max = 0;
	acmhip_plan_stats st{};
	const int variant = pick_variant();

	/* levels 13 and 14: four rows / two rows are one 128 KB tile of the lean kernel.  Whether a batch is worth its one
	 * lead-in tile per workgroup is known only from all its streams: counted here, decided before the streams are cut */
	bool k2_high[16] = {};
	for (uint32_t lv = ACM_K1_MAX_LEVEL + 1; lv <= ACM_K2_MAX_LEVEL; lv++) {
		const uint32_t TH = (uint32_t)acmk_tile2_rows(lv);
		uint64_t whole = 0;
		for (size_t i = 0; i < n && TH; i++) {
			const acmhip_stream_desc &s = streams[i];
			if (s.level == lv && s.row_begin == 0 && !has_patch[i])
				whole += std::min<uint64_t>(s.nrows, s.n_emit >> lv) / TH;
		}
		const size_t gridh = (size_t)acmk_tile2_grid(lv, dev->cus);
		k2_high[lv] = k2_allowed && prefix_allowed && TH && gridh && (lean == 1 ? whole > 0 : whole >= 8 * gridh);
	}

	for (size_t i = 0; i < n; i++) {
		const acmhip_stream_desc &s = streams[i];
		if (s.level > 15 || s.rows == 0 || s.rows > 4095 || (s.idx_off & 7) || (s.pcm_off & 7) ||
		    s.row_begin > s.nrows ||
		    s.n_emit > ((uint64_t)(s.nrows - s.row_begin) << s.level)) {
			set_err("stream %zu: invalid descriptor (level %u rows %u nrows %u row_begin %u n_emit %llu idx_off %llu pcm_off %llu)",
				i, s.level, s.rows, s.nrows, s.row_begin, (unsigned long long)s.n_emit,
				(unsigned long long)s.idx_off, (unsigned long long)s.pcm_off);
			return ACMHIP_ERR_ARG;
		}
		ds[i] = AcmDevStream{};
		AcmDevStream d{};               /* filled here, stored below: ds grows while windows are cut */
		d.idx_off = s.idx_off;
		d.hdr_off = s.hdr_off;
		d.pcm_off = s.pcm_off;
		d.n_emit = s.n_emit;
		d.level = s.level;
		d.rows = s.rows;
		d.nrows = s.nrows;
		d.row_begin = s.row_begin;
		d.halo_row = s.row_begin >= 2 ? s.row_begin - 2 : 0;
		d.scratch_off = 0;
		d.pad = 0;
		st.samples += s.n_emit;
		ds[i] = d;
		if (s.n_emit == 0)
			continue;

		const bool fused = !(flags & ACMHIP_PLAN_STAGEWISE) && fused_ok(s, variant);
		if (fused && has_patch[i]) {
			on_tile_kernel[i] = 1;
			const uint32_t T = (uint32_t)acmk_fused_tile_rows(s.level, variant) - 2;
			const uint64_t cols = 1ull << s.level;
			const uint64_t emit_rows = (s.n_emit + cols - 1) >> s.level;
			const std::vector<uint64_t> &pr = patch_rows[i];
			for (uint64_t r = 0; r < emit_rows; r += T) {
				const uint64_t row0 = s.row_begin + r;
				const uint64_t lo = row0 >= 2 ? row0 - 2 : 0, hi = std::min<uint64_t>(row0 + T, s.nrows);
				auto it = std::lower_bound(pr.begin(), pr.end(), lo);
				if (it == pr.end() || *it >= hi) {
					tiles_extra[s.level].push_back(AcmTile{ (uint32_t)i, (int32_t)row0, 0u, 0u });   /* clean tile: halo flavour */
					continue;
				}
				AcmDevStream w = d;
				w.pcm_off = s.pcm_off + (r << s.level);
				w.n_emit = std::min<uint64_t>((uint64_t)T << s.level, s.n_emit - (r << s.level));
				w.row_begin = (uint32_t)row0;
				w.halo_row = (uint32_t)lo;
				w.nrows = (uint32_t)hi;
				const uint64_t elems = (hi - lo) << s.level;
				w.scratch_off = plane;
				plane += (elems + 63) & ~63ull;
				windows.push_back(PatchWindow{ (uint32_t)i, lo, hi, w.scratch_off });
				const uint32_t id = (uint32_t)ds.size();
				ds.push_back(w);
				lists[s.level].push_back(id);
				sw_all.push_back(id);
				grp_max_elems[s.level] = std::max(grp_max_elems[s.level], elems);
				grp_max_emit[s.level] = std::max(grp_max_emit[s.level], (uint64_t)w.n_emit);
				sw_max = std::max(sw_max, elems);
			}
			st.fused_streams++;
		} else if (fused) {
			const uint32_t T = (uint32_t)acmk_fused_tile_rows(s.level, variant) - 2;
			const uint64_t cols = 1ull << s.level;
			const uint64_t emit_rows = (s.n_emit + cols - 1) >> s.level;
			for (uint64_t r = 0; r < emit_rows; r += T)
				tiles[s.level].push_back(AcmTile{ (uint32_t)i, (int32_t)(s.row_begin + r), 0u, 0u });
			/* the lean kernel takes the whole tiles of a stream that is decoded from its row 0; the ragged tail
			 * (and every other kind of stream) stays with the general kernel, as halo tiles */
			const uint32_t T2 = (uint32_t)acmk_tile2_rows(s.level);
			uint64_t rows2 = 0;
			if (k2_allowed && T2 && s.row_begin == 0) {
				const uint64_t full_rows = std::min<uint64_t>(s.nrows, s.n_emit >> s.level);
				rows2 = full_rows / T2 * T2;
				const int cr = cut_lean(i, 0, rows2, T2);
				if (cr != ACMHIP_OK)
					return cr;
			} else if (k2_allowed && T2 && s.row_begin >= T2 && s.row_begin % T2 == 0 && packed && packed[i].ntiles &&
				   packed[i].form == ACMHIP_FORM_BYTEPLANE && acmk_tile2m_rows(s.level) > 0 && T2 % (uint32_t)acmk_tile2m_rows(s.level) == 0) {
				/* a window that starts on a tile boundary of a stream with a byte-plane form (a block range of a device-parsed batch):
				 * its whole tiles go to the lean kernels too, behind a lead-in record for the tile in front of it */
				const uint64_t full_rows = std::min<uint64_t>(s.nrows - s.row_begin, s.n_emit >> s.level);
				rows2 = full_rows / T2 * T2;
				if (rows2) {
					const int cr = cut_lean(i, s.row_begin, rows2, T2);
					if (cr != ACMHIP_OK)
						return cr;
				}
			}
			for (uint64_t r = rows2; r < emit_rows; r += T)
				tiles_rest[s.level].push_back(AcmTile{ (uint32_t)i, (int32_t)(s.row_begin + r), 0u, 0u });
			if (acmk_fused_has_carry(s.level, variant)) {
				/* carry mode: T + 2 payload rows per tile; a stream that does not start at its row 0 gets a
				 * lead-in tile in front (rows that do not exist count as zeros, which is exact: no output
				 * depends on anything further back than two rows) */
				const uint32_t TC = T + 2;
				std::vector<AcmTile> &tc = tiles_carry[s.level];
				if (s.row_begin > 0)
					tc.push_back(AcmTile{ (uint32_t)i, (int32_t)s.row_begin - (int32_t)TC, ACM_TILE_FRESH | ACM_TILE_DISCARD, 0u });
				for (uint64_t r = 0; r < emit_rows; r += TC)
					tc.push_back(AcmTile{ (uint32_t)i, (int32_t)(s.row_begin + r), (r == 0 && s.row_begin == 0) ? ACM_TILE_FRESH : 0u, 0u });
			}
			st.fused_streams++;
		} else if (!(flags & ACMHIP_PLAN_STAGEWISE) && s.level <= ACM_SMALL_MAX_LEVEL && !has_patch[i]) {
			small_lists[s.level].push_back((uint32_t)i);
			grp_max_emit[s.level] = std::max(grp_max_emit[s.level], (uint64_t)s.n_emit);
			st.fused_streams++;
		} else if (prefix_allowed && s.level > ACM_K1_MAX_LEVEL) {
			uint32_t sid = (uint32_t)i;             /* the stream the prefix + plane pair works on: this one, or the rest of it */
			AcmDevStream src = d;
			if (k2_high[s.level] && s.row_begin == 0 && !has_patch[i]) {
				/* the whole tiles go to the lean tile kernel; whatever is left - the ragged tail - is a window of the
				 * stream (a pseudo stream behind the n real ones) for the prefix + plane pair */
				const uint32_t T2 = (uint32_t)acmk_tile2_rows(s.level);
				const uint64_t rows2 = std::min<uint64_t>(s.nrows, s.n_emit >> s.level) / T2 * T2;
				const int cr = cut_lean(i, 0, rows2, T2);
				if (cr != ACMHIP_OK)
					return cr;
				if (rows2 > 0) {
					src.pcm_off = s.pcm_off + (rows2 << s.level);
					src.n_emit = s.n_emit - (rows2 << s.level);
					src.row_begin = (uint32_t)rows2;
					src.halo_row = (uint32_t)rows2 - 2;
					if (src.n_emit) {
						sid = (uint32_t)ds.size();
						ds.push_back(src);
					}
				}
			}
			if (src.n_emit) {
			/* levels 13-15: level - 12 stages by the stage-wise kernels into a plane (scaled, see acmk_launch_unpack), then the
			 * plane is a level-12 stream of 2^(level-12) times as many rows for the tile kernel: stage k of level L has the
			 * stride of stage k - j of level L - j, and the "+1" belongs to stage 0 alone (decode.c:555-571) */
			const uint32_t j = s.level - 12;
			const uint64_t elems = (uint64_t)(src.nrows - src.halo_row) << s.level;
			ds[sid].scratch_off = plane;
			if (sid < n)
				plane_shift[sid] = (uint8_t)(16 - s.level);
			AcmDevStream w{};
			w.idx_off = plane;                              /* int32 units into the plane */
			w.pcm_off = src.pcm_off;
			w.n_emit = src.n_emit;
			w.level = 12;
			w.rows = 1;
			w.nrows = (src.nrows - src.halo_row) << j;
			w.row_begin = (src.row_begin - src.halo_row) << j;
			w.halo_row = w.row_begin >= 2 ? w.row_begin - 2 : 0;
			plane += (elems + 63) & ~63ull;
			const uint32_t id = (uint32_t)ds.size();
			ds.push_back(w);
			const uint32_t T12 = (uint32_t)acmk_plane_tile_rows() - 2;
			const uint64_t emit_rows = (src.n_emit + 4095) >> 12;
			for (uint64_t r = 0; r < emit_rows; r += T12)
				prefix_tiles[s.level].push_back(AcmTile{ id, (int32_t)(w.row_begin + r), 0u, 0u });
			{
				/* the carry-mode table of the same stream (see tiles_carry above) */
				const uint32_t TC = T12 + 2;
				std::vector<AcmTile> &tc = prefix_tiles_carry[s.level];
				if (w.row_begin > 0)
					tc.push_back(AcmTile{ id, (int32_t)w.row_begin - (int32_t)TC, ACM_TILE_FRESH | ACM_TILE_DISCARD, 0u });
				for (uint64_t r = 0; r < emit_rows; r += TC)
					tc.push_back(AcmTile{ id, (int32_t)(w.row_begin + r), (r == 0 && w.row_begin == 0) ? ACM_TILE_FRESH : 0u, 0u });
			}
			prefix_lists[s.level].push_back(sid);
			grp_max_elems[s.level] = std::max(grp_max_elems[s.level], elems);
			sw_max = std::max(sw_max, elems);
			}
			st.fused_streams++;
		} else {
			const uint64_t elems = (uint64_t)(s.nrows - d.halo_row) << s.level;
			d.scratch_off = plane;
			ds[i].scratch_off = plane;
			plane += (elems + 63) & ~63ull;
			lists[s.level].push_back((uint32_t)i);
			sw_all.push_back((uint32_t)i);
			grp_max_elems[s.level] = std::max(grp_max_elems[s.level], elems);
			grp_max_emit[s.level] = std::max(grp_max_emit[s.level], (uint64_t)s.n_emit);
			sw_max = std::max(sw_max, elems);
			st.stagewise_streams++;
		}
	}

#ifdef ACM_TUNING
	/* EXPERIMENT (ACM_K3_SEG=S; measured and not kept, profiles/r5_placement.txt): the chunk kernel's table in time-major order - wavefront v's run is segments v, v + W, v + 2 W, ... of S
	 * chunks each (W wavefronts), every segment behind lead-in records for the chunks in front of it: at any moment the W wavefronts
	 * work inside one window of W x S chunks that moves through the arenas, instead of all over them */
	if (const char *e = ACM_TUNING_ENV("ACM_K3_SEG")) {
		const size_t S = (size_t)atoi(e);
		for (uint32_t lv = 0; lv < 16 && S > 0; lv++) {
			const size_t W = (size_t)acmk_tile2m_run_waves(lv, dev->cus);
			std::vector<AcmTile2> &T = tiles2m[lv];
			bool plainly_cut = W > 0 && !T.empty();
			for (const AcmTile2 &r : T)
				plainly_cut = plainly_cut && !(r.flags & ACM_TILE_DISCARD);
			if (!plainly_cut)
				continue;
			const size_t lead = (size_t)acmk_tile2m_lead_in(lv);
			std::vector<size_t> seg;                /* first record of every segment */
			for (size_t k = 0, in_seg = 0; k < T.size(); k++, in_seg++)
				if (k == 0 || (T[k].flags & ACM_TILE_FRESH) || in_seg == S) {
					seg.push_back(k);
					in_seg = 0;
				}
			seg.push_back(T.size());
			const size_t nseg = seg.size() - 1;
			std::vector<AcmTile2> N;
			N.reserve(T.size() + nseg * lead);
			for (size_t v = 0; v < W; v++)
				for (size_t g = v; g < nseg; g += W) {
					const size_t a = seg[g], b = seg[g + 1];
					if (!(T[a].flags & ACM_TILE_FRESH)) {
						size_t from = a;
						for (size_t n = 0; n < lead && from > 0; n++) {
							from--;
							if (T[from].flags & ACM_TILE_FRESH)
								break;
						}
						for (size_t k = from; k < a; k++) {
							AcmTile2 r = T[k];
							r.flags |= ACM_TILE_DISCARD;
							N.push_back(r);
						}
					}
					N.insert(N.end(), T.begin() + (long)a, T.begin() + (long)b);
				}
			T.swap(N);
		}
	}
#endif

	/* H1 patches -> plane coordinates */
	std::vector<AcmDevPatch> dp;
	std::vector<std::vector<size_t>> win_of(windows.empty() ? 0 : n);
	for (size_t w = 0; w < windows.size(); w++)
		win_of[windows[w].stream].push_back(w);
	for (size_t p = 0; p < npatches; p++) {
		const AcmDevStream &d = ds[patches[p].stream];
		if (on_tile_kernel[patches[p].stream]) {
			/* a tile-kernel stream: the patch lands in every window that can see it (its own tile, and the next one
			 * when it sits in that tile's two halo rows); a patch no window sees (behind the last emitted tile, or
			 * outside a windowed decode) changes nothing that is emitted and has no place in the plane: the stream
			 * owns no plane run of its own (scratch_off 0 is somebody else's) */
			const uint64_t row = patches[p].sample >> d.level;
			if (win_of.empty())
				continue;
			for (size_t w : win_of[patches[p].stream]) {
				const PatchWindow &pw = windows[w];
				if (row >= pw.lo_row && row < pw.hi_row)
					dp.push_back(AcmDevPatch{ pw.scratch_off + (patches[p].sample - (pw.lo_row << d.level)), patches[p].value, 0 });
			}
			continue;
		}
		const uint64_t first = (uint64_t)d.halo_row << d.level;
		const uint64_t end = (uint64_t)d.nrows << d.level;
		if (d.n_emit == 0 || patches[p].sample < first || patches[p].sample >= end)
			continue;
		dp.push_back(AcmDevPatch{ d.scratch_off + (patches[p].sample - first),
					  (int32_t)((uint32_t)patches[p].value << plane_shift[patches[p].stream]), 0 });
	}
	for (const AcmDevPatch &q : dp)
		if (q.dst >= plane) {
			set_err("internal: H1 patch lands at %llu of a %llu-element plane", (unsigned long long)q.dst, (unsigned long long)plane);
			return ACMHIP_ERR_ARG;
		}

	acmhip_plan *pl = new (std::nothrow) acmhip_plan;
	if (!pl)
		return ACMHIP_ERR_NOMEM;
	pl->dev = dev;
	pl->variant = variant;
	pl->form_only = (flags & ACMHIP_PLAN_FORM_ONLY) != 0;
	pl->form_rows = std::move(form_rows);
	int rc = to_device(pl, ds, &pl->d_streams);
	for (uint32_t lv = 0; lv < 16 && rc == ACMHIP_OK; lv++) {
		if (!tiles[lv].empty() || !tiles_extra[lv].empty() || !tiles2[lv].empty() || !tiles2p[lv].empty() || !tiles2m[lv].empty()) {
			LevelGroup g;
			g.level = lv;
			if (!tiles_extra[lv].empty()) {
				g.ntiles_extra = (uint32_t)tiles_extra[lv].size();
				rc = to_device(pl, tiles_extra[lv], &g.d_tiles_extra);
				st.tiles += g.ntiles_extra;
				st.launches += 1;
				if (tiles[lv].empty())
					st.launches -= 1;               /* the launch counted below does not happen */
			}
			const size_t grid = (size_t)acmk_fused_grid(lv, variant, dev->cus);
			/* the lean kernel replays one tile per workgroup as a lead-in: worth it from a few tiles per workgroup on */
			const size_t grid2 = (size_t)acmk_tile2_grid(lv, dev->cus);
			const size_t n2 = tiles2[lv].size() + tiles2p[lv].size() + tiles2m[lv].size();
			/* streams that came with a packed form always take the lean kernel: their caller may have staged nothing else for
			 * these rows (acm_batch_decode with ACM_BATCH_STAGE_PACKED uploads the int16 form of the ragged tails only) */
			const bool k2 = grid2 && (lv > ACM_K1_MAX_LEVEL ? n2 > 0         /* levels 13, 14: decided before the cut */
						  : lean == 1 ? n2 > 0 : (n2 >= 8 * grid2 || !tiles2p[lv].empty() || !tiles2m[lv].empty()));
			g.carry = !k2 && !tiles_carry[lv].empty() && carry_wanted(tiles_carry[lv].size(), grid, (size_t)acmk_fused_tile_rows(lv, variant), flags);
			const std::vector<AcmTile> &use = k2 ? tiles_rest[lv] : g.carry ? tiles_carry[lv] : tiles[lv];
			g.ntiles = (uint32_t)use.size();
			rc = to_device(pl, use, &g.d_tiles);
			if (k2 && rc == ACMHIP_OK && !pl->d_sink)
				rc = plan_malloc(pl, (void **)&pl->d_sink, ACM_K2_SINK_BYTES);
			if (k2 && rc == ACMHIP_OK) {
				g.ntiles2 = (uint32_t)tiles2[lv].size();
				rc = to_device(pl, tiles2[lv], &g.d_tiles2);
				st.tiles += g.ntiles2;
				st.launches += g.ntiles2 ? 1 : 0;
				if (rc == ACMHIP_OK && !tiles2p[lv].empty()) {
					g.ntiles2p = (uint32_t)tiles2p[lv].size();
					rc = to_device(pl, tiles2p[lv], &g.d_tiles2p);
					if (rc == ACMHIP_OK)
						rc = to_device(pl, tiles2p_plain[lv], &g.d_tiles2p_plain);
					st.tiles += g.ntiles2p;
					st.packed_tiles += g.ntiles2p;
					st.launches += 1;
				}
				if (rc == ACMHIP_OK && !tiles2m[lv].empty()) {
					g.ntiles2m = (uint32_t)tiles2m[lv].size();
					rc = to_device(pl, tiles2m[lv], &g.d_tiles2m);
					g.ntiles2m_plain = (uint32_t)tiles2m_plain[lv].size();
					if (rc == ACMHIP_OK)
						rc = to_device(pl, tiles2m_plain[lv], &g.d_tiles2m_plain);
					st.tiles += g.ntiles2m;
					st.mform_tiles += g.ntiles2m;
					st.launches += 1;
				}
				if (g.ntiles == 0)
					st.launches -= 1;
			}
			pl->fused.push_back(g);
			st.tiles += g.ntiles;
			st.launches += 1;
		}
		if (lv <= ACM_SMALL_MAX_LEVEL && !small_lists[lv].empty() && rc == ACMHIP_OK) {
			LevelGroup g;
			g.level = lv;
			g.nlist = (uint32_t)small_lists[lv].size();
			g.max_emit = grp_max_emit[lv];
			rc = to_device(pl, small_lists[lv], &g.d_list);
			pl->small.push_back(g);
			st.launches += 1;
		}
		if (!prefix_lists[lv].empty() && rc == ACMHIP_OK) {
			LevelGroup g;
			g.level = lv;
			g.prefix_stages = lv - 12;
			for (uint32_t id : prefix_lists[lv])
				g.prefix_patched = g.prefix_patched || (id < n && has_patch[id]);
			g.nlist = (uint32_t)prefix_lists[lv].size();
			g.max_elems = grp_max_elems[lv];
			rc = to_device(pl, prefix_lists[lv], &g.d_list);
			if (rc == ACMHIP_OK) {
				g.carry = carry_wanted(prefix_tiles_carry[lv].size(), (size_t)acmk_plane_grid(dev->cus), (size_t)acmk_plane_tile_rows(), flags);
				const std::vector<AcmTile> &use = g.carry ? prefix_tiles_carry[lv] : prefix_tiles[lv];
				g.ntiles = (uint32_t)use.size();
				rc = to_device(pl, use, &g.d_tiles);
			}
			pl->prefix.push_back(g);
			st.tiles += g.ntiles;
			st.launches += g.prefix_patched ? 2 + g.prefix_stages : 2;           /* prefix sweep (or unpack + stages), tile kernel */
		}
		if (!lists[lv].empty() && rc == ACMHIP_OK) {
			LevelGroup g;
			g.level = lv;
			g.nlist = (uint32_t)lists[lv].size();
			g.max_elems = grp_max_elems[lv];
			g.max_emit = grp_max_emit[lv];
			rc = to_device(pl, lists[lv], &g.d_list);
			pl->stagewise.push_back(g);
			st.launches += lv + 1;          /* stages + emit */
		}
	}
	if (rc == ACMHIP_OK && plane > 0) {
		pl->n_sw_all = (uint32_t)sw_all.size();
		pl->sw_max_elems = sw_max;
		pl->plane_elems = plane;
		if (!sw_all.empty()) {
			rc = to_device(pl, sw_all, &pl->d_sw_all);
			st.launches += 1;                       /* unpack */
		}
		if (rc == ACMHIP_OK && !dp.empty()) {
			pl->npatches = dp.size();
			rc = to_device(pl, dp, &pl->d_patches);
			st.launches += 1;
		}
		for (int b = 0; b < 2 && rc == ACMHIP_OK; b++)
			rc = plan_malloc(pl, (void **)&pl->d_plane[b], plane * sizeof(int32_t));
	}
	if (rc == ACMHIP_OK)
		rc = upload_done(pl, (flags & ACMHIP_PLAN_UPLOAD_ASYNC) != 0);
	if (rc != ACMHIP_OK) {
		acmhip_plan_destroy(pl);
		return rc;
	}
	if (pl->fused.size() > 1 && !(ACM_TUNING_ENV("ACM_PLAN_STREAMS") && atoi(ACM_TUNING_ENV("ACM_PLAN_STREAMS")) <= 1)) {
		hipError_t e = hipEventCreateWithFlags(&pl->ev_fork, hipEventDisableTiming);
		for (int k = 0; k < 2 && e == hipSuccess; k++) {
			e = hipEventCreateWithFlags(&pl->ev_join[k], hipEventDisableTiming);
			if (e == hipSuccess && !dev->side[k])
				e = hipStreamCreateWithFlags(&dev->side[k], hipStreamNonBlocking);
		}
		if (e != hipSuccess) {
			rc = hip_fail(e, "side streams of a multi-level plan");
			acmhip_plan_destroy(pl);
			return rc;
		}
	}
	pl->stats = st;
	*out = pl;
	return ACMHIP_OK;
}

extern "C" int acmhip_plan_form_rows(const acmhip_plan *plan, size_t stream, uint64_t *rows)
{
	if (!plan || !rows || stream >= plan->form_rows.size())
		return ACMHIP_ERR_ARG;
	*rows = plan->form_rows[stream];
	return ACMHIP_OK;
}

extern "C" int acmhip_plan_get_stats(const acmhip_plan *plan, acmhip_plan_stats *out)
{
	if (!plan || !out)
		return ACMHIP_ERR_ARG;
	*out = plan->stats;
	return ACMHIP_OK;
}

#define LAUNCHTRY(call) do { int e_ = (call); if (e_ != 0) { \
	if (e_ > 0) return hip_fail((hipError_t)e_, #call); \
	set_err("%s: unsupported configuration", #call); return ACMHIP_ERR_ARG; } } while (0)

extern "C" int acmhip_plan_launch(acmhip_plan *pl, const int16_t *d_idx, const acmhip_blkhdr *d_hdr,
				  int16_t *d_pcm, unsigned fmt)
{
	if (!pl || fmt > 3)
		return ACMHIP_ERR_ARG;
	void *st = (void *)pl->dev->stream;
	if (pl->ev_upload)
		HIPTRY(hipStreamWaitEvent(pl->dev->stream, pl->ev_upload, 0));
	if (pl->form_only && !pl->mform)
		for (const LevelGroup &g : pl->fused)
			if (g.ntiles2m) {
				set_err("a plan cut with ACMHIP_PLAN_FORM_ONLY is launched without its byte-plane form bound");
				return ACMHIP_ERR_ARG;
			}

	const bool spread = pl->ev_fork != nullptr;
	if (spread) {
		HIPTRY(hipEventRecord(pl->ev_fork, pl->dev->stream));
		for (int k = 0; k < 2; k++)
			HIPTRY(hipStreamWaitEvent(pl->dev->side[k], pl->ev_fork, 0));
	}
	/* The lean kernels of every level group, ONE BEHIND THE OTHER on the device stream: each of them fills the chip by itself (the planner
	 * hands a level's tiles to them only from eight per workgroup on), and the chunk kernel's workgroup IS a CU - two of them overlapping
	 * on separate streams share the CUs between them and both run at half rate until the shorter one ends, with whatever comes third
	 * waiting for a whole CU's LDS (the configs[2] corpus, levels 7-9 in one plan: 1.78 ms a step with the groups spread over three
	 * streams, round 6).  What IS spread over the side streams are the small kernels behind them - the ragged tails on acm_fused_tile -
	 * which fill the gaps the big ones leave */
	for (const LevelGroup &g : pl->fused) {
		LAUNCHTRY(acmk_launch_tile2(g.level, pl->dev->cus, g.d_tiles2, g.ntiles2, d_idx, d_hdr, d_pcm, pl->d_sink, fmt, st));
		if (pl->pk_chunks)
			LAUNCHTRY(acmk_launch_tile2p(g.level, pl->dev->cus, g.d_tiles2p, g.ntiles2p, pl->pk_chunks, pl->pk_blob, d_hdr, d_pcm, pl->d_sink, fmt, st));
		else
			LAUNCHTRY(acmk_launch_tile2(g.level, pl->dev->cus, g.d_tiles2p_plain, g.ntiles2p, d_idx, d_hdr, d_pcm, pl->d_sink, fmt, st));
		if (pl->mform)
			LAUNCHTRY(acmk_launch_tile2m(g.level, pl->dev->cus, g.d_tiles2m, g.ntiles2m, pl->mform, pl->mform_pairs, d_hdr, d_pcm, pl->d_sink, fmt, st));
		else
			LAUNCHTRY(acmk_launch_tile2(g.level, pl->dev->cus, g.d_tiles2m_plain, g.ntiles2m_plain, d_idx, d_hdr, d_pcm, pl->d_sink, fmt, st));
	}
	size_t gi = 0;
	for (const LevelGroup &g : pl->fused) {
		void *gs = (!spread || gi % 3 == 0) ? st : (void *)pl->dev->side[gi % 3 - 1];
		gi++;
		LAUNCHTRY(acmk_launch_fused(g.level, pl->variant, pl->dev->cus, g.carry, pl->d_streams, g.d_tiles, g.ntiles, d_idx, d_hdr, d_pcm, fmt, gs));
		LAUNCHTRY(acmk_launch_fused(g.level, pl->variant, pl->dev->cus, 0, pl->d_streams, g.d_tiles_extra, g.ntiles_extra, d_idx, d_hdr, d_pcm, fmt, gs));
	}
	if (spread) {
		for (int k = 0; k < 2; k++) {
			HIPTRY(hipEventRecord(pl->ev_join[k], pl->dev->side[k]));
			HIPTRY(hipStreamWaitEvent(pl->dev->stream, pl->ev_join[k], 0));
		}
	}

	for (const LevelGroup &g : pl->small)
		LAUNCHTRY(acmk_launch_small(g.level, pl->d_streams, g.d_list, g.nlist, g.max_emit, d_idx, d_hdr, d_pcm, fmt, st));

	if (pl->n_sw_all || !pl->prefix.empty()) {
		if (pl->n_sw_all)
			LAUNCHTRY(acmk_launch_unpack(pl->d_streams, pl->d_sw_all, pl->n_sw_all, pl->sw_max_elems,
						     d_idx, d_hdr, pl->d_plane[0], 0, st));
		for (const LevelGroup &g : pl->prefix)
			if (g.prefix_patched)
				LAUNCHTRY(acmk_launch_unpack(pl->d_streams, g.d_list, g.nlist, g.max_elems, d_idx, d_hdr, pl->d_plane[0], 16 - g.level, st));
		LAUNCHTRY(acmk_launch_patch(pl->d_patches, pl->npatches, pl->d_plane[0], st));
		for (const LevelGroup &g : pl->stagewise) {
			int cur = 0;
			for (uint32_t k = 0; k < g.level; k++, cur ^= 1)
				LAUNCHTRY(acmk_launch_stage(pl->d_streams, g.d_list, g.nlist, g.max_elems, g.level, k,
							    pl->d_plane[cur], pl->d_plane[cur ^ 1], 0, st));
			LAUNCHTRY(acmk_launch_emit(pl->d_streams, g.d_list, g.nlist, g.max_emit, pl->d_plane[cur],
						   d_pcm, fmt, st));
		}
		for (const LevelGroup &g : pl->prefix) {
			int cur = 0;
			if (!g.prefix_patched) {
				/* unpack and all level - 12 stages in one sweep, straight into the plane the tile kernel reads */
				LAUNCHTRY(acmk_launch_prefix(pl->d_streams, g.d_list, g.nlist, g.max_elems, g.level, d_idx, d_hdr, pl->d_plane[1], st));
				cur = 1;
			} else {
				for (uint32_t k = 0; k < g.prefix_stages; k++, cur ^= 1)
					LAUNCHTRY(acmk_launch_stage(pl->d_streams, g.d_list, g.nlist, g.max_elems, g.level, k,
								    pl->d_plane[cur], pl->d_plane[cur ^ 1], 16 - g.level, st));
			}
			LAUNCHTRY(acmk_launch_fused_plane(pl->dev->cus, g.carry, pl->d_streams, g.d_tiles, g.ntiles, pl->d_plane[cur], d_pcm, fmt, st));
		}
	}
	return ACMHIP_OK;
}

extern "C" int acmhip_plan_bind_packed(acmhip_plan *pl, const acmhip_packed_chunk *d_chunks, const uint8_t *d_blob)
{
	if (!pl)
		return ACMHIP_ERR_ARG;
	if ((d_chunks == nullptr) != (d_blob == nullptr)) {
		set_err("acmhip_plan_bind_packed: both tables, or none");
		return ACMHIP_ERR_ARG;
	}
	pl->pk_chunks = d_chunks;
	pl->pk_blob = d_blob;
	return ACMHIP_OK;
}

extern "C" int acmhip_plan_bind_mform(acmhip_plan *pl, const uint8_t *d_mform, const acmhip_mform_pair *d_pairs)
{
	if (!pl)
		return ACMHIP_ERR_ARG;
	if ((d_mform == nullptr) != (d_pairs == nullptr)) {
		set_err("acmhip_plan_bind_mform: the arena and its pair table, or neither");
		return ACMHIP_ERR_ARG;
	}
	pl->mform = d_mform;
	pl->mform_pairs = d_pairs;
	return ACMHIP_OK;
}

extern "C" int acmhip_plan_time(acmhip_plan *pl, const int16_t *d_idx, const acmhip_blkhdr *d_hdr,
				int16_t *d_pcm, unsigned fmt, int reps, float *ms_total)
{
	if (!pl || reps < 1 || !ms_total)
		return ACMHIP_ERR_ARG;
	hipEvent_t a, b;
	HIPTRY(hipEventCreate(&a));
	HIPTRY(hipEventCreate(&b));
	HIPTRY(hipEventRecord(a, pl->dev->stream));
	int rc = ACMHIP_OK;
	for (int i = 0; i < reps && rc == ACMHIP_OK; i++)
		rc = acmhip_plan_launch(pl, d_idx, d_hdr, d_pcm, fmt);
	hipError_t e = hipEventRecord(b, pl->dev->stream);
	if (e == hipSuccess)
		e = hipEventSynchronize(b);
	if (e == hipSuccess)
		e = hipEventElapsedTime(ms_total, a, b);
	(void)hipEventDestroy(a);
	(void)hipEventDestroy(b);
	if (rc != ACMHIP_OK)
		return rc;
	if (e != hipSuccess)
		return hip_fail(e, "event timing");
	return ACMHIP_OK;
}
