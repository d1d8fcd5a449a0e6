// Sanitizer harness for the HOST half of the product (bit reader, filler parsers, stream control): built with
// g++ -fsanitize=address,undefined (GPU sanitizers are not available on the pool; this is the CPU build).
// Feeds mutated / truncated ACM images through acm_stage_file and through the libacm.h API - decode-and-discard reads and reads into a
// buffer, which the library's host synthesis serves (acm_host_synth.cpp, acmhip_set_host_synth_limit(UINT64_MAX): never the device).
// The device entry points are stubbed: a call into any of them would mean the host path reached for the GPU, a failure here.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "acm_hip.h"
#include "libacm.h"

static int device_calls = 0;
extern "C" {
const char *acmhip_last_error(void) { return "stub"; }
int acmhip_device_open(int, void *, acmhip_device **) { device_calls++; return ACMHIP_ERR_NO_DEVICE; }
int acmhip_device_sync(acmhip_device *) { device_calls++; return ACMHIP_ERR_NO_DEVICE; }
void *acmhip_device_stream(acmhip_device *) { return nullptr; }
int acmk_warmup(void *) { device_calls++; return ACMHIP_ERR_NO_DEVICE; }
int acmhip_malloc(acmhip_device *, size_t, void **) { device_calls++; return ACMHIP_ERR_NO_DEVICE; }
int acmhip_free(acmhip_device *, void *) { return 0; }
int acmhip_upload(acmhip_device *, void *, const void *, size_t) { device_calls++; return ACMHIP_ERR_NO_DEVICE; }
int acmhip_download(acmhip_device *, void *, const void *, size_t) { device_calls++; return ACMHIP_ERR_NO_DEVICE; }
int acmhip_plan_create(acmhip_device *, const acmhip_stream_desc *, size_t, const acmhip_patch *, size_t, unsigned, acmhip_plan **) { device_calls++; return ACMHIP_ERR_NO_DEVICE; }
void acmhip_plan_destroy(acmhip_plan *) {}
int acmhip_plan_launch(acmhip_plan *, const int16_t *, const acmhip_blkhdr *, int16_t *, unsigned) { device_calls++; return ACMHIP_ERR_NO_DEVICE; }
/* tile geometries the stagers ask the kernels' translation unit for (acm_kernels.hip is not in this build): the shipped ones */
int acmk_tile2_rows(uint32_t level) { return level >= 6 && level <= 11 ? 8192 >> level : level == 12 || level == 13 ? 4 : level == 14 ? 2 : 0; }
int acmk_tile2m_rows(uint32_t level) { return level == 7 ? 64 : level >= 8 && level <= 11 ? 2048 >> level : level == 12 ? 1 : level == 13 || level == 14 ? 2 : 0; }
int acmk_tile2m_stages(uint32_t level) { return level == 7 ? 3 : level >= 8 && level <= 14 ? 6 : 0; }
int acmk_tile2m_lead_in(uint32_t) { return 1; }
int acmk_tile2p_rows(uint32_t level) { return level >= 6 && level <= 9 ? 8192 >> level : 0; }
int acmk_tile2p_group_rows(uint32_t level) { return level == 6 ? 32 : 16; }
int acmk_tile2p_slots(uint32_t) { return 28; }
int acmk_tile2p_waves(uint32_t) { return 4; }
int acmk_tile2p_pad_shift(uint32_t) { return 5; }
}

struct Mem { const uint8_t *p; size_t len, pos; unsigned max_read; };
static int rd(void *ptr, int size, int n, void *arg)
{
	Mem *m = (Mem *)arg;
	size_t want = (size_t)size * n;
	if (m->max_read && want > m->max_read) want = m->max_read;
	if (want > m->len - m->pos) want = m->len - m->pos;
	memcpy(ptr, m->p + m->pos, want);
	m->pos += want;
	return (int)(want / size);
}
static int sk(void *arg, int off, int) { Mem *m = (Mem *)arg; m->pos = (size_t)off > m->len ? m->len : (size_t)off; return 0; }
static int ln(void *arg) { return (int)((Mem *)arg)->len; }

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 11); }

static void exercise(const std::vector<uint8_t> &img)
{
	acm_stage_info si;
	if (acm_stage_probe(img.data(), img.size(), 0, &si) == ACM_OK) {
		const uint64_t bl = (uint64_t)si.rows * si.cols;
		uint64_t need = (si.total_values + bl - 1) / bl;
		if (need * bl < (1u << 22)) {                 /* keep allocations sane for mutated headers */
			std::vector<int16_t> idx(need * bl);
			std::vector<acmhip_blkhdr> hdr(need);
			std::vector<acmhip_patch> pt(4096);
			acm_stage_file(img.data(), img.size(), (int)(rnd() % 4) - 1, idx.data(), hdr.data(), need, pt.data(), pt.size(), &si);
			/* the fused staging: the byte-plane form written by the parsing pass (and every way it falls back) */
			const uint64_t nrows = (need * si.rows) & ~1ull;
			if (acmhip_mform_tile_rows(si.level) > 0 || (rnd() & 7) == 0) {
				std::vector<uint8_t> blob(acmhip_mform_bytes(si.level, nrows) + 256);
				std::vector<acmhip_mform_pair> pairs(acmhip_mform_pairs(nrows) + 32);
				uint64_t mf_rows = 0, mf_bytes = 0;
				acm_stage_file_mform(img.data(), img.size(), (int)(rnd() % 4) - 1, idx.data(), hdr.data(), need, &si, blob.data(), 4096, pairs.data(),
						     &mf_rows, &mf_bytes);
				if (mf_rows) {
					std::vector<int16_t> back(mf_rows * si.cols);
					if (mf_bytes > blob.size() || acmhip_mform_unrows(si.level, blob.data() - 4096, pairs.data(), mf_rows, back.data()) != ACMHIP_OK) {
						fprintf(stderr, "fused staging wrote a form that does not read back\n");
						abort();
					}
				}
			}
		}
	}
	Mem m{ img.data(), img.size(), 0, (rnd() & 3) ? 0u : 1u + rnd() % 9 };
	acm_io_callbacks io{ rd, sk, nullptr, ln };
	ACMStream *s = nullptr;
	if (acm_open_decoder(&s, &m, io, (int)(rnd() % 4) - 1) != ACM_OK)
		return;
	if ((uint64_t)s->block_len * 2 > (1u << 24)) {        /* window buffers scale with block_len: skip absurd headers */
		acm_close(s);
		return;
	}
	for (int k = 0; k < 40; k++) {
		switch (rnd() % 5) {
		case 0: acm_seek_pcm(s, rnd() % (acm_pcm_total(s) + 5)); break;
		case 1: acm_seek_time(s, rnd() % (acm_time_total(s) + 5)); break;
		case 2: {       /* PCM for real: parsed windows through the host synthesis, every output format */
			std::vector<uint8_t> out(2 + rnd() % 20000);
			acm_read_loop(s, out.data(), (unsigned)out.size(), (int)(rnd() & 1), 2, (int)(rnd() & 1));
			break;
		}
		default: acm_read_loop(s, NULL, 2 + rnd() % 20000, 0, 2, 1); break;
		}
		(void)acm_raw_tell(s); (void)acm_pcm_tell(s); (void)acm_time_tell(s); (void)acm_bitrate(s);
	}
	acm_close(s);
}

int main(int argc, char **argv)
{
	int iters = argc > 2 ? atoi(argv[2]) : 300, files = 0;
	acmhip_set_host_synth_limit(~0ull);
	for (int a = 1; a < argc; a++) {
		if (a == 2) continue;
		FILE *f = fopen(argv[a], "rb");
		if (!f) continue;
		std::vector<uint8_t> base;
		uint8_t tmp[4096]; size_t n;
		while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) base.insert(base.end(), tmp, tmp + n);
		fclose(f);
		files++;
		exercise(base);
		for (int i = 0; i < iters && !base.empty(); i++) {
			std::vector<uint8_t> img = base;
			int flips = 1 + rnd() % 4;
			for (int k = 0; k < flips; k++) img[rnd() % img.size()] ^= (uint8_t)(1u << (rnd() & 7));
			if (rnd() % 3 == 0) img.resize(rnd() % (img.size() + 1));
			exercise(img);
		}
	}
	if (device_calls) { fprintf(stderr, "host path called into the device %d times\n", device_calls); return 2; }
	printf("fuzz ok: %d files x %d mutations\n", files, iters);
	return 0;
}
