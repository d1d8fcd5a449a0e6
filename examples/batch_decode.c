/*
 * batch_decode.c - minimal C user of the batch front end (include/acm_hip.h).
 *
 *   cc -Iinclude examples/batch_decode.c -Llibacm_amd/lib -lacm_hip -Wl,-rpath,$PWD/libacm_amd/lib -o batch_decode
 *   ./batch_decode a.acm b.acm ...        -> one line per file: status, words, FNV-1a of the PCM (s16le)
 *
 * Everything the reference's acmtool does per file in its read loop (acmtool.c:274-291) happens in one
 * acm_batch_decode call: threaded (or device-side) bit parsing, one synthesis launch per level, PCM back.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "acm_hip.h"

static unsigned char *slurp(const char *path, size_t *len)
{
	FILE *f = fopen(path, "rb");
	unsigned char *p;
	long n;
	if (!f)
		return NULL;
	fseek(f, 0, SEEK_END);
	n = ftell(f);
	fseek(f, 0, SEEK_SET);
	p = malloc(n > 0 ? (size_t)n : 1);
	*len = (p && n > 0) ? fread(p, 1, (size_t)n, f) : 0;
	fclose(f);
	return p;
}

int main(int argc, char **argv)
{
	const int n = argc - 1;
	acm_batch_item *items = calloc((size_t)(n > 0 ? n : 1), sizeof(*items));
	acm_batch_opts opts;
	acm_batch_timing tm;
	acmhip_device *dev = NULL;
	int i, rc;

	if (n < 1) {
		fprintf(stderr, "usage: %s file.acm ...\n", argv[0]);
		return 2;
	}
	for (i = 0; i < n; i++) {
		acm_stage_info si;
		items[i].data = slurp(argv[i + 1], &items[i].len);
		if (items[i].data && acm_stage_probe(items[i].data, items[i].len, 0, &si) == 0) {
			items[i].pcm_cap = si.total_values;
			items[i].pcm = calloc(si.total_values ? si.total_values : 1, sizeof(int16_t));
		}
	}
	memset(&opts, 0, sizeof(opts));
	opts.fmt = ACMHIP_FMT_S16LE;
	opts.parse = ACM_BATCH_PARSE_AUTO;
	rc = acmhip_device_open(0, NULL, &dev);
	if (rc == ACMHIP_OK)
		rc = acm_batch_decode(dev, items, (size_t)n, &opts, &tm);
	if (rc != ACMHIP_OK) {
		fprintf(stderr, "batch decode failed: %s\n", acmhip_last_error());
		return 1;
	}
	for (i = 0; i < n; i++) {
		uint32_t h = 2166136261u;
		const unsigned char *b = (const unsigned char *)items[i].pcm;
		uint64_t k;
		for (k = 0; b && k < items[i].words * 2; k++)
			h = (h ^ b[k]) * 16777619u;
		printf("%s status %d words %llu fnv1a %08x\n", argv[i + 1], items[i].status, (unsigned long long)items[i].words, h);
	}
	fprintf(stderr, "%llu samples, %.3f s wall (parse %.3f, h2d %.3f, kernel %.4f, d2h %.3f)\n",
		(unsigned long long)tm.samples, tm.total_s, tm.stage_s, tm.h2d_s, tm.kernel_s, tm.d2h_s);
	acmhip_device_close(dev);
	return 0;
}
