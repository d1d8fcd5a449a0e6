// Diagnostic: where do the cycles of acm_fused_tile's tile loop go?  Builds the real kernel source with
// ACM_STAMPS (s_memtime stamps per phase, per wave) on synthetic staged data and prints phase shares.
// Timing-only; never part of the product build.   hipcc -O3 --offload-arch=gfx950 -I include -I libacm_amd/csrc
#define ACM_STAMPS 1
#include "../../libacm_amd/csrc/acm_kernels.hip"
#include <cstdio>
#include <vector>
int main(int argc, char **argv) {
  int level = argc > 1 ? atoi(argv[1]) : 7; int variant = argc > 2 ? atoi(argv[2]) : 0;
  const uint32_t rows = 16, nblocks = level == 7 ? 1000 : 250, nstreams = 1024;
  const uint64_t cols = 1ull << level, per = (uint64_t)nblocks * rows * cols;
  std::vector<AcmDevStream> ds(nstreams); std::vector<AcmTile> tiles;
  const uint32_t T = acmk_fused_tile_rows(level, variant) - 2;
  for (uint32_t i = 0; i < nstreams; i++) {
    AcmDevStream &d = ds[i]; d = AcmDevStream{};
    d.idx_off = i * per; d.hdr_off = (uint64_t)i * nblocks; d.pcm_off = i * per; d.n_emit = per;
    d.level = level; d.rows = rows; d.nrows = nblocks * rows; d.row_begin = 0; d.halo_row = 0;
    for (uint32_t r = 0; r < d.nrows; r += T) tiles.push_back(AcmTile{i, r});
  }
  int16_t *d_idx, *d_pcm; acmhip_blkhdr *d_hdr; AcmDevStream *d_s; AcmTile *d_t;
  (void)hipMalloc(&d_idx, per * nstreams * 2); (void)hipMalloc(&d_pcm, per * nstreams * 2);
  (void)hipMalloc(&d_hdr, (size_t)nstreams * nblocks * 8); (void)hipMalloc(&d_s, ds.size() * sizeof(AcmDevStream));
  (void)hipMalloc(&d_t, tiles.size() * sizeof(AcmTile));
  (void)hipMemset(d_idx, 1, per * nstreams * 2); (void)hipMemset(d_hdr, 1, (size_t)nstreams * nblocks * 8);
  (void)hipMemcpy(d_s, ds.data(), ds.size() * sizeof(AcmDevStream), hipMemcpyHostToDevice);
  (void)hipMemcpy(d_t, tiles.data(), tiles.size() * sizeof(AcmTile), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; rep++) {
    (void)hipEventRecord(e0);
    acmk_launch_fused(level, variant, 256, d_s, d_t, (uint32_t)tiles.size(), d_idx, d_hdr, d_pcm, 0, nullptr);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  }
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  static unsigned long long h[2048][8];
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_acm_stamps), sizeof(h));
  double sum[8] = {0}; int n = 0;
  for (int w = 0; w < 2048; w++) { if (!h[w][1]) continue; n++; for (int k = 0; k < 8; k++) sum[k] += (double)h[w][k]; }
  double tot = 0; for (int k = 0; k < 6; k++) tot += sum[k];
  const char *names[6] = {"top barrier (wait rowval/prev store)", "first pass (wait HBM loads + unpack + butterflies + LDS store)",
                          "prefetch issue (next ctx, hdr, idx loads)", "LDS passes", "barrier before write-out", "write-out (LDS gather + HBM store)"};
  printf("level %d variant %d: %.3f ms per launch with stamps, %zu tiles, %d waves sampled, s_memtime ticks per wave %.0f\n", level, variant, ms, tiles.size(), n, tot / n);
  for (int k = 0; k < 6; k++) printf("  %-70s %5.1f %%\n", names[k], 100.0 * sum[k] / tot);
  return 0;
}
