// Diagnostic: where do the cycles of the chunk kernel (acm_chunk) go?  Builds the real kernel source with ACM_STAMPS (s_memtime stamps per
// phase, per wavefront) on synthetic byte-plane data and prints the phase shares.  Timing only (the PCM is not looked at).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -I libacm_amd/csrc -o profiles/ubench/phases_k3.bin profiles/ubench/phases_k3.hip
//   ./phases_k3 <level> [rows per block = 16] [percent of the blocks at 16 bits = 56] [percent at 12 bits = 0]
#ifndef NO_STAMPS
#define ACM_STAMPS 1              /* (-DNO_STAMPS: the plain kernel, for counter collections and timing) */
#endif
#include "../../libacm_amd/csrc/acm_kernels.hip"
#include <cstdio>
#include <vector>
int main(int argc, char **argv) {
  const int level = argc > 1 ? atoi(argv[1]) : 9;
  const uint32_t rows = argc > 2 ? atoi(argv[2]) : 16, pct16 = argc > 3 ? atoi(argv[3]) : 56, pct12 = argc > 4 ? atoi(argv[4]) : 0;
  const uint32_t nstreams = 1024, nrows = (uint32_t)((1u << 21) >> level), nblocks = (nrows + rows - 1) / rows;
  const uint64_t cols = 1ull << level, per = (uint64_t)nrows * cols;
  const uint32_t TR = (uint32_t)acmk_tile2m_rows(level);
  if (acmk_tile2m_stages(level) != 6) { printf("level %d has no chunk kernel\n", level); return 1; }
  std::vector<AcmTile2> tiles; std::vector<uint32_t> pairs; uint64_t at = 0; uint32_t seed = 12345;
  for (uint32_t i = 0; i < nstreams; i++) {
    const uint32_t p0 = (uint32_t)pairs.size();
    pairs.push_back((uint32_t)((at >> 6) << 2 | ACMHIP_BP_BYTE)); at += 2 * cols;            /* the pair of zeros in front */
    uint32_t cls = ACMHIP_BP_WORD;
    for (uint32_t p = 0; p < nrows / 2; p++) {
      if ((2 * p) % rows < 2) { seed = seed * 1664525u + 1013904223u; const uint32_t d100 = (seed >> 16) % 100; cls = d100 < pct16 ? ACMHIP_BP_WORD : d100 < pct16 + pct12 ? ACMHIP_BP_NIB12 : ACMHIP_BP_BYTE; }
      pairs.push_back((uint32_t)((at >> 6) << 2 | cls)); at += (cls == ACMHIP_BP_WORD ? 4 : cls == ACMHIP_BP_NIB12 ? 3 : 2) * cols;
    }
    for (uint32_t r = 0; r + TR <= nrows; r += TR) {
      const uint64_t rh = r >= 2 ? r - 2 : 0;
      tiles.push_back(AcmTile2{ p0 + r / 2, i * per + r * cols, (uint32_t)(i * nblocks + rh / rows), (uint32_t)(rh % rows),
                                (uint32_t)(((1ull << 32) + rows - 1) / rows), (r == 0 ? ACM_TILE_FRESH : 0u) | (r == 1 ? ACM_TILE_ROW1 : 0u) | ((r & 1) ? ACM_TILE_ODD : 0u) |
                                ((rh % rows) + (r + TR - 1 - rh) < rows ? ACM_TILE_ONEBLOCK : 0u) });
    }
  }
  for (int k = 0; k < 64; k++) pairs.push_back(0);
  uint8_t *d_blob; int16_t *d_pcm, *d_sink; acmhip_blkhdr *d_hdr; AcmTile2 *d_t; uint32_t *d_pairs;
  (void)hipMalloc(&d_sink, ACM_K2_SINK_BYTES); (void)hipMalloc(&d_blob, at + 4096); (void)hipMalloc(&d_pcm, per * nstreams * 2);
  (void)hipMalloc(&d_hdr, (size_t)nstreams * nblocks * 8); (void)hipMalloc(&d_t, tiles.size() * sizeof(AcmTile2)); (void)hipMalloc(&d_pairs, pairs.size() * 4);
  (void)hipMemset(d_blob, 3, at + 4096);
  std::vector<acmhip_blkhdr> hdr((size_t)nstreams * nblocks); for (size_t k = 0; k < hdr.size(); k++) { hdr[k].val = 1 + (uint32_t)(k * 7919u) % 255; hdr[k].pwr = 8; }
  (void)hipMemcpy(d_hdr, hdr.data(), hdr.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_t, tiles.data(), tiles.size() * sizeof(AcmTile2), hipMemcpyHostToDevice);
  (void)hipMemcpy(d_pairs, pairs.data(), pairs.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0, best = 1e9;
  for (int rep = 0; rep < 8; rep++) {
    (void)hipEventRecord(e0);
    int rc = acmk_launch_tile2m(level, 256, d_t, (uint32_t)tiles.size(), d_blob, d_pairs, d_hdr, d_pcm, d_sink, 0, nullptr);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    if (rc) { printf("launch failed %d\n", rc); return 1; }
    (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
#ifdef NO_STAMPS
  printf("acm_chunk level %d, %u rows per block, %u %% / %u %% of the blocks at 16 / 12 bits (staged %.2f B/sample), %zu chunks, no stamps: %.3f ms per launch (best of 8; %.3f of the 8 TB/s roofline at 4 B/sample)\n",
         level, rows, pct16, pct12, (double)at / (per * nstreams), tiles.size(), best, per * nstreams * 4.0 / best / 8e9);
  return 0;
#else
  static unsigned long long h[2048][8];
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_acm_stamps), sizeof(h));
  double sum[8] = {0}; int n = 0;
  for (int w = 0; w < 2048; w++) { if (!h[w][1]) continue; n++; for (int k = 0; k < 8; k++) sum[k] += (double)h[w][k]; }
  double tot = 0; for (int k = 0; k < 7; k++) tot += sum[k];
  const char *names[7] = {"carry reset + row values", "first pass (matrix passes, scaling, LDS store)", "issue of the next chunk's loads", "LDS passes",
                          "-", "write-out (LDS gather + PCM stores)", "end-of-iteration wait for the prefetched loads"};
  const double chunks_per_wave = tiles.size() / 4096.0;
  printf("acm_chunk level %d, %u rows per block, %u %% of the blocks at 16 bits: %.3f ms per launch with stamps (best of 8; %.3f of the 8 TB/s roofline at 4 B/sample), "
         "%zu chunks, %d waves sampled, ticks per chunk %.0f (staged %.2f B/sample)\n", level, rows, pct16, best, per * nstreams * 4.0 / best / 8e9, tiles.size(), n,
         tot / n / chunks_per_wave, (double)at / (per * nstreams));
  for (int k = 0; k < 7; k++) if (k != 4) printf("  %-62s %5.1f %%  %8.0f ticks per chunk\n", names[k], 100.0 * sum[k] / tot, sum[k] / n / chunks_per_wave);
  return 0;
#endif
}
