// Diagnostic: where do the cycles of acm_tile2 (argv[2] = 1: its matrix-core build) go?  Builds the real kernel source with ACM_STAMPS
// (s_memtime stamps per phase, per wave) on synthetic staged data and prints the phase shares.  Timing only.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -I libacm_amd/csrc -o phases_k2 profiles/ubench/phases_k2.hip
#define ACM_STAMPS 1
#include "../../libacm_amd/csrc/acm_kernels.hip"
#include <cstdio>
#include <vector>
int main(int argc, char **argv) {
  const int level = argc > 1 ? atoi(argv[1]) : 9; const int mform = argc > 2 ? atoi(argv[2]) : 0;
  const uint32_t rows = 16, nblocks = (uint32_t)((1u << 21) / (rows << level)), nstreams = 1024;
  const uint64_t cols = 1ull << level, per = (uint64_t)nblocks * rows * cols;
  const uint32_t TR = (uint32_t)acmk_tile2_rows(level);
  std::vector<AcmTile2> tiles;
  for (uint32_t i = 0; i < nstreams; i++)
    for (uint32_t r = 0; r + TR <= nblocks * rows; r += TR) {
      const uint64_t rh = r >= 2 ? r - 2 : 0;
      tiles.push_back(AcmTile2{ i * per + r * cols, i * per + r * cols, (uint32_t)(i * nblocks + rh / rows), (uint32_t)(rh % rows),
                                (uint32_t)(((1ull << 32) + rows - 1) / rows), r == 0 ? ACM_TILE_FRESH : 0u });
    }
  int16_t *d_idx, *d_pcm, *d_sink; acmhip_blkhdr *d_hdr; AcmTile2 *d_t; (void)hipMalloc(&d_sink, ACM_K2_SINK_BYTES);
  (void)hipMalloc(&d_idx, per * nstreams * 2 + (64 << 10));   d_idx += 16 << 10;       /* the matrix-core build reads two rows in front of a stream */ (void)hipMalloc(&d_pcm, per * nstreams * 2);
  (void)hipMalloc(&d_hdr, (size_t)nstreams * nblocks * 8); (void)hipMalloc(&d_t, tiles.size() * sizeof(AcmTile2));
  (void)hipMemset(d_idx, 1, per * nstreams * 2); (void)hipMemset(d_hdr, 1, (size_t)nstreams * nblocks * 8);
  (void)hipMemcpy(d_t, tiles.data(), tiles.size() * sizeof(AcmTile2), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 5; rep++) {
    (void)hipEventRecord(e0);
    (mform ? acmk_launch_tile2m(level, 256, d_t, (uint32_t)tiles.size(), (const uint8_t *)d_idx, d_hdr, d_pcm, d_sink, 0, nullptr) : acmk_launch_tile2(level, 256, d_t, (uint32_t)tiles.size(), d_idx, d_hdr, d_pcm, d_sink, 0, nullptr));
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  static unsigned long long h[2048][8];
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_acm_stamps), sizeof(h));
  double sum[8] = {0}; int n = 0;
  for (int w = 0; w < 2048; w++) { if (!h[w][1]) continue; n++; for (int k = 0; k < 8; k++) sum[k] += (double)h[w][k]; }
  double tot = 0; for (int k = 0; k < 7; k++) tot += sum[k];
  const char *names[7] = {"carry reset + row values + top barrier", "first pass (unpack + butterflies + LDS store)",
                          "issue of the next tile's loads", "LDS passes", "barrier before write-out", "write-out (LDS gather + PCM stores)",
                          "end-of-iteration wait for the prefetched loads"};
  printf("%s level %d: %.3f ms per launch with stamps (%.1f Gsamples/s), %zu tiles, %d waves sampled, s_memtime ticks per wave %.0f\n",
         mform ? "matrix-core build," : "vector-ALU build,", level, ms, per * nstreams / ms / 1e6, tiles.size(), n, tot / n);
  for (int k = 0; k < 7; k++) printf("  %-62s %5.1f %%  %8.0f ticks per tile\n", names[k], 100.0 * sum[k] / tot, sum[k] / n / (tiles.size() / 1024.0));
  return 0;
}
