/* profiles/sign_search.c - which positions of which butterfly stage to STORE NEGATED at the exact-32-bit levels (acm_kernels.hip:
 * sign_field).  gcc -O3 -o sign_search sign_search.c && ./sign_search 2 && ./sign_search 3     (G = 3 takes ~45 s on one core)
 *
 * Model: a pass of G stages over a body of BODY = 2^(G+1) walk positions, periodic.  Stage t (0..G-1) has stride d = 2^(G-1-t)
 * in walk units and sigma(u) = -1 where bit G-1-t of u is set:  y[u] = 2 x[u-d] + sigma (x[u-2d] + x[u])   (decode.c:518-519).
 * A field has one bit per position: 1 = the value there is stored negated.  With input field Tin and output field Tout the stored
 * output is  A 2 X1 + B2 X2 + B0 X0  with A = Tout(u) Tin(u-d), B2 = Tout(u) sigma Tin(u-2d), B0 = Tout(u) sigma Tin(u); it costs
 * two ops (add / sub + v_lshl_add_u32) unless A = -1 or B2 = B0 = -1: three (exact 32-bit: no one-op t - 2 z).
 * Exact backward dynamic programme over all 2^BODY fields per stage.  Inputs of a pass: plain / negated where the TOP bit of u is
 * set (= what the pass before can leave: its bit 0 is this pass's top bit) / their negations / FREE (first pass: the signs ride on
 * the unpack multiply).  Outputs: plain / negated at ODD u / negations.  Printed: cost and the fields of stage 0 .. G. */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>

static int G, B; static uint32_t MASK;
static inline uint32_t rot(uint32_t f, int d) { d %= B; return ((f << d) | (f >> (B - d))) & MASK; }   /* field value at u-d -> position u */
static uint32_t SIG[8];
static inline int cost3(uint32_t r1, uint32_t r2x, uint32_t r0x, uint32_t tout)
{
	return __builtin_popcount(((tout ^ r1) | ((tout ^ r2x) & (tout ^ r0x))) & MASK);
}
int main(int argc, char **argv)
{
	G = atoi(argv[1]); B = 2 << G; MASK = (B == 32) ? 0xFFFFFFFFu : ((1u << B) - 1);
	const uint32_t NF = 1u << B;
	uint32_t negtop = 0; for (int u = 0; u < B; u++) if ((u >> G) & 1) negtop |= 1u << u;
	uint32_t odd = 0; for (int u = 0; u < B; u++) if (u & 1) odd |= 1u << u;
	/* inputs: plain / negated where the TOP bit of u is set (what the previous pass can leave: its bit 0 is this pass's bit G);
	 * outputs: plain / negated at ODD u (bit 0 of u = the next pass's top bit) */
	const uint32_t ins[4] = { 0, negtop, MASK, negtop ^ MASK }; const char *names[4] = { "P", "N", "-P", "-N" };
	const uint32_t outs[4] = { 0, odd, MASK, odd ^ MASK };
	uint8_t *V = malloc((size_t)NF * (G + 1)); uint32_t *arg = malloc((size_t)NF * (G + 1) * sizeof(uint32_t));
	for (int o = 0; o < 4; o++) {
		memset(V, 255, (size_t)NF * (G + 1));
		V[(size_t)G * NF + outs[o]] = 0;
		for (int t = G - 1; t >= 0; t--) {
			const int d = 1 << (G - 1 - t), pb = G - 1 - t;
			uint32_t sig = 0;
			for (int u = 0; u < B; u++) if ((u >> pb) & 1) sig |= 1u << u;
			/* candidate outputs sorted by their value: stop early */
			const uint8_t *Vn = V + (size_t)(t + 1) * NF;
			for (uint32_t tin = 0; tin < NF; tin++) {
				int best = 255; uint32_t ba = 0;
				const uint32_t r1 = rot(tin, d), r2x = rot(tin, 2 * d) ^ sig, r0x = tin ^ sig;
				for (uint32_t tout = 0; tout < NF; tout++) {
					const int v = Vn[tout];
					if (v >= best) continue;
					const int c = v + cost3(r1, r2x, r0x, tout);
					if (c < best) { best = c; ba = tout; }
				}
				V[(size_t)t * NF + tin] = (uint8_t)best; arg[(size_t)t * NF + tin] = ba;
			}
		}
		/* report: boundary inputs and the free-input optimum */
		for (int i = 0; i < 4; i++) {
			uint32_t f = ins[i]; printf("G=%d in %-3s out %-3s cost %2d of %d  fields:", G, names[i], names[o], V[f], G * B);
			for (int t = 0; t <= G; t++) { printf(" %0*x", B / 4, f); if (t < G) f = arg[(size_t)t * NF + f]; }
			printf("\n");
		}
		int best = 255; uint32_t bf = 0;
		for (uint32_t f = 0; f < NF; f++) if (V[f] < best) { best = V[f]; bf = f; }
		{ uint32_t f = bf; printf("G=%d in FREE out %-3s cost %2d of %d  fields:", G, names[o], best, G * B);
		  for (int t = 0; t <= G; t++) { printf(" %0*x", B / 4, f); if (t < G) f = arg[(size_t)t * NF + f]; } printf("\n"); }
	}
	return 0;
}
