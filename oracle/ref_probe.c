/*
 * ref_probe.c - exposes three file-static routines of the REAL reference
 * decoder so that tests can feed them raw block matrices (fixture family F9,
 * SURVEY.md 8c).  It textually includes the reference's src/decode.c from
 * where it lies (path passed by oracle/Makefile as REF_DECODE_C); no reference
 * source is copied into this repository.  Authoring-container only.
 * TEST INFRASTRUCTURE ONLY.
 */
#include REF_DECODE_C

/* juggle_block (src/decode.c:528) on caller-owned block + wrapbuf */
void refprobe_juggle_block(unsigned level, unsigned rows, int *block, int *wrapbuf)
{
	ACMStream s;
	memset(&s, 0, sizeof(s));
	s.info.acm_level = level;
	s.info.acm_cols = 1u << level;
	s.info.acm_rows = rows;
	s.block = block;
	s.wrapbuf = wrapbuf;
	juggle_block(&s);
}

/* output_values (src/decode.c:657) */
int refprobe_output(int *src, unsigned char *dst, int n, int level, int be, int wordlen, int sgned)
{
	return output_values(src, dst, n, level, be, wordlen, sgned);
}
