/*
 * acmtool - command line front end of the MI355X-native ACM decoder.
 *
 * Command-line compatible with the reference tool (/root/reference/src/acmtool.c:
 * option letters :416, messages, WAV layout :193-229, zero padding :293-310,
 * header patcher :322-362) so that scripts written for it keep working; the
 * decoding itself goes through libacm.h and therefore through the GPU.
 *
 * Extension (no reference counterpart): -B decodes all listed files as ONE
 * batch on the GPU (acm_batch_decode) instead of one after the other.
 */
#include <errno.h>
#include <getopt.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include "acm_hip.h"
#include "libacm.h"

#define TOOL_BANNER "acmtool - libacm version " LIBACM_VERSION
#define IO_CHUNK    (16 * 1024)

struct settings {
	int raw;            /* -r: no WAV header */
	int force_chans;    /* -m / -s */
	int no_output;      /* -n */
	int quiet;          /* -q (also implied by -o -) */
};

static struct settings cfg;

/* "<name>: Length: m:ss Chans:c(h) Freq:f A:level/rows kbps:k" (reference :51-53) */
static void print_summary(const char *name, ACMStream *acm)
{
	const ACMInfo *inf = acm_info(acm);
	unsigned secs = acm_time_total(acm) / 1000;
	int kbps = (int)(acm_bitrate(acm) / 1000);

	if (cfg.quiet)
		return;
	printf("%s: Length:%2d:%02d Chans:%d(%d) Freq:%d A:%d/%d kbps:%d\n",
	       name, secs / 60, secs % 60, acm_channels(acm), acm->info.acm_channels,
	       acm_rate(acm), inf->acm_level, inf->acm_rows, kbps);
}

/* ---- RIFF/WAVE header: 44 bytes, PCM, 16 bit ---- */
static unsigned char *le16(unsigned char *p, unsigned v)
{
	p[0] = (unsigned char)(v & 0xFF);
	p[1] = (unsigned char)((v >> 8) & 0xFF);
	return p + 2;
}

static unsigned char *le32(unsigned char *p, unsigned v)
{
	return le16(le16(p, v & 0xFFFF), v >> 16);
}

static unsigned char *tag(unsigned char *p, const char *s)
{
	size_t n = strlen(s);
	memcpy(p, s, n);
	return p + n;
}

static int emit_wav_header(FILE *out, ACMStream *acm)
{
	unsigned char h[44], *p = h;
	unsigned chans = acm_channels(acm);
	unsigned rate = acm_rate(acm);
	unsigned data = acm_pcm_total(acm) * ACM_WORD * chans;

	p = tag(p, "RIFF");
	p = le32(p, 4 + 8 + 16 + 8 + data);
	p = tag(p, "WAVEfmt ");
	p = le32(p, 16);
	p = le16(p, 1);                          /* PCM */
	p = le16(p, chans);
	p = le32(p, rate);
	p = le32(p, rate * chans * ACM_WORD);    /* bytes per second */
	p = le16(p, ACM_WORD * 8 * chans / 8);   /* block align */
	p = le16(p, ACM_WORD * 8);
	p = tag(p, "data");
	p = le32(p, data);
	return fwrite(h, 1, sizeof(h), out) == sizeof(h) ? 0 : -1;
}

static char *swap_extension(const char *name, const char *ext)
{
	char *out = malloc(strlen(name) + strlen(ext) + 2);
	char *dot;
	strcpy(out, name);
	dot = strrchr(out, '.');
	if (dot)
		*dot = 0;
	strcat(out, ext);
	return out;
}

/* pad with silence up to the length the header promised (reference :293-310) */
static int pad_output(const char *name, FILE *out, char *buf, int done, int total)
{
	memset(buf, 0, IO_CHUNK);
	if (done < total)
		fprintf(stderr, "%s: adding filler_samples: %d\n", name, total - done);
	while (done < total) {
		int n = total - done < IO_CHUNK ? total - done : IO_CHUNK;
		if (out && (int)fwrite(buf, 1, (size_t)n, out) != n)
			break;
		done += n;
	}
	return done;
}

/*
 * Leaving costs as much as arriving: when a process that used the GPU ends, the kernel gives back its device memory and
 * unpins its pinned arenas before the parent's wait() returns - 0.16-0.27 s behind the last output byte of a 0.8 s batch run
 * (profiles/r3_cli_probe.txt), 0.08 s even for a program that only initialised the runtime.  With ACMTOOL_DETACH=1 the decode
 * therefore runs in a CHILD (forked before anything touches HIP); when every output is written and closed the child sends its
 * exit code through a pipe and goes on to die at its own pace, and the process the caller waits for returns that code at once.
 * Off by default (the GPU context of the detached child outlives the command, back-to-back commands stack them):
 * the numbers the documents lead with are those of the one-process run.  While detached:
 *   - a terminating signal sent to the waiting process is passed on to the child (a supervisor's kill, timeout(1)), and the
 *     child asks the kernel for SIGTERM should its parent die another way, until it has reported its code;
 *   - a child that ends any other way than through fast_exit() closes the pipe without a byte: the parent waits for it and
 *     passes its status on, 128 + signal for a signalled one;
 *   - never under a profiler or any other tool library that initialises the GPU before main() (rocprofv3 preloads one): a
 *     forked child cannot use a HIP runtime it inherited.
 */
static int done_fd = -1;
static volatile sig_atomic_t detached_child = 0;

static void pass_signal_on(int sig)
{
	if (detached_child > 0)
		kill((pid_t)detached_child, sig);
}

static int tool_library_preloaded(void)
{
	const char *pre = getenv("LD_PRELOAD");
	return getenv("ROCP_TOOL_LIBRARIES") || getenv("HSA_TOOLS_LIB") || getenv("ROCPROFILER_REGISTER_ENABLED") ||
	       (pre && (strstr(pre, "rocprof") || strstr(pre, "roctracer") || strstr(pre, "rocprofiler")));
}

static void detach_teardown(void)
{
	static const int passed_on[] = { SIGTERM, SIGINT, SIGHUP, SIGQUIT };
	int fds[2];
	pid_t pid;
	size_t k;
	const char *yes = getenv("ACMTOOL_DETACH");
	if (!yes || !atoi(yes) || tool_library_preloaded() || pipe(fds) != 0)
		return;
	fflush(NULL);
	pid = fork();
	if (pid < 0) {
		close(fds[0]);
		close(fds[1]);
		return;
	}
	if (pid == 0) {
		close(fds[0]);
		done_fd = fds[1];
		prctl(PR_SET_PDEATHSIG, SIGTERM);       /* until the code has been reported (fast_exit) */
		if (getppid() == 1)                     /* the parent is gone already */
			_exit(128 + SIGTERM);
		return;                 /* the child does the work */
	}
	close(fds[1]);
	detached_child = (sig_atomic_t)pid;
	for (k = 0; k < sizeof(passed_on) / sizeof(passed_on[0]); k++) {
		struct sigaction sa;
		memset(&sa, 0, sizeof(sa));
		sa.sa_handler = pass_signal_on;
		sigaction(passed_on[k], &sa, NULL);     /* no SA_RESTART: read() below returns EINTR and goes round again */
	}
	{
		unsigned char code = 0;
		ssize_t n;
		int st = 0;
		do
			n = read(fds[0], &code, 1);
		while (n < 0 && errno == EINTR);
		if (n == 1)
			_exit(code);
		while (waitpid(pid, &st, 0) < 0 && errno == EINTR)
			;
		_exit(WIFEXITED(st) ? WEXITSTATUS(st) : WIFSIGNALED(st) ? 128 + WTERMSIG(st) : 1);
	}
}

/* all outputs written and closed: end the process without the HIP runtime's teardown, and tell the waiting parent first */
static void fast_exit(int code)
{
	fflush(NULL);
	if (done_fd >= 0) {
		unsigned char c = (unsigned char)code;
		ssize_t n;
		prctl(PR_SET_PDEATHSIG, 0);             /* the parent leaves now; what is left is the teardown */
		do
			n = write(done_fd, &c, 1);
		while (n < 0 && errno == EINTR);
		/* whoever captures this command's output waits for every holder of the pipes to close them */
		close(0);
		close(1);
		close(2);
	}
	_exit(code);
}

static void decode_one(const char *src, const char *dst)
{
	ACMStream *acm;
	FILE *out = NULL;
	char *buf;
	int done = 0, total, got;
	int err = acm_open_file(&acm, src, cfg.force_chans);

	if (err < 0) {
		fprintf(stderr, "%s: %s\n", src, acm_strerror(err));
		return;
	}
	/* a long stream goes to the GPU: let it come up on a thread of the library's own while this one parses.  A short one is synthesised
	 * on the host by acm_read() itself (acmhip_host_synth_limit) and never pays for the runtime */
	if ((unsigned long long)acm_pcm_total(acm) * acm_channels(acm) >= acmhip_host_synth_limit())
		acmhip_prewarm();
	if (!cfg.no_output) {
		if (strcmp(dst, "-") == 0) {
			out = stdout;
			cfg.quiet = 1;
		} else {
			out = fopen(dst, "wb");
		}
		if (!out) {
			perror(dst);
			acm_close(acm);
			return;
		}
	}
	print_summary(src, acm);
	if (out && !cfg.raw && emit_wav_header(out, acm) < 0) {
		perror(dst);
		fclose(out);
		acm_close(acm);
		return;
	}

	buf = malloc(IO_CHUNK);
	total = (int)(acm_pcm_total(acm) * acm_channels(acm) * ACM_WORD);
	while (done < total) {
		got = acm_read_loop(acm, buf, IO_CHUNK / 2, 0, 2, 1);   /* 8 KiB requests, as the reference */
		if (got == 0)
			break;
		if (got < 0) {
			fprintf(stderr, "%s: %s\n", src, acm_strerror(got));
			break;
		}
		if (out && (int)fwrite(buf, 1, (size_t)got, out) != got) {
			fprintf(stderr, "%s: write error\n", dst);
			break;
		}
		done += got;
	}
	pad_output(src, out, buf, done, total);

	acm_close(acm);
	if (out)
		fclose(out);
	free(buf);
}

/* -M / -S: rewrite the channel count in the 14-byte header (reference :322-362) */
static void patch_channels(const char *name, int chans)
{
	static const unsigned char magic[4] = { 0x97, 0x28, 0x03, 0x01 };
	unsigned char hdr[14];
	int old;
	FILE *f = fopen(name, "rb+");

	if (!f) {
		perror(name);
		return;
	}
	if (fread(hdr, 1, sizeof(hdr), f) != sizeof(hdr)) {
		fprintf(stderr, "%s: cannot read header\n", name);
		goto done;
	}
	if (memcmp(hdr, magic, sizeof(magic)) != 0) {
		fprintf(stderr, "%s: not an ACM file\n", name);
		goto done;
	}
	old = hdr[8] | (hdr[9] << 8);
	if (old != 1 && old != 2) {
		fprintf(stderr, "%s: suspicios number of channels: %d\n", name, old);
		goto done;
	}
	if (fseek(f, 0, SEEK_SET)) {
		perror(name);
		goto done;
	}
	hdr[8] = (unsigned char)chans;
	if (fwrite(hdr, 1, sizeof(hdr), f) != sizeof(hdr))
		perror(name);
done:
	fclose(f);
}

static void info_one(const char *name)
{
	ACMStream *acm;
	int err = acm_open_file(&acm, name, cfg.force_chans);
	if (err < 0) {
		printf("%s: %s\n", name, acm_strerror(err));    /* stdout, like the reference (:375) */
		return;
	}
	print_summary(name, acm);
	acm_close(acm);
}

/* -B: all files through one acm_batch_decode() */
static int slurp(const char *name, unsigned char **data, size_t *len)
{
	FILE *f = fopen(name, "rb");
	long n;
	if (!f)
		return -1;
	fseek(f, 0, SEEK_END);
	n = ftell(f);
	fseek(f, 0, SEEK_SET);
	*data = malloc(n > 0 ? (size_t)n : 1);
	*len = n > 0 ? fread(*data, 1, (size_t)n, f) : 0;
	fclose(f);
	return 0;
}

/*
 * -B: batch mode (no reference counterpart; the files it leaves behind are the ones acmtool.c:231-316 writes one by one).
 * Four stages run beside each other on groups of files that fit a memory budget:
 *   reader   slurps the files of a group and reads their headers
 *   stager   bit-parses the group on the host pool (acm_batch_prestage: needs no device, so it runs from the first
 *            millisecond on, while the HIP runtime is still coming up - a third of the run on the 4000-file corpus)
 *   decoder  (this thread) gives the group one of three pinned PCM arenas and runs acm_batch_decode on it: what is left of
 *            that call is upload, synthesis and read-back
 *   writers  write the WAV / raw files from a small thread pool, then release the group and its arena
 * At most batch_groups_ahead groups exist at a time (read or parsed, waiting for the device), three of them with a PCM
 * arena, whatever the length of the file list; the arenas are pinned once and reused, and the read-back engine writes
 * every file's PCM straight into them (ACM_BATCH_PCM_PINNED).
 */
typedef struct barena {                 /* PCM of one group; pinned when the device hands it out, else malloc */
	void *mem;
	size_t cap;
	int pinned, busy;
} barena;

typedef struct bgroup {
	int first, n;                   /* names[first .. first + n) */
	acm_batch_item *items;
	barena *arena;
	size_t pcm_words;               /* arena words this group needs */
	acm_batch_prestaged *pre;       /* the group's files, bit-parsed (stager) */
	double pre_s;
	acm_batch_timing tm;
	int rc;
	struct bgroup *next;
} bgroup;

typedef struct bqueue {                 /* unbounded FIFO; the group budget below bounds what is in flight */
	pthread_mutex_t mu;
	pthread_cond_t cv;
	bgroup *head, *tail;
	int closed;
} bqueue;

static void bq_init(bqueue *q)
{
	pthread_mutex_init(&q->mu, NULL);
	pthread_cond_init(&q->cv, NULL);
	q->head = q->tail = NULL;
	q->closed = 0;
}

static void bq_push(bqueue *q, bgroup *g)       /* g == NULL closes the queue */
{
	pthread_mutex_lock(&q->mu);
	if (!g) {
		q->closed = 1;
	} else {
		g->next = NULL;
		if (q->tail)
			q->tail->next = g;
		else
			q->head = g;
		q->tail = g;
	}
	pthread_cond_broadcast(&q->cv);
	pthread_mutex_unlock(&q->mu);
}

static bgroup *bq_pop(bqueue *q)
{
	bgroup *g;
	pthread_mutex_lock(&q->mu);
	while (!q->head && !q->closed)
		pthread_cond_wait(&q->cv, &q->mu);
	g = q->head;
	if (g) {
		q->head = g->next;
		if (!q->head)
			q->tail = NULL;
	}
	pthread_mutex_unlock(&q->mu);
	return g;
}

static struct {
	int nfiles;
	char **names;
	size_t budget;                  /* bytes of file images + PCM per group */
	bqueue to_stage, to_decode, to_write;
	pthread_mutex_t mu;
	pthread_cond_t cv;
	int groups_alive;               /* read but not yet written out and freed */
	acm_batch_timing total;
	barena arenas[3];               /* one per group in flight, reused: pinning and unpinning a gigabyte costs 0.1-0.2 s each */
} bt;

#define BATCH_GROUPS_IN_FLIGHT 3        /* == number of bt.arenas */
static int batch_groups_ahead = 6;      /* groups read / parsed ahead of the device: file images + staged indices, ~1.3 x the budget each
					 * (ACMTOOL_GROUPS_AHEAD; what is touched here has to be given back at exit, 0.1 s per gigabyte) */
#define BATCH_WRITERS 8

/* ACMTOOL_BATCH_TRACE=1: wall-clock notes of the three stages on stderr (diagnostics; profiles/cli_batch_probe.sh) */
static int bt_trace;
static double bt_t0;
static double bt_now(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + ts.tv_nsec * 1e-9;
}
#define BT_NOTE(...) do { if (bt_trace) { fprintf(stderr, "[batch %.3f] ", bt_now() - bt_t0); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); } } while (0)

static void *batch_reader(void *unused)
{
	int i = 0;
	(void)unused;
	while (i < bt.nfiles) {
		bgroup *g = calloc(1, sizeof(*g));
		size_t bytes = 0, pcm_words = 0;
		pthread_mutex_lock(&bt.mu);
		while (bt.groups_alive >= batch_groups_ahead)
			pthread_cond_wait(&bt.cv, &bt.mu);
		bt.groups_alive++;
		pthread_mutex_unlock(&bt.mu);
		g->first = i;
		BT_NOTE("reader: group from file %d", i);
		g->items = calloc((size_t)(bt.nfiles - i), sizeof(*g->items));
		while (i < bt.nfiles && (g->n == 0 || bytes < bt.budget)) {
			acm_batch_item *it = &g->items[g->n];
			unsigned char *data = NULL;
			acm_stage_info si;
			if (slurp(bt.names[i], &data, &it->len) == 0) {
				it->data = data;
				if (acm_stage_probe(data, it->len, cfg.force_chans, &si) == ACM_OK) {
					/* what the file's bytes can hold, not what its header promises: a 19-byte file that claims
					 * 2^32-1 samples gets one block of arena, not 8 GB of pinned memory (the silence behind a
					 * short stream is written by write_one, not kept in the arena) */
					uint64_t fits = acm_batch_pcm_words(it, 1, cfg.force_chans);
					it->pcm_cap = si.total_values < fits ? si.total_values : (size_t)fits;
				}
				bytes += it->len + 2 * it->pcm_cap;
				pcm_words += (it->pcm_cap + 63) & ~(size_t)63;
			}
			g->n++;
			i++;
		}
		BT_NOTE("reader: %d files read, %zu MB of PCM to come", g->n, pcm_words * 2 >> 20);
		g->pcm_words = pcm_words;
		bq_push(&bt.to_stage, g);
	}
	bq_push(&bt.to_stage, NULL);
	return NULL;
}

/* the host half of the decode, group by group, as soon as the files are in memory */
static void *batch_stager(void *unused)
{
	bgroup *g;
	acm_batch_opts opts;
	(void)unused;
	memset(&opts, 0, sizeof(opts));
	opts.force_chans = cfg.force_chans;
	while ((g = bq_pop(&bt.to_stage)) != NULL) {
		if (acm_batch_prestage(g->items, (size_t)g->n, &opts, &g->pre, &g->pre_s) != ACMHIP_OK)
			g->pre = NULL;          /* out of memory: acm_batch_decode parses the group itself */
		BT_NOTE("stager: group of %d files parsed in %.3f s", g->n, g->pre_s);
		bq_push(&bt.to_decode, g);
	}
	bq_push(&bt.to_decode, NULL);
	return NULL;
}

/* one arena per group, every file's PCM on a 128-byte boundary inside it; pinned memory lets the read-back copy engine
 * write it without a bounce buffer.  Called by the decoder (the device is up by then: pinning needs the runtime) */
static void batch_give_arena(bgroup *g, int only_group)
{
	size_t at = 0;
	int k;
	if (g->pcm_words) {
		barena *a = NULL;
		pthread_mutex_lock(&bt.mu);
		for (;;) {
			for (k = 0; k < BATCH_GROUPS_IN_FLIGHT && !a; k++)
				if (!bt.arenas[k].busy)
					a = &bt.arenas[k];
			if (a)
				break;
			pthread_cond_wait(&bt.cv, &bt.mu);      /* the writers release them */
		}
		a->busy = 1;
		pthread_mutex_unlock(&bt.mu);
		if (a->cap < g->pcm_words * 2) {
			/* kept when big enough; a new one is sized for a whole budget unless this is the only group */
			size_t want = g->pcm_words * 2;
			if (!only_group && want < bt.budget)
				want = bt.budget;
			if (a->mem) {
				if (a->pinned)
					acmhip_host_free(a->mem);
				else
					free(a->mem);
			}
			a->pinned = acmhip_host_alloc(want, &a->mem) == ACMHIP_OK;
			if (!a->pinned)
				a->mem = malloc(want);
			a->cap = a->mem ? want : 0;
			if (!a->mem)
				fprintf(stderr, "acmtool: cannot allocate %zu MB for the PCM of %d files\n", want >> 20, g->n);
		}
		g->arena = a;
	}
	for (k = 0; k < g->n; k++) {
		if (g->items[k].pcm_cap && g->arena && g->arena->mem) {
			g->items[k].pcm = (int16_t *)g->arena->mem + at;
			at += (g->items[k].pcm_cap + 63) & ~(size_t)63;
		}
	}
}

typedef struct bwrite_job {
	bgroup *g;
	int next;                       /* next file of the group to write (under bt.mu) */
} bwrite_job;

static void write_one(const char *name, const acm_batch_item *it)
{
	char *dst = swap_extension(name, cfg.raw ? ".raw" : ".wav");
	FILE *out = fopen(dst, "wb");
	if (!out) {
		perror(dst);
	} else {
		unsigned chans = it->channels ? it->channels : 1;
		unsigned whole = it->total_values / chans * chans;      /* acm_pcm_total * channels */
		if (!cfg.raw) {
			unsigned char h[44], *p = h;                    /* acmtool.c:193-229 */
			p = tag(p, "RIFF");
			p = le32(p, 4 + 8 + 16 + 8 + whole * ACM_WORD);
			p = tag(p, "WAVEfmt ");
			p = le32(p, 16);
			p = le16(p, 1);
			p = le16(p, chans);
			p = le32(p, it->rate);
			p = le32(p, it->rate * chans * ACM_WORD);
			p = le16(p, ACM_WORD * 8 * chans / 8);
			p = le16(p, ACM_WORD * 8);
			p = tag(p, "data");
			p = le32(p, whole * ACM_WORD);
			fwrite(h, 1, sizeof(h), out);
		}
		{
			/* the decoded words, then silence for what the stream did not deliver (acmtool.c:293-310) */
			static const unsigned char zeros[8192];
			size_t have = it->words < whole ? (size_t)it->words : whole, rest = ((size_t)whole - have) * 2;
			fwrite(it->pcm, 2, have, out);
			while (rest) {
				size_t n = rest < sizeof(zeros) ? rest : sizeof(zeros);
				fwrite(zeros, 1, n, out);
				rest -= n;
			}
		}
		fclose(out);
	}
	free(dst);
}

static void *batch_write_worker(void *arg)
{
	bwrite_job *job = arg;
	for (;;) {
		int k;
		pthread_mutex_lock(&bt.mu);
		k = job->next++;
		pthread_mutex_unlock(&bt.mu);
		if (k >= job->g->n)
			return NULL;
		if (job->g->items[k].data && job->g->items[k].pcm && !cfg.no_output)
			write_one(bt.names[job->g->first + k], &job->g->items[k]);
	}
}

static void *batch_writer(void *unused)
{
	bgroup *g;
	(void)unused;
	while ((g = bq_pop(&bt.to_write)) != NULL) {
		pthread_t th[BATCH_WRITERS];
		bwrite_job job;
		int k, nth;
		/* messages in file order, exactly what a one-by-one run prints; then the files themselves in parallel */
		for (k = 0; k < g->n && g->rc == ACMHIP_OK; k++) {
			acm_batch_item *it = &g->items[k];
			const char *name = bt.names[g->first + k];
			unsigned chans, whole;
			if (!it->data) {
				fprintf(stderr, "%s: %s\n", name, acm_strerror(ACM_ERR_OPEN));
				continue;
			}
			if (!it->pcm) {
				if (it->pcm_cap)
					fprintf(stderr, "%s: out of memory\n", name);
				else
					fprintf(stderr, "%s: %s\n", name, acm_strerror(it->status));
				continue;
			}
			if (it->status < 0 && it->words == 0)
				fprintf(stderr, "%s: %s\n", name, acm_strerror(it->status));
			if (!cfg.quiet)
				printf("%s: Chans:%u Freq:%u A:%u/%u words:%llu/%u\n", name, it->channels, it->rate,
				       it->level, it->rows, (unsigned long long)it->words, it->total_values);
			if (cfg.no_output)
				continue;
			chans = it->channels ? it->channels : 1;
			whole = it->total_values / chans * chans;
			if (it->words < whole) {
				/* silence for what the stream did not deliver (acmtool.c:293-310) */
				fprintf(stderr, "%s: adding filler_samples: %d\n", name, (int)((whole - it->words) * ACM_WORD));
			}
		}
		BT_NOTE("writer: group of %d files", g->n);
		job.g = g;
		job.next = 0;
		nth = g->rc == ACMHIP_OK ? (g->n < BATCH_WRITERS ? g->n : BATCH_WRITERS) : 0;
		for (k = 0; k < nth; k++)
			pthread_create(&th[k], NULL, batch_write_worker, &job);
		for (k = 0; k < nth; k++)
			pthread_join(th[k], NULL);
		BT_NOTE("writer: files written");
		for (k = 0; k < g->n; k++)
			free((void *)g->items[k].data);
		free(g->items);
		BT_NOTE("writer: group released");
		pthread_mutex_lock(&bt.mu);
		if (g->arena)
			g->arena->busy = 0;
		free(g);
		bt.groups_alive--;
		pthread_cond_broadcast(&bt.cv);
		pthread_mutex_unlock(&bt.mu);
	}
	return NULL;
}

static int decode_batch(int nfiles, char **names)
{
	acm_batch_opts opts;
	acmhip_device *dev = NULL;
	pthread_t reader, stager, writer;
	bgroup *g;
	int rc, failed = 0;
	const char *mb = getenv("ACMTOOL_BATCH_MB"), *bb = getenv("ACMTOOL_BATCH_BYTES");

	memset(&opts, 0, sizeof(opts));
	opts.force_chans = cfg.force_chans;
	opts.parse = ACM_BATCH_PARSE_AUTO;
	memset(&bt, 0, sizeof(bt));
	bt.nfiles = nfiles;
	bt.names = names;
	bt.budget = (size_t)(mb && atoi(mb) > 0 ? atoi(mb) : 128) << 20;       /* file images + PCM per group: small groups keep the pinned
											 * footprint (0.2 s per gigabyte to pin, as much to unpin) low */
	if (bb && atol(bb) > 0)
		bt.budget = (size_t)atol(bb);
	bq_init(&bt.to_stage);
	bq_init(&bt.to_decode);
	bq_init(&bt.to_write);
	pthread_mutex_init(&bt.mu, NULL);
	pthread_cond_init(&bt.cv, NULL);

	if (getenv("ACMTOOL_GROUPS_AHEAD") && atoi(getenv("ACMTOOL_GROUPS_AHEAD")) >= BATCH_GROUPS_IN_FLIGHT)
		batch_groups_ahead = atoi(getenv("ACMTOOL_GROUPS_AHEAD"));
	bt_trace = getenv("ACMTOOL_BATCH_TRACE") != NULL;
	bt_t0 = bt_now();
	/* the reader starts first: it reads the first group's files while this thread brings the HIP runtime up (~0.2 s) */
	pthread_create(&reader, NULL, batch_reader, NULL);
	pthread_create(&stager, NULL, batch_stager, NULL);
	pthread_create(&writer, NULL, batch_writer, NULL);
	rc = acmhip_device_open(0, NULL, &dev);
	if (rc != ACMHIP_OK) {
		fprintf(stderr, "acmtool: batch decode failed: %s\n", acmhip_last_error());
		failed = 1;             /* the groups still flow through the queues so that both threads end */
	}
	BT_NOTE("device open");
	while ((g = bq_pop(&bt.to_decode)) != NULL) {
		BT_NOTE("decoder: group of %d files", g->n);
		batch_give_arena(g, g->first == 0 && g->n == nfiles);
		BT_NOTE("decoder: arena ready");
		opts.flags = (g->arena && g->arena->pinned) ? ACM_BATCH_PCM_PINNED : 0;
		opts.prestaged = g->pre;
		g->rc = failed ? ACMHIP_ERR_ARG : acm_batch_decode(dev, g->items, (size_t)g->n, &opts, &g->tm);
		BT_NOTE("decoder: call returned");
		acm_batch_prestage_free(g->pre);
		g->pre = NULL;
		g->tm.stage_s += g->pre_s;      /* the parsing happened in the stager */
		BT_NOTE("decoder: done (parse %.3f h2d %.3f kernel %.3f d2h %.3f total %.3f, device-parsed %llu)", g->tm.stage_s, g->tm.h2d_s,
			g->tm.kernel_s, g->tm.d2h_s, g->tm.total_s, (unsigned long long)g->tm.device_parsed);
		if (g->rc != ACMHIP_OK && !failed) {
			fprintf(stderr, "acmtool: batch decode failed: %s\n", acmhip_last_error());
			failed = 1;
		}
		if (g->rc == ACMHIP_OK) {
			bt.total.samples += g->tm.samples;
			bt.total.alloc_s += g->tm.alloc_s;
			bt.total.stage_s += g->tm.stage_s;
			bt.total.h2d_s += g->tm.h2d_s;
			bt.total.kernel_s += g->tm.kernel_s;
			bt.total.d2h_s += g->tm.d2h_s;
			bt.total.total_s += g->tm.total_s;
		}
		bq_push(&bt.to_write, g);
	}
	bq_push(&bt.to_write, NULL);
	pthread_join(reader, NULL);
	pthread_join(stager, NULL);
	pthread_join(writer, NULL);
	if (!cfg.quiet && !failed)
		printf("batch: %llu samples, alloc %.3fs parse %.3fs h2d %.3fs kernel %.3fs d2h %.3fs total %.3fs\n",
		       (unsigned long long)bt.total.samples, bt.total.alloc_s, bt.total.stage_s, bt.total.h2d_s,
		       bt.total.kernel_s, bt.total.d2h_s, bt.total.total_s);
	/* every output file is closed: leave without unpinning the arenas (as slow as pinning them) or taking the HIP
	 * runtime down (~0.08 s) - the process ends here anyway */
	BT_NOTE("done");
	if (bt_trace) {
		struct timespec ts;
		clock_gettime(CLOCK_REALTIME, &ts);
		fprintf(stderr, "[batch] wall clock at exit %.3f, the batch started %.3f s before\n", ts.tv_sec + ts.tv_nsec * 1e-9, bt_now() - bt_t0);
	}
	fast_exit(failed);
	return failed;
}

static void usage(int code)
{
	printf("%s\n", TOOL_BANNER);
	printf("Play:   acmtool -p [-q][-m|-s] acmfile [acmfile ...]\n");
	printf("Decode: acmtool -d [-q][-m|-s] [-r|-n] -o wavfile acmfile\n");
	printf("        acmtool -d [-q][-m|-s] [-r|-n] acmfile [acmfile ...]\n");
	printf("Other:  acmtool -i acmfile [acmfile ...]\n");
	printf("        acmtool -M|-S acmfile [acmfile ...]\n");
	printf("Commands:\n");
	printf("  -p     play file(s)\n");
	printf("  -d     decode audio into WAV files\n");
	printf("  -i     show info about ACM files\n");
	printf("  -M     modify ACM header to have 1 channel\n");
	printf("  -S     modify ACM header to have 2 channels\n");
	printf("Switches:\n");
	printf("  -m     force mono\n");
	printf("  -s     force stereo (default)\n");
	printf("  -r     raw output - no wav header\n");
	printf("  -q     be quiet\n");
	printf("  -n     no output - for benchmarking\n");
	printf("  -o FN  output to file, can be used if single source file\n");
	exit(code);
}

int main(int argc, char *argv[])
{
	enum { CMD_NONE = 0, CMD_PLAY = 1, CMD_DECODE = 2, CMD_INFO = 4, CMD_CHANS = 8 };
	int cmds = 0, ncmds = 0, set_chans = 0, batch = 0, c, i;
	const char *outname = NULL;

	while ((c = getopt(argc, argv, "pdiMSqhrmsnvo:B")) != -1) {
		switch (c) {
		case 'p': cmds |= CMD_PLAY; break;
		case 'd': cmds |= CMD_DECODE; break;
		case 'i': cmds |= CMD_INFO; break;
		case 'M': cmds |= CMD_CHANS; set_chans = 1; break;
		case 'S': cmds |= CMD_CHANS; set_chans = 2; break;
		case 'q': cfg.quiet = 1; break;
		case 'm': cfg.force_chans = 1; break;
		case 's': cfg.force_chans = 2; break;
		case 'r': cfg.raw = 1; break;
		case 'n': cfg.no_output = 1; break;
		case 'o': outname = optarg; break;
		case 'B': batch = 1; break;
		case 'h': usage(0); break;
		case 'v':
			printf("%s\n", TOOL_BANNER);
			return 0;
		default:
			fprintf(stderr, "bad arg: -%c\n", c);
			usage(1);
		}
	}
	for (i = 1; i <= CMD_CHANS; i <<= 1)
		ncmds += (cmds & i) != 0;
	if (ncmds != 1) {
		fprintf(stderr, "only one command at a time please\n");
		usage(1);
	}

	if (cmds == CMD_PLAY) {
		/* live playback needs libao, which this build never links (reference :479-482) */
		fprintf(stderr, "For audio output, please compile with libao.\n");
		return 1;
	}
	if (cmds == CMD_INFO) {
		for (i = optind; i < argc; i++)
			info_one(argv[i]);
		return 0;
	}
	if (cmds == CMD_CHANS) {
		for (i = optind; i < argc; i++)
			patch_channels(argv[i], set_chans);
		return 0;
	}

	/* decode */
	if (optind == argc)
		usage(1);
	detach_teardown();              /* from here on this is the child: nothing has touched the GPU yet */
	/* ACMTOOL_HOST_LIMIT=<samples>: streams shorter than this are synthesised on the host (the library's default: 128 M; 0 = every
	 * stream on the GPU where there is one) */
	if (getenv("ACMTOOL_HOST_LIMIT"))
		acmhip_set_host_synth_limit(strtoull(getenv("ACMTOOL_HOST_LIMIT"), NULL, 10));
	if (batch)
		return decode_batch(argc - optind, argv + optind);
	if (outname) {
		if (optind + 1 != argc)
			usage(1);
		decode_one(argv[optind], outname);
		fast_exit(0);
	}
	for (i = optind; i < argc; i++) {
		char *dst = swap_extension(argv[i], cfg.raw ? ".raw" : ".wav");
		decode_one(argv[i], dst);
		free(dst);
	}
	fast_exit(0);
	return 0;
}
