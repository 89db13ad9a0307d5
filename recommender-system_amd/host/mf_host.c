/*
 * mf_host.c -- host-side helpers of the MI355X matFact build (see include/matfact_host.h).
 * Pure C, no GPU dependency.  Each function cites the reference code whose behaviour it keeps.
 */
#define _GNU_SOURCE
#include "../../include/matfact_host.h"

#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ parsing */

const char *mf_host_parse_strerror(int status)
{
	switch (status) {
	case MF_PARSE_OK: return "ok";
	case MF_PARSE_OPEN: return "Unable to open input file.";           /* matFact.c:76 */
	case MF_PARSE_INT: return "Error in int argument.";                /* util.c:14 */
	case MF_PARSE_DOUBLE: return "Error in double argument.";          /* util.c:20 */
	case MF_PARSE_THREE_INTS: return "Error in multiple int argument."; /* util.c:26 */
	case MF_PARSE_ENTRY: return "Error in non-zero entry.";            /* util.c:32 */
	case MF_PARSE_CLOSE: return "Unable to close input file.";         /* matFact.c:109 */
	case MF_PARSE_NOMEM: return "Out of memory.";
	default: return "Unknown parse error.";
	}
}

/* the tokens fscanf("%d") / fscanf("%lf") would take: skip white space, then the longest valid prefix */
static int take_int(const char **cur, const char *end, int *out)
{
	const char *p = *cur;
	while (p < end && (*p == ' ' || (*p >= '\t' && *p <= '\r'))) ++p;
	if (p >= end) return 0;
	int neg = 0;
	if (*p == '+' || *p == '-') neg = (*p++ == '-');
	if (p >= end || *p < '0' || *p > '9') return 0;
	long long v = 0;
	while (p < end && *p >= '0' && *p <= '9') {
		v = v * 10 + (*p++ - '0');
		if (v > 0x7fffffffLL + 1) v = 0x7fffffffLL + 1;   /* saturate; the reference overflows (UB) */
	}
	*out = (int) (neg ? -v : v);
	*cur = p;
	return 1;
}

static int take_double(const char **cur, const char *end, double *out)
{
	const char *p = *cur;
	while (p < end && (*p == ' ' || (*p >= '\t' && *p <= '\r'))) ++p;
	if (p >= end) return 0;
	char *stop = NULL;
	const double v = strtod(p, &stop);   /* the buffer is NUL-terminated by the callers */
	if (stop == p) return 0;
	*out = v;
	*cur = stop;
	return 1;
}

static int parse_buffer_sequential(const char *text, size_t len, mf_problem *p)
{
	memset(p, 0, sizeof *p);
	const char *cur = text, *end = text + len;
	int iters, feats, users, items, nnz;
	double alpha;
	if (!take_int(&cur, end, &iters)) return MF_PARSE_INT;
	if (!take_double(&cur, end, &alpha)) return MF_PARSE_DOUBLE;
	if (!take_int(&cur, end, &feats)) return MF_PARSE_INT;
	if (!take_int(&cur, end, &users) || !take_int(&cur, end, &items) || !take_int(&cur, end, &nnz))
		return MF_PARSE_THREE_INTS;
	mf_entry *e = malloc(sizeof(mf_entry) * (size_t) (nnz > 0 ? nnz : 1));
	if (!e) return MF_PARSE_NOMEM;
	for (int n = 0; n < nnz; n++) {
		int r, c;
		double v;
		if (!take_int(&cur, end, &r) || !take_int(&cur, end, &c) || !take_double(&cur, end, &v)) {
			free(e);
			return MF_PARSE_ENTRY;
		}
		e[n].row = r;
		e[n].col = c;
		e[n].value = v;
	}
	p->iters = iters;
	p->alpha = alpha;
	p->features = feats;
	p->users = users;
	p->items = items;
	p->nnz = nnz > 0 ? nnz : 0;
	p->entries = e;
	return MF_PARSE_OK;
}

/* ---- parallel body parse.  The entries are 3*nnz white-space separated tokens; threads count the
 * tokens of their slice of the buffer, a prefix sum tells every slice which token it starts with, and the slices
 * parse in parallel.  A token must be consumed WHOLE by its field (fscanf would split "3.5" read with %d into "3"
 * and ".5"): anything else -- and any failure -- abandons the fast path, and the sequential parser above decides,
 * so results and error strings are always those of the sequential (== fscanf) semantics. */
static inline int is_space(char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

/* whole-token int: optional sign, digits only */
static int token_int(const char *b, const char *e, int *out)
{
	int neg = 0;
	if (b < e && (*b == '+' || *b == '-')) neg = (*b++ == '-');
	if (b >= e || e - b > 10) return 0;
	long long v = 0;
	for (; b < e; ++b) {
		if (*b < '0' || *b > '9') return 0;
		v = v * 10 + (*b - '0');
	}
	if (v > 0x7fffffffLL) return 0;
	*out = (int) (neg ? -v : v);
	return 1;
}

/* whole-token decimal "ddd[.ddd]" with <= 15 significant digits: mantissa and 10^k are exact doubles, so the one
 * IEEE division is the correctly rounded value (Clinger's fast path) == strtod's result; anything else -> strtod */
static int token_double(const char *b, const char *e, double *out)
{
	static const double p10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
	const char *s = b;
	int neg = 0, digits = 0, frac = 0, seen_dot = 0;
	if (s < e && (*s == '+' || *s == '-')) neg = (*s++ == '-');
	long long m = 0;
	for (; s < e; ++s) {
		if (*s == '.' && !seen_dot) {
			seen_dot = 1;
		} else if (*s >= '0' && *s <= '9') {
			if (++digits > 15) break;
			m = m * 10 + (*s - '0');
			frac += seen_dot;
		} else {
			break;
		}
	}
	if (s == e && digits > 0 && digits <= 15) {
		const double v = (double) m / p10[frac];
		*out = neg ? -v : v;
		return 1;
	}
	char tmp[64];
	const size_t n = (size_t) (e - b);
	if (n == 0 || n >= sizeof tmp) return 0;
	memcpy(tmp, b, n);
	tmp[n] = '\0';
	char *stop = NULL;
	const double v = strtod(tmp, &stop);
	if (stop != tmp + n) return 0;
	*out = v;
	return 1;
}

int mf_host_parse_buffer(const char *text, size_t len, mf_problem *p)
{
	memset(p, 0, sizeof *p);
	const char *cur = text, *end = text + len;
	int iters, feats, users, items, nnz;
	double alpha;
	if (!take_int(&cur, end, &iters) || !take_double(&cur, end, &alpha) || !take_int(&cur, end, &feats) ||
	    !take_int(&cur, end, &users) || !take_int(&cur, end, &items) || !take_int(&cur, end, &nnz) || nnz < (1 << 16) ||
	    (cur < end && !is_space(*cur)))
		return parse_buffer_sequential(text, len, p);   /* small inputs and every irregular header */

	int nthr = 1;
#ifdef _OPENMP
	nthr = omp_get_max_threads();
#endif
	if (nthr > 64) nthr = 64;
	const size_t body = (size_t) (end - cur);
	size_t start[65], count[65];
	for (int t = 0; t <= nthr; t++) {
		const char *s = cur + body * (size_t) t / (size_t) nthr;
		if (t > 0 && t < nthr)
			while (s < end && !is_space(*s)) ++s;   /* never cut a token */
		start[t] = (size_t) (s - text);
	}
	start[nthr] = len;
	#pragma omp parallel for schedule(static, 1) num_threads(nthr)
	for (int t = 0; t < nthr; t++) {
		size_t c = 0;
		const char *s = text + start[t], *e = text + start[t + 1];
		while (s < e) {
			while (s < e && is_space(*s)) ++s;
			if (s >= e) break;
			++c;
			while (s < e && !is_space(*s)) ++s;
		}
		count[t] = c;
	}
	size_t first[65];
	first[0] = 0;
	for (int t = 0; t < nthr; t++) first[t + 1] = first[t] + count[t];
	const size_t need = 3 * (size_t) nnz;
	if (first[nthr] < need) return parse_buffer_sequential(text, len, p);   /* too few tokens: exact error below */

	mf_entry *ent = malloc(sizeof(mf_entry) * (size_t) nnz);
	if (!ent) return MF_PARSE_NOMEM;
	int bad = 0;
	#pragma omp parallel for schedule(static, 1) num_threads(nthr) reduction(| : bad)
	for (int t = 0; t < nthr; t++) {
		size_t tok = first[t];
		const char *s = text + start[t], *e = text + start[t + 1];
		while (s < e && tok < need && !bad) {
			while (s < e && is_space(*s)) ++s;
			if (s >= e) break;
			const char *b = s;
			while (s < e && !is_space(*s)) ++s;
			mf_entry *en = ent + tok / 3;
			const int field = (int) (tok % 3);
			if (field == 0)
				bad |= !token_int(b, s, &en->row);
			else if (field == 1)
				bad |= !token_int(b, s, &en->col);
			else
				bad |= !token_double(b, s, &en->value);
			++tok;
		}
	}
	if (bad) {
		free(ent);
		return parse_buffer_sequential(text, len, p);
	}
	p->iters = iters;
	p->alpha = alpha;
	p->features = feats;
	p->users = users;
	p->items = items;
	p->nnz = nnz;
	p->entries = ent;
	return MF_PARSE_OK;
}

int mf_host_parse_file(const char *path, mf_problem *p)
{
	memset(p, 0, sizeof *p);
	FILE *fp = fopen(path, "r");
	if (!fp) return MF_PARSE_OPEN;
	size_t cap = 1 << 16, len = 0;
	char *buf = malloc(cap + 1);
	if (!buf) {
		fclose(fp);
		return MF_PARSE_NOMEM;
	}
	for (;;) {
		const size_t got = fread(buf + len, 1, cap - len, fp);
		len += got;
		if (got == 0) break;
		if (len == cap) {
			cap *= 2;
			char *nb = realloc(buf, cap + 1);
			if (!nb) {
				free(buf);
				fclose(fp);
				return MF_PARSE_NOMEM;
			}
			buf = nb;
		}
	}
	buf[len] = '\0';
	if (fclose(fp) == EOF) {
		free(buf);
		return MF_PARSE_CLOSE;
	}
	const int rc = mf_host_parse_buffer(buf, len, p);
	free(buf);
	return rc;
}

/* ---- binary cache of parsed `.in` files: the per-token fscanf of util.c:30-34 is the wall-clock floor
 * of a repeat run once the iterations are on the GPU.  A cache file holds the header and the entries exactly as the
 * parser produced them (the reference's own 16-byte non_zero_entry structs) under a key made of the CONTENT of the
 * `.in` (64-bit FNV-1a over 1-MiB blocks, hashed in parallel) and its size: an edited file misses unless its text
 * collides in 64 bits at equal length (not a cryptographic guarantee).  A hit maps the cache file, checks the hash of
 * the stored entries against the header (a damaged body is a miss) and points mf_problem.entries into the mapping (no
 * copy, no parse); anything unexpected -- no directory, short file, other magic, other key, other body hash -- falls
 * back to the parser.  The entries are served as parsed: like the reference's parser (util.c:30-34) this layer does not
 * range-check indices; mf_plan_create / mf_backend_run* do, and answer MF_ERR_ARGUMENT. */
typedef struct {
	char magic[8];              /* "MFCACHE2" */
	uint64_t content_hash;      /* of the .in text */
	uint64_t content_size;
	int32_t users, items, features, iters;
	double alpha;
	int64_t nnz;
	uint64_t body_hash;         /* of the entries as stored: a damaged cache body is a miss, not a wrong answer */
} mf_cache_header;              /* 64 bytes: entries start at byte 64 */

/* live mappings of cache files (mf_problem.entries points 64 bytes into one): a list under a mutex, so any number of
 * problems may be open and concurrent callers do not race */
typedef struct mf_mapping {
	void *base;
	size_t size;
	struct mf_mapping *next;
} mf_mapping;
static mf_mapping *g_mappings;
static pthread_mutex_t g_mappings_mu = PTHREAD_MUTEX_INITIALIZER;

static uint64_t fnv1a64(const unsigned char *p, size_t n)
{
	uint64_t h = 1469598103934665603ull;
	for (size_t i = 0; i < n; ++i) {
		h ^= p[i];
		h *= 1099511628211ull;
	}
	return h;
}

static uint64_t content_hash(const unsigned char *p, size_t n)
{
	const size_t blk = (size_t) 1 << 20, nblk = (n + blk - 1) / blk;
	uint64_t *part = malloc(sizeof(uint64_t) * (nblk ? nblk : 1));
	if (!part) return fnv1a64(p, n);
	#pragma omp parallel for schedule(static)
	for (long b = 0; b < (long) nblk; ++b) {
		const size_t lo = (size_t) b * blk, hi = lo + blk < n ? lo + blk : n;
		part[b] = fnv1a64(p + lo, hi - lo);
	}
	uint64_t h = 1469598103934665603ull ^ (uint64_t) n;
	for (size_t b = 0; b < nblk; ++b) h = (h ^ part[b]) * 1099511628211ull;
	free(part);
	return h;
}

int mf_host_parse_file_cached(const char *path, const char *cache_dir, mf_problem *p, int *cache_hit)
{
	if (cache_hit) *cache_hit = 0;
	if (!cache_dir || !cache_dir[0]) return mf_host_parse_file(path, p);
	memset(p, 0, sizeof *p);
	const int fd = open(path, O_RDONLY);
	if (fd < 0) return MF_PARSE_OPEN;
	struct stat st;
	if (fstat(fd, &st) != 0 || st.st_size <= 0) {
		close(fd);
		return mf_host_parse_file(path, p);
	}
	const size_t len = (size_t) st.st_size;
	void *text = mmap(NULL, len, PROT_READ, MAP_PRIVATE, fd, 0);
	close(fd);
	if (text == MAP_FAILED) return mf_host_parse_file(path, p);
	const uint64_t key = content_hash(text, len);
	char name[4096];
	snprintf(name, sizeof name, "%s/%016llx-%zu.mfcache", cache_dir, (unsigned long long) key, len);

	/* ---- hit? */
	const int cfd = open(name, O_RDONLY);
	if (cfd >= 0) {
		struct stat cst;
		mf_cache_header h;
		if (fstat(cfd, &cst) == 0 && (size_t) cst.st_size >= sizeof h &&
		    pread(cfd, &h, sizeof h, 0) == (ssize_t) sizeof h && memcmp(h.magic, "MFCACHE2", 8) == 0 &&
		    h.content_hash == key && h.content_size == (uint64_t) len && h.nnz >= 0 && h.users >= 0 && h.items >= 0 &&
		    (uint64_t) cst.st_size == sizeof h + (uint64_t) h.nnz * sizeof(mf_entry)) {
			void *map = mmap(NULL, (size_t) cst.st_size, PROT_READ, MAP_PRIVATE, cfd, 0);
			mf_mapping *node = map != MAP_FAILED ? malloc(sizeof *node) : NULL;
			/* the body is trusted only as far as its own hash: what was parsed is what is served */
			if (node && content_hash((const unsigned char *) map + sizeof h, (size_t) h.nnz * sizeof(mf_entry)) == h.body_hash) {
				node->base = map;
				node->size = (size_t) cst.st_size;
				pthread_mutex_lock(&g_mappings_mu);
				node->next = g_mappings;
				g_mappings = node;
				pthread_mutex_unlock(&g_mappings_mu);
				p->users = h.users;
				p->items = h.items;
				p->features = h.features;
				p->iters = h.iters;
				p->alpha = h.alpha;
				p->nnz = h.nnz;
				p->entries = (const mf_entry *) ((const char *) map + sizeof h);   /* nnz == 0: one past the header, never read */
				close(cfd);
				munmap(text, len);
				if (cache_hit) *cache_hit = 1;
				return MF_PARSE_OK;
			}
			free(node);
			if (map != MAP_FAILED) munmap(map, (size_t) cst.st_size);
		}
		close(cfd);
	}

	/* ---- miss: parse (the parser wants a NUL behind the text), then write the cache beside a temporary name */
	char *buf = malloc(len + 1);
	if (!buf) {
		munmap(text, len);
		return MF_PARSE_NOMEM;
	}
	memcpy(buf, text, len);
	buf[len] = '\0';
	munmap(text, len);
	const int rc = mf_host_parse_buffer(buf, len, p);
	free(buf);
	if (rc != MF_PARSE_OK) return rc;
	char tmp[4200];
	snprintf(tmp, sizeof tmp, "%s.%ld.tmp", name, (long) getpid());
	FILE *out = fopen(tmp, "wb");
	if (out) {
		mf_cache_header h;
		memset(&h, 0, sizeof h);
		memcpy(h.magic, "MFCACHE2", 8);
		h.content_hash = key;
		h.content_size = (uint64_t) len;
		h.users = p->users;
		h.items = p->items;
		h.features = p->features;
		h.iters = p->iters;
		h.alpha = p->alpha;
		h.nnz = p->nnz;
		const size_t n = (size_t) p->nnz;
		h.body_hash = content_hash((const unsigned char *) p->entries, n * sizeof(mf_entry));
		const int ok = fwrite(&h, sizeof h, 1, out) == 1 && (n == 0 || fwrite(p->entries, sizeof(mf_entry), n, out) == n);
		if (fclose(out) == 0 && ok)
			(void) rename(tmp, name);   /* atomic: a reader sees the old state or the whole file */
		else
			(void) remove(tmp);
	}
	return MF_PARSE_OK;
}

void mf_host_free_problem(mf_problem *p)
{
	if (!p) return;
	/* a cached problem points sizeof(header) bytes into its mapping -- also when it has no entries at all */
	mf_mapping *hit = NULL;
	if (p->entries) {
		const char *want = (const char *) p->entries - sizeof(mf_cache_header);
		pthread_mutex_lock(&g_mappings_mu);
		for (mf_mapping **q = &g_mappings; *q; q = &(*q)->next)
			if ((const char *) (*q)->base == want) {
				hit = *q;
				*q = hit->next;
				break;
			}
		pthread_mutex_unlock(&g_mappings_mu);
	}
	if (hit) {
		munmap(hit->base, hit->size);
		free(hit);
	} else
		free((void *) p->entries);
	p->entries = NULL;
	p->nnz = 0;
}

/* ------------------------------------------------------------------------- glibc TYPE_3 random() */

/* srandom(): r[0] = seed (0 -> 1); r[i] = 16807 * r[i-1] mod (2^31 - 1) by Schrage's split; front cursor
 * 3 ahead of the back cursor; the first 310 outputs are discarded.  random(): ring[f] += ring[b];
 * result = that word >> 1.  (srandom(0) in mat2d.c:62 therefore behaves as seed 1.) */
void mf_host_srandom(mf_rand *g, unsigned seed)
{
	if (seed == 0) seed = 1;
	int32_t word = (int32_t) seed;
	g->ring[0] = word;
	for (int i = 1; i < 31; i++) {
		const long hi = word / 127773, lo = word % 127773;
		long w = 16807 * lo - 2836 * hi;
		if (w < 0) w += 2147483647;
		word = (int32_t) w;
		g->ring[i] = word;
	}
	g->f = 3;
	g->b = 0;
	for (int i = 0; i < 310; i++) (void) mf_host_random(g);
}

int32_t mf_host_random(mf_rand *g)
{
	const uint32_t v = (uint32_t) g->ring[g->f] + (uint32_t) g->ring[g->b];
	g->ring[g->f] = (int32_t) v;
	if (++g->f >= 31) g->f = 0;
	if (++g->b >= 31) g->b = 0;
	return (int32_t) (v >> 1);
}

#define MF_RAND_MAX 2147483647

/* ---- jump-ahead.  The generator is the linear recurrence s[n] = s[n-31] + s[n-3] (mod 2^32) over a 31-word
 * state, so "advance by N draws" is a multiplication of the state by the N-th power of its 31x31 companion
 * matrix (uint32 arithmetic IS arithmetic mod 2^32).  With the powers A^(2^i) at hand a jump costs ~30 small
 * matrix-vector products, which lets every thread start its slice of the initial factors at the right draw:
 * the reference draws 1e9 numbers one after the other for its largest sample (7.6 s on one core). */
typedef struct { uint32_t m[31][31]; } mf_mat31;

static void mat31_mul(mf_mat31 *c, const mf_mat31 *a, const mf_mat31 *b)
{
	for (int i = 0; i < 31; i++)
		for (int j = 0; j < 31; j++) {
			uint32_t acc = 0;
			for (int k = 0; k < 31; k++) acc += a->m[i][k] * b->m[k][j];
			c->m[i][j] = acc;
		}
}

#define MF_JUMP_BITS 48
typedef struct { mf_mat31 pw[MF_JUMP_BITS]; } mf_jump_table;

static void jump_table_init(mf_jump_table *t)
{
	/* state vector w[i] = s[n-31+i] (oldest first); one draw: w'[i] = w[i+1], w'[30] = w[0] + w[28] */
	memset(&t->pw[0], 0, sizeof t->pw[0]);
	for (int i = 0; i < 30; i++) t->pw[0].m[i][i + 1] = 1;
	t->pw[0].m[30][0] = 1;
	t->pw[0].m[30][28] = 1;
	for (int b = 1; b < MF_JUMP_BITS; b++) mat31_mul(&t->pw[b], &t->pw[b - 1], &t->pw[b - 1]);
}

/* g := g advanced by n draws */
static void rand_jump(mf_rand *g, const mf_jump_table *t, uint64_t n)
{
	uint32_t w[31], v[31];
	for (int i = 0; i < 31; i++) w[i] = (uint32_t) g->ring[(g->f + i) % 31];
	for (int b = 0; b < MF_JUMP_BITS && (n >> b); b++) {
		if (!((n >> b) & 1)) continue;
		for (int i = 0; i < 31; i++) {
			uint32_t acc = 0;
			for (int k = 0; k < 31; k++) acc += t->pw[b].m[i][k] * w[k];
			v[i] = acc;
		}
		memcpy(w, v, sizeof w);
	}
	for (int i = 0; i < 31; i++) g->ring[i] = (int32_t) w[i];
	g->f = 0;      /* the oldest word is the next one to be replaced ... */
	g->b = 28;     /* ... and the partner lies three draws back: b = f - 3 (mod 31) */
}

void mf_host_init_factors_block(int users, int items, int features, int u0, int count, double *L_block,
                                double *R)
{
	mf_rand g0;
	mf_host_srandom(&g0, 0);
	const double norm = (double) features;
	const uint64_t K = (uint64_t) features;
	const int64_t draws = (int64_t) count * features + (R ? (int64_t) items * features : 0);
	if (draws < (1 << 22)) {
		/* below ~4e6 draws (30 ms) one generator in draw order beats starting a thread team */
		mf_rand g = g0;
		for (int64_t u = 0; u < users; u++) {
			if (u >= u0 && u < (int64_t) u0 + count) {
				double *dst = L_block + (u - u0) * features;
				for (int k = 0; k < features; k++)
					dst[k] = ((double) mf_host_random(&g) / (double) MF_RAND_MAX) / norm;
			} else {
				for (int k = 0; k < features; k++) (void) mf_host_random(&g);
			}
		}
		if (R)
			for (int k = 0; k < features; k++)
				for (int64_t j = 0; j < items; j++)
					R[j * features + k] = ((double) mf_host_random(&g) / (double) MF_RAND_MAX) / norm;
		return;
	}
	mf_jump_table *tab = malloc(sizeof *tab);
	if (!tab) abort();
	jump_table_init(tab);
	int want_threads = (int) (draws >> 20);   /* at least ~1e6 draws per thread, at most 32 threads */
	if (want_threads > 32) want_threads = 32;
	if (want_threads < 1) want_threads = 1;
#ifdef _OPENMP
	if (want_threads > omp_get_max_threads()) want_threads = omp_get_max_threads();
#endif
	/* L[u][k] in row-major draw order (mat2d.c:65-67): draw number u*K + k; rows of the block are cut over the threads */
#pragma omp parallel num_threads(want_threads)
	{
#ifdef _OPENMP
		const int nt = omp_get_num_threads(), tid = omp_get_thread_num();
#else
		const int nt = 1, tid = 0;
#endif
		const int64_t r0 = (int64_t) count * tid / nt, r1 = (int64_t) count * (tid + 1) / nt;
		if (r1 > r0) {
			mf_rand g = g0;
			rand_jump(&g, tab, ((uint64_t) u0 + (uint64_t) r0) * K);
			for (int64_t t = r0 * features; t < r1 * features; t++)
				L_block[t] = ((double) mf_host_random(&g) / (double) MF_RAND_MAX) / norm;
		}
		/* R_init[k][j] in row-major draw order (mat2d.c:69-71), draw number users*K + k*items + j, stored
		 * transposed (mat2d.c:115-124).  Items are cut over the threads; a thread keeps one generator PER k,
		 * positioned at (k, j0), with the 31-word states stored word-major: all K generators advance in lockstep,
		 * so the inner loop over k is a plain vector add and row j of R is written contiguously. */
		if (R) {
			const int64_t j0 = (int64_t) items * tid / nt, j1 = (int64_t) items * (tid + 1) / nt;
			uint32_t *st = j1 > j0 ? malloc(sizeof(uint32_t) * 31 * (size_t) features) : NULL;
			if (j1 > j0 && !st) abort();
			if (st) {
				mf_rand gk = g0;
				rand_jump(&gk, tab, (uint64_t) users * K + (uint64_t) j0);
				for (int k = 0; k < features; k++) {
					for (int i = 0; i < 31; i++) st[(size_t) i * features + k] = (uint32_t) gk.ring[(gk.f + i) % 31];
					if (k + 1 < features) rand_jump(&gk, tab, (uint64_t) items);   /* same j0, next k */
				}
				int f = 0, b = 28;   /* word 0 is the oldest: replaced next; partner three draws back */
				for (int64_t j = j0; j < j1; j++) {
					uint32_t *wf = st + (size_t) f * features;
					const uint32_t *wb = st + (size_t) b * features;
					double *dst = R + j * features;
					for (int k = 0; k < features; k++) {
						const uint32_t v = wf[k] + wb[k];
						wf[k] = v;
						dst[k] = ((double) (int32_t) (v >> 1) / (double) MF_RAND_MAX) / norm;
					}
					if (++f >= 31) f = 0;
					if (++b >= 31) b = 0;
				}
				free(st);
			}
		}
	}
	free(tab);
}

void mf_host_init_factors(int users, int items, int features, double *L, double *R)
{
	mf_host_init_factors_block(users, items, features, 0, users, L, R);
}

void mf_host_split_entries(const mf_entry *e, int64_t nnz, int32_t *row, int32_t *col, double *val)
{
	for (int64_t n = 0; n < nnz; n++) {
		row[n] = e[n].row;
		col[n] = e[n].col;
		val[n] = e[n].value;
	}
}

/* ---------------------------------------------------------------------------------- partitioning */

int mf_host_partition_users(int users, int parts, int by_entries, const int64_t *row_ptr, int32_t *begin)
{
	if (users < 0 || parts < 1 || !begin || (by_entries && !row_ptr)) return -1;
	if (!by_entries) {
		for (int p = 0; p <= parts; p++) begin[p] = (int32_t) mf_host_block_low(p, parts, users);
		return 0;
	}
	const int64_t total = row_ptr[users];
	begin[0] = 0;
	int u = 0;
	for (int p = 1; p < parts; p++) {
		const int64_t target = total * p / parts;
		/* first row boundary at or after the target, never behind the previous cut */
		while (u < users && row_ptr[u] < target) u++;
		begin[p] = u;
	}
	begin[parts] = users;
	return 0;
}

static int least_prime_factor(int n)
{
	for (int d = 2; (int64_t) d * d <= n; d++)
		if (n % d == 0) return d;
	return n;
}

int mf_host_balanced_grid(int users, int items, int nproc, int32_t size[2])
{
	if (nproc < 1 || users < 0 || items < 0 || !size) return -1;
	/* most square factorisation, larger factor first */
	int small = 1;
	for (int d = 1; (int64_t) d * d <= nproc; d++)
		if (nproc % d == 0) small = d;
	int along = nproc / small, across = small;   /* along = the side that follows the longer matrix dimension */
	const int longer = users > items ? users : items, shorter = users > items ? items : users;
	const int ratio = shorter > 0 ? longer / shorter : 1;
	if (ratio > 1) {
		const int limit = nproc < ratio ? nproc : ratio;
		while (along < limit && across > 1) {
			const int f = least_prime_factor(across);
			if ((int64_t) along * f > limit) break;
			along *= f;
			across /= f;
		}
	}
	size[0] = items > users ? across : along;
	size[1] = items > users ? along : across;
	return 0;
}

int mf_host_write_out(FILE *f, const int32_t *best, int users)
{
	for (int i = 0; i < users; i++)
		if (best[i] >= 0 && fprintf(f, "%d\n", best[i]) < 0) return -1;
	return 0;
}

/* ------------------------------------------------------------------------------------ checkpoint */

int mf_host_checkpoint_write(const char *path, const mf_problem *p, int iters_done, const double *L, const double *R)
{
	char tmp[4096];
	if (snprintf(tmp, sizeof tmp, "%s.tmp", path) >= (int) sizeof tmp) return -1;
	FILE *f = fopen(tmp, "wb");
	if (!f) return -1;
	mf_checkpoint_header h;
	memset(&h, 0, sizeof h);
	memcpy(h.magic, "MFCKPT1", 8);
	h.users = p->users;
	h.items = p->items;
	h.features = p->features;
	h.iters_done = iters_done;
	h.nnz = p->nnz;
	h.alpha = p->alpha;
	const size_t nl = (size_t) p->users * p->features, nr = (size_t) p->items * p->features;
	int ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(L, sizeof(double), nl, f) == nl &&
	         fwrite(R, sizeof(double), nr, f) == nr;
	if (fclose(f) == EOF) ok = 0;
	if (!ok || rename(tmp, path) != 0) {   /* write-then-rename: a crash never leaves a torn checkpoint */
		remove(tmp);
		return -1;
	}
	return 0;
}

int mf_host_checkpoint_read(const char *path, const mf_problem *p, int *iters_done, double *L, double *R)
{
	FILE *f = fopen(path, "rb");
	if (!f) return -1;
	mf_checkpoint_header h;
	const size_t nl = (size_t) p->users * p->features, nr = (size_t) p->items * p->features;
	int ok = fread(&h, sizeof h, 1, f) == 1 && memcmp(h.magic, "MFCKPT1", 8) == 0 && h.users == p->users &&
	         h.items == p->items && h.features == p->features && h.nnz == p->nnz && h.alpha == p->alpha &&
	         h.iters_done >= 0 && h.iters_done <= p->iters;
	ok = ok && fread(L, sizeof(double), nl, f) == nl && fread(R, sizeof(double), nr, f) == nr;
	fclose(f);
	if (!ok) return -1;
	*iters_done = h.iters_done;
	return 0;
}

/* ------------------------------------------------------------------------------------- synthetic */

static inline uint64_t splitmix64(uint64_t *x)
{
	uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

static inline uint64_t row_state(const mf_synth *s, int u)
{
	uint64_t x = s->seed ^ (0xD1B54A32D192ED03ull * (uint64_t) (u + 1));
	(void) splitmix64(&x);
	return x;
}

static inline int32_t raw_count(const mf_synth *s, int u)
{
	uint64_t x = row_state(s, u);
	const uint64_t span = (uint64_t) (s->max_row - s->min_row + 1);
	int64_t m = s->min_row + (int64_t) (splitmix64(&x) % span);
	if (m > s->items) m = s->items;
	if (m < 0) m = 0;
	return (int32_t) m;
}

/* target_nnz > 0: the raw draws m(u) ~ U[min_row, max_row] are rescaled so that they sum to target_nnz EXACTLY
 * (SURVEY 8d): with C(u) the prefix sum of the raw draws and S their total, m'(u) = floor(C(u+1) T / S) - floor(C(u) T / S)
 * -- a pure function of (seed, u) again, within one of m(u) T / S, clamped to the item count (the clamp can only bite
 * when rows are nearly full; the generators below are not meant for that).  Costs one O(users) pass per call. */
int64_t mf_host_synth_counts(const mf_synth *s, int u0, int count, int32_t *counts)
{
	int64_t total = 0;
	if (s->target_nnz <= 0) {
		#pragma omp parallel for schedule(static) reduction(+ : total)
		for (int i = 0; i < count; i++) {
			counts[i] = raw_count(s, u0 + i);
			total += counts[i];
		}
		return total;
	}
	int64_t before = 0, all = 0;
	#pragma omp parallel for schedule(static) reduction(+ : before, all)
	for (int u = 0; u < s->users; u++) {
		const int32_t m = raw_count(s, u);
		all += m;
		if (u < u0) before += m;
	}
	if (all <= 0) {
		memset(counts, 0, sizeof(int32_t) * (size_t) count);
		return 0;
	}
	/* sequential prefix over the block (cheap: one draw per user); products stay below 2^63 up to ~9e18 */
	int64_t c = before;
	for (int i = 0; i < count; i++) {
		const int64_t lo = (int64_t) ((__int128) c * s->target_nnz / all);
		c += raw_count(s, u0 + i);
		const int64_t hi = (int64_t) ((__int128) c * s->target_nnz / all);
		int64_t m = hi - lo;
		if (m > s->items) m = s->items;
		counts[i] = (int32_t) m;
		total += m;
	}
	return total;
}

/* ---- column draws.  MF_SYNTH_STRATIFIED: row u takes one column in each of its m strata of [0, items) (round 1-2's
 * generator: distinct, ascending, uniform marginal, every item ends up with a near-identical count).
 * MF_SYNTH_UNIFORM: m DISTINCT columns drawn uniformly (SURVEY 8d's headline generator).  MF_SYNTH_ZIPF: m distinct
 * columns drawn from Zipf(1.0) item popularity, P(rank r) ~ 1 / (r + 1), ranks scattered over the ids by a fixed
 * bijection -- the most popular items are rated by (nearly) every user: hot columns. */
static int cmp_i32(const void *a, const void *b)
{
	const int32_t x = *(const int32_t *) a, y = *(const int32_t *) b;
	return (x > y) - (x < y);
}

/* Zipf(1.0) CDF over the ranks, built once per item count */
static double *g_zipf_cdf;
static int32_t g_zipf_items;
static pthread_mutex_t g_zipf_mu = PTHREAD_MUTEX_INITIALIZER;

static const double *zipf_cdf(int32_t items)
{
	pthread_mutex_lock(&g_zipf_mu);
	if (g_zipf_items != items) {
		free(g_zipf_cdf);
		g_zipf_cdf = malloc(sizeof(double) * (size_t) (items > 0 ? items : 1));
		g_zipf_items = g_zipf_cdf ? items : 0;
		if (g_zipf_cdf) {
			double acc = 0.0;
			for (int32_t r = 0; r < items; r++) {
				acc += 1.0 / (double) (r + 1);
				g_zipf_cdf[r] = acc;
			}
			for (int32_t r = 0; r < items; r++) g_zipf_cdf[r] /= acc;
		}
	}
	const double *t = g_zipf_cdf;
	pthread_mutex_unlock(&g_zipf_mu);
	return t;
}

static inline int32_t zipf_rank(const double *cdf, int32_t items, uint64_t z)
{
	const double u = (double) (z >> 11) * (1.0 / 9007199254740992.0);   /* [0, 1) */
	int32_t lo = 0, hi = items - 1;
	while (lo < hi) {
		const int32_t mid = lo + (hi - lo) / 2;
		if (cdf[mid] > u)
			hi = mid;
		else
			lo = mid + 1;
	}
	return lo;
}

/* rank -> item id: a bijection of [0, items) that scatters the popular ranks (multiplier coprime with items) */
static inline int32_t scatter(int64_t r, int32_t items, int64_t mult, int64_t shift)
{
	return (int32_t) ((r * mult + shift) % items);
}

static int64_t coprime_multiplier(int32_t items)
{
	int64_t a = (int64_t) (0.6180339887 * (double) items) | 1;
	for (;; a += 2) {
		int64_t x = a, y = items;
		while (y) {
			const int64_t t = x % y;
			x = y;
			y = t;
		}
		if (x == 1) return a;
	}
}

/* m distinct draws into out[0..m), ascending: draw what is missing, sort, drop duplicates, repeat (uniform draws over
 * 1e5 items: a duplicate every ~20 rows; Zipf: a handful of rounds).  More than half of the items: the complement. */
static void draw_distinct(const mf_synth *s, uint64_t *x, int64_t m, int32_t *out, const double *cdf, int64_t mult,
                          int64_t shift)
{
	const int32_t items = s->items;
	if (m <= 0) return;
	if (s->mode == MF_SYNTH_UNIFORM && 2 * m > items) {   /* choose the items LEFT OUT */
		const int64_t k = items - m;
		int32_t *skip = malloc(sizeof(int32_t) * (size_t) (k > 0 ? k : 1));
		if (!skip) abort();
		mf_synth sub = *s;
		draw_distinct(&sub, x, k, skip, cdf, mult, shift);   /* k <= items / 2: no further recursion */
		int64_t o = 0, q = 0;
		for (int32_t j = 0; j < items; j++) {
			if (q < k && skip[q] == j)
				++q;
			else
				out[o++] = j;
		}
		free(skip);
		return;
	}
	int64_t have = 0;
	while (have < m) {
		for (int64_t t = have; t < m; t++) {
			const uint64_t z = splitmix64(x);
			out[t] = s->mode == MF_SYNTH_ZIPF ? scatter(zipf_rank(cdf, items, z), items, mult, shift)
			                                  : (int32_t) ((z >> 8) % (uint64_t) items);
		}
		qsort(out, (size_t) m, sizeof(int32_t), cmp_i32);
		int64_t w = 0;
		for (int64_t t = 0; t < m; t++)
			if (w == 0 || out[t] != out[w - 1]) out[w++] = out[t];
		have = w;
	}
}

int mf_host_synth_fill(const mf_synth *s, int u0, int count, const int32_t *counts, int32_t *row,
                       int32_t *col, double *val)
{
	int64_t *off = malloc(sizeof(int64_t) * ((size_t) count + 1));
	if (!off) return -1;
	off[0] = 0;
	for (int i = 0; i < count; i++) off[i + 1] = off[i] + counts[i];
	const double *cdf = s->mode == MF_SYNTH_ZIPF ? zipf_cdf(s->items) : NULL;
	if (s->mode == MF_SYNTH_ZIPF && !cdf) {
		free(off);
		return -1;
	}
	const int64_t mult = s->items > 1 ? coprime_multiplier(s->items) : 1;
	const int64_t shift = s->items > 0 ? (int64_t) ((s->seed * 0x9E3779B97F4A7C15ull) >> 33) % s->items : 0;
	#pragma omp parallel for schedule(dynamic, 1024)
	for (int i = 0; i < count; i++) {
		const int u = u0 + i;
		uint64_t x = row_state(s, u);
		(void) splitmix64(&x);   /* the draw that fixed the (raw) count */
		const int64_t m = counts[i];
		int64_t o = off[i];
		if (s->mode == MF_SYNTH_STRATIFIED) {
			/* m strata of [0, items), one uniformly drawn column in each; rating uniform in {1,...,5} */
			for (int64_t t = 0; t < m; t++, o++) {
				const int64_t lo = t * s->items / m, hi = (t + 1) * s->items / m;
				const uint64_t z = splitmix64(&x);
				row[o] = u;
				col[o] = (int32_t) (lo + (int64_t) ((z >> 8) % (uint64_t) (hi - lo)));
				val[o] = (double) (1 + (int) (z & 0xff) % 5);
			}
			continue;
		}
		draw_distinct(s, &x, m, col + o, cdf, mult, shift);
		for (int64_t t = 0; t < m; t++, o++) {
			row[o] = u;
			val[o] = (double) (1 + (int) (splitmix64(&x) % 5));
		}
	}
	free(off);
	return 0;
}
