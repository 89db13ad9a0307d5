/*
 * matfact_host.h -- host-side C helpers around the HIP backend: the pieces of the reference's `main`
 * (matFact.c:61-137) that stay on the CPU.  Plain C, no GPU dependency; libmatfact_host.so.
 *
 *   mf_host_parse_file_cached  the same behind a binary cache keyed by the file's content
 *   mf_host_parse_file      the `.in` reader          matFact.c:72-110, util.c:12-34 (same grammar, same
 *                                                     error strings, returned instead of die()'d)
 *   mf_host_init_factors    initial L and R           mat2d.c:61-72 + mat2d.c:115-124 (matFact.c:113-120)
 *                           with an own restatement of glibc's TYPE_3 random() so the result does not
 *                           depend on the C library in use
 *   mf_host_block_*         BLOCK_LOW/HIGH/SIZE/OWNER mpiutil.h:8-13
 *   mf_host_partition_users contiguous user blocks for P shards, by row count (the reference's rule)
 *                           or balanced by entry count
 *   mf_host_balanced_grid   rows x cols of the 2-D process grid       mpiutil.c:54-88 (create_balanced_grid)
 *   mf_host_write_out       the `.out` writer         matFact.c:24-25
 *   mf_host_synth_*         deterministic synthetic instances (the reference ships no generator;
 *                           SURVEY.md section 8d defines the shapes)
 */
#ifndef MATFACT_HOST_H
#define MATFACT_HOST_H

#include <stdint.h>
#include <stdio.h>
#include "matfact_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* parse errors: the message is exactly what the reference passes to die() (util.c:12-34, matFact.c:75-109) */
typedef enum mf_parse_status {
	MF_PARSE_OK = 0,
	MF_PARSE_OPEN = 1,        /* "Unable to open input file."      */
	MF_PARSE_INT = 2,         /* "Error in int argument."          */
	MF_PARSE_DOUBLE = 3,      /* "Error in double argument."       */
	MF_PARSE_THREE_INTS = 4,  /* "Error in multiple int argument." */
	MF_PARSE_ENTRY = 5,       /* "Error in non-zero entry."        */
	MF_PARSE_CLOSE = 6,       /* "Unable to close input file."     */
	MF_PARSE_NOMEM = 7        /* allocation failed (the reference would crash) */
} mf_parse_status;

const char *mf_host_parse_strerror(int status);

/* Fills *p; p->entries is malloc'ed and must be released with mf_host_free_problem. */
int mf_host_parse_file(const char *path, mf_problem *p);
int mf_host_parse_buffer(const char *text, size_t len, mf_problem *p);
void mf_host_free_problem(mf_problem *p);
/* The same through a binary cache (the reference re-parses with fscanf on every run, util.c:30-34):
 * cache_dir/<content hash>-<size>.mfcache holds the header and the entries as parsed, keyed by the CONTENT of the
 * `.in` (a 64-bit hash + the size: an edited file misses).  Miss: parse, then write the cache (best effort,
 * write-then-rename).  Hit: the cache file is mapped, the hash of its entries checked against its header, and p->entries
 * points into the mapping -- no parse, no copy; *cache_hit = 1.  Thread-safe; any number of cached problems may be open.
 * cache_dir NULL or "": plain mf_host_parse_file.  Release with mf_host_free_problem either way. */
int mf_host_parse_file_cached(const char *path, const char *cache_dir, mf_problem *p, int *cache_hit);

/* glibc-compatible random(): srandom(seed) / random(), TYPE_3 additive feedback, r[i] = r[i-3] + r[i-31] */
typedef struct mf_rand {
	int32_t ring[31];
	int f, b;   /* front / back cursors into the 31-word ring */
} mf_rand;
void mf_host_srandom(mf_rand *g, unsigned seed);
int32_t mf_host_random(mf_rand *g);

/* L: users x K, R: items x K (transposed storage), both row-major */
void mf_host_init_factors(int users, int items, int features, double *L, double *R);
/* only rows [u0, u0+count) of L are stored (L_block: count x K); R complete.  Same values as above. */
void mf_host_init_factors_block(int users, int items, int features, int u0, int count, double *L_block,
                                double *R);

/* AoS -> SoA */
void mf_host_split_entries(const mf_entry *e, int64_t nnz, int32_t *row, int32_t *col, double *val);

static inline int64_t mf_host_block_low(int id, int p, int64_t n) { return (int64_t) id * n / p; }
static inline int64_t mf_host_block_high(int id, int p, int64_t n) { return mf_host_block_low(id + 1, p, n) - 1; }
static inline int64_t mf_host_block_size(int id, int p, int64_t n)
{
	return mf_host_block_high(id, p, n) - mf_host_block_low(id, p, n) + 1;
}
static inline int mf_host_block_owner(int64_t index, int p, int64_t n) { return (int) ((p * (index + 1) - 1) / n); }

/* begin[0..parts]: user boundaries.  by_entries = 0: BLOCK_LOW rule; 1: cut at row boundaries so that each
 * part holds about nnz/parts entries (row_ptr = CSR row pointer of the whole instance, users+1 long). */
int mf_host_partition_users(int users, int parts, int by_entries, const int64_t *row_ptr, int32_t *begin);

/* The process grid of the MPI variant (create_balanced_grid, mpiutil.c:54-88): size[0] grid rows (user blocks)
 * x size[1] grid columns (item blocks), size[0]*size[1] == nproc.  Starts from the most square factorisation
 * (what MPI_Dims_create gives two free dimensions, larger first) and moves prime factors to the long side while
 * it stays <= min(nproc, long/short aspect ratio of the matrix); the long side of the grid is laid along the
 * long side of the matrix.  cfg4 (1e6 x 1e5) on 8 ranks: 8x1; a square matrix: 4x2. */
int mf_host_balanced_grid(int users, int items, int nproc, int32_t size[2]);

/* one line per user with best >= 0 */
int mf_host_write_out(FILE *f, const int32_t *best, int users);

/* ---- iteration checkpoint (the reference has none): a binary file holding the header of the
 * instance it belongs to, the number of iterations already done, and L and R.  Resuming reproduces the
 * uninterrupted run bit for bit (the iteration is deterministic). */
typedef struct mf_checkpoint_header {
	char magic[8];            /* "MFCKPT1\0" */
	int32_t users, items, features;
	int32_t iters_done;
	int64_t nnz;
	double alpha;
} mf_checkpoint_header;
int mf_host_checkpoint_write(const char *path, const mf_problem *p, int iters_done, const double *L, const double *R);
/* L (users x K) and R (items x K) must be allocated by the caller; fails (-1) if the file does not belong to p */
int mf_host_checkpoint_read(const char *path, const mf_problem *p, int *iters_done, double *L, double *R);

/* ---- synthetic instances (the reference ships no generator): row u has m(u) distinct sorted columns,
 * m(u) ~ U[min_row, max_row], ratings ~ U{1..5}; everything is a pure function of (seed, u), so any rank can generate
 * any block of users.  target_nnz > 0 rescales the counts so that they sum to target_nnz exactly.  Columns:
 *   MF_SYNTH_STRATIFIED  one column per stratum of width items / m(u) (rounds 1-2: every item ends up with a
 *                        near-identical count -- the friendliest column distribution)
 *   MF_SYNTH_UNIFORM     m(u) distinct uniform columns (the survey's headline generator)
 *   MF_SYNTH_ZIPF        m(u) distinct columns from Zipf(1.0) item popularity, ranks scattered over the ids: the most
 *                        popular items are rated by nearly every user (hot columns) */
enum { MF_SYNTH_STRATIFIED = 0, MF_SYNTH_UNIFORM = 1, MF_SYNTH_ZIPF = 2 };
typedef struct mf_synth {
	uint64_t seed;
	int32_t users, items;
	int32_t min_row, max_row;
	int32_t mode;         /* MF_SYNTH_* */
	int32_t reserved;
	int64_t target_nnz;   /* 0: the raw draws */
} mf_synth;
/* entries per user for users [u0, u0+count): counts[count]; returns their sum */
int64_t mf_host_synth_counts(const mf_synth *s, int u0, int count, int32_t *counts);
/* fills row/col/val (sum of counts entries) for users [u0, u0+count), sorted by (row, col) */
int mf_host_synth_fill(const mf_synth *s, int u0, int count, const int32_t *counts, int32_t *row,
                       int32_t *col, double *val);

#ifdef __cplusplus
}
#endif
#endif
