/*
 * matFact.c -- the drop-in command line: `matFact <file.in>` with the reference's argv handling, input
 * grammar, stdout format and error convention (matFact.c:61-137, util.c:7-10), the iteration loop and the
 * recommendation step running on an MI355X through the C ABI of include/matfact_hip.h.
 *
 * stdout carries ONLY the recommendations (one index per line), byte-identical to the reference's `.out`
 * files; timing goes to stderr and only when MATFACT_TIMING is set (the root-dir reference build appends a
 * `time : %f` line to stdout, benchmark.h:23; the hand-in build prints none).
 */
#define _POSIX_C_SOURCE 200809L
#include "../../include/matfact_hip.h"
#include "../../include/matfact_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/*
 * MATFACT_MATS=<path>: also write the reference's debug dump format (samples/inst{0,1,2}.mats): the dense
 * rating matrix, then L, R (printed K x I, i.e. un-transposed) and B = L R^T with "%f " per element -- initially,
 * after each of the first MATFACT_MATS_ITERS iterations (default 0; inst0.mats holds 5) and at the end.
 * Uses the resident-plan API so that every number printed comes from the GPU path.
 */
static void mats_matrix(FILE *f, const char *title, const double *m, int rows, int cols, int transposed)
{
	fprintf(f, "%s\n", title);
	for (int i = 0; i < rows; i++) {
		for (int j = 0; j < cols; j++)
			fprintf(f, "%f ", transposed ? m[(size_t) j * rows + i] : m[(size_t) i * cols + j]);
		fprintf(f, "\n");
	}
}

static int mats_state(FILE *f, mf_plan *plan, const mf_problem *p, double *L, double *R, double *B, int initial)
{
	int rc = mf_plan_download_factors(plan, L, R);
	if (rc == MF_OK) rc = mf_plan_predict(plan, B);
	if (rc != MF_OK) return rc;
	mats_matrix(f, initial ? "Initial matrix L" : "Matrix L", L, p->users, p->features, 0);
	mats_matrix(f, initial ? "Initial matrix R" : "Matrix R", R, p->features, p->items, 1);
	mats_matrix(f, initial ? "Initial matrix B" : "Matrix B", B, p->users, p->items, 0);
	return MF_OK;
}

static int run_with_mats(const char *path, const mf_problem *p, double *L, double *R, int32_t *best, int device)
{
	/* the dense A and B of the dump exist only for small instances (mf_plan_predict refuses more than 2^26 elements);
	 * indices are checked BEFORE A is filled: the parser does not range-check them, only the device build does */
	const size_t nb = (size_t) p->users * (size_t) p->items;
	if (p->users < 0 || p->items < 0 || nb > ((size_t) 1 << 26)) return MF_ERR_UNSUPPORTED;
	for (int64_t n = 0; n < p->nnz; n++)
		if (p->entries[n].row < 0 || p->entries[n].row >= p->users || p->entries[n].col < 0 ||
		    p->entries[n].col >= p->items)
			return MF_ERR_ARGUMENT;
	int rc = MF_OK;
	FILE *f = fopen(path, "w");
	double *B = calloc(nb ? nb : 1, sizeof(double));
	int32_t *row = malloc(sizeof(int32_t) * (size_t) (p->nnz ? p->nnz : 1));
	int32_t *col = malloc(sizeof(int32_t) * (size_t) (p->nnz ? p->nnz : 1));
	double *val = malloc(sizeof(double) * (size_t) (p->nnz ? p->nnz : 1));
	mf_plan *plan = NULL;
	if (!f) rc = MF_ERR_ARGUMENT;
	if (rc == MF_OK && (!B || !row || !col || !val)) rc = MF_ERR_NO_MEMORY;
	if (rc != MF_OK) goto done;
	mf_host_split_entries(p->entries, p->nnz, row, col, val);
	for (int64_t n = 0; n < p->nnz; n++) B[(size_t) row[n] * p->items + col[n]] = val[n];
	mats_matrix(f, "Initial matrix A", B, p->users, p->items, 0);

	mf_shard s = {p->users, p->items, p->features, 0, p->users, p->nnz, row, col, val, p->alpha, device, 0, {0, 0}, {0, 0}, 0, 0};
	rc = mf_plan_create(&plan, &s);
	if (rc == MF_OK) rc = mf_plan_upload_factors(plan, L, R);
	if (rc == MF_OK) rc = mats_state(f, plan, p, L, R, B, 1);
	int shown = getenv("MATFACT_MATS_ITERS") ? atoi(getenv("MATFACT_MATS_ITERS")) : 0;
	if (shown > p->iters) shown = p->iters;
	for (int it = 0; rc == MF_OK && it < shown; it++) {
		rc = mf_plan_iterate(plan, 1);
		if (rc == MF_OK) {
			fprintf(f, "Iter=%d\n", it);
			rc = mats_state(f, plan, p, L, R, B, 0);
		}
	}
	if (rc == MF_OK) rc = mf_plan_iterate(plan, p->iters - shown);
	if (rc == MF_OK) {
		fprintf(f, "Final:\n");
		rc = mats_state(f, plan, p, L, R, B, 0);
	}
	if (rc == MF_OK) rc = mf_plan_recommend(plan, best);
done:   /* the one way out: everything that was acquired is released, whatever failed */
	mf_plan_destroy(plan);
	free(B);
	free(row);
	free(col);
	free(val);
	if (f && fclose(f) == EOF && rc == MF_OK) rc = MF_ERR_ARGUMENT;
	return rc;
}

/*
 * MATFACT_CHECKPOINT=<path> [MATFACT_CHECKPOINT_EVERY=n, default 1000]: write (L, R, iterations done) every n
 * iterations; MATFACT_RESUME=<path>: start from such a file instead of the random initial factors.  The final
 * factors and recommendations are bit-identical to an uninterrupted run.
 */
static int run_with_checkpoints(const mf_problem *p, double *L, double *R, int32_t *best, int device, int start_iter)
{
	const char *ck = getenv("MATFACT_CHECKPOINT");
	int every = getenv("MATFACT_CHECKPOINT_EVERY") ? atoi(getenv("MATFACT_CHECKPOINT_EVERY")) : 1000;
	if (every < 1) every = 1;
	int32_t *row = malloc(sizeof(int32_t) * (size_t) (p->nnz ? p->nnz : 1));
	int32_t *col = malloc(sizeof(int32_t) * (size_t) (p->nnz ? p->nnz : 1));
	double *val = malloc(sizeof(double) * (size_t) (p->nnz ? p->nnz : 1));
	if (!row || !col || !val) return MF_ERR_NO_MEMORY;
	mf_host_split_entries(p->entries, p->nnz, row, col, val);
	mf_shard s = {p->users, p->items, p->features, 0, p->users, p->nnz, row, col, val, p->alpha, device, 0, {0, 0}, {0, 0}, 0, 0};
	mf_plan *plan = NULL;
	int rc = mf_plan_create(&plan, &s);
	if (rc == MF_OK) rc = mf_plan_upload_factors(plan, L, R);
	int done = start_iter;
	while (rc == MF_OK && done < p->iters) {
		int step = p->iters - done;
		if (ck && step > every - done % every) step = every - done % every;
		rc = mf_plan_iterate(plan, step);
		done += step;
		if (rc == MF_OK && ck && done < p->iters) {
			rc = mf_plan_download_factors(plan, L, R);
			if (rc == MF_OK && mf_host_checkpoint_write(ck, p, done, L, R) != 0) rc = MF_ERR_ARGUMENT;
		}
	}
	if (rc == MF_OK) rc = mf_plan_recommend(plan, best);
	if (rc == MF_OK) rc = mf_plan_download_factors(plan, L, R);
	mf_plan_destroy(plan);
	free(row);
	free(col);
	free(val);
	return rc;
}

/* util.c:7-10 */
static void die(const char *error)
{
	fprintf(stderr, "Error: %s\n", error);
	exit(-1);
}

static double now(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec;
}

int main(int argc, char **argv)
{
	if (argc != 2) {
		fprintf(stderr, "Run ./matFact.out file");   /* matFact.c:65 */
		die("Missing input file name.");
	}
	const double t0 = now();

	mf_problem prob;
	/* MATFACT_CACHE=<dir>: binary cache of parsed inputs keyed by the file's content (util.c:30-34 re-parses every
	 * run); unset: the plain parser */
	int cache_hit = 0;
	const int prc = mf_host_parse_file_cached(argv[1], getenv("MATFACT_CACHE"), &prob, &cache_hit);
	if (prc != MF_PARSE_OK) die(mf_host_parse_strerror(prc));
	const double t1 = now();

	const size_t nl = (size_t) prob.users * (size_t) prob.features;
	const size_t nr = (size_t) prob.items * (size_t) prob.features;
	double *L = malloc(sizeof(double) * (nl ? nl : 1));
	double *R = malloc(sizeof(double) * (nr ? nr : 1));
	int32_t *best = malloc(sizeof(int32_t) * (size_t) (prob.users > 0 ? prob.users : 1));
	if (!L || !R || !best) die("Out of memory.");
	mf_host_init_factors(prob.users, prob.items, prob.features, L, R);
	const double t2 = now();

	int device = 0;
	if (getenv("MATFACT_DEVICE")) device = atoi(getenv("MATFACT_DEVICE"));
	const char *mats = getenv("MATFACT_MATS");
	const char *devlist = getenv("MATFACT_DEVICES");   /* e.g. "0,1,2,3,4,5,6,7": row-shard over these GPUs */
	int rc, start_iter = 0;
	if (getenv("MATFACT_RESUME")) {
		if (mf_host_checkpoint_read(getenv("MATFACT_RESUME"), &prob, &start_iter, L, R) != 0)
			die("MATFACT_RESUME: cannot read the checkpoint or it belongs to another instance.");
	}
	if (getenv("MATFACT_CHECKPOINT") || getenv("MATFACT_RESUME")) {
		rc = run_with_checkpoints(&prob, L, R, best, device, start_iter);
	} else if (mats) {
		rc = run_with_mats(mats, &prob, L, R, best, device);
	} else if (devlist) {
		int devs[16], nd = 0;
		for (const char *c = devlist; *c && nd < 16;) {
			char *stop;
			devs[nd++] = (int) strtol(c, &stop, 10);
			if (stop == c) die("MATFACT_DEVICES: expected a comma-separated list of device ordinals.");
			c = *stop == ',' ? stop + 1 : stop;
		}
		rc = mf_backend_run_multi(&prob, L, R, best, devs, nd);
	} else {
		rc = mf_backend_run_top1(&prob, L, R, best, device);   /* only the list is printed: no copy-back of L and R */
	}
	if (rc != MF_OK) {
		fprintf(stderr, "matFact (hip backend): %s %s\n", mf_backend_strerror(rc),
		        rc == MF_ERR_HIP ? mf_backend_last_hip_error() : "");
		die("GPU backend failed.");
	}
	const double t3 = now();

	mf_host_write_out(stdout, best, prob.users);
	fflush(stdout);

	if (getenv("MATFACT_TIMING"))
		fprintf(stderr, "parse%s %.6f init %.6f gpu(run) %.6f total %.6f\n", cache_hit ? "(cache)" : "", t1 - t0, t2 - t1,
		        t3 - t2, now() - t0);
	free(best);
	free(L);
	free(R);
	mf_host_free_problem(&prob);
	return 0;
}
