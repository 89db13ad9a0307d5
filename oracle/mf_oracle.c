/*
 * mf_oracle.c -- CPU ORACLE for the matrix-factorisation hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and the cpu_baseline leg of bench.py may load it.  The product
 * path (recommender-system_amd/) never links, imports or calls anything in oracle/.
 *
 * It restates, in plain C over SoA arrays, the arithmetic of the reference
 * (vladstojna/recommender-system) so that the HIP path can be compared bit for bit:
 *
 *   orc_init_factors   <- mat2d_random_fill_LR (mat2d.c:61-72, RAND01 mat2d.h:4) followed by
 *                         mat2d_transpose (mat2d.c:115-124; call site matFact.c:113-120)
 *   orc_factorize      <- matrix_factorization iteration loop (matFact.c:36-54) with
 *                         mat2d_dot_product (mat2d.c:126-139) and mat2d_copy (mat2d.c:34-36)
 *   orc_recommend      <- mat2d_prod (mat2d.c:100-113) + print_output (matFact.c:10-27),
 *                         without materialising B (as matFact-mpi.c:82-96 does)
 *   orc_predict_row    <- one row of mat2d_prod, for spot checks at sizes where B is too big
 *   orc_factorize_omp  <- the REDUCTION=1 OpenMP algorithm of matFact-omp.c:35-144
 *                         (CPU baseline, `cpu_baseline.kind = "port"`)
 *   orc_shard_step     <- one iteration of the MPI block update (matFact-mpi.c:185-210) for a
 *                         row shard: aux = (root ? old : 0) + local sums; the caller SUM-reduces.
 *   orc_tile_step      <- the same for one tile (user block x item block) of the 2-D process grid:
 *                         both factors start from old on their communicator's root only (:187-188).
 *
 * Parity pin: checked against the reference itself (oracle/_ref, built from the reference's own
 * sources by oracle/Makefile) on every bundled sample and against the bundled samples' .out files
 * and inst{0,1,2}.mats -- see tests/test_oracle_pinned.py and tests/golden/make_golden.py.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC   (never -ffast-math / -march=native
 * without -ffp-contract=off: a fused multiply-add changes the bits).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- init: srandom(0); L row-major draws, then R_init (K x I) row-major draws; R = R_init^T ---- */
void orc_init_factors(int users, int items, int feats, double *L, double *R)
{
	srandom(0);
	const double norm = (double) feats;
	for (int64_t t = 0; t < (int64_t) users * feats; t++)
		L[t] = ((double) random() / (double) RAND_MAX) / norm;
	/* the reference draws R_init[k][j] for k outer, j inner and then transposes */
	for (int k = 0; k < feats; k++)
		for (int j = 0; j < items; j++)
			R[(int64_t) j * feats + k] = ((double) random() / (double) RAND_MAX) / norm;
}

/* sequential-k dot product, accumulator starts at 0.0, separate multiply and add */
static inline double seq_dot(const double *a, const double *b, int feats)
{
	double s = 0;
	for (int k = 0; k < feats; k++)
		s += a[k] * b[k];
	return s;
}

/* ---- the serial iteration loop (file order over the entries; frozen copies per iteration) ---- */
void orc_factorize(int users, int items, int feats, int64_t nnz,
                   const int32_t *row, const int32_t *col, const double *val,
                   int iters, double alpha, double *L, double *R)
{
	const int64_t nl = (int64_t) users * feats, nr = (int64_t) items * feats;
	double *Ls = malloc(sizeof(double) * (nl ? nl : 1));
	double *Rs = malloc(sizeof(double) * (nr ? nr : 1));
	for (int it = 0; it < iters; it++) {
		memcpy(Ls, L, sizeof(double) * nl);
		memcpy(Rs, R, sizeof(double) * nr);
		for (int64_t n = 0; n < nnz; n++) {
			const double *ls = Ls + (int64_t) row[n] * feats;
			const double *rs = Rs + (int64_t) col[n] * feats;
			double *l = L + (int64_t) row[n] * feats;
			double *r = R + (int64_t) col[n] * feats;
			/* alpha * 2 * (a - dot): C evaluates (alpha*2) first */
			const double e = alpha * 2 * (val[n] - seq_dot(ls, rs, feats));
			for (int k = 0; k < feats; k++) {
				l[k] = l[k] - e * (-rs[k]);
				r[k] = r[k] - e * (-ls[k]);
			}
		}
	}
	free(Ls);
	free(Rs);
}

/* one predicted row: out[j] = sum_k L[i][k] * R[j][k], sequential k from 0.0 */
void orc_predict_row(int items, int feats, const double *Li, const double *R, double *out)
{
	for (int j = 0; j < items; j++) {
		double b = 0;
		for (int k = 0; k < feats; k++)
			b = b + Li[k] * R[(int64_t) j * feats + k];
		out[j] = b;
	}
}

/*
 * Masked row arg-max with the reference's cursor semantics: entries are (row, col)-sorted, a cursor
 * walks them; a rated (i, j) is skipped; the first unrated j seeds the maximum and a later j wins only
 * on strict '>'.  best[i] = -1 when every item of user i is rated (the reference prints no line).
 * The reference reads entries[nnz] after the last entry (UB); the oracle treats that as "no match".
 */
void orc_recommend(int users, int items, int feats, int64_t nnz,
                   const int32_t *row, const int32_t *col,
                   const double *L, const double *R, int32_t *best)
{
	double *b = malloc(sizeof(double) * (items ? items : 1));
	int64_t aix = 0;
	for (int i = 0; i < users; i++) {
		orc_predict_row(items, feats, L + (int64_t) i * feats, R, b);
		int max = -1;
		for (int j = 0; j < items; j++) {
			if (!(aix < nnz && row[aix] == i && col[aix] == j)) {
				if (max == -1 || b[j] > b[max])
					max = j;
			} else {
				aix++;
			}
		}
		best[i] = max;
	}
	free(b);
}

/*
 * One iteration of the block update for a contiguous user shard [u0, u0+users_loc): the MPI variant
 * computes into aux buffers that start from the old factor on the communicator root and from zero
 * elsewhere; summing the aux buffers over the communicator gives the new factor.  L is private to the
 * shard (row_comm of size 1 in the 8x1 grid), so L_new is complete; R_aux must be SUM-reduced by the caller.
 * row[] holds GLOBAL user ids.
 */
void orc_shard_step(int u0, int users_loc, int items, int feats, int64_t nnz_loc,
                    const int32_t *row, const int32_t *col, const double *val, double alpha,
                    const double *L_old, const double *R_old, int r_is_root,
                    double *L_new, double *R_aux)
{
	memcpy(L_new, L_old, sizeof(double) * (int64_t) users_loc * feats);
	if (r_is_root)
		memcpy(R_aux, R_old, sizeof(double) * (int64_t) items * feats);
	else
		memset(R_aux, 0, sizeof(double) * (int64_t) items * feats);
	for (int64_t n = 0; n < nnz_loc; n++) {
		const int64_t i = row[n] - u0;
		const double *ls = L_old + i * feats;
		const double *rs = R_old + (int64_t) col[n] * feats;
		double *l = L_new + i * feats;
		double *r = R_aux + (int64_t) col[n] * feats;
		const double e = alpha * 2 * (val[n] - seq_dot(ls, rs, feats));
		for (int k = 0; k < feats; k++) {
			l[k] = l[k] - e * (-rs[k]);
			r[k] = r[k] - e * (-ls[k]);
		}
	}
}

/*
 * One iteration of the block update for a TILE of the 2-D grid (matFact-mpi.c:185-205): users
 * [u0, u0+users_loc) x items [j0, j0+items_loc), row[]/col[] GLOBAL ids of the entries inside the tile.
 * L_aux starts from L_old on the row communicator's root and R_aux from R_old on the column communicator's
 * root, from zero elsewhere (:187-188); the caller SUM-reduces L_aux over the grid row and R_aux over the
 * grid column (:207-208).
 */
void orc_tile_step(int u0, int users_loc, int j0, int items_loc, int feats, int64_t nnz_loc,
                   const int32_t *row, const int32_t *col, const double *val, double alpha,
                   const double *L_old, const double *R_old, int l_is_root, int r_is_root,
                   double *L_aux, double *R_aux)
{
	const size_t nl = sizeof(double) * (size_t) users_loc * feats, nr = sizeof(double) * (size_t) items_loc * feats;
	if (l_is_root) memcpy(L_aux, L_old, nl); else memset(L_aux, 0, nl);
	if (r_is_root) memcpy(R_aux, R_old, nr); else memset(R_aux, 0, nr);
	for (int64_t n = 0; n < nnz_loc; n++) {
		const int64_t i = row[n] - u0, j = col[n] - j0;
		const double *ls = L_old + i * feats, *rs = R_old + j * feats;
		double *l = L_aux + i * feats, *r = R_aux + j * feats;
		const double e = alpha * 2 * (val[n] - seq_dot(ls, rs, feats));
		for (int k = 0; k < feats; k++) {
			l[k] = l[k] - e * (-rs[k]);
			r[k] = r[k] - e * (-ls[k]);
		}
	}
}

/*
 * OpenMP CPU baseline: the REDUCTION=1 algorithm of matFact-omp.c -- static split of the entries over
 * threads, atomic updates on the side with MORE rows, a private per-thread partial of the other side that
 * is summed into the factor after the loop; when items > users the entries are processed in column order.
 * Returns the seconds spent in the iteration loop only (the reference times the whole program).
 * `order` = 0: entries as given; the caller passes column-sorted arrays when items > users.
 */
double orc_factorize_omp(int users, int items, int feats, int64_t nnz,
                         const int32_t *row, const int32_t *col, const double *val,
                         int iters, double alpha, double *L, double *R, int *threads_used)
{
	const int64_t nl = (int64_t) users * feats, nr = (int64_t) items * feats;
	const int reduce_L = items > users;         /* private partial holds the SMALLER side */
	const int64_t np = reduce_L ? nl : nr;
	double *Ls = malloc(sizeof(double) * (nl ? nl : 1));
	double *Rs = malloc(sizeof(double) * (nr ? nr : 1));
	double **partial = NULL;
	int nthreads = 1;
	struct timespec t0, t1;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	#pragma omp parallel
	{
		#pragma omp single
		{
#ifdef _OPENMP
			nthreads = omp_get_num_threads();
#endif
			partial = malloc(sizeof(double *) * nthreads);
		}
		int tid = 0;
#ifdef _OPENMP
		tid = omp_get_thread_num();
#endif
		partial[tid] = malloc(sizeof(double) * (np ? np : 1));
		double *mine = partial[tid];
		for (int it = 0; it < iters; it++) {
			#pragma omp for schedule(static)
			for (int64_t t = 0; t < nl; t++) Ls[t] = L[t];
			#pragma omp for schedule(static)
			for (int64_t t = 0; t < nr; t++) Rs[t] = R[t];
			memset(mine, 0, sizeof(double) * np);
			#pragma omp for schedule(static)
			for (int64_t n = 0; n < nnz; n++) {
				const int64_t i = row[n], j = col[n];
				const double *ls = Ls + i * feats, *rs = Rs + j * feats;
				const double e = alpha * 2 * (val[n] - seq_dot(ls, rs, feats));
				if (reduce_L) {
					double *r = R + j * feats, *p = mine + i * feats;
					for (int k = 0; k < feats; k++) {
						const double d = e * (-ls[k]);
						#pragma omp atomic
						r[k] -= d;
						p[k] = p[k] - e * (-rs[k]);
					}
				} else {
					double *l = L + i * feats, *p = mine + j * feats;
					for (int k = 0; k < feats; k++) {
						const double d = e * (-rs[k]);
						#pragma omp atomic
						l[k] -= d;
						p[k] = p[k] - e * (-ls[k]);
					}
				}
			}
			/* sum the private partials into the reduced side, thread after thread, rows split */
			double *target = reduce_L ? L : R;
			for (int t = 0; t < nthreads; t++) {
				const double *src = partial[t];
				#pragma omp for schedule(static) nowait
				for (int64_t x = 0; x < np; x++) target[x] = target[x] + src[x];
			}
			#pragma omp barrier
		}
		free(mine);
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	free(partial);
	free(Ls);
	free(Rs);
	if (threads_used) *threads_used = nthreads;
	return (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
}

/* thread count of the OpenMP legs (libgomp may have read OMP_NUM_THREADS long before we are loaded) */
void orc_set_threads(int n)
{
#ifdef _OPENMP
	if (n > 0) omp_set_num_threads(n);
#else
	(void) n;
#endif
}

/* serial loop with the same timing convention, for a 1-core baseline figure */
double orc_factorize_timed(int users, int items, int feats, int64_t nnz,
                           const int32_t *row, const int32_t *col, const double *val,
                           int iters, double alpha, double *L, double *R)
{
	struct timespec t0, t1;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	orc_factorize(users, items, feats, nnz, row, col, val, iters, alpha, L, R);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	return (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
}
