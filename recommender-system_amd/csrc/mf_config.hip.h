// mf_config.hip.h -- every environment switch of the library, read ONCE per plan (mf_plan_create / the level-1 calls /
// mf_backend_run_multi) into a struct the plan keeps.  Nothing reads the environment per launch or per iteration, so a
// plan's behaviour is fixed at its creation and mf_plan_describe prints the switches that differ from their defaults.
//
// Two classes:
//   * documented switches select between forms the library SHIPS -- the tests
//     force each form through them and bench.py --check uses them for its reference run; always honoured;
//   * experiment switches tune constants inside one form (chunk and segment sizes, schedules).  They are compiled in only
//     with -DMF_EXPERIMENTS (make EXPERIMENTS=1; tools/ scripts build their own copy): the shipped library ignores them.
#pragma once
#include <cstdlib>
#include <cstring>
#include <string>

namespace {

struct mf_config {
	// ---- documented
	enum IterMode { kIterAuto, kIterSweeps, kIterEs };
	IterMode iter_mode = kIterAuto;   // MF_ITER_MODE=auto|sweeps|es: two sweeps or errors + resident streams
	bool sweep_reg = false;           // MF_SWEEP_IMPL=reg: register-staged sweep only
	bool skew = true;                 // MF_SWEEP_SKEW=0: no split of long rows
	bool sweep_long_set = false;      // MF_SWEEP_LONG=<entries>: a row at least this long takes the extreme-row path
	double sweep_long = 0.0;
	int sweep_nch = 0;                // MF_SWEEP_NCH=<1..64>: entries per chunk (0: the rule of choose_sweep)
	int sweep_db = -1;                // MF_SWEEP_DB=0|1: intra-wave double-buffered sweep off / forced (-1: by occupancy)
	int sweep_pair = -1;              // MF_SWEEP_PAIR=0|1: wave-pair sweep (loader + compute wave per row) off / forced (-1: by the plan)
	int es_sw = 0;                    // MF_ES_SW=8|4|2: slice width of the resident streams launch to try first
	bool row_pitch = true;            // MF_ROW_PITCH=0: dense device rows
	bool resident = true;             // MF_RESIDENT=0: no single-launch loop for toy instances
	bool graph = true;                // MF_GRAPH=0: no HIP-graph replay of small iterations
	double graph_max = 2e6;           // MF_GRAPH_MAX: nnz*K below which iterations are replayed from a graph
	bool rec_exact = false;           // MF_RECOMMEND_IMPL=exact: the exact recommendation kernel only
	bool rec_ares = true;             // MF_RECOMMEND_ARES=0: no LDS-resident L image in the MFMA pass
	bool rec_bdma = true;             // MF_RECOMMEND_BDMA=0: R chunks staged through registers
	int rec_half = 1;                 // MF_RECOMMEND_HALF=0: no 64-user workgroups (two per CU) in the MFMA pass; all: also for
	                                  // K that is no multiple of 20 (the general form of that kernel: slower than the 128-user one)
	int rec_split = -1;               // MF_RECOMMEND_SPLIT=0|n: item split of small recommendations off / n splits (-1: rule)
	bool build_host = false;          // MF_BUILD=host: CSR/CSC bucketed on the host
	bool os_dpp = true;               // MF_OS_DPP=0: ordered sums by plain v_add_f64 (no DPP broadcast)
	bool multi_force = false;         // MF_MULTI_FORCE=1: sharded path even with one shard
	bool multi_rccl = false;          // MF_MULTI_REDUCE=rccl|peer
	bool multi_threads = true;        // MF_MULTI_THREADS=0: all shards enqueued from the calling thread
	// ---- experiments (-DMF_EXPERIMENTS)
	int sweep_few = 2048;             // MF_SWEEP_FEW: row count below which a sweep takes the large chunk
	int sweep_seg = 64;               // MF_SWEEP_SEG: entries per segment of the products launch
	int sweep_pnch = 0;               // MF_SWEEP_PNCH: chunk size of the products launch (0: the sweep's)
	size_t os_lds = 0;                // MF_OS_LDS: LDS request of the ordered sums (bounds their waves per CU)
	bool sum_under = false;           // MF_SWEEP_SUM_ORDER=under
	bool no_defer = false;            // MF_SWEEP_NO_DEFER
	bool nosort = false;              // MF_SWEEP_NOSORT
	bool rest_coop = false;           // MF_SWEEP_REST=coop
	int sweep_trio = 0;               // MF_SWEEP_TRIO=1 / MF_SWEEP_TRIO_U=1: the pair form's compute wave split into a phase-A and a phase-B wave wherever pairs run / on the user side
	int sweep_pair_kind[2] = {-1, -1};  // MF_SWEEP_PAIR_I / MF_SWEEP_PAIR_U = 0|1: the wave-pair form of the item / user sweep alone
	int es_row_cost = 0;              // MF_ES_ROW_COST: entries a row end counts for when the streams launch cuts its runs (0: rule)
	int es_active = 0;                // MF_ES_ACTIVE=1..8: waves per workgroup of the streams launch that own rows (0: rule)
	int es_nch = 0;                   // MF_ES_NCH: segment size of the errors launch
	int db_rows = 0;                  // MF_SWEEP_DB_ROWS: row count below which the double-buffered sweep is chosen (0: rule)
	int sweep_long_kind[2] = {0, 0};  // MF_SWEEP_LONG_I / MF_SWEEP_LONG_U: the extreme-row threshold of the item / user sweep alone
	int side_prio = -1;               // MF_SIDE_PRIO=0|1: the side stream of the extreme-row path at low / high priority (-1: rule)
	bool rec_wide = false;            // MF_RECOMMEND_WIDE: K=128 in the eight-wave, 16-users-per-wave shape of K=256
	int pf_rows = 0;                  // MF_SWEEP_PF_ROWS: launches of up to this many rows take the pipelined-phases form (0: 262144)
	int pair_loaders = 0;             // MF_SWEEP_PAIR_LOADERS=1|2: loader waves of the wave-pair form (0: rule)
	int pair_nch = 0;                 // MF_SWEEP_PAIR_NCH: chunk size of the wave-pair form (0: 32)
	int db_nch = 0;                   // MF_SWEEP_DB_NCH: its chunk size (0: 16)
	bool sweep_pf = true;             // MF_SWEEP_PF=0: phases A / B as hipcc schedules them (two steps per LDS round trip)
	int sweep_prio = -1;              // MF_SWEEP_PRIO: rows at least this long run at raised wave priority (0: none, -1: rule)
	int sweep_mid = 0;                // MF_SWEEP_MID: rows at least this long (and below the extreme threshold) get their own launch (0: none, -1: a quarter of the threshold)
	bool mid_coop = false;            // MF_SWEEP_MID_KERNEL=coop: the mid-length rows through the row-cooperative kernel
	int mid_nch = 0;                  // MF_SWEEP_MID_NCH: its chunk size (0: 32 or what fits a third of the LDS)

	static bool is0(const char *v) { return v && v[0] == '0'; }
	static bool eq(const char *v, const char *s) { return v && strcmp(v, s) == 0; }

	static mf_config from_env()
	{
		mf_config c;
		const char *v;
		if ((v = getenv("MF_ITER_MODE"))) c.iter_mode = eq(v, "sweeps") ? kIterSweeps : eq(v, "es") ? kIterEs : kIterAuto;
		c.sweep_reg = eq(getenv("MF_SWEEP_IMPL"), "reg");
		c.skew = !is0(getenv("MF_SWEEP_SKEW"));
		if ((v = getenv("MF_SWEEP_LONG"))) {
			c.sweep_long_set = true;
			c.sweep_long = atof(v);
		}
		if ((v = getenv("MF_SWEEP_NCH"))) {
			const int n = atoi(v);
			if (n >= 1 && n <= 64) c.sweep_nch = n;
		}
		if ((v = getenv("MF_SWEEP_DB"))) c.sweep_db = is0(v) ? 0 : 1;
		if ((v = getenv("MF_SWEEP_PAIR"))) c.sweep_pair = is0(v) ? 0 : 1;
		if ((v = getenv("MF_ES_SW"))) c.es_sw = atoi(v);
		c.row_pitch = !is0(getenv("MF_ROW_PITCH"));
		c.resident = !is0(getenv("MF_RESIDENT"));
		c.graph = !is0(getenv("MF_GRAPH"));
		if ((v = getenv("MF_GRAPH_MAX"))) c.graph_max = atof(v);
		c.rec_exact = eq(getenv("MF_RECOMMEND_IMPL"), "exact");
		c.rec_ares = !is0(getenv("MF_RECOMMEND_ARES"));
		c.rec_bdma = !is0(getenv("MF_RECOMMEND_BDMA"));
		if ((v = getenv("MF_RECOMMEND_HALF"))) c.rec_half = is0(v) ? 0 : eq(v, "all") ? 2 : 1;
		if ((v = getenv("MF_RECOMMEND_SPLIT"))) c.rec_split = atoi(v);
		c.build_host = eq(getenv("MF_BUILD"), "host");
		c.os_dpp = !is0(getenv("MF_OS_DPP"));
		c.multi_force = eq(getenv("MF_MULTI_FORCE"), "1");
		c.multi_rccl = eq(getenv("MF_MULTI_REDUCE"), "rccl");
		c.multi_threads = !is0(getenv("MF_MULTI_THREADS"));
#ifdef MF_EXPERIMENTS
		if ((v = getenv("MF_SWEEP_FEW"))) c.sweep_few = atoi(v);
		if ((v = getenv("MF_SWEEP_SEG"))) c.sweep_seg = atoi(v) > 16 ? atoi(v) : 16;
		if ((v = getenv("MF_SWEEP_PNCH"))) c.sweep_pnch = atoi(v);
		if ((v = getenv("MF_OS_LDS"))) c.os_lds = (size_t) atoll(v);
		c.sum_under = eq(getenv("MF_SWEEP_SUM_ORDER"), "under");
		c.no_defer = getenv("MF_SWEEP_NO_DEFER") != nullptr;
		c.nosort = getenv("MF_SWEEP_NOSORT") != nullptr;
		c.rest_coop = eq(getenv("MF_SWEEP_REST"), "coop");
		if ((v = getenv("MF_ES_NCH"))) c.es_nch = atoi(v);
		if ((v = getenv("MF_ES_ACTIVE"))) c.es_active = atoi(v);
		if ((v = getenv("MF_ES_ROW_COST"))) c.es_row_cost = atoi(v);
		if ((v = getenv("MF_SWEEP_TRIO"))) c.sweep_trio = is0(v) ? 0 : 1;
		if ((v = getenv("MF_SWEEP_TRIO_U")) && !is0(v)) c.sweep_trio = 2;
		if ((v = getenv("MF_SWEEP_PAIR_I"))) c.sweep_pair_kind[0] = is0(v) ? 0 : 1;
		if ((v = getenv("MF_SWEEP_PAIR_U"))) c.sweep_pair_kind[1] = is0(v) ? 0 : 1;
		if ((v = getenv("MF_SWEEP_DB_ROWS"))) c.db_rows = atoi(v);
		if ((v = getenv("MF_SWEEP_DB_NCH"))) c.db_nch = atoi(v);
		if ((v = getenv("MF_SWEEP_PAIR_NCH"))) c.pair_nch = atoi(v);
		if ((v = getenv("MF_SWEEP_PAIR_LOADERS"))) c.pair_loaders = atoi(v);
		if ((v = getenv("MF_SWEEP_PF_ROWS"))) c.pf_rows = atoi(v);
		c.rec_wide = getenv("MF_RECOMMEND_WIDE") != nullptr;
		if ((v = getenv("MF_SWEEP_LONG_I"))) c.sweep_long_kind[0] = atoi(v);
		if ((v = getenv("MF_SWEEP_LONG_U"))) c.sweep_long_kind[1] = atoi(v);
		if ((v = getenv("MF_SIDE_PRIO"))) c.side_prio = is0(v) ? 0 : 1;
		if ((v = getenv("MF_SWEEP_PF"))) c.sweep_pf = !is0(v);
		if ((v = getenv("MF_SWEEP_PRIO"))) c.sweep_prio = atoi(v);
		if ((v = getenv("MF_SWEEP_MID"))) c.sweep_mid = atoi(v);
		if ((v = getenv("MF_SWEEP_MID_NCH"))) c.mid_nch = atoi(v);
		c.mid_coop = eq(getenv("MF_SWEEP_MID_KERNEL"), "coop");
#endif
		return c;
	}

	// the switches that differ from their defaults, for mf_plan_describe ("" when none does)
	std::string describe() const
	{
		const mf_config d;
		std::string s;
		auto add = [&](const char *name, const std::string &val) { s += std::string(s.empty() ? "" : ",") + name + "=" + val; };
		if (iter_mode != d.iter_mode) add("MF_ITER_MODE", iter_mode == kIterSweeps ? "sweeps" : "es");
		if (sweep_reg) add("MF_SWEEP_IMPL", "reg");
		if (!skew) add("MF_SWEEP_SKEW", "0");
		if (sweep_long_set) add("MF_SWEEP_LONG", std::to_string(sweep_long));
		if (sweep_nch) add("MF_SWEEP_NCH", std::to_string(sweep_nch));
		if (sweep_db >= 0) add("MF_SWEEP_DB", std::to_string(sweep_db));
		if (sweep_pair >= 0) add("MF_SWEEP_PAIR", std::to_string(sweep_pair));
		if (es_sw) add("MF_ES_SW", std::to_string(es_sw));
		if (!row_pitch) add("MF_ROW_PITCH", "0");
		if (!resident) add("MF_RESIDENT", "0");
		if (!graph) add("MF_GRAPH", "0");
		if (graph_max != d.graph_max) add("MF_GRAPH_MAX", std::to_string(graph_max));
		if (rec_exact) add("MF_RECOMMEND_IMPL", "exact");
		if (!rec_ares) add("MF_RECOMMEND_ARES", "0");
		if (!rec_bdma) add("MF_RECOMMEND_BDMA", "0");
		if (rec_half != 1) add("MF_RECOMMEND_HALF", rec_half ? "all" : "0");
		if (rec_split >= 0) add("MF_RECOMMEND_SPLIT", std::to_string(rec_split));
		if (build_host) add("MF_BUILD", "host");
		if (!os_dpp) add("MF_OS_DPP", "0");
		if (multi_force) add("MF_MULTI_FORCE", "1");
		if (multi_rccl) add("MF_MULTI_REDUCE", "rccl");
		if (!multi_threads) add("MF_MULTI_THREADS", "0");
		if (sweep_few != d.sweep_few) add("MF_SWEEP_FEW", std::to_string(sweep_few));
		if (sweep_seg != d.sweep_seg) add("MF_SWEEP_SEG", std::to_string(sweep_seg));
		if (sweep_pnch) add("MF_SWEEP_PNCH", std::to_string(sweep_pnch));
		if (os_lds) add("MF_OS_LDS", std::to_string(os_lds));
		if (sum_under) add("MF_SWEEP_SUM_ORDER", "under");
		if (no_defer) add("MF_SWEEP_NO_DEFER", "1");
		if (nosort) add("MF_SWEEP_NOSORT", "1");
		if (rest_coop) add("MF_SWEEP_REST", "coop");
		if (es_nch) add("MF_ES_NCH", std::to_string(es_nch));
		if (es_active) add("MF_ES_ACTIVE", std::to_string(es_active));
		if (es_row_cost) add("MF_ES_ROW_COST", std::to_string(es_row_cost));
		if (sweep_trio) add(sweep_trio == 2 ? "MF_SWEEP_TRIO_U" : "MF_SWEEP_TRIO", "1");
		if (sweep_pair_kind[0] >= 0) add("MF_SWEEP_PAIR_I", std::to_string(sweep_pair_kind[0]));
		if (sweep_pair_kind[1] >= 0) add("MF_SWEEP_PAIR_U", std::to_string(sweep_pair_kind[1]));
		if (db_rows) add("MF_SWEEP_DB_ROWS", std::to_string(db_rows));
		if (db_nch) add("MF_SWEEP_DB_NCH", std::to_string(db_nch));
		if (pair_nch) add("MF_SWEEP_PAIR_NCH", std::to_string(pair_nch));
		if (pair_loaders) add("MF_SWEEP_PAIR_LOADERS", std::to_string(pair_loaders));
		if (pf_rows) add("MF_SWEEP_PF_ROWS", std::to_string(pf_rows));
		if (rec_wide) add("MF_RECOMMEND_WIDE", "1");
		if (sweep_long_kind[0]) add("MF_SWEEP_LONG_I", std::to_string(sweep_long_kind[0]));
		if (sweep_long_kind[1]) add("MF_SWEEP_LONG_U", std::to_string(sweep_long_kind[1]));
		if (side_prio >= 0) add("MF_SIDE_PRIO", std::to_string(side_prio));
		if (!sweep_pf) add("MF_SWEEP_PF", "0");
		if (sweep_prio >= 0) add("MF_SWEEP_PRIO", std::to_string(sweep_prio));
		if (sweep_mid != 0) add("MF_SWEEP_MID", std::to_string(sweep_mid));
		if (mid_nch) add("MF_SWEEP_MID_NCH", std::to_string(mid_nch));
		if (mid_coop) add("MF_SWEEP_MID_KERNEL", "coop");
		return s;
	}
};

}  // namespace
