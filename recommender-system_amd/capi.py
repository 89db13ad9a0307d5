"""ctypes bindings of the two in-tree libraries -- the same stubs INTEGRATION.md shows for the reference.

  csrc/libmatfact_hip.so   include/matfact_hip.h   (HIP kernels, C ABI; replaces matFact.c:29 / :10 / mat2d.c:100)
  host/libmatfact_host.so  include/matfact_host.h  (parser, init, partition, synthetic generator)

Loading is strict: a missing library raises ImportError; nothing here computes on the CPU in its place.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MF_HIP_LIB: alternative build of the same library (A/B timing of kernel variants); never a different backend
HIP_LIB_PATH = os.environ.get("MF_HIP_LIB") or os.path.join(_HERE, "csrc", "libmatfact_hip.so")
HOST_LIB_PATH = os.path.join(_HERE, "host", "libmatfact_host.so")
CLI_PATH = os.path.join(_HERE, "host", "matFact")

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")

MF_OK = 0
MF_ERR_ARGUMENT, MF_ERR_NO_DEVICE, MF_ERR_HIP, MF_ERR_NO_MEMORY, MF_ERR_UNSUPPORTED, MF_ERR_STATE = -1, -2, -3, -4, -5, -6

# every symbol include/matfact_hip.h declares (tests check the library exports each one)
HIP_SYMBOLS = [
    "mf_backend_strerror", "mf_backend_last_hip_error", "mf_backend_abi_version", "mf_backend_device_count",
    "mf_backend_factorize", "mf_backend_recommend", "mf_backend_run", "mf_backend_run_multi", "mf_backend_run_top1",
    "mf_backend_multi_last_timing", "mf_backend_multi_last_counters",
    "mf_plan_create", "mf_backend_row_pitch", "mf_plan_row_pitch", "mf_plan_destroy", "mf_plan_set_stream", "mf_plan_upload_factors",
    "mf_plan_download_factors", "mf_plan_iterate", "mf_plan_sweep_items", "mf_plan_sweep_users",
    "mf_plan_items_next", "mf_plan_items_current", "mf_plan_flip", "mf_plan_recommend", "mf_plan_recommend_info",
    "mf_plan_sweep_users_seeded", "mf_plan_users_next", "mf_plan_users_current", "mf_plan_recommend_scored",
    "mf_plan_recommend_scored_users", "mf_plan_recommend_filter", "mf_backend_recommend_margin",
    "mf_plan_predict", "mf_plan_synchronize",
    "mf_plan_timing", "mf_plan_timing_read", "mf_plan_describe",
]
HOST_SYMBOLS = [
    "mf_host_parse_strerror", "mf_host_parse_file", "mf_host_parse_buffer", "mf_host_free_problem",
    "mf_host_parse_file_cached",
    "mf_host_srandom", "mf_host_random", "mf_host_init_factors", "mf_host_init_factors_block",
    "mf_host_split_entries", "mf_host_partition_users", "mf_host_balanced_grid", "mf_host_write_out", "mf_host_checkpoint_write",
    "mf_host_checkpoint_read", "mf_host_synth_counts",
    "mf_host_synth_fill",
]


class HipBackendError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = hip().mf_backend_strerror(status).decode()
        detail = hip().mf_backend_last_hip_error().decode() if status == MF_ERR_HIP else ""
        super().__init__("%s: %s (%d) %s" % (where, msg, status, detail))


class Entry(C.Structure):  # mf_entry == non_zero_entry (datatypes.h:10-15)
    _fields_ = [("row", C.c_int32), ("col", C.c_int32), ("value", C.c_double)]


class Problem(C.Structure):  # mf_problem
    _fields_ = [("users", C.c_int32), ("items", C.c_int32), ("features", C.c_int32), ("iters", C.c_int32),
                ("alpha", C.c_double), ("nnz", C.c_int64), ("entries", C.POINTER(Entry))]


class Shard(C.Structure):  # mf_shard
    _fields_ = [("users_total", C.c_int32), ("items", C.c_int32), ("features", C.c_int32),
                ("user_begin", C.c_int32), ("user_count", C.c_int32), ("nnz", C.c_int64),
                ("row", C.c_void_p), ("col", C.c_void_p), ("val", C.c_void_p), ("alpha", C.c_double),
                ("device", C.c_int32), ("flags", C.c_int32), ("items_ext", C.c_void_p * 2),
                ("users_ext", C.c_void_p * 2), ("items_pitch", C.c_int32), ("users_pitch", C.c_int32)]


# mf_candidate: the partial scan state of mf_plan_recommend_scored, as a numpy record layout
# mf_filter: pass-1 (matrix-core) report of mf_plan_recommend_filter
FILTER_DTYPE = np.dtype([("best", np.float64), ("second", np.float64), ("arg", np.int32), ("nonfinite", np.int32)])
CANDIDATE_DTYPE = np.dtype([("score", np.float64), ("best", np.int32), ("first", np.int32),
                            ("first_nan", np.int32), ("reserved", np.int32)])


class Synth(C.Structure):  # mf_synth
    _fields_ = [("seed", C.c_uint64), ("users", C.c_int32), ("items", C.c_int32), ("min_row", C.c_int32),
                ("max_row", C.c_int32), ("mode", C.c_int32), ("reserved", C.c_int32), ("target_nnz", C.c_int64)]


SYNTH_MODES = {"stratified": 0, "uniform": 1, "zipf": 2}


class Rand(C.Structure):  # mf_rand
    _fields_ = [("ring", C.c_int32 * 31), ("f", C.c_int), ("b", C.c_int)]


_hip = None
_host = None


def hip():
    """The HIP backend library.  Raises ImportError when it has not been built (no fallback)."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback" % HIP_LIB_PATH)
        lib = C.CDLL(HIP_LIB_PATH)
        P = C.c_void_p
        lib.mf_backend_strerror.restype = C.c_char_p
        lib.mf_backend_strerror.argtypes = [C.c_int]
        lib.mf_backend_last_hip_error.restype = C.c_char_p
        lib.mf_backend_run.argtypes = [C.POINTER(Problem), _f64p, _f64p, _i32p, C.c_int]
        lib.mf_backend_run_multi.argtypes = [C.POINTER(Problem), _f64p, _f64p, _i32p, _i32p, C.c_int]
        lib.mf_backend_multi_last_timing.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double),
                                                     C.POINTER(C.c_double), C.POINTER(C.c_int * 3)]
        lib.mf_backend_multi_last_counters.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int)]
        lib.mf_backend_run_top1.argtypes = [C.POINTER(Problem), _f64p, _f64p, _i32p, C.c_int]
        lib.mf_backend_factorize.argtypes = [C.POINTER(Problem), _f64p, _f64p, C.c_int]
        lib.mf_backend_recommend.argtypes = [C.POINTER(Problem), _f64p, _f64p, _i32p, C.c_int]
        lib.mf_plan_create.argtypes = [C.POINTER(P), C.POINTER(Shard)]
        lib.mf_backend_row_pitch.argtypes = [C.c_int]
        lib.mf_plan_row_pitch.argtypes = [P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        lib.mf_plan_destroy.argtypes = [P]
        lib.mf_plan_destroy.restype = None
        lib.mf_plan_set_stream.argtypes = [P, P]
        lib.mf_plan_upload_factors.argtypes = [P, P, P]
        lib.mf_plan_download_factors.argtypes = [P, P, P]
        lib.mf_plan_iterate.argtypes = [P, C.c_int]
        lib.mf_plan_sweep_items.argtypes = [P, C.c_int]
        lib.mf_plan_sweep_users.argtypes = [P]
        lib.mf_plan_items_next.argtypes = [P]
        lib.mf_plan_items_next.restype = P
        lib.mf_plan_items_current.argtypes = [P]
        lib.mf_plan_items_current.restype = P
        lib.mf_plan_flip.argtypes = [P]
        lib.mf_plan_sweep_users_seeded.argtypes = [P, C.c_int]
        lib.mf_plan_users_next.argtypes = [P]
        lib.mf_plan_users_next.restype = P
        lib.mf_plan_users_current.argtypes = [P]
        lib.mf_plan_users_current.restype = P
        lib.mf_plan_recommend_scored.argtypes = [P, P]
        lib.mf_plan_recommend_scored_users.argtypes = [P, _i32p, C.c_int32, P]
        lib.mf_plan_recommend_filter.argtypes = [P, P, _f64p, C.POINTER(C.c_double)]
        lib.mf_backend_recommend_margin.argtypes = [C.c_int]
        lib.mf_backend_recommend_margin.restype = C.c_double
        lib.mf_plan_recommend.argtypes = [P, _i32p]
        lib.mf_plan_recommend_info.argtypes = [P, C.POINTER(C.c_int64)]
        lib.mf_plan_predict.argtypes = [P, _f64p]
        lib.mf_plan_synchronize.argtypes = [P]
        lib.mf_plan_timing.argtypes = [P, C.c_int]
        lib.mf_plan_timing_read.argtypes = [P, C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                            C.POINTER(C.c_int64), C.POINTER(C.c_double)]
        lib.mf_plan_describe.argtypes = [P, C.c_char_p, C.c_int]
        _hip = lib
    return _hip


def host():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise ImportError("%s is missing: run __graft_entry__.build()" % HOST_LIB_PATH)
        lib = C.CDLL(HOST_LIB_PATH)
        lib.mf_host_parse_strerror.restype = C.c_char_p
        lib.mf_host_parse_strerror.argtypes = [C.c_int]
        lib.mf_host_parse_file.argtypes = [C.c_char_p, C.POINTER(Problem)]
        lib.mf_host_parse_file_cached.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(Problem), C.POINTER(C.c_int)]
        lib.mf_host_parse_buffer.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Problem)]
        lib.mf_host_free_problem.argtypes = [C.POINTER(Problem)]
        lib.mf_host_free_problem.restype = None
        lib.mf_host_srandom.argtypes = [C.POINTER(Rand), C.c_uint]
        lib.mf_host_srandom.restype = None
        lib.mf_host_random.argtypes = [C.POINTER(Rand)]
        lib.mf_host_random.restype = C.c_int32
        lib.mf_host_init_factors.argtypes = [C.c_int, C.c_int, C.c_int, _f64p, _f64p]
        lib.mf_host_init_factors.restype = None
        lib.mf_host_init_factors_block.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f64p, C.c_void_p]
        lib.mf_host_init_factors_block.restype = None
        lib.mf_host_split_entries.argtypes = [C.POINTER(Entry), C.c_int64, _i32p, _i32p, _f64p]
        lib.mf_host_split_entries.restype = None
        lib.mf_host_partition_users.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, _i32p]
        lib.mf_host_balanced_grid.argtypes = [C.c_int, C.c_int, C.c_int, _i32p]
        lib.mf_host_synth_counts.argtypes = [C.POINTER(Synth), C.c_int, C.c_int, _i32p]
        lib.mf_host_synth_counts.restype = C.c_int64
        lib.mf_host_synth_fill.argtypes = [C.POINTER(Synth), C.c_int, C.c_int, _i32p, _i32p, _i32p, _f64p]
        _host = lib
    return _host


def _check(status, where):
    if status != MF_OK:
        raise HipBackendError(status, where)


# ------------------------------------------------------------------------------------ host helpers
class Instance:
    """A parsed instance in SoA form (what mf_shard wants)."""

    def __init__(self, iters, alpha, feats, users, items, row, col, val):
        self.iters, self.alpha, self.feats = int(iters), float(alpha), int(feats)
        self.users, self.items = int(users), int(items)
        self.row = np.ascontiguousarray(row, np.int32)
        self.col = np.ascontiguousarray(col, np.int32)
        self.val = np.ascontiguousarray(val, np.float64)

    @property
    def nnz(self):
        return int(self.row.shape[0])


class ParseError(ValueError):
    def __init__(self, status):
        self.status = status
        super().__init__(host().mf_host_parse_strerror(status).decode())


def parse_file(path):
    """`.in` reader through the C parser (mf_host_parse_file); gz files are inflated first."""
    if str(path).endswith(".gz"):
        import gzip
        with gzip.open(path, "rb") as f:
            return parse_text(f.read())
    p = Problem()
    rc = host().mf_host_parse_file(os.fsencode(path), C.byref(p))
    if rc != 0:
        raise ParseError(rc)
    return _from_problem(p)


def parse_file_cached(path, cache_dir):
    """(instance, cache_hit) through the binary cache of mf_host_parse_file_cached."""
    p = Problem()
    hit = C.c_int(0)
    rc = host().mf_host_parse_file_cached(os.fsencode(path), os.fsencode(cache_dir), C.byref(p), C.byref(hit))
    if rc != 0:
        raise ParseError(rc)
    return _from_problem(p), bool(hit.value)


def parse_text(text):
    if isinstance(text, str):
        text = text.encode()
    p = Problem()
    rc = host().mf_host_parse_buffer(text, len(text), C.byref(p))
    if rc != 0:
        raise ParseError(rc)
    return _from_problem(p)


def _from_problem(p):
    n = int(p.nnz)
    row, col, val = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.float64)
    host().mf_host_split_entries(p.entries, n, row, col, val)
    inst = Instance(p.iters, p.alpha, p.features, p.users, p.items, row, col, val)
    host().mf_host_free_problem(C.byref(p))
    return inst


def init_factors(users, items, feats):
    """Initial L (users x K) and R (items x K): mat2d_random_fill_LR + transpose, own glibc random()."""
    L = np.empty((users, feats), np.float64)
    R = np.empty((items, feats), np.float64)
    host().mf_host_init_factors(users, items, feats, L, R)
    return L, R


def init_factors_block(users, items, feats, u0, count, want_r=True):
    Lb = np.empty((count, feats), np.float64)
    R = np.empty((items, feats), np.float64) if want_r else None
    host().mf_host_init_factors_block(users, items, feats, u0, count, Lb,
                                      R.ctypes.data if R is not None else None)
    return Lb, R


def partition_users(users, parts, row_ptr=None):
    """begin[0..parts]; by entry count when the CSR row pointer (int64, users+1) is given, else BLOCK_LOW."""
    begin = np.empty(parts + 1, np.int32)
    if row_ptr is not None:
        row_ptr = np.ascontiguousarray(row_ptr, np.int64)
        rc = host().mf_host_partition_users(users, parts, 1, row_ptr.ctypes.data, begin)
    else:
        rc = host().mf_host_partition_users(users, parts, 0, None, begin)
    if rc != 0:
        raise ValueError("mf_host_partition_users failed")
    return begin


def row_pitch(feats):
    """Row pitch (doubles) the backend gives factor buffers it owns for this K (K, or K padded to whole 128-byte
    lines); caller-owned buffers may use the same and declare it (Plan(items_pitch=..., users_pitch=...))."""
    return int(hip().mf_backend_row_pitch(int(feats)))


def recommend_margin(feats):
    """8*(K+8)*2^-53: the factor of ||l||*||r|| above which a matrix-core score gap certifies the arg-max."""
    return float(hip().mf_backend_recommend_margin(int(feats)))


def balanced_grid(users, items, nproc):
    """(grid rows, grid cols) of the reference's 2-D process grid (create_balanced_grid, mpiutil.c:54-88)."""
    size = np.empty(2, np.int32)
    if host().mf_host_balanced_grid(int(users), int(items), int(nproc), size) != 0:
        raise ValueError("mf_host_balanced_grid failed")
    return int(size[0]), int(size[1])


def synth_counts(seed, users, items, min_row, max_row, u0=0, count=None, columns="stratified", target_nnz=0):
    count = users - u0 if count is None else count
    s = Synth(seed, users, items, min_row, max_row, SYNTH_MODES[columns], 0, int(target_nnz))
    counts = np.empty(count, np.int32)
    total = host().mf_host_synth_counts(C.byref(s), u0, count, counts)
    return counts, int(total)


def synth_block(seed, users, items, min_row, max_row, u0=0, count=None, columns="stratified", target_nnz=0):
    """(row, col, val) of users [u0, u0+count) of the synthetic instance, (row, col)-sorted.  columns: "stratified" (one
    column per stratum), "uniform" (distinct uniform columns) or "zipf" (Zipf(1.0) item popularity); target_nnz > 0
    rescales the row counts so that the WHOLE instance has exactly that many entries."""
    count = users - u0 if count is None else count
    counts, total = synth_counts(seed, users, items, min_row, max_row, u0, count, columns, target_nnz)
    row, col, val = np.empty(total, np.int32), np.empty(total, np.int32), np.empty(total, np.float64)
    s = Synth(seed, users, items, min_row, max_row, SYNTH_MODES[columns], 0, int(target_nnz))
    if host().mf_host_synth_fill(C.byref(s), u0, count, counts, row, col, val) != 0:
        raise MemoryError("mf_host_synth_fill")
    return row, col, val


def device_count():
    return hip().mf_backend_device_count()


def file_sha256(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def kernel_source_hash():
    """Hash (16 hex digits) of the kernel sources under csrc/: measurements kept in files (PMC traffic) are tied to
    it, so a record is never quoted for kernels it was not measured on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.hip.h"))):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


# ------------------------------------------------------------------------------------ level 1
def _problem(inst, iters=None):
    ent = (Entry * max(inst.nnz, 1))()
    buf = np.ctypeslib.as_array(ent).view(np.dtype([("row", np.int32), ("col", np.int32), ("value", np.float64)]))
    buf["row"][:inst.nnz] = inst.row
    buf["col"][:inst.nnz] = inst.col
    buf["value"][:inst.nnz] = inst.val
    p = Problem(inst.users, inst.items, inst.feats, inst.iters if iters is None else iters, inst.alpha,
                inst.nnz, ent)
    return p, ent


def backend_run(inst, L, R, iters=None, device=0):
    """mf_backend_run: L, R updated in place, returns best[users]."""
    p, keep = _problem(inst, iters)
    best = np.empty(inst.users, np.int32)
    _check(hip().mf_backend_run(C.byref(p), L, R, best, device), "mf_backend_run")
    return best


def backend_run_multi(inst, L, R, devices, iters=None):
    """mf_backend_run_multi: one process, len(devices) shards (ordinals may repeat)."""
    p, keep = _problem(inst, iters)
    best = np.empty(inst.users, np.int32)
    dev = np.ascontiguousarray(devices, np.int32)
    _check(hip().mf_backend_run_multi(C.byref(p), L, R, best, dev, len(dev)), "mf_backend_run_multi")
    return best


def multi_last_timing():
    """Host wall-clock and counters of the last backend_run_multi: dict(setup_s, iterate_s, recommend_s, shards, reducer,
    sliced, enqueue_s, entry_passes, host_threads)."""
    a, b, c, e = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    info = (C.c_int * 3)()
    passes, threads = C.c_int64(), C.c_int()
    _check(hip().mf_backend_multi_last_timing(C.byref(a), C.byref(b), C.byref(c), C.byref(info)),
           "mf_backend_multi_last_timing")
    _check(hip().mf_backend_multi_last_counters(C.byref(e), C.byref(passes), C.byref(threads)),
           "mf_backend_multi_last_counters")
    return {"setup_s": a.value, "iterate_s": b.value, "recommend_s": c.value, "shards": info[0],
            "reducer": "rccl" if info[1] else "peer", "sliced": bool(info[2]), "enqueue_s": e.value,
            "entry_passes": passes.value, "host_threads": threads.value}


def backend_run_top1(inst, L0, R0, iters=None, device=0):
    """mf_backend_run_top1: initial factors in, only the recommendation list out."""
    p, keep = _problem(inst, iters)
    best = np.empty(inst.users, np.int32)
    _check(hip().mf_backend_run_top1(C.byref(p), L0, R0, best, device), "mf_backend_run_top1")
    return best


def backend_factorize(inst, L, R, iters=None, device=0):
    p, keep = _problem(inst, iters)
    _check(hip().mf_backend_factorize(C.byref(p), L, R, device), "mf_backend_factorize")


def backend_recommend(inst, L, R, device=0):
    p, keep = _problem(inst)
    best = np.empty(inst.users, np.int32)
    _check(hip().mf_backend_recommend(C.byref(p), L, R, best, device), "mf_backend_recommend")
    return best


# ------------------------------------------------------------------------------------ level 2
class Plan:
    """mf_plan: one shard resident on one GPU."""

    def __init__(self, users_total, items, feats, alpha, row, col, val, user_begin=0, user_count=None,
                 device=0, items_ext=None, flags=0, users_ext=None, items_pitch=0, users_pitch=0):
        self.users_total, self.items, self.feats = int(users_total), int(items), int(feats)
        self.user_begin = int(user_begin)
        self.user_count = int(users_total - user_begin if user_count is None else user_count)
        self._keep = (np.ascontiguousarray(row, np.int32), np.ascontiguousarray(col, np.int32),
                      np.ascontiguousarray(val, np.float64))
        s = Shard()
        s.users_total, s.items, s.features = self.users_total, self.items, self.feats
        s.user_begin, s.user_count = self.user_begin, self.user_count
        s.nnz = int(self._keep[0].shape[0])
        s.row, s.col, s.val = (a.ctypes.data for a in self._keep)
        s.alpha, s.device, s.flags = float(alpha), int(device), int(flags)
        if items_ext is not None:
            s.items_ext[0], s.items_ext[1] = int(items_ext[0]), int(items_ext[1])
        if users_ext is not None:
            s.users_ext[0], s.users_ext[1] = int(users_ext[0]), int(users_ext[1])
        s.items_pitch, s.users_pitch = int(items_pitch), int(users_pitch)
        self.nnz = s.nnz
        self._h = C.c_void_p()
        _check(hip().mf_plan_create(C.byref(self._h), C.byref(s)), "mf_plan_create")
        self._keep = None  # the plan copied everything it needs into HBM

    def close(self):
        if getattr(self, "_h", None):
            hip().mf_plan_destroy(self._h)
            self._h = None

    __del__ = close

    def set_stream(self, stream_ptr):
        _check(hip().mf_plan_set_stream(self._h, stream_ptr), "mf_plan_set_stream")

    def upload(self, L_block, R):
        L_block = np.ascontiguousarray(L_block, np.float64)
        R = np.ascontiguousarray(R, np.float64)
        assert L_block.shape == (self.user_count, self.feats) and R.shape == (self.items, self.feats)
        _check(hip().mf_plan_upload_factors(self._h, L_block.ctypes.data, R.ctypes.data), "mf_plan_upload_factors")

    def download(self, want_l=True, want_r=True):
        Lb = np.empty((self.user_count, self.feats), np.float64) if want_l else None
        R = np.empty((self.items, self.feats), np.float64) if want_r else None
        _check(hip().mf_plan_download_factors(self._h, Lb.ctypes.data if want_l else None,
                                              R.ctypes.data if want_r else None), "mf_plan_download_factors")
        return Lb, R

    def iterate(self, iters):
        _check(hip().mf_plan_iterate(self._h, int(iters)), "mf_plan_iterate")

    def sweep_items(self, seed_from_old=True):
        _check(hip().mf_plan_sweep_items(self._h, 1 if seed_from_old else 0), "mf_plan_sweep_items")

    def sweep_users(self, seed_from_old=True):
        _check(hip().mf_plan_sweep_users_seeded(self._h, 1 if seed_from_old else 0), "mf_plan_sweep_users_seeded")

    def users_next_ptr(self):
        return hip().mf_plan_users_next(self._h)

    def users_current_ptr(self):
        return hip().mf_plan_users_current(self._h)

    def items_next_ptr(self):
        return hip().mf_plan_items_next(self._h)

    def items_current_ptr(self):
        return hip().mf_plan_items_current(self._h)

    def flip(self):
        _check(hip().mf_plan_flip(self._h), "mf_plan_flip")

    def recommend(self):
        best = np.empty(self.user_count, np.int32)
        _check(hip().mf_plan_recommend(self._h, best), "mf_plan_recommend")
        return best

    def recommend_scored(self):
        """Partial scan state per user (CANDIDATE_DTYPE records); item ids relative to this plan's item block."""
        out = np.zeros(self.user_count, CANDIDATE_DTYPE)
        _check(hip().mf_plan_recommend_scored(self._h, out.ctypes.data), "mf_plan_recommend_scored")
        return out

    def recommend_scored_users(self, users):
        """The same for the listed users (local ids) only; record t belongs to users[t]."""
        users = np.ascontiguousarray(users, np.int32)
        out = np.zeros(users.shape[0], CANDIDATE_DTYPE)
        _check(hip().mf_plan_recommend_scored_users(self._h, users, users.shape[0], out.ctypes.data),
               "mf_plan_recommend_scored_users")
        return out

    def recommend_filter(self):
        """Pass 1 alone: (FILTER_DTYPE records, ||L[i]|| per user, max ||R[j]|| over this plan's items)."""
        out = np.zeros(self.user_count, FILTER_DTYPE)
        norm = np.zeros(self.user_count, np.float64)
        rmax = C.c_double()
        _check(hip().mf_plan_recommend_filter(self._h, out.ctypes.data, norm, C.byref(rmax)),
               "mf_plan_recommend_filter")
        return out, norm, rmax.value

    def pitches(self):
        """(users_pitch, items_pitch) of the plan's L and R buffers in doubles."""
        a, b = C.c_int32(), C.c_int32()
        _check(hip().mf_plan_row_pitch(self._h, C.byref(a), C.byref(b)), "mf_plan_row_pitch")
        return a.value, b.value

    def recommend_info(self):
        n = C.c_int64()
        _check(hip().mf_plan_recommend_info(self._h, C.byref(n)), "mf_plan_recommend_info")
        return n.value

    def predict(self):
        B = np.empty((self.user_count, self.items), np.float64)
        _check(hip().mf_plan_predict(self._h, B), "mf_plan_predict")
        return B

    def synchronize(self):
        _check(hip().mf_plan_synchronize(self._h), "mf_plan_synchronize")

    def timing(self, enable=True):
        _check(hip().mf_plan_timing(self._h, 1 if enable else 0), "mf_plan_timing")

    def timing_read(self):
        il, ul = C.c_int64(), C.c_int64()
        ims, ums = C.c_double(), C.c_double()
        _check(hip().mf_plan_timing_read(self._h, C.byref(il), C.byref(ims), C.byref(ul), C.byref(ums)),
               "mf_plan_timing_read")
        return {"item_launches": il.value, "item_ms": ims.value, "user_launches": ul.value, "user_ms": ums.value}

    def describe(self):
        buf = C.create_string_buffer(512)
        _check(hip().mf_plan_describe(self._h, buf, 512), "mf_plan_describe")
        return buf.value.decode()
