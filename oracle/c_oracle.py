"""ctypes wrapper of oracle/libsrps_oracle.so (srps_oracle.c).  TEST INFRASTRUCTURE ONLY: the
checker's C restatement and the CPU baseline ("port") of bench.py.  PARITY UNPINNED (see
srps_oracle.py)."""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libsrps_oracle.so")
if not os.path.exists(_PATH):
    raise ImportError(f"{_PATH} not built (make -C oracle)")
_L = C.CDLL(_PATH)
f32 = np.float32
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


def _f(a): return a.ctypes.data_as(_fp)
def _i(a): return a.ctypes.data_as(_ip)


_L.oc_assemble.restype = C.c_long
_L.oc_energy.restype = C.c_double
_L.oc_num_threads.restype = C.c_int


def num_threads() -> int:
    return int(_L.oc_num_threads())


def set_threads(n: int):
    _L.oc_set_threads(C.c_int(n))


# The OpenMP team is bounded by what the process may really use (cpu_budget.py: affinity and cgroup quota) -- again before every call,
# because another library of the process (torch) sets the runtime's thread count to the machine's core count when it is imported.
import cpu_budget as _budget
_EFF = int(os.environ.get("SRPS_ORACLE_THREADS", "0")) or _budget.effective_cpus()


def _bound():
    if _L.oc_num_threads() > _EFF:
        _L.oc_set_threads(C.c_int(_EFF))



class Structure:
    """neighbour lists and KT blocks of a mask (make_gradient SRPS.cu:23-71, KT SRPS.cu:170-193)"""

    def __init__(self, h, w, sf, mask):
        _bound()
        mask = np.ascontiguousarray(mask, dtype=f32)
        P = C.c_int(0); Ps = C.c_int(0)
        _L.oc_count(h, w, sf, _f(mask), C.byref(P), C.byref(Ps))
        self.h, self.w, self.sf, self.P, self.Ps = h, w, sf, P.value, Ps.value
        self.imask = np.empty(self.P, np.int32); self.nb = np.empty(4 * self.P, np.int32)
        self.blk = np.empty(self.P, np.int32); self.blk_pix = np.empty(max(1, self.Ps * sf * sf), np.int32)
        rc = _L.oc_structure(h, w, sf, _f(mask), _i(self.imask), _i(self.nb), _i(self.blk), _i(self.blk_pix))
        assert rc == 0


def tensor(st: Structure, s, rho, dz, xx, yy, fx, fy, I):
    _bound()
    n_img, n_ch, P = I.shape
    M = np.empty(6 * P, f32); q = np.empty(3 * P, f32)
    a = [np.ascontiguousarray(v, dtype=f32) for v in (s, rho, dz, xx, yy, I)]
    _L.oc_tensor(P, n_img, n_ch, _f(a[0]), _f(a[1]), _f(a[2]), _f(a[3]), _f(a[4]), C.c_float(fx), C.c_float(fy), _f(a[5]), _f(M), _f(q))
    return M, q


def mf_apply(st: Structure, M, x, lam=1.0):
    _bound()
    x = np.ascontiguousarray(x, f32); y = np.empty(st.P, f32); work = np.empty(3 * st.P + st.Ps + 8, f32)
    _L.oc_mf_apply(st.P, st.Ps, st.sf, _i(st.nb), _i(st.blk), _i(st.blk_pix), _f(M), C.c_float(lam), _f(x), _f(y), _f(work))
    return y


def gradient(st: Structure, x):
    """zx = Dx x, zy = Dy x (rows of make_gradient, SRPS.cu:29-47)"""
    _bound()
    x = np.ascontiguousarray(x, f32); gx = np.empty(st.P, f32); gy = np.empty(st.P, f32)
    _L.oc_gradient(st.P, _i(st.nb), _f(x), _f(gx), _f(gy))
    return gx, gy


def rhs(st: Structure, q, z0s, lam=1.0):
    _bound()
    out = np.empty(st.P, f32); z0s = np.ascontiguousarray(z0s, f32)
    _L.oc_rhs(st.P, st.sf, _i(st.nb), _i(st.blk), _f(q), _f(z0s), C.c_float(lam), _f(out))
    return out


def assemble(st: Structure, M, lam=1.0):
    _bound()
    cap = st.P * (13 + st.sf * st.sf)
    rowptr = np.empty(st.P + 1, np.int32); col = np.empty(cap, np.int32); val = np.empty(cap, f32)
    nnz = _L.oc_assemble(st.P, st.sf, _i(st.nb), _i(st.blk), _i(st.blk_pix), _f(M), C.c_float(lam), _i(rowptr), _i(col), _f(val), C.c_long(cap))
    assert nnz >= 0, "assemble: capacity exceeded"
    return rowptr, col[:nnz], val[:nnz]


def csr_spmv(rowptr, col, val, x):
    _bound()
    x = np.ascontiguousarray(x, f32); y = np.empty(rowptr.size - 1, f32)
    _L.oc_csr_spmv(rowptr.size - 1, _i(rowptr), _i(col), _f(val), _f(x), _f(y))
    return y


def cg_csr(rowptr, col, val, x, b, tol=1e-9, max_iter=100, fixed_iters=0):
    _bound()
    return _L.oc_cg_csr(rowptr.size - 1, _i(rowptr), _i(col), _f(val), _f(x), _f(b), C.c_float(tol), max_iter, fixed_iters)


def cg_mf(st: Structure, M, x, b, lam=1.0, tol=1e-9, max_iter=100, fixed_iters=0):
    _bound()
    return _L.oc_cg_mf(st.P, st.Ps, st.sf, _i(st.nb), _i(st.blk), _i(st.blk_pix), _f(M), C.c_float(lam), _f(x), _f(b),
                       C.c_float(tol), max_iter, fixed_iters)


def energy(st: Structure, s, rho, dz, xx, yy, fx, fy, I, z0s, z, lam=1.0):
    _bound()
    n_img, n_ch, P = I.shape
    a = [np.ascontiguousarray(v, dtype=f32) for v in (s, rho, dz, xx, yy, I, z0s, z)]
    return float(_L.oc_energy(P, st.Ps, n_img, n_ch, st.sf, _i(st.nb), _i(st.blk_pix), _f(a[0]), _f(a[1]), _f(a[2]), _f(a[3]),
                              _f(a[4]), C.c_float(fx), C.c_float(fy), _f(a[5]), _f(a[6]), _f(a[7]), C.c_float(lam)))


def depth_estimation(st: Structure, s, rho, I, xx, yy, dz, z0s, z, fx, fy, assembled=True):
    """one depth step (dc.cu:636-786) in C: tensor, system, rhs, residual, CG (101 steps), energy"""
    M, q = tensor(st, s, rho, dz, xx, yy, fx, fy, I)
    b = rhs(st, q, z0s)
    if assembled:
        rp, ci, v = assemble(st, M)
        b = (b - csr_spmv(rp, ci, v, z)).astype(f32)
        it = cg_csr(rp, ci, v, z, b)
    else:
        b = (b - mf_apply(st, M, z)).astype(f32)
        it = cg_mf(st, M, z, b)
    return energy(st, s, rho, dz, xx, yy, fx, fy, I, z0s, z), it


def bench_cg_csr(rowptr, col, val, b0, iters, reps, threads):
    """seconds of each of `reps` solves of `iters` CG steps on `threads` threads, after a warm-up (oc_bench_cg_csr)"""
    _L.oc_bench_cg_csr.restype = C.c_int
    sec = np.zeros(reps, np.float64)
    b0 = np.ascontiguousarray(b0, f32)
    rc = _L.oc_bench_cg_csr(rowptr.size - 1, _i(rowptr), _i(col), _f(val), _f(b0), int(iters), int(reps), int(threads), sec.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0, "oc_bench_cg_csr: out of memory"
    return sec


def bench_cg_mf(st, M, b0, iters, reps, threads, lam=1.0):
    """the same for the matrix-free operator (oc_bench_cg_mf)"""
    _L.oc_bench_cg_mf.restype = C.c_int
    sec = np.zeros(reps, np.float64)
    b0 = np.ascontiguousarray(b0, f32); M = np.ascontiguousarray(M, f32)
    rc = _L.oc_bench_cg_mf(st.P, st.Ps, st.sf, _i(st.nb), _i(st.blk), _i(st.blk_pix), _f(M), C.c_float(lam), _f(b0), int(iters), int(reps), int(threads),
                           sec.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0, "oc_bench_cg_mf: out of memory"
    return sec


def bench_cpu_baseline(h, w, sf, mask, budget_s=20.0, reps=5):
    """bench.py cpu_baseline leg: BOTH CPU variants of BASELINE.md section 3 on the SAME HR grid -- the reference's formulation (assembled
    CSR + unfused BLAS-1 CG, dc.cu:229-279: `value`) and the matrix-free operator under the same recurrence (`matrix_free`: the stronger
    baseline, a third of the bytes).  Per variant: median of `reps` timed solves after two warm-up solves, at ONE thread and at all host
    threads, matrix / tensor and vectors first touched by the threads that stream them; the caller binds the threads through
    the environment (oracle/cpu_baseline_main.py is started with OMP_PROC_BIND=spread OMP_PLACES=cores).  The tensor is synthetic
    (values do not change the cost of an iteration)."""
    st = Structure(h, w, sf, mask)
    rng = np.random.default_rng(0)
    M = np.abs(rng.normal(size=(6, st.P))).astype(f32); M[[1, 2, 4]] *= 0.1
    M = M.reshape(-1)
    rp, ci, v = assemble(st, M)
    b = rng.normal(size=st.P).astype(f32)
    nthr = num_threads()
    runners = {"assembled_csr": lambda it, rp_, thr: bench_cg_csr(rp, ci, v, b, it, rp_, thr),
               "matrix_free": lambda it, rp_, thr: bench_cg_mf(st, M, b, it, rp_, thr)}
    res = {}
    for kind, run in runners.items():
        out = {}
        for label, thr in (("all", nthr), ("one", 1)):
            t_probe = float(run(3, 1, thr)[0]) / 3                                          # seconds per step, after the entry's own warm-ups
            share = budget_s * 0.5 * (0.5 if nthr > 1 else 1.0) / (reps + 2)                 # per solve, warm-ups included; half the budget per variant
            iters = int(max(3, min(101, share / max(t_probe, 1e-6))))
            sec = run(iters, reps, thr)
            out[label] = {"it_per_s": iters / float(np.median(sec)), "iters_per_solve": iters, "solves": [iters / float(t) for t in sec],
                          "spread": float((sec.max() - sec.min()) / np.median(sec)), "threads": thr}
            if nthr == 1:
                out["one"] = out["all"]
                break
        res[kind] = out
    a, o = res["assembled_csr"]["all"], res["assembled_csr"]["one"]
    ma, mo = res["matrix_free"]["all"], res["matrix_free"]["one"]
    return {"value": a["it_per_s"], "unit": "cg_iterations/s", "cores": nthr, "kind": "port",
            "value_1_thread": o["it_per_s"],
            "sample": f"median of {reps} solves of {a['iters_per_solve']} CG steps each (after two warm-up solves) of the assembled-CSR {h}x{w} system "
                      f"({v.size / st.P:.1f} nnz/row), C/OpenMP restatement of devicecalls.cu:229-279 on {nthr} threads; "
                      f"1 thread: {reps} solves of {o['iters_per_solve']} steps",
            "solves_it_per_s": a["solves"], "solves_it_per_s_1_thread": o["solves"], "spread_over_solves": a["spread"],
            "matrix_free": {"value": ma["it_per_s"], "unit": "cg_iterations/s", "cores": nthr, "value_1_thread": mo["it_per_s"],
                            "sample": f"median of {reps} solves of {ma['iters_per_solve']} CG steps each (after two warm-up solves) of the SAME system applied matrix-free "
                                      f"(oc_mf_apply: 6 tensor planes, 4 neighbour indices and the block index per unknown), same recurrence, {nthr} threads; "
                                      f"1 thread: {reps} solves of {mo['iters_per_solve']} steps",
                            "solves_it_per_s": ma["solves"], "spread_over_solves": ma["spread"],
                            "note": "BASELINE.md section 3's second CPU variant: the formulation the GPU path uses, on the host cores"},
            "thread_binding": {k: os.environ.get(k) for k in ("OMP_PROC_BIND", "OMP_PLACES", "OMP_NUM_THREADS")},
            "first_touch": "matrix / tensor and vectors copied and first touched by the threads that stream them (static schedule)"}


# ---- the whole alternating loop, matrix-free, in one arithmetic type (srps_solve_mf.c) ---------------------------------------------
_dp = C.POINTER(C.c_double)


class _Real:
    """entry points of one build of srps_solve_mf.c: 'f64' (oc64_*, the calibration reference) or 'f32' (oc32_*)"""

    def __init__(self, precision):
        self.pfx = {"f64": "oc64", "f32": "oc32"}[precision]
        self.dtype = {"f64": np.float64, "f32": np.float32}[precision]
        self.ctype = {"f64": C.c_double, "f32": C.c_float}[precision]
        assert getattr(_L, self.pfx + "_sizeof_real")() == np.dtype(self.dtype).itemsize

    def fn(self, name):
        return getattr(_L, f"{self.pfx}_{name}")

    def p(self, a):
        assert a.dtype == self.dtype and a.flags.c_contiguous
        return a.ctypes.data_as(C.POINTER(self.ctype))


def solve_mf(st: Structure, I, z0s, z_init, xx, yy, fx, fy, precision="f64", max_outer=None, keep_z=False, lam=1.0):
    """SRPS::execute's loop (SRPS.cu:272-335) with every phase matrix-free in `precision` (srps_solve_mf.c): start values of
    SRPS.cu:209-217 / dc.cu:133-139, first normals from z_init, then lighting -> albedo -> depth -> stop rule on THIS run's energies
    (SRPS.cu:297-302) -> normals.  I: float32 [n][c][P] compact (not copied).  Returns energies, the final state, the CG step counts
    per pass and, with keep_z, the depth after every pass."""
    _bound()
    R = _Real(precision)
    T = R.dtype
    n_img, n_ch, P = I.shape
    assert P == st.P and I.dtype == f32 and I.flags.c_contiguous
    cast = lambda a: np.ascontiguousarray(np.asarray(a), dtype=T)
    s = np.zeros((n_img, n_ch, 4), T); s[:, :, 2] = -1                       # SRPS.cu:209-217
    rho = np.full((n_ch, P), 0.5, T)                                         # dc.cu:133-139
    z, z0s, xx, yy = cast(z_init).copy(), cast(z0s), cast(xx), cast(yy)
    N = np.empty((4, P), T); dz = np.empty(P, T)
    fxr, fyr = R.ctype(float(T(f32(fx)))), R.ctype(float(T(f32(fy))))        # K is float in the reference (SRPS.cu:269)
    R.fn("normals")(P, _i(st.nb), R.p(z), R.p(xx), R.p(yy), fxr, fyr, R.p(N), R.p(dz))
    energies, steps, z_pass = [], [], []
    last_error, iteration = float("nan"), 1
    OUTER_TOL, OUTER_MAX = 5e-3, 10                                          # SRPS.cu:85-86
    while True:
        it_l = np.zeros(n_img * n_ch, np.int32); it_a = np.zeros(n_ch, np.int32)
        rc = R.fn("lighting")(P, n_img, n_ch, R.p(rho.reshape(-1)), R.p(N.reshape(-1)), _f(I), R.p(s.reshape(-1)), _i(it_l))
        assert rc == 0
        rc = R.fn("albedo")(P, n_img, n_ch, R.p(s.reshape(-1)), R.p(N.reshape(-1)), _f(I), R.p(rho.reshape(-1)), _i(it_a))
        assert rc == 0
        e = C.c_double(0); it_d = C.c_int(0)
        rc = R.fn("depth")(P, st.Ps, n_img, n_ch, st.sf, _i(st.nb), _i(st.blk), _i(st.blk_pix), R.p(s.reshape(-1)), R.p(rho.reshape(-1)),
                           R.p(dz), R.p(xx), R.p(yy), fxr, fyr, _f(I), R.p(z0s), R.ctype(lam), R.p(z), C.byref(e), C.byref(it_d))
        assert rc == 0
        error = float(T(e.value)) if precision == "f32" else e.value          # the reference's energy is a float (dc.cu:785)
        energies.append(error)
        steps.append(dict(lighting=it_l.tolist(), albedo=it_a.tolist(), depth=int(it_d.value)))
        if keep_z:
            z_pass.append(z.copy())
        with np.errstate(invalid="ignore", divide="ignore"):
            rel_err = abs(T(last_error) - T(error)) / abs(T(error))
        stop = (error > last_error) or (rel_err < T(OUTER_TOL)) or (iteration > OUTER_MAX)   # SRPS.cu:298-301
        last_error = error
        iteration += 1
        if stop or (max_outer is not None and len(energies) >= max_outer):
            break
        R.fn("normals")(P, _i(st.nb), R.p(z), R.p(xx), R.p(yy), fxr, fyr, R.p(N.reshape(-1)), R.p(dz))   # SRPS.cu:304-313
    return dict(energies=energies, z=z, rho=rho, s=s, N=N, dz=dz, steps=steps, z_pass=z_pass, precision=precision)
