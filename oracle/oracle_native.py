"""ctypes wrapper of oracle/oracle_kernels.cpp (C++17/OpenMP restatement) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py import this.  The shared
object is compiled for the CPU it runs on (-march=native) on first use, so the prebuilt file of one
machine is never loaded on another."""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import platform
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def _cpu_tag() -> str:
    model = platform.processor()
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") or line.startswith("flags"):
                    model += line
                    if line.startswith("flags"):
                        break
    except OSError:
        pass
    return hashlib.sha1(model.encode()).hexdigest()[:10]


def _build() -> str:
    out_dir = os.path.join(_HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, f"liboracle_kernels_{_cpu_tag()}.so")
    src = os.path.join(_HERE, "oracle_kernels.cpp")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        cmd = ["g++", "-O3", "-march=native", "-std=c++17", "-fPIC", "-fopenmp", "-shared", "-o", out, src]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("building the native oracle failed:\n" + res.stderr[-3000:])
    return out


_lib = None


def load():
    global _lib
    if _lib is None:
        lib = C.CDLL(_build())
        lib.oracle_num_threads.restype = C.c_int
        lib.oracle_set_num_threads.argtypes = [C.c_int]
        if "OMP_NUM_THREADS" not in os.environ:
            lib.oracle_set_num_threads(effective_cpu_count())
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def effective_cpu_count() -> int:
    """min(affinity mask, cgroup cpu.max quota): the host cores this process may really use."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(n, 1)


def num_threads() -> int:
    return load().oracle_num_threads()


def set_num_threads(t: int):
    load().oracle_set_num_threads(int(t))


def mf_apply(n, h, cell_dofs, coef, con, x):
    lib = load()
    n_a = np.asarray(n, dtype=np.int32)
    h_a = np.asarray(h, dtype=np.float64)
    cd = np.ascontiguousarray(cell_dofs, dtype=np.int32)
    co = np.ascontiguousarray(coef, dtype=np.float64)
    cn = np.ascontiguousarray(con, dtype=np.uint8)
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    lib.oracle_mf_apply(_p(n_a), _p(h_a), _p(cd), _p(co), _p(cn), _p(x), _p(y))
    return y


def mf_diagonal(n, h, cell_dofs, coef, con):
    """compute_diagonal of the matrix-free operator (constrained entries one)."""
    lib = load()
    n_a = np.asarray(n, dtype=np.int32)
    h_a = np.asarray(h, dtype=np.float64)
    cd = np.ascontiguousarray(cell_dofs, dtype=np.int32)
    co = np.ascontiguousarray(coef, dtype=np.float64)
    cn = np.ascontiguousarray(con, dtype=np.uint8)
    d = np.empty(cn.shape[0])
    lib.oracle_mf_diagonal(_p(n_a), _p(h_a), _p(cd), _p(co), _p(cn), _p(d))
    return d


def csr_spmv(A, x):
    lib = load()
    rp = np.ascontiguousarray(A.indptr, dtype=np.int32)
    cl = np.ascontiguousarray(A.indices, dtype=np.int32)
    vl = np.ascontiguousarray(A.data, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty(A.shape[0])
    lib.oracle_csr_spmv(C.c_int64(A.shape[0]), _p(rp), _p(cl), _p(vl), _p(x), _p(y))
    return y


class _Csr(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("n_cols", C.c_int64), ("rp", C.c_void_p), ("col", C.c_void_p),
                ("val", C.c_void_p)]


class _AmgLevel(C.Structure):
    _fields_ = [("A", _Csr), ("P", _Csr), ("Pt", _Csr), ("degree", C.c_int32), ("lmin", C.c_double),
                ("lmax", C.c_double)]


def _csr_struct(M, keep):
    M = M.tocsr()
    M.sort_indices()
    arrs = (np.ascontiguousarray(M.indptr, dtype=np.int32), np.ascontiguousarray(M.indices, dtype=np.int32),
            np.ascontiguousarray(M.data, dtype=np.float64))
    keep.append(arrs)
    return _Csr(M.shape[0], M.shape[1], arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data)


def vcycles(n, h, cell_dofs, coef, con, dinv, degree, lmin, lmax, R, Ac, coarse_iters, b, x0, n_cycles,
            want_history=True, amg_levels=None, amg_pre_smoothing_levels=None):
    """n_cycles V-cycles (matrix-free fine level, Chebyshev(degree); coarse 'solve' = PCG(coarse_iters), or
    one V-cycle of the aggregation hierarchy `amg_levels` = [(A_l, P_l or None, (deg, lmin, lmax) or None)], its levels
    from `amg_pre_smoothing_levels` on without pre-smoother);
    returns (x, history or None)."""
    lib = load()
    n_a = np.asarray(n, dtype=np.int32)
    h_a = np.asarray(h, dtype=np.float64)
    cd = np.ascontiguousarray(cell_dofs, dtype=np.int32)
    co = np.ascontiguousarray(coef, dtype=np.float64)
    cn = np.ascontiguousarray(con, dtype=np.uint8)
    dinv = np.ascontiguousarray(dinv, dtype=np.float64)
    R = R.tocsr()
    R.sort_indices()
    Rt = R.T.tocsr()
    Rt.sort_indices()
    Ac = Ac.tocsr()
    Ac.sort_indices()
    arrs = []
    for M in (R, Rt, Ac):
        arrs += [np.ascontiguousarray(M.indptr, dtype=np.int32), np.ascontiguousarray(M.indices, dtype=np.int32),
                 np.ascontiguousarray(M.data, dtype=np.float64)]
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.array(x0, dtype=np.float64, copy=True)
    hist = np.zeros(n_cycles + 1) if want_history else None
    keep = []
    n_amg = 0
    lv_arr = None
    if amg_levels:
        import scipy.sparse as sp
        n_amg = len(amg_levels)
        lv_arr = (_AmgLevel * n_amg)()
        for l, (A_l, P_l, cheb) in enumerate(amg_levels):
            lv_arr[l].A = _csr_struct(A_l, keep)
            if P_l is not None:
                lv_arr[l].P = _csr_struct(P_l, keep)
                lv_arr[l].Pt = _csr_struct(P_l.T, keep)
                lv_arr[l].degree, lv_arr[l].lmin, lv_arr[l].lmax = cheb
            else:
                empty = sp.csr_matrix((A_l.shape[0], 0))
                lv_arr[l].P = _csr_struct(empty, keep)
                lv_arr[l].Pt = _csr_struct(empty.T, keep)
        coarse_iters = 0
    lib.oracle_vcycles(_p(n_a), _p(h_a), _p(cd), _p(co), _p(cn), _p(dinv), C.c_int(degree), C.c_double(lmin),
                       C.c_double(lmax), C.c_int64(R.shape[0]), *[_p(a) for a in arrs], C.c_int(coarse_iters),
                       C.c_int(n_amg), lv_arr, _p(b), _p(x), C.c_int(n_cycles),
                       _p(hist) if want_history else None,
                       C.c_int((1 << 20) if amg_pre_smoothing_levels is None else int(amg_pre_smoothing_levels)))
    return x, hist
