"""Thin Python mirror of the mfmg classes over the C ABI (include/mfmg_hip.h).

Vectors are torch CUDA tensors (float64, contiguous) -- torch only owns the memory;
the arithmetic happens in libmfmg_hip.so on the stream of the Context."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import lib as _lib
from .lib import MeshDesc, check


def params_to_info(params: dict, indent: int = 0) -> str:
    """dict -> boost::property_tree INFO text (tests/data/hierarchy_input.info)."""
    out = []
    pad = " " * indent
    for k, v in params.items():
        key = f'"{k}"' if (" " in str(k) or ":" in str(k)) else str(k)
        if isinstance(v, dict):
            out.append(f"{pad}{key}\n{pad}{{\n{params_to_info(v, indent + 2)}{pad}}}\n")
        else:
            if isinstance(v, bool):
                v = "true" if v else "false"
            elif isinstance(v, float):
                v = repr(v)
            sval = str(v)
            if " " in sval or sval == "":
                sval = f'"{sval}"'
            out.append(f"{pad}{key} {sval}\n")
    return "".join(out)


def info_to_params(text: str) -> dict:
    """boost::property_tree INFO text -> nested dict (the inverse of params_to_info; what
    boost::property_tree::info_parser::read_info does for tests/hierarchy_driver.cc:255-256): `key value`,
    `key { children }` with the brace on the same or the next line, "quoted strings", `;` comments.  Values stay
    strings except true / false, integers and floats."""
    import shlex

    def convert(v: str):
        if v == "true":
            return True
        if v == "false":
            return False
        for cast in (int, float):
            try:
                return cast(v)
            except ValueError:
                pass
        return v

    tokens = []
    for line in text.splitlines():
        lex = shlex.shlex(line, posix=True)
        lex.whitespace_split = True
        lex.commenters = ";"
        lex.quotes = '"'
        toks = list(lex)
        # braces glued to a word do not occur in the files of the reference; split the plain cases anyway
        for t in toks:
            tokens.append(("tok", t))
        tokens.append(("eol", None))

    def parse(pos):
        node = {}
        while pos < len(tokens):
            kind, t = tokens[pos]
            if kind == "eol":
                pos += 1
                continue
            if t == "}":
                return node, pos + 1
            key = t
            pos += 1
            value = None
            if pos < len(tokens) and tokens[pos][0] == "tok" and tokens[pos][1] not in ("{", "}"):
                value = tokens[pos][1]
                pos += 1
            q = pos
            while q < len(tokens) and tokens[q][0] == "eol":
                q += 1
            if q < len(tokens) and tokens[q] == ("tok", "{"):
                child, pos = parse(q + 1)
                node[key] = child
            else:
                node[key] = convert(value) if value is not None else ""
        return node, pos

    return parse(0)[0]


def _dev_ptr(t: torch.Tensor, n: Optional[int] = None, dtype=torch.float64) -> int:
    if not isinstance(t, torch.Tensor):
        raise TypeError("expected a torch tensor")
    if t.device.type != "cuda":
        raise ValueError("vectors must live on the GPU (the HIP path has no CPU fallback)")
    if t.dtype != dtype or not t.is_contiguous():
        raise ValueError(f"vectors must be contiguous {dtype} tensors")
    if n is not None and t.numel() != n:
        raise ValueError(f"vector has {t.numel()} entries, expected {n}")
    return t.data_ptr()


def memory_inventory() -> str:
    """Live bytes of the library's device buffers per (setup section | kind of structure), as text."""
    lib = _lib.load()
    buf = C.create_string_buffer(1 << 16)
    check(lib.mfmg_hip_memory_inventory(buf, len(buf)))
    return buf.value.decode()


class Context:
    """CudaHandle twin: a HIP stream + scratch (source/cuda/cuda_handle.cu:17-56)."""

    def __init__(self, stream: Optional[int] = None, own_stream: bool = False):
        """By default the context runs on torch's current stream, so that tensor operations issued
        through torch and the library's kernels are ordered; `own_stream` asks the library for a
        private non-blocking stream (the caller then synchronises explicitly)."""
        self._lib = _lib.load()
        h = C.c_void_p()
        if own_stream:
            arg = C.c_void_p(-1)
        else:
            if stream is None and torch.cuda.is_available():
                stream = torch.cuda.current_stream().cuda_stream
            arg = C.c_void_p(stream) if stream else None
        check(self._lib.mfmg_hip_context_create(arg, C.byref(h)))
        self.handle = h

    def synchronize(self):
        check(self._lib.mfmg_hip_context_synchronize(self.handle))

    @property
    def stream(self) -> int:
        return self._lib.mfmg_hip_context_stream(self.handle)

    def torch_stream(self) -> "torch.cuda.ExternalStream":
        return torch.cuda.ExternalStream(self.stream)

    def set_cell_constant_layout(self, enable: bool):
        """Operators created afterwards keep one coefficient per cell when a cell's eight are equal (default on)."""
        check(self._lib.mfmg_hip_context_set_cell_constant_layout(self.handle, int(bool(enable))))

    def set_stored_diagonal(self, enable: bool):
        """Cell-constant operators created afterwards keep D^-1 in their chunk records (default: derived in the kernel)."""
        check(self._lib.mfmg_hip_context_set_stored_diagonal(self.handle, int(bool(enable))))

    def set_mf_fused_terms(self, n_terms: int):
        """Matrix-free operators created afterwards can run up to n_terms (1..3) smoother terms per sweep (default 3)."""
        check(self._lib.mfmg_hip_context_set_mf_fused_terms(self.handle, int(n_terms)))

    def set_mf_shell(self, mode):
        """Distributed fine operator: the shell of tiles 'beside' (default) / 'after' the interior tiles, or as 'slabs'."""
        check(self._lib.mfmg_hip_context_set_mf_shell(self.handle, {"beside": 0, "": 0, "after": 1, "slabs": 2}[mode]))

    def set_mf_emulate_split(self, axes):
        """One rank, measurement only: the launches of a rank with neighbours along 'z', 'yz' or 'xyz' ('' / None: off)."""
        check(self._lib.mfmg_hip_context_set_mf_emulate_split(self.handle, {None: 0, "": 0, "z": 1, "yz": 2, "xyz": 3, "1": 3}[axes]))

    def set_galerkin_on_device(self, enable: bool):
        """Hierarchies created afterwards form R A R^T of a matrix-free A by probing on the device (default) or on the host."""
        check(self._lib.mfmg_hip_context_set_galerkin_on_device(self.handle, int(bool(enable))))

    def set_overlap_exchange(self, enable: bool):
        """Distributed runs: overlap the fine-level halo exchange with interior operator tiles (default on)."""
        check(self._lib.mfmg_hip_context_set_overlap_exchange(self.handle, int(bool(enable))))

    def profile_enable(self, enabled: bool = True, only: str = ""):
        """HIP-event timing per kernel name; `only` restricts it to one name (every timed launch costs two
        event records on the stream)."""
        check(self._lib.mfmg_hip_profile_select(self.handle, only.encode() if only else None))
        check(self._lib.mfmg_hip_profile_enable(self.handle, 1 if enabled else 0))

    def profile_query(self, kernel: str):
        """(launches, total milliseconds, summed algorithmic bytes) of one kernel since profile_enable."""
        n, ms, by = C.c_int64(), C.c_double(), C.c_double()
        check(self._lib.mfmg_hip_profile_query(self.handle, kernel.encode(), C.byref(n), C.byref(ms), C.byref(by)))
        return n.value, ms.value, by.value

    def dot(self, x: torch.Tensor, y: torch.Tensor) -> float:
        r = C.c_double()
        check(self._lib.mfmg_hip_vector_dot(self.handle, x.numel(), _dev_ptr(x), _dev_ptr(y), C.byref(r)))
        return r.value

    def l2_norm(self, x: torch.Tensor) -> float:
        r = C.c_double()
        check(self._lib.mfmg_hip_vector_l2_norm(self.handle, x.numel(), _dev_ptr(x), C.byref(r)))
        return r.value

    def set(self, x: torch.Tensor, value: float):
        check(self._lib.mfmg_hip_vector_set(self.handle, x.numel(), value, _dev_ptr(x)))

    def add(self, x: torch.Tensor, a: float, v: torch.Tensor):
        check(self._lib.mfmg_hip_vector_add(self.handle, x.numel(), a, _dev_ptr(v, x.numel()), _dev_ptr(x)))

    def sadd(self, x: torch.Tensor, s: float, a: float, v: torch.Tensor):
        check(self._lib.mfmg_hip_vector_sadd(self.handle, x.numel(), s, a, _dev_ptr(v, x.numel()), _dev_ptr(x)))

    def cell_contraction(self, u: torch.Tensor, c: torch.Tensor, v: torch.Tensor, cell_size, variant: str = "mfma"):
        """v[m, cell] = c[cell] * sum_k K_ref[m, k] u[k, cell] (BASELINE.json configs[4]); u, v: [8, n] contiguous CUDA
        tensors, float32 or float64; variant "valu" or "mfma"."""
        n = c.numel()
        dt = u.dtype
        assert dt in (torch.float32, torch.float64) and u.shape == (8, n) and v.shape == (8, n)
        hs = (C.c_double * 3)(*[float(x) for x in cell_size])
        check(self._lib.mfmg_hip_cell_contraction(self.handle, 1 if dt == torch.float32 else 0, 1 if variant == "mfma" else 0, n,
                                                  _dev_ptr(u, 8 * n, dt), _dev_ptr(c, n, dt), _dev_ptr(v, 8 * n, dt), hs))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self._lib.mfmg_hip_context_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def _csr_arrays(a):
    """scipy.sparse matrix (or (row_ptr, col, val, shape)) -> int32/int32/float64 numpy arrays."""
    if hasattr(a, "tocsr"):
        a = a.tocsr()
        a.sort_indices()
        return (np.ascontiguousarray(a.indptr, dtype=np.int32), np.ascontiguousarray(a.indices, dtype=np.int32),
                np.ascontiguousarray(a.data, dtype=np.float64), a.shape)
    rp, cl, vl, shape = a
    return (np.ascontiguousarray(rp, dtype=np.int32), np.ascontiguousarray(cl, dtype=np.int32),
            np.ascontiguousarray(vl, dtype=np.float64), shape)


class SparseMatrixDevice:
    """SparseMatrixDevice<double> + CudaMatrixOperator (include/mfmg/cuda/sparse_matrix_device.cuh,
    source/cuda/cuda_matrix_operator.cu)."""

    def __init__(self, ctx: Context, matrix=None, _handle=None, _borrowed=False, _keepalive=None):
        self._lib = _lib.load()
        self.ctx = ctx
        self._borrowed = _borrowed
        self._keepalive = _keepalive
        if _handle is not None:
            self.handle = _handle
        else:
            rp, cl, vl, shape = _csr_arrays(matrix)
            h = C.c_void_p()
            check(self._lib.mfmg_hip_csr_create(ctx.handle, shape[0], shape[1], len(vl), rp.ctypes.data,
                                                cl.ctypes.data, vl.ctypes.data, C.byref(h)))
            self.handle = h

    @property
    def shape(self):
        m, n, z = C.c_int64(), C.c_int64(), C.c_int64()
        check(self._lib.mfmg_hip_csr_shape(self.handle, C.byref(m), C.byref(n), C.byref(z)))
        return (m.value, n.value)

    @property
    def nnz(self):
        z = C.c_int64()
        check(self._lib.mfmg_hip_csr_shape(self.handle, None, None, C.byref(z)))
        return z.value

    def set_kernel(self, lanes_per_row: int = 0, use_lds: int = -1):
        check(self._lib.mfmg_hip_csr_set_kernel(self.handle, lanes_per_row, use_lds))

    def regular_rows(self) -> bool:
        v = C.c_int()
        check(self._lib.mfmg_hip_csr_regular_rows(self.handle, C.byref(v)))
        return bool(v.value)

    def stencil_classes(self):
        """(classes of non-regular nodes sharing a stencil, rows left to the stored values)."""
        k, r = C.c_int(), C.c_int64()
        check(self._lib.mfmg_hip_csr_stencil_classes(self.handle, C.byref(k), C.byref(r)))
        return k.value, r.value

    def solve(self, params, b, x):
        """CudaSolver(handle, op, params)->apply(b, x): the coarse solver `solver.type` names, built for this matrix,
        applied once from a zero guess."""
        info = params if isinstance(params, str) else params_to_info(params)
        n = self.shape[0]
        check(self._lib.mfmg_hip_csr_solve(self.handle, info.encode(), _dev_ptr(b, n), _dev_ptr(x, n)))

    def float_storage(self) -> bool:
        """The values the kernels read are kept in float (all representable in it: lossless)."""
        v = C.c_int()
        check(self._lib.mfmg_hip_csr_float_storage(self.handle, C.byref(v)))
        return bool(v.value)

    def set_regular_rows(self, enable: bool):
        check(self._lib.mfmg_hip_csr_set_regular_rows(self.handle, int(bool(enable))))

    def get_kernel(self):
        a, b = C.c_int(), C.c_int()
        check(self._lib.mfmg_hip_csr_get_kernel(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def vmult(self, dst: torch.Tensor, src: torch.Tensor):
        m, n = self.shape
        check(self._lib.mfmg_hip_csr_vmult(self.handle, _dev_ptr(src, n), _dev_ptr(dst, m)))

    def apply(self, x: torch.Tensor, y: torch.Tensor, mode: int = _lib.NO_TRANS):
        m, n = self.shape
        nx, ny = (n, m) if mode == _lib.NO_TRANS else (m, n)
        check(self._lib.mfmg_hip_csr_apply(self.handle, _dev_ptr(x, nx), _dev_ptr(y, ny), mode))

    def transpose(self) -> "SparseMatrixDevice":
        h = C.c_void_p()
        check(self._lib.mfmg_hip_csr_transpose(self.handle, C.byref(h)))
        return SparseMatrixDevice(self.ctx, _handle=h)

    def multiply(self, b: "SparseMatrixDevice") -> "SparseMatrixDevice":
        h = C.c_void_p()
        check(self._lib.mfmg_hip_csr_multiply(self.handle, b.handle, C.byref(h)))
        return SparseMatrixDevice(self.ctx, _handle=h)

    def inverse_diagonal(self, dinv: torch.Tensor):
        check(self._lib.mfmg_hip_csr_inverse_diagonal(self.handle, _dev_ptr(dinv, self.shape[0])))

    def residual(self, x, b, res):
        m, n = self.shape
        check(self._lib.mfmg_hip_csr_residual(self.handle, _dev_ptr(x, n), _dev_ptr(b, m), _dev_ptr(res, m)))

    def smoother_step(self, dinv, b, x, x_prev, alpha, beta, out):
        n = self.shape[0]
        check(self._lib.mfmg_hip_csr_smoother_step(self.handle, _dev_ptr(dinv, n), _dev_ptr(b, n), _dev_ptr(x, n),
                                                   _dev_ptr(x_prev, n) if x_prev is not None else None,
                                                   alpha, beta, _dev_ptr(out, n)))

    def to_scipy(self):
        import scipy.sparse as sp
        m, n = self.shape
        nnz = self.nnz
        rp = np.empty(m + 1, dtype=np.int32)
        cl = np.empty(nnz, dtype=np.int32)
        vl = np.empty(nnz, dtype=np.float64)
        check(self._lib.mfmg_hip_csr_download(self.handle, rp.ctypes.data, cl.ctypes.data, vl.ctypes.data))
        return sp.csr_matrix((vl, cl, rp), shape=(m, n))

    def __del__(self):
        try:
            if getattr(self, "handle", None) and not self._borrowed:
                self._lib.mfmg_hip_csr_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class MatrixFreeLaplace:
    """CudaMatrixFreeOperator + LaplaceOperator (source/cuda/cuda_matrix_free_operator.cu,
    tests/laplace_matrix_free.hpp:121-156) on the GPU."""

    def __init__(self, ctx: Context, problem):
        self._lib = _lib.load()
        self.ctx = ctx
        self.problem = problem  # keeps the mesh arrays alive during construction
        desc = problem.mesh_desc()
        h = C.c_void_p()
        check(self._lib.mfmg_hip_mf_laplace_create(ctx.handle, C.byref(desc), C.byref(h)))
        self.handle = h
        self.n_dofs = problem.n_dofs

    def vmult(self, dst: torch.Tensor, src: torch.Tensor):
        check(self._lib.mfmg_hip_mf_laplace_vmult(self.handle, _dev_ptr(src, self.n_dofs), _dev_ptr(dst, self.n_dofs)))

    def residual(self, x, b, res):
        n = self.n_dofs
        check(self._lib.mfmg_hip_mf_laplace_residual(self.handle, _dev_ptr(x, n), _dev_ptr(b, n), _dev_ptr(res, n)))

    def smoother_step(self, b, x, x_prev, alpha, beta, out):
        n = self.n_dofs
        check(self._lib.mfmg_hip_mf_laplace_smoother_step(self.handle, _dev_ptr(b, n), _dev_ptr(x, n),
                                                          _dev_ptr(x_prev, n) if x_prev is not None else None,
                                                          alpha, beta, _dev_ptr(out, n)))

    def sweep_available(self, n_terms: int) -> bool:
        v = C.c_int()
        check(self._lib.mfmg_hip_mf_laplace_sweep_available(self.handle, int(n_terms), C.byref(v)))
        return bool(v.value)

    def smoother_sweep(self, alpha, beta, b, x, out, out_prev=None):
        """len(alpha) smoother terms in one sweep; out = the last iterate, out_prev (optional) the one before.
        x = None: from the zero vector, which is not read (three terms, the default arithmetic; raises where unavailable)."""
        n, k = self.n_dofs, len(alpha)
        a = (C.c_double * k)(*[float(v) for v in alpha])
        be = (C.c_double * k)(*[float(v) for v in beta])
        check(self._lib.mfmg_hip_mf_laplace_smoother_sweep(self.handle, k, a, be, _dev_ptr(b, n), _dev_ptr(x, n) if x is not None else None, _dev_ptr(out, n),
                                                           _dev_ptr(out_prev, n) if out_prev is not None else None))

    def set_sweep_tile(self, waves: int, ty: int, tz: int):
        check(self._lib.mfmg_hip_mf_laplace_set_sweep_tile(self.handle, waves, ty, tz))

    def set_sweep_reference(self, on: bool):
        """The sweep's cell kernel: mode space (default) or the arithmetic of the one-term kernel, bit for bit."""
        check(self._lib.mfmg_hip_mf_laplace_set_sweep_reference(self.handle, int(bool(on))))

    def get_sweep_tile(self, n_terms: int):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        check(self._lib.mfmg_hip_mf_laplace_get_sweep_tile(self.handle, int(n_terms), C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def diagonal_inverse(self) -> torch.Tensor:
        out = torch.empty(self.n_dofs, dtype=torch.float64, device="cuda")
        check(self._lib.mfmg_hip_mf_laplace_diagonal_inverse(self.handle, _dev_ptr(out)))
        self.ctx.synchronize()
        return out

    def diagonal(self) -> torch.Tensor:
        out = torch.empty(self.n_dofs, dtype=torch.float64, device="cuda")
        check(self._lib.mfmg_hip_mf_laplace_diagonal(self.handle, _dev_ptr(out)))
        self.ctx.synchronize()
        return out

    def cell_constant_layout(self) -> bool:
        v = C.c_int()
        check(self._lib.mfmg_hip_mf_laplace_cell_constant_layout(self.handle, C.byref(v)))
        return bool(v.value)

    def diagonal_in_record(self) -> bool:
        v = C.c_int()
        check(self._lib.mfmg_hip_mf_laplace_diagonal_in_record(self.handle, C.byref(v)))
        return bool(v.value)

    def ids_computed(self) -> bool:
        """The kernel computes the DoF ids from the position instead of reading them from its records."""
        v = C.c_int()
        check(self._lib.mfmg_hip_mf_laplace_ids_computed(self.handle, C.byref(v)))
        return bool(v.value)

    def get_tile(self):
        """(waves, ty, tz) of the next launch."""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        check(self._lib.mfmg_hip_mf_laplace_get_tile(self.handle, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def set_tile(self, ty: int, tz: int, waves: int = None):
        check(self._lib.mfmg_hip_mf_laplace_set_tile(self.handle, ty, tz))
        if waves is not None:
            check(self._lib.mfmg_hip_mf_laplace_set_tile_waves(self.handle, waves))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self._lib.mfmg_hip_mf_laplace_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class MatrixFreeLaplaceF32:
    """FP32 instance of the matrix-free operator (BASELINE.json configs[4]); vectors are float32 tensors."""

    def __init__(self, ctx: Context, problem):
        self._lib = _lib.load()
        self.ctx = ctx
        desc = problem.mesh_desc()
        h = C.c_void_p()
        check(self._lib.mfmg_hip_mf_laplace_f32_create(ctx.handle, C.byref(desc), C.byref(h)))
        self.handle = h
        self.n_dofs = problem.n_dofs

    def _p(self, t):
        return _dev_ptr(t, self.n_dofs, torch.float32)

    def vmult(self, dst, src):
        check(self._lib.mfmg_hip_mf_laplace_f32_vmult(self.handle, self._p(src), self._p(dst)))

    def residual(self, x, b, res):
        check(self._lib.mfmg_hip_mf_laplace_f32_residual(self.handle, self._p(x), self._p(b), self._p(res)))

    def smoother_step(self, b, x, x_prev, alpha, beta, out):
        check(self._lib.mfmg_hip_mf_laplace_f32_smoother_step(self.handle, self._p(b), self._p(x),
                                                              self._p(x_prev) if x_prev is not None else None,
                                                              alpha, beta, self._p(out)))

    def sweep_available(self, n_terms: int) -> bool:
        v = C.c_int()
        check(self._lib.mfmg_hip_mf_laplace_f32_sweep_available(self.handle, int(n_terms), C.byref(v)))
        return bool(v.value)

    def smoother_sweep(self, alpha, beta, b, x, out, out_prev=None):
        k = len(alpha)
        a = (C.c_float * k)(*[float(v) for v in alpha])
        be = (C.c_float * k)(*[float(v) for v in beta])
        check(self._lib.mfmg_hip_mf_laplace_f32_smoother_sweep(self.handle, k, a, be, self._p(b), self._p(x), self._p(out),
                                                               self._p(out_prev) if out_prev is not None else None))

    def set_sweep_reference(self, on: bool):
        check(self._lib.mfmg_hip_mf_laplace_f32_set_sweep_reference(self.handle, int(bool(on))))

    def cell_constant_layout(self) -> bool:
        v = C.c_int()
        check(self._lib.mfmg_hip_mf_laplace_f32_cell_constant_layout(self.handle, C.byref(v)))
        return bool(v.value)

    def ids_computed(self) -> bool:
        v = C.c_int()
        check(self._lib.mfmg_hip_mf_laplace_f32_ids_computed(self.handle, C.byref(v)))
        return bool(v.value)

    def diagonal_inverse(self) -> torch.Tensor:
        out = torch.empty(self.n_dofs, dtype=torch.float32, device="cuda")
        check(self._lib.mfmg_hip_mf_laplace_f32_diagonal_inverse(self.handle, self._p(out)))
        self.ctx.synchronize()
        return out

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self._lib.mfmg_hip_mf_laplace_f32_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class Hierarchy:
    """mfmg::Hierarchy<VectorType> (include/mfmg/common/hierarchy.hpp:155-373)."""

    def __init__(self, ctx: Context, evaluator_type: str, problem, params: dict | str):
        self._lib = _lib.load()
        self.ctx = ctx
        info = params if isinstance(params, str) else params_to_info(params)
        desc = problem.mesh_desc()
        h = C.c_void_p()
        check(self._lib.mfmg_hip_hierarchy_create(ctx.handle, evaluator_type.encode(), C.byref(desc),
                                                  info.encode(), C.byref(h)))
        self.handle = h

    @property
    def n_levels(self) -> int:
        n = C.c_int32()
        check(self._lib.mfmg_hip_hierarchy_n_levels(self.handle, C.byref(n)))
        return n.value

    def level_size(self, level: int) -> int:
        n = C.c_int64()
        check(self._lib.mfmg_hip_hierarchy_level_size(self.handle, level, C.byref(n)))
        return n.value

    def apply(self, b: torch.Tensor, x: torch.Tensor):
        n = self.level_size(0)
        check(self._lib.mfmg_hip_hierarchy_apply(self.handle, _dev_ptr(b, n), _dev_ptr(x, n)))

    def apply_f32(self, b: torch.Tensor, x: torch.Tensor):
        """The same cycle on float32 vectors with the fine level in FP32 (parameter "fine level precision": "float")."""
        n = self.level_size(0)
        check(self._lib.mfmg_hip_hierarchy_apply_f32(self.handle, _dev_ptr(b, n, torch.float32), _dev_ptr(x, n, torch.float32)))

    def vmult(self, x: torch.Tensor, b: torch.Tensor):
        n = self.level_size(0)
        check(self._lib.mfmg_hip_hierarchy_vmult(self.handle, _dev_ptr(x, n), _dev_ptr(b, n)))

    def operator_apply(self, level: int, x, y, mode: int = _lib.NO_TRANS):
        n = self.level_size(level)
        check(self._lib.mfmg_hip_hierarchy_operator_apply(self.handle, level, _dev_ptr(x, n), _dev_ptr(y, n), mode))

    def smoother_apply(self, level: int, b, x):
        n = self.level_size(level)
        check(self._lib.mfmg_hip_hierarchy_smoother_apply(self.handle, level, _dev_ptr(b, n), _dev_ptr(x, n)))

    def restrictor_apply(self, level: int, vin, vout, mode: int = _lib.NO_TRANS):
        nf, nc = self.level_size(level - 1), self.level_size(level)
        ni, no = (nf, nc) if mode == _lib.NO_TRANS else (nc, nf)
        check(self._lib.mfmg_hip_hierarchy_restrictor_apply(self.handle, level, _dev_ptr(vin, ni), _dev_ptr(vout, no), mode))

    def ap_apply(self, level: int, vin, vout):
        """vout = (A R^T) vin for the A R^T of `level` (hierarchies built with keep_ap = true)."""
        nf, nc = self.level_size(level - 1), self.level_size(level)
        check(self._lib.mfmg_hip_hierarchy_ap_apply(self.handle, level, _dev_ptr(vin, nc), _dev_ptr(vout, nf)))

    def coarse_apply(self, b, x):
        n = self.level_size(self.n_levels - 1)
        check(self._lib.mfmg_hip_hierarchy_coarse_apply(self.handle, _dev_ptr(b, n), _dev_ptr(x, n)))

    def set_restrictor(self, matrix):
        rp, cl, vl, shape = _csr_arrays(matrix)
        check(self._lib.mfmg_hip_hierarchy_set_restrictor(self.handle, shape[0], shape[1], len(vl), rp.ctypes.data,
                                                          cl.ctypes.data, vl.ctypes.data))

    def restrictor(self) -> SparseMatrixDevice:
        h = C.c_void_p()
        check(self._lib.mfmg_hip_hierarchy_get_restrictor(self.handle, C.byref(h)))
        return SparseMatrixDevice(self.ctx, _handle=h, _borrowed=True, _keepalive=self)

    def fine_operator(self) -> SparseMatrixDevice:
        """The assembled operator of level 0 (HipMeshEvaluator hierarchies)."""
        h = C.c_void_p()
        check(self._lib.mfmg_hip_hierarchy_get_fine_operator(self.handle, C.byref(h)))
        return SparseMatrixDevice(self.ctx, _handle=h, _borrowed=True, _keepalive=self)

    def coarse_operator(self) -> SparseMatrixDevice:
        h = C.c_void_p()
        check(self._lib.mfmg_hip_hierarchy_get_coarse_operator(self.handle, C.byref(h)))
        return SparseMatrixDevice(self.ctx, _handle=h, _borrowed=True, _keepalive=self)

    def coarse_amg_levels(self):
        """[(A_l, P_l or None, (degree, lambda_min, lambda_max) or None)] of the multilevel coarse solver."""
        n = C.c_int32()
        check(self._lib.mfmg_hip_hierarchy_coarse_amg_levels(self.handle, C.byref(n)))
        out = []
        for l in range(n.value):
            h = C.c_void_p()
            check(self._lib.mfmg_hip_hierarchy_coarse_amg_get(self.handle, l, 0, C.byref(h)))
            A = SparseMatrixDevice(self.ctx, _handle=h, _borrowed=True, _keepalive=self).to_scipy()
            P = cheb = None
            if l + 1 < n.value:
                check(self._lib.mfmg_hip_hierarchy_coarse_amg_get(self.handle, l, 1, C.byref(h)))
                P = SparseMatrixDevice(self.ctx, _handle=h, _borrowed=True, _keepalive=self).to_scipy()
                d, lo, hi = C.c_int32(), C.c_double(), C.c_double()
                check(self._lib.mfmg_hip_hierarchy_coarse_amg_smoother(self.handle, l, C.byref(d), C.byref(lo), C.byref(hi)))
                cheb = (d.value, lo.value, hi.value)
            out.append((A, P, cheb))
        return out

    def coarse_amg_kernels(self, regular_rows=None):
        """[(level, which, rows, kernel kind, stencil classes, listed rows)] of the matrices of the multilevel coarse
        solver (which: 0 A_l, 1 P_l, 2 its stored transpose); regular_rows True / False switches the table-driven paths
        of all of them on / off first (tests: both must give the same cycle to rounding)."""
        n = C.c_int32()
        check(self._lib.mfmg_hip_hierarchy_coarse_amg_levels(self.handle, C.byref(n)))
        out = []
        for l in range(n.value):
            for which in (0, 1, 2):
                if which > 0 and l + 1 == n.value:
                    continue
                h = C.c_void_p()                              # (one borrowed view at a time)
                check(self._lib.mfmg_hip_hierarchy_coarse_amg_get(self.handle, l, which, C.byref(h)))
                m = SparseMatrixDevice(self.ctx, _handle=h, _borrowed=True, _keepalive=self)
                if regular_rows is not None:
                    m.set_regular_rows(regular_rows)
                out.append((l, which, m.shape[0], m.get_kernel()[1]) + tuple(m.stencil_classes()))
        return out

    def restrict_residual(self, x: torch.Tensor, b: torch.Tensor, b_coarse: torch.Tensor, level: int = 1):
        """b_coarse = R (A x - b) as the cycle computes it (hierarchy.hpp:281-290): one kernel where the rows of R A
        repeat themselves, the fused residual followed by the restriction otherwise."""
        nf, nc = self.level_size(level - 1), self.level_size(level)
        check(self._lib.mfmg_hip_hierarchy_restrict_residual(self.handle, level, _dev_ptr(x, nf), _dev_ptr(b, nf),
                                                             _dev_ptr(b_coarse, nc)))

    def residual_restriction_classes(self, level: int = 1) -> int:
        """Agglomerate classes of the one-pass residual restriction; 0 when the two-step form is in use."""
        n = C.c_int32()
        check(self._lib.mfmg_hip_hierarchy_residual_restriction_classes(self.handle, level, C.byref(n)))
        return n.value

    def coarse_amg_gather_level(self) -> int:
        """Index of the first aggregation level that is gathered and solved redundantly on every rank (-1: one rank)."""
        n = C.c_int32()
        check(self._lib.mfmg_hip_hierarchy_coarse_amg_gather_level(self.handle, C.byref(n)))
        return n.value

    def coarse_amg_gather_rows(self) -> int:
        """Global rows of that level (0 on one rank)."""
        l = self.coarse_amg_gather_level()
        return self.coarse_amg_shapes()[l][0] if l >= 0 else 0

    def coarse_amg_shapes(self):
        """[(rows, nnz(A_l), nnz(P_l) or 0)] of the multilevel coarse solver (no download)."""
        n = C.c_int32()
        check(self._lib.mfmg_hip_hierarchy_coarse_amg_levels(self.handle, C.byref(n)))
        out = []
        for l in range(n.value):
            h = C.c_void_p()
            check(self._lib.mfmg_hip_hierarchy_coarse_amg_get(self.handle, l, 0, C.byref(h)))
            A = SparseMatrixDevice(self.ctx, _handle=h, _borrowed=True, _keepalive=self)
            rows, an, pn = A.shape[0], A.nnz, 0
            if l + 1 < n.value:
                hp = C.c_void_p()
                check(self._lib.mfmg_hip_hierarchy_coarse_amg_get(self.handle, l, 1, C.byref(hp)))
                pn = SparseMatrixDevice(self.ctx, _handle=hp, _borrowed=True, _keepalive=self).nnz
            out.append((rows, an, pn))
        return out

    def smoother_sweep_terms(self):
        """(terms per sweep of the in-place smoother apply, of the out-of-place one); 0 = one launch per term."""
        a, b = C.c_int(), C.c_int()
        check(self._lib.mfmg_hip_hierarchy_smoother_sweep_terms(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def smoother_info(self):
        d, lo, hi = C.c_int32(), C.c_double(), C.c_double()
        check(self._lib.mfmg_hip_hierarchy_smoother_info(self.handle, C.byref(d), C.byref(lo), C.byref(hi)))
        return d.value, lo.value, hi.value

    def solve_cg(self, b, x, tolerance: float = 1e-6, max_iterations: int = 1000):
        """tests/hierarchy_driver.cc:103-116: CG on the fine operator preconditioned by this hierarchy
        (build it with "is preconditioner" true).  Returns (iterations, residual history)."""
        import numpy as np
        n = self.level_size(0)
        it, res = C.c_int32(), C.c_double()
        hist = np.zeros(max_iterations + 1)
        check(self._lib.mfmg_hip_hierarchy_solve_cg(self.handle, _dev_ptr(b, n), _dev_ptr(x, n), tolerance, max_iterations,
                                                    C.byref(it), C.byref(res), hist.ctypes.data_as(C.POINTER(C.c_double)),
                                                    len(hist)))
        return it.value, hist[: it.value + 1]

    def operator_tile(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        check(self._lib.mfmg_hip_hierarchy_operator_tile(self.handle, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def sweep_tile(self, n_terms: int):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        check(self._lib.mfmg_hip_hierarchy_sweep_tile(self.handle, int(n_terms), C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def set_sweep_tile(self, waves: int, ty: int, tz: int):
        check(self._lib.mfmg_hip_hierarchy_set_sweep_tile(self.handle, waves, ty, tz))

    def set_operator_tile(self, waves: int, ty: int, tz: int):
        check(self._lib.mfmg_hip_hierarchy_set_operator_tile(self.handle, waves, ty, tz))

    def timer_report(self) -> str:
        buf = C.create_string_buffer(8192)
        check(self._lib.mfmg_hip_hierarchy_timer_report(self.handle, buf, len(buf)))
        return buf.value.decode()

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self._lib.mfmg_hip_hierarchy_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


# ---- host-side setup pieces (no GPU) ------------------------------------------------------
def _host_csr_to_scipy(lib, h):
    import scipy.sparse as sp
    m, n, z = C.c_int64(), C.c_int64(), C.c_int64()
    check(lib.mfmg_hip_host_csr_shape(h, C.byref(m), C.byref(n), C.byref(z)))
    rp = np.empty(m.value + 1, dtype=np.int32)
    cl = np.empty(z.value, dtype=np.int32)
    vl = np.empty(z.value, dtype=np.float64)
    check(lib.mfmg_hip_host_csr_get(h, rp.ctypes.data, cl.ctypes.data, vl.ctypes.data))
    lib.mfmg_hip_host_csr_destroy(h)
    return sp.csr_matrix((vl, cl, rp), shape=(m.value, n.value))


def host_assemble_matrix(problem, semantics: str = "assembled"):
    lib = _lib.load()
    assert problem.device.type == "cpu"
    desc = problem.mesh_desc()
    h = C.c_void_p()
    check(lib.mfmg_hip_host_assemble_matrix(C.byref(desc), 0 if semantics == "assembled" else 1, C.byref(h)))
    return _host_csr_to_scipy(lib, h)


def host_build_restrictor(problem, params: dict | str, matrix_free: bool):
    lib = _lib.load()
    assert problem.device.type == "cpu"
    desc = problem.mesh_desc()
    info = params if isinstance(params, str) else params_to_info(params)
    h = C.c_void_p()
    check(lib.mfmg_hip_host_build_restrictor(C.byref(desc), info.encode(), 1 if matrix_free else 0, C.byref(h)))
    return _host_csr_to_scipy(lib, h)


def host_galerkin(problem, R, semantics: str = "assembled"):
    lib = _lib.load()
    assert problem.device.type == "cpu"
    desc = problem.mesh_desc()
    rp, cl, vl, shape = _csr_arrays(R)
    h = C.c_void_p()
    check(lib.mfmg_hip_host_galerkin(C.byref(desc), 0 if semantics == "assembled" else 1, shape[0], len(vl),
                                     rp.ctypes.data, cl.ctypes.data, vl.ctypes.data, C.byref(h)))
    return _host_csr_to_scipy(lib, h)


def host_amg_build(A, near_null=None, params: dict | str = "", grid_dims=None, node_of_row=None,
                   component_of_row=None):
    """Smoothed-aggregation hierarchy of the multilevel coarse solver on the host: [(A_l, P_l or None)]."""
    lib = _lib.load()
    rp, cl, vl, shape = _csr_arrays(A)
    info = params if isinstance(params, str) else params_to_info(params)
    nn = None if near_null is None else np.ascontiguousarray(near_null, dtype=np.float64)
    gd = None if grid_dims is None else np.ascontiguousarray(grid_dims, dtype=np.int32)
    nr = None if node_of_row is None else np.ascontiguousarray(node_of_row, dtype=np.int32)
    cr = None if component_of_row is None else np.ascontiguousarray(component_of_row, dtype=np.int32)
    h = C.c_void_p()
    check(lib.mfmg_hip_host_amg_build(shape[0], len(vl), rp.ctypes.data, cl.ctypes.data, vl.ctypes.data,
                                      nn.ctypes.data if nn is not None else None,
                                      gd.ctypes.data if gd is not None else None,
                                      nr.ctypes.data if nr is not None else None,
                                      cr.ctypes.data if cr is not None else None, info.encode(), C.byref(h)))
    n = C.c_int32()
    check(lib.mfmg_hip_host_amg_n_levels(h, C.byref(n)))
    out = []
    for l in range(n.value):
        m = C.c_void_p()
        check(lib.mfmg_hip_host_amg_get(h, l, 0, C.byref(m)))
        Al = _host_csr_to_scipy(lib, m)
        Pl = None
        if l + 1 < n.value:
            check(lib.mfmg_hip_host_amg_get(h, l, 1, C.byref(m)))
            Pl = _host_csr_to_scipy(lib, m)
        out.append((Al, Pl))
    lib.mfmg_hip_host_amg_destroy(h)
    return out


def params_get(info: str, path: str) -> str:
    lib = _lib.load()
    buf = C.create_string_buffer(1024)
    check(lib.mfmg_hip_host_params_get(info.encode(), path.encode(), buf, len(buf)))
    return buf.value.decode()
