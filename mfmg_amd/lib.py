"""ctypes binding of libmfmg_hip.so (the C ABI declared in include/mfmg_hip.h).

The shared library is the product; this module only loads it.  There is no
Python / CPU fallback: if the library is missing the import fails loudly."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MFMG_HIP_LIBRARY: another build of the same library, e.g. a measurement build of scratch/r04_fp32_ablation.sh; never a fallback)
LIB_PATH = os.environ.get("MFMG_HIP_LIBRARY") or os.path.join(_HERE, "libmfmg_hip.so")

SUCCESS = 0
ERROR_RUNTIME = 1
ERROR_NOT_IMPLEMENTED = 2
ERROR_INVALID_ARGUMENT = 3
ERROR_DEVICE = 4
NO_TRANS = 0
TRANS = 1


class MfmgError(RuntimeError):
    """std::runtime_error thrown by ASSERT_THROW (include/mfmg/common/exceptions.hpp:48-52)."""


class MfmgNotImplementedError(NotImplementedError):
    """NotImplementedExc (include/mfmg/common/exceptions.hpp:54-73)."""


class MfmgInvalidArgument(ValueError):
    pass


class MfmgDeviceError(RuntimeError):
    pass


class MeshDesc(C.Structure):
    """mfmg_hip_mesh_desc"""
    _fields_ = [
        ("dim", C.c_int32),
        ("n_cells", C.c_int32 * 3),
        ("cell_size", C.c_double * 3),
        ("n_dofs", C.c_int64),
        ("cell_dofs", C.c_void_p),
        ("coefficient", C.c_void_p),
        ("constrained", C.c_void_p),
        ("arrays_on_device", C.c_int32),
    ]


def build_library(verbose: bool = False) -> str:
    """Compile libmfmg_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libmfmg_hip.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
    if verbose:
        print(res.stdout[-2000:])
    return LIB_PATH


_lib = None


def _declared_abi_version():
    import re
    header = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mfmg_hip.h")
    try:
        m = re.search(r"#define\s+MFMG_HIP_ABI_VERSION\s+(\d+)", open(header).read())
    except OSError:
        return None
    return int(m.group(1)) if m else None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the mfmg HIP path has no Python/CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    # a library built from another revision of include/mfmg_hip.h would be called with the wrong callback layouts
    declared = _declared_abi_version()
    try:
        lib.mfmg_hip_abi_version.restype = C.c_int
        built = int(lib.mfmg_hip_abi_version())
    except AttributeError:
        built = None
    if declared is not None and built != declared:
        raise ImportError(
            f"{LIB_PATH} was built with ABI version {built}, include/mfmg_hip.h declares {declared}: rebuild it with "
            "`python -c 'import __graft_entry__ as g; g.build()'`")
    vp, i32, i64, dbl, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_size_t
    P = C.POINTER
    sig = {
        "mfmg_hip_last_error": (C.c_char_p, []),
        "mfmg_hip_version": (C.c_char_p, []),
        "mfmg_hip_context_create": (C.c_int, [vp, P(vp)]),
        "mfmg_hip_context_destroy": (C.c_int, [vp]),
        "mfmg_hip_context_synchronize": (C.c_int, [vp]),
        "mfmg_hip_context_stream": (vp, [vp]),
        "mfmg_hip_context_set_communicator": (C.c_int, [vp, i32, i32, i32, i32]),
        "mfmg_hip_context_set_communicator_box": (C.c_int, [vp, i32, P(i32), P(i32), P(i32)]),
        "mfmg_hip_context_exchange_volume": (C.c_int, [vp, P(i64), P(i64)]),
        "mfmg_hip_context_halo_box": (C.c_int, [vp, i32, vp]),
        "mfmg_hip_rccl_unique_id": (C.c_int, [vp]),
        "mfmg_hip_rccl_available": (C.c_int, []),
        "mfmg_hip_abi_version": (C.c_int, []),
        "mfmg_hip_memory_inventory": (C.c_int, [C.c_char_p, sz]),
        "mfmg_hip_context_use_rccl": (C.c_int, [vp, vp]),
        "mfmg_hip_context_use_host_transport": (C.c_int, [vp, vp, vp, vp, vp]),
        "mfmg_hip_context_use_reflecting_transport": (C.c_int, [vp]),
        "mfmg_hip_context_use_reflecting_transport_delay": (C.c_int, [vp, dbl]),
        "mfmg_hip_context_transport_loopback_time": (C.c_int, [vp, i64, C.c_int, P(dbl)]),
        "mfmg_hip_context_transport_name": (C.c_int, [vp, C.c_char_p, sz]),
        "mfmg_hip_context_transport_ranks": (C.c_int, [vp, P(C.c_int)]),
        "mfmg_hip_context_exchange_count": (C.c_int, [vp, P(i64)]),
        "mfmg_hip_context_transport_selftest": (C.c_int, [vp, i64, P(dbl)]),
        "mfmg_hip_context_exchange": (C.c_int, [vp, i32, vp, C.c_int]),
        "mfmg_hip_context_owned_dot": (C.c_int, [vp, i32, vp, vp, P(dbl)]),
        "mfmg_hip_context_halo_space": (C.c_int, [vp, i32, vp]),
        "mfmg_hip_context_set_overlap_exchange": (C.c_int, [vp, C.c_int]),
        "mfmg_hip_context_set_cell_constant_layout": (C.c_int, [vp, C.c_int]),
        "mfmg_hip_context_set_galerkin_on_device": (C.c_int, [vp, C.c_int]),
        "mfmg_hip_context_set_stored_diagonal": (C.c_int, [vp, C.c_int]),
        "mfmg_hip_mf_laplace_diagonal_in_record": (C.c_int, [vp, P(C.c_int)]),
        "mfmg_hip_context_set_mf_fused_terms": (C.c_int, [vp, C.c_int]),
        "mfmg_hip_context_set_low_ghost_cells": (C.c_int, [vp, C.c_int32]),
        "mfmg_hip_context_set_mf_shell": (C.c_int, [vp, C.c_int]),
        "mfmg_hip_context_set_mf_emulate_split": (C.c_int, [vp, C.c_int]),
        "mfmg_hip_mf_laplace_sweep_available": (C.c_int, [vp, C.c_int, P(C.c_int)]),
        "mfmg_hip_mf_laplace_smoother_sweep": (C.c_int, [vp, C.c_int, P(dbl), P(dbl), vp, vp, vp, vp]),
        "mfmg_hip_mf_laplace_set_sweep_tile": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
        "mfmg_hip_mf_laplace_set_sweep_reference": (C.c_int, [vp, C.c_int]),
        "mfmg_hip_mf_laplace_f32_set_sweep_reference": (C.c_int, [vp, C.c_int]),
        "mfmg_hip_mf_laplace_get_sweep_tile": (C.c_int, [vp, C.c_int, P(C.c_int), P(C.c_int), P(C.c_int)]),
        "mfmg_hip_mf_laplace_f32_sweep_available": (C.c_int, [vp, C.c_int, P(C.c_int)]),
        "mfmg_hip_mf_laplace_f32_smoother_sweep": (C.c_int, [vp, C.c_int, P(C.c_float), P(C.c_float), vp, vp, vp, vp]),
        "mfmg_hip_mf_laplace_ids_computed": (C.c_int, [vp, P(C.c_int)]),
        "mfmg_hip_mf_laplace_f32_ids_computed": (C.c_int, [vp, P(C.c_int)]),
        "mfmg_hip_mf_laplace_cell_constant_layout": (C.c_int, [vp, P(C.c_int)]),
        "mfmg_hip_context_halo_layout": (C.c_int, [vp, i32, P(i64), P(i64), P(i64), P(i64)]),
        "mfmg_hip_cell_contraction": (C.c_int, [vp, C.c_int, C.c_int, i64, vp, vp, vp, P(dbl)]),
        "mfmg_hip_profile_enable": (C.c_int, [vp, C.c_int]),
        "mfmg_hip_profile_select": (C.c_int, [vp, C.c_char_p]),
        "mfmg_hip_profile_query": (C.c_int, [vp, C.c_char_p, P(i64), P(dbl), P(dbl)]),
        "mfmg_hip_malloc": (C.c_int, [P(vp), sz]),
        "mfmg_hip_free": (C.c_int, [vp]),
        "mfmg_hip_copy_to_dev": (C.c_int, [vp, vp, sz]),
        "mfmg_hip_copy_to_host": (C.c_int, [vp, vp, sz]),
        "mfmg_hip_vector_set": (C.c_int, [vp, i64, dbl, vp]),
        "mfmg_hip_vector_add": (C.c_int, [vp, i64, dbl, vp, vp]),
        "mfmg_hip_vector_sadd": (C.c_int, [vp, i64, dbl, dbl, vp, vp]),
        "mfmg_hip_vector_dot": (C.c_int, [vp, i64, vp, vp, P(dbl)]),
        "mfmg_hip_vector_l2_norm": (C.c_int, [vp, i64, vp, P(dbl)]),
        "mfmg_hip_csr_create": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, P(vp)]),
        "mfmg_hip_csr_destroy": (C.c_int, [vp]),
        "mfmg_hip_csr_shape": (C.c_int, [vp, P(i64), P(i64), P(i64)]),
        "mfmg_hip_csr_set_kernel": (C.c_int, [vp, C.c_int, C.c_int]),
        "mfmg_hip_csr_get_kernel": (C.c_int, [vp, P(C.c_int), P(C.c_int)]),
        "mfmg_hip_csr_regular_rows": (C.c_int, [vp, P(C.c_int)]),
        "mfmg_hip_csr_set_regular_rows": (C.c_int, [vp, C.c_int]),
        "mfmg_hip_csr_stencil_classes": (C.c_int, [vp, P(C.c_int), P(i64)]),
        "mfmg_hip_csr_float_storage": (C.c_int, [vp, P(C.c_int)]),
        "mfmg_hip_csr_solve": (C.c_int, [vp, C.c_char_p, vp, vp]),
        "mfmg_hip_csr_vmult": (C.c_int, [vp, vp, vp]),
        "mfmg_hip_csr_apply": (C.c_int, [vp, vp, vp, C.c_int]),
        "mfmg_hip_csr_transpose": (C.c_int, [vp, P(vp)]),
        "mfmg_hip_csr_multiply": (C.c_int, [vp, vp, P(vp)]),
        "mfmg_hip_csr_download": (C.c_int, [vp, vp, vp, vp]),
        "mfmg_hip_csr_inverse_diagonal": (C.c_int, [vp, vp]),
        "mfmg_hip_csr_smoother_step": (C.c_int, [vp, vp, vp, vp, vp, dbl, dbl, vp]),
        "mfmg_hip_csr_residual": (C.c_int, [vp, vp, vp, vp]),
        "mfmg_hip_mf_laplace_create": (C.c_int, [vp, P(MeshDesc), P(vp)]),
        "mfmg_hip_mf_laplace_destroy": (C.c_int, [vp]),
        "mfmg_hip_mf_laplace_size": (C.c_int, [vp, P(i64)]),
        "mfmg_hip_mf_laplace_vmult": (C.c_int, [vp, vp, vp]),
        "mfmg_hip_mf_laplace_diagonal_inverse": (C.c_int, [vp, vp]),
        "mfmg_hip_mf_laplace_diagonal": (C.c_int, [vp, vp]),
        "mfmg_hip_mf_laplace_residual": (C.c_int, [vp, vp, vp, vp]),
        "mfmg_hip_mf_laplace_smoother_step": (C.c_int, [vp, vp, vp, vp, dbl, dbl, vp]),
        "mfmg_hip_mf_laplace_set_tile": (C.c_int, [vp, C.c_int, C.c_int]),
        "mfmg_hip_mf_laplace_set_tile_waves": (C.c_int, [vp, C.c_int]),
        "mfmg_hip_mf_laplace_get_tile": (C.c_int, [vp, P(C.c_int), P(C.c_int), P(C.c_int)]),
        "mfmg_hip_mf_laplace_f32_create": (C.c_int, [vp, P(MeshDesc), P(vp)]),
        "mfmg_hip_mf_laplace_f32_destroy": (C.c_int, [vp]),
        "mfmg_hip_mf_laplace_f32_cell_constant_layout": (C.c_int, [vp, P(C.c_int)]),
        "mfmg_hip_mf_laplace_f32_vmult": (C.c_int, [vp, vp, vp]),
        "mfmg_hip_mf_laplace_f32_diagonal_inverse": (C.c_int, [vp, vp]),
        "mfmg_hip_mf_laplace_f32_residual": (C.c_int, [vp, vp, vp, vp]),
        "mfmg_hip_mf_laplace_f32_smoother_step": (C.c_int, [vp, vp, vp, vp, C.c_float, C.c_float, vp]),
        "mfmg_hip_hierarchy_create": (C.c_int, [vp, C.c_char_p, P(MeshDesc), C.c_char_p, P(vp)]),
        "mfmg_hip_hierarchy_destroy": (C.c_int, [vp]),
        "mfmg_hip_hierarchy_apply": (C.c_int, [vp, vp, vp]),
        "mfmg_hip_hierarchy_apply_f32": (C.c_int, [vp, vp, vp]),
        "mfmg_hip_hierarchy_vmult": (C.c_int, [vp, vp, vp]),
        "mfmg_hip_hierarchy_solve_cg": (C.c_int, [vp, vp, vp, C.c_double, C.c_int32, P(C.c_int32), P(C.c_double), P(C.c_double), C.c_int32]),
        "mfmg_hip_hierarchy_n_levels": (C.c_int, [vp, P(i32)]),
        "mfmg_hip_hierarchy_level_size": (C.c_int, [vp, i32, P(i64)]),
        "mfmg_hip_hierarchy_operator_apply": (C.c_int, [vp, i32, vp, vp, C.c_int]),
        "mfmg_hip_hierarchy_smoother_apply": (C.c_int, [vp, i32, vp, vp]),
        "mfmg_hip_hierarchy_restrictor_apply": (C.c_int, [vp, i32, vp, vp, C.c_int]),
        "mfmg_hip_hierarchy_ap_apply": (C.c_int, [vp, i32, vp, vp]),
        "mfmg_hip_hierarchy_coarse_apply": (C.c_int, [vp, vp, vp]),
        "mfmg_hip_hierarchy_set_restrictor": (C.c_int, [vp, i64, i64, i64, vp, vp, vp]),
        "mfmg_hip_hierarchy_get_restrictor": (C.c_int, [vp, P(vp)]),
        "mfmg_hip_hierarchy_get_coarse_operator": (C.c_int, [vp, P(vp)]),
        "mfmg_hip_hierarchy_get_fine_operator": (C.c_int, [vp, P(vp)]),
        "mfmg_hip_hierarchy_smoother_info": (C.c_int, [vp, P(i32), P(dbl), P(dbl)]),
        "mfmg_hip_hierarchy_smoother_sweep_terms": (C.c_int, [vp, P(C.c_int), P(C.c_int)]),
        "mfmg_hip_hierarchy_sweep_tile": (C.c_int, [vp, C.c_int, P(C.c_int), P(C.c_int), P(C.c_int)]),
        "mfmg_hip_hierarchy_set_sweep_tile": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
        "mfmg_hip_hierarchy_operator_tile": (C.c_int, [vp, P(C.c_int), P(C.c_int), P(C.c_int)]),
        "mfmg_hip_hierarchy_set_operator_tile": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
        "mfmg_hip_hierarchy_timer_report": (C.c_int, [vp, C.c_char_p, sz]),
        "mfmg_hip_host_csr_shape": (C.c_int, [vp, P(i64), P(i64), P(i64)]),
        "mfmg_hip_host_csr_get": (C.c_int, [vp, vp, vp, vp]),
        "mfmg_hip_host_csr_destroy": (C.c_int, [vp]),
        "mfmg_hip_host_assemble_matrix": (C.c_int, [P(MeshDesc), C.c_int, P(vp)]),
        "mfmg_hip_host_build_restrictor": (C.c_int, [P(MeshDesc), C.c_char_p, C.c_int, P(vp)]),
        "mfmg_hip_host_galerkin": (C.c_int, [P(MeshDesc), C.c_int, i64, i64, vp, vp, vp, P(vp)]),
        "mfmg_hip_host_params_get": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, sz]),
        "mfmg_hip_host_amg_build": (C.c_int, [i64, i64, vp, vp, vp, vp, vp, vp, vp, C.c_char_p, P(vp)]),
        "mfmg_hip_host_amg_n_levels": (C.c_int, [vp, P(i32)]),
        "mfmg_hip_host_amg_get": (C.c_int, [vp, i32, i32, P(vp)]),
        "mfmg_hip_host_amg_destroy": (C.c_int, [vp]),
        "mfmg_hip_hierarchy_coarse_amg_levels": (C.c_int, [vp, P(i32)]),
        "mfmg_hip_hierarchy_coarse_amg_gather_level": (C.c_int, [vp, P(i32)]),
        "mfmg_hip_hierarchy_restrict_residual": (C.c_int, [vp, i32, vp, vp, vp]),
        "mfmg_hip_hierarchy_residual_restriction_classes": (C.c_int, [vp, i32, P(i32)]),
        "mfmg_hip_hierarchy_coarse_amg_get": (C.c_int, [vp, i32, i32, P(vp)]),
        "mfmg_hip_hierarchy_coarse_amg_smoother": (C.c_int, [vp, i32, P(i32), P(dbl), P(dbl)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    lib._declared = sorted(sig)
    _lib = lib
    return lib


def check(status: int) -> None:
    if status == SUCCESS:
        return
    msg = load().mfmg_hip_last_error().decode(errors="replace")
    if status == ERROR_NOT_IMPLEMENTED:
        raise MfmgNotImplementedError(msg)
    if status == ERROR_INVALID_ARGUMENT:
        raise MfmgInvalidArgument(msg)
    if status == ERROR_DEVICE:
        raise MfmgDeviceError(msg)
    raise MfmgError(msg)
