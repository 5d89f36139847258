"""mfmg_amd: MI355X-native V-cycle apply path behind the mfmg operator API.

Python is plumbing here (device memory through torch, process groups through
torch.distributed); every numerical operation runs in libmfmg_hip.so."""
from . import lib
from .api import (Context, Hierarchy, MatrixFreeLaplace, MatrixFreeLaplaceF32, SparseMatrixDevice, host_assemble_matrix,
                  host_amg_build, host_build_restrictor, host_galerkin, info_to_params, memory_inventory, params_to_info)
from .laplace import LaplaceProblem, material_property
from .distributed import BoxPartition, HaloTransport, SlabPartition, box_grid

__all__ = [
    "lib", "Context", "Hierarchy", "MatrixFreeLaplace", "MatrixFreeLaplaceF32", "SparseMatrixDevice", "LaplaceProblem",
    "material_property", "SlabPartition", "BoxPartition", "box_grid", "HaloTransport", "host_assemble_matrix", "host_build_restrictor", "host_galerkin", "host_amg_build", "params_to_info", "info_to_params", "memory_inventory",
]
