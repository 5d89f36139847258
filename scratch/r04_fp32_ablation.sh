#!/bin/bash
# VERDICT r03 item 6 (BASELINE configs[4]: FP32, "cell-local evaluation as batched dense contraction on MFMA"): what does the cell
# arithmetic cost the FP32 eight-coefficient operator kernel?  A measurement build of the library in /tmp with the cell kernel
# replaced by one multiply per corner (mf_device.hpp: MFMG_MF_ABLATE_CELL; loads, lane shifts, carries, stores kept), timed
# against the product build on the same box: if the whole arithmetic is worth < 10 % of the launch, matrix cores cannot win.
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
set -e
mkdir -p /tmp/abl
cd $R/mfmg_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fopenmp -DMFMG_MF_ABLATE_CELL=1 -c mf_laplace.hip -o /tmp/abl/mf_laplace.hip.o
OBJS=$(ls build/*.o | grep -v "build/mf_laplace.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fopenmp -o /tmp/abl/libmfmg_hip.so /tmp/abl/mf_laplace.hip.o $OBJS -ldl
cd $R
for mat in linear constant; do
  echo "== FP32 smoother apply (Chebyshev(3), one launch per term), 257^3 DoFs, material $mat"
  MFMG_MF_F32_TIME=$mat python3 scratch/fp32_smoother_time.py product
  MFMG_HIP_LIBRARY=/tmp/abl/libmfmg_hip.so MFMG_MF_F32_TIME=$mat python3 scratch/fp32_smoother_time.py "cell arithmetic removed"
done
