#!/bin/bash
# same-box A/B of two builds of the library (scratch/lib_old.so, scratch/lib_new.so): ms per Chebyshev(3) sweep at 257^3 and 512^3
cd ${GRAFT_REPO_ROOT:?}
for rep in 1 2; do
for v in old new; do
  cp scratch/lib_$v.so mfmg_amd/libmfmg_hip.so
  for n in 257 512; do
    echo -n "$v $n: "; python scratch/sweep_time.py $n 3 "0,0,0" 2>&1 | tail -1
  done
done
done
