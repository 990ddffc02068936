#!/bin/bash
# A library of its own for timing experiments that skip work and give WRONG results by design (sgs_profile modes > 1:
# gmg_sgs.hpp kSwExperiments).  The shipped csrc/libgmgcoulomb.so is built without GMG_EXPERIMENTS and refuses those modes.
# Use: GMG_DEVICE_LIB=$(bash tools/build_experiments.sh) GMG_OPTIONS=sgs_profile=2 python tools/sgs_probe.py 20 5 1 2
set -e
cd "$(dirname "$0")/.."
P=geometric-multigrid-preconditioners-for-long-range-coulomb-interaction_amd/csrc
OUT=${OUT:-/tmp/libgmgcoulomb_experiments.so}
hipcc -O3 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -fno-jump-tables -std=c++17 -DGMG_EXPERIMENTS -Wall -Wno-unused-function \
  -o $OUT $P/gmg_coulomb.hip -pthread -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib 1>&2
echo $OUT
