#!/bin/bash
# dev experiment: phases of k_big_f16x3<gates> (timing build with in-kernel stamps)
cd ${GRAFT_REPO_ROOT:-/root/repo}
touch gvl_amd/csrc/gvl_gemm16.hip
GVL_BUILD_DEFS="-DGVL_G_STAMPS $1" python -c "from gvl_amd import build; build.build()" > /dev/null 2>&1
python tools/gates_probe.py --reps 10 2>&1 | grep "k_gates\|round" | tail -12
touch gvl_amd/csrc/gvl_gemm16.hip
python -c "from gvl_amd import build; build.build()" > /dev/null 2>&1
