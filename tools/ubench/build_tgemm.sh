#!/bin/bash
# build tools/_bin/tgemm_<variant> for each ablation variant:  build_tgemm.sh  (variants: name=defs ...)
root=$(cd $(dirname $0)/../.. && pwd)
mkdir -p $root/tools/_bin
build() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I $root/include -I $root/gvl_amd/csrc $2 -o $root/tools/_bin/tgemm_$1 $root/tools/ubench/tgemm_bench.hip 2>&1 | grep -v warning | grep -i "error" ; }
build base "-DGVL_WG_STAMPS" &
build nomfma "-DGVL_WG_STAMPS -DGVL_WG_NO_MFMA" &
build nofrag "-DGVL_WG_STAMPS -DGVL_WG_NO_MFMA -DGVL_WG_NO_FRAG" &
build nostore "-DGVL_WG_STAMPS -DGVL_WG_NO_MFMA -DGVL_WG_NO_FRAG -DGVL_WG_NO_STORE" &
build noload "-DGVL_WG_STAMPS -DGVL_WG_NO_LOAD" &
build mfmaonly "-DGVL_WG_STAMPS -DGVL_WG_NO_LOAD -DGVL_WG_NO_STORE -DGVL_WG_NO_FRAG" &
wait
ls -la $root/tools/_bin/tgemm_*
