#!/bin/bash
# Timing-only builds of the eight-wavefront k_gemm_f16x3 (which part of a K stage costs what; DESIGN_LOG.md section 4.4):
#   base | nodma (no LDS-DMA in the loop) | nolds (operands from registers) | nobar | mfmaonly | nogroups (compiler's own
#   instruction order).  Build here (hipcc cross-compiles), run on the GPU box:  for v in tools/_bin/*; do $v; done
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
for v in "base:" "nodma:-DGVL_ABLATE_DMA" "nolds:-DGVL_ABLATE_LDS" "nobar:-DGVL_ABLATE_BARRIER" \
         "mfmaonly:-DGVL_ABLATE_DMA -DGVL_ABLATE_LDS -DGVL_ABLATE_BARRIER" "nogroups:-DGVL_NO_SCHED_GROUPS"; do
  name=${v%%:*}; flags=${v#*:}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I gvl_amd/csrc $flags -DVARIANT="\"$name\"" \
        tools/gemm16_ablate.hip gvl_amd/csrc/gvl_msda.hip -o tools/_bin/$name
done
ls tools/_bin
