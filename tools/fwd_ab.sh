#!/bin/bash
# A/B of the forward sampling kernel: the shipped library against a timing build in tools/_bin/libgvl_msda_dev.so (made HERE before
# the call with `python -m gvl_amd.build --dev <source> [-D...]`: the GPU box only runs it).  Per-workgroup phase stamps of the
# decoder launch (in situ and back to back) + interleaved eval bench lines.  Results under gpurun_out/fwd_ab/.
# Round 5 used it for (1) -DGVL_FWD_GROUPED=1 (four sample steps' LDS reads requested together; not kept) and (2) the
# instruction-count work on the sample loop against the previous commit's source.
out=$PWD/gpurun_out/fwd_ab; mkdir -p $out
dev=$PWD/tools/_bin/libgvl_msda_dev.so
python3 tools/fwd_phase_stamps.py 2>/dev/null | grep -v "^backward" > $out/stamps_shipped.txt
GVL_LIB_PATH=$dev python3 tools/fwd_phase_stamps.py 2>/dev/null | grep -v "^backward" > $out/stamps_dev.txt
for k in 1 2 3; do
  GVL_LIB_PATH=$dev python3 bench.py --mode eval --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 > $out/dev_$k.json
  python3 bench.py --mode eval --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 > $out/shipped_$k.json
done
