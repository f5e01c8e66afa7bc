#!/bin/bash
# dev experiment: tile-walk group size of the split-fp16 GEMM kernels (row tiles per XCD group), rebuilt on the GPU box
cd ${GRAFT_REPO_ROOT:-/root/repo}
for g in 8 4 16 2 8; do
  GVL_BUILD_DEFS="-DGVL_GROUPM=$g" python -c "from gvl_amd import build; build.build(force=True)" > /dev/null 2>&1
  echo "== GVL_GROUPM=$g"
  python tools/x1_probe.py 2>&1 | grep "h product\|argmax form x3\|gate product + cell x3\|vocabulary"
done
python -c "from gvl_amd import build; build.build(force=True)" > /dev/null 2>&1
