#!/bin/bash
# dev experiment: cost of the fused argmax epilogue's arithmetic in the vocabulary product (timing-only build, results wrong)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for d in "" "-DGVL_ABLATE_EPI" "" "-DGVL_ABLATE_EPI"; do
  GVL_BUILD_DEFS="$d" python -c "from gvl_amd import build; build.build(force=True)" > /dev/null 2>&1
  echo "== defs '$d'"
  python tools/x1_probe.py 2>&1 | grep "argmax form"
done
python -c "from gvl_amd import build; build.build(force=True)" > /dev/null 2>&1
