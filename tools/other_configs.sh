#!/bin/bash
# the bench at the other BASELINE configurations -> gpurun_out/r06_other_configs.json (one JSON object of bench lines)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
python - <<'PY'
import json, subprocess, sys
runs = {
    "r06_yc2_f32": ["--cfg", "yc2_tsn_dvc", "--T", "512", "--queries", "100"],
    "r06_yc2_bf16": ["--cfg", "yc2_tsn_dvc", "--T", "512", "--queries", "100", "--dtype", "bf16"],
    "r06_cfgA_bf16": ["--dtype", "bf16"],
    "r06_cfgA_T512": ["--T", "512"],
    "r06_cfgA_fixed_layout": ["--fixed-layout"],
}
out = {}
for name, args in runs.items():
    p = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "16"] + args, capture_output=True, text=True)
    try:
        d = json.loads(p.stdout.strip().splitlines()[-1])
        keep = {k: d.get(k) for k in ("value", "unit", "ms_per_step", "train_step_ms", "dtype", "config", "eval_graphs", "train_graphs")}
        keep["fwd_roofline"] = {k: d["roofline"].get(k) for k in ("kernel_us", "frac", "algorithmic_bytes")}
        keep["bwd_roofline"] = (d.get("train_roofline") or {}).get("launches")
        keep["args"] = " ".join(args)
        out[name] = keep
    except Exception as e:                                  # noqa: BLE001
        out[name] = {"error": str(e), "stderr": p.stderr[-400:]}
json.dump(out, open("gpurun_out/r06_other_configs.json", "w"), indent=1)
for k, v in out.items():
    print(k, v.get("value"), v.get("ms_per_step"), v.get("train_step_ms"), (v.get("fwd_roofline") or {}).get("frac"))
PY
