"""ATen / runtime launches of the median train step of a tools/step_timeline.py --full listing, by kind: count, total us"""
import collections
import re
import sys

c = collections.defaultdict(lambda: [0, 0.0])
started = False
for line in open(sys.argv[1]):
    if line.startswith("# start_us"):
        started = True
        continue
    if not started:
        continue
    p = line.split(None, 3)
    if len(p) < 4:
        continue
    try:
        d = float(p[1])
    except ValueError:
        continue
    k = p[3].strip()
    if "at::native" in k or "rocclr" in k:
        m = re.search(r"(direct_copy_kernel|CUDAFunctor_add|FillFunctor|masked_fill|binary_internal::\w+|reduce_kernel<\d+, \d+, at::native::ReduceOp<\w+, at::native::\w+"
                      r"|CatArray\w+|index\w+|clamp\w*|sigmoid\w*|threshold\w*|copyBuffer|fillBuffer\w*|neg_kernel|\w*dropout\w*|\w*gather\w*"
                      r"|upsample\w+|bitwise\w+|Compare\w+|multi_tensor\w+|masked_scale\w*|arange\w*)", k)
        key = m.group(0)[:70] if m else k[:100]
        c[key][0] += 1
        c[key][1] += d
tot_n = sum(v[0] for v in c.values())
tot_t = sum(v[1] for v in c.values())
print(f"{tot_n:4d} {tot_t:8.1f}  ALL ATen / runtime launches")
for k, (n, t) in sorted(c.items(), key=lambda x: -x[1][1]):
    print(f"{n:4d} {t:8.1f}  {k}")
