"""Build libgvl_msda.so (HIP, gfx950) in-tree with hipcc.  Used by __graft_entry__.build() and `python -m gvl_amd.build`."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", n) for n in ("gvl_msda.hip", "gvl_cap.hip", "gvl_cap_train.hip", "gvl_criterion.hip", "gvl_lsap_dev.hip", "gvl_proj.hip", "gvl_gemm16.hip", "gvl_layers.hip", "gvl_train_layers.hip", "gvl_train_gemm.hip", "gvl_mha_train.hip", "gvl_optim.hip", "gvl_lsap.cpp")]
OUT = os.path.join(HERE, "libgvl_msda.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics", "-pthread",
         "-Wall", "-Wno-unused-function"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0)")


def needs_build():
    if not os.path.exists(OUT):
        return True
    deps = [s for s in SRC if os.path.exists(s)] + [os.path.join(ROOT, "include", "gvl_msda.h"), __file__,
                                                    os.path.join(HERE, "csrc", "gvl_common.hpp"),
                                                    os.path.join(HERE, "csrc", "gvl_gemm16_common.hpp")]
    return any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps)


def _deps(src):
    return [src, os.path.join(ROOT, "include", "gvl_msda.h"), __file__, os.path.join(HERE, "csrc", "gvl_common.hpp"),
            os.path.join(HERE, "csrc", "gvl_gemm16_common.hpp")]


def build(force=False, verbose=False, save_temps=None):
    """one object per source file under build/obj (only the stale ones are recompiled, up to 8 at a time), then one link"""
    if not force and not needs_build():
        return OUT
    srcs = [s for s in SRC if os.path.exists(s)]
    cc = hipcc()
    objdir = os.path.join(ROOT, "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    if save_temps:
        os.makedirs(save_temps, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"] + ["-I", os.path.join(ROOT, "include"), "-c"]

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if (not force and os.path.exists(obj)
                and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in _deps(src) if os.path.exists(d))):
            return obj
        cmd = [cc] + cflags + ["-o", obj, src]
        if save_temps:
            cmd += [f"-save-temps={save_temps}", "-Rpass-analysis=kernel-resource-usage"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=save_temps or ROOT)
        return obj
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=ROOT)
    return OUT


OUT_DEV = os.path.join(ROOT, "tools", "_bin", "libgvl_msda_dev.so")     # (git-ignored, outside the package: never beside the product library)


def build_dev(sources, defs, verbose=False):
    """Timing / ablation builds of single kernels (`tools/*_ablate.sh`, `*_stamps.sh`): the named source files compiled with `defs`
    (-DGVL_...) into build/obj_dev, linked with the SHIPPED library's other objects into tools/_bin/libgvl_msda_dev.so -- the shipped
    library and its objects are never touched.  A process uses it through GVL_LIB_PATH (gvl_amd/_lib.py)."""
    objdir, devdir = os.path.join(ROOT, "build", "obj"), os.path.join(ROOT, "build", "obj_dev")
    cc = hipcc()
    os.makedirs(devdir, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"] + ["-I", os.path.join(ROOT, "include"), "-c"] + list(defs)
    given = {os.path.basename(n): os.path.abspath(n) for n in sources}       # (a copy elsewhere, e.g. an older revision of the
    names = set(given)                                                       #  file, replaces the library's source of that name)
    unknown = names - {os.path.basename(s) for s in SRC}
    if unknown:
        raise ValueError(f"build_dev: not a source of the library: {sorted(unknown)}")
    objs = []
    for src in [s for s in SRC if os.path.exists(s)]:
        base = os.path.basename(src)
        if base in names:
            obj = os.path.join(devdir, base + ".o")
            use = given[base] if os.path.exists(given[base]) else src
            cmd = [cc] + cflags + ["-I", os.path.dirname(src), "-o", obj, use]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd, cwd=ROOT)
        else:
            obj = os.path.join(objdir, base + ".o")
            fresh = os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in _deps(src) if os.path.exists(d))
            if not fresh:
                # a checkout that carries the library but not (or no longer) its objects: this file's plain object goes to
                # build/obj_dev as well -- neither build/obj nor the shipped library is written from here
                obj = os.path.join(devdir, base + ".plain.o")
                plain = [f for f in FLAGS if f != "-shared"] + ["-I", os.path.join(ROOT, "include"), "-c"]
                if not (os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in _deps(src) if os.path.exists(d))):
                    subprocess.check_call([cc] + plain + ["-o", obj, src], cwd=ROOT)
        objs.append(obj)
    subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", OUT_DEV] + objs, cwd=ROOT)
    return OUT_DEV


if __name__ == "__main__":
    if "--dev" in sys.argv:                  # python -m gvl_amd.build --dev gvl_gemm16.hip -DGVL_V_NO_EPI ...
        rest = sys.argv[sys.argv.index("--dev") + 1:]
        print(build_dev([a for a in rest if not a.startswith("-")], [a for a in rest if a.startswith("-")], verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True,
                    save_temps=os.path.join(ROOT, "build", "temps") if "--save-temps" in sys.argv else None))
