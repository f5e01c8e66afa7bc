"""The C ABI used from plain C: tests/c_abi/c_abi_smoke.c is compiled with gcc against include/gvl_msda.h, linked with
gvl_amd/libgvl_msda.so, the HIP runtime (device memory) and the C oracle (oracle/libgvl_oracle.so), and run -- no
Python or torch in the process that exercises the boundary."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_drives_the_library(tmp_path):
    gcc = shutil.which("gcc")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    assert gcc and os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime_api.h"))
    lib_dir, ora_dir = os.path.join(ROOT, "gvl_amd"), os.path.join(ROOT, "oracle")
    if not os.path.exists(os.path.join(ora_dir, "libgvl_oracle.so")):
        subprocess.check_call(["make", "-s", "-C", ora_dir])
    exe = str(tmp_path / "c_abi_smoke")
    subprocess.check_call([gcc, "-std=c11", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(rocm, "include"), os.path.join(ROOT, "tests", "c_abi", "c_abi_smoke.c"),
                           "-L", lib_dir, "-lgvl_msda", "-L", ora_dir, "-lgvl_oracle", "-L", os.path.join(rocm, "lib"),
                           "-lamdhip64", "-lm", f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{ora_dir}",
                           f"-Wl,-rpath,{os.path.join(rocm, 'lib')}", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "C ABI OK" in out.stdout and out.stdout.count("pad=") == 2
