import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A checkout without the built artefacts (they are git-ignored) is built once, the way __graft_entry__.build()
    does: hipcc -> gvl_amd/libgvl_msda.so, gcc -> oracle/libgvl_oracle.so.  Nothing is substituted when that fails:
    the tests that need the library then fail with its own "missing library" error."""
    import subprocess
    from gvl_amd import build as b
    try:
        if b.needs_build():                                 # missing OR older than its sources
            b.build()
        if not os.path.exists(os.path.join(ROOT, "oracle", "libgvl_oracle.so")):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    except Exception as e:                                   # noqa: BLE001
        print(f"[conftest] could not build the native pieces: {e}", file=sys.stderr)


class _EnvMonkeyPatch(pytest.MonkeyPatch):
    """libgvl_msda.so caches its GVL_* switches per process (no getenv() on a launch path): a test that flips one through
    monkeypatch must make the library read it again -- on the change and when it is undone"""

    @staticmethod
    def _reload():
        from gvl_amd import _lib
        _lib.reload_env()

    def setenv(self, *a, **k):
        super().setenv(*a, **k)
        self._reload()

    def delenv(self, *a, **k):
        super().delenv(*a, **k)
        self._reload()

    def undo(self):
        super().undo()
        self._reload()


@pytest.fixture
def monkeypatch():
    mp = _EnvMonkeyPatch()
    yield mp
    mp.undo()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _no_leaked_library_state(request):
    """process-wide state of libgvl_msda.so that a test (or the product code it drives) may change for a scope -- the number of fp16
    products per fp32 product (MSDA.f16_products) -- must be back at its default when the test ends: a leak makes every LATER test
    in the process run at 11-bit operands (seen as a 1.6e-4 error in an unrelated test)"""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    from gvl_amd import MultiScaleDeformableAttention as MSDA
    now = MSDA.f16_products_now()
    if now != 3:
        MSDA.f16_products(3).__enter__()                    # (do not let one leak fail every later test too)
        pytest.fail(f"{request.node.nodeid} left the library at {now} fp16 product(s) per fp32 product")
