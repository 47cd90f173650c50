"""AddressSanitizer + UndefinedBehaviorSanitizer over the HOST code of libdygnn_hip.so (the C++ CSR builder, every entry point's
argument validation, the planning / packing code): the `asan` build variant (dyglib_amd/_build.py) is loaded into a child Python
with clang's ASan runtime preloaded and runs the boundary tests and the property tests against it.  GPU ASan is not available on
this pool (it needs xnack+ code objects), so the gfx950 kernels are covered by the parity tests instead."""
import os
import subprocess
import sys

import pytest

from dyglib_amd import _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def asan_env():
    _build.build(verbose=False, variant="asan")
    env = dict(os.environ)
    env.update(LD_PRELOAD=_build.asan_runtime(), DYGNN_LIB_VARIANT="asan", ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", PYTHONPATH=ROOT)
    return env


@pytest.mark.timeout(900)
def test_boundary_and_property_tests_are_clean_under_asan_ubsan(asan_env):
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_capi_cpu.py", "tests/test_properties_cpu.py", "tests/test_sampling_strategies.py", "-m", "not gpu",
                        "-x", "-q", "-p", "no:cacheprovider"],          # incl. the MT19937 draw replay (csr_host.cpp) against numpy
                       cwd=ROOT, env=asan_env, capture_output=True, text=True)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert "libdygnn_hip_asan.so" in subprocess.run(
        [sys.executable, "-c", "from dyglib_amd import _capi; print(_capi.load()._name)"], cwd=ROOT, env=asan_env, capture_output=True, text=True).stdout


OVERFLOW = """
import ctypes as C, numpy as np
from dyglib_amd import _capi
lib = _capi.load()
libc = C.CDLL(None); libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]
src = np.array([1, 2, 3, 1], dtype=np.int64); dst = np.array([2, 3, 1, 3], dtype=np.int64)
eid = np.arange(1, 5, dtype=np.int64); ts = np.arange(4, dtype=np.float64)
indptr = libc.malloc(3 * 8)                    # num_nodes + 1 = 5 entries are written: two too few
nbr = np.empty(8, dtype=np.int32); eo = np.empty(8, dtype=np.int32); to = np.empty(8, dtype=np.float64)
lib.dygnn_csr_build_host(4, src.ctypes.data, dst.ctypes.data, eid.ctypes.data, ts.ctypes.data, 4, indptr, nbr.ctypes.data, eo.ctypes.data, to.ctypes.data)
print("survived")
"""


def test_the_sanitizer_is_live(asan_env):
    """negative control: a caller buffer that is too small must be reported, otherwise the clean run above proves nothing"""
    r = subprocess.run([sys.executable, "-c", OVERFLOW], cwd=ROOT, env=asan_env, capture_output=True, text=True)
    assert r.returncode != 0 and "AddressSanitizer: heap-buffer-overflow" in r.stderr and "survived" not in r.stdout
