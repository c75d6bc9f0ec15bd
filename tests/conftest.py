import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have(path):
    return os.path.exists(os.path.join(ROOT, path))


@pytest.fixture(scope="session", autouse=True)
def built_libraries():
    """Builds whatever is missing (hipcc cross-compiles without a GPU)."""
    need = ["actinon_amd/lib/libactinon_hip.so", "actinon_amd/lib/libactinon_host.so", "oracle/libacn_oracle.so",
            "oracle/libacn_oracle_libm.so"]
    if not all(_have(p) for p in need):
        subprocess.check_call(["make", "-C", ROOT, "-j", str(min(8, os.cpu_count() or 1)), "all"])
    yield


@pytest.fixture(scope="session")
def oracle():
    from oracle_binding import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def oracle_libm():
    from oracle_binding import Oracle
    return Oracle(libm=True)


@pytest.fixture(scope="session")
def detmath_cpu():
    """acn_detmath.h compiled for the host (gcc, -ffp-contract=off) behind a tiny array shim."""
    import ctypes as C
    src = os.path.join(ROOT, "tests", "csrc", "detmath_cpu.c")
    out = os.path.join(ROOT, "build", "libdetmath_cpu.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(
            os.path.join(ROOT, "actinon_amd/csrc/acn_detmath.h"))):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-I",
                               os.path.join(ROOT, "actinon_amd/csrc"), src, "-o", out, "-lm"])
    lib = C.CDLL(out)
    lib.detmath_eval.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    return lib
