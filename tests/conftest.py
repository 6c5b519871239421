import importlib.util
import os
import sys

import pytest

# The library keeps up to nine HIP streams busy; the runtime's default of four hardware queues would make unrelated
# streams share a queue.  A host sets this before its first HIP call; the library reads no environment.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

try:  # torch bundles its own libamdhip64: import it BEFORE libgpsbb.so is loaded so both share one HIP runtime
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_package():
    """Import pluto-gps-sim_amd/ (not a valid identifier, so by path) as `pluto_gps_sim_amd`."""
    name = "pluto_gps_sim_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "pluto-gps-sim_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    mod = load_package()
    mod.build()
    return mod


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding as ob
    return ob.Oracle()


@pytest.fixture(scope="session")
def synth(pkg):
    """One GPU handle for the whole session; the HIP path must load — no fallback."""
    s = pkg.Synth(0)
    yield s
    s.close()


GOLDEN = os.path.join(ROOT, "tests", "golden")


# The reference's channel_t (plutogpssim.h:152-174, LP64) as numpy records, with and without FLOAT_CARR_PHASE: the package's
# (bench.py's fill_block leg uses them too); tests/test_ref_layout.py checks every offset against offsetof() on the real header.
REF_CHANNEL_DTYPE = load_package().REF_CHANNEL_DTYPE
REF_CHANNEL_FIXED_DTYPE = load_package().REF_CHANNEL_FIXED_DTYPE
