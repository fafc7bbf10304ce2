import importlib
import os
import sys

import pytest

# PyTorch (test plumbing for device buffers) bundles its own libamdhip64.so.  Whichever HIP runtime a process
# loads first answers to that SONAME from then on: load torch's BEFORE the product library pulls in /opt/rocm's, or a
# later `import torch` finds the other runtime under its name and reports "No HIP GPUs are available".  (The product
# itself never needs torch; bench.py imports torch first for the same reason.)
try:
    import torch  # noqa: F401
except Exception:  # a CPU-only environment without torch: the tests that need it import it themselves
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


@pytest.fixture(scope="session")
def ebo():
    """The product's ctypes plumbing (event-based-odomety_amd/__init__.py)."""
    return importlib.import_module("event-based-odomety_amd")


@pytest.fixture(scope="session")
def ebo_ab():
    """A second instance of the same plumbing bound to libebo_hip_ab.so (-DEBO_AB): the only build that reads
    the EBO_* switches of csrc/ab_env.h.  For the tests that force an implementation, a block shape or an ablation
    and compare it with what the shipped library picks by itself."""
    import importlib.util
    pkg = os.path.join(ROOT, "event-based-odomety_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("event_based_odomety_amd_ab", pkg,
                                                  submodule_search_locations=[os.path.dirname(pkg)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("event-based-odomety_amd.synth")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    import orc as _orc
    _orc.lib()
    return _orc
