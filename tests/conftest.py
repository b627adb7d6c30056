import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _make(rel):
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, rel)], check=True, capture_output=True)


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    # build the HIP library / C++ driver if the tree was not built yet (hipcc cross-compiles without a GPU)
    _make(os.path.join("binary-fluctuating-lattice-boltzmann_amd", "csrc"))
    _make(os.path.join("binary-fluctuating-lattice-boltzmann_amd", "cpp"))
    return ge.load_package()


@pytest.fixture(scope="session")
def ob():
    import oracle_binding
    oracle_binding.lib()
    return oracle_binding
