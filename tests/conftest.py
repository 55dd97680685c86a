import os
import sys

import numpy as np
import pytest
import torch

# the CPU oracle is a chain of small ops: beyond ~16 threads it gets slower, not faster (128-core GPU hosts)
torch.set_num_threads(min(16, os.cpu_count() or 1))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# test infrastructure only: run the suite against a VARIANT build of the library (same ABI; scripts/build_file_variant.sh,
# `python -m ladiff_amd.build --tag=...`) for a same-box A/B.  The product never reads this variable.
if os.environ.get("LADIFF_TEST_LIB"):
    from ladiff_amd import _lib as _lib_for_variant
    _lib_for_variant.LIB_PATH = os.path.join(ROOT, os.environ["LADIFF_TEST_LIB"])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: soak tests (a few seconds each on the GPU; `-m gpu` still selects them)")


def load_golden(name):
    import torch
    with np.load(os.path.join(GOLDEN, name + ".npz")) as f:
        return {k: (torch.from_numpy(f[k]) if f[k].dtype.kind in "fiub" else f[k]) for k in f.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden
