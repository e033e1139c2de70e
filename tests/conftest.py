import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Store-free iterations that regenerate rows are taken above a measured size only (cost_sweep.hip: SGPMP_STORE_FREE_BREAK_EVEN);
# the parity tests run small shapes and want the store-free form all the same.  Read once per context, at sgpmp_create;
# test_store_free_steps_are_taken_where_they_pay puts the default back.
os.environ.setdefault("SGPMP_STORE_FREE_MIN_BYTES", "1")
# Likewise the launch of small steps: up to 512 items a Panda step goes out as fused_step_small_kernel (one workgroup per item;
# bit-identical).  The parity tests are small by necessity and are there for the kernel the big configurations run, so they keep
# it; test_small_step_launch_* compare the two and run the small one against the oracle.
os.environ.setdefault("SGPMP_NO_SMALL_STEP", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _noise_restatement_follows_the_library():
    """oracle/native_noise.py restates the in-kernel Philox stream; the round count is a build parameter of the
    library (csrc/rng.h SGPMP_PHILOX_ROUNDS), so the checker is told which one the library under test reports."""
    try:
        from oracle import native_noise
        from stoch_gpmp_amd import _lib
        native_noise.DEFAULT_ROUNDS = int(_lib.load().sgpmp_philox_rounds())
    except (ImportError, OSError, AttributeError):
        pass                                      # (library not built: the tests that need it fail on their own)
    yield


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped where no HIP device exists (the build container); on a GPU box they
    run and fail loudly if libsgpmp.so is missing."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
