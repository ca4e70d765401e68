import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Multi-process GPU tests fork their ranks from a fork server that is started HERE, before any test
# has initialised the GPU: a process that already holds a GPU context must not exec another program
# (which both the "spawn" start method and subprocess do), while forking the clean server is fine.
import multiprocessing as _mp
from multiprocessing import forkserver as _forkserver

try:
    _mp.get_context("forkserver")
    _forkserver.ensure_running()
except Exception:  # pragma: no cover - platforms without the forkserver start method
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def split_prefixed(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


@pytest.fixture
def golden():
    return load_golden
