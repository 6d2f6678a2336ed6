import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "medical-cross-modality-domain-adaptation_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: BASELINE-batch (B=16) oracle comparisons, ~1 min of host time each (still part of -m gpu)")


def pkg(sub=None):
    return importlib.import_module(PKG + ("." + sub if sub else ""))


@pytest.fixture(scope="session")
def built():
    """libpnp_hip.so built in-tree (cross-compiles on CPU-only hosts)"""
    import __graft_entry__ as ge
    return ge.build()


@pytest.fixture(scope="session")
def dev(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
