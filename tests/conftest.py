import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def models():
    from retto_amd import synth
    return synth.synth_models(0)


@pytest.fixture(scope="session")
def oracle_session(models):
    from oracle.pipeline import OracleSession
    det, cls, rec, dic = models
    return OracleSession(det, cls, rec, dic)


@pytest.fixture(scope="session")
def hip_session():
    """The HIP session. Fails loudly (never skips silently into a CPU path) when the
    library or the device is missing."""
    import retto_amd
    s = retto_amd.RettoSession(retto_amd.synthetic_session_config(0))
    yield s
    s.close()


def rand_page(h, w, seed):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)
