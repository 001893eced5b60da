import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gold():
    class G:
        def npz(self, name):
            return np.load(os.path.join(GOLD, name))

        def json(self, name):
            import json
            return json.load(open(os.path.join(GOLD, name)))
    return G()


@pytest.fixture(scope="session")
def fbank_tag_state():
    """(checkpoint dict, cpu model object) of the synthetic fbank-tag anonymizer, seed 0"""
    import satools_amd  # noqa: F401
    from satools_amd import synthetic
    return synthetic.checkpoint("hifigan_bn_tdnnf_600h_vq_48_v1")


def rms(a):
    a = np.asarray(a, dtype=np.float64)
    return float(np.sqrt(np.mean(a * a)))
