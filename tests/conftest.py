import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# No pretrained asset exists here or on the GPU box (no network): every model in the tests runs on the closed-form synthetic
# weights of msmd_amd.synth, which Wav2Vec2Model.from_pretrained only hands out when asked (tests/test_host_cpu.py checks
# that it raises otherwise and that it loads a local Hugging Face checkpoint when one exists).
os.environ.setdefault("MSMD_SYNTHETIC_WEIGHTS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """GPU-marked tests are skipped (not failed) where no GPU exists, e.g. a plain `pytest tests/` here."""
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _reset_training_noise():
    """The training-noise switch is module-level state: start every test from eval-mode arithmetic."""
    yield
    try:
        from msmd_amd import autograd as ag
        ag.TrainNoise.active = False
        ag.TrainNoise.graph_safe = False
        ag.TrainNoise.spec_masks = None
        ag.DIRECT_GRAD = False
    except Exception:
        pass
