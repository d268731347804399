import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def monkeypatch(monkeypatch):
    """pytest's monkeypatch, with one addition: the library reads its NND_* diagnostic switches once at load
    (nnd_reload_switches re-reads them), so setting / deleting one through this fixture also makes the library re-read them,
    and they are re-read once more after the environment has been restored."""
    from nndepth_amd._lib import lib
    setenv, delenv = monkeypatch.setenv, monkeypatch.delenv

    def setenv_(name, value, *a, **k):
        setenv(name, value, *a, **k)
        if name.startswith("NND_"):
            lib.nnd_reload_switches()

    def delenv_(name, *a, **k):
        delenv(name, *a, **k)
        if name.startswith("NND_"):
            lib.nnd_reload_switches()

    monkeypatch.setenv, monkeypatch.delenv = setenv_, delenv_
    yield monkeypatch
    monkeypatch.undo()
    lib.nnd_reload_switches()


@pytest.fixture(scope="session")
def gold():
    def load(name):
        return dict(np.load(os.path.join(GOLD, name)))
    return load


@pytest.fixture(scope="session")
def raft_sd():
    """Deterministic BaseRAFTStereo(context_dim=64) weights (the ones the goldens were made with)."""
    from nndepth_amd import weightgen
    from oracle import torch_ref as R
    return weightgen.fill_state_dict(R.raft_stereo_spec())


@pytest.fixture(scope="session")
def cre_sd():
    """Deterministic CREStereoBase weights (the ones tests/golden/cre_*.npz were made with)."""
    from nndepth_amd import weightgen
    from oracle import cre_ref as C
    return weightgen.fill_state_dict(C.cre_stereo_spec())


@pytest.fixture(scope="session")
def tartanair_frames():
    from PIL import Image
    frames = []
    for side in ("left", "right"):
        img = np.asarray(Image.open(os.path.join(GOLD, f"tartanair_000000_{side}.png")).convert("RGB"))
        t = torch.from_numpy(img.copy()).permute(2, 0, 1).float().unsqueeze(0)
        t = torch.nn.functional.interpolate(t, (544, 960), mode="bilinear")  # inference.py:55-60 preprocessing
        frames.append((t - 127.5) / 127.5)
    return frames


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))
