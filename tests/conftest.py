import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir() -> Path:
    return GOLDEN


@pytest.fixture(scope="session")
def tiny_state_dict():
    from fitclip_amd import synth
    return synth.make_state_dict(synth.TINY, seed=42)


@pytest.fixture(scope="session")
def vitb16_state_dict():
    from fitclip_amd import synth
    return synth.make_state_dict(synth.VIT_B_16, seed=42)
