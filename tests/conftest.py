import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def pointnet_sd():
    from interpret_quality_amd import synth
    return synth.to_torch(synth.pointnet_state_dict(0))


def assert_close_elementwise(got, want, rtol=1e-4, floor_frac=0.05):
    """Element-wise relative check beside the norm-wise one: |got - want| <= rtol * max(|want|, floor_frac * max|want|) for
    EVERY element.  The floor keeps the bar meaningful for entries that are small because of cancellation (their absolute
    error is set by the magnitude of the summed terms, not by their own value): such an entry is held to
    rtol * floor_frac * max|want| = 5e-6 of the largest entry, twenty times tighter than the norm-wise bar."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    bound = rtol * np.maximum(np.abs(want), floor_frac * np.abs(want).max())
    ratio = np.abs(got - want) / np.maximum(bound, 1e-300)
    assert ratio.max() <= 1.0, "element-wise error %.2f x the bound at index %s (got %r, want %r)" % (
        ratio.max(), np.unravel_index(ratio.argmax(), ratio.shape), got.reshape(-1)[ratio.argmax()], want.reshape(-1)[ratio.argmax()])
