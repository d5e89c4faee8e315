"""CPU: launch sizing from free memory and the halve-and-retry on an allocation failure (interpret_quality_amd/workspace.py),
with the device's memory figures stubbed."""
import pytest
import torch

from interpret_quality_amd import workspace
from interpret_quality_amd._lib import IqError


class _Eng:
    device = torch.device("cpu")
    _ws = None


def _need(k):          # 4 MB per cloud + a fixed 16 MB table, like the engines' workspace_bytes
    return (16 << 20) + k * (4 << 20)


def test_fit_step_is_the_largest_launch_that_fits(monkeypatch):
    monkeypatch.setattr(workspace, "available_bytes", lambda device, held=0: (16 << 20) + 1000 * (4 << 20) + 123)
    assert workspace.fit_step(4096, _need, "cpu") == 1000
    assert workspace.fit_step(800, _need, "cpu") == 800                    # fits as asked
    assert workspace.fit_step(4096, _need, "cpu", held=_need(4096)) == 4096  # the engine's own workspace already covers it
    monkeypatch.setattr(workspace, "available_bytes", lambda device, held=0: 1 << 20)
    with pytest.raises(IqError, match="not enough free device memory"):
        workspace.fit_step(4096, _need, "cpu")


def test_run_in_steps_covers_every_row_once_and_halves_on_out_of_memory(monkeypatch):
    monkeypatch.setattr(workspace, "available_bytes", lambda device, held=0: _need(300))
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    eng, calls, fail = _Eng(), [], {"left": 2}

    def call(lo, hi):
        if hi - lo > 100 and fail["left"] > 0:     # "another process took the memory": the first two attempts do not fit
            fail["left"] -= 1
            raise torch.OutOfMemoryError("stub")
        calls.append((lo, hi))
        return torch.arange(lo, hi).reshape(-1, 1)

    before = workspace.STATS["oom_retries"]
    out = workspace.run_in_steps(eng, 1000, 4096, _need, call)
    assert torch.equal(out.reshape(-1), torch.arange(1000))
    assert workspace.STATS["oom_retries"] == before + 2 and workspace.STATS["last_step"] == 75      # 300 -> 150 -> 75
    assert calls[0] == (0, 75) and calls[-1][1] == 1000 and all(a[1] == b[0] for a, b in zip(calls, calls[1:]))
    # a failure that halving cannot cure travels on
    with pytest.raises(torch.OutOfMemoryError):
        workspace.run_in_steps(eng, 1000, 4096, _need, lambda lo, hi: (_ for _ in ()).throw(torch.OutOfMemoryError("stub")))
    # an empty launch still reaches the engine (it returns the empty logits tensor)
    assert workspace.run_in_steps(eng, 0, 4096, _need, lambda lo, hi: torch.zeros((0, 10))).shape == (0, 10)
