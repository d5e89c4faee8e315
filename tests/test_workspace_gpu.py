"""GPU: launches sized from the memory that is free (interpret_quality_amd/workspace.py).

Round 3's record (gpurun_out/test_failures/sweep_250.txt): `torch.OutOfMemoryError: Tried to allocate 17.39 GiB ... 99.80 GiB
allocated` in dgcnn.py when two stage processes shared one GPU - every engine sized its workspace for 4096 clouds per launch
whatever was free.  Here a dummy tensor pins all but a few GB of the card and (1) the engines of DGCNN and PointConv, in this
process, evaluate 3300 coalitions to the same bits as with the card empty, in several smaller launches; (2) the scale-sweep
stage script of both families, as a SECOND process on the pinned card, writes the same artefact files."""
import argparse
import gc
import os
import sys

import numpy as np
import pytest
import torch

from interpret_quality_amd import final_common, hip_ops, synth, workspace
from test_dist_gpu import REPO, _artefacts, _assert_same, _env, _run, _run_chains

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def _release():
    gc.collect()
    torch.cuda.empty_cache()


def _pin(leave_bytes):
    """A tensor that leaves about ``leave_bytes`` of device memory free."""
    _release()
    free, _ = torch.cuda.mem_get_info(dev())
    n = free - leave_bytes
    assert n > 0, "card already fuller than the test assumes (%d free)" % free
    return torch.empty(int(n), dtype=torch.uint8, device=dev())


def _model(name):
    from interpret_quality_amd.dgcnn import DGCNN_cls
    from interpret_quality_amd.pointconv import PointConvDensityClsSsg
    cls, sd = {"dgcnn": (DGCNN_cls, synth.dgcnn_state_dict), "pointconv": (PointConvDensityClsSsg, synth.pointconv_state_dict)}[name]
    m = cls(argparse.Namespace(dataset="modelnet10", k=20) if name == "dgcnn" else None)
    m.load_state_dict(synth.to_torch(sd(0)))
    return m.to(dev()).eval()


@pytest.mark.parametrize("name", ["dgcnn", "pointconv"])
def test_engine_fits_its_launches_into_a_nearly_full_card(name):
    pts, label = synth.make_cloud(4)
    data = torch.from_numpy(pts).unsqueeze(0).to(dev())
    lbl = torch.tensor([label], device=dev())
    region_id = hip_ops.region_assign(data[0].contiguous(), hip_ops.fps(data, 32)[0].contiguous()).cpu().numpy().astype(np.int64)
    orders = synth.make_orders(100, 32, seed=3)
    a = argparse.Namespace(model=name, softmax_type="modified", num_points=1024, num_regions=32, num_samples=100, shapley_batch_size=20, verbose=False)
    m = _model(name)
    phi, logits = final_common.shap_sampling_all_regions_batch(m, data, lbl, region_id, orders, a)      # the card to itself: one launch
    full_step = workspace.STATS["last_step"]
    logits = logits.cpu()
    del m
    _release()
    fitted, retries = workspace.STATS["fitted_below_cap"], workspace.STATS["oom_retries"]
    ballast = _pin(6 << 30)                                       # 6 GB left of 288: DGCNN wants 14 GB, PointConv 31 GB for one launch
    try:
        m = _model(name)
        phi2, logits2 = final_common.shap_sampling_all_regions_batch(m, data, lbl, region_id, orders, a)
        assert workspace.STATS["fitted_below_cap"] > fitted or workspace.STATS["oom_retries"] > retries
        assert 0 < workspace.STATS["last_step"] < full_step       # several smaller launches
        assert torch.equal(logits2.cpu(), logits) and np.array_equal(phi, phi2)
    finally:
        del ballast
        m = None
        _release()


def test_stage_scripts_share_a_nearly_full_card_with_another_process(tmp_path):
    """The reference's way of using one card for two scripts (README.md:87): this process holds the card but for ~12 GB, the
    scale-sweep stage of DGCNN and PointConv runs as a child process next to it and writes the files it writes on an empty card."""
    out = {}
    for tag in ("empty", "pinned"):
        work = tmp_path / tag
        work.mkdir()
        ballast = _pin(12 << 30) if tag == "pinned" else None
        try:
            chains = []
            for model in ("dgcnn", "pointconv"):
                common = ["--model", model, "--dataset", "modelnet10", "--synthetic", "--num_clouds", "1"]
                chains.append([([sys.executable, os.path.join(REPO, "final_shapley_value.py")] + common + ["--num_samples_save", "100"], work, _env()),
                               ([sys.executable, os.path.join(REPO, "final_scale_center_enum_all.py")] + common, work, _env())])
            if tag == "empty":      # the reference run on an empty card: the two models side by side (start-up dominates; the
                _run_chains(chains)  # dataset's FPS index file is written under a temporary name and renamed: two writers are safe)
            else:                   # the pinned card: one child at a time next to the ballast, as before
                for chain in chains:
                    for step in chain:
                        _run(*step)
        finally:
            del ballast
            _release()
        out[tag] = _artefacts(work)
    _assert_same(out["empty"], out["pinned"])
    assert any(k.endswith("scale_all/region_shapley_value.npy") for k in out["empty"])
