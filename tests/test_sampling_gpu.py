"""GPU: coalition sampling on the device (csrc/iq_sample.hip) against NumPy's legacy generator, which is where the reference
draws its permutations (final_shapley_value.py:59-72 after tools/final_util.py:113-120), and the keep-mask constructors
against the host constructions (tools/final_common.py:56-60, final_point_binary_interaction_logits.py:45-52).  Bit-exact."""
import numpy as np
import pytest
import torch

from interpret_quality_amd import final_common, hip_ops, interaction

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def _numpy_permutations(s, r):
    return np.stack([np.random.permutation(np.arange(0, r, 1)) for _ in range(s)]) if s else np.zeros((0, r), dtype=np.int64)


@pytest.mark.parametrize("seed,s,r", [(1, 1000, 32), (1, 64, 8), (7, 300, 64), (3, 50, 2), (5, 40, 1), (2, 1, 33), (11, 5000, 32)])
def test_device_permutations_continue_numpys_stream(seed, s, r):
    """Same permutations as np.random.permutation drawn s times, and the generator state handed back equals NumPy's own
    state after those draws: whatever the host draws next (final_gen_pair.py's pairs and contexts) is unchanged."""
    np.random.seed(seed)
    np.random.rand(seed)                                  # start somewhere inside a 624-word block, as a running script does
    start = np.random.get_state()
    want = _numpy_permutations(s, r)
    after = np.random.get_state()
    tail_want = np.random.randint(0, 1 << 30, size=8)

    np.random.set_state(start)
    state = hip_ops.mt_state_to_device(dev())
    got = hip_ops.sample_permutations(state, s, r)
    st = hip_ops.mt_state_to_host(state, set_global=True)
    assert np.array_equal(got.cpu().numpy(), want)
    assert st[2] == after[2] and np.array_equal(st[1], after[1])
    assert np.array_equal(np.random.randint(0, 1 << 30, size=8), tail_want)


def test_a_cached_gaussian_of_the_host_generator_survives_device_sampling():
    """np.random.randn() draws Gaussians in pairs and caches the second (has_gauss, cached_gaussian of the legacy state): the device
    consumes 32-bit words only, so the cache must come back untouched."""
    np.random.seed(9)
    np.random.randn()                                     # leaves one Gaussian cached
    start = np.random.get_state()
    assert start[3] == 1
    want_perm = np.random.permutation(np.arange(32))
    want_next = np.random.randn(3)
    np.random.set_state(start)
    state = hip_ops.mt_state_to_device(dev())
    got = hip_ops.sample_permutations(state, 1, 32)
    hip_ops.mt_state_to_host(state, set_global=True)
    assert np.array_equal(got.cpu().numpy()[0], want_perm)
    assert np.array_equal(np.random.randn(3), want_next)


def test_device_permutations_in_several_calls_equal_one_call():
    """The state lives on the device between calls (bench.py keeps it there across steps: no host round trip)."""
    np.random.seed(4)
    want = _numpy_permutations(700, 32)
    np.random.seed(4)
    state = hip_ops.mt_state_to_device(dev())
    parts = [hip_ops.sample_permutations(state, n, 32) for n in (1, 99, 250, 350)]
    assert np.array_equal(torch.cat(parts).cpu().numpy(), want)


@pytest.mark.parametrize("s,r", [(1000, 32), (3, 8), (17, 64), (5, 1)])
def test_prefix_keep_masks_kernel_equals_the_host_construction(s, r):
    rng = np.random.default_rng(s * 100 + r)
    orders = np.stack([rng.permutation(r) for _ in range(s)]).astype(np.int64)
    want = final_common.prefix_keep_masks(orders, r)
    got = hip_ops.prefix_keep_masks(hip_ops.as_i32(orders, dev())).cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want)
    assert got[0] == 0 and got[r] == np.uint64((1 << r) - 1 if r < 64 else 0xFFFFFFFFFFFFFFFF)


@pytest.mark.parametrize("p,c,m,r", [(300, 100, 15, 32), (7, 1, 0, 32), (5, 1, 30, 32), (4, 6, 9, 64), (1, 1, 1, 8)])
def test_context_keep_masks_kernel_equals_the_host_construction(p, c, m, r):
    rng = np.random.default_rng(p + 10 * c + 100 * m)
    pairs = np.stack([rng.choice(r, size=2, replace=False) for _ in range(p)]).astype(np.int64)
    ctx = np.zeros((p, c, m), dtype=np.int64)
    for a in range(p):
        rest = np.setdiff1d(np.arange(r), pairs[a])
        for b in range(c):
            ctx[a, b] = rng.choice(rest, size=m, replace=False)
    want = interaction.context_keep_masks(pairs, ctx, r)
    got = hip_ops.context_keep_masks(hip_ops.as_i32(pairs, dev()), hip_ops.as_i32(ctx, dev())).cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want)


def test_gen_context_on_the_device_equals_the_host_loop(tmp_path):
    """final_gen_pair.py:18-43: the contexts np.random.choice(rest, m, replace=False) draws - the first m entries of a fresh
    permutation of the R - 2 other regions - continued on the device: every ratio's context list (shape, dtype, entries) and
    the generator state afterwards equal the reference's host loop."""
    import argparse
    from interpret_quality_amd import gen_pair
    a = argparse.Namespace(num_regions=32, num_pairs_random=40, num_save_context_max=100, ratio=interaction.DEFAULT_RATIOS)
    np.random.seed(1)
    pairs = gen_pair.gen_pair_random(a)
    start = np.random.get_state()
    gen_pair.gen_context(pairs, str(tmp_path) + "/host_", a)
    after = np.random.get_state()
    np.random.set_state(start)
    a.device = dev()
    gen_pair.gen_context(pairs, str(tmp_path) + "/dev_", a)
    got = np.random.get_state()
    assert got[2] == after[2] and np.array_equal(got[1], after[1])
    for ratio in a.ratio:
        h = np.load(str(tmp_path) + "/host_ratio%d_context_list.npy" % int(ratio * 100))
        d = np.load(str(tmp_path) + "/dev_ratio%d_context_list.npy" % int(ratio * 100))
        assert h.shape == d.shape and h.dtype == d.dtype and np.array_equal(h, d), ratio


def test_a_malformed_generator_state_is_refused_on_the_host_and_marked_on_the_device():
    """np.random.set_state accepts any position word; the generator itself never leaves 0..624.  The host wrapper refuses such a
    state; a device state tensor that carries one anyway (or that a failed call poisoned) makes iq_sample_permutations draw
    nothing: rows of -1 (every range check rejects them) and a state that mt_state_to_host refuses."""
    from interpret_quality_amd._lib import IqError
    np.random.seed(5)
    st = np.random.get_state()
    with pytest.raises(IqError, match="outside 0..624"):
        hip_ops.mt_state_to_device(dev(), ("MT19937", st[1], 700, st[3], st[4]))
    state = hip_ops.mt_state_to_device(dev(), st)
    state[624] = 1000
    orders = hip_ops.sample_permutations(state, 7, 32)
    assert bool((orders == -1).all())
    with pytest.raises(IqError, match="did not complete"):
        hip_ops.mt_state_to_host(state, set_global=False)
    with pytest.raises(IqError):
        hip_ops.check_index_range(orders, 0, 32, "orders")
