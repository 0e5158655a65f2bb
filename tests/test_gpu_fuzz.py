"""A slice of tools/fuzz_trainer.py in the suite: random train-step configurations (ray / sample / hypothesis counts on
both sides of every tile and wave boundary, mask modes, lindisp, threshold, every precision) stepped by the default
Trainer, by the Trainer with every fusion switched off and by the HIP-graph replay; the exact rows also against
autograd through the oracle.  The 800-configuration run is profiles/r06_fuzz_trainer.json."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [0, 1, 2, 4, 7, 8, 11, 23])
def test_random_configuration_fused_plain_and_graphed_steps_agree(dev, seed):
    import fuzz_trainer as F
    c, g = F.config(seed)
    bs = F.batches(c, g, 3, dev)
    A, B, G = (F.run(c, bs, dev, m) for m in "ABG")
    assert torch.equal(A[1], G[1]) and torch.equal(A[2], G[2]) and A[0] == G[0], (c, "graph replay != eager")
    assert torch.equal(A[3], G[3]), (c, "first gradient: graph replay != eager")
    assert bool(torch.isfinite(A[1]).all() and torch.isfinite(A[3]).all()), c
    assert abs(A[0][0] - B[0][0]) <= 1e-6 * abs(B[0][0]), (c, A[0], B[0])
    assert F.rel_l2(A[3], B[3]) < F.PARAM_BAR[c["precision"]], (c, F.rel_l2(A[3], B[3]))


@pytest.mark.parametrize("seed", [5, 6, 356])
def test_random_exact_configuration_vs_oracle_autograd(dev, seed):
    """Loss to 2e-4 and the coarse network's gradient to 1e-3 against autograd through the oracle; seeds 5 and 356 are
    the two of 197 oracle-checked configurations that miss the bar because a ReLU sits within rounding of zero (3.4e-3 /
    4.9e-3) - and meet it at 1e-6 once the biases of those units are moved 1e-5 off zero on both sides."""
    import fuzz_trainer as F
    c, g = F.config(seed)
    assert c["precision"] == "f32" and c["N"] * (c["Ns"] + c["Ni"]) <= 40000, c
    bs = F.batches(c, g, 1, dev)
    loss_rel, grad_rel, nudged, after = F.oracle_check(c, bs, dev)
    assert loss_rel < 2e-4, (c, loss_rel)
    assert grad_rel < 1e-3 or (nudged > 0 and after < 1e-5), (c, grad_rel, nudged, after)
    if seed in (5, 356):
        assert nudged > 0 and after < 1e-5, (grad_rel, nudged, after)


@pytest.mark.parametrize("seed", [3, 4, 8, 11, 14, 19, 25, 370])
def test_random_render_configuration_vs_oracle_chunked_graphed_and_fast(dev, seed):
    """tools/fuzz_render.py: the exact render of a random configuration against the oracle (coarse stage element-wise,
    the rest by PSNR / norm - seed 370, where one ray of 222 carries the depth norm, by its distance to an fp64
    evaluation), chunked and graph-replayed renders bit for bit, the fast inference precision against the exact one.
    The 800-configuration run is profiles/r06_fuzz_render.json."""
    import fuzz_render as F
    c, g = F.config(seed)
    row = F.one(c, g, dev, True)
    assert "rgb_psnr" in row, "a configuration the oracle finishes in seconds"
    assert row["ok"], row
    assert row["z_vals0_bitwise"] and row["coarse_elementwise_excess"] <= 0
    if seed == 370:
        assert "vs_fp64" in row
