"""CPU: the oracle's differentiable path pinned on the REFERENCE's own gradients.

tests/golden/grad_<case>.npz were recorded by tests/golden/gen_golden.py from the reference model in train mode:
cross-entropy of its logits, `loss.backward()` (engine.py:60-76), per parameter the L2 norm and <= 512 strided entries of
`p.grad`.  torch.autograd over the oracle's functional forward must reproduce them (fp32 on both sides: 1e-4 relative to the
parameter's gradient norm, i.e. summation-order noise only).  The GPU tests then compare the HIP backward with this oracle on
the device's own decisions (tests/test_hip_train.py).
"""
import os

import numpy as np
import pytest
import torch

from tests._params import GOLDEN_CASES, GRAD_CASES, grad_sample_index, oracle_param_grads

CPU_CASES = [n for n in GRAD_CASES if GOLDEN_CASES[n]["embed_dim"] <= 128] + ["topk_small_kr07"]


@pytest.mark.parametrize("name", CPU_CASES)
def test_oracle_gradients_match_the_reference(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"grad_{name}.npz"))
    loss, logits, grads = oracle_param_grads(GOLDEN_CASES[name])
    assert abs(loss - float(g["loss"])) <= 1e-5 * max(1.0, abs(float(g["loss"])))
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=0, atol=3e-5)
    names = [str(n) for n in g["param_names"]]
    assert set(names) == set(grads)
    for n in names:
        flat = grads[n].reshape(-1)
        ref_norm = float(g["norm:" + n])
        got_norm = float(flat.double().norm())
        assert abs(got_norm - ref_norm) <= 1e-4 * ref_norm + 1e-9, (n, got_norm, ref_norm)
        smp = flat[torch.from_numpy(grad_sample_index(flat.numel()))].numpy()
        err = np.abs(smp - g["sample:" + n]).max()
        assert err <= 1e-4 * ref_norm / np.sqrt(max(1, flat.numel())) * 30 + 1e-9, (n, err, ref_norm)
