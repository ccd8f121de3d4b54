"""CPU: the oracle's train-mode forward + torch autograd reproduces the REAL reference's loss,
logits, gradients and BatchNorm running statistics (tests/golden/training_grads.npz)."""
import numpy as np
import torch

from conftest import golden_state_dict
from oracle import policy_oracle as orc


def test_policy_training_step_matches_reference(training_golden, policy_golden):
    z, meta = training_golden
    zp, _ = policy_golden
    for ci, m in enumerate(meta):
        if m['kind'] != 'policy':
            continue
        sd = golden_state_dict(zp, m['K'])
        sd = {k: v.clone() for k, v in sd.items()}
        params = {k: v.requires_grad_(True) for k, v in sd.items()
                  if v.dtype == torch.float32 and 'running' not in k}
        sd.update(params)
        obs = torch.from_numpy(z['g%d_obs' % ci].astype(np.float32))
        S = torch.from_numpy(z['g%d_S' % ci])
        tgt = torch.from_numpy(z['g%d_target' % ci].astype(np.float32))
        out = orc.policy_forward(sd, S, obs, training=True)
        loss = orc.policy_loss(out, tgt)
        loss.backward()
        assert abs(loss.item() - float(z['g%d_loss' % ci])) <= 1e-6
        assert np.abs(torch.stack(out, 1).detach().numpy() - z['g%d_logits' % ci]).max() <= 5e-6
        for j, name in enumerate(m['param_names']):
            g = sd[name].grad.double()
            want = z['g%d_gradsum' % ci][j]
            assert abs(g.norm().item() - want[2]) <= 1e-5 * max(1.0, want[2]), name
            key = 'g%d_grad/%s' % (ci, name)
            if key in z.files:
                assert np.abs(sd[name].grad.numpy() - z[key]).max() <= 2e-6 * max(1.0, np.abs(z[key]).max()), name
        for key in z.files:
            if key.startswith('g%d_buf/' % ci) and 'num_batches' not in key:
                name = key.split('/', 1)[1]
                assert np.abs(sd[name].detach().numpy() - z[key]).max() <= 1e-6, name
