"""CPU: pin the oracle (oracle/policy_oracle.py) to the golden vectors produced by the real
reference (oracle/gen_golden.py).  Tolerance 5e-6 absolute on O(1) values (observed: bit-equal
up to ~1e-6: same op sequence, BLAS blocking differs with layout)."""
import numpy as np
import pytest
import torch

from oracle import policy_oracle as orc
from conftest import golden_state_dict, multilayer_state_dict

TOL = 5e-6


def _case(z, i, m):
    h = torch.from_numpy(z['c%d_h' % i])
    S = torch.from_numpy(z['c%d_S' % i])
    x = torch.from_numpy(z['c%d_x' % i])
    b = torch.from_numpy(z['c%d_b' % i]) if m['has_bias'] else None
    y = z['c%d_y' % i]
    return h, S, x, b, y


def test_lsigf_family_matches_reference(lsigf_golden):
    z, meta = lsigf_golden
    fn = {'LSIGF': orc.lsigf, 'BatchLSIGF': orc.batch_lsigf,
          'GraphFilter': lambda h, S, x, b: orc.graph_filter(h, b, S, x),
          'GraphFilterBatch': lambda h, S, x, b: orc.graph_filter_batch(h, b, S, x)}
    kinds = set()
    for i, m in enumerate(meta):
        h, S, x, b, y = _case(z, i, m)
        got = fn[m['kind']](h, S, x, b).numpy()
        assert got.shape == y.shape, (i, m)
        assert np.abs(got - y).max() <= TOL, (i, m, np.abs(got - y).max())
        kinds.add(m['kind'])
    assert kinds == set(fn)


def test_f64_einsum_statement_agrees(lsigf_golden):
    z, meta = lsigf_golden
    for i, m in enumerate(meta):
        if m['kind'] not in ('LSIGF', 'BatchLSIGF'):
            continue
        h, S, x, b, y = _case(z, i, m)
        ref = orc.lsigf_f64(h.numpy(), S.numpy(), x.numpy(), None if b is None else b.numpy())
        scale = max(1.0, np.abs(ref).max())
        assert np.abs(ref - y).max() <= 2e-5 * scale, (i, m)


def test_policy_matches_reference(policy_golden):
    z, meta = policy_golden
    for i, m in enumerate(meta):
        sd = golden_state_dict(z, m['K'])
        obs = torch.from_numpy(z['p%d_obs' % i])
        S = torch.from_numpy(z['p%d_S' % i])
        with torch.no_grad():
            out = orc.policy_forward(sd, S, obs)
            feat = orc.policy_features(sd, obs)
        assert len(out) == m['N'] and out[0].shape == (m['B'], 5)
        logits = torch.stack(out, dim=1).numpy()
        assert np.abs(feat.numpy() - z['p%d_feat' % i]).max() <= TOL
        assert np.abs(logits - z['p%d_logits' % i]).max() <= TOL, (i, m)
        want = torch.from_numpy(z['p%d_logits' % i]).argmax(-1)
        assert torch.equal(orc.decode_actions(out), want)


def test_policy_large_teams_match_reference(policy_golden, policy_large_golden):
    """BASELINE configs 3 / 5 shapes (50, 64, 100 agents; K = 2, 3, 4; fp64 / fp32 / sparse asymmetric GSOs):
    the oracle against logits computed by the reference itself."""
    zp, _ = policy_golden
    z, meta = policy_large_golden
    assert {m['N'] for m in meta} == {50, 64, 100} and {m['K'] for m in meta} == {2, 3, 4}
    for i, m in enumerate(meta):
        sd = golden_state_dict(zp, m['K'])
        obs = torch.from_numpy(z['q%d_obs' % i])
        S = torch.from_numpy(z['q%d_S' % i])
        with torch.no_grad():
            out = orc.policy_forward(sd, S, obs)
        logits = torch.stack(out, dim=1).numpy()
        assert logits.shape == (m['B'], m['N'], 5)
        assert np.abs(logits - z['q%d_logits' % i]).max() <= TOL, (i, m)
        clear = orc.top2_margin(out) > 1e-5
        want = torch.from_numpy(z['q%d_logits' % i]).argmax(-1)
        assert torch.equal(orc.decode_actions(out)[clear], want[clear])


def test_state_dict_contract(policy_golden):
    z, _ = policy_golden
    sd = golden_state_dict(z)
    mine = orc.init_state_dict(3)
    assert set(sd) == set(mine)
    for k in sd:
        assert tuple(sd[k].shape) == tuple(mine[k].shape), k
        assert sd[k].dtype == mine[k].dtype, k
    assert sum(v.numel() for k, v in sd.items()
               if 'running' not in k and 'num_batches' not in k) == 206501


def test_synth_gso_properties():
    S = orc.synth_gso_geometric(8, 10, 20, seed=3)
    assert S.dtype == np.float64 and S.shape == (8, 10, 10)
    assert np.allclose(S, S.transpose(0, 2, 1))
    assert (np.diagonal(S, axis1=1, axis2=2) == 0).all()
    deg = (S != 0).sum(-1)
    assert deg.min() >= 1                      # connected => no isolated node
    assert 2.0 < deg.mean() < 6.0              # ~3.4 under the reference rule at (10, 20x20)
    A = orc.synth_gso_sparse(4, 50, 6.6, seed=1)
    assert not torch.allclose(A, A.transpose(1, 2))


def test_multilayer_policy_matches_reference(policy_golden, multilayer_golden):
    """L = 2 graph-filter layers and / or E = 2 edge features (reference re-wired as editing
    decentralplanner.py:130-131 / :208 would): the oracle's layer loop against the reference's."""
    zp, _ = policy_golden
    zm, meta = multilayer_golden
    seen = set()
    for ci, m in enumerate(meta):
        sd = multilayer_state_dict(zp, zm, ci)
        obs = torch.from_numpy(zm['m%d_obs' % ci]).float()
        S = torch.from_numpy(zm['m%d_S' % ci])
        with torch.no_grad():
            out = orc.policy_forward(sd, S.squeeze(1) if m['E'] == 1 else S, obs)
        logits = torch.stack(out, dim=1).numpy()
        assert np.abs(logits - zm['m%d_logits' % ci]).max() <= TOL, (ci, m)
        seen.add((len(m['dims']), m['E']))
    assert {(2, 1), (1, 2), (2, 2)} <= seen


def test_bias_per_node_and_wide_cases_present(lsigf_golden):
    _, meta = lsigf_golden
    assert sum(1 for m in meta if m.get('bias_per_node')) >= 4
    assert sum(1 for m in meta if m['F'] > 128) >= 4
