"""GPU parity tests (run with -m gpu on the MI355X): the HIP path, called through the C ABI via
the reference-shaped Python modules, against (i) the golden vectors produced by the real
reference and (ii) the CPU oracle on seeded inputs at the BASELINE.json sizes.

Tolerances (north_star): logits / filter outputs within 1e-4 absolute (fp32), action argmax
bit-exact on every row whose oracle top-2 margin exceeds 1e-5."""
import numpy as np
import pytest
import torch

from conftest import golden_state_dict
from oracle import policy_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need the GPU box'
    from gnn_pathplanning_amd import _native
    _native.lib()                      # fail loudly if libgnnpp.so is missing
    return torch.device('cuda:0')


@pytest.fixture(params=['fp32', 'fp32_mfma', 'split_f16'], ids=['prec-bf16x3', 'prec-fp32-mfma', 'prec-split-f16'])
def enc_variant(request, dev):
    """Run a test under all three arithmetics (include/gnnpp.h GNNPP_PREC_*): the planners a test builds take the
    precision from the module default, which this fixture sets: 'fp32' = bf16x3 (the shipped default),
    'fp32_mfma' = exact fp32 MFMA, 'split_f16' = the opt-in fast mode."""
    import gnn_pathplanning_amd.decentralplanner as dp
    old = dp.DEFAULT_PRECISION
    dp.DEFAULT_PRECISION = request.param
    yield request.param
    dp.DEFAULT_PRECISION = old


class Cfg:
    def __init__(self, n, k, device):
        self.num_agents, self.nGraphFilterTaps, self.device = n, k, device


def _net(n, k, dev, sd):
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    net = DecentralPlannerNet(Cfg(n, k, dev)).to(dev).eval()
    net.load_state_dict(sd)
    return net


def test_lsigf_golden_all_cases(dev, lsigf_golden):
    import gnn_pathplanning_amd.graphML as gml
    z, meta = lsigf_golden
    for i, m in enumerate(meta):
        h = torch.from_numpy(z['c%d_h' % i]).to(dev)
        S = torch.from_numpy(z['c%d_S' % i]).to(dev)
        x = torch.from_numpy(z['c%d_x' % i]).to(dev)
        b = torch.from_numpy(z['c%d_b' % i]).to(dev) if m['has_bias'] else None
        want = z['c%d_y' % i]
        if m['kind'] == 'LSIGF':
            y = gml.LSIGF(h, S, x, b)
        elif m['kind'] == 'BatchLSIGF':
            y = gml.BatchLSIGF(h, S, x, b)
        else:
            cls = gml.GraphFilter if m['kind'] == 'GraphFilter' else gml.GraphFilterBatch
            mod = cls(m['G'], m['F'], m['K'], m['E'], True).to(dev)
            with torch.no_grad():
                mod.weight.copy_(h)
                mod.bias.copy_(b)
            mod.addGSO(S)
            y = mod(x)
        assert tuple(y.shape) == want.shape, (i, m)
        err = np.abs(y.detach().cpu().numpy() - want).max()
        assert err <= TOL * max(1.0, np.abs(want).max()), (i, m, err)


def test_policy_golden(dev, policy_golden, enc_variant):
    z, meta = policy_golden
    for i, m in enumerate(meta):
        sd = golden_state_dict(z, m['K'])
        net = _net(m['N'], m['K'], dev, sd)
        obs = torch.from_numpy(z['p%d_obs' % i]).to(dev)
        S = torch.from_numpy(z['p%d_S' % i]).to(dev)
        net.addGSO(S)
        out = net(obs)
        assert isinstance(out, list) and len(out) == m['N'] and out[0].shape == (m['B'], 5)
        assert all(o.is_contiguous() for o in out)
        got = torch.stack(out, 1).cpu().numpy()
        want = z['p%d_logits' % i]
        assert np.abs(got - want).max() <= TOL, (i, m, np.abs(got - want).max())
        assert (got.argmax(-1) == want.argmax(-1)).all()
        feat = net.encode(obs).cpu().numpy()            # [B,N,128]
        assert np.abs(feat - z['p%d_feat' % i].transpose(0, 2, 1)).max() <= TOL
        acts = net.decode_actions(net.forward_logits(obs)).cpu().numpy()
        assert (acts == want.argmax(-1)).all()


@pytest.mark.parametrize('B,N,W', [(1, 10, 20), (16, 100, 100), (6, 7, 12)])
def test_forward_list_is_a_real_list_and_is_recycled_only_when_dropped(dev, B, N, W):
    """the reference's return type (list of N tensors [B,5], decentralplanner.py:304-318) costs the
    host ~30 us per step at N = 100.  forward() re-uses the previous step's buffer and view objects when the caller has
    dropped them (decentralplanner._OutputSlot) and is otherwise exactly what it was: a real list (C-level consumers
    such as torch.stack work), contiguous [B,5] elements, held results never overwritten."""
    K = 3
    sd = orc.init_state_dict(K, seed=11)
    net = _net(N, K, dev, sd)
    obs = [orc.synth_obs(B, N, seed=s).to(dev) for s in (1, 2, 3)]
    S = [torch.from_numpy(orc.synth_gso_geometric(B, N, W, seed=s)).float().to(dev) for s in (1, 2, 3)]

    def run(i):
        net.addGSO(S[i])
        return net(obs[i])

    def ref(i):
        net.addGSO(S[i])
        return net.forward_logits(obs[i]).clone()
    want = [ref(i) for i in range(3)]
    slot = net._out_slot
    out = run(0)
    assert type(out) is list and len(out) == N and out[0].shape == (B, 5) and all(o.is_contiguous() for o in out)
    assert torch.equal(torch.stack(out, 1), want[0].permute(1, 0, 2)) and torch.equal(torch.stack(list(iter(out))), want[0])
    assert out[-1].data_ptr() == out[0].data_ptr() + (N - 1) * B * 5 * 4 and len(out[2:5]) == min(3, max(0, N - 2))
    held, p_held = out, out[0].data_ptr()
    out1 = run(1)                                           # `held` is alive: fresh memory
    assert out1[0].data_ptr() != p_held
    assert torch.equal(torch.stack(held), want[0]) and torch.equal(torch.stack(out1), want[1])
    p1 = out1[0].data_ptr()
    del out, out1
    f0, r0 = slot.fresh, slot.recycled
    out2 = run(2)                                           # out1 was dropped: its buffer and view objects again
    assert (slot.fresh, slot.recycled) == (f0, r0 + 1) and out2[0].data_ptr() == p1
    assert torch.equal(torch.stack(out2), want[2]) and torch.equal(torch.stack(held), want[0])
    one = out2[N // 2]                                      # a single element kept: never overwritten
    del out2
    out3 = run(0)
    assert slot.fresh == f0 + 1 and torch.equal(one, want[2][N // 2]) and torch.equal(torch.stack(out3), want[0])
    del out3, one, held
    for i in (1, 2, 0, 1):                                  # the steady state of a rollout loop: every step recycled
        o = run(i)
        assert torch.equal(torch.stack(o), want[i])
        del o
    assert slot.fresh <= f0 + 2
    # a HIP-graph capture never takes a recycled (non-pool) buffer, and what it leaves behind is not recycled into
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run(0)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    n_entries = len(slot.entries)
    with torch.cuda.graph(g):
        og = run(1)
    assert len(slot.entries) == n_entries
    eager = run(2)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(torch.stack(og), want[1]) and torch.equal(torch.stack(eager), want[2])
    assert all(a.data_ptr() != b.data_ptr() for a, b in zip(og, eager))


def test_policy_large_teams_golden(dev, policy_golden, policy_large_golden, enc_variant):
    """Teams of 50 / 64 / 100 agents against logits computed by the REFERENCE (tests/golden/policy_large.npz): the
    two-kernel policy step with policy_filter_kernel (default) and with the general filter kernel."""
    from gnn_pathplanning_amd import _native
    L = _native.lib()
    zp, _ = policy_golden
    z, meta = policy_large_golden
    try:
        for i, m in enumerate(meta):
            net = _net(m['N'], m['K'], dev, golden_state_dict(zp, m['K']))
            obs = torch.from_numpy(z['q%d_obs' % i]).to(dev)
            S = torch.from_numpy(z['q%d_S' % i]).to(dev)
            want = z['q%d_logits' % i]
            net.addGSO(S)
            for mode in (1, 0):
                assert L.gnnpp_set_tuning(9, mode) == 0
                got = torch.stack(net(obs), 1).cpu().numpy()
                assert np.abs(got - want).max() <= TOL, (i, m, mode, np.abs(got - want).max())
                srt = np.sort(want, -1)
                clear = srt[..., -1] - srt[..., -2] > 1e-5
                assert (got.argmax(-1)[clear] == want.argmax(-1)[clear]).all()
    finally:
        L.gnnpp_set_tuning(9, 1)


@pytest.mark.parametrize('B,N,K,W', [(1, 10, 3, 20), (512, 10, 3, 20), (256, 50, 3, 50),
                                     (128, 100, 2, 100), (128, 100, 3, 100), (128, 100, 4, 100),
                                     (37, 7, 3, 12), (5, 64, 3, 40)])
def test_policy_vs_oracle_baseline_sizes(dev, B, N, K, W, enc_variant):
    sd = orc.init_state_dict(K, seed=1337 + K)
    net = _net(N, K, dev, sd)
    obs = orc.synth_obs(B, N, seed=B + N)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, W, seed=N + K))     # fp64 like the simulator
    with torch.no_grad():
        want = orc.policy_forward(sd, S, obs)
    net.addGSO(S.to(dev))
    got = [g.cpu() for g in net(obs.to(dev))]
    err = max((g - w).abs().max().item() for g, w in zip(got, want))
    assert err <= TOL, err
    margin = orc.top2_margin(want)
    ids_want = orc.decode_actions(want)
    ids_got = torch.stack([g.argmax(-1) for g in got], 1)
    clear = margin > 1e-5
    assert torch.equal(ids_got[clear], ids_want[clear])
    assert clear.float().mean() > 0.99
    # fp32 GSO (training-style input) gives the same answer as the fp64 one rounded on load
    net.addGSO(S.float().to(dev))
    got32 = [g.cpu() for g in net(obs.to(dev))]
    assert max((a - b).abs().max().item() for a, b in zip(got, got32)) <= 1e-6


@pytest.mark.parametrize('B,N,K', [(512, 10, 3), (256, 50, 3), (128, 100, 4)])
def test_filter_vs_oracle_asymmetric_gso(dev, B, N, K):
    import gnn_pathplanning_amd.graphML as gml
    g = torch.Generator().manual_seed(B + N + K)
    h = (torch.rand(128, 1, K, 128, generator=g) * 2 - 1) / (128 * K) ** 0.5
    b = torch.randn(128, 1, generator=g) * 0.1
    x = orc.synth_features(B, 128, N, seed=N)
    S = orc.synth_gso_sparse(B, N, {10: 3.45, 50: 6.6, 100: 8.4}[N], seed=K).unsqueeze(1)
    want = orc.batch_lsigf(h, S, x, b)
    y = gml.BatchLSIGF(h.to(dev), S.to(dev), x.to(dev), b.to(dev)).cpu()
    scale = max(1.0, want.abs().max().item())
    assert (y - want).abs().max().item() <= TOL * scale
    # the transposed GSO must give a DIFFERENT answer (column gather, not row gather)
    yT = gml.BatchLSIGF(h.to(dev), S.transpose(2, 3).contiguous().to(dev), x.to(dev), b.to(dev)).cpu()
    assert (yT - want).abs().max().item() > 1e-2


@pytest.mark.parametrize('N,E,Nin', [(130, 1, 130), (150, 2, 141)])
def test_filter_larger_than_one_workgroup(dev, N, E, Nin):
    """Graphs with more nodes than one workgroup's LDS holds (N > 112) run as dense exact-fp32 GEMMs
    (graphML._lsigf_large on gnnpp_gemm_kmajor): BatchLSIGF (fp64 GSO, several edge features, Nin < N through the
    module), LSIGF with one shared GSO, and the per-node bias -- against the oracle."""
    import gnn_pathplanning_amd.graphML as gml
    g = torch.Generator().manual_seed(N + E)
    B, K, G, F_out = 3, 3, 48, 40
    h = (torch.rand(F_out, E, K, G, generator=g) * 2 - 1) / (G * K * E) ** 0.5
    b = torch.randn(F_out, 1, generator=g) * 0.1
    bn = torch.randn(F_out, N, generator=g) * 0.1
    x = orc.synth_features(B, G, N, seed=N)
    S = torch.stack([orc.synth_gso_sparse(B, N, 8.0, seed=e + 1) for e in range(E)], 1)       # [B,E,N,N]
    want = orc.batch_lsigf(h, S, x, b)
    y = gml.BatchLSIGF(h.to(dev), S.double().to(dev), x.to(dev), b.to(dev)).cpu()
    assert (y - want).abs().max().item() <= TOL * max(1.0, want.abs().max().item())
    want_n = orc.batch_lsigf(h, S, x, bn)
    y_n = gml.BatchLSIGF(h.to(dev), S.to(dev), x.to(dev), bn.to(dev)).cpu()
    assert (y_n - want_n).abs().max().item() <= TOL * max(1.0, want_n.abs().max().item())
    want_s = orc.lsigf(h, S[0], x, b)                                                          # one GSO for the batch
    y_s = gml.LSIGF(h.to(dev), S[0].to(dev), x.to(dev), b.to(dev)).cpu()
    assert (y_s - want_s).abs().max().item() <= TOL * max(1.0, want_s.abs().max().item())
    gf = gml.GraphFilterBatch(G, F_out, K, E).to(dev)
    with torch.no_grad():
        gf.weight.copy_(h); gf.bias.copy_(b)
    gf.addGSO(S.to(dev))
    with torch.no_grad():
        y_m = gf(x[:, :, :Nin].to(dev)).cpu()                                                  # zero-padded nodes
    xp = torch.zeros_like(x); xp[:, :, :Nin] = x[:, :, :Nin]
    want_m = orc.batch_lsigf(h, S, xp, b)[:, :, :Nin]
    assert y_m.shape == (B, F_out, Nin)
    assert (y_m - want_m).abs().max().item() <= TOL * max(1.0, want_m.abs().max().item())


def test_planner_with_more_agents_than_one_workgroup(dev):
    """DecentralPlannerNet with 120 agents (the reference has no limit): encoder kernel + dense-GEMM filter + head
    against the oracle; identical actions on clear rows."""
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    B, N = 2, 120

    class Cfg:
        num_agents, nGraphFilterTaps, device = N, 3, dev
    sd = orc.init_state_dict(3, seed=21)
    net = DecentralPlannerNet(Cfg()).to(dev).eval()
    net.load_state_dict(sd)
    obs = orc.synth_obs(B, N, seed=4)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, 100, seed=4))                           # float64
    with torch.no_grad():
        want = torch.stack(orc.policy_forward(sd, S, obs), 0)                                  # [N,B,5]
        net.addGSO(S.to(dev))
        out = net(obs.to(dev))
    got = torch.stack([o.cpu() for o in out], 0)
    assert len(out) == N and (got - want).abs().max().item() <= 1e-4
    clear = orc.top2_margin(list(want)) > 1e-5
    assert torch.equal(got.argmax(-1).t()[clear], want.argmax(-1).t()[clear])


def test_filter_algebra_properties(dev):
    import gnn_pathplanning_amd.graphML as gml
    g = torch.Generator().manual_seed(99)
    B, G, F_out, N = 6, 128, 128, 10
    x = torch.randn(B, G, N, generator=g).to(dev)
    S = orc.synth_gso_sparse(B, N, 3.0, seed=1).unsqueeze(1).to(dev)
    h = (torch.randn(F_out, 1, 3, G, generator=g) / 20).to(dev)
    # K = 1: pure per-node linear map, GSO irrelevant
    y1 = gml.BatchLSIGF(h[:, :, :1].contiguous(), S, x)
    ref1 = torch.einsum('fg,bgn->bfn', h[:, 0, 0].cpu(), x.cpu())
    assert (y1.cpu() - ref1).abs().max() <= TOL
    # S = 0: only tap 0 survives
    y0 = gml.BatchLSIGF(h, torch.zeros_like(S), x)
    assert (y0.cpu() - ref1).abs().max() <= TOL
    # linearity in x
    x2 = torch.randn(B, G, N, generator=g).to(dev)
    ya, yb, yab = gml.BatchLSIGF(h, S, x), gml.BatchLSIGF(h, S, x2), gml.BatchLSIGF(h, S, x + x2)
    assert (ya + yb - yab).abs().max().item() <= 5e-5
    # permutation equivariance: relabel the nodes of every graph
    perm = torch.randperm(N, generator=g).to(dev)
    Sp = S[:, :, perm][:, :, :, perm].contiguous()
    yp = gml.BatchLSIGF(h, Sp, x[:, :, perm].contiguous())
    assert (yp - ya[:, :, perm]).abs().max().item() <= 5e-5
    # shared-GSO LSIGF == BatchLSIGF with the GSO repeated
    S1 = S[0]
    yl = gml.LSIGF(h, S1, x)
    yb2 = gml.BatchLSIGF(h, S1.unsqueeze(0).repeat(B, 1, 1, 1), x)
    assert (yl - yb2).abs().max().item() <= 1e-6


def test_encoder_ragged_tiles(dev, enc_variant):
    """M = B*N not a multiple of the 16-agent tile, including a single agent."""
    sd = orc.init_state_dict(3, seed=3)
    for B, N in ((1, 1), (1, 17), (3, 11), (2, 16)):
        net = _net(N, 3, dev, sd)
        obs = orc.synth_obs(B, N, seed=B * 31 + N)
        want = orc.policy_features(sd, obs).permute(0, 2, 1)
        got = net.encode(obs.to(dev)).cpu()
        assert (got - want).abs().max().item() <= TOL, (B, N)


def test_non_binary_observations(dev, enc_variant):
    """The kernel must not assume {0,1} inputs."""
    sd = orc.init_state_dict(3, seed=5)
    net = _net(4, 3, dev, sd)
    g = torch.Generator().manual_seed(0)
    obs = torch.randn(3, 4, 3, 11, 11, generator=g)
    want = orc.policy_features(sd, obs).permute(0, 2, 1)
    got = net.encode(obs.to(dev)).cpu()
    assert (got - want).abs().max().item() <= 2e-4 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize('prec', ['fp32', 'split_f16'])
@pytest.mark.parametrize('B,N,W,K', [(512, 10, 20, 3), (33, 16, 24, 3), (5, 1, 8, 3), (64, 7, 12, 3), (600, 14, 20, 3),
                                     (2048, 10, 20, 3), (512, 10, 20, 2), (33, 16, 24, 4), (64, 7, 12, 4),
                                     (5, 3, 8, 2)])
def test_fused_policy_kernel_equals_two_kernels(dev, B, N, W, K, prec):
    """For N <= 16 and K = 2, 3, 4 taps the policy step is ONE kernel (a workgroup per graph: encoder, dense-MFMA
    shifts, tap contraction, head).  Split-f16: the same arithmetic in the same order as the encoder kernel + filter
    kernel, so the logits must be identical.  bf16x3 (default): identical encoder, but the two-kernel path contracts
    the taps on the exact fp32 MFMA and the fused one on bf16x3 planes -- equal to a few ulps.  fp64 and fp32 GSOs."""
    from gnn_pathplanning_amd import _native
    L = _native.lib()
    sd = orc.init_state_dict(K, seed=31)
    net = _net(N, K, dev, sd)
    net.precision = prec
    obs = orc.synth_obs(B, N, seed=B + 3 * N).to(dev)
    S64 = torch.from_numpy(orc.synth_gso_geometric(B, N, W, seed=N))
    try:
        for S in (S64.to(dev), S64.float().to(dev)):
            net.addGSO(S)
            outs = []
            for mode in (1, 0, 1):
                assert L.gnnpp_set_tuning(6, mode) == 0
                outs.append(net.forward_logits(obs).clone())
            assert torch.equal(outs[0], outs[2])
            if prec == 'split_f16':
                assert torch.equal(outs[0], outs[1]), (outs[0] - outs[1]).abs().max().item()
            else:
                assert (outs[0] - outs[1]).abs().max().item() <= 2e-6 * max(1.0, outs[1].abs().max().item())
    finally:
        L.gnnpp_set_tuning(6, 1)
    want = torch.stack(orc.policy_forward(sd, S64.float(), obs.cpu()), 0)
    assert (outs[0].cpu() - want).abs().max().item() <= TOL


@pytest.mark.parametrize('B,N,K,real_obs', [(512, 10, 3, False), (37, 12, 3, False), (64, 1, 2, False), (9, 5, 4, True),
                                            (130, 9, 3, True), (33, 11, 2, False), (65, 7, 3, False), (512, 8, 4, False),
                                            (17, 2, 3, False), (256, 10, 3, True)])
def test_column_packed_policy_kernel_is_bit_identical(dev, B, N, K, real_obs):
    """for teams of <= 12 agents the one-launch policy kernel puts (agent, position) pairs on the
    MFMA columns of its two 5x5 layers (GNNPP_TUNE_POLICY_CP, default on).  Every logit must keep its BITS: each
    output sums the same plane products in the same order, a tap that leaves the image adds exact zeros.  All team-size
    classes (waves without a tile at N = 1 / 2, a last tile with missing columns, 19 tiles at N = 12), K = 2, 3, 4,
    fp64 and fp32 GSOs, binary and real-valued (three-plane L0) observations; and against the oracle."""
    from gnn_pathplanning_amd import _native
    L = _native.lib()
    sd = orc.init_state_dict(K, seed=40 + N)
    net = _net(N, K, dev, sd)
    obs = orc.synth_obs(B, N, seed=B + N)
    if real_obs:
        obs = obs * torch.randn(obs.shape, generator=torch.Generator().manual_seed(N))
    S64 = torch.from_numpy(orc.synth_gso_geometric(B, N, 20, seed=N + K))
    obs_d = obs.to(dev)
    try:
        for S in (S64.to(dev), S64.float().to(dev)):
            net.addGSO(S)
            outs = []
            for cp in (1, 0, 1):
                assert L.gnnpp_set_tuning(13, cp) == 0 and L.gnnpp_get_tuning(13) == cp
                outs.append(net.forward_logits(obs_d).clone())
            assert torch.isfinite(outs[0]).all()
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (outs[0] - outs[1]).abs().max().item()
    finally:
        L.gnnpp_set_tuning(13, 1)
    want = torch.stack(orc.policy_forward(sd, S64.float(), obs), 0)
    assert (outs[0].cpu() - want).abs().max().item() <= TOL * max(1.0, want.abs().max().item())


@pytest.mark.parametrize('M,tile', [(1600, 0), (1600, 7), (23, 5), (3000, 12), (257, 0), (640, 3), (40, 1)])
def test_column_packed_encoder_tiles_are_bit_identical(dev, M, tile):
    """GNNPP_TUNE_ENCODER_CP_TILE: the unfused encoder's LATENCY form -- column-packed tiles of ceil(M / 256) <= 12
    agents, one per CU, for launches of at most 2048 agents (the per-GPU shards of the 8-GPU configs: 16 graphs of
    100 agents) -- writes the same features, bit for bit, as 16-agent tiles; ragged last tiles, unaligned tile starts,
    binary and real-valued observations; and equals the oracle."""
    import ctypes
    from gnn_pathplanning_amd import _native
    L = _native.lib()
    sd = orc.init_state_dict(3, seed=60)
    net = _net(10, 3, dev, sd)
    enc = net.packed_encoder()
    obs = orc.synth_obs(1, M, seed=M)
    if M % 2:
        obs = obs * torch.randn(obs.shape, generator=torch.Generator().manual_seed(M))
    obs_d = obs.reshape(M, 3, 11, 11).contiguous().to(dev)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())                # noqa: E731
    feats = []
    try:
        for knob in (tile, 16, tile):
            assert L.gnnpp_set_tuning(14, knob) == 0
            feat = torch.full((M, 128), float('nan'), device=dev)
            assert L.gnnpp_encoder_fwd(vp(obs_d), vp(enc), vp(feat), M, 0, None, _native.stream_ptr(dev)) == 0
            feats.append(feat)
    finally:
        L.gnnpp_set_tuning(14, 0)
    assert torch.isfinite(feats[0]).all()
    assert torch.equal(feats[0], feats[1]) and torch.equal(feats[0], feats[2]), (feats[0] - feats[1]).abs().max().item()
    want = orc.policy_features(sd, obs).permute(0, 2, 1).reshape(M, 128)
    assert (feats[0].cpu() - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())


def test_encoder_negative_and_zero_batchnorm_scales(dev, enc_variant):
    """A trained BatchNorm may have gamma < 0 or = 0.  The bf16x3 L0 pools its RAW accumulators and applies the affine
    map once per window (the sign of the folded scale lives in the packed weights, |scale| in the table): must equal
    the reference, binary and real-valued observations, every precision."""
    sd = orc.init_state_dict(3, seed=14)
    g = torch.Generator().manual_seed(2)
    for bn, c_n in (('ConvLayers.1', 32), ('ConvLayers.5', 32), ('ConvLayers.8', 64)):
        sgn = (torch.rand(c_n, generator=g) < 0.5).float() * 2 - 1
        sd[bn + '.weight'] = sd[bn + '.weight'].abs() * sgn
        sd[bn + '.weight'][3] = 0.0
    for (B, N), binary in (((4, 10), True), ((1, 23), False)):
        net = _net(N, 3, dev, sd)
        obs = orc.synth_obs(B, N, seed=N)
        if not binary:
            obs = obs * torch.randn(obs.shape, generator=g)
        want = orc.policy_features(sd, obs).permute(0, 2, 1)
        got = net.encode(obs.to(dev)).cpu()
        assert (got - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item()), (B, N, binary)


def test_encoder_dynamic_range(dev, enc_variant):
    """Weights and activations spread over several decades (the split-f16 schedule rescales the
    weights per layer and keeps subnormal lo halves): the relative error must stay at fp32 level."""
    sd = orc.init_state_dict(3, seed=11)
    for name, f in (('ConvLayers.0.weight', 30.0), ('ConvLayers.4.weight', 0.004),
                    ('ConvLayers.7.weight', 55.0), ('ConvLayers.11.weight', 0.02),
                    ('ConvLayers.14.weight', 7.0), ('compressMLP.0.weight', 0.3)):
        assert name in sd, sorted(sd)
        sd[name] = sd[name] * f
    net = _net(6, 3, dev, sd)
    g = torch.Generator().manual_seed(2)
    obs = torch.randn(5, 6, 3, 11, 11, generator=g) * 3.0
    want = orc.policy_features(sd, obs).permute(0, 2, 1)
    got = net.encode(obs.to(dev)).cpu()
    scale = want.abs().max().item()
    assert scale > 0 and torch.isfinite(got).all()
    assert (got - want).abs().max().item() <= 2e-5 * scale, ((got - want).abs().max().item(), scale)


def test_state_dict_roundtrip_and_cache_invalidation(dev, policy_golden):
    z, meta = policy_golden
    sd = golden_state_dict(z, 3)
    net = _net(10, 3, dev, sd)
    got_sd = net.state_dict()
    assert list(got_sd.keys()) == list(sd.keys())
    for k in sd:
        assert torch.equal(got_sd[k].cpu(), sd[k]), k
    obs = torch.from_numpy(z['p0_obs']).to(dev)
    S = torch.from_numpy(z['p0_S']).to(dev)
    net.addGSO(S)
    a = torch.stack(net(obs), 1)
    # in-place parameter update (what an optimizer step or load_state_dict does) must repack
    sd2 = orc.init_state_dict(3, seed=123)
    net.load_state_dict(sd2)
    b = torch.stack(net(obs), 1).cpu()
    with torch.no_grad():
        want = torch.stack(orc.policy_forward(sd2, S.cpu(), obs.cpu()), 1)
    assert (b - want).abs().max().item() <= TOL
    assert (a.cpu() - b).abs().max().item() > 1e-3
    with torch.no_grad():
        net.ConvLayers[1].running_mean.add_(0.5)          # a BN buffer change must repack too
    c = torch.stack(net(obs), 1).cpu()
    assert (c - b).abs().max().item() > 1e-4


def test_gso_larger_than_num_agents(dev):
    """Nin < N: the reference zero-pads the missing nodes (graphML.py:2464-2476)."""
    sd = orc.init_state_dict(3, seed=8)
    net = _net(6, 3, dev, sd)
    obs = orc.synth_obs(2, 6, seed=1)
    S = orc.synth_gso_sparse(2, 9, 3.0, seed=2)
    with torch.no_grad():
        feat = orc.policy_features(sd, obs)
        shared = torch.relu(orc.graph_filter_batch(sd['GFL.0.weight'], sd['GFL.0.bias'],
                                                   S.unsqueeze(1), feat))
        want = torch.stack([torch.nn.functional.linear(shared[:, :, n], sd['actionsMLP.0.weight'],
                                                       sd['actionsMLP.0.bias']) for n in range(6)], 1)
    net.addGSO(S.to(dev))
    got = torch.stack(net(obs.to(dev)), 1).cpu()
    assert (got - want).abs().max().item() <= TOL


def test_error_conventions(dev):
    import gnn_pathplanning_amd.graphML as gml
    from gnn_pathplanning_amd import _native
    sd = orc.init_state_dict(3)
    net = _net(10, 3, dev, sd)
    with pytest.raises(AssertionError):
        net.addGSO(torch.zeros(2, 1, 10, 10, device=dev))            # must be 3-D when E == 1
    with pytest.raises(TypeError):
        _net(10, 3, dev, sd)(torch.zeros(1, 10, 3, 11, 11, device=dev))   # no GSO yet
    net.addGSO(torch.zeros(1, 10, 10, device=dev))
    with pytest.raises(_native.GnnppError):
        net(torch.zeros(1, 10, 3, 11, 11))                          # CPU tensor: no fallback
    net.train()                                                     # train mode: differentiable
    net.addGSO(torch.zeros(2, 10, 10, device=dev))
    out = net(torch.zeros(2, 10, 3, 11, 11, device=dev))
    assert isinstance(out, list) and len(out) == 10 and out[0].requires_grad
    gf = gml.GraphFilterBatch(8, 8, 2).to(dev)
    with pytest.raises(AssertionError):
        gf.addGSO(torch.zeros(10, 10, device=dev))
    with pytest.raises(RuntimeError):
        gml.LSIGF(torch.zeros(4, 1, 2, 4, device=dev), torch.zeros(1, 5, 5, device=dev).double(),
                  torch.zeros(1, 4, 5, device=dev))


def test_multilayer_and_edge_feature_planners(dev, policy_golden, multilayer_golden, enc_variant):
    """L = 2 graph-filter layers / E = 2 edge features (the generality decentralplanner.py:205-224,
    266-276, 293-315 has; reference re-wired to produce tests/golden/policy_multilayer.npz) through
    DecentralPlannerNet's optional config fields."""
    from conftest import multilayer_state_dict
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    zp, _ = policy_golden
    zm, meta = multilayer_golden
    for ci, m in enumerate(meta):
        class C:
            num_agents, nGraphFilterTaps, device = m['N'], list(m['taps']), dev
            dimNodeSignals, numEdgeFeatures = list(m['dims']), m['E']
        net = DecentralPlannerNet(C()).to(dev).eval()
        assert net.L == len(m['dims']) and net.E == m['E'] and net.F == [128] + m['dims']
        sd = multilayer_state_dict(zp, zm, ci)
        net.load_state_dict(sd)                                   # strict: same key set as the re-wired reference
        obs = torch.from_numpy(zm['m%d_obs' % ci]).float().to(dev)
        S = torch.from_numpy(zm['m%d_S' % ci]).to(dev)
        net.addGSO(S.squeeze(1) if m['E'] == 1 else S)
        got = torch.stack(net(obs), 1).cpu().numpy()
        want = zm['m%d_logits' % ci]
        assert np.abs(got - want).max() <= TOL, (ci, m, np.abs(got - want).max())
        assert (got.argmax(-1) == want.argmax(-1)).all()
        # train mode runs the same layers differentiably
        net.train()
        net.addGSO(S.squeeze(1) if m['E'] == 1 else S)
        out = net(obs)
        assert len(out) == m['N'] and out[0].requires_grad
        net.eval()


@pytest.mark.parametrize('prec', ['fp32', 'split_f16'])
def test_filter_split_f16_wide_dynamic_range(dev, prec):
    """The FILTER's contraction (G = 128; split-f16 and the default) with trained-scale taps spread over several
    decades and features from 1e-3 to 1e3: relative error stays at fp32 level against the float64 statement."""
    import gnn_pathplanning_amd.graphML as gml
    g = torch.Generator().manual_seed(17)
    B, N, K = 64, 10, 3
    for tap_scale, feat_scale in ((1.0, 1.0), (60.0, 300.0), (0.003, 1e-3), (25.0, 2e-2)):
        h = torch.randn(128, 1, K, 128, generator=g) * tap_scale / (128 * K) ** 0.5
        h[:, :, 1] *= 0.01                                         # taps of very different magnitude
        h[:7] *= 40.0                                              # and a few dominant output features
        b = torch.randn(128, 1, generator=g) * tap_scale * 0.1
        x = torch.relu(torch.randn(B, 128, N, generator=g)) * feat_scale
        x[:, ::5] *= 1e-3
        S = torch.from_numpy(orc.synth_gso_geometric(B, N, 20, seed=3)).unsqueeze(1)
        ref = orc.lsigf_f64(h.numpy(), S.numpy(), x.numpy(), b.numpy())
        y = gml.BatchLSIGF(h.to(dev), S.to(dev), x.to(dev), b.to(dev), precision=prec).cpu().numpy()
        scale = np.abs(ref).max()
        assert np.isfinite(y).all() and np.abs(y - ref).max() <= 2e-5 * scale, (tap_scale, feat_scale,
                                                                               np.abs(y - ref).max() / scale)


def test_filter_precision_is_per_call_and_per_instance(dev):
    """No process-wide precision switch in the functional / module API of the graph filter: the
    functions take `precision=` per call, the modules fix `self.precision` at construction.  Two threads, each on
    its own stream, run the SAME inputs under different arithmetics at the same time, many times over: every result
    of a thread is bit-identical to that arithmetic's single-threaded result (and the two arithmetics do differ, so
    a leak from one thread into the other would be seen)."""
    import threading
    import gnn_pathplanning_amd.graphML as gml
    assert not hasattr(gml, 'PRECISION')                          # the r03 module global is gone
    g = torch.Generator().manual_seed(5)
    B, N, K = 256, 10, 3
    h = (torch.randn(128, 1, K, 128, generator=g) / (128 * K) ** 0.5).to(dev)
    b = (0.1 * torch.randn(128, 1, generator=g)).to(dev)
    x = torch.relu(torch.randn(B, 128, N, generator=g)).to(dev)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, 20, seed=3)).unsqueeze(1).to(dev)
    want = {p: gml.BatchLSIGF(h, S, x, b, precision=p).clone() for p in ('fp32', 'split_f16')}
    assert not torch.equal(want['fp32'], want['split_f16'])       # (feature-major calls: exact fp32 MFMA vs f16 pairs)
    assert (want['fp32'] - want['split_f16']).abs().max().item() <= 1e-4
    # modules: the arithmetic is the instance's
    mods = {}
    for p in ('fp32', 'split_f16'):
        m = gml.GraphFilterBatch(128, 128, K, 1, True, precision=p).to(dev)
        with torch.no_grad():
            m.weight.copy_(h)
            m.bias.copy_(b)
        mods[p] = m
        assert m.precision == p
    assert gml.GraphFilterBatch(4, 4, 2).precision == gml.DEFAULT_PRECISION == 'fp32'
    with pytest.raises(Exception):
        gml.GraphFilterBatch(4, 4, 2, precision='fp64')
    torch.cuda.synchronize()
    bad = []

    def worker(p, use_module):
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st), torch.no_grad():
            for _ in range(40):
                if use_module:
                    mods[p].addGSO(S)
                    y = mods[p](x)
                else:
                    y = gml.BatchLSIGF(h, S, x, b, precision=p)
                if not torch.equal(y, want[p]):
                    bad.append(p)
        st.synchronize()
    for use_module in (False, True):
        ts = [threading.Thread(target=worker, args=(p, use_module)) for p in ('fp32', 'split_f16')]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    assert not bad, bad


def test_range_guard_and_exact_fallback(dev):
    """precision='split_f16' (opt-in): |activation| >= 65504 cannot go through the f16 pipe: the kernels raise the
    range flag (range_policy 'flag': check_range -> GnnppError), nothing is raised for in-range inputs, and the
    default range_policy 'strict' re-runs the call with the fp32-equivalent arithmetic and matches the oracle."""
    from gnn_pathplanning_amd import _native
    sd = orc.init_state_dict(3, seed=12)
    B, N = 9, 10
    obs = orc.synth_obs(B, N, seed=4)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, 20, seed=4))
    net = _net(N, 3, dev, sd)
    net.precision, net.range_policy = 'split_f16', 'flag'
    net.addGSO(S.to(dev))
    net(obs.to(dev))
    net.check_range()                                              # in range: no error
    big = obs.clone()
    big[3, 2, 0, 5, 5] = 2.0e5                                      # one huge observation value
    net(big.to(dev))
    with pytest.raises(_native.GnnppError, match='f16 range'):
        net.check_range()
    net.check_range()                                              # the flag was reset by the raise
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2['ConvLayers.0.weight'] *= 3.0e4                             # activations of L0 beyond 65504
    with torch.no_grad():
        want = torch.stack(orc.policy_forward(sd2, S, obs), 0)
    assert torch.isfinite(want).all()
    net2 = _net(N, 3, dev, sd2)
    net2.precision = 'split_f16'
    assert net2.range_policy == 'strict'
    net2.addGSO(S.to(dev))
    got = net2.forward_logits(obs.to(dev)).cpu()
    assert not net2.range_exceeded()                               # consumed by the fallback
    assert (got - want).abs().max().item() <= 2e-5 * want.abs().max().item()


def test_default_precision_has_no_input_domain(dev):
    """drive the module API exactly like agents/decentralplannerlocal.py:575-588 (addGSO, forward,
    argmax -- no check_range, no flag) with a checkpoint whose activations overflow the f16 range: the DEFAULT
    precision must match the oracle, on a second stream as well, and the same model must give the same bits from two
    streams at once (per-call arithmetic: nothing process-wide is toggled)."""
    sd = orc.init_state_dict(3, seed=12)
    sd['ConvLayers.0.weight'] *= 3.0e4                              # L0 activations ~1e5 .. 1e6 (> 65504)
    sd['ConvLayers.4.weight'] *= 1.0e-3
    B, N = 24, 10
    obs = orc.synth_obs(B, N, seed=4)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, 20, seed=4))
    with torch.no_grad():
        want = orc.policy_forward(sd, S, obs)
    assert all(torch.isfinite(w).all() for w in want)
    net = _net(N, 3, dev, sd)
    assert net.precision == 'fp32'
    net.addGSO(S.to(dev))
    got = net(obs.to(dev))                                          # the reference's call protocol, nothing else
    scale = max(w.abs().max().item() for w in want)
    err = max((g.cpu() - w).abs().max().item() for g, w in zip(got, want))
    assert err <= 2e-5 * scale, (err, scale)
    ids = torch.stack([g.argmax(-1) for g in got], 1).cpu()
    margin = orc.top2_margin(want)
    clear = margin > 1e-5 * scale
    assert (ids[clear] == orc.decode_actions(want)[clear]).all()
    assert not net.range_exceeded() and net._range_flag is None     # no guard exists in this mode
    # two streams at once: a split-f16 planner falling back on stream A must not change what stream B computes
    net_b = _net(N, 3, dev, sd)
    net_f = _net(N, 3, dev, sd)
    net_f.precision = 'split_f16'                                   # strict: overflows, re-runs with 'fp32'
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    obs_d, S_d = obs.to(dev), S.to(dev)
    torch.cuda.synchronize()
    outs_b = []
    for _ in range(4):
        with torch.cuda.stream(sa):
            net_f.addGSO(S_d)
            a = net_f.forward_logits(obs_d)
        with torch.cuda.stream(sb):
            net_b.addGSO(S_d)
            outs_b.append(net_b.forward_logits(obs_d))
    torch.cuda.synchronize()
    ref = torch.stack(got, 0)
    assert all(torch.equal(o, ref) for o in outs_b)
    assert (a.cpu() - torch.stack(want, 0)).abs().max().item() <= 2e-5 * scale


@pytest.mark.parametrize('lo,hi', [(1e-30, 1e-24), (1e-12, 1e-3), (1e3, 1e9), (1e20, 1e30), (1e-30, 1e30)])
def test_default_precision_adversarial_ranges(dev, lo, hi):
    """bf16x3 represents every finite fp32 operand exactly, with fp32's exponent range: encoder features for
    observations / weights whose magnitudes are spread log-uniformly over [lo, hi] (far outside the f16 range on
    either side; products stay inside fp32's) agree with the float64 statement of the same network at fp32 relative
    accuracy -- no guard, no rescaling, no fallback involved."""
    g = torch.Generator().manual_seed(int(abs(np.log10(lo)) * 100 + abs(np.log10(hi))))
    sd = orc.init_state_dict(3, seed=21)
    B, N = 3, 6

    def spread(shape):
        e = torch.rand(shape, generator=g) * (np.log10(hi) - np.log10(lo)) + np.log10(lo)
        return (10.0 ** e.double()).float() * (torch.randint(0, 2, shape, generator=g) * 2 - 1).float()
    obs = spread((B, N, 3, 11, 11)) * (torch.rand(B, N, 3, 11, 11, generator=g) < 0.3)
    # first-layer weights scaled so that L0's outputs are O(1) again: the later layers see ordinary activations, the
    # first layer sees operands of magnitude lo .. hi times weights of magnitude 1/hi .. 1/lo
    sd['ConvLayers.0.weight'] = sd['ConvLayers.0.weight'] / float(np.sqrt(lo * hi))
    net = _net(N, 3, dev, sd)
    got = net.encode(obs.to(dev)).cpu().double()
    want = orc.policy_features({k: v.double() for k, v in sd.items()}, obs.double()).permute(0, 2, 1)
    assert torch.isfinite(got).all()
    scale = want.abs().max().item()
    assert scale > 0 and (got - want).abs().max().item() <= 3e-5 * scale, ((got - want).abs().max().item(), scale)


@pytest.mark.parametrize('case', ['golden', 'binary_c2', 'band_1e-30_1e-20', 'band_1e10_1e20', 'band_1e-6_1e6'])
def test_default_precision_error_is_comparable_to_the_exact_fp32_mfma(dev, policy_golden, case):
    """"fp32-equivalent" stated as a COMPARISON, per output and not as one max-scaled bound: on the
    same inputs, the error of the default arithmetic (bf16x3 planes) against the float64 statement of the network is at
    most twice the error of the exact fp32 MFMA schedule (an fmaf chain: the reference's own kind of arithmetic) -- for
    every one of the 128 encoder features and every logit column: RMS over the samples within 2x (+ two ulps of that
    output's magnitude, at least the median output's: dead-ReLU features; a sum of a few hundred fp32 terms differs by
    an ulp or two between ANY two summation orders), maximum over the samples within 4x (+ four ulps); the RMS over ALL
    outputs within 2x with no slack.
    Golden inputs of the reference, the bench's binary batch, and the adversarial magnitude bands."""
    z, meta = policy_golden
    g = torch.Generator().manual_seed(len(case))
    if case == 'golden':
        sd = golden_state_dict(z, 3)
        i = [k for k, m in enumerate(meta) if m['K'] == 3 and m['N'] == 10][0]
        obs = torch.from_numpy(z['p%d_obs' % i].astype(np.float32))
        S = torch.from_numpy(np.ascontiguousarray(z['p%d_S' % i])).double()
        if S.dim() == 4:
            S = S.squeeze(1)
    elif case == 'binary_c2':
        sd = orc.init_state_dict(3, seed=1337)
        obs = orc.synth_obs(128, 10, seed=1337)
        S = torch.from_numpy(orc.synth_gso_geometric(128, 10, 20, seed=1337))
    else:
        lo, hi = (float(t) for t in case.split('_')[1:])
        sd = orc.init_state_dict(3, seed=21)
        B, N = 24, 6
        e = torch.rand((B, N, 3, 11, 11), generator=g) * (np.log10(hi) - np.log10(lo)) + np.log10(lo)
        obs = (10.0 ** e.double()).float() * (torch.randint(0, 2, e.shape, generator=g) * 2 - 1).float()
        obs = obs * (torch.rand(e.shape, generator=g) < 0.3)
        sd['ConvLayers.0.weight'] = sd['ConvLayers.0.weight'] / float(np.sqrt(lo * hi))
        S = torch.from_numpy(orc.synth_gso_geometric(B, N, 12, seed=5))
    B, N = obs.shape[0], obs.shape[1]
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        ref_feat = torch.stack([orc.encoder_one_agent(sd64, obs[:, n].double()) for n in range(N)], 1)   # [B,N,128] f64
    y = orc.lsigf_f64(sd['GFL.0.weight'].numpy(), S.unsqueeze(1).numpy(), ref_feat.permute(0, 2, 1).numpy(),
                      sd['GFL.0.bias'].numpy())
    y = torch.relu(torch.from_numpy(y)).permute(0, 2, 1)                                                # [B,N,128]
    ref_logits = y @ sd64['actionsMLP.0.weight'].t() + sd64['actionsMLP.0.bias']                        # [B,N,5]
    net = _net(N, 3, dev, sd)
    net.addGSO(S.to(dev))
    err = {}
    for prec in ('fp32', 'fp32_mfma'):
        net.precision = prec
        feat = net.encode(obs.to(dev)).cpu().double().reshape(B, N, 128)
        logits = net.forward_logits(obs.to(dev)).cpu().double().permute(1, 0, 2)                        # [B,N,5]
        assert torch.isfinite(feat).all() and torch.isfinite(logits).all()
        err[prec] = ((feat - ref_feat).abs().reshape(-1, 128), (logits - ref_logits).abs().reshape(-1, N * 5)
                     if False else (logits - ref_logits).abs().permute(1, 2, 0).reshape(N * 5, B).t())
    for which, ref in ((0, ref_feat.abs().reshape(-1, 128)), (1, ref_logits.abs().permute(1, 2, 0).reshape(N * 5, B).t())):
        e_b3, e_mf = err['fp32'][which], err['fp32_mfma'][which]
        # an output's own magnitude, but not less than the typical output's: a feature that ReLU holds at (or near) zero
        # is the sum of terms of the typical size, and neither arithmetic resolves that sum finer than fp32 does
        colmax = ref.max(0).values
        ulp = torch.maximum(colmax, colmax.median()) * 2.0 ** -23
        # per output: RMS error over the samples within a factor of two (+ two ulps); the MAXIMUM over the samples -- a noisy
        # statistic of two dozen to a few thousand samples -- within a factor of four (+ four ulps)
        r_b3, r_mf = e_b3.pow(2).mean(0).sqrt(), e_mf.pow(2).mean(0).sqrt()
        bad = r_b3 > 2.0 * r_mf + 2.0 * ulp
        assert not bad.any(), (case, which, 'rms', int(bad.sum()), (r_b3 / (r_mf + ulp)).max().item())
        worst_b3, worst_mf = e_b3.max(0).values, e_mf.max(0).values
        bad = worst_b3 > 4.0 * worst_mf + 4.0 * ulp
        assert not bad.any(), (case, which, 'max', int(bad.sum()), (worst_b3 / (worst_mf + ulp)).max().item())
        rms_b3, rms_mf = e_b3.pow(2).mean().sqrt().item(), e_mf.pow(2).mean().sqrt().item()
        assert rms_b3 <= 2.0 * rms_mf + 1e-300, (case, which, rms_b3, rms_mf)


def test_default_precision_denormal_and_huge_weights(dev):
    """Weights in fp32's denormal range (|w| ~ 1e-40: exactly representable by bf16 planes only down to their own
    subnormals -- the contribution is below fp32's resolution of the sum either way) and weights near 1e30 with
    observations near 1e-30: finite, and equal to the float64 statement at fp32 relative accuracy."""
    g = torch.Generator().manual_seed(77)
    sd = orc.init_state_dict(3, seed=22)
    w0 = sd['ConvLayers.0.weight']
    w0[:, 0] *= 1e30                                               # channel 0: huge weights x tiny observations
    w0[:, 1] = torch.randn(w0[:, 1].shape, generator=g) * 1e-40    # channel 1: denormal weights
    B, N = 2, 5
    obs = torch.rand(B, N, 3, 11, 11, generator=g)
    obs[:, :, 0] *= 1e-30
    net = _net(N, 3, dev, sd)
    got = net.encode(obs.to(dev)).cpu().double()
    want = orc.policy_features({k: v.double() for k, v in sd.items()}, obs.double()).permute(0, 2, 1)
    assert torch.isfinite(got).all()
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 3e-5 * scale


def test_replaced_parameter_objects_are_seen_by_the_next_forward(dev):
    """a Parameter / buffer / sub-module OBJECT replaced by hand (no version counter, no
    load_state_dict, no train() transition in between) must change the logits of the VERY NEXT forward -- the
    memoised tensor list is validated by object identity on every call."""
    sd = orc.init_state_dict(3, seed=9)
    net = _net(6, 3, dev, sd)
    obs = orc.synth_obs(3, 6, seed=2).to(dev)
    S = torch.from_numpy(orc.synth_gso_geometric(3, 6, 12, seed=2)).to(dev)
    net.addGSO(S)

    def check():
        got = net.forward_logits(obs).cpu()
        cur = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        with torch.no_grad():
            want = torch.stack(orc.policy_forward(cur, S.cpu(), obs.cpu()), 0)
        assert (got - want).abs().max().item() <= TOL * max(1.0, want.abs().max().item())
        return got
    a = check()
    g = torch.Generator().manual_seed(1)
    conv = net.ConvLayers[4]
    conv.weight = torch.nn.Parameter((torch.randn(conv.weight.shape, generator=g) * 0.06).to(dev))   # new Parameter object
    b = check()
    assert (a - b).abs().max().item() > 1e-3
    bn = net.ConvLayers[8]
    bn.running_var = (torch.rand(bn.running_var.shape, generator=g) + 0.5).to(dev)                 # new buffer object
    c = check()
    assert (b - c).abs().max().item() > 1e-4
    lin = torch.nn.Linear(128, 5).to(dev)
    net.actionsMLP[0] = lin                                                                           # new sub-module
    d = check()
    assert (c - d).abs().max().item() > 1e-3
    net.GFL[0].weight = torch.nn.Parameter(net.GFL[0].weight.detach() * 0.5)
    e = check()
    assert (d - e).abs().max().item() > 1e-4


def test_unseen_parameter_updates_and_invalidate_packed(dev):
    """`p.data` edits do not bump torch's version counters: invalidate_packed() (or a train()/eval()
    transition) makes the next forward repack."""
    sd = orc.init_state_dict(3, seed=2)
    net = _net(5, 3, dev, sd)
    obs = orc.synth_obs(2, 5, seed=1).to(dev)
    S = torch.from_numpy(orc.synth_gso_geometric(2, 5, 12, seed=1)).to(dev)
    net.addGSO(S)
    a = net.forward_logits(obs).clone()
    net.actionsMLP[0].weight.data.mul_(3.0)
    net.GFL[0].weight.data.mul_(0.5)
    net.ConvLayers[0].weight.data.mul_(1.5)
    net.invalidate_packed()
    b = net.forward_logits(obs).clone()
    sd2 = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    with torch.no_grad():
        want = torch.stack(orc.policy_forward(sd2, S.cpu(), obs.cpu()), 0)
    assert (b.cpu() - want).abs().max().item() <= TOL and (a - b).abs().max().item() > 1e-3
    net.GFL[0].weight.data.mul_(2.0)
    net.train(); net.eval()                                        # the transition repacks as well
    c = net.forward_logits(obs)
    sd3 = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    with torch.no_grad():
        want3 = torch.stack(orc.policy_forward(sd3, S.cpu(), obs.cpu()), 0)
    assert (c.cpu() - want3).abs().max().item() <= TOL
    # a module moved to another dtype is read through casted copies, not misread
    net64 = _net(5, 3, dev, sd).double()
    net64.addGSO(S)
    d = net64.forward_logits(obs)
    assert (d - a).abs().max().item() <= 1e-6


@pytest.mark.parametrize('B,N,K,f64', [(16, 100, 3, 0), (9, 100, 2, 1), (4, 80, 4, 0), (32, 100, 4, 0), (64, 50, 3, 1),
                                       (128, 100, 3, 0), (16, 72, 3, 0)])
def test_policy_filter_n_way_split(dev, B, N, K, f64):
    """up to ceil(N / 16) workgroups per graph in policy_filter_kernel / lsigf_kernel
    (csrc/lsigf_kernel.hip lsigf_plan: as many parts as keep one workgroup per CU; 16 graphs of 100 agents -- the
    shard one GPU of eight holds of config 5 -- run as 112 one-tile workgroups).  A batch that MIXES sparse graphs,
    near-cliques, hubs and isolated nodes: the heuristic's split, every forced split 1 .. 7 and the general filter
    kernel on the same inputs.  Exact-fp32 contraction: every split gives the one-workgroup logits bit for bit (a
    row's arithmetic does not depend on the partition).  Default precision: equal to rounding (a finer split can move
    a team from the fp32 MFMA to bf16x3 planes, which then fit the LDS) and within TOL of the float64 statement."""
    from gnn_pathplanning_amd import _native
    L = _native.lib()
    g = torch.Generator().manual_seed(B * 17 + N + K)
    h = torch.randn(128, 1, K, 128, generator=g) / (128 * K) ** 0.5
    x = torch.relu(torch.randn(B, N, 128, generator=g))
    dens = torch.tensor([0.06 if b % 3 else 0.35 for b in range(B)]).reshape(B, 1, 1)     # every third graph: dense
    S = ((torch.rand(B, N, N, generator=g) < dens) * torch.rand(B, N, N, generator=g))
    S[1 % B, :, N // 2] = torch.rand(N, generator=g)                                     # a hub in a sparse graph
    S[:, :, 5] = 0                                                                       # an isolated node everywhere
    S = S * (1 - torch.eye(N))
    S = S.double() if f64 else S.float()
    bias, aw, ab = torch.randn(128, generator=g) / 4, torch.randn(5, 128, generator=g) / 8, torch.randn(5, generator=g)
    packed = torch.empty(L.gnnpp_filter_packed_floats(128, 128, K, 1), dtype=torch.float32, device=dev)
    hd = h.to(dev)
    assert L.gnnpp_filter_pack(hd.data_ptr(), packed.data_ptr(), 128, 128, K, 1, None) == 0
    xd, Sd, bd, awd, abd = x.to(dev), S.to(dev), bias.to(dev), aw.to(dev), ab.to(dev)

    def run(prec):
        lg = torch.full((N, B, 5), float('nan'), device=dev)
        assert L.gnnpp_filter_head_fwd(xd.data_ptr(), Sd.data_ptr(), packed.data_ptr(), bd.data_ptr(), awd.data_ptr(),
                                       abd.data_ptr(), lg.data_ptr(), B, N, 128, 128, K, 1, f64, prec, None, None) == 0
        torch.cuda.synchronize()
        return lg.cpu()
    outs = {}
    try:
        for prec in (1, 0):
            for split in (0, 1, 2, 3, 4, 5, 6, 7):           # 0 = the heuristic
                assert L.gnnpp_set_tuning(7, split) == 0
                outs[prec, split] = run(prec)
            outs[prec, 'again'] = run(prec)                  # (a repeated launch: the same bits)
        assert L.gnnpp_set_tuning(7, 0) == 0 and L.gnnpp_set_tuning(9, 0) == 0
        general = run(0)                                     # the general filter kernel, its own n-way split
    finally:
        L.gnnpp_set_tuning(9, 1)
        L.gnnpp_set_tuning(7, 0)
    z = x.double()
    y = torch.zeros(B, N, 128, dtype=torch.float64)
    Sf = S.float().double()
    for k in range(K):
        y += z @ h[:, 0, k, :].double().t()
        z = torch.einsum('bmn,bmg->bng', Sf, z)
    want = (torch.relu(y + bias.double()) @ aw.double().t() + ab.double()).permute(1, 0, 2)
    scale = max(1.0, want.abs().max().item())
    for split in (0, 1, 2, 3, 4, 5, 6, 7, 'again'):
        assert torch.equal(outs[1, split], outs[1, 1]), split
        assert torch.isfinite(outs[0, split]).all()
        assert (outs[0, split].double() - want).abs().max().item() <= TOL * scale, split
        assert (outs[0, split] - outs[0, 1]).abs().max().item() <= 4e-6 * scale, split
    assert torch.equal(outs[0, 'again'], outs[0, 7])
    assert (general - outs[0, 1]).abs().max().item() <= 4e-6 * scale


@pytest.mark.parametrize('B,N,K,f64', [(256, 50, 3, 0), (128, 100, 3, 1), (128, 100, 2, 0), (7, 17, 3, 0), (300, 64, 4, 1),
                                       (5, 89, 3, 0), (5, 90, 3, 1), (130, 97, 1, 0), (40, 33, 2, 0), (64, 100, 4, 0)])
@pytest.mark.parametrize('prec', [0, 1, 2])
def test_policy_filter_kernel_vs_general_filter(dev, B, N, K, f64, prec):
    """Filter + ReLU + head of the policy step for 17..100 agents: policy_filter_kernel (default) against the
    general filter kernel on the same inputs -- equal to rounding (the head's eight partial sums instead of one
    chain) -- and against an fp64 restatement within TOL.  Shapes on both sides of every switch: one / two
    workgroups per graph (B <= 128 large graphs), partial logits in their own LDS / in the dead S slab (N >= 90),
    1..4 row tiles per wave, K = 1..4, fp32 / fp64 GSO."""
    from gnn_pathplanning_amd import _native
    L = _native.lib()
    g = torch.Generator().manual_seed(B * 131 + N * 7 + K)
    h = torch.randn(128, 1, K, 128, generator=g) / (128 * K) ** 0.5
    x = torch.relu(torch.randn(B, N, 128, generator=g))
    S = ((torch.rand(B, N, N, generator=g) < 8.0 / N) * torch.rand(B, N, N, generator=g))
    S[0, :, N // 2] = torch.rand(N, generator=g)                           # hubs: nodes that gather from (almost)
    S[B // 2, :, N - 1] = torch.rand(N, generator=g)                       # everybody take the whole-wave path
    S[-1, :, 0] = (torch.rand(N, generator=g) < 0.5) * torch.rand(N, generator=g)
    S = S * (1 - torch.eye(N))
    S = S.double() if f64 else S.float()
    bias, aw, ab = torch.randn(128, generator=g) / 4, torch.randn(5, 128, generator=g) / 8, torch.randn(5, generator=g)
    packed = torch.empty(L.gnnpp_filter_packed_floats(128, 128, K, 1), dtype=torch.float32, device=dev)
    hd = h.to(dev)
    assert L.gnnpp_filter_pack(hd.data_ptr(), packed.data_ptr(), 128, 128, K, 1, None) == 0
    xd, Sd, bd, awd, abd = x.to(dev), S.to(dev), bias.to(dev), aw.to(dev), ab.to(dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    outs = []
    try:
        for mode in (1, 0, 1):
            assert L.gnnpp_set_tuning(9, mode) == 0
            lg = torch.full((N, B, 5), float('nan'), device=dev)
            assert L.gnnpp_filter_head_fwd(xd.data_ptr(), Sd.data_ptr(), packed.data_ptr(), bd.data_ptr(),
                                           awd.data_ptr(), abd.data_ptr(), lg.data_ptr(), B, N, 128, 128, K, 1, f64,
                                           prec, flag.data_ptr(), None) == 0
            torch.cuda.synchronize()
            outs.append(lg.cpu())
    finally:
        L.gnnpp_set_tuning(9, 1)
    assert flag.item() == 0
    assert torch.equal(outs[0], outs[2])                                  # deterministic
    z = x.double()
    y = torch.zeros(B, N, 128, dtype=torch.float64)
    Sf = S.float().double()
    for k in range(K):
        y += z @ h[:, 0, k, :].double().t()
        z = torch.einsum('bmn,bmg->bng', Sf, z)
    want = (torch.relu(y + bias.double()) @ aw.double().t() + ab.double()).permute(1, 0, 2)
    scale = max(1.0, want.abs().max().item())
    assert (outs[0].double() - want).abs().max().item() <= TOL * scale
    assert (outs[0] - outs[1]).abs().max().item() <= 4e-6 * scale
    assert not torch.equal(outs[0], outs[1]) or K == 0 or prec == 1       # (the new kernel did run; under the exact
                                                                          # fp32 MFMA both may round identically)


def test_torch_ops_equal_the_module_api_and_trace(dev):
    """torch.ops.gnnpp.{lsigf, policy_logits, decode_actions}: the same kernels as the module API (bit-identical), and a
    function built on them compiles with fullgraph=True (no graph break on the ctypes call: the ops are opaque to
    dynamo, their fake implementations give the shapes)."""
    import gnn_pathplanning_amd.ops  # noqa: F401
    from gnn_pathplanning_amd import graphML as gml
    sd = orc.init_state_dict(3, seed=21)
    B, N = 6, 10
    obs = orc.synth_obs(B, N, seed=8).to(dev)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, 20, seed=8)).to(dev)            # fp64, [B,1,N,N] after addGSO
    net = _net(N, 3, dev, sd)
    net.addGSO(S)
    want = net.forward_logits(obs)
    gf, act = net.GFL[0], net.actionsMLP[0]
    got = torch.ops.gnnpp.policy_logits(obs, net.S, net.packed_encoder(), gf.packed_taps(), gf.bias, act.weight,
                                        act.bias, 3, 0)
    assert torch.equal(got, want)
    assert torch.equal(torch.ops.gnnpp.decode_actions(got), net.decode_actions(want))
    # the op takes the caller's packs: a tap count that does not match the pack, a truncated encoder pack or a GSO of
    # another shape is an error, never an out-of-bounds read
    from gnn_pathplanning_amd._native import GnnppError
    for bad in (lambda: torch.ops.gnnpp.policy_logits(obs, net.S, net.packed_encoder(), gf.packed_taps(), gf.bias,
                                                      act.weight, act.bias, 4, 0),
                lambda: torch.ops.gnnpp.policy_logits(obs, net.S, net.packed_encoder()[:-8].contiguous(),
                                                      gf.packed_taps(), gf.bias, act.weight, act.bias, 3, 0),
                lambda: torch.ops.gnnpp.policy_logits(obs, net.S[:, :, :9, :9].contiguous(), net.packed_encoder(),
                                                      gf.packed_taps(), gf.bias, act.weight, act.bias, 3, 0)):
        with pytest.raises(GnnppError):
            bad()
    g = torch.Generator().manual_seed(3)
    h = (torch.randn(96, 1, 3, 128, generator=g) / 20).to(dev)
    x = torch.randn(B, 128, N, generator=g).to(dev)
    b = torch.randn(96, 1, generator=g).to(dev)
    y = torch.ops.gnnpp.lsigf(h, net.S, x, b, False, 0)
    assert torch.equal(y, gml.BatchLSIGF(h, net.S, x, b))

    enc, taps, gb, aw, ab = net.packed_encoder(), gf.packed_taps(), gf.bias.detach(), act.weight.detach(), act.bias.detach()

    def step(o, s):
        lg = torch.ops.gnnpp.policy_logits(o, s, enc, taps, gb, aw, ab, 3, 0)
        return torch.ops.gnnpp.decode_actions(lg), lg * 1.0

    compiled = torch.compile(step, backend='aot_eager', fullgraph=True)
    ids, lg = compiled(obs, net.S)
    assert torch.equal(lg, want) and torch.equal(ids, net.decode_actions(want))


@pytest.mark.parametrize('N,K,B', [(10, 3, 3000), (16, 2, 1100), (7, 4, 2500), (3, 1, 4000)])
def test_pipeline_filter_kernel_equals_the_small_graph_kernel(dev, N, K, B):
    """lsigf_pipe_b3_kernel (persistent 8-wave producer / consumer workgroups, several groups each at these sizes)
    computes every row with the arithmetic of lsigf_small_b3_kernel: bit-identical outputs; both within tolerance of the
    float64 statement.  Ragged last group, K = 1 (no shifts), fp64 GSOs."""
    import ctypes
    from gnn_pathplanning_amd import _native
    from gnn_pathplanning_amd.graphML import pack_filter_taps
    L = _native.lib()
    g = torch.Generator().manual_seed(N * 100 + K)
    h = (torch.randn(128, 1, K, 128, generator=g) / (128 * K) ** 0.5)
    taps = pack_filter_taps(h.to(dev))
    bias = (torch.randn(128, generator=g) / 4).to(dev)
    x = torch.relu(torch.randn(B * N, 128, generator=g)).to(dev)
    S64 = ((torch.rand(B, N, N, generator=g) < 0.4) * torch.rand(B, N, N, generator=g)).double()
    S = S64.to(dev) if K == 2 else S64.float().to(dev)
    outs = []
    try:
        for mode, pgrid in ((2, 0), (3, 0), (3, 7)):
            assert L.gnnpp_set_tuning(10, mode) == 0 and L.gnnpp_set_tuning(12, pgrid) == 0
            y = torch.full_like(x, float('nan'))
            rc = L.gnnpp_lsigf_fwd(x.data_ptr(), S.data_ptr(), taps.data_ptr(), bias.data_ptr(), y.data_ptr(), B, N, N,
                                   128, 128, K, 1, int(S.dtype is torch.float64), 1, 1, 1, 1, 0, 0, None,
                                   _native.stream_ptr(dev))
            assert rc == 0
            outs.append(y)
    finally:
        L.gnnpp_set_tuning(10, 1)
        L.gnnpp_set_tuning(12, 0)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    want = orc.lsigf_f64(h.numpy(), S64.float().unsqueeze(1).numpy()[:64], x.cpu().reshape(B, N, 128)[:64].permute(0, 2, 1).numpy(),
                         bias.cpu().numpy().reshape(128, 1))
    got = outs[1].cpu().reshape(B, N, 128)[:64].permute(0, 2, 1).numpy()
    want = np.maximum(want, 0)
    assert np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize('B,N,K,W', [(1, 10, 3, 20), (16, 100, 3, 100), (64, 10, 3, 20)])
def test_graphed_policy_step_equals_eager(dev, B, N, K, W):
    """rollout.GraphedPolicyStep (addGSO + forward as one HIP-graph replay, the latency regime: one case per step, or
    the 16-graph shard of an 8-GPU run): the replay on NEW inputs gives bit for bit what the eager call gives on them,
    agrees with the oracle, a second replay does not depend on the first, and a wrong shape is refused."""
    from gnn_pathplanning_amd.rollout import GraphedPolicyStep
    sd = orc.init_state_dict(K, seed=1337 + K)
    net = _net(N, K, dev, sd)
    obs0 = orc.synth_obs(B, N, seed=1).to(dev)
    S0 = torch.from_numpy(orc.synth_gso_geometric(B, N, W, seed=1)).float().to(dev)
    step = GraphedPolicyStep(net, obs0, S0)
    for seed in (2, 3):
        obs = orc.synth_obs(B, N, seed=seed)
        S = torch.from_numpy(orc.synth_gso_geometric(B, N, W, seed=seed)).float()
        got = [g.clone() for g in step(obs.to(dev), S.to(dev))]
        with torch.no_grad():
            net.addGSO(S.to(dev))
            eager = net(obs.to(dev))
            want = orc.policy_forward(sd, S, obs)
        assert len(got) == N and all(torch.equal(a, b) for a, b in zip(got, eager))
        assert max((g.cpu() - w).abs().max().item() for g, w in zip(got, want)) <= TOL
    with pytest.raises(ValueError):
        step(obs0[:, :-1], S0[:, :-1, :-1]) if N > 1 else step(obs0[:0], S0[:0])
    net.train()
    with pytest.raises(ValueError):
        GraphedPolicyStep(net, obs0, S0)
    with pytest.raises(ValueError):
        step(obs0, S0)                                       # a captured step refuses a model switched to train()
    net.eval()


def test_graphed_policy_step_follows_weight_changes(dev):
    """the graph bakes in the addresses of the model's packed weight copies.  After load_state_dict, an
    in-place parameter update or invalidate_packs() the step must neither replay the OLD weights nor read buffers the
    pack caches have dropped: it re-captures, and its logits equal the eager call's with the NEW weights bit for bit
    (and agree with the oracle on them)."""
    from gnn_pathplanning_amd.rollout import GraphedPolicyStep
    from gnn_pathplanning_amd import _native
    B, N, K, W = 8, 10, 3, 20
    sd_a, sd_b = orc.init_state_dict(K, seed=11), orc.init_state_dict(K, seed=12)
    net = _net(N, K, dev, sd_a)
    obs = orc.synth_obs(B, N, seed=5)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, W, seed=5)).float()
    obs_d, S_d = obs.to(dev), S.to(dev)
    step = GraphedPolicyStep(net, obs_d, S_d)
    first = [g.clone() for g in step(obs_d, S_d)]
    assert step.recaptures == 0
    step(obs_d, S_d)
    assert step.recaptures == 0                              # unchanged weights: plain replays
    held_before = [b.data_ptr() for b in step._held if torch.is_tensor(b)]
    net.load_state_dict(sd_b)
    junk = [torch.full((1 << 18,), float('nan'), device=dev) for _ in range(8)]   # reuse whatever the caches freed
    got = [g.clone() for g in step(obs_d, S_d)]
    assert step.recaptures == 1
    with torch.no_grad():
        net.addGSO(S_d)
        eager = net(obs_d)
        want = orc.policy_forward(sd_b, S, obs)
    assert all(torch.equal(a, b) for a, b in zip(got, eager))
    assert max((g.cpu() - w).abs().max().item() for g, w in zip(got, want)) <= TOL
    assert not all(torch.equal(a, b) for a, b in zip(got, first))
    del junk
    # an in-place update (what an optimizer step does) and the explicit invalidation are both seen
    with torch.no_grad():
        net.actionsMLP[0].bias.add_(0.25)
    got2 = [g.clone() for g in step(obs_d, S_d)]
    assert step.recaptures == 2
    assert all(torch.allclose(a, b + 0.25, atol=1e-6) for a, b in zip(got2, got))
    _native.invalidate_packs()
    step(obs_d, S_d)
    assert step.recaptures == 3
    assert held_before                                       # (the step did hold device buffers)
    # split_f16 + strict range policy synchronises with the host after every forward: refused at capture
    net.precision, net.range_policy = 'split_f16', 'strict'
    with pytest.raises(ValueError):
        GraphedPolicyStep(net, obs_d, S_d)


@pytest.mark.parametrize('fused', [1, 0])
def test_non_finite_observations_flush_is_pinned(dev, enc_variant, fused):
    """VERDICT r04 item 7 ("Numerics"): a NaN / +Inf pixel makes every logit of its graph NaN in the
    reference (torch.relu keeps it, graphs/models/decentralplanner.py:166; the dense x @ S spreads it,
    utils/graphUtils/graphML.py:2350); the kernels' ReLU flushes non-finite activations, so the device logits of
    those graphs are FINITE -- the documented deviation, pinned against the oracle here in all three arithmetics and
    both dispatch paths -- and the clean graphs of the same batch keep parity (nothing leaks across graphs)."""
    from gnn_pathplanning_amd import _native
    from test_emu_kernels import _non_finite_case
    sd_t, obs_t, S, want = _non_finite_case()
    assert np.isnan(want[1]).all() and np.isnan(want[2]).all() and np.isfinite(want[[0, 3]]).all()
    net = _net(10, 3, dev, sd_t)
    net.range_policy = 'flag'
    L = _native.lib()
    old = L.gnnpp_get_tuning(6)
    L.gnnpp_set_tuning(6, fused)
    try:
        with torch.no_grad():
            net.addGSO(S.to(dev))
            got = torch.stack([g.cpu() for g in net(obs_t.to(dev))], 1).numpy()
    finally:
        L.gnnpp_set_tuning(6, old)
    assert np.abs(got[[0, 3]] - want[[0, 3]]).max() <= TOL
    assert np.isfinite(got).all()


@pytest.mark.parametrize('B,N,G,F_out,K,E', [(3, 100, 128, 128, 3, 1), (5, 37, 24, 20, 3, 2), (16, 64, 128, 96, 2, 1)])
def test_general_filter_n_way_split_is_bit_identical(dev, B, N, G, F_out, K, E):
    """lsigf_kernel with 1 .. 7 workgroups per graph (GNNPP_TUNE_FILTER_SPLIT; v320, the heuristic included): output in
    both layouts AND the tap dump of the training entry point (gnnpp_lsigf_fwd_save) are the one-workgroup results bit
    for bit -- every part runs the early shifts on all rows, a row's last shift / contraction / epilogue does not depend
    on which part owns it -- and agree with the float64 statement."""
    from gnn_pathplanning_amd import _native
    import gnn_pathplanning_amd.graphML as gml
    L = _native.lib()
    g = torch.Generator().manual_seed(1000 * N + G + K)
    h = torch.randn(F_out, E, K, G, generator=g) / (G * K) ** 0.5
    x = torch.randn(B, N, G, generator=g)
    S = (torch.rand(B, E, N, N, generator=g) < 0.12).float() * torch.rand(B, E, N, N, generator=g)
    bias = torch.randn(F_out, generator=g) / 4
    hd, xd, Sd, bd = h.to(dev), x.to(dev), S.to(dev), bias.to(dev)
    packed = gml.pack_filter_taps(hd)
    outs = {}
    try:
        for split in (1, 0, 2, 3, 5, 7):
            assert L.gnnpp_set_tuning(7, split) == 0 and L.gnnpp_set_tuning(1, 1) == 0
            y = torch.full((B, N, F_out), float('nan'), device=dev)
            zs = torch.full((E * K, B * N, G), float('nan'), device=dev)
            rc = L.gnnpp_lsigf_fwd_save(xd.data_ptr(), Sd.data_ptr(), packed.data_ptr(), bd.data_ptr(), y.data_ptr(),
                                        zs.data_ptr(), B, N, N, G, F_out, K, E, 0, 1, 0, 1, 1, 1, 0, 1, None,
                                        _native.stream_ptr(dev))
            assert rc == 0
            yf = gml.BatchLSIGF(hd, Sd, xd.permute(0, 2, 1).contiguous(), bd.reshape(F_out, 1), precision='fp32_mfma')
            torch.cuda.synchronize()
            outs[split] = (y.cpu(), zs.cpu(), yf.cpu())
    finally:
        L.gnnpp_set_tuning(7, 0)
        L.gnnpp_set_tuning(1, 0)
    for split, (y, zs, yf) in outs.items():
        assert torch.equal(y, outs[1][0]) and torch.equal(zs, outs[1][1]) and torch.equal(yf, outs[1][2]), split
    want = np.maximum(orc.lsigf_f64(h.numpy(), S.numpy(), x.permute(0, 2, 1).numpy(), bias.numpy().reshape(F_out, 1)), 0)
    got = outs[7][0].permute(0, 2, 1).numpy()
    assert np.abs(got - want).max() <= TOL * max(1.0, np.abs(want).max())
    assert np.abs(outs[7][2].numpy() - orc.lsigf_f64(h.numpy(), S.numpy(), x.permute(0, 2, 1).numpy(),
                                                      bias.numpy().reshape(F_out, 1))).max() <= TOL * max(1.0, np.abs(want).max())


def test_bench_prints_one_parseable_line_of_at_most_6_kb(dev, tmp_path):
    """VERDICT r05 item 1, end to end on the GPU: `python bench.py` (short regions, the secondary block and the CPU baseline
    off to keep the test short) prints ONE JSON line of <= 6 144 bytes carrying the contract keys, `roofline` and `parity`,
    names the side file it wrote, and that side file holds the full record the line was derived from."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    details = str(tmp_path / 'bench_full.json')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '10', '--warmup', '3',
                        '--repeats', '3', '--no-secondary', '--no-cpu-baseline', '--pmc', 'off', '--details-file', details],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) <= 6144, (len(lines), [len(x) for x in lines])
    d = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'parity', 'summary', 'details_file'):
        assert key in d, key
    assert d['n_gpus'] == 1 and d['steps'] == 10 and d['warmup'] == 3 and d['vs_baseline'] is None
    assert d['value'] > 1e7 and abs(d['value'] - 5120 / (d['ms_per_step'] * 1e-3)) <= 1e-3 * d['value']
    assert d['roofline']['bound'] == 'mfma' and 0 < d['roofline']['frac'] < 1 / 6
    assert d['parity']['max_abs_dlogit'] <= 1e-4 and d['parity']['argmax_equal_on_clear_rows'] is True
    full = json.load(open(details))
    assert full['value'] == pytest.approx(d['value'], rel=1e-5) and 'step_breakdown_us' in full
    sys.path.insert(0, root)
    import bench
    assert bench.driver_line(full, d['details_file']) == lines[0]
