"""GPU (-m gpu): training path -- graph-filter forward/backward on the HIP kernels (autograd
Function), train-mode DecentralPlannerNet, loss/gradient/BN-statistics parity with the REAL
reference (tests/golden/training_grads.npz) and with the CPU oracle's autograd."""
import numpy as np
import pytest
import torch

from conftest import golden_state_dict
from oracle import policy_oracle as orc

pytestmark = pytest.mark.gpu
RTOL = 2e-4        # gradients: relative to the tensor's max magnitude (fp32, different summation order)


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    from gnn_pathplanning_amd import _native
    _native.lib()
    return torch.device('cuda:0')


def close(a, b, rtol=RTOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    # + absolute floor: a conv bias in front of train-mode BatchNorm has an exactly-zero gradient
    # (both sides are 1e-8 .. 1e-7 roundoff noise)
    return np.abs(a - b).max() <= rtol * np.abs(b).max() + 1e-6


def test_graph_filter_gradients_match_reference(dev, training_golden):
    import gnn_pathplanning_amd.graphML as gml
    z, meta = training_golden
    idx = 0
    for m in meta:
        if m['kind'] not in ('GraphFilter', 'GraphFilterBatch'):
            continue
        k = 'f%d_' % idx
        idx += 1
        mod = getattr(gml, m['kind'])(m['G'], m['F'], m['K'], m['E'], True).to(dev)
        with torch.no_grad():
            mod.weight.copy_(torch.from_numpy(z[k + 'h']))
            mod.bias.copy_(torch.from_numpy(z[k + 'b']))
        x = torch.from_numpy(z[k + 'x']).to(dev).requires_grad_(True)
        mod.addGSO(torch.from_numpy(z[k + 'S']).to(dev))
        y = mod(x)
        assert y.requires_grad
        (y * torch.from_numpy(z[k + 'cot']).to(dev)).sum().backward()
        assert close(y.detach().cpu(), z[k + 'y'], 1e-4), m
        assert close(x.grad.cpu(), z[k + 'dx']), m
        assert close(mod.weight.grad.cpu(), z[k + 'dh']), m
        assert close(mod.bias.grad.cpu(), z[k + 'db']), m


@pytest.mark.parametrize('B,N,K,E,G,F_out', [(8, 10, 3, 1, 128, 128), (3, 50, 2, 1, 128, 128),
                                             (2, 100, 3, 1, 128, 128), (5, 6, 4, 2, 24, 40)])
def test_lsigf_autograd_vs_oracle(dev, B, N, K, E, G, F_out):
    import gnn_pathplanning_amd.graphML as gml
    g = torch.Generator().manual_seed(B * N + K)
    h = (torch.randn(F_out, E, K, G, generator=g) / (G * K) ** 0.5)
    b = torch.randn(F_out, 1, generator=g) * 0.1
    x = torch.randn(B, G, N, generator=g)
    S = orc.synth_gso_sparse(B * E, N, 4.0, seed=K).reshape(B, E, N, N)
    cot = torch.randn(B, F_out, N, generator=g)
    hc, bc, xc = h.clone().requires_grad_(True), b.clone().requires_grad_(True), x.clone().requires_grad_(True)
    (orc.batch_lsigf(hc, S, xc, bc) * cot).sum().backward()
    hg, bg, xg = (t.to(dev).requires_grad_(True) for t in (h, b, x))
    (gml.BatchLSIGF(hg, S.to(dev), xg, bg) * cot.to(dev)).sum().backward()
    assert close(xg.grad.cpu(), xc.grad) and close(hg.grad.cpu(), hc.grad) and close(bg.grad.cpu(), bc.grad)
    # only some inputs require grad
    x2 = x.to(dev).requires_grad_(True)
    y2 = gml.BatchLSIGF(h.to(dev), S.to(dev), x2, b.to(dev))
    (y2 * cot.to(dev)).sum().backward()
    assert close(x2.grad.cpu(), xc.grad)


def test_policy_training_step_matches_reference(dev, training_golden, policy_golden):
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from gnn_pathplanning_amd.training import policy_loss
    z, meta = training_golden
    zp, _ = policy_golden
    for ci, m in enumerate(meta):
        if m['kind'] != 'policy':
            continue

        class Cfg:
            num_agents, nGraphFilterTaps, device = m['N'], m['K'], dev
        net = DecentralPlannerNet(Cfg()).to(dev)
        net.load_state_dict(golden_state_dict(zp, m['K']))
        net.train()
        obs = torch.from_numpy(z['g%d_obs' % ci].astype(np.float32)).to(dev)
        S = torch.from_numpy(z['g%d_S' % ci]).to(dev)
        tgt = torch.from_numpy(z['g%d_target' % ci].astype(np.float32)).to(dev)
        net.addGSO(S)
        out = net(obs)
        assert isinstance(out, list) and len(out) == m['N'] and out[0].requires_grad
        loss = policy_loss(out, tgt)
        loss.backward()
        assert abs(loss.item() - float(z['g%d_loss' % ci])) <= 1e-5
        assert np.abs(torch.stack(out, 1).detach().cpu().numpy() - z['g%d_logits' % ci]).max() <= 1e-4
        grads = dict((n, p.grad) for n, p in net.named_parameters())
        assert list(grads) == m['param_names']
        for j, name in enumerate(m['param_names']):
            want = z['g%d_gradsum' % ci][j]
            assert abs(grads[name].double().norm().item() - want[2]) <= 5e-4 * max(1e-3, want[2]), name
            key = 'g%d_grad/%s' % (ci, name)
            if key in z.files:
                assert close(grads[name].cpu(), z[key], 5e-4), name
        bufs = dict(net.named_buffers())
        for key in z.files:
            if key.startswith('g%d_buf/' % ci):
                name = key.split('/', 1)[1]
                assert close(bufs[name].cpu().float(), z[key].astype(np.float32), 1e-4), name


def test_multilayer_training_matches_reference(dev, multilayer_training_golden, policy_golden):
    """gradient goldens for planners with L = 2 graph-filter layers and / or E = 2 edge features
    (tests/golden/training_multilayer.npz: the re-wired REAL reference in train mode, loss.backward()): logits, loss,
    the norm of every parameter's gradient, and the full gradients of every graph-filter layer, the head and
    compressMLP."""
    from conftest import multilayer_state_dict
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from gnn_pathplanning_amd.training import policy_loss
    z, meta = multilayer_training_golden
    zp, _ = policy_golden
    for ci, m in enumerate(meta):
        class C:
            num_agents, nGraphFilterTaps, device = m['N'], list(m['taps']), dev
            dimNodeSignals, numEdgeFeatures = list(m['dims']), m['E']
        net = DecentralPlannerNet(C()).to(dev)
        net.load_state_dict(multilayer_state_dict(zp, z, ci, prefix='t'))
        net.train()
        obs = torch.from_numpy(z['t%d_obs' % ci].astype(np.float32)).to(dev)
        S = torch.from_numpy(z['t%d_S' % ci]).to(dev)
        tgt = torch.from_numpy(z['t%d_target' % ci].astype(np.float32)).to(dev)
        net.addGSO(S.squeeze(1) if m['E'] == 1 else S)
        out = net(obs)
        loss = policy_loss(out, tgt)
        loss.backward()
        assert abs(loss.item() - float(z['t%d_loss' % ci])) <= 1e-5, (ci, loss.item())
        assert np.abs(torch.stack(out, 1).detach().cpu().numpy() - z['t%d_logits' % ci]).max() <= 1e-4
        grads = dict((n, p.grad) for n, p in net.named_parameters())
        assert list(grads) == m['param_names']
        full = 0
        for j, name in enumerate(m['param_names']):
            want = z['t%d_gradsum' % ci][j]
            assert abs(grads[name].double().norm().item() - want[2]) <= 5e-4 * max(1e-3, want[2]), (ci, name)
            key = 't%d_grad/%s' % (ci, name)
            if key in z.files:
                assert close(grads[name].cpu(), z[key], 5e-4), (ci, name)
                full += 1
        assert full >= 2 * len(m['dims']) + 3


def test_train_step_reduces_loss_and_eval_repacks(dev):
    """A few Adam steps on one batch (agents/decentralplannerlocal.py:59 settings) lower the loss,
    and the fused eval path afterwards uses the UPDATED weights (pack cache invalidation)."""
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from gnn_pathplanning_amd.training import train_step

    class Cfg:
        num_agents, nGraphFilterTaps, device = 10, 3, dev
    torch.manual_seed(0)
    net = DecentralPlannerNet(Cfg()).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5)
    B = 16
    obs = orc.synth_obs(B, 10, seed=2).to(dev)
    S = torch.from_numpy(orc.synth_gso_geometric(B, 10, 20, seed=2)).float().to(dev)
    g = torch.Generator().manual_seed(1)
    tgt = torch.nn.functional.one_hot(torch.randint(0, 5, (B, 10), generator=g), 5).float().to(dev)
    net.train()
    losses = [train_step(net, opt, obs, tgt, S).item() for _ in range(12)]
    assert losses[-1] < losses[0] - 0.05, losses
    net.eval()
    net.addGSO(S)
    got = torch.stack(net(obs), 1).cpu()
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    with torch.no_grad():
        want = torch.stack(orc.policy_forward(sd, S.cpu(), obs.cpu()), 1)
    assert (got - want).abs().max().item() <= 1e-4


def test_prepowered_gso_family_matches_reference(dev, training_golden):
    """matrixPowersBatch / batchLSIGF / GraphFilterBatchGSO (graphML.py:2063-2271)."""
    import gnn_pathplanning_amd.graphML as gml
    z, meta = training_golden
    idx = 0
    for m in meta:
        if m['kind'] != 'GraphFilterBatchGSO':
            continue
        k = 'p%d_' % idx
        idx += 1
        mod = gml.GraphFilterBatchGSO(m['G'], m['F'], m['K'], m['E'], True).to(dev)
        with torch.no_grad():
            mod.weight.copy_(torch.from_numpy(z[k + 'h']))
            mod.bias.copy_(torch.from_numpy(z[k + 'b']))
        S = torch.from_numpy(z[k + 'S']).to(dev)
        mod.addGSO(S)
        assert tuple(mod.SK.shape) == z[k + 'SK'].shape
        assert np.abs(mod.SK.cpu().numpy() - z[k + 'SK']).max() <= 1e-5
        x = torch.from_numpy(z[k + 'x']).to(dev)
        with torch.no_grad():
            y = mod(x)
        assert close(y.cpu(), z[k + 'y'], 1e-4), m
        assert 'number_nodes=%d, batch_size=%d' % (m['N'], m['B']) in mod.extra_repr()
        # differentiable too, and equal to the chained-shift layer on the same GSO
        xg = x.clone().requires_grad_(True)
        mod(xg).sum().backward()
        ref = gml.GraphFilterBatch(m['G'], m['F'], m['K'], m['E'], True).to(dev)
        with torch.no_grad():
            ref.weight.copy_(mod.weight)
            ref.bias.copy_(mod.bias)
        ref.addGSO(S if S.dim() == 4 else S.unsqueeze(1))
        xr = x.clone().requires_grad_(True)
        ref(xr).sum().backward()
        assert close(xg.grad.cpu(), xr.grad.cpu(), 1e-3)


def test_graphed_train_step_equals_eager(dev):
    """HIP-graph replay of the train step follows the eager step exactly (same kernels, same order)."""
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from gnn_pathplanning_amd.training import GraphedTrainStep, train_step

    class Cfg:
        num_agents, nGraphFilterTaps, device = 10, 3, dev
    B = 8
    g = torch.Generator().manual_seed(3)
    batches = []
    for i in range(6):
        obs = orc.synth_obs(B, 10, seed=40 + i).to(dev)
        S = torch.from_numpy(orc.synth_gso_geometric(B, 10, 20, seed=40 + i)).float().to(dev)
        tgt = torch.nn.functional.one_hot(torch.randint(0, 5, (B, 10), generator=g), 5).float().to(dev)
        batches.append((obs, tgt, S))
    sd0 = orc.init_state_dict(3, seed=9)

    def fresh():
        net = DecentralPlannerNet(Cfg()).to(dev).train()
        net.load_state_dict(sd0)
        return net, torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5, capturable=True)
    net_e, opt_e = fresh()
    for _ in range(3):                       # GraphedTrainStep runs 3 warm-up steps (capture only records)
        train_step(net_e, opt_e, *batches[0])
    eager = [train_step(net_e, opt_e, *b).item() for b in batches[1:]]
    net_g, opt_g = fresh()
    step = GraphedTrainStep(net_g, opt_g, *batches[0])
    graphed = [step(*b).item() for b in batches[1:]]
    assert np.allclose(eager, graphed, rtol=2e-4, atol=1e-6), (eager, graphed)
    for (k, a), (_, b) in zip(net_e.state_dict().items(), net_g.state_dict().items()):
        if a.dtype.is_floating_point:
            assert (a - b).abs().max().item() <= 2e-4 * max(1.0, a.abs().max().item()), k


@pytest.mark.parametrize('B', [8, 64])
def test_backward_weight_gradient_fork_is_bit_identical(dev, B):
    """r05: gnnpp_encoder_train_bwd runs the weight-gradient kernels of every layer on a second HIP stream (forks
    behind each layer's BatchNorm backward, joins before the call hands the stream back; GNNPP_TUNE_TRAIN_FORK = 2:
    always, 1: from 4096 agent-samples on, 0: never).  The
    same kernels on the same data: every gradient, the loss and the parameters after several optimisation steps are
    the one-stream run's BIT FOR BIT -- eager, and as a captured HIP graph (where the fork / join are graph edges) --
    and a race on the double-buffered dz / the shared partial-sum buffers would show up as run-to-run differences."""
    from gnn_pathplanning_amd import _native
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from gnn_pathplanning_amd.training import FusedAdam, GraphedTrainStep, train_step
    L = _native.lib()

    class Cfg:
        num_agents, nGraphFilterTaps, device = 10, 3, dev
    g = torch.Generator().manual_seed(5)
    batches = []
    for i in range(5):
        obs = orc.synth_obs(B, 10, seed=70 + i).to(dev)
        S = torch.from_numpy(orc.synth_gso_geometric(B, 10, 20, seed=70 + i)).float().to(dev)
        tgt = torch.nn.functional.one_hot(torch.randint(0, 5, (B, 10), generator=g), 5).float().to(dev)
        batches.append((obs, tgt, S))
    sd0 = orc.init_state_dict(3, seed=13)

    def run(fork, graphed, merged=0):
        # merged (r06b, GNNPP_TUNE_TRAIN_WGRAD_MERGED, the default): all five weight gradients as ONE launch behind the
        # chain -- the fork rule only applies to the per-layer launches (merged = 0)
        assert L.gnnpp_set_tuning(18, merged) == 0 and L.gnnpp_get_tuning(18) == merged
        assert L.gnnpp_set_tuning(15, fork) == 0 and L.gnnpp_get_tuning(15) == fork
        net = DecentralPlannerNet(Cfg()).to(dev).train()
        net.load_state_dict(sd0)
        opt = FusedAdam(net.parameters(), lr=1e-3, weight_decay=1e-5)
        losses, grads = [], None
        if graphed:
            step = GraphedTrainStep(net, opt, *batches[0])        # (3 eager warm-up steps, then the capture)
            losses = [step(*b).item() for b in batches[1:]]
        else:
            for _ in range(3):
                train_step(net, opt, *batches[0])
            for b in batches[1:]:
                losses.append(train_step(net, opt, *b).item())
            grads = [p.grad.clone() for p in net.parameters()]
        torch.cuda.synchronize()
        return losses, grads, [p.detach().clone() for p in net.parameters()]
    try:
        ref = run(0, False)
        for fork, graphed, merged in ((2, False, 0), (2, False, 0), (2, True, 0), (0, True, 0), (1, False, 0),
                                      (1, False, 1), (1, True, 1), (2, True, 1)):
            got = run(fork, graphed, merged)
            assert got[0] == ref[0], (fork, graphed, got[0], ref[0])
            assert all(torch.equal(a, b) for a, b in zip(got[2], ref[2])), (fork, graphed)
            if got[1] is not None:
                assert all(torch.equal(a, b) for a, b in zip(got[1], ref[1])), (fork, graphed)
        assert L.gnnpp_set_tuning(15, 3) == -1 and L.gnnpp_set_tuning(18, 2) == -1
    finally:
        L.gnnpp_set_tuning(15, 1)
        L.gnnpp_set_tuning(18, 1)


def test_small_cotangents_keep_relative_accuracy(dev):
    """dy of 1e-5 .. 1e-8 (what CrossEntropy / (B N) hands the filter at B = 64) sits in the f16
    subnormal range: the input-gradient launch therefore contracts on the exact fp32 MFMA, and the
    gradients keep fp32 RELATIVE accuracy (no absolute floor in this check).  A per-node
    bias [F,N] trains too."""
    import gnn_pathplanning_amd.graphML as gml
    g = torch.Generator().manual_seed(5)
    B, N, K, G, F_out = 16, 10, 3, 128, 128
    h = torch.randn(F_out, 1, K, G, generator=g) / (G * K) ** 0.5
    b = torch.randn(F_out, N, generator=g) * 0.1
    x = torch.randn(B, G, N, generator=g)
    S = orc.synth_gso_sparse(B, N, 3.5, seed=9).unsqueeze(1)
    for cot_scale in (1.0, 1e-5, 1e-8):
        cot = torch.randn(B, F_out, N, generator=g) * cot_scale
        hc, bc, xc = (t.clone().double().requires_grad_(True) for t in (h, b, x))
        yc = torch.from_numpy(orc.lsigf_f64(h.numpy(), S.numpy(), x.numpy(), b.numpy()))
        # float64 autograd reference of the same algebra
        z, y64 = xc, 0
        for k in range(K):
            if k:
                z = torch.matmul(z, S[:, 0].double())
            y64 = y64 + torch.einsum('fg,bgn->bfn', hc[:, 0, k], z)
        y64 = y64 + bc
        (y64 * cot.double()).sum().backward()
        hd, bd, xd = (t.clone().to(dev).requires_grad_(True) for t in (h, b, x))
        y = gml.BatchLSIGF(hd, S.to(dev), xd, bd)
        (y * cot.to(dev)).sum().backward()
        assert (y.detach().cpu().double() - yc).abs().max().item() <= 1e-4 * yc.abs().max().item()
        for got, want in ((xd.grad, xc.grad), (hd.grad, hc.grad), (bd.grad, bc.grad)):
            err = (got.cpu().double() - want).abs().max().item()
            assert err <= 3e-5 * want.abs().max().item(), (cot_scale, err, want.abs().max().item())


def test_graphed_train_step_then_eval_uses_new_weights(dev):
    """HIP-graph replays update the parameters behind torch's version counters: the eval path must
    still see the new weights (PackCache served stale packs)."""
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from gnn_pathplanning_amd.training import GraphedTrainStep
    B, N = 8, 6

    class C:
        num_agents, nGraphFilterTaps, device = N, 3, dev
    net = DecentralPlannerNet(C()).to(dev)
    net.load_state_dict(orc.init_state_dict(3, seed=6))
    obs = orc.synth_obs(B, N, seed=2).to(dev)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, 12, seed=2)).float().to(dev)
    g = torch.Generator().manual_seed(1)
    tgt = torch.nn.functional.one_hot(torch.randint(0, 5, (B, N), generator=g), 5).float().to(dev)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-2, capturable=True)
    step = GraphedTrainStep(net, opt, obs, tgt, S)
    for _ in range(2):
        step(obs, tgt, S)
    net.eval()
    net.addGSO(S)
    a = net.forward_logits(obs).clone()                    # packs built from the current weights
    net.train()
    for _ in range(5):
        step(obs, tgt, S)                                   # replays: weights move, versions do not
    net.eval()
    net.addGSO(S)
    b = net.forward_logits(obs).clone()
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    with torch.no_grad():
        want = torch.stack(orc.policy_forward(sd, S.cpu(), obs.cpu()), 0)
    assert (b.cpu() - want).abs().max().item() <= 1e-4
    assert (a - b).abs().max().item() > 1e-3


@pytest.mark.parametrize('B,N', [(64, 10), (7, 3), (2, 16), (130, 4)])
def test_hip_training_encoder_matches_aten_path(dev, B, N):
    """The hand-written train-mode encoder (csrc/train_encoder.hip, forward + backward) against the same
    forward on stock aten / MIOpen ops (agents as convolution groups) at BASELINE config 4's batch:
    loss, logits, every parameter gradient and the BatchNorm running statistics."""
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from gnn_pathplanning_amd.training import policy_loss

    class C:
        num_agents, nGraphFilterTaps, device = N, 3, dev
    sd = orc.init_state_dict(3, seed=B + N)
    obs = orc.synth_obs(B, N, seed=3).to(dev)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, 20, seed=3)).float().to(dev)
    g = torch.Generator().manual_seed(2)
    tgt = torch.nn.functional.one_hot(torch.randint(0, 5, (B, N), generator=g), 5).float().to(dev)
    res = []
    for which in ('hip', 'aten'):
        net = DecentralPlannerNet(C()).to(dev)
        net.load_state_dict(sd)
        net.train()
        net.addGSO(S)
        out = net(obs) if which == 'hip' else net._forward_train_aten(obs)
        loss = policy_loss(out, tgt)
        loss.backward()
        res.append((loss.item(), torch.stack(out, 1).detach().cpu(),
                    {k: p.grad.detach().cpu() for k, p in net.named_parameters()},
                    {k: b.detach().cpu().clone() for k, b in net.named_buffers() if 'running' in k}))
    (l0, o0, g0, r0), (l1, o1, g1, _) = res
    assert abs(l0 - l1) <= 1e-5 and (o0 - o1).abs().max().item() <= 1e-4
    for k in g0:
        if k.startswith('ConvLayers.') and k.endswith('.bias') and int(k.split('.')[1]) in (0, 4, 7, 11, 14):
            continue                                       # conv bias before train-mode BN: zero gradient, roundoff
        assert close(g0[k], g1[k], 5e-4), (k, (g0[k] - g1[k]).abs().max().item(), g1[k].abs().max().item())
    # running statistics after ONE forward against the oracle's per-agent-call loop
    sd2 = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        orc.policy_forward(sd2, S.cpu(), obs.cpu(), training=True)
    for k, v in r0.items():
        assert (v - sd2[k]).abs().max().item() <= 2e-5, k


def test_c4_batch_gradients_match_the_oracle(dev):
    """VERDICT r05 (weak 4): at the FULL size of BASELINE config 4's per-GPU batch (64 graphs x 10 agents, K = 3) every
    gradient of the HIP training step against the pinned oracle -- oracle.policy_forward(training=True) (the
    reference's per-agent-call loop, itself checked against tests/golden/training_grads.npz by
    tests/test_oracle_training.py) + torch autograd on the CPU --, with the tolerances of the golden test: loss 1e-5,
    logits 1e-4, gradients 5e-4 of the tensor's largest magnitude, running statistics 1e-4.  Through train_step()'s own
    loss launch (gnnpp_policy_loss) as well as through policy_loss()."""
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from gnn_pathplanning_amd.training import _policy_loss_and_grad, policy_loss
    B, N, K = 64, 10, 3

    class C:
        num_agents, nGraphFilterTaps, device = N, K, dev
    sd0 = orc.init_state_dict(K, seed=1337)
    obs = orc.synth_obs(B, N, seed=1337)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, 20, seed=1337)).float()
    tgt = torch.nn.functional.one_hot(torch.randint(0, 5, (B, N), generator=torch.Generator().manual_seed(0)), 5).float()
    sd = {k: v.clone() for k, v in sd0.items()}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype == torch.float32 and 'running' not in k}
    sd.update(params)
    want_out = orc.policy_forward(sd, S, obs, training=True)
    want_loss = orc.policy_loss(want_out, tgt)
    want_loss.backward()
    for fused_loss in (False, True):
        net = DecentralPlannerNet(C()).to(dev)
        net.load_state_dict(sd0)
        net.train()
        net.addGSO(S.to(dev))
        out = net(obs.to(dev))
        if fused_loss:
            loss, dlogits = _policy_loss_and_grad(out.stacked, tgt.to(dev))
            out.stacked.backward(dlogits)
        else:
            loss = policy_loss(out, tgt.to(dev))
            loss.backward()
        assert abs(loss.item() - want_loss.item()) <= 1e-5
        assert (torch.stack(list(out), 1).detach().cpu() - torch.stack(want_out, 1).detach()).abs().max().item() <= 1e-4
        names = [n for n, _ in net.named_parameters()]
        assert sorted(names) == sorted(params)
        for name, p in net.named_parameters():
            assert close(p.grad.cpu(), sd[name].grad, 5e-4), \
                (fused_loss, name, (p.grad.cpu() - sd[name].grad).abs().max().item(), sd[name].grad.abs().max().item())
        for name, b in net.named_buffers():
            if 'running' in name:
                assert close(b.cpu(), sd[name].detach(), 1e-4), name


@pytest.mark.parametrize('B,N', [(64, 10), (7, 3), (33, 50)])
def test_running_statistics_inside_the_last_batchnorm_launch(dev, B, N):
    """r06b: the BatchNorm running-statistics update rides in the forward's LAST BatchNorm launch
    (GNNPP_TUNE_TRAIN_RUNNING_FUSED): the N workgroups of a channel tile publish the statistics they computed with an
    agent-scope release / ticket / acquire hand-off (per-XCD L2s are not coherent) and the last one to arrive runs the N
    sequential updates.  Against the launch of its own (knob 19 = 0): the same running_mean / running_var /
    num_batches_tracked BIT FOR BIT over many repetitions (a stale cross-XCD read would show up as a difference), and
    against the oracle's per-agent-call loop."""
    from gnn_pathplanning_amd import _native
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    L = _native.lib()

    class C:
        num_agents, nGraphFilterTaps, device = N, 3, dev
    sd0 = orc.init_state_dict(3, seed=B + N)
    obs = [orc.synth_obs(B, N, seed=s).to(dev) for s in range(4)]
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, 20 if N <= 10 else 50, seed=1)).float().to(dev)

    def run(fused, reps):
        assert L.gnnpp_set_tuning(19, fused) == 0 and L.gnnpp_get_tuning(19) == fused
        net = DecentralPlannerNet(C()).to(dev)
        net.load_state_dict(sd0)
        net.train()
        net.addGSO(S)
        with torch.no_grad():
            for r in range(reps):
                net(obs[r % 4])
        torch.cuda.synchronize()
        return {k: b.detach().clone() for k, b in net.named_buffers()}
    try:
        reps = 60
        ref = run(0, reps)
        for _ in range(3):
            got = run(1, reps)
            for k in ref:
                assert torch.equal(got[k], ref[k]), k
        one = run(1, 1)
    finally:
        L.gnnpp_set_tuning(19, 1)
    sd2 = {k: v.clone() for k, v in sd0.items()}
    with torch.no_grad():
        orc.policy_forward(sd2, S.cpu(), obs[0].cpu(), training=True)
    for k, v in one.items():
        if 'running' in k:
            assert (v.cpu() - sd2[k]).abs().max().item() <= 2e-5, k
        elif 'num_batches' in k:
            assert int(v) == int(sd0[k]) + N, k


def test_deferred_parameter_gradient_products(dev):
    """r06b: the products that only yield parameter gradients (the action head's and the graph filter's dW / dh / db) are
    not on the backward chain; they wait in _native's queue and ride with the compress layer's backward launch.  Queued
    only while the parameter has no `.grad` yet (autograd then stores the result tensor without reading it): a SECOND
    backward without zero_grad() must accumulate correctly, and a pass that never reaches the flushing node (gradients
    of the head alone) is completed by the engine's final callback."""
    from gnn_pathplanning_amd import _native
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from gnn_pathplanning_amd.training import policy_loss
    B, N, K = 16, 10, 3

    class C:
        num_agents, nGraphFilterTaps, device = N, K, dev
    sd0 = orc.init_state_dict(K, seed=21)
    obs = orc.synth_obs(B, N, seed=21).to(dev)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, 20, seed=21)).float().to(dev)
    tgt = torch.nn.functional.one_hot(torch.randint(0, 5, (B, N), generator=torch.Generator().manual_seed(1)), 5).float().to(dev)

    def fresh():
        net = DecentralPlannerNet(C()).to(dev)
        net.load_state_dict(sd0)
        net.train()
        net.addGSO(S)
        return net
    # deferral is opt-in per backward pass (training.train_step): a plain loss.backward() computes every gradient inside
    # the node that returns it -- a tensor hook (or DistributedDataParallel's accumulate hooks) reads it right there
    net0 = fresh()
    seen = {}
    net0.actionsMLP[0].weight.register_hook(lambda g: seen.__setitem__('head', g.clone()))
    net0.GFL[0].weight.register_hook(lambda g: seen.__setitem__('taps', g.clone()))
    queued = []
    orig_defer = _native.defer_gemms
    _native.defer_gemms = lambda specs, prms: (queued.append(len(specs)), orig_defer(specs, prms))[1]
    try:
        policy_loss(net0(obs), tgt).backward()
        assert queued == [] and torch.equal(seen['head'], net0.actionsMLP[0].weight.grad)
        assert torch.equal(seen['taps'], net0.GFL[0].weight.grad)
        net = fresh()
        with _native.allow_deferred_gemms():
            policy_loss(net(obs), tgt).backward()
        assert queued == [2, 2]                              # the head's and the filter's dW / dh + db waited
    finally:
        _native.defer_gemms = orig_defer
    assert not _native._deferred_gemms                       # everything queued was launched inside the pass
    once = {k: p.grad.clone() for k, p in net.named_parameters()}
    for k, p in net0.named_parameters():
        assert torch.equal(p.grad, once[k]), k               # the same kernels on the same data: the same bits
    net.load_state_dict(sd0)                                 # (the first forward moved the running statistics)
    with _native.allow_deferred_gemms():
        policy_loss(net(obs), tgt).backward()                # accumulate: `.grad` exists -> nothing may be deferred
    for k, p in net.named_parameters():
        assert close(p.grad.cpu(), 2 * once[k].cpu(), 1e-5), k
    # a partial pass: only the head's parameters -> the compress layer's node (the flushing one) never runs
    net2 = fresh()
    out = net2(obs)
    with _native.allow_deferred_gemms():
        gw, gb = torch.autograd.grad(policy_loss(out, tgt), [net2.actionsMLP[0].weight, net2.actionsMLP[0].bias])
    assert not _native._deferred_gemms
    assert torch.equal(gw, once['actionsMLP.0.weight']) and torch.equal(gb, once['actionsMLP.0.bias'])
    # ... and the filter's alone
    net3 = fresh()
    with _native.allow_deferred_gemms():
        gh, = torch.autograd.grad(policy_loss(net3(obs), tgt), [net3.GFL[0].weight])
    assert not _native._deferred_gemms and torch.equal(gh, once['GFL.0.weight'])


def test_fused_adam_and_loss_match_torch(dev):
    """The one-launch pieces of the optimisation step against stock torch on the same model and batches:
    policy_loss_fused == policy_loss (value and gradient of every parameter), and FusedAdam (gnnpp_adam_step)
    tracks torch.optim.Adam(lr, weight_decay) over several steps."""
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet, LogitList
    from gnn_pathplanning_amd.training import FusedAdam, policy_loss, policy_loss_fused

    class Cfg:
        num_agents, nGraphFilterTaps, device = 10, 3, dev
    B = 24
    obs = orc.synth_obs(B, 10, seed=8).to(dev)
    S = torch.from_numpy(orc.synth_gso_geometric(B, 10, 20, seed=8)).float().to(dev)
    g = torch.Generator().manual_seed(4)
    tgt = torch.nn.functional.one_hot(torch.randint(0, 5, (B, 10), generator=g), 5).float().to(dev)
    torch.manual_seed(5)
    net_a = DecentralPlannerNet(Cfg()).to(dev).train()
    net_b = DecentralPlannerNet(Cfg()).to(dev).train()
    net_b.load_state_dict(net_a.state_dict())
    opt_a = torch.optim.Adam(net_a.parameters(), lr=1e-3, weight_decay=1e-5)
    opt_b = FusedAdam(net_b.parameters(), lr=1e-3, weight_decay=1e-5)
    for it in range(5):
        losses = []
        for net, opt, lossf in ((net_a, opt_a, policy_loss), (net_b, opt_b, policy_loss_fused)):
            opt.zero_grad()
            net.addGSO(S)
            out = net(obs)
            assert isinstance(out, LogitList) and len(out) == 10 and out[0].shape == (B, 5)
            loss = lossf(out, tgt)
            loss.backward()
            losses.append(loss.item())
        # identical weights at it = 0; afterwards the two optimizers' last bits differ and the (inert, see below) conv
        # biases walk apart by +-lr per step, which fp32 rounding turns into ~1e-5 of loss
        assert abs(losses[0] - losses[1]) <= (2e-6 if it == 0 else 1e-4) * max(1.0, abs(losses[0])), (it, losses)
        if it == 0:                                      # identical weights: gradients comparable one to one
            for (k, pa), pb in zip(net_a.named_parameters(), net_b.parameters()):
                assert (pa.grad - pb.grad).abs().max().item() <= 1e-6 + 1e-5 * pa.grad.abs().max().item(), k
        opt_a.step()
        opt_b.step()
    conv_bias = {'ConvLayers.%d.bias' % i for i in (0, 4, 7, 11, 14)}
    for (k, pa), pb in zip(net_a.named_parameters(), net_b.parameters()):
        if k in conv_bias:
            # a bias in front of a train-mode BatchNorm has a gradient of exactly zero: what arrives is rounding
            # noise (~1e-9), which Adam's normalisation turns into full-size +-lr steps -- two correct
            # implementations whose last bits differ walk these (inert) parameters apart
            continue
        # 5 steps of lr 1e-3: updates ~5e-3.  Entries whose gradient is noise-level get full-size Adam steps whose SIGN
        # follows the noise, and the (inert) biases above feed that noise differently into the two nets from the second
        # step on: the trajectories agree to a fraction of one step, not to rounding.  (The update rule itself is pinned
        # to rounding, on well-conditioned gradients, by test_fused_adam_checkpoint_round_trip.)
        # -> bound the worst entry by two full steps and the typical entry tightly
        assert (pa - pb).abs().max().item() <= 2e-3, k
        assert (pa - pb).abs().mean().item() <= 1e-4, k


def test_fused_adam_checkpoint_round_trip(dev):
    """save -> load -> step: the reference checkpoints optimizer.state_dict()
    (agents/decentralplannerlocal.py:125-134).  After load_state_dict() the device step counter and the moments
    must be the loaded ones (bias correction continues at t, not 0) and the cached pointer tables must not
    point at the freed moment tensors: a resumed FusedAdam keeps tracking a resumed torch.optim.Adam."""
    import io
    from gnn_pathplanning_amd.training import FusedAdam
    torch.manual_seed(3)
    shapes = [(33, 7), (128,), (5, 128), (64, 32, 3, 3)]
    pa = [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = torch.optim.Adam(pa, lr=1e-2, weight_decay=1e-4)
    ob = FusedAdam(pb, lr=1e-2, weight_decay=1e-4)

    def grads(step):
        g = torch.Generator().manual_seed(100 + step)
        return [torch.randn(*s, generator=g).to(dev) for s in shapes]

    def run(o, ps, step):
        for p, gr in zip(ps, grads(step)):
            p.grad = gr.clone()
        o.step()
    for it in range(4):
        run(oa, pa, it); run(ob, pb, it)
    # checkpoint both, continue in FRESH optimizers (ob2 also took a step before the load: stale pointer tables)
    bufa, bufb = io.BytesIO(), io.BytesIO()
    torch.save(oa.state_dict(), bufa); torch.save(ob.state_dict(), bufb)
    assert any(isinstance(k, str) and k.startswith('gnnpp_group_') for k in ob.state_dict()['state'])
    pa2 = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    pb2 = [torch.nn.Parameter(p.detach().clone()) for p in pb]
    oa2 = torch.optim.Adam(pa2, lr=1e-2, weight_decay=1e-4)
    ob2 = FusedAdam(pb2, lr=1e-2, weight_decay=1e-4)
    snapshot = [p.detach().clone() for p in pb2]
    run(ob2, pb2, 99)                                    # builds tables on moments the load will replace
    with torch.no_grad():
        for p, s in zip(pb2, snapshot):
            p.copy_(s)
    bufa.seek(0); bufb.seek(0)
    oa2.load_state_dict(torch.load(bufa, map_location='cpu'))
    ob2.load_state_dict(torch.load(bufb, map_location='cpu'))
    assert float(ob2.state['gnnpp_group_0']['counter'][0]) == 4.0
    for it in range(4, 8):
        run(oa2, pa2, it); run(ob2, pb2, it)
        run(oa, pa, it); run(ob, pb, it)                 # the uninterrupted runs
    for a, b, a2, b2 in zip(pa, pb, pa2, pb2):
        assert (a2 - a).abs().max().item() <= 1e-6, 'torch Adam itself resumes exactly'
        assert (b2 - b).abs().max().item() <= 1e-6, 'resumed FusedAdam == uninterrupted FusedAdam'
        assert (a2 - b2).abs().max().item() <= 2e-5
    assert float(ob2.state['gnnpp_group_0']['counter'][0]) == 8.0


def test_graphed_train_step_with_fused_adam(dev):
    """The whole step with FusedAdam captured in a HIP graph (the step counter lives on the device) replays
    to the same losses as eager steps from the same start."""
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from gnn_pathplanning_amd.training import FusedAdam, GraphedTrainStep, train_step

    class Cfg:
        num_agents, nGraphFilterTaps, device = 10, 3, dev
    B = 16
    batches = []
    for i in range(6):
        obs = orc.synth_obs(B, 10, seed=60 + i).to(dev)
        S = torch.from_numpy(orc.synth_gso_geometric(B, 10, 20, seed=60 + i)).float().to(dev)
        g = torch.Generator().manual_seed(i)
        tgt = torch.nn.functional.one_hot(torch.randint(0, 5, (B, 10), generator=g), 5).float().to(dev)
        batches.append((obs, tgt, S))

    def make():
        torch.manual_seed(11)
        net = DecentralPlannerNet(Cfg()).to(dev).train()
        return net, FusedAdam(net.parameters(), lr=1e-3, weight_decay=1e-5)
    net_e, opt_e = make()
    for _ in range(3):                                   # GraphedTrainStep warms up with 3 steps on batch 0
        train_step(net_e, opt_e, *batches[0])
    eager = [train_step(net_e, opt_e, *b).item() for b in batches]
    net_g, opt_g = make()
    step = GraphedTrainStep(net_g, opt_g, *batches[0])
    first = step(*batches[0]).item()                     # the capture itself is not a step
    graphed = [first] + [step(*b).item() for b in batches[1:]]
    for a, b in zip(eager, graphed):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(a)), (eager, graphed)


def test_train_mode_gso_with_more_nodes_than_agents(dev):
    """a GSO with more nodes than numAgents (the reference's GraphFilterBatch zero-pads the signal,
    graphML.py:2464-2469) works in TRAIN mode like in eval mode: same logits as the train-mode forward on the padded
    graph restricted to the team, gradients flow."""
    from gnn_pathplanning_amd import _native
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet

    class Cfg:
        num_agents, nGraphFilterTaps, device = 6, 3, dev
    torch.manual_seed(3)
    net = DecentralPlannerNet(Cfg()).to(dev).train()
    B, N, Ns = 5, 6, 9
    obs = orc.synth_obs(B, N, seed=5).to(dev)
    S = torch.from_numpy(orc.synth_gso_geometric(B, Ns, 14, seed=5)).float().to(dev)
    net.addGSO(S)
    out = net(obs)
    assert len(out) == N and out[0].shape == (B, 5)
    torch.stack(list(out), 0).square().sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    # oracle: the reference's train-mode forward with the [B,Ns,Ns] GSO (it pads the [B,128,N] signal itself)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    net2 = DecentralPlannerNet(Cfg()).to(dev).train()
    net2.load_state_dict(sd)
    with torch.no_grad():
        want = orc.policy_forward(sd, S.cpu(), obs.cpu(), training=True)
    net2.addGSO(S)
    got = net2(obs)
    err = max((g.detach().cpu() - w).abs().max().item() for g, w in zip(got, want))
    assert err <= 1e-4, err
    with pytest.raises(_native.GnnppError):
        net2.addGSO(S[:, :4, :4])                          # fewer nodes than agents: a clear error, not an assert
        net2(obs)


@pytest.mark.parametrize('N,E,Nin,per_node', [(120, 1, 120, False), (130, 2, 125, True)])
def test_training_on_graphs_larger_than_one_workgroup(dev, N, E, Nin, per_node):
    """VERDICT r02 "missing" 5: the reference trains on any graph size (BatchLSIGF is a chain of matmuls,
    graphML.py:2273-2367).  N > 112 nodes: forward AND backward as dense exact-fp32 GEMMs on gnnpp_gemm_kmajor
    (_LSIGFFunction's large path) -- output, dh, dx, db against torch autograd over the oracle's restatement; shared and
    per-sample GSOs, several edge features, fewer signal nodes than graph nodes, per-node bias, fused ReLU."""
    import gnn_pathplanning_amd.graphML as gml
    g = torch.Generator().manual_seed(N + E)
    B, K, G, F_out = 3, 3, 40, 24
    h0 = (torch.rand(F_out, E, K, G, generator=g) * 2 - 1) / (G * K * E) ** 0.5
    b0 = torch.randn(F_out, N if per_node else 1, generator=g) * 0.1
    x0 = torch.randn(B, G, Nin, generator=g)
    S = torch.stack([orc.synth_gso_sparse(B, N, 6.0, seed=e + 3) for e in range(E)], 1)        # [B,E,N,N]
    w = torch.randn(B, F_out, Nin, generator=g)                                                 # cotangent
    # reference: the oracle's restatement on CPU, zero-padded like the module (graphML.py:2464-2470)
    hr, br, xr = h0.clone().requires_grad_(), b0.clone().requires_grad_(), x0.clone().requires_grad_()
    xp = torch.cat([xr, xr.new_zeros(B, G, N - Nin)], 2) if Nin != N else xr
    yr = orc.batch_lsigf(hr, S, xp, br)[:, :, :Nin]
    (yr * w).sum().backward()
    gf = gml.GraphFilterBatch(G, F_out, K, E, bias=True).to(dev)
    with torch.no_grad():
        gf.weight.copy_(h0)
        if per_node:
            gf.bias = torch.nn.Parameter(b0.clone().to(dev))
        else:
            gf.bias.copy_(b0)
    gf.addGSO(S.to(dev))
    xd = x0.clone().to(dev).requires_grad_()
    y = gf(xd)
    assert y.shape == (B, F_out, Nin)
    (y * w.to(dev)).sum().backward()
    scale = max(1.0, yr.abs().max().item())
    assert (y.detach().cpu() - yr.detach()).abs().max().item() <= 1e-4 * scale
    for got, want, name in ((gf.weight.grad, hr.grad, 'dh'), (xd.grad, xr.grad, 'dx'), (gf.bias.grad, br.grad, 'db')):
        assert got is not None, name
        assert (got.cpu() - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item()), name
    # functional forms: shared GSO (LSIGF) and the node-major + ReLU form the train-mode planner uses
    h1, x1 = h0.clone().to(dev).requires_grad_(), torch.randn(B, G, N, generator=g).to(dev).requires_grad_()
    ys = gml.LSIGF(h1, S[0].to(dev), x1, None)
    hs, xs = h0.clone().requires_grad_(), x1.detach().cpu().clone().requires_grad_()
    ys_r = orc.lsigf(hs, S[0], xs, None)
    ys.square().sum().backward(); ys_r.square().sum().backward()
    assert (ys.detach().cpu() - ys_r.detach()).abs().max().item() <= 1e-4 * max(1.0, ys_r.abs().max().item())
    assert (h1.grad.cpu() - hs.grad).abs().max().item() <= 2e-4 * max(1.0, hs.grad.abs().max().item())
    assert (x1.grad.cpu() - xs.grad).abs().max().item() <= 2e-4 * max(1.0, xs.grad.abs().max().item())
    h2, x2 = h0.clone().to(dev).requires_grad_(), torch.randn(B, N, G, generator=g).to(dev).requires_grad_()
    yn = gml._LSIGFFunction.apply(h2, S.to(dev), x2, None, True, None, True, True)             # node-major, ReLU fused
    hn, xn = h0.clone().requires_grad_(), x2.detach().cpu().clone().requires_grad_()
    yn_r = torch.relu(orc.batch_lsigf(hn, S, xn.permute(0, 2, 1), None)).permute(0, 2, 1)
    yn.square().sum().backward(); yn_r.square().sum().backward()
    assert (yn.detach().cpu() - yn_r.detach()).abs().max().item() <= 1e-4 * max(1.0, yn_r.abs().max().item())
    assert (h2.grad.cpu() - hn.grad).abs().max().item() <= 2e-4 * max(1.0, hn.grad.abs().max().item())
    assert (x2.grad.cpu() - xn.grad).abs().max().item() <= 2e-4 * max(1.0, xn.grad.abs().max().item())


def test_planner_training_step_with_more_agents_than_one_workgroup(dev):
    """A train-mode forward + backward of DecentralPlannerNet with 120 agents (GSO rows beyond one workgroup's LDS):
    loss and the gradients of the graph filter, the action head and compressMLP against torch autograd over the oracle's
    train-mode restatement (per-agent-call BatchNorm statistics)."""
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from gnn_pathplanning_amd.training import policy_loss
    B, N = 3, 120

    class C:
        num_agents, nGraphFilterTaps, device = N, 3, dev
    sd = orc.init_state_dict(3, seed=77)
    obs = orc.synth_obs(B, N, seed=5)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, 100, seed=5)).float()
    g = torch.Generator().manual_seed(4)
    tgt = torch.nn.functional.one_hot(torch.randint(0, 5, (B, N), generator=g), 5).float()
    net = DecentralPlannerNet(C()).to(dev)
    net.load_state_dict(sd)
    net.train()
    net.addGSO(S.to(dev))
    out = net(obs.to(dev))
    loss = policy_loss(out, tgt.to(dev))
    loss.backward()
    sdr = {k: (v.clone().requires_grad_() if v.is_floating_point() and 'running' not in k else v.clone())
           for k, v in sd.items()}
    out_r = orc.policy_forward(sdr, S, obs, training=True)
    loss_r = orc.policy_loss(out_r, tgt)
    loss_r.backward()
    assert abs(loss.item() - loss_r.item()) <= 2e-5 * max(1.0, abs(loss_r.item()))
    assert (torch.stack(list(out), 0).detach().cpu() - torch.stack(out_r, 0).detach()).abs().max().item() <= 2e-4
    grads = dict(net.named_parameters())
    for k in ('GFL.0.weight', 'GFL.0.bias', 'actionsMLP.0.weight', 'actionsMLP.0.bias', 'compressMLP.0.weight',
              'ConvLayers.14.weight'):
        want = sdr[k].grad
        got = grads[k].grad.cpu()
        assert close(got, want, 1e-3), (k, (got - want).abs().max().item(), want.abs().max().item())


@pytest.mark.parametrize('G,F_out', [(512, 16), (16, 512)])
def test_training_with_features_wider_than_one_workgroup_holds(dev, G, F_out):
    """50 nodes fit the kernels, but 512 input features per node do not fit a workgroup's LDS
    (GNNPP_ERR_UNSUPPORTED from the forward launch at G = 512; from the input-gradient launch -- the transposed filter,
    whose input features are F -- at F = 512).  Training must still work: the dense path takes over for exactly the
    launch that does not fit.  Output and gradients against torch autograd over the oracle."""
    import gnn_pathplanning_amd.graphML as gml
    g = torch.Generator().manual_seed(G)
    B, N, K, E = 2, 50, 3, 1
    h0 = (torch.rand(F_out, E, K, G, generator=g) * 2 - 1) / (G * K) ** 0.5
    b0 = torch.randn(F_out, 1, generator=g) * 0.1
    x0 = torch.randn(B, G, N, generator=g)
    S = orc.synth_gso_sparse(B, N, 5.0, seed=2).unsqueeze(1)
    hr, br, xr = h0.clone().requires_grad_(), b0.clone().requires_grad_(), x0.clone().requires_grad_()
    yr = orc.batch_lsigf(hr, S, xr, br)
    yr.square().sum().backward()
    hd, bd, xd = (t.clone().to(dev).requires_grad_() for t in (h0, b0, x0))
    y = gml.BatchLSIGF(hd, S.to(dev), xd, bd)
    y.square().sum().backward()
    assert (y.detach().cpu() - yr.detach()).abs().max().item() <= 1e-4 * max(1.0, yr.abs().max().item())
    for got, want, name in ((hd.grad, hr.grad, 'dh'), (xd.grad, xr.grad, 'dx'), (bd.grad, br.grad, 'db')):
        assert (got.cpu() - want).abs().max().item() <= 2e-4 * max(1.0, want.abs().max().item()), name


def _dp_rank(rank, world, port, q):
    """One data-parallel rank (gloo collectives, BOTH ranks on cuda:0): two real training steps of the planner with
    FlatBucketDP; reports a gradient check of the first step and the parameters after the second."""
    import os
    import torch.distributed as dist
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
        from gnn_pathplanning_amd.training import FlatBucketDP, FusedAdam, policy_loss
        dev = torch.device('cuda', 0)
        B, N = 4, 10

        class C:
            num_agents, nGraphFilterTaps, device = N, 3, dev
        torch.manual_seed(100 + rank)                            # DIFFERENT initial weights per rank: the broadcast must fix it
        net = DecentralPlannerNet(C()).to(dev).train()
        dp = FlatBucketDP(net)
        opt = FusedAdam(net.parameters(), lr=1e-3, weight_decay=1e-5)
        obs = orc.synth_obs(B, N, seed=50 + rank).to(dev)        # each rank its own shard
        S = torch.from_numpy(orc.synth_gso_geometric(B, N, 20, seed=50 + rank)).float().to(dev)
        g = torch.Generator().manual_seed(rank)
        tgt = torch.nn.functional.one_hot(torch.randint(0, 5, (B, N), generator=g), 5).float().to(dev)
        grad_err = None
        for it in range(2):
            opt.zero_grad()
            net.addGSO(S)
            policy_loss(net(obs), tgt).backward()
            if it == 0:
                local = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu()
                both = [torch.empty_like(local) for _ in range(world)]
                dist.all_gather(both, local)                     # (gloo on CPU copies: the reference for the average)
            born_in_bucket = all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(dp.params, dp.views))
            kernels = None
            if it == 1:                                          # kernel launches inside the exchange
                try:
                    from torch.profiler import profile, ProfilerActivity
                    torch.cuda.synchronize()
                    with profile(activities=[ProfilerActivity.CUDA]) as prof:
                        dp.reduce_gradients()
                        torch.cuda.synchronize()
                    kernels = sum(1 for e in prof.events() if 'cuda' in str(e.device_type).lower()
                                  and 'memcpy' not in e.name.lower() and 'memset' not in e.name.lower())
                except Exception:                                # (a profiler that is not available: the counter below still holds)
                    kernels = None
                    dp.reduce_gradients()
            else:
                dp.reduce_gradients()
            copies = dp.last_reduce_copies
            if it == 0:
                got = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu()
                grad_err = (got - sum(both) / world).abs().max().item()
            opt.step()
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu()
        q.put((rank, grad_err, flat.numpy(), born_in_bucket, copies, kernels))
    finally:
        dist.destroy_process_group()


def test_two_rank_data_parallel_training_on_one_gpu(dev):
    """a real model step PER RANK.  Two processes share the one GPU, collectives over gloo (RCCL
    refuses two ranks on one device): FlatBucketDP broadcasts rank 0's weights, every step averages the gradients of
    the two ranks' different shards (checked against an independent all_gather), and after two FusedAdam steps both
    ranks hold bit-identical parameters.  The gradients live in the bucket: no per-parameter copies around the all-reduce."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, e0, f0, born0, cp0, k0), (_, e1, f1, born1, cp1, k1) = res
    assert e0 <= 1e-7 and e1 <= 1e-7, (e0, e1)                   # the reduced gradient IS the mean of the two ranks'
    assert np.array_equal(f0, f1)                                # identical replicas after two steps
    # r05: every gradient of the planner's step is BORN in the flat bucket (gradient sinks): the exchange copies nothing,
    # and launches at most the scale (+ whatever the collective itself launches): <= 3 kernels
    assert born0 and born1 and cp0 == 0 and cp1 == 0
    assert (k0 is None or k0 <= 3) and (k1 is None or k1 <= 3), (k0, k1)



def test_torch_ops_lsigf_autograd_equals_the_module_path_and_traces(dev):
    """torch.ops.gnnpp.lsigf carries a registered autograd formula (gnnpp::lsigf_backward = the
    kernels of graphML._LSIGFFunction).  Gradients of the taps, the signal and the bias through the DISPATCHER equal the
    module path's bit for bit (same kernels), with and without the fused ReLU, batched and shared GSOs; a training-style
    function on the op compiles with fullgraph=True (AOTAutograd traces forward AND backward through the ops' fake
    implementations) and its gradients match eager."""
    import gnn_pathplanning_amd.ops  # noqa: F401
    from gnn_pathplanning_amd import graphML as gml
    g = torch.Generator().manual_seed(11)
    B, N, K = 9, 10, 3
    h0 = (torch.randn(96, 1, K, 128, generator=g) / 20).to(dev)
    x0 = torch.randn(B, 128, N, generator=g).to(dev)
    b0 = torch.randn(96, 1, generator=g).to(dev)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, 20, seed=4)).float().unsqueeze(1).to(dev)
    cot = torch.randn(B, 96, N, generator=g).to(dev)
    for S_use, fn in ((S, gml.BatchLSIGF), (S[0], gml.LSIGF)):
        for relu in (False, True):
            grads = []
            for path in ('ops', 'module'):
                h, x, b = (t.clone().requires_grad_(True) for t in (h0, x0, b0))
                if path == 'ops':
                    y = torch.ops.gnnpp.lsigf(h, S_use, x, b, relu, 0)
                else:
                    y = fn(h, S_use, x, b)
                    if relu:
                        y = torch.relu(y)
                y.backward(cot)
                grads.append((y.detach(), h.grad, x.grad, b.grad))
            for a, m in zip(grads[0], grads[1]):
                assert torch.equal(a, m), (relu, (a - m).abs().max().item())
            if S_use is S and not relu:
                grads_ref_dh, grads_ref_dx = grads[1][1], grads[1][2]

    # r05: only the gradients somebody asked for are computed, and the packed taps (forward and transposed)
    # are cached on the weight object: two steps on the same parameter pack twice in all (once per form), not per call
    packs = []
    real_pack = gml.pack_filter_taps
    gml.pack_filter_taps = lambda hh: (packs.append(tuple(hh.shape)), real_pack(hh))[1]
    try:
        h = h0.clone().requires_grad_(True)
        x_only = x0.clone().requires_grad_(True)
        for _ in range(2):
            y = torch.ops.gnnpp.lsigf(h.detach(), S, x_only, b0, False, 0)       # taps frozen: only dx is needed
            (dx_only,) = torch.autograd.grad(y, [x_only], cot)
        assert torch.equal(dx_only, grads_ref_dx)
        n_frozen = len(packs)
        for _ in range(3):
            torch.ops.gnnpp.lsigf(h, S, x_only, b0, False, 0).backward(cot)
        assert len(packs) - n_frozen <= 3                     # forward op (uncached by design) x 3 ... and NOT 3 x 3
        assert torch.equal(h.grad / 3, grads_ref_dh) or (h.grad / 3 - grads_ref_dh).abs().max().item() <= 1e-5
    finally:
        gml.pack_filter_taps = real_pack

    def loss_fn(h, x, b):
        y = torch.ops.gnnpp.lsigf(h, S, x, b, True, 0)
        return (y * cot).sum()
    eager = []
    for f in (loss_fn, torch.compile(loss_fn, backend='aot_eager', fullgraph=True)):
        h, x, b = (t.clone().requires_grad_(True) for t in (h0, x0, b0))
        f(h, x, b).backward()
        eager.append((h.grad, x.grad, b.grad))
    for a, m in zip(eager[0], eager[1]):
        assert torch.equal(a, m)
