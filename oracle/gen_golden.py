"""Generate the golden vectors under tests/golden/ by running the REAL reference on CPU.

Runs only in the build container (needs /root/reference, which never travels to the GPU box).
The reference is imported read-only with the two tiny sys.modules stubs described in SURVEY.md
section 8c; nothing from it is copied: only tensors (inputs, parameters, outputs) are stored.

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz

Outputs
  tests/golden/lsigf_cases.npz    LSIGF / BatchLSIGF / GraphFilter / GraphFilterBatch I/O
  tests/golden/policy_model.npz   DecentralPlannerNet parameters (reference init + randomised BN
                                  running stats) and forward I/O for several (N, K, B)
  tests/golden/policy_multilayer.npz  planners with L = 2 graph-filter layers and / or E = 2
  tests/golden/policy_large.npz   forward I/O of teams of 50 / 64 / 100 agents (parameters of policy_model.npz);
                                  `python oracle/gen_golden.py large` writes only this file
  tests/golden/training_grads.npz train-mode forward/backward, standalone filter gradients
Every random draw is seeded (module constructors included): re-running reproduces the files bit for bit.
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


def import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    stub = types.ModuleType('torchsummaryX')
    stub.summary = lambda *a, **k: None
    sys.modules['torchsummaryX'] = stub
    for name, path in (('utils', REF + '/utils'), ('utils.graphUtils', REF + '/utils/graphUtils')):
        pkg = types.ModuleType(name)
        pkg.__path__ = [path]
        sys.modules[name] = pkg
    from graphs.models.decentralplanner import DecentralPlannerNet
    import utils.graphUtils.graphML as gml
    return DecentralPlannerNet, gml


class Cfg:
    def __init__(self, n, k):
        self.num_agents = n
        self.nGraphFilterTaps = k
        self.device = torch.device('cpu')


def gen_lsigf(gml):
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle.policy_oracle import synth_gso_geometric, synth_gso_sparse
    g = torch.Generator().manual_seed(20260926)
    store, meta = {}, []

    def rnd(*shape):
        return torch.randn(*shape, generator=g)

    def add(kind, h, S, x, b, y, extra=None):
        i = len(meta)
        store['c%d_h' % i] = h.numpy()
        store['c%d_S' % i] = S.numpy()
        store['c%d_x' % i] = x.numpy()
        if b is not None:
            store['c%d_b' % i] = b.numpy()
        store['c%d_y' % i] = y.numpy()
        m = {'kind': kind, 'has_bias': b is not None, 'F': h.shape[0], 'E': h.shape[1],
             'K': h.shape[2], 'G': h.shape[3], 'S_dtype': str(S.dtype).replace('torch.', '')}
        m.update(extra or {})
        meta.append(m)

    # functional forms -----------------------------------------------------------------
    for K in (1, 2, 3, 4):
        for N in (1, 2, 10, 50, 100):
            G, F_out = (128, 128) if (K == 3 and N in (10, 50)) else (16, 24)
            B = 2
            E = 2 if (K == 2 and N == 10) else 1
            h = rnd(F_out, E, K, G) / (G * K) ** 0.5
            b = rnd(F_out, 1) * 0.1 if (K + N) % 2 == 0 else None
            x = torch.relu(rnd(B, G, N))
            # shared GSO (LSIGF): asymmetric sparse
            S1 = synth_gso_sparse(E, N, 3.5, seed=K * 1000 + N)
            with torch.no_grad():
                add('LSIGF', h, S1, x, b, gml.LSIGF(h, S1, x, b))
            # per-sample GSO (BatchLSIGF): geometric fp64 for even K, sparse asymmetric fp32 else
            if K % 2 == 0:
                Sb = torch.from_numpy(synth_gso_geometric(B * E, N, max(4, int(N ** 0.5 * 6)),
                                                          seed=K * 77 + N)).reshape(B, E, N, N)
            else:
                Sb = synth_gso_sparse(B * E, N, 3.5, seed=K * 31 + N).reshape(B, E, N, N)
            with torch.no_grad():
                add('BatchLSIGF', h, Sb, x, b, gml.BatchLSIGF(h, Sb, x, b))

    # per-node bias b[F,N] (graphML.py:2300-2302) and filters wider than one 128-feature launch ------
    for (G, F_out, K, N, E, wide) in ((16, 24, 3, 10, 1, False), (128, 128, 3, 10, 1, False),
                                      (20, 150, 3, 7, 2, True), (128, 256, 2, 10, 1, True)):
        B = 2
        h = rnd(F_out, E, K, G) / (G * K) ** 0.5
        x = torch.relu(rnd(B, G, N))
        b = rnd(F_out, 1) * 0.1 if wide else rnd(F_out, N) * 0.1
        S1 = synth_gso_sparse(E, N, 3.5, seed=G * 3 + F_out)
        Sb = synth_gso_sparse(B * E, N, 3.5, seed=G * 5 + F_out).reshape(B, E, N, N)
        with torch.no_grad():
            add('LSIGF', h, S1, x, b, gml.LSIGF(h, S1, x, b), {'bias_per_node': not wide})
            add('BatchLSIGF', h, Sb, x, b, gml.BatchLSIGF(h, Sb, x, b), {'bias_per_node': not wide})

    # module forms incl. the Nin < N zero-padding path --------------------------------------
    for (G, F_out, K, N, Nin) in ((8, 12, 3, 10, 7), (128, 128, 3, 10, 10), (5, 3, 2, 6, 6)):
        B = 3
        torch.manual_seed(G * 1000 + F_out)            # the modules draw their taps from the global RNG
        gf = gml.GraphFilter(G, F_out, K, 1, True)
        gfb = gml.GraphFilterBatch(G, F_out, K, 1, True)
        x = torch.relu(rnd(B, G, Nin))
        S1 = synth_gso_sparse(1, N, 3.0, seed=G * 13 + N)
        Sb = synth_gso_sparse(B, N, 3.0, seed=G * 17 + N).reshape(B, 1, N, N)
        with torch.no_grad():
            gf.addGSO(S1)
            add('GraphFilter', gf.weight.detach(), S1, x, gf.bias.detach(), gf(x), {'Nin': Nin})
            gfb.addGSO(Sb)
            add('GraphFilterBatch', gfb.weight.detach(), Sb, x, gfb.bias.detach(), gfb(x),
                {'Nin': Nin})
    store['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, 'lsigf_cases.npz'), **store)
    print('lsigf_cases: %d cases' % len(meta))


def gen_policy(DecentralPlannerNet):
    from oracle.policy_oracle import synth_gso_geometric, synth_gso_sparse, synth_obs
    torch.manual_seed(1337)
    base = DecentralPlannerNet(Cfg(10, 3)).eval()
    # default BN running stats make eval-BN nearly the identity: randomise them
    g = torch.Generator().manual_seed(4242)
    with torch.no_grad():
        for m in base.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(0.1 * torch.randn(m.num_features, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.num_features, generator=g))
                m.bias.copy_(0.05 * torch.randn(m.num_features, generator=g))
        base.compressMLP[0].bias.copy_(0.05 * torch.randn(128, generator=g))
        base.actionsMLP[0].bias.copy_(0.05 * torch.randn(5, generator=g))
    sd = {k: v.clone() for k, v in base.state_dict().items()}
    store = {'sd/' + k: v.numpy() for k, v in sd.items()}
    meta = []
    gfl_extra = {}
    for (N, K, B, gso) in ((10, 3, 4, 'geo64'), (10, 2, 2, 'sparse32'), (10, 4, 2, 'geo64'),
                           (1, 3, 2, 'geo64'), (3, 3, 2, 'sparse32'), (23, 3, 1, 'geo64')):
        net = DecentralPlannerNet(Cfg(N, K)).eval()
        sdk = dict(sd)
        if K != 3:
            if K not in gfl_extra:
                gfl_extra[K] = net.state_dict()['GFL.0.weight'].clone()
                store['gfl_w_K%d' % K] = gfl_extra[K].numpy()
            sdk['GFL.0.weight'] = gfl_extra[K]
        net.load_state_dict(sdk)
        obs = synth_obs(B, N, seed=N * 100 + K)
        if gso == 'geo64':
            S = torch.from_numpy(synth_gso_geometric(B, N, 20, seed=N * 7 + K))
        else:
            S = synth_gso_sparse(B, N, 3.0, seed=N * 5 + K)
        feats = {}
        hook = net.GFL.register_forward_pre_hook(lambda mod, inp: feats.__setitem__('x', inp[0].clone()))
        with torch.no_grad():
            net.addGSO(S)
            out = net(obs)
        hook.remove()
        i = len(meta)
        store['p%d_obs' % i] = obs.numpy()
        store['p%d_S' % i] = S.numpy()
        store['p%d_feat' % i] = feats['x'].numpy()
        store['p%d_logits' % i] = torch.stack(out, dim=1).numpy()      # [B,N,5]
        meta.append({'N': N, 'K': K, 'B': B, 'gso': gso})
    store['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, 'policy_model.npz'), **store)
    print('policy_model: %d cases' % len(meta))


def gen_policy_large(DecentralPlannerNet):
    """Large teams (the shapes of BASELINE configs 3 and 5): the reference's forward with the parameters stored in
    policy_model.npz -> policy_large.npz (inputs, logits; the features are not kept: the encoder is per agent and
    covered by policy_model.npz)."""
    from oracle.policy_oracle import synth_gso_geometric, synth_gso_sparse, synth_obs
    z = np.load(os.path.join(OUT, 'policy_model.npz'))
    sd = {k[3:]: torch.from_numpy(np.array(z[k])) for k in z.files if k.startswith('sd/')}
    store, meta = {}, []
    for (N, K, B, gso, W) in ((50, 3, 2, 'geo64', 50), (100, 2, 1, 'sparse32', 0), (100, 4, 2, 'geo64', 100),
                              (64, 3, 3, 'geo32', 40), (100, 3, 1, 'geo64', 100)):
        net = DecentralPlannerNet(Cfg(N, K)).eval()
        sdk = dict(sd)
        if K != 3:
            sdk['GFL.0.weight'] = torch.from_numpy(np.array(z['gfl_w_K%d' % K]))
        net.load_state_dict(sdk)
        obs = synth_obs(B, N, seed=N * 100 + K)
        if gso == 'sparse32':
            S = synth_gso_sparse(B, N, 6.0, seed=N * 5 + K)
        else:
            S = torch.from_numpy(synth_gso_geometric(B, N, W, seed=N * 7 + K))
            if gso == 'geo32':
                S = S.float()
        with torch.no_grad():
            net.addGSO(S)
            out = net(obs)
        i = len(meta)
        store['q%d_obs' % i] = obs.numpy()
        store['q%d_S' % i] = S.numpy()
        store['q%d_logits' % i] = torch.stack(out, dim=1).numpy()      # [B,N,5]
        meta.append({'N': N, 'K': K, 'B': B, 'gso': gso})
    store['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, 'policy_large.npz'), **store)
    print('policy_large: %d cases' % len(meta))


def gen_multilayer(DecentralPlannerNet, gml):
    """Planners with SEVERAL graph-filter layers and E > 1 edge features -> policy_multilayer.npz.
    The reference builds / runs L layers and E features generically (decentralplanner.py:205-224,
    266-276, 293-315) but fixes L = 1, E = 1 in its source (:130-131, :208); here the reference
    object is re-wired after construction exactly as editing those lines would (its own
    GraphFilterBatch modules, its own addGSO / forward)."""
    from oracle.policy_oracle import synth_gso_geometric, synth_gso_sparse, synth_obs
    z = np.load(os.path.join(OUT, 'policy_model.npz'))
    sd_enc = {k[3:]: torch.from_numpy(np.array(z[k])) for k in z.files if k.startswith('sd/')}
    store, meta = {}, []
    for ci, (N, B, dims, taps, E, gso) in enumerate(((10, 3, [64, 128], [3, 2], 1, 'geo64'),
                                                     (10, 2, [128], [3], 2, 'sparse32'),
                                                     (5, 2, [32, 48], [2, 3], 2, 'sparse32'),
                                                     (50, 2, [128, 128], [3, 3], 1, 'geo64'))):
        torch.manual_seed(8000 + ci)
        net = DecentralPlannerNet(Cfg(N, 3))
        net.load_state_dict(sd_enc)                    # encoder (and the soon replaced GFL / head)
        F = [128] + dims
        layers = []
        for l in range(len(dims)):
            layers += [gml.GraphFilterBatch(F[l], F[l + 1], taps[l], E, True), torch.nn.ReLU(inplace=True)]
        net.GFL = torch.nn.Sequential(*layers)
        net.L, net.F, net.K, net.E = len(dims), F, taps, E
        head = torch.nn.Linear(F[-1], 5)
        torch.nn.init.xavier_normal_(head.weight)
        with torch.no_grad():
            head.bias.copy_(0.05 * torch.randn(5))
        net.actionsMLP = torch.nn.Sequential(head)
        net.eval()
        obs = synth_obs(B, N, seed=300 + ci)
        if gso == 'geo64':
            S = torch.from_numpy(synth_gso_geometric(B * E, N, 20, seed=310 + ci)).reshape(B, E, N, N)
        else:
            S = synth_gso_sparse(B * E, N, 3.0, seed=310 + ci).reshape(B, E, N, N)
        with torch.no_grad():
            net.addGSO(S.squeeze(1) if E == 1 else S)
            out = net(obs)
        k = 'm%d_' % ci
        for l in range(len(dims)):
            store[k + 'GFL.%d.weight' % (2 * l)] = net.GFL[2 * l].weight.detach().numpy()
            store[k + 'GFL.%d.bias' % (2 * l)] = net.GFL[2 * l].bias.detach().numpy()
        store[k + 'actionsMLP.0.weight'] = head.weight.detach().numpy()
        store[k + 'actionsMLP.0.bias'] = head.bias.detach().numpy()
        store[k + 'obs'] = obs.numpy().astype(np.uint8)
        store[k + 'S'] = S.numpy()
        store[k + 'logits'] = torch.stack(out, dim=1).numpy()                  # [B,N,5]
        meta.append({'N': N, 'B': B, 'dims': dims, 'taps': taps, 'E': E, 'gso': gso})
    store['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, 'policy_multilayer.npz'), **store)
    print('policy_multilayer: %d cases' % len(meta))


def gen_training(DecentralPlannerNet, gml):
    """Train-mode forward + backward of the reference (agents/decentralplannerlocal.py:283-317) and
    standalone graph-filter gradients -> tests/golden/training_grads.npz."""
    from oracle.policy_oracle import synth_gso_geometric, synth_gso_sparse, synth_obs
    z = np.load(os.path.join(OUT, 'policy_model.npz'))
    sd = {k[3:]: torch.from_numpy(np.array(z[k])) for k in z.files if k.startswith('sd/')}
    store, meta = {}, []
    FULL = ('GFL.0.weight', 'GFL.0.bias', 'actionsMLP.0.weight', 'actionsMLP.0.bias',
            'compressMLP.0.bias', 'ConvLayers.0.weight', 'ConvLayers.0.bias', 'ConvLayers.1.weight',
            'ConvLayers.15.weight', 'ConvLayers.15.bias', 'ConvLayers.11.bias')
    for ci, (N, K, B) in enumerate(((10, 3, 4), (3, 2, 6))):
        net = DecentralPlannerNet(Cfg(N, K))
        sdk = dict(sd)
        if K != 3:
            sdk['GFL.0.weight'] = torch.from_numpy(np.array(z['gfl_w_K%d' % K]))
        net.load_state_dict(sdk)
        net.train()
        obs = synth_obs(B, N, seed=900 + ci)
        S = torch.from_numpy(synth_gso_geometric(B, N, 20, seed=800 + ci)).float()
        g = torch.Generator().manual_seed(700 + ci)
        tgt = torch.nn.functional.one_hot(torch.randint(0, 5, (B, N), generator=g), 5).float()
        net.addGSO(S)
        predict = net(obs)
        bt = tgt.permute(1, 0, 2)
        loss = 0
        ce = torch.nn.CrossEntropyLoss()
        for n in range(N):
            loss = loss + ce(predict[n], torch.max(bt[n], 1)[1])
        loss = loss / N
        loss.backward()
        store['g%d_obs' % ci] = obs.numpy().astype(np.uint8)
        store['g%d_S' % ci] = S.numpy()
        store['g%d_target' % ci] = tgt.numpy().astype(np.uint8)
        store['g%d_logits' % ci] = torch.stack([p.detach() for p in predict], 1).numpy()
        store['g%d_loss' % ci] = np.array(loss.item(), dtype=np.float64)
        names, summ = [], []
        for name, p in net.named_parameters():
            names.append(name)
            gr = p.grad.double()
            summ.append([gr.sum().item(), gr.abs().sum().item(), gr.norm().item()])
            if name in FULL:
                store['g%d_grad/%s' % (ci, name)] = p.grad.numpy()
        store['g%d_gradsum' % ci] = np.array(summ)
        for name, b in net.named_buffers():
            if name.startswith(('ConvLayers.1.', 'ConvLayers.15.')):
                store['g%d_buf/%s' % (ci, name)] = b.numpy()
        meta.append({'kind': 'policy', 'N': N, 'K': K, 'B': B, 'param_names': names})
    # standalone graph-filter gradients
    g = torch.Generator().manual_seed(4711)
    for ci, (cls, G, F_out, K, E, N, Nin, B) in enumerate((('GraphFilterBatch', 16, 24, 3, 2, 7, 5, 3),
                                                          ('GraphFilter', 128, 128, 3, 1, 10, 10, 2),
                                                          ('GraphFilterBatch', 128, 128, 4, 1, 10, 10, 2))):
        torch.manual_seed(6000 + ci)                   # module taps come from the global RNG
        mod = getattr(gml, cls)(G, F_out, K, E, True)
        x = torch.randn(B, G, Nin, generator=g, requires_grad=True)
        if cls == 'GraphFilter':
            S = synth_gso_sparse(E, N, 3.0, seed=50 + ci)
        else:
            S = synth_gso_sparse(B * E, N, 3.0, seed=50 + ci).reshape(B, E, N, N)
        cot = torch.randn(B, F_out, Nin, generator=g)
        mod.addGSO(S)
        y = mod(x)
        (y * cot).sum().backward()
        k = 'f%d_' % ci
        store[k + 'h'] = mod.weight.detach().numpy()
        store[k + 'b'] = mod.bias.detach().numpy()
        store[k + 'S'] = S.numpy()
        store[k + 'x'] = x.detach().numpy()
        store[k + 'cot'] = cot.numpy()
        store[k + 'y'] = y.detach().numpy()
        store[k + 'dx'] = x.grad.numpy()
        store[k + 'dh'] = mod.weight.grad.numpy()
        store[k + 'db'] = mod.bias.grad.numpy()
        meta.append({'kind': cls, 'G': G, 'F': F_out, 'K': K, 'E': E, 'N': N, 'Nin': Nin, 'B': B})
    # pre-powered GSO family: matrixPowersBatch / batchLSIGF / GraphFilterBatchGSO
    for ci, (G, F_out, K, E, N, B) in enumerate(((12, 20, 3, 1, 9, 3), (128, 128, 3, 1, 10, 2),
                                                 (8, 8, 4, 2, 6, 2))):
        torch.manual_seed(7000 + ci)
        mod = gml.GraphFilterBatchGSO(G, F_out, K, E, True)
        S = synth_gso_sparse(B * E, N, 3.0, seed=70 + ci).reshape(B, E, N, N)
        if E == 1 and ci == 0:
            S = S.squeeze(1)                                       # the 3-D addGSO form
        x = torch.randn(B, G, N, generator=g)
        with torch.no_grad():
            mod.addGSO(S)
            y = mod(x)
        k = 'p%d_' % ci
        store[k + 'h'] = mod.weight.detach().numpy()
        store[k + 'b'] = mod.bias.detach().numpy()
        store[k + 'S'] = S.numpy()
        store[k + 'SK'] = mod.SK.numpy()
        store[k + 'x'] = x.numpy()
        store[k + 'y'] = y.numpy()
        meta.append({'kind': 'GraphFilterBatchGSO', 'G': G, 'F': F_out, 'K': K, 'E': E, 'N': N, 'B': B})
    store['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, 'training_grads.npz'), **store)
    print('training_grads: %d cases' % len(meta))


def gen_multilayer_training(DecentralPlannerNet, gml):
    """Train-mode forward + loss.backward() of planners with SEVERAL graph-filter layers / E > 1 edge features
    (the re-wired reference of gen_multilayer, agents/decentralplannerlocal.py:283-317's loss) ->
    tests/golden/training_multilayer.npz: logits, loss, sum / abs-sum / norm of EVERY gradient and the full
    gradients of the graph-filter layers and the head."""
    from oracle.policy_oracle import synth_gso_geometric, synth_gso_sparse, synth_obs
    z = np.load(os.path.join(OUT, 'policy_model.npz'))
    sd_enc = {k[3:]: torch.from_numpy(np.array(z[k])) for k in z.files if k.startswith('sd/')}
    store, meta = {}, []
    for ci, (N, B, dims, taps, E, gso) in enumerate(((10, 4, [64, 128], [3, 2], 1, 'geo64'),
                                                     (5, 3, [128], [3], 2, 'sparse32'),
                                                     (6, 3, [32, 48], [2, 3], 2, 'sparse32'))):
        torch.manual_seed(8100 + ci)
        net = DecentralPlannerNet(Cfg(N, 3))
        net.load_state_dict(sd_enc)
        F = [128] + dims
        layers = []
        for l in range(len(dims)):
            layers += [gml.GraphFilterBatch(F[l], F[l + 1], taps[l], E, True), torch.nn.ReLU(inplace=True)]
        net.GFL = torch.nn.Sequential(*layers)
        net.L, net.F, net.K, net.E = len(dims), F, taps, E
        head = torch.nn.Linear(F[-1], 5)
        torch.nn.init.xavier_normal_(head.weight)
        with torch.no_grad():
            head.bias.copy_(0.05 * torch.randn(5))
        net.actionsMLP = torch.nn.Sequential(head)
        net.train()
        obs = synth_obs(B, N, seed=400 + ci)
        if gso == 'geo64':
            S = torch.from_numpy(synth_gso_geometric(B * E, N, 20, seed=410 + ci)).float().reshape(B, E, N, N)
        else:
            S = synth_gso_sparse(B * E, N, 3.0, seed=410 + ci).reshape(B, E, N, N)
        g = torch.Generator().manual_seed(420 + ci)
        tgt = torch.nn.functional.one_hot(torch.randint(0, 5, (B, N), generator=g), 5).float()
        net.addGSO(S.squeeze(1) if E == 1 else S)
        predict = net(obs)
        bt = tgt.permute(1, 0, 2)
        ce = torch.nn.CrossEntropyLoss()
        loss = 0
        for n in range(N):
            loss = loss + ce(predict[n], torch.max(bt[n], 1)[1])
        loss = loss / N
        loss.backward()
        k = 't%d_' % ci
        for l in range(len(dims)):
            store[k + 'GFL.%d.weight' % (2 * l)] = net.GFL[2 * l].weight.detach().numpy()
            store[k + 'GFL.%d.bias' % (2 * l)] = net.GFL[2 * l].bias.detach().numpy()
        store[k + 'actionsMLP.0.weight'] = head.weight.detach().numpy()
        store[k + 'actionsMLP.0.bias'] = head.bias.detach().numpy()
        store[k + 'obs'] = obs.numpy().astype(np.uint8)
        store[k + 'S'] = S.numpy()
        store[k + 'target'] = tgt.numpy().astype(np.uint8)
        store[k + 'logits'] = torch.stack([p.detach() for p in predict], 1).numpy()
        store[k + 'loss'] = np.array(loss.item(), dtype=np.float64)
        names, summ = [], []
        for name, p in net.named_parameters():
            names.append(name)
            gr = p.grad.double()
            summ.append([gr.sum().item(), gr.abs().sum().item(), gr.norm().item()])
            if name.startswith(('GFL.', 'actionsMLP.')) or name == 'compressMLP.0.weight':
                store[k + 'grad/' + name] = p.grad.numpy()
        store[k + 'gradsum'] = np.array(summ)
        meta.append({'N': N, 'B': B, 'dims': dims, 'taps': taps, 'E': E, 'gso': gso, 'param_names': names})
    store['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, 'training_multilayer.npz'), **store)
    print('training_multilayer: %d cases' % len(meta))


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    Net, gml = import_reference()
    sys.path.insert(0, os.path.dirname(HERE))
    if len(sys.argv) > 1 and sys.argv[1] == 'training':
        gen_training(Net, gml)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'multilayer_training':
        gen_multilayer_training(Net, gml)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'large':
        gen_policy_large(Net)
        sys.exit(0)
    gen_lsigf(gml)
    gen_policy(Net)
    gen_policy_large(Net)
    gen_multilayer(Net, gml)
    gen_training(Net, gml)
    gen_multilayer_training(Net, gml)
