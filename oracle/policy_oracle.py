"""CPU oracle for the GNN policy forward pass.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (torch CPU ops, fp32) of the reference's hot path.  It is the
*checker*, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under ``gnn_pathplanning_amd/``
imports it, and the product path raises if the HIP library is missing.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the real reference from
``/root/reference`` in the build container and stores its inputs/outputs under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function below against those
vectors (max |diff| <= 5e-6 on O(1) values; observed 0 .. 1.1e-6).

The op *sequence* deliberately follows the reference (per-agent encoder loop, repeat/cat based
shift register, permute+reshape before the tap contraction) so that timing this oracle on the
host cores is a faithful "reference CPU path" baseline.

Reference (paths relative to the upstream repo root):
  utils/graphUtils/graphML.py:48-141      LSIGF        -> lsigf()
  utils/graphUtils/graphML.py:2273-2367   BatchLSIGF   -> batch_lsigf()
  utils/graphUtils/graphML.py:1200-1219   GraphFilter.forward       -> graph_filter()
  utils/graphUtils/graphML.py:2458-2477   GraphFilterBatch.forward  -> graph_filter_batch()
  graphs/models/decentralplanner.py:278-318  DecentralPlannerNet.forward -> policy_forward()
  utils/multirobotsim_dcenlocal.py:589-591   action decode (LogSoftmax, argmax) -> decode_actions()
"""
import math

import numpy as np
import torch
import torch.nn.functional as tF

# Architecture constants of "DCP v1.4" (graphs/models/decentralplanner.py:89-98, 155-177).
CONV_CHANNELS = (3, 32, 32, 64, 64, 128)
CONV_KEYS = (0, 4, 7, 11, 14)        # indices of the Conv2d modules inside ConvLayers
BN_KEYS = (1, 5, 8, 12, 15)          # indices of the BatchNorm2d modules inside ConvLayers
POOL_AFTER = (True, False, True, False, True)   # MaxPool2d(2) after conv layers 0, 2, 4 (:169-172)
BN_EPS = 1e-5
NUM_ACTIONS = 5
FOV = 11


# --------------------------------------------------------------------------------------------
# graph filter
# --------------------------------------------------------------------------------------------
def _taps_then_contract(h, z_taps, b):
    """Shared tail of both LSIGF flavours (graphML.py:130-141 / :2355-2367).

    z_taps: [B, E, K, G, N];  h: [F, E, K, G];  returns [B, F, N].
    """
    F_out, E, K, G = h.shape
    B, N = z_taps.shape[0], z_taps.shape[4]
    rows = z_taps.permute(0, 4, 1, 2, 3).reshape(B, N, E * K * G)
    y = torch.matmul(rows, h.reshape(F_out, E * K * G).permute(1, 0)).permute(0, 2, 1)
    if b is not None:
        y = y + b
    return y


def lsigf(h, S, x, b=None):
    """One GSO shared by the whole batch.  graphML.py:48-141.

    h [F,E,K,G], S [E,N,N], x [B,G,N], b [F,1] or None -> [B,F,N].  Right-multiplication
    x <- x @ S (:124), no dtype cast of S.
    """
    F_out, E, K, G = h.shape
    assert S.shape[0] == E
    N = S.shape[1]
    assert S.shape[2] == N
    B = x.shape[0]
    assert x.shape[1] == G and x.shape[2] == N
    cur = x.reshape(B, 1, G, N)
    Sb = S.reshape(1, E, N, N)
    z = cur.reshape(B, 1, 1, G, N).repeat(1, E, 1, 1, 1)
    for _ in range(1, K):
        cur = torch.matmul(cur, Sb)
        z = torch.cat((z, cur.reshape(B, E, 1, G, N)), dim=2)
    return _taps_then_contract(h, z, b)


def batch_lsigf(h, S, x, b=None):
    """One GSO per sample.  graphML.py:2273-2367.  S [B,E,N,N] is cast with .float() at every
    shift (:2350), so float64 GSOs from the simulator are accepted."""
    F_out, E, K, G = h.shape
    assert S.shape[1] == E
    N = S.shape[2]
    assert S.shape[3] == N
    B = x.shape[0]
    assert x.shape[1] == G and x.shape[2] == N
    cur = x.reshape(B, 1, G, N)
    Sb = S.reshape(B, E, N, N)
    z = cur.reshape(B, 1, 1, G, N).repeat(1, E, 1, 1, 1)
    for _ in range(1, K):
        cur = torch.matmul(cur, Sb.float())
        z = torch.cat((z, cur.reshape(B, E, 1, G, N)), dim=2)
    return _taps_then_contract(h, z, b)


def _pad_nodes(x, N):
    B, G, Nin = x.shape
    if Nin < N:
        x = torch.cat((x, torch.zeros(B, G, N - Nin, dtype=x.dtype, device=x.device)), dim=2)
    return x, Nin


def graph_filter(weight, bias, S, x):
    """GraphFilter.forward, graphML.py:1200-1219 (zero-pad nodes to N, filter, slice back)."""
    assert S.dim() == 3
    xp, Nin = _pad_nodes(x, S.shape[1])
    u = lsigf(weight, S, xp, bias)
    return u[:, :, :Nin] if Nin < S.shape[1] else u


def graph_filter_batch(weight, bias, S, x):
    """GraphFilterBatch.forward, graphML.py:2458-2477."""
    assert S.dim() == 4
    xp, Nin = _pad_nodes(x, S.shape[2])
    u = batch_lsigf(weight, S, xp, bias)
    return u[:, :, :Nin] if Nin < S.shape[2] else u


def lsigf_f64(h, S, x, b=None):
    """Independent float64 einsum statement of the same algebra (no shared code with the two
    functions above); used by the property tests.  S is [E,N,N] or [B,E,N,N]."""
    h = np.asarray(h, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    S = np.asarray(S, dtype=np.float64)
    F_out, E, K, G = h.shape
    B = x.shape[0]
    if S.ndim == 3:
        S = np.broadcast_to(S[None], (B,) + S.shape)
    y = np.zeros((B, F_out, x.shape[2]))
    for e in range(E):
        z = x
        for k in range(K):
            if k > 0:
                z = np.einsum('bgm,bmn->bgn', z, S[:, e])
            y += np.einsum('fg,bgn->bfn', h[:, e, k, :], z)
    if b is not None:
        y = y + np.asarray(b, dtype=np.float64)
    return y


# --------------------------------------------------------------------------------------------
# full policy
# --------------------------------------------------------------------------------------------
def encoder_one_agent(sd, obs_agent, training=False):
    """ConvLayers + flatten + compressMLP for one agent's mini-batch [B,3,11,11] -> [B,128].
    decentralplanner.py:286-289 with the Sequential of :155-177.  training=True reproduces the
    reference's train mode: BatchNorm uses THIS call's batch statistics and updates the running
    statistics in `sd` in place (momentum 0.1), once per agent call."""
    t = obs_agent
    for li in range(5):
        t = tF.conv2d(t, sd['ConvLayers.%d.weight' % CONV_KEYS[li]],
                      sd['ConvLayers.%d.bias' % CONV_KEYS[li]], stride=1, padding=1)
        bn = 'ConvLayers.%d.' % BN_KEYS[li]
        t = tF.batch_norm(t, sd[bn + 'running_mean'], sd[bn + 'running_var'],
                          sd[bn + 'weight'], sd[bn + 'bias'], training=training, momentum=0.1,
                          eps=BN_EPS)
        t = tF.relu(t)
        if POOL_AFTER[li]:
            t = tF.max_pool2d(t, kernel_size=2)
    flat = t.reshape(t.shape[0], -1)
    return tF.relu(tF.linear(flat, sd['compressMLP.0.weight'], sd['compressMLP.0.bias']))


def policy_forward(sd, S, obs, training=False):
    """DecentralPlannerNet.addGSO + forward (decentralplanner.py:266-318); eval mode by default.

    sd: state_dict (CPU tensors), S: [B,N,N] (E = 1, :271-272) or [B,E,N,N] (:274-276), fp32 or fp64,
    obs: [B,N,3,11,11] fp32.  The graph-filter layers are the `GFL.{2l}` entries of sd -- one in the
    reference's own configuration, L of them when :130-131 list several (:293-301 runs them all).
    Returns a list of N tensors [B,5] exactly like the reference.
    """
    assert S.dim() in (3, 4)
    B, N = obs.shape[0], obs.shape[1]
    S4 = S.unsqueeze(1) if S.dim() == 3 else S
    feat = torch.zeros(B, 128, N)
    for n in range(N):
        feat[:, :, n] = encoder_one_agent(sd, obs[:, n], training)
    shared, l = feat, 0
    while 'GFL.%d.weight' % (2 * l) in sd:
        shared = tF.relu(graph_filter_batch(sd['GFL.%d.weight' % (2 * l)], sd.get('GFL.%d.bias' % (2 * l)),
                                            S4, shared))
        l += 1
    out = []
    for n in range(N):
        out.append(tF.linear(shared[:, :, n].reshape(B, -1),
                             sd['actionsMLP.0.weight'], sd['actionsMLP.0.bias']))
    return out


def policy_loss(out, target):
    """agents/decentralplannerlocal.py:305-312: mean over agents of CE(predict[n], argmax target)."""
    loss = 0
    for n in range(len(out)):
        loss = loss + tF.cross_entropy(out[n], target[:, n].argmax(-1))
    return loss / len(out)


def policy_features(sd, obs):
    """Encoder output in the [B,128,N] layout the reference hands to the graph filter."""
    B, N = obs.shape[0], obs.shape[1]
    feat = torch.zeros(B, 128, N)
    for n in range(N):
        feat[:, :, n] = encoder_one_agent(sd, obs[:, n])
    return feat


def decode_actions(action_list):
    """multirobotsim_dcenlocal.py:589-591: LogSoftmax(dim=-1) then torch.max(.,1)[1].
    Returns int64 [B,N]."""
    ids = [torch.max(torch.log_softmax(a, dim=-1), 1)[1] for a in action_list]
    return torch.stack(ids, dim=1)


def top2_margin(action_list):
    """Gap between the best and second-best logit, [B,N]; near-ties are excluded from the
    bit-exact argmax gate (SURVEY.md section 8d parity gate)."""
    logits = torch.stack(action_list, dim=1)
    top = torch.topk(logits, 2, dim=-1).values
    return top[..., 0] - top[..., 1]


# --------------------------------------------------------------------------------------------
# parameter construction (reference init rules) and synthetic inputs
# --------------------------------------------------------------------------------------------
def init_state_dict(K, seed=1337, randomize_bn_stats=True):
    """Random parameters following graphs/weights_initializer.py:11-23 and
    graphML.py:2442-2447; BN running stats optionally randomised so eval-BN is non-trivial
    (BASELINE.md section 4)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def xavier(shape, fan_in, fan_out):
        std = math.sqrt(2.0 / (fan_in + fan_out))
        return torch.randn(shape, generator=g) * std

    for li in range(5):
        cin, cout = CONV_CHANNELS[li], CONV_CHANNELS[li + 1]
        sd['ConvLayers.%d.weight' % CONV_KEYS[li]] = xavier((cout, cin, 3, 3), cin * 9, cout * 9)
        bound = 1.0 / math.sqrt(cin * 9)    # torch Conv2d default bias init (left untouched)
        sd['ConvLayers.%d.bias' % CONV_KEYS[li]] = (torch.rand(cout, generator=g) * 2 - 1) * bound
        bn = 'ConvLayers.%d.' % BN_KEYS[li]
        sd[bn + 'weight'] = 1.0 + 0.02 * torch.randn(cout, generator=g)
        sd[bn + 'bias'] = torch.zeros(cout)
        if randomize_bn_stats:
            sd[bn + 'running_mean'] = 0.1 * torch.randn(cout, generator=g)
            sd[bn + 'running_var'] = 0.5 + torch.rand(cout, generator=g)
        else:
            sd[bn + 'running_mean'] = torch.zeros(cout)
            sd[bn + 'running_var'] = torch.ones(cout)
        sd[bn + 'num_batches_tracked'] = torch.tensor(0, dtype=torch.int64)
    sd['compressMLP.0.weight'] = xavier((128, 128), 128, 128)
    sd['compressMLP.0.bias'] = torch.zeros(128)
    stdv = 1.0 / math.sqrt(128 * K)
    sd['GFL.0.weight'] = (torch.rand(128, 1, K, 128, generator=g) * 2 - 1) * stdv
    sd['GFL.0.bias'] = (torch.rand(128, 1, generator=g) * 2 - 1) * stdv
    sd['actionsMLP.0.weight'] = xavier((NUM_ACTIONS, 128), 128, NUM_ACTIONS)
    sd['actionsMLP.0.bias'] = torch.zeros(NUM_ACTIONS)
    return sd


def synth_obs(B, N, seed=1337, p=0.1):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(B, N, 3, FOV, FOV, generator=g) < p).float()


def synth_features(B, G, N, seed=1337):
    g = torch.Generator().manual_seed(seed)
    return torch.relu(torch.randn(B, G, N, generator=g))


def _connected(adj):
    n = adj.shape[0]
    seen = np.zeros(n, dtype=bool)
    stack = [0]
    seen[0] = True
    while stack:
        i = stack.pop()
        for j in np.nonzero(adj[i])[0]:
            if not seen[j]:
                seen[j] = True
                stack.append(j)
    return bool(seen.all())


def synth_gso_geometric(B, N, W, seed=1337, r0=6.0, dtype=np.float64):
    """Communication GSO by the simulator's rule (multirobotsim_dcenlocal.py:320-365): N distinct
    cells on a WxW grid, A = (dist < R) with zero diagonal, R grown x1.1 from r0 until connected,
    S = D^-1/2 A D^-1/2 (zero-degree rows -> 0).  Connectivity is decided by graph search instead
    of the Laplacian spectrum (same boolean).  Returns numpy [B,N,N]."""
    rng = np.random.default_rng(seed)
    out = np.zeros((B, N, N), dtype=np.float64)
    for b in range(B):
        cells = rng.choice(W * W, size=N, replace=False)
        pos = np.stack([cells // W, cells % W], axis=1).astype(np.float64)
        d = np.sqrt(((pos[:, None, :] - pos[None, :, :]) ** 2).sum(-1))
        R = r0 / 1.1
        while True:
            R *= 1.1
            A = (d < R).astype(np.float64)
            np.fill_diagonal(A, 0.0)
            if N == 1 or _connected(A):
                break
        deg = A.sum(1)
        zero = np.abs(deg) < 1e-9
        deg[zero] = 1.0
        inv = np.sqrt(1.0 / deg)
        inv[zero] = 0.0
        out[b] = (inv[:, None] * A) * inv[None, :]
    return out.astype(dtype)


def synth_gso_sparse(B, N, mean_degree, seed=1337):
    """Random sparse *asymmetric* GSO: Bernoulli(d/N) mask x U(0,1); exercises the
    column-gather direction."""
    g = torch.Generator().manual_seed(seed)
    p = min(1.0, float(mean_degree) / max(N, 1))
    mask = (torch.rand(B, N, N, generator=g) < p).float()
    return mask * torch.rand(B, N, N, generator=g)
