"""CPU: the train-mode encoder kernels (csrc/train_encoder.hip: forward AND backward), compiled unmodified
for the host emulation, against torch autograd over the oracle's restatement of the reference's train
mode (per-agent ConvLayers calls, BatchNorm with that call's batch statistics; oracle/policy_oracle.py
encoder_one_agent(training=True), itself pinned to tests/golden/training_grads.npz)."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu'))

pytestmark = pytest.mark.skipif(not os.path.exists('/opt/rocm/lib/llvm/bin/clang++'),
                                reason='host clang++ from ROCm not present')

CONV = (0, 4, 7, 11, 14)
BN = (1, 5, 8, 12, 15)


class Grads(ctypes.Structure):
    _fields_ = [('conv_w', ctypes.c_void_p * 5), ('conv_b', ctypes.c_void_p * 5),
                ('bn_w', ctypes.c_void_p * 5), ('bn_b', ctypes.c_void_p * 5)]


def reference(sd, obs, cot):
    """feat [N,B,128] (flattened ConvLayers output per agent call) and every gradient, torch autograd."""
    from oracle import policy_oracle as orc
    import torch.nn.functional as tF
    p = {k: v.clone().requires_grad_(v.dtype.is_floating_point and 'running' not in k) for k, v in sd.items()}
    run = {k: v.clone() for k, v in sd.items() if 'running' in k}
    B, N = obs.shape[0], obs.shape[1]
    feats = []
    for n in range(N):
        t = obs[:, n]
        for li in range(5):
            t = tF.conv2d(t, p['ConvLayers.%d.weight' % CONV[li]], p['ConvLayers.%d.bias' % CONV[li]], padding=1)
            bn = 'ConvLayers.%d.' % BN[li]
            t = tF.batch_norm(t, run[bn + 'running_mean'], run[bn + 'running_var'], p[bn + 'weight'],
                              p[bn + 'bias'], training=True, momentum=0.1, eps=orc.BN_EPS)
            t = tF.relu(t)
            if orc.POOL_AFTER[li]:
                t = tF.max_pool2d(t, 2)
        feats.append(t.reshape(B, 128))
    feat = torch.stack(feats, 0)                           # [N,B,128]
    (feat * cot).sum().backward()
    return feat.detach(), p, run


@pytest.mark.parametrize('N,B,seed', [(2, 3, 0), (3, 5, 1)])
def test_emu_train_encoder_forward_backward(N, B, seed):
    import emu_lib as el
    from oracle import policy_oracle as orc
    lib = el.load()
    lib.gnnpp_encoder_train_workspace_floats.restype = ctypes.c_size_t
    sd = orc.init_state_dict(3, seed=40 + seed)
    g = torch.Generator().manual_seed(seed)
    obs = (torch.rand(B, N, 3, 11, 11, generator=g) < 0.25).float() + 0.1 * torch.randn(B, N, 3, 11, 11, generator=g)
    cot = torch.randn(N, B, 128, generator=g)
    want_feat, p_ref, run_ref = reference(sd, obs, cot)

    keep = []
    P = el.EncParams()
    arrs = {}
    for i in range(5):
        for field, key in (('conv_w', 'ConvLayers.%d.weight' % CONV[i]), ('conv_b', 'ConvLayers.%d.bias' % CONV[i]),
                           ('bn_w', 'ConvLayers.%d.weight' % BN[i]), ('bn_b', 'ConvLayers.%d.bias' % BN[i]),
                           ('bn_mean', 'ConvLayers.%d.running_mean' % BN[i]),
                           ('bn_var', 'ConvLayers.%d.running_var' % BN[i])):
            a = el.f32(sd[key].numpy().copy()); keep.append(a); arrs[key] = a
            getattr(P, field)[i] = a.ctypes.data
    a = el.f32(sd['compressMLP.0.weight'].numpy()); keep.append(a); P.fc_w = a.ctypes.data
    a = el.f32(sd['compressMLP.0.bias'].numpy()); keep.append(a); P.fc_b = a.ctypes.data
    P.bn_eps = 1e-5
    ws = np.zeros(lib.gnnpp_encoder_train_workspace_floats(N, B), np.float32)
    feat = np.full((N, B, 128), np.nan, np.float32)
    obs_np = el.f32(obs.numpy())
    rc = lib.gnnpp_encoder_train_fwd(ctypes.byref(P), el.ptr(obs_np), el.ptr(ws), el.ptr(feat), B, N,
                                     ctypes.c_float(0.1), 1, None)
    assert rc == 0
    assert np.abs(feat - want_feat.numpy()).max() <= 2e-5 * max(1.0, want_feat.abs().max().item())
    for i in range(5):                                       # N sequential running-statistics updates
        for nm in ('running_mean', 'running_var'):
            key = 'ConvLayers.%d.%s' % (BN[i], nm)
            assert np.abs(arrs[key] - run_ref[key].numpy()).max() <= 1e-5, key

    G = Grads()
    outs = {}
    for i in range(5):
        for field, key in (('conv_w', 'ConvLayers.%d.weight' % CONV[i]), ('conv_b', 'ConvLayers.%d.bias' % CONV[i]),
                           ('bn_w', 'ConvLayers.%d.weight' % BN[i]), ('bn_b', 'ConvLayers.%d.bias' % BN[i])):
            o = np.full(tuple(sd[key].shape), np.nan, np.float32); outs[key] = o
            getattr(G, field)[i] = o.ctypes.data
    cot_np = el.f32(cot.numpy())
    rc = lib.gnnpp_encoder_train_bwd(ctypes.byref(P), el.ptr(obs_np), el.ptr(ws), el.ptr(cot_np), ctypes.byref(G),
                                     B, N, None)
    assert rc == 0
    for key, o in outs.items():
        want = p_ref[key].grad.numpy()
        scale = np.abs(want).max()
        # conv biases in front of train-mode BatchNorm: exactly zero in exact arithmetic, roundoff on both sides
        tol = 2e-4 * scale + (2e-4 if ('bias' in key and int(key.split('.')[1]) in CONV) else 1e-6)
        assert np.isfinite(o).all() and np.abs(o - want).max() <= tol, (key, np.abs(o - want).max(), scale)
