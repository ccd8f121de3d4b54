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


@pytest.mark.parametrize('N,B,seed,fbn,ext_pack', [(2, 3, 0, 0, 0), (3, 5, 1, 1, 1)])
def test_emu_train_encoder_forward_backward(N, B, seed, fbn, ext_pack):
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
    nbt = [np.full(1, 7, np.int64) for _ in range(5)]              # BatchNorm2d.num_batches_tracked: += N
    # ext_pack: the caller-owned weight pack of gnnpp_train_pack (v330: one launch per weight version, shared with the
    # graph filter's taps) instead of the pack the forward call builds in its workspace
    tp = None
    if ext_pack:
        lib.gnnpp_train_pack_floats.restype = ctypes.c_size_t
        tp = np.full(lib.gnnpp_train_pack_floats(), np.nan, np.float32)
        assert tp.ctypes.data % 16 == 0
        assert lib.gnnpp_train_pack(ctypes.byref(P), el.ptr(tp), None, None, None, 0, 0, 0, 0, None) == 0
    rc = lib.gnnpp_encoder_train_fwd(ctypes.byref(P), el.ptr(obs_np), el.ptr(ws), el.ptr(feat), B, N,
                                     ctypes.c_float(0.1), 1, (ctypes.c_void_p * 5)(*[a.ctypes.data for a in nbt]),
                                     fbn, el.ptr(tp) if tp is not None else None, None)
    assert rc == 0
    assert all(int(a[0]) == 7 + N for a in nbt)
    # r06b: the running-statistics update rides in the forward's last BatchNorm launch (GNNPP_TUNE_TRAIN_RUNNING_FUSED,
    # default); the launch of its own (knob 19 = 0) gives the same bits
    assert lib.gnnpp_get_tuning(19) == 1 and lib.gnnpp_set_tuning(19, 0) == 0
    try:
        arrs2 = {k: el.f32(sd[k].numpy().copy()) for k in arrs if 'running' in k}
        P2 = el.EncParams()
        ctypes.memmove(ctypes.byref(P2), ctypes.byref(P), ctypes.sizeof(P))
        for i in range(5):
            P2.bn_mean[i] = arrs2['ConvLayers.%d.running_mean' % BN[i]].ctypes.data
            P2.bn_var[i] = arrs2['ConvLayers.%d.running_var' % BN[i]].ctypes.data
        ws2, feat2 = np.zeros_like(ws), np.full((N, B, 128), np.nan, np.float32)
        nbt2 = [np.full(1, 7, np.int64) for _ in range(5)]
        rc = lib.gnnpp_encoder_train_fwd(ctypes.byref(P2), el.ptr(obs_np), el.ptr(ws2), el.ptr(feat2), B, N,
                                         ctypes.c_float(0.1), 1, (ctypes.c_void_p * 5)(*[a.ctypes.data for a in nbt2]),
                                         fbn, el.ptr(tp) if tp is not None else None, None)
        assert rc == 0 and all(int(a[0]) == 7 + N for a in nbt2)
        for k, a2 in arrs2.items():
            assert np.array_equal(a2, arrs[k]), k
        # update_running = 0 (BatchNorm2d(track_running_stats=False)): the running-statistic pointers arrive as NULL -- the
        # fused update must stay off (r06c fix: it dereferenced them) and the features are the same
        assert lib.gnnpp_set_tuning(19, 1) == 0
        ws3, feat3 = np.zeros_like(ws), np.full((N, B, 128), np.nan, np.float32)
        rc = lib.gnnpp_encoder_train_fwd(ctypes.byref(P2), el.ptr(obs_np), el.ptr(ws3), el.ptr(feat3), B, N,
                                         ctypes.c_float(0.1), 0, None, fbn, el.ptr(tp) if tp is not None else None, None)
        assert rc == 0 and np.array_equal(feat3, feat)
        for k, a2 in arrs2.items():
            assert np.array_equal(a2, arrs[k]), k            # (untouched by that call)
    finally:
        assert lib.gnnpp_set_tuning(19, 1) == 0
    if fbn:                                                  # feat_sample_major: the same rows as [B,N,128]
        feat = np.ascontiguousarray(feat.reshape(B, N, 128).transpose(1, 0, 2))
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
    cot_np = el.f32(cot.permute(1, 0, 2).numpy() if fbn else cot.numpy())
    rc = lib.gnnpp_encoder_train_bwd(ctypes.byref(P), el.ptr(obs_np), el.ptr(ws), el.ptr(cot_np), ctypes.byref(G),
                                     B, N, fbn, el.ptr(tp) if tp is not None else None, None)
    assert rc == 0
    # r06b: the weight gradients of all five layers come from ONE launch behind the chain (GNNPP_TUNE_TRAIN_WGRAD_MERGED,
    # default); the per-layer launches of r05 (knob 18 = 0) give the same bits
    assert lib.gnnpp_get_tuning(18) == 1 and lib.gnnpp_set_tuning(18, 0) == 0
    try:
        G2, outs2 = Grads(), {}
        for i in range(5):
            for field, key in (('conv_w', 'ConvLayers.%d.weight' % CONV[i]), ('conv_b', 'ConvLayers.%d.bias' % CONV[i]),
                               ('bn_w', 'ConvLayers.%d.weight' % BN[i]), ('bn_b', 'ConvLayers.%d.bias' % BN[i])):
                o = np.full(tuple(sd[key].shape), np.nan, np.float32); outs2[key] = o
                getattr(G2, field)[i] = o.ctypes.data
        rc = lib.gnnpp_encoder_train_bwd(ctypes.byref(P), el.ptr(obs_np), el.ptr(ws), el.ptr(cot_np), ctypes.byref(G2),
                                         B, N, fbn, el.ptr(tp) if tp is not None else None, None)
        assert rc == 0
        for key in outs:
            assert np.array_equal(outs[key], outs2[key]), key
    finally:
        assert lib.gnnpp_set_tuning(18, 1) == 0
    for key, o in outs.items():
        want = p_ref[key].grad.numpy()
        scale = np.abs(want).max()
        # conv biases in front of train-mode BatchNorm: exactly zero in exact arithmetic, roundoff on both sides
        tol = 2e-4 * scale + (2e-4 if ('bias' in key and int(key.split('.')[1]) in CONV) else 1e-6)
        assert np.isfinite(o).all() and np.abs(o - want).max() <= tol, (key, np.abs(o - want).max(), scale)


class AdamTensors(ctypes.Structure):
    _fields_ = [('p', ctypes.c_void_p * 32), ('g', ctypes.c_void_p * 32), ('m', ctypes.c_void_p * 32),
                ('v', ctypes.c_void_p * 32), ('numel', ctypes.c_longlong * 32), ('count', ctypes.c_int)]


@pytest.mark.parametrize('batch,M,N,K', [(3, 128, 128, 70), (1, 5, 128, 33), (1, 1, 40, 57), (2, 70, 20, 300)])
def test_emu_gemm_kmajor(batch, M, N, K):
    """gnnpp_gemm_kmajor (split contraction, ordered partial sums) against numpy, strided operands included:
    the first case is the graph filter's tap gradient written straight into the [F,E,K,G] layout."""
    import emu_lib as el
    lib = el.load()
    lib.gnnpp_gemm_workspace_floats.restype = ctypes.c_size_t
    ll = ctypes.c_longlong
    rng = np.random.default_rng(batch * 1000 + M)
    A = rng.standard_normal((M, K)).astype(np.float32)              # shared by every batch entry (a_sb = 0)
    Bm = rng.standard_normal((batch, K, N)).astype(np.float32)
    C = np.full((M, batch, N), np.nan, np.float32)                  # C_b(m,n) at m*batch*N + b*N + n
    nws = lib.gnnpp_gemm_workspace_floats(batch, M, N, K)
    ws = np.zeros(max(nws, 1), np.float32)
    rc = lib.gnnpp_gemm_kmajor(el.ptr(A), ll(0), ll(K), ll(1), el.ptr(Bm), ll(K * N), ll(N), el.ptr(C), ll(N),
                               ll(batch * N), batch, M, N, K, el.ptr(ws), None)
    assert rc == 0
    want = np.einsum('mk,bkn->mbn', A.astype(np.float64), Bm.astype(np.float64))
    np.testing.assert_allclose(C, want, rtol=0, atol=2e-5 * np.sqrt(K))
    # transposed A (the Linear weight gradient dW = dY^T X: A(m,k) = dY[k][m])
    dY = rng.standard_normal((K, M)).astype(np.float32)
    C2 = np.full((M, N), np.nan, np.float32)
    rc = lib.gnnpp_gemm_kmajor(el.ptr(dY), ll(0), ll(1), ll(M), el.ptr(Bm[0]), ll(0), ll(N), el.ptr(C2), ll(0),
                               ll(N), 1, M, N, K, el.ptr(ws), None)
    assert rc == 0
    np.testing.assert_allclose(C2, dY.astype(np.float64).T @ Bm[0].astype(np.float64), rtol=0,
                               atol=2e-5 * np.sqrt(K))


@pytest.mark.parametrize('B,N', [(3, 2), (64, 10), (130, 9)])
def test_emu_policy_loss(B, N):
    """gnnpp_policy_loss == mean over agents of torch CrossEntropyLoss on argmax labels, and its gradient
    (agents/decentralplannerlocal.py:296-312); ties in the target resolve to the first maximum."""
    import emu_lib as el
    import torch.nn.functional as tF
    lib = el.load()
    g = torch.Generator().manual_seed(B + N)
    logits = (3 * torch.randn(N, B, 5, generator=g)).requires_grad_(True)
    labels = torch.randint(0, 5, (B, N), generator=g)
    target = tF.one_hot(labels, 5).float()
    target[0, 0] = 0.0                                             # an all-equal row: label 0
    labels[0, 0] = 0
    loss = sum(tF.cross_entropy(logits[n], labels[:, n]) for n in range(N)) / N
    loss.backward()
    lg = el.f32(logits.detach().numpy()); tg = el.f32(target.numpy())
    out = np.zeros(1, np.float32); dl = np.full_like(lg, np.nan)
    assert lib.gnnpp_policy_loss(el.ptr(lg), el.ptr(tg), el.ptr(out), el.ptr(dl), B, N, 5, 0, None) == 0
    assert abs(out[0] - loss.item()) <= 2e-6 * max(1.0, abs(loss.item()))
    np.testing.assert_allclose(dl, logits.grad.numpy(), rtol=0, atol=2e-7)
    # the same rows laid out [B,N,5] (what the train-mode forward computes): same loss, transposed gradient
    lg2 = np.ascontiguousarray(lg.transpose(1, 0, 2)); out2 = np.zeros(1, np.float32); dl2 = np.full_like(lg2, np.nan)
    assert lib.gnnpp_policy_loss(el.ptr(lg2), el.ptr(tg), el.ptr(out2), el.ptr(dl2), B, N, 5, 1, None) == 0
    assert out2[0] == out[0] and np.array_equal(dl2.transpose(1, 0, 2), dl)


def test_emu_adam_matches_torch():
    """gnnpp_adam_step against torch.optim.Adam (L2 weight decay, bias correction) over several steps and
    tensors of ragged sizes (one launch covers them all; the step counter lives in `state`)."""
    import emu_lib as el
    lib = el.load()
    g = torch.Generator().manual_seed(3)
    shapes = [(32, 3, 3, 3), (32,), (128, 128), (5,), (1030,)]
    ps = [torch.randn(*s, generator=g).requires_grad_(True) for s in shapes]
    opt = torch.optim.Adam(ps, lr=1e-3, weight_decay=1e-5)
    mine = [el.f32(p.detach().numpy().copy()) for p in ps]
    m = [np.zeros_like(a) for a in mine]; v = [np.zeros_like(a) for a in mine]
    state = np.zeros(8, np.float32)                          # [steps, 2 factors, arrival counter, betas, 2 factors of the next step]
    cf = ctypes.c_float
    for it in range(4):
        grads = [torch.randn(*s, generator=g) for s in shapes]
        for p, gr in zip(ps, grads):
            p.grad = gr.clone()
        opt.step()
        gn = [el.f32(gr.numpy()) for gr in grads]
        tb = AdamTensors()
        for i in range(len(ps)):
            tb.p[i], tb.g[i], tb.m[i], tb.v[i] = mine[i].ctypes.data, gn[i].ctypes.data, m[i].ctypes.data, v[i].ctypes.data
            tb.numel[i] = mine[i].size
        tb.count = len(ps)
        rc = lib.gnnpp_adam_step(ctypes.byref(tb), el.ptr(state), cf(1e-3), cf(0.9), cf(0.999), cf(1e-8), cf(1e-5),
                                 1, None)
        assert rc == 0 and state[0] == it + 1 and state[3] == 0      # (ticked by the launch's last workgroup, re-armed)
        for a, p in zip(mine, ps):
            np.testing.assert_allclose(a, p.detach().numpy(), rtol=0, atol=3e-7)


class GemmDesc(ctypes.Structure):
    _fields_ = [('A', ctypes.c_void_p), ('a_sb', ctypes.c_longlong), ('a_sm', ctypes.c_longlong),
                ('a_sk', ctypes.c_longlong), ('B', ctypes.c_void_p), ('b_sb', ctypes.c_longlong),
                ('b_sk', ctypes.c_longlong), ('C', ctypes.c_void_p), ('c_sb', ctypes.c_longlong),
                ('c_sm', ctypes.c_longlong), ('batch', ctypes.c_int), ('M', ctypes.c_int), ('N', ctypes.c_int),
                ('K', ctypes.c_int), ('mask', ctypes.c_void_p)]


def test_emu_gemm_multi_linear_backward():
    """gnnpp_gemm_kmajor_multi: the three products of a Linear layer's backward pass (dx = dY W, dW = dY^T X,
    db = 1^T dY) in one call, against numpy -- one of them split over the contraction, one not."""
    import emu_lib as el
    lib = el.load()
    lib.gnnpp_gemm_multi_workspace_floats.restype = ctypes.c_size_t
    rng = np.random.default_rng(5)
    R, I, O = 90, 128, 5
    dY = rng.standard_normal((R, O)).astype(np.float32)
    X = rng.standard_normal((R, I)).astype(np.float32)
    Wt = rng.standard_normal((O, I)).astype(np.float32)
    ones = np.ones(R, np.float32)
    dx = np.full((R, I), np.nan, np.float32); dW = np.full((O, I), np.nan, np.float32); db = np.full(O, np.nan, np.float32)
    arr = (GemmDesc * 3)()
    for d, (A, a_st, Bm, b_st, C, c_st, M, N, K) in zip(arr, (
            (dY, (0, O, 1), Wt, (0, I), dx, (0, I), R, I, O),
            (dY, (0, 1, O), X, (0, I), dW, (0, I), O, I, R),
            (ones, (0, 0, 1), dY, (0, O), db, (0, O), 1, O, R))):
        d.A, d.a_sb, d.a_sm, d.a_sk = A.ctypes.data, *a_st
        d.B, d.b_sb, d.b_sk = Bm.ctypes.data, *b_st
        d.C, d.c_sb, d.c_sm = C.ctypes.data, *c_st
        d.batch, d.M, d.N, d.K = 1, M, N, K
    ws = np.zeros(max(lib.gnnpp_gemm_multi_workspace_floats(arr, 3), 1), np.float32)
    assert lib.gnnpp_gemm_kmajor_multi(arr, 3, el.ptr(ws), None) == 0
    np.testing.assert_allclose(dx, dY.astype(np.float64) @ Wt.astype(np.float64), rtol=0, atol=2e-5)
    np.testing.assert_allclose(dW, dY.astype(np.float64).T @ X.astype(np.float64), rtol=0, atol=2e-4)
    np.testing.assert_allclose(db, dY.astype(np.float64).sum(0), rtol=0, atol=2e-4)
    # v330: a ReLU backward folded into a product -- C is stored as 0 where mask <= 0, whether the product is written
    # directly (dx: contraction 5) or through the split reduction (dW with a mask of its own shape: contraction 90)
    mask_x = rng.standard_normal((R, I)).astype(np.float32)
    mask_w = rng.standard_normal((O, I)).astype(np.float32)
    arr[0].mask, arr[1].mask = mask_x.ctypes.data, mask_w.ctypes.data
    dx0, dW0 = dx.copy(), dW.copy()
    dx[:] = np.nan; dW[:] = np.nan
    assert lib.gnnpp_gemm_kmajor_multi(arr, 3, el.ptr(ws), None) == 0
    np.testing.assert_array_equal(dx, np.where(mask_x > 0, dx0, 0))
    np.testing.assert_array_equal(dW, np.where(mask_w > 0, dW0, 0))


@pytest.mark.parametrize('R,I,O,relu', [(90, 128, 128, 1), (37, 128, 5, 0), (16, 64, 40, 1)])
def test_emu_linear_fwd(R, I, O, relu):
    """gnnpp_linear_fwd (v330): y = x W^T + b (+ ReLU) of compressMLP / actionsMLP in the training step against numpy
    (decentralplanner.py:187-195, :232-243); ragged row and feature tiles."""
    import emu_lib as el
    lib = el.load()
    rng = np.random.default_rng(R + O)
    x = rng.standard_normal((R, I)).astype(np.float32)
    W = (rng.standard_normal((O, I)) / np.sqrt(I)).astype(np.float32)
    b = rng.standard_normal(O).astype(np.float32)
    y = np.full((R, O), np.nan, np.float32)
    assert lib.gnnpp_linear_fwd(el.ptr(x), el.ptr(W), el.ptr(b), el.ptr(y), R, I, O, relu, None) == 0
    want = x.astype(np.float64) @ W.astype(np.float64).T + b
    if relu:
        want = np.maximum(want, 0)
    np.testing.assert_allclose(y, want, rtol=0, atol=3e-6)
    assert lib.gnnpp_linear_fwd(el.ptr(x), el.ptr(W), None, el.ptr(y), R, 100, O, relu, None) == -2     # I % 64 != 0


def test_emu_train_pack_filter_taps_and_masked_input_gradient():
    """gnnpp_train_pack (v330) writes the fp32 fragments of the forward AND the transposed taps straight from h: both
    must equal the first region of gnnpp_filter_pack of h / of h.permute(3,1,2,0); gnnpp_lsigf_input_grad on the
    transposed taps = gnnpp_lsigf_fwd_save(s_transposed) of the same, with the mask of the folded ReLU applied."""
    import emu_lib as el
    lib = el.load()
    rng = np.random.default_rng(9)
    F, E, K, G = 40, 2, 3, 24
    h = rng.standard_normal((F, E, K, G)).astype(np.float32)
    hT = np.ascontiguousarray(h.transpose(3, 1, 2, 0))
    want_f, want_t = el.pack_filter(lib, h), el.pack_filter(lib, hT)
    fwd = np.full_like(want_f, np.nan)
    tr = np.full_like(want_t, np.nan)
    assert lib.gnnpp_train_pack(None, None, el.ptr(h), el.ptr(fwd), el.ptr(tr), G, F, K, E, None) == 0
    nf = E * K * ((F + 15) // 16) * ((G + 15) // 16) * 256           # floats of the fp32 fragment region
    np.testing.assert_array_equal(fwd[:nf], want_f[:nf])
    np.testing.assert_array_equal(tr[:nf], want_t[:nf])
    assert np.isnan(fwd[nf:]).all() and np.isnan(tr[nf:]).all()       # nothing else is written
    B, N = 3, 7
    S = (rng.random((B, E, N, N)) < 0.4) * rng.standard_normal((B, E, N, N))
    S = el.f32(S)
    dy = el.f32(rng.standard_normal((B, N, F)))
    mask = el.f32(rng.standard_normal((B, N, G)))
    ref = np.full((B, N, G), np.nan, np.float32)
    rc = lib.gnnpp_lsigf_fwd_save(el.ptr(dy), el.ptr(S), el.ptr(want_t), None, el.ptr(ref), None, B, N, N, F, G, K, E,
                                  0, 1, 1, 1, 1, 0, 0, 0, None, None)
    assert rc == 0
    for m in (None, mask):
        dx = np.full((B, N, G), np.nan, np.float32)
        rc = lib.gnnpp_lsigf_input_grad(el.ptr(dy), el.ptr(S), el.ptr(tr), el.ptr(m) if m is not None else None,
                                        el.ptr(dx), B, N, G, F, K, E, 0, 1, 1, None)
        assert rc == 0
        np.testing.assert_array_equal(dx, ref if m is None else np.where(m > 0, ref, 0))
