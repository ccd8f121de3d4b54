"""torch.ops.gnnpp.* -- the hot-path operators registered with torch.library.

    import gnn_pathplanning_amd.ops          # registers the ops
    y      = torch.ops.gnnpp.lsigf(h, S, x, bias, relu=False, precision=0)        # BatchLSIGF / LSIGF
    logits = torch.ops.gnnpp.policy_logits(obs, S, enc_packed, taps_packed, gf_bias, act_w, act_b, K, precision=0)
    ids    = torch.ops.gnnpp.decode_actions(logits)

The module classes (DecentralPlannerNet, GraphFilter*) call the C ABI through ctypes directly; these registrations
exist so that torch.compile / torch.export / FX see the same kernels as OPAQUE operators with known output shapes
(fake implementations below) instead of graph-breaking on ctypes calls.  They are thin: every op is one C entry point
of include/gnnpp.h on the current stream and fails loudly (GnnppError) without the HIP library or on CPU tensors.
`lsigf` is DIFFERENTIABLE through the dispatcher (r04): its registered autograd formula calls `gnnpp::lsigf_backward`
-- the kernels of graphML._LSIGFFunction: input gradient = the transposed filter, tap gradient on gnnpp_gemm_kmajor --
so a training step written on torch.ops.gnnpp.lsigf traces under torch.compile (AOTAutograd sees both ops through their
fake implementations).  `policy_logits` / `decode_actions` are inference ops (training the whole planner goes through
the modules: decentralplanner._EncoderTrainFunction).  `precision` is GNNPP_PREC_* (0 = fp32-equivalent default).  References: utils/graphUtils/graphML.py:48-141, 2273-2367; graphs/models/decentralplanner.py:266-318;
utils/multirobotsim_dcenlocal.py:589-591.
"""
from typing import Optional, Tuple

import torch

from . import _native
from . import graphML as gml



def _ptr(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


@torch.library.custom_op('gnnpp::lsigf', mutates_args=())
def lsigf(h: torch.Tensor, S: torch.Tensor, x: torch.Tensor, bias: Optional[torch.Tensor] = None,
          relu: bool = False, precision: int = 0) -> torch.Tensor:
    """h [F,E,K,G], S [E,N,N] (shared) or [B,E,N,N], x [B,G,N] -> y [B,F,N]   (graphML.py:2273-2367)."""
    batched = S.dim() == 4
    return gml._lsigf_device(h, S, x, bias, batched, x.shape[2], packed=gml._cached_pack(h, False), relu=relu,
                             precision=int(precision))


@lsigf.register_fake
def _(h, S, x, bias=None, relu=False, precision=0):
    return x.new_empty((x.shape[0], h.shape[0], x.shape[2]), dtype=torch.float32)


@torch.library.custom_op('gnnpp::lsigf_backward', mutates_args=())
def lsigf_backward(h: torch.Tensor, S: torch.Tensor, x: torch.Tensor, bias: Optional[torch.Tensor], dy: torch.Tensor,
                   relu: bool = False, precision: int = 0, needs: int = 7
                   ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(dh, dx, dbias) of y = lsigf(h, S, x, bias, relu) for the cotangent dy [B,F,N] (agents/decentralplannerlocal.py:314
    `loss.backward()` through graphML.py:2273-2367).  The forward is re-run with its tap signals kept (one more filter
    launch: the op pair carries no hidden state between forward and backward), then graphML._LSIGFFunction's backward:
    dx = the same kernel on dy with transposed taps and S^T, dh = one split-K GEMM, dbias = a reduction.  No gradient
    for S.  `needs`: bit 0 = dh, bit 1 = dx, bit 2 = dbias -- a gradient nobody asked for is neither computed nor
    launched and comes back as an EMPTY tensor (so does dbias when there is no bias).  The packed taps (forward and
    transposed) are cached on the weight OBJECT the caller passed (graphML._cached_pack): a training loop that passes
    its parameter packs once per weight version, not once per call."""
    batched = S.dim() == 4
    want_h, want_x, want_b = bool(needs & 1), bool(needs & 2), bool(needs & 4) and bias is not None
    if not (want_h or want_x or want_b):
        return h.new_empty(0), h.new_empty(0), h.new_empty(0)
    packed = gml._cached_pack(h, False)
    packed_T = gml._cached_pack(h, True) if want_x else None
    with torch.enable_grad(), _native.no_grad_sinks():        # (h.detach() keeps the parameter's address: no bucket views out)
        hh, xx = h.detach().requires_grad_(want_h), x.detach().requires_grad_(want_x)
        bb = bias.detach().requires_grad_(want_b) if bias is not None else None
        y = gml._LSIGFFunction.apply(hh, S.detach(), xx, bb, batched, packed, False, bool(relu), int(precision),
                                     packed_T)
        wrt = ([hh] if want_h else []) + ([xx] if want_x else []) + ([bb] if want_b else [])
        grads = list(torch.autograd.grad(y, wrt, dy.detach().contiguous()))
    dh = grads.pop(0) if want_h else h.new_empty(0)
    dx = grads.pop(0) if want_x else h.new_empty(0)
    db = grads.pop(0) if want_b else h.new_empty(0)
    return dh, dx, db


@lsigf_backward.register_fake
def _(h, S, x, bias, dy, relu=False, precision=0, needs=7):
    e = h.new_empty(0)
    return (torch.empty_like(h, dtype=torch.float32) if needs & 1 else e,
            torch.empty_like(x, dtype=torch.float32) if needs & 2 else e,
            torch.empty_like(bias, dtype=torch.float32) if (needs & 4 and bias is not None) else e)


def _lsigf_setup_context(ctx, inputs, output):
    h, S, x, bias, relu, precision = inputs
    ctx.save_for_backward(h, S, x, bias)
    ctx.relu, ctx.precision, ctx.has_bias = bool(relu), int(precision), bias is not None


def _lsigf_autograd(ctx, dy):
    h, S, x, bias = ctx.saved_tensors
    ng = ctx.needs_input_grad                                # (h, S, x, bias, relu, precision)
    needs = int(ng[0]) | int(ng[2]) << 1 | int(bool(ng[3]) and ctx.has_bias) << 2
    dh, dx, db = torch.ops.gnnpp.lsigf_backward(h, S, x, bias, dy, ctx.relu, ctx.precision, needs)
    return (dh if ng[0] else None), None, (dx if ng[2] else None), (db if (ctx.has_bias and ng[3]) else None), None, None


lsigf.register_autograd(_lsigf_autograd, setup_context=_lsigf_setup_context)


@torch.library.custom_op('gnnpp::policy_logits', mutates_args=())
def policy_logits(obs: torch.Tensor, S: torch.Tensor, enc_packed: torch.Tensor, taps_packed: torch.Tensor,
                  gf_bias: Optional[torch.Tensor], act_w: torch.Tensor, act_b: torch.Tensor, K: int,
                  precision: int = 0) -> torch.Tensor:
    """The whole policy step of the single-layer planner (addGSO + forward, decentralplanner.py:266-318):
    obs [B,N,3,11,11], S [B,1,N,N] fp32 | fp64, the packed encoder (DecentralPlannerNet.packed_encoder()), the packed
    taps of GFL[0] (GraphFilterBatch.packed_taps()), its bias [128] or None, actionsMLP weight [5,128] / bias [5]
    -> logits [N,B,5].  Teams of <= 112 agents; split-f16 (precision 2) is unguarded here (no range flag)."""
    B, N = obs.shape[0], obs.shape[1]
    dev = _native.require_gpu(obs, S, enc_packed, taps_packed, act_w, act_b)
    # The packs are opaque buffers the kernels index by layout constants: a pack of another build / tap count, or a
    # GSO of another shape, would be read out of bounds.  (The module API is protected by its PackCache; this op
    # takes the caller's word, so it checks.)
    L = _native.lib()
    K = int(K)
    if tuple(obs.shape[2:]) != (3, 11, 11) or K < 1:
        raise _native.GnnppError('policy_logits: obs must be [B,N,3,11,11] and K >= 1')
    if S.dim() == 3:
        S = S.unsqueeze(1)
    if tuple(S.shape) != (B, 1, N, N):
        raise _native.GnnppError('policy_logits: S must be [B,1,N,N] = %r, got %r' % ((B, 1, N, N), tuple(S.shape)))
    for name, t, want in (('enc_packed', enc_packed, L.gnnpp_encoder_packed_floats()),
                          ('taps_packed', taps_packed, L.gnnpp_filter_packed_floats(128, 128, K, 1))):
        if t.dtype is not torch.float32 or not t.is_contiguous() or t.numel() != want:
            raise _native.GnnppError('policy_logits: %s must be a contiguous fp32 buffer of %d floats (this build, '
                                     'K = %d), got %d x %s' % (name, want, K, t.numel(), t.dtype))
    if tuple(act_w.shape) != (5, 128) or act_b.numel() != 5 or (gf_bias is not None and gf_bias.numel() != 128):
        raise _native.GnnppError('policy_logits: act_w [5,128], act_b [5], gf_bias [128] expected')
    obs_c = obs.detach().contiguous().float()
    S_c = S.detach().contiguous()
    if S_c.dtype not in (torch.float32, torch.float64):
        S_c = S_c.float()
    aw, ab = act_w.detach().contiguous().float(), act_b.detach().contiguous().float()
    gb = gf_bias.detach().reshape(-1).contiguous().float() if gf_bias is not None else None
    ws = torch.empty(B * N, 128, dtype=torch.float32, device=dev)
    logits = torch.empty(N, B, 5, dtype=torch.float32, device=dev)
    with _native.device_guard(dev):
        rc = _native.lib().gnnpp_policy_fwd(_ptr(obs_c), _ptr(S_c), _ptr(enc_packed), _ptr(taps_packed), _ptr(gb),
                                            _ptr(aw), _ptr(ab), _ptr(ws), _ptr(logits), B, N, int(K), 1,
                                            int(S_c.dtype is torch.float64), int(precision), None,
                                            _native.stream_ptr(dev))
    _native.check(rc, 'gnnpp_policy_fwd')
    return logits


@policy_logits.register_fake
def _(obs, S, enc_packed, taps_packed, gf_bias, act_w, act_b, K, precision=0):
    return obs.new_empty((obs.shape[1], obs.shape[0], 5), dtype=torch.float32)


@torch.library.custom_op('gnnpp::decode_actions', mutates_args=())
def decode_actions(logits: torch.Tensor) -> torch.Tensor:
    """logits [N,B,5] -> int32 [B,N]: argmax of the LogSoftmax the simulator applies, first maximum wins
    (utils/multirobotsim_dcenlocal.py:589-591)."""
    N, B = logits.shape[0], logits.shape[1]
    dev = _native.require_gpu(logits)
    lg = logits.detach().contiguous().float()
    acts = torch.empty(B, N, dtype=torch.int32, device=dev)
    with _native.device_guard(dev):
        rc = _native.lib().gnnpp_decode_actions(_ptr(lg), _ptr(acts), B, N, _native.stream_ptr(dev))
    _native.check(rc, 'gnnpp_decode_actions')
    return acts


@decode_actions.register_fake
def _(logits):
    return logits.new_empty((logits.shape[1], logits.shape[0]), dtype=torch.int32)
