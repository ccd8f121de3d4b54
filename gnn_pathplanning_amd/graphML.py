"""Graph-filter operators with the reference's API, computed by libgnnpp.so on MI355X.

Mirrors the part of utils/graphUtils/graphML.py that the path planner uses:

    LSIGF(h, S, x, b=None)            graphML.py:48-141      one GSO shared by the batch
    BatchLSIGF(h, S, x, b=None)       graphML.py:2273-2367   one GSO per sample
    GraphFilter(G, F, K, E=1, bias)   graphML.py:1111-1230
    GraphFilterBatch(G, F, K, E, b)   graphML.py:2369-2488
    matrixPowersBatch(S, K)           graphML.py:2063-2113   S^k, k = 0..K-1, per sample
    batchLSIGF(h, SK, x, bias=None)   graphML.py:2115-2178   filter on pre-powered GSOs
    GraphFilterBatchGSO(G, F, K, E,b) graphML.py:2180-2271

Same names, argument meaning, shape asserts, parameter names/shapes (`weight [F,E,K,G]`,
`bias [F,1]`), init rule and `addGSO` / `forward` / `extra_repr` behaviour, so the reference's
callers run unchanged.  Layouts at this boundary are the reference's feature-major
x[B,G,N] -> y[B,F,N]; the kernel transposes through LDS on load/store (no extra HBM pass).

Autograd: when gradients are enabled and an input requires grad, the call goes through
`_LSIGFFunction`: forward = the HIP kernel (also dumping the tap signals z_k), backward =
  dx = the SAME kernel on dy with taps h.permute(3,1,2,0) and S^T (the input gradient of a graph
       filter is a graph filter), dW = one library GEMM per tap (dy . z_k^T), db = sum(dy).
No gradient flows to S (it is data in the reference too).  There is no CPU path: tensors must be
on a HIP device.
"""
import ctypes
import math

import torch
import torch.nn as nn

from . import _native

zeroTolerance = 1e-9    # kept for API parity (graphML.py:42-43)
infiniteNumber = 1e12

# Arithmetic of the tap contraction (include/gnnpp.h GNNPP_PREC_*): 'fp32' (default, fp32-equivalent: bf16x3 operand
# planes on the small-graph kernels -- N <= 16, G = F = 128, >= 64 workgroups: lsigf_small_dispatch -- and the exact
# fp32 MFMA everywhere else), 'fp32_mfma' (exact fp32 MFMA on every path, bitwise an fmaf chain and independent of
# the batch size; GNNPP_TUNE_FILTER_SMALL=0 gives the same single schedule under 'fp32'), or the opt-in 'split_f16'
# (22-bit operands, |x| < 65504, unguarded at this level).
# There is NO process-wide switch: the functions take `precision=` per call (None = DEFAULT_PRECISION), the modules
# carry `self.precision`, fixed at CONSTRUCTION from the `precision=` argument (None = DEFAULT_PRECISION at that
# moment).  Nothing reads a module global at call time, so two threads / streams cannot change each other's
# arithmetic (tests/test_gpu_parity.py::test_filter_precision_is_per_call_and_per_instance).
DEFAULT_PRECISION = 'fp32'

_MAX_F_PER_LAUNCH = 128
MAX_NODES = 112         # rows one workgroup holds in LDS (GNNPP_MAX_NODES=100 guaranteed at G=F=128)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def pack_filter_taps(h):
    """h [F,E,K,G] (device, fp32) -> MFMA-fragment-ordered buffer (gnnpp_filter_pack)."""
    _native.require_gpu(h)
    L = _native.lib()
    F_out, E, K, G = h.shape
    hc = h.detach().contiguous().float()
    packed = torch.empty(L.gnnpp_filter_packed_floats(G, F_out, K, E), dtype=torch.float32,
                         device=h.device)
    with _native.device_guard(h.device):
        _native.check(L.gnnpp_filter_pack(_ptr(hc), _ptr(packed), G, F_out, K, E,
                                          _native.stream_ptr(h.device)), 'gnnpp_filter_pack')
    return packed


def _bias_arg(b, F_out, N):
    """The reference's bias is F x 1 (one value per feature) or F x N (per feature and node,
    graphML.py:2300-2302); both are applied in the kernel's epilogue."""
    if b is None:
        return None, 0
    if b.numel() == F_out:
        return b.detach().contiguous().float().reshape(-1), 0
    assert b.shape[0] == F_out and b.shape[-1] == N, 'bias must be [F,1] or [F,N]'
    return b.detach().contiguous().float(), 1


def _large_tap_signals(h, S, x, batched):
    """Z [B,N,E*K,G]: every tap signal z_{e,k} = S_e^T z_{e,k-1} of a node-major x [B,N,G], one gnnpp_gemm_kmajor call per
    (e, k >= 1) over the batch; and the fp32 GSO it used."""
    dev = _native.require_gpu(h, S, x)
    F_out, E, K, G = h.shape
    B, N, _ = x.shape
    S32 = S.detach()
    S32 = (S32 if S32.dtype is torch.float32 else S32.float()).contiguous()       # S.float() of the reference (:2350)
    EKG = E * K * G
    Z = torch.empty(B, N, E * K, G, dtype=torch.float32, device=dev)
    xc = x.detach().float()
    for e in range(E):
        Z[:, :, e * K] = xc
        for k in range(1, K):
            Se = S32[:, e] if batched else S32[e]                                   # [B,N,N] view | [N,N] view
            _native.gemm_kmajor(Se, (E * N * N if batched else 0, 1, N),           # A(m = n_out, k = m_in) = S[m_in][n_out]
                                Z[:, :, e * K + k - 1], (N * EKG, EKG),             # B(k = m_in, n = g)
                                Z[:, :, e * K + k], (N * EKG, EKG), B, N, G, N)
    return Z, S32


def _lsigf_large(h, S, x, b, batched, relu=False, keep=False):
    """The same filter for graphs of MORE than MAX_NODES nodes (whose rows do not fit one workgroup's LDS), node-major:
    x [B,N,G] -> [B,N,F].  Dense and exact fp32 throughout on gnnpp_gemm_kmajor (fp32 MFMA, ordered partial sums):
      z_{e,k} = S_e^T z_{e,k-1}  (rows = nodes: z_k[n][g] = sum_m S[m][n] z_{k-1}[m][g], graphML.py:2345-2352),
      one GEMM per (e, k >= 1) over the batch, written into its column block of Z [B*N, E*K*G];
      y = Z . h^T  (one GEMM, contraction E*K*G), then bias and ReLU.
    The reference has no size limit (BatchLSIGF is a chain of torch.matmul); this keeps the drop-in true for any N
    at dense-GEMM speed -- the reference's configurations (N <= 100) never take this path.
    keep: also return (Z, S32) for the backward pass (_LSIGFFunction, large graphs)."""
    dev = _native.require_gpu(h, S, x, b)
    F_out, E, K, G = h.shape
    B, N, _ = x.shape
    EKG = E * K * G
    Z, S32 = _large_tap_signals(h, S, x, batched)
    hT = h.detach().float().permute(1, 2, 3, 0).reshape(EKG, F_out).contiguous()   # [(e,k,g), f]
    y = torch.empty(B, N, F_out, dtype=torch.float32, device=dev)
    _native.gemm_kmajor(Z, (0, EKG, 1), hT, (0, F_out), y, (0, F_out), 1, B * N, F_out, EKG)
    if b is not None:
        bb = b.detach().float()
        y += bb.reshape(1, 1, F_out) if bb.numel() == F_out else bb.t().reshape(1, N, F_out)
    if relu:
        torch.relu_(y)
    return (y, Z, S32) if keep else y


def _lsigf_large_backward(h, S32, Z, dy, batched, need_dh, need_dx):
    """Gradients of the dense large-graph filter for dy [B,N,F] (already masked by the ReLU), on gnnpp_gemm_kmajor:
      dh[f,(e,k,g)] = sum_(b,n) dy[(b,n),f] Z[(b,n),(e,k,g)]                      one GEMM, contraction B*N;
      dZ[(b,n),(e,k,g)] = sum_f dy[(b,n),f] h[f,(e,k,g)]                          one GEMM;
      dz_{e,k-1} += S_e dz_{e,k}  (the adjoint of z_k = S_e^T z_{k-1}), k = K-1 .. 1   one GEMM per (e, k) over the batch;
      dx = sum_e dz_{e,0}."""
    F_out, E, K, G = h.shape
    B, N, _ = dy.shape
    EKG = E * K * G
    dev = dy.device
    dh = dx = None
    if need_dh:
        dh = torch.empty(F_out, E, K, G, dtype=torch.float32, device=dev)
        _native.gemm_kmajor(dy, (0, 1, F_out), Z, (0, EKG), dh, (0, EKG), 1, F_out, EKG, B * N)
    if need_dx:
        dZ = torch.empty(B, N, E * K, G, dtype=torch.float32, device=dev)
        h2 = h.detach().float().reshape(F_out, EKG).contiguous()
        _native.gemm_kmajor(dy, (0, F_out, 1), h2, (0, EKG), dZ, (0, EKG), 1, B * N, EKG, F_out)
        tmp = torch.empty(B, N, G, dtype=torch.float32, device=dev)
        for e in range(E):
            Se = S32[:, e] if batched else S32[e]
            for k in range(K - 1, 0, -1):
                _native.gemm_kmajor(Se, (E * N * N if batched else 0, N, 1),        # A(m = m_in, k = n_out) = S[m_in][n_out]
                                    dZ[:, :, e * K + k], (N * EKG, EKG), tmp, (N * G, G), B, N, G, N)
                dZ[:, :, e * K + k - 1] += tmp
        dx = dZ[:, :, 0].clone() if E == 1 else dZ[:, :, ::K].sum(dim=2)
    return dh, dx


def _lsigf_device(h, S, x, b, batched, Nin, packed=None, relu=False, transposed=False,
                  save_taps=False, node_major=False, precision=None, out_mask=None):
    """Shared driver: h [F,E,K,G], S [E,N,N] | [B,E,N,N], x [B,G,Nin] -> y [B,F,Nin]
    (and, with save_taps, zs [E*K, B*N, G]).  Any F: the C entry point splits wide filters.
    node_major: x [B,N,G] -> y [B,N,F] (rows = nodes, the layout the kernel keeps in LDS anyway).
    precision: GNNPP_PREC_* (or its name) of the tap contraction; None = DEFAULT_PRECISION (fp32-equivalent: bf16x3
    planes on the small-graph kernels, the exact fp32 MFMA elsewhere); 2 = the opt-in split-f16 schedule, unguarded
    here (no range flag is passed)."""
    dev = _native.require_gpu(h, S, x, b)
    L = _native.lib()
    precision = _native.precision_code(DEFAULT_PRECISION if precision is None else precision)
    F_out, E, K, G = h.shape
    N = S.shape[-1]
    B = x.shape[0]
    if N > MAX_NODES:
        if transposed or save_taps:
            raise _native.GnnppError('graphs with N=%d > %d nodes: the LDS-resident kernels do not apply '
                                     '(training goes through _LSIGFFunction\'s dense path)' % (N, MAX_NODES))
        if node_major:
            return _lsigf_large(h, S, x, b, batched, relu)
        xn = torch.zeros(B, N, G, dtype=torch.float32, device=dev)                 # zero padding of missing nodes
        xn[:, :Nin] = x.detach().permute(0, 2, 1)
        return _lsigf_large(h, S, xn, b, batched, relu)[:, :Nin].permute(0, 2, 1).contiguous()
    xc = x.detach().contiguous()
    if xc.dtype != torch.float32:
        xc = xc.float()
    Sc = S.detach().contiguous()
    if Sc.dtype not in (torch.float32, torch.float64):
        Sc = Sc.float()
    if packed is None:
        packed = pack_filter_taps(h)
    bias, per_node = _bias_arg(b, F_out, N)
    nm = int(node_major)
    y = torch.empty((B, Nin, F_out) if node_major else (B, F_out, Nin), dtype=torch.float32, device=dev)
    zs = torch.empty(E * K, B * N, G, dtype=torch.float32, device=dev) if save_taps else None
    s64 = int(Sc.dtype == torch.float64)
    with _native.device_guard(dev):
        if out_mask is not None:
            # input gradient with the ReLU backward of the layer below folded in (gnnpp_lsigf_input_grad): h is the
            # transposed taps' SHAPE [G,E,K,F] here, x the output gradient dy [B,N,F], y = dx [B,N,G]
            assert transposed and node_major and b is None and not relu and Nin == N
            rc = L.gnnpp_lsigf_input_grad(_ptr(xc), _ptr(Sc), _ptr(packed), _ptr(out_mask), _ptr(y), B, N, F_out, G,
                                          K, E, s64, int(batched), 1, _native.stream_ptr(dev))
        elif transposed or save_taps:
            rc = L.gnnpp_lsigf_fwd_save(_ptr(xc), _ptr(Sc), _ptr(packed), _ptr(bias), _ptr(y), _ptr(zs),
                                        B, N, Nin, G, F_out, K, E, s64, int(batched), int(transposed),
                                        nm, nm, int(relu), per_node, int(precision), None,
                                        _native.stream_ptr(dev))
        else:
            rc = L.gnnpp_lsigf_fwd(_ptr(xc), _ptr(Sc), _ptr(packed), _ptr(bias), _ptr(y),
                                   B, N, Nin, G, F_out, K, E, s64, int(batched), nm, nm, int(relu),
                                   per_node, int(precision), None, _native.stream_ptr(dev))
    if rc == -2 and L.gnnpp_lsigf_fits(N, G, F_out, K, E) != 0:
        _native.check(rc, 'gnnpp_lsigf_fwd')     # GNNPP_ERR_UNSUPPORTED for another reason than LDS room: an error
    if rc == -2 and not (transposed or save_taps):
        # GNNPP_ERR_UNSUPPORTED below MAX_NODES, confirmed by gnnpp_lsigf_fits: the graph's rows do not fit the kernel's
        # LDS budget (wide input features, or 101..112 nodes at G = F = 128).  The reference has no such limit: the
        # dense exact-fp32 form.
        if node_major:
            return _lsigf_large(h, S, x, b, batched, relu)
        xn = torch.zeros(B, N, G, dtype=torch.float32, device=dev)
        xn[:, :Nin] = x.detach().permute(0, 2, 1)
        return _lsigf_large(h, S, xn, b, batched, relu)[:, :Nin].permute(0, 2, 1).contiguous()
    if rc == -2 and (transposed or save_taps):
        return None                                # the training driver (_LSIGFFunction) takes its dense path instead
    _native.check(rc, 'gnnpp_lsigf_fwd')
    return (y, zs) if save_taps else y


_weight_packs = {}            # (id(weight), transposed) -> (key, weakref(weight), packed buffer)


def _cached_pack(h, transposed):
    """Packed taps of `h` (forward filter) or of h.permute(3,1,2,0) (the input-gradient filter), cached PER WEIGHT
    OBJECT and weight version: a training step needs each once per filter layer, however many calls share the weight
    (a planner with several graph-filter layers keeps one entry per layer).  The entry holds a weak reference to the
    weight: a dead object's id / address can be reused by an unrelated tensor."""
    key = (h._version, h.data_ptr(), _native._pack_generation)
    slot = (id(h), bool(transposed))
    ent = _weight_packs.get(slot)
    if ent is None or ent[0] != key or ent[1]() is not h:
        if len(_weight_packs) > 64:
            _weight_packs.clear()
        import weakref
        src = h.detach().permute(3, 1, 2, 0).contiguous() if transposed else h
        ent = (key, weakref.ref(h), pack_filter_taps(src))
        _weight_packs[slot] = ent
    return ent[2]


def _packed_transposed_taps(h):
    return _cached_pack(h, True)


class _LSIGFFunction(torch.autograd.Function):
    """y = LSIGF(h, S, x, b) with gradients for h, x and b (none for S)."""

    @staticmethod
    def forward(ctx, h, S, x, b, batched, packed, node_major=False, relu=False, precision=None, packed_T=None,
                fold=0):
        """node_major: x [B,N,G] -> y [B,N,F] (train-mode planner: no transposing copies around the filter);
        relu: y = relu(filter) in the same launch (the mask for the backward pass is y > 0);
        precision: of the forward contraction (the gradient filters run in the default arithmetic);
        packed_T: the packed taps of the input-gradient filter when the caller already holds them (ops.py: `h` is a
        per-call alias of the parameter there, which the per-object cache cannot recognise);
        fold (node-major training path only; the caller vouches for both): bit 0 -- the incoming output gradient is
        ALREADY masked by this filter's ReLU (the product that computed it did so: _LinearFunction `mask_dx`), bit 1 --
        x is the output of a ReLU whose backward is folded into this filter's input-gradient launch (dx is returned
        masked by x > 0; the layer below must not mask again), bit 2 -- the tap / bias gradient products wait in
        _native's queue for a later launch of the same backward pass (_native.defer_gemms; only while h / b have no
        `.grad` yet)."""
        Nin = x.shape[1] if node_major else x.shape[2]
        ctx.packed_T = packed_T
        ctx.fold = int(fold) if node_major else 0
        ctx.x_mask = x.detach() if (ctx.fold & 2) else None          # (read by the dense form's backward only)
        ctx.bias_ref = b
        ctx.batched, ctx.Nin, ctx.has_bias = batched, Nin, b is not None
        ctx.bias_shape = None if b is None else tuple(b.shape)
        ctx.param_ptrs = (h.data_ptr(), b.data_ptr() if b is not None else 0)    # (_native.grad_out: gradient sinks)
        ctx.node_major, ctx.relu = node_major, relu
        N = S.shape[-1]
        ctx.large = N > MAX_NODES
        res = None if ctx.large else _lsigf_device(h, S, x, b, batched, Nin, packed, relu=relu, save_taps=True,
                                                   node_major=node_major, precision=precision)
        if res is not None:
            y, zs = res
            ctx.save_for_backward(h, S, zs, y if relu else None)
            return y
        # graphs whose rows do not fit one workgroup's LDS (N > MAX_NODES, or fewer nodes with wide input features:
        # GNNPP_ERR_UNSUPPORTED): the dense exact-fp32 form, forward and backward (the reference has no size limit:
        # graphML.py:2273-2367); node-major inside, zero rows for nodes the signal does not have
        ctx.large = True
        xn = x.detach().float() if node_major else x.detach().float().permute(0, 2, 1)
        if Nin != N:
            xn = torch.cat([xn, xn.new_zeros(xn.shape[0], N - Nin, xn.shape[2])], 1)
        y, Z, S32 = _lsigf_large(h, S, xn.contiguous(), b, batched, relu, keep=True)
        ctx.save_for_backward(h, S32, Z, y if relu else None)
        y = y[:, :Nin]
        return y.contiguous() if node_major else y.permute(0, 2, 1).contiguous()

    @staticmethod
    def backward(ctx, dy):
        h, S, zs, yrelu = ctx.saved_tensors
        F_out, E, K, G = h.shape
        N = S.shape[-1]
        B = dy.shape[0]
        dy = dy.contiguous().float()
        if ctx.large:
            dyn = dy if ctx.node_major else dy.permute(0, 2, 1)                     # [B,Nin,F]
            if ctx.Nin != N:
                dyn = torch.cat([dyn, dyn.new_zeros(B, N - ctx.Nin, F_out)], 1)
            dyn = dyn.contiguous()
            if ctx.relu and not (ctx.fold & 1):
                dyn = torch.ops.aten.threshold_backward(dyn, yrelu, 0)
            dh, dxn = _lsigf_large_backward(h, S, zs, dyn, ctx.batched, ctx.needs_input_grad[0], ctx.needs_input_grad[2])
            dx = db = None
            if dxn is not None:
                dxn = dxn[:, :ctx.Nin]
                dx = dxn.contiguous() if ctx.node_major else dxn.permute(0, 2, 1).contiguous()
                if ctx.fold & 2:                                                   # (the promise made to the layer below)
                    dx = torch.ops.aten.threshold_backward(dx, ctx.x_mask, 0)
            if ctx.has_bias and ctx.needs_input_grad[3]:
                if ctx.bias_shape[-1] == 1 or len(ctx.bias_shape) == 1:
                    db = dyn.sum(dim=(0, 1)).reshape(ctx.bias_shape)
                else:                                                              # per-node bias [F,N]
                    db = dyn.sum(dim=0).t().contiguous().reshape(ctx.bias_shape)
            return dh, None, dx, db, None, None, None, None, None, None, None
        if ctx.relu and not (ctx.fold & 1):
            dy = torch.ops.aten.threshold_backward(dy, yrelu, 0)          # dy where y > 0, else 0
        if ctx.node_major:
            return _LSIGFFunction._backward_node_major(ctx, h, S, zs, dy)
        dh = dx = db = None
        if ctx.needs_input_grad[2]:
            hT = h.detach().permute(3, 1, 2, 0)                          # [G,E,K,F] (shape only)
            dx = _lsigf_device(hT, S, dy, None, ctx.batched, ctx.Nin,
                               ctx.packed_T if ctx.packed_T is not None else _packed_transposed_taps(h),
                               transposed=True)
            if dx is None:                                               # (wide F: the dense adjoint, needs no Z)
                dx = _LSIGFFunction._dense_dx(ctx, h, S, dy.permute(0, 2, 1)).permute(0, 2, 1).contiguous()
        if ctx.needs_input_grad[0]:
            dyp = dy if ctx.Nin == N else torch.nn.functional.pad(dy, (0, N - ctx.Nin))
            dy2 = dyp.permute(1, 0, 2).reshape(F_out, B * N)             # [F, B*N] (one copy)
            # dh[f,e,k,g] = sum_(b,n) dy2[f,(b,n)] zs[(e,k),(b,n),g]: E*K small GEMMs with a long contraction,
            # split over workgroups by gnnpp_gemm_kmajor and written straight into the [F,E,K,G] layout
            dh = _native.grad_out(ctx.param_ptrs[0], (F_out, E, K, G), dy.device)
            _native.gemm_kmajor(dy2, (0, B * N, 1), zs, (B * N * G, G), dh, (G, E * K * G), E * K, F_out, G,
                                B * N)
        if ctx.has_bias and ctx.needs_input_grad[3]:
            if ctx.bias_shape[-1] == 1 or len(ctx.bias_shape) == 1:
                db = dy.sum(dim=(0, 2)).reshape(ctx.bias_shape)
            else:                                                         # per-node bias [F,N]
                db = torch.nn.functional.pad(dy.sum(dim=0), (0, N - ctx.Nin)).reshape(ctx.bias_shape)
        return dh, None, dx, db, None, None, None, None, None, None, None

    @staticmethod
    def _dense_dx(ctx, h, S, dyn):
        """dx [B,Nin,G] for dy [B,Nin,F] (node-major) through the dense adjoint (no saved tap signals needed)."""
        N, B = S.shape[-1], dyn.shape[0]
        if ctx.Nin != N:
            dyn = torch.cat([dyn, dyn.new_zeros(B, N - ctx.Nin, dyn.shape[2])], 1)
        S32 = S.detach()
        S32 = (S32 if S32.dtype is torch.float32 else S32.float()).contiguous()
        _, dxn = _lsigf_large_backward(h, S32, None, dyn.contiguous(), ctx.batched, False, True)
        return dxn[:, :ctx.Nin]

    @staticmethod
    def _backward_node_major(ctx, h, S, zs, dy):
        """dy [B,N,F] (rows (b,n), the row order of the saved tap signals): the input gradient is the
        transposed filter on dy, node-major in and out; dh and db come from ONE multi-product GEMM launch."""
        F_out, E, K, G = h.shape
        B, N = dy.shape[0], dy.shape[1]
        assert ctx.Nin == N
        dh = dx = db = None
        if ctx.needs_input_grad[2]:
            hT = h.detach().permute(3, 1, 2, 0)
            # (fold bit 1: tap signal 0 of edge feature 0 IS the filter's input x -- the mask of the ReLU that made it)
            mask = zs[0] if (ctx.fold & 2) else None
            dx = _lsigf_device(hT, S, dy, None, ctx.batched, N,
                               ctx.packed_T if ctx.packed_T is not None else _packed_transposed_taps(h),
                               transposed=True, node_major=True, out_mask=mask)
            if dx is None:
                dx = _LSIGFFunction._dense_dx(ctx, h, S, dy)
                if mask is not None:
                    dx = torch.ops.aten.threshold_backward(dx.contiguous(), mask.reshape(dx.shape), 0)
        specs, prms = [], []
        if ctx.needs_input_grad[0]:
            dh = _native.grad_out(ctx.param_ptrs[0], (F_out, E, K, G), dy.device)
            specs.append((dy, (0, 1, F_out), zs, (B * N * G, G), dh, (G, E * K * G), E * K, F_out, G, B * N))
            prms.append(h)
        if ctx.has_bias and ctx.needs_input_grad[3]:
            if ctx.bias_shape[-1] == 1 or len(ctx.bias_shape) == 1:
                db = _native.grad_out(ctx.param_ptrs[1], ctx.bias_shape, dy.device)          # sum over (b, n)
                specs.append((_ones(B * N, dy.device), (0, 0, 1), dy, (0, F_out), db, (0, F_out), 1, 1, F_out, B * N))
                prms.append(ctx.bias_ref)
            else:                                                         # per-node bias [F,N]: sum over b
                db = dy.sum(dim=0).t().contiguous().reshape(ctx.bias_shape)
        if specs:
            if ((ctx.fold & 4) and _native.deferral_allowed() and h.grad is None
                    and (ctx.bias_ref is None or ctx.bias_ref.grad is None)):
                _native.defer_gemms(specs, prms)
            else:
                _native.gemm_kmajor_multi(specs)
        return dh, None, dx, db, None, None, None, None, None, None, None


_ones_cache = {}


def _ones(n, device):
    key = (str(device), n)
    if key not in _ones_cache:
        _ones_cache[key] = torch.ones(n, dtype=torch.float32, device=device)
    return _ones_cache[key]


def _wants_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def LSIGF(h, S, x, b=None, precision=None):
    """Linear shift-invariant graph filter, one GSO for the whole batch (graphML.py:48-141).

    h [F,E,K,G] filter taps, S [E,N,N], x [B,G,N], b [F,1] (or [F,N]) -> [B,F,N].
    precision (keyword, not in the reference's signature): arithmetic of THIS call, None = DEFAULT_PRECISION.
    """
    F_out, E, K, G = h.shape
    assert S.shape[0] == E
    N = S.shape[1]
    assert S.shape[2] == N
    assert x.shape[1] == G
    assert x.shape[2] == N
    if S.dtype != x.dtype:
        # the reference multiplies x @ S without a cast (:124) and torch refuses mixed dtypes
        raise RuntimeError('expected S and x to have the same dtype, but got: %s != %s'
                           % (x.dtype, S.dtype))
    if _wants_grad(h, x, b):
        return _LSIGFFunction.apply(h, S, x, b, False, None, False, False, precision)
    return _lsigf_device(h, S, x, b, batched=False, Nin=N, precision=precision)


def BatchLSIGF(h, S, x, b=None, precision=None):
    """Same filter with one GSO per sample (graphML.py:2273-2367).  S [B,E,N,N] may be float64:
    it is rounded to fp32 on load like the reference's `S.float()` (:2350).  precision: as LSIGF."""
    F_out, E, K, G = h.shape
    assert S.shape[1] == E
    N = S.shape[2]
    assert S.shape[3] == N
    assert x.shape[1] == G
    assert x.shape[2] == N
    assert S.shape[0] == x.shape[0]
    if _wants_grad(h, x, b):
        return _LSIGFFunction.apply(h, S, x, b, True, None, False, False, precision)
    return _lsigf_device(h, S, x, b, batched=True, Nin=N, precision=precision)


class _GraphFilterBase(nn.Module):
    _batched = False

    def __init__(self, G, F, K, E=1, bias=True, precision=None):
        super().__init__()
        self.G = G
        self.F = F
        self.K = K
        self.E = E
        self.S = None
        # arithmetic of this INSTANCE's forward, fixed here (a keyword the reference's constructor does not have)
        self.precision = DEFAULT_PRECISION if precision is None else precision
        _native.precision_code(self.precision)             # (unknown names fail at construction)
        self.weight = nn.parameter.Parameter(torch.Tensor(F, E, K, G))
        if bias:
            self.bias = nn.parameter.Parameter(torch.Tensor(F, 1))
        else:
            self.register_parameter('bias', None)
        self._packed = _native.PackCache()
        self.reset_parameters()

    def reset_parameters(self):
        # graphML.py:1183-1189 / :2442-2447
        stdv = 1. / math.sqrt(self.G * self.K)
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)
        _native.invalidate_packs()              # `.data` writes are invisible to the version counters

    def packed_taps(self):
        """Fragment-ordered taps, repacked only when `weight` changed."""
        return self._packed.get((self.weight,), lambda: pack_filter_taps(self.weight))

    def forward(self, x):
        if self.S is None:
            raise TypeError('addGSO() must be called before forward()')
        _native.require_gpu(self.weight, self.S, x)
        Nin = x.shape[2]
        assert x.shape[1] == self.G
        assert Nin <= self.N
        if not self._batched and self.S.dtype != x.dtype:
            raise RuntimeError('expected S and x to have the same dtype, but got: %s != %s'
                               % (x.dtype, self.S.dtype))
        if self._batched:
            assert self.S.shape[0] == x.shape[0]
        # zero padding of the missing nodes and the final index_select (graphML.py:1206-1218 /
        # :2464-2476) are folded into the kernel through Nin
        if _wants_grad(self.weight, x, self.bias):
            return _LSIGFFunction.apply(self.weight, self.S, x, self.bias, self._batched,
                                        self.packed_taps(), False, False, self.precision)
        return _lsigf_device(self.weight, self.S, x, self.bias, self._batched, Nin,
                             packed=self.packed_taps(), precision=self.precision)

    def forward_node_major(self, x, relu=False, packed=None, packed_T=None, fold=0):
        """The same filter on x [B,N,G] -> [B,N,F] (rows = nodes; optionally followed by ReLU in the same launch),
        differentiable: the train-mode planner's path, which keeps every activation node-major so that no
        transposing copy sits between encoder, filter and action head.  Needs all N nodes (no Nin < N)."""
        if self.S is None:
            raise TypeError('addGSO() must be called before forward()')
        _native.require_gpu(self.weight, self.S, x)
        assert x.shape[2] == self.G and x.shape[1] == self.N
        if self._batched:
            assert self.S.shape[0] == x.shape[0]
        # packed / packed_T: the caller's packs of the CURRENT weight (the planner's one-launch gnnpp_train_pack: fp32
        # fragments only, which is all the exact-fp32 training launches read); fold: _LSIGFFunction.forward
        return _LSIGFFunction.apply(self.weight, self.S, x, self.bias, self._batched,
                                    packed if packed is not None else self.packed_taps(), True,
                                    bool(relu), self.precision, packed_T, fold)

    def extra_repr(self):
        s = 'in_features=%d, out_features=%d, ' % (self.G, self.F)
        s += 'filter_taps=%d, ' % self.K + 'edge_features=%d, ' % self.E
        s += 'bias=%s, ' % (self.bias is not None)
        s += 'GSO stored' if self.S is not None else 'no GSO stored'
        return s


class GraphFilter(_GraphFilterBase):
    """Graph filtering layer with ONE GSO for the batch (graphML.py:1111-1230).
    addGSO(S [E,N,N]); forward(x [B,G,Nin]) -> [B,F,Nin]."""
    _batched = False

    def addGSO(self, S):
        assert len(S.shape) == 3
        assert S.shape[0] == self.E
        self.N = S.shape[1]
        assert S.shape[2] == self.N
        self.S = S


class GraphFilterBatch(_GraphFilterBase):
    """Graph filtering layer with one GSO per sample (graphML.py:2369-2488).
    addGSO(S [B,E,N,N]); forward(x [B,G,Nin]) -> [B,F,Nin]."""
    _batched = True

    def addGSO(self, S):
        assert len(S.shape) == 4
        assert S.shape[1] == self.E
        self.N = S.shape[2]
        assert S.shape[3] == self.N
        self.S = S


def matrixPowersBatch(S, K):
    """S^k for k = 0..K-1 per batch element (graphML.py:2063-2113).  S [B,N,N] -> [B,K,N,N];
    S [B,E,N,N] -> [B,E,K,N,N].  The result tensor is allocated once and power k is written in place
    from power k-1 (K-2 batched products, no concatenation copies); the return value is a [B,E,K,N,N] view of
    that power-major buffer.  fp32 GSOs on the GPU: each product is ONE gnnpp_gemm_kmajor launch (exact fp32 MFMA,
    batch = B*E, r06; r05: torch.bmm); anything else (fp64 GSOs keep the reference's fp64 powers; CPU tensors of the
    host-side tests) goes through torch.bmm."""
    assert S.dim() in (3, 4)
    scalar = S.dim() == 3
    S4 = S.unsqueeze(1) if scalar else S
    B, E, N = S4.shape[0], S4.shape[1], S4.shape[2]
    assert S4.shape[3] == N
    flat = S4.reshape(B * E, N, N)
    P = flat.new_zeros(max(K, 1), B * E, N, N)             # power-major: every P[k] is contiguous
    P[0].diagonal(dim1=-2, dim2=-1).fill_(1)
    if K > 1:
        P[1] = flat
    own = flat.is_cuda and flat.dtype is torch.float32
    if own and K > 2 and not flat.is_contiguous():
        flat = flat.contiguous()
    for k in range(2, K):
        if own:     # P[k][b] (m, n) = sum_j P[k-1][b] (m, j) * S[b] (j, n): A rows contiguous in j, B rows contiguous in n
            _native.gemm_kmajor(P[k - 1], (N * N, N, 1), flat, (N * N, N), P[k], (N * N, N), B * E, N, N, N)
        else:
            torch.bmm(P[k - 1], flat, out=P[k])
    SK = P.reshape(max(K, 1), B, E, N, N).permute(1, 2, 0, 3, 4)
    return SK.squeeze(1) if scalar else SK


def batchLSIGF(h, SK, x, bias=None, precision=None):
    """Graph filter on given per-sample matrices SK [B,E,K,N,N] (graphML.py:2115-2178):
    y[b] = bias + sum_{e,k} h[:,e,k,:] . (x[b] SK[b,e,k]).  Each (e,k) pair is an independent
    one-hop shift of x, i.e. the LSIGF kernel with E*K "edge features", two taps and a zero tap 0.
    precision: arithmetic of this call (keyword; None = DEFAULT_PRECISION)."""
    F_out, E, K, G = h.shape
    B = SK.shape[0]
    assert SK.shape[1] == E
    assert SK.shape[2] == K
    N = SK.shape[3]
    assert SK.shape[4] == N
    assert x.shape[0] == B
    assert x.shape[1] == G
    assert x.shape[2] == N
    h2 = torch.zeros(F_out, E * K, 2, G, dtype=h.dtype, device=h.device)
    h2[:, :, 1, :] = h.reshape(F_out, E * K, G)
    S2 = SK.reshape(B, E * K, N, N)
    if _wants_grad(h, x, bias):
        return _LSIGFFunction.apply(h2, S2, x, bias, True, None, False, False, precision)
    return _lsigf_device(h2, S2, x, bias, batched=True, Nin=N, precision=precision)


class GraphFilterBatchGSO(GraphFilter):
    """Graph filtering layer with a different GSO per sample, powers precomputed at addGSO
    (graphML.py:2180-2271).  addGSO(S [B,N,N] | [B,E,N,N]); forward(x [B,G,N]) -> [B,F,N]."""

    def __init__(self, G, F, K, E=1, bias=True, precision=None):
        super().__init__(G, F, K, E, bias, precision)

    def addGSO(self, S):
        if len(S.shape) == 3 and S.shape[1] == S.shape[2]:
            self.S = S.unsqueeze(1)
        elif len(S.shape) == 4 and S.shape[1] == self.E and S.shape[2] == S.shape[3]:
            self.S = S
        self.N = self.S.shape[2]
        self.B = self.S.shape[0]
        self.SK = matrixPowersBatch(self.S, self.K)

    def forward(self, x):
        return batchLSIGF(self.weight, self.SK, x, self.bias, self.precision)

    def extra_repr(self):
        s = 'in_features=%d, out_features=%d, ' % (self.G, self.F)
        s += 'filter_taps=%d, ' % self.K + 'edge_features=%d, ' % self.E
        s += 'bias=%s, ' % (self.bias is not None)
        if self.S is not None:
            s += 'GSO stored: number_nodes=%d, batch_size=%d' % (self.N, self.B)
        else:
            s += 'no GSO stored'
        return s
