"""Graph-filter operators with the reference's API, computed by libgnnpp.so on MI355X.

Mirrors the part of utils/graphUtils/graphML.py that the path planner uses:

    LSIGF(h, S, x, b=None)            graphML.py:48-141      one GSO shared by the batch
    BatchLSIGF(h, S, x, b=None)       graphML.py:2273-2367   one GSO per sample
    GraphFilter(G, F, K, E=1, bias)   graphML.py:1111-1230
    GraphFilterBatch(G, F, K, E, b)   graphML.py:2369-2488

Same names, argument meaning, shape asserts, parameter names/shapes (`weight [F,E,K,G]`,
`bias [F,1]`), init rule and `addGSO` / `forward` / `extra_repr` behaviour, so the reference's
callers run unchanged.  Layouts at this boundary are the reference's feature-major
x[B,G,N] -> y[B,F,N]; the kernel transposes through LDS on load/store (no extra HBM pass).

Forward only in this round: outputs carry no autograd graph (backward is SURVEY.md section 8f
row 2).  There is no CPU path: tensors must be on a HIP device.
"""
import ctypes
import math

import torch
import torch.nn as nn

from . import _native

zeroTolerance = 1e-9    # kept for API parity (graphML.py:42-43)
infiniteNumber = 1e12

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


def _lsigf_device(h, S, x, b, batched, Nin, packed=None, relu=False):
    """Shared driver: h [F,E,K,G], S [E,N,N] | [B,E,N,N], x [B,G,Nin] -> y [B,F,Nin]."""
    dev = _native.require_gpu(h, S, x, b)
    L = _native.lib()
    F_out, E, K, G = h.shape
    N = S.shape[-1]
    B = x.shape[0]
    if N > MAX_NODES:
        raise _native.GnnppError('graphs with N=%d > %d nodes are not supported yet' % (N, MAX_NODES))
    if F_out > _MAX_F_PER_LAUNCH:
        # wide filters: split the output features (each chunk recomputes the cheap shifts)
        outs = []
        for f0 in range(0, F_out, _MAX_F_PER_LAUNCH):
            f1 = min(F_out, f0 + _MAX_F_PER_LAUNCH)
            bb = None if b is None else b[f0:f1]
            outs.append(_lsigf_device(h[f0:f1], S, x, bb, batched, Nin, None, relu))
        return torch.cat(outs, dim=1)
    xc = x.detach().contiguous()
    if xc.dtype != torch.float32:
        xc = xc.float()
    Sc = S.detach().contiguous()
    if Sc.dtype not in (torch.float32, torch.float64):
        Sc = Sc.float()
    if packed is None:
        packed = pack_filter_taps(h)
    fused_bias = None
    if b is not None and b.numel() == F_out:
        fused_bias = b.detach().contiguous().float().reshape(-1)
    y = torch.empty(B, F_out, Nin, dtype=torch.float32, device=dev)
    with _native.device_guard(dev):
        rc = L.gnnpp_lsigf_fwd(_ptr(xc), _ptr(Sc), _ptr(packed), _ptr(fused_bias), _ptr(y),
                               B, N, Nin, G, F_out, K, E, int(Sc.dtype == torch.float64),
                               int(batched), 0, 0, int(relu and (b is None or fused_bias is not None)),
                               _native.stream_ptr(dev))
    _native.check(rc, 'gnnpp_lsigf_fwd')
    if b is not None and fused_bias is None:      # per-node bias [F,N]: not fused
        y = y + b.detach()[:, :Nin]
        if relu:
            y = torch.relu_(y)
    return y


def LSIGF(h, S, x, b=None):
    """Linear shift-invariant graph filter, one GSO for the whole batch (graphML.py:48-141).

    h [F,E,K,G] filter taps, S [E,N,N], x [B,G,N], b [F,1] (or [F,N]) -> [B,F,N].
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
    return _lsigf_device(h, S, x, b, batched=False, Nin=N)


def BatchLSIGF(h, S, x, b=None):
    """Same filter with one GSO per sample (graphML.py:2273-2367).  S [B,E,N,N] may be float64:
    it is rounded to fp32 on load like the reference's `S.float()` (:2350)."""
    F_out, E, K, G = h.shape
    assert S.shape[1] == E
    N = S.shape[2]
    assert S.shape[3] == N
    assert x.shape[1] == G
    assert x.shape[2] == N
    assert S.shape[0] == x.shape[0]
    return _lsigf_device(h, S, x, b, batched=True, Nin=N)


class _GraphFilterBase(nn.Module):
    _batched = False

    def __init__(self, G, F, K, E=1, bias=True):
        super().__init__()
        self.G = G
        self.F = F
        self.K = K
        self.E = E
        self.S = None
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
        return _lsigf_device(self.weight, self.S, x, self.bias, self._batched, Nin,
                             packed=self.packed_taps())

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
