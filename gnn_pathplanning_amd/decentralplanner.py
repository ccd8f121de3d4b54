"""DecentralPlannerNet with the reference's API, computed by libgnnpp.so on MI355X.

Mirrors graphs/models/decentralplanner.py:13-318 ("DCP v1.4", :89-98): per-agent CNN encoder
(5 x [conv3x3 + BatchNorm + ReLU], MaxPool after layers 0/2/4) -> Linear(128,128)+ReLU ->
GraphFilterBatch(128,128,K,E=1)+ReLU -> Linear(128,5).

  * constructor reads config.num_agents, config.nGraphFilterTaps, config.device (:18, :131, :283);
  * identical sub-module tree, hence identical state_dict keys/shapes (ConvLayers.{0,1,4,5,...},
    compressMLP.0, GFL.0, actionsMLP.0) so reference checkpoints load unchanged;
  * addGSO(S [B,N,N]) (:266-276) then forward(inputTensor [B,N,3,11,11]) -> list of N tensors [B,5]
    (:278-318).

The sub-modules are parameter containers: forward() does NOT call them.  The whole step is two
HIP kernels behind one C call (gnnpp_policy_fwd): the fused encoder over all B*N agents, then the
K-tap graph filter + ReLU + action head.  Packed/BN-folded weights are cached and rebuilt whenever
a parameter or running statistic changes.

Eval mode (the rollout loop, agents/decentralplannerlocal.py:489-592) is the fully fused HIP path.

Train mode (agents/decentralplannerlocal.py:283-317) keeps the reference's exact semantics: the
encoder runs once per agent so BatchNorm normalises with per-agent-call batch statistics and
updates its running statistics N times per forward (decentralplanner.py:284-290).  That needs
batch-wide reductions per agent between the layers, so train mode is a layer-by-layer schedule of
hand-written HIP kernels, forward and backward (csrc/train_encoder.hip behind
gnnpp_encoder_train_fwd / _bwd), plus the graph filter (forward, input gradient, tap gradient) on
lsigf_kernel through graphML._LSIGFFunction; compressMLP and the action head forward on gnnpp_linear_fwd, their
backward products on gnnpp_gemm_kmajor_multi.
There is no CPU path.
"""
import ctypes
import sys

import torch
import torch.nn as nn

from . import _native
from . import graphML as gml
from .weights_initializer import weights_init

# The architecture the kernels implement (graphs/models/decentralplanner.py:93-197: 3 -> 32 -> 32 -> 64 -> 64 -> 128
# channels of 3x3 convolutions on the 11x11 field of view, a 2x2 max-pool after layers 0, 2 and 4, 128 features,
# 5 actions): (input channels, output channels, pooled) per layer, and where that puts Conv2d / BatchNorm2d in
# the reference's nn.Sequential -- the state_dict keys.
_ENCODER_LAYERS = ((3, 32, True), (32, 32, False), (32, 64, True), (64, 64, False), (64, 128, True))
_FEATURES = 128
_ACTIONS = 5
_CONV_IDX = (0, 4, 7, 11, 14)
_BN_IDX = (1, 5, 8, 12, 15)
# precision of a planner whose config does not name one (see DecentralPlannerNet.precision)
DEFAULT_PRECISION = 'fp32'


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


_enc_params_cache = {}


def _encoder_params(ps, buffers, eps):
    """struct gnnpp_encoder_params of the 20 parameter tensors (+ the 10 running-statistics buffers): filling a
    ctypes struct costs more host time than a kernel launch, and the pointers of a model do not move from step to
    step, so the struct is kept per pointer tuple."""
    key = tuple(t.data_ptr() for t in ps) + (tuple(t.data_ptr() for t in buffers) if buffers is not None else ()) \
        + (float(eps),)
    p = _enc_params_cache.get(key)
    if p is None:
        if len(_enc_params_cache) > 64:
            _enc_params_cache.clear()
        p = _native.EncoderParams()
        for i in range(5):
            p.conv_w[i], p.conv_b[i] = ps[4 * i].data_ptr(), ps[4 * i + 1].data_ptr()
            p.bn_w[i], p.bn_b[i] = ps[4 * i + 2].data_ptr(), ps[4 * i + 3].data_ptr()
            if buffers is not None:
                p.bn_mean[i], p.bn_var[i] = buffers[2 * i].data_ptr(), buffers[2 * i + 1].data_ptr()
        p.bn_eps = float(eps)
        _enc_params_cache[key] = p
    return p


class _EncoderTrainFunction(torch.autograd.Function):
    """Train-mode ConvLayers of ALL agents on the HIP kernels of csrc/train_encoder.hip.

    obs [B,N,3,11,11] -> feat [B,N,128] (row (b,n) = the flattened ConvLayers output of agent call n, i.e. what the
    reference hands to compressMLP at decentralplanner.py:287-289), with the reference's per-agent-call
    BatchNorm statistics; the running statistics receive their N momentum updates in place.  backward =
    gnnpp_encoder_train_bwd: gradients of the 20 conv / BatchNorm parameters (no gradient to obs: it is
    data).  `tensors` = [conv_w, conv_b, bn_w, bn_b] x 5 layers; `buffers` = [running_mean, running_var] x 5."""

    @staticmethod
    def forward(ctx, obs, buffers, counters, momentum, eps, enc_pack, *tensors):
        L = _native.lib()
        B, N = obs.shape[0], obs.shape[1]
        dev = obs.device
        ps = [t.detach() if (t.dtype is torch.float32 and t.is_contiguous()) else t.detach().contiguous().float()
              for t in tensors]
        p = _encoder_params(ps, buffers, eps)
        ws = torch.empty(L.gnnpp_encoder_train_workspace_floats(N, B), dtype=torch.float32, device=dev)
        feat = torch.empty(B, N, 128, dtype=torch.float32, device=dev)     # sample-major: node-major rows
        with _native.device_guard(dev):
            _native.check(L.gnnpp_encoder_train_fwd(ctypes.byref(p), _ptr(obs), _ptr(ws), _ptr(feat), B, N,
                                                    ctypes.c_float(momentum), int(buffers is not None),
                                                    (ctypes.c_void_p * 5)(*[c.data_ptr() for c in counters])
                                                    if counters is not None else None, 1, _ptr(enc_pack),
                                                    _native.stream_ptr(dev)), 'gnnpp_encoder_train_fwd')
        ctx.save_for_backward(obs, ws, *ps)
        ctx.eps = float(eps)
        ctx.enc_pack = enc_pack                    # (the pack of THESE weights: the backward call reads its other half)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        L = _native.lib()
        obs, ws = ctx.saved_tensors[0], ctx.saved_tensors[1]
        ps = ctx.saved_tensors[2:]
        B, N = obs.shape[0], obs.shape[1]
        dev = obs.device
        p, g = _encoder_params(ps, None, ctx.eps), _native.EncoderGrads()
        grads = [_native.grad_out(t.data_ptr(), t.shape, dev) for t in ps]       # (FlatBucketDP: slices of its bucket)
        for i in range(5):
            g.conv_w[i], g.conv_b[i] = grads[4 * i].data_ptr(), grads[4 * i + 1].data_ptr()
            g.bn_w[i], g.bn_b[i] = grads[4 * i + 2].data_ptr(), grads[4 * i + 3].data_ptr()
        d = dfeat.contiguous().float()
        with _native.device_guard(dev):
            _native.check(L.gnnpp_encoder_train_bwd(ctypes.byref(p), _ptr(obs), _ptr(ws), _ptr(d), ctypes.byref(g),
                                                    B, N, 1, _ptr(ctx.enc_pack), _native.stream_ptr(dev)),
                          'gnnpp_encoder_train_bwd')
        return (None, None, None, None, None, None) + tuple(grads)


class _LinearFunction(torch.autograd.Function):
    """y = x W^T + b (optionally followed by ReLU) for x [..., I] with a few hundred rows (compressMLP, the action head
    in train mode).  Forward: gnnpp_linear_fwd -- one launch, bias and ReLU in its epilogue (r05: a library GEMM + an
    aten ReLU).  The backward's three products are "small output, long or short contraction" shapes that a library
    GEMM serves with one macro tile -- they run on gnnpp_gemm_kmajor (contraction split over workgroups,
    deterministic):  dx = dy W,  dW = dy^T x,  db = 1^T dy.
      relu      0: none; 1: y = relu(.), the backward masks dy itself; 2: y = relu(.) and the CONSUMER of y promises
                to hand back a gradient that is already masked by y > 0 (the graph filter's input-gradient launch
                does: graphML._LSIGFFunction fold bit 1)
      mask_dx   x is itself the output of a ReLU whose backward is folded into THIS function's dx product (dx is
                stored as 0 where x <= 0; the producer of x must then not mask again: _LSIGFFunction fold bit 0)
      defer     1: the products that only yield parameter gradients (dW, db) wait in _native's queue for a later launch
                of the same backward pass (r06b: they are not on the backward chain) -- only while W / b have no `.grad`
                yet, so that autograd stores the result tensors without reading them; 2: this backward launch carries
                the queue with it (one multiply + one reduce launch for everything queued so far and its own products)"""

    @staticmethod
    def forward(ctx, x, W, b, relu=0, mask_dx=False, defer=0):
        ctx.param_ptrs = (W.data_ptr(), b.data_ptr() if b is not None else 0)
        ctx.relu, ctx.mask_dx, ctx.defer = int(relu), bool(mask_dx), int(defer)
        O, I = W.shape
        xd, Wd = x.detach(), W.detach()
        y = None
        if (x.is_cuda and xd.dtype is torch.float32 and Wd.dtype is torch.float32 and xd.is_contiguous()
                and Wd.is_contiguous() and I % 64 == 0 and (b is None or b.dtype is torch.float32)):
            R = xd.numel() // I
            y = torch.empty(x.shape[:-1] + (O,), dtype=torch.float32, device=x.device)
            with _native.device_guard(x.device):
                rc = _native.lib().gnnpp_linear_fwd(_ptr(xd), _ptr(Wd), _ptr(b.detach().contiguous()) if b is not None
                                                    else None, _ptr(y), R, I, O, int(relu != 0),
                                                    _native.stream_ptr(x.device))
            if rc == -2:
                y = None                                      # GNNPP_ERR_UNSUPPORTED (alignment): the library GEMM
            else:
                _native.check(rc, 'gnnpp_linear_fwd')
        if y is None:
            y = torch.nn.functional.linear(xd, Wd, b.detach() if b is not None else None)
            if relu:
                y = torch.relu_(y)
        ctx.save_for_backward(x, W, y if relu == 1 else None, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, yrelu, b = ctx.saved_tensors
        O, I = W.shape
        dy2 = dy.reshape(-1, O).contiguous().float()
        if yrelu is not None:
            dy2 = torch.ops.aten.threshold_backward(dy2, yrelu.reshape(-1, O), 0)
        x2 = x.detach().reshape(-1, I).contiguous().float()
        R = dy2.shape[0]
        dx = dW = db = None
        specs, pspecs = [], []                                 # the three products: ONE launch (+ one for the sums)
        if ctx.needs_input_grad[0]:
            dx = torch.empty(R, I, dtype=torch.float32, device=dy.device)
            specs.append((dy2, (0, O, 1), W.detach().contiguous().float(), (0, I), dx, (0, I), 1, R, I, O,
                          x2 if ctx.mask_dx else None))
        if ctx.needs_input_grad[1]:
            dW = _native.grad_out(ctx.param_ptrs[0], (O, I), dy.device)
            pspecs.append((dy2, (0, 1, O), x2, (0, I), dW, (0, I), 1, O, I, R))
        if ctx.needs_input_grad[2]:
            db = _native.grad_out(ctx.param_ptrs[1], (O,), dy.device)
            pspecs.append((_ones(R, dy.device), (0, 0, 1), dy2, (0, O), db, (0, O), 1, 1, O, R))
        if ctx.defer == 2:
            _native.flush_deferred_gemms(specs + pspecs)
        elif (ctx.defer == 1 and _native.deferral_allowed() and W.grad is None and (b is None or b.grad is None)):
            _native.defer_gemms(pspecs, ([W] if dW is not None else []) + ([b] if db is not None else []))
            if specs:
                _native.gemm_kmajor_multi(specs)
        elif specs or pspecs:
            _native.gemm_kmajor_multi(specs + pspecs)
        return (dx.reshape(x.shape) if dx is not None else None), dW, db, None, None, None


_ones_cache = {}


def _ones(n, device):
    key = (str(device), n)
    if key not in _ones_cache:
        _ones_cache[key] = torch.ones(n, dtype=torch.float32, device=device)
    return _ones_cache[key]


class LogitList(list):
    """The reference's return value -- a python list of N tensors [B,5] -- built as the N views of ONE
    tensor [N,B,5], which stays reachable as .stacked.  Autograd sees one unbind (its backward is one
    concatenation) instead of N selects (N x zero-fill + copy + accumulate per step), and
    training.policy_loss() takes .stacked directly instead of re-stacking the list."""

    def __init__(self, stacked):
        super().__init__(stacked.unbind(0))
        self.stacked = stacked


_getrefcount = sys.getrefcount
_tensor_use_count = getattr(torch._C.TensorBase, '_use_count', None)
_storage_use_count = getattr(torch._C, '_storage_Use_Count', None)


def _stream_capturing():
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


class _OutputSlot:
    """Where eval-mode forward() gets its logits buffer AND the reference's list of N views from.

    The reference returns a python list of N tensors [B,5] (decentralplanner.py:304-318).  Building N view objects
    costs the host ~0.3 us each -- 30 us at N = 100, more than the per-GPU shard of config 5 leaves room for (VERDICT
    r05 item 4: 25.5 M agent-steps/s through forward() vs 36.8 M through forward_logits()).  A `list` subclass that
    materialises lazily does not survive C-level consumers (torch.stack(out, 1) reads ob_item directly and would see
    an empty list), so the list stays a real one and the cost is removed instead: the [N,B,5] buffer of the previous
    step and its N views are handed out AGAIN when nothing can observe that -- the python refcount of every view, the
    TensorImpl use count of every view (autograd SavedVariables, DLPack capsules) and the use count of the storage
    (derived views, .detach(), reshapes) are all back at the values they had when only this slot held them, i.e. the
    caller has dropped the previous list and everything made from it.  Otherwise a fresh buffer + fresh views are
    built, exactly as before.  One entry per (N, B, device, stream): a buffer is only re-used on the stream that
    wrote it last.  Never during a HIP-graph capture (the captured kernels' addresses must come from the graph's
    private pool)."""
    __slots__ = ('entries', 'views', 'recycled', 'fresh')

    def __init__(self):
        self.entries, self.views, self.recycled, self.fresh = {}, None, 0, 0

    def clear(self):
        self.entries.clear()
        self.views = None

    # a copied / pickled model starts with an empty slot (the entries are this process's device buffers)
    def __deepcopy__(self, memo):
        return _OutputSlot()

    def __reduce__(self):
        return (_OutputSlot, ())

    def acquire(self, N, B, dev, stream):
        if _tensor_use_count is None or _storage_use_count is None or _stream_capturing():
            self.views = None                            # forward() falls back to logits.unbind(0)
            return torch.empty(N, B, _ACTIONS, dtype=torch.float32, device=dev)
        key = (N, B, dev.index, stream)
        e = self.entries.get(key)
        if e is not None:
            stacked, views, stg, ref_py, ref_st = e
            if (sum(map(_getrefcount, views)) == ref_py and sum(map(_tensor_use_count, views)) == N
                    and _storage_use_count(stg._cdata) == ref_st):
                self.views = views
                self.recycled += 1
                return stacked
        stacked = torch.empty(N, B, _ACTIONS, dtype=torch.float32, device=dev)
        views = stacked.unbind(0)                        # a tuple: the caller's list is its own copy
        stg = stacked.untyped_storage()
        if len(self.entries) > 8:
            self.entries.clear()
        self.entries[key] = (stacked, views, stg, sum(map(_getrefcount, views)), _storage_use_count(stg._cdata))
        self.views = views
        self.fresh += 1
        return stacked


class DecentralPlannerNet(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.S = None
        self.numAgents = self.config.num_agents

        # The module tree below IS the state_dict contract with the reference (graphs/models/decentralplanner.py:
        # ConvLayers.{0,1,4,5,7,8,11,12,14,15}, compressMLP.0, GFL.0, actionsMLP.0); the kernels fix the
        # architecture (_ENCODER_LAYERS: 11x11 observations -> 128 features), so it is written out as a table.
        convs = []
        for cin, cout, pool in _ENCODER_LAYERS:
            convs += [nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1, bias=True), nn.BatchNorm2d(cout),
                      nn.ReLU(inplace=True)]
            if pool:
                convs.append(nn.MaxPool2d(kernel_size=2))
        self.ConvLayers = nn.Sequential(*convs)
        assert [i for i, m in enumerate(convs) if isinstance(m, nn.Conv2d)] == list(_CONV_IDX)
        self.compressMLP = nn.Sequential(nn.Linear(_FEATURES, _FEATURES, bias=True), nn.ReLU(inplace=True))
        self.numFeatures2Share = _FEATURES

        # Graph-filter layers.  The reference fixes ONE layer in its source (dimNodeSignals = [2**7],
        # nGraphFilterTaps = [config.nGraphFilterTaps] at :130-131, E = 1 at :208) but builds and runs L layers /
        # E edge features generically (:205-224, :266-276, :293-298).  The same generality is reachable here
        # without editing source, through optional config fields: a LIST in `nGraphFilterTaps`,
        # `dimNodeSignals` (list of layer widths), `numEdgeFeatures`.
        taps = self.config.nGraphFilterTaps
        self.K = list(taps) if isinstance(taps, (list, tuple)) else [taps]
        widths = list(getattr(self.config, 'dimNodeSignals', None) or [_FEATURES] * len(self.K))
        assert len(widths) == len(self.K)
        self.L = len(self.K)
        self.F = [_FEATURES] + widths
        self.E = int(getattr(self.config, 'numEdgeFeatures', 1))
        self.bias = True
        gfl = []
        for l in range(self.L):
            gfl += [gml.GraphFilterBatch(self.F[l], self.F[l + 1], self.K[l], self.E, self.bias),
                    nn.ReLU(inplace=True)]
        self.GFL = nn.Sequential(*gfl)
        self.actionsMLP = nn.Sequential(nn.Linear(self.F[-1], _ACTIONS, bias=True))
        self.apply(weights_init)
        self._enc_cache = _native.PackCache()
        self._head_cache = _native.PackCache()
        self._train_pack_cache = _native.PackCache()
        self._ws = None
        self._ws_key = None
        self._ws_all = {}                          # feature workspaces by (rows, stream)
        self._out_slot = _OutputSlot()             # eval-mode forward(): recycled logits buffer + its N views
        # Arithmetic of the matrix-pipe contractions (include/gnnpp.h GNNPP_PREC_*), passed to the kernels PER CALL:
        #   'fp32' (default)  fp32-equivalent bf16x3 operand split: no input domain, nothing for the caller to poll --
        #                     what an unchanged caller of the reference (agents/decentralplannerlocal.py:575-588) gets;
        #   'fp32_mfma'       exact fp32 MFMA (bitwise an fmaf chain);
        #   'split_f16'       opt-in fast mode, NARROWER than fp32 (22-bit operands, |activation| < 65504).  Its
        #                     range guard is a device int the kernels raise; range_policy 'strict' (default for this
        #                     mode) reads it back after every forward and re-runs an out-of-range call with 'fp32';
        #                     'flag' leaves the polling to the caller (check_range(); BatchedRollout.run does it).
        self.precision = getattr(self.config, 'precision', None) or DEFAULT_PRECISION
        self.range_policy = getattr(self.config, 'range_policy', 'strict')
        self._range_flag = None

    # ------------------------------------------------------------------------------------
    def addGSO(self, S):
        # B x N x N, or B x E x N x N when E > 1 (decentralplanner.py:266-276)
        # (a plain attribute: written through __dict__, nn.Module.__setattr__ costs ~2.5 us per call)
        if self.E == 1:
            assert len(S.shape) == 3
            self.__dict__['S'] = S.unsqueeze(1)
        else:
            assert len(S.shape) == 4
            assert S.shape[1] == self.E
            self.__dict__['S'] = S

    def _encoder_tensors(self):
        """The 32 tensors the packed encoder depends on.  Walking nn.Sequential / __getattr__ costs
        ~40 us per call, so the list is memoised together with the (dict, key) slot each tensor -- and each
        sub-module on the way to it -- lives in: every call re-reads those ~50 slots (plain dict lookups, ~3 us)
        and compares OBJECT identities, so a Parameter, buffer or sub-module replaced by hand
        (`conv.weight = nn.Parameter(...)`, `bn.running_mean = t`, `net.actionsMLP[0] = nn.Linear(..)`) is seen by
        the very next forward (the previous every-64th-call refresh left a 63-forward window)."""
        d = self.__dict__
        t = d.get('_enc_tensors')
        if t is not None:
            for store, key, obj in d['_enc_slots']:
                if store.get(key) is not obj:
                    t = None
                    break
        if t is None:
            t, slots = [], []
            for ci, bi in zip(_CONV_IDX, _BN_IDX):
                conv, bn = self.ConvLayers[ci], self.ConvLayers[bi]
                slots += [(self.ConvLayers._modules, str(ci), conv), (self.ConvLayers._modules, str(bi), bn)]
                for mod in (conv, bn):
                    for nm in ('weight', 'bias'):
                        t.append(mod._parameters[nm])
                        slots.append((mod._parameters, nm, t[-1]))
                for nm in ('running_mean', 'running_var'):
                    t.append(bn._buffers[nm])
                    slots.append((bn._buffers, nm, t[-1]))
            fc = self.compressMLP[0]
            slots.append((self.compressMLP._modules, '0', fc))
            for nm in ('weight', 'bias'):
                t.append(fc._parameters[nm])
                slots.append((fc._parameters, nm, t[-1]))
            gfs, act = tuple(self.GFL[2 * l] for l in range(self.L)), self.actionsMLP[0]
            slots += [(self.GFL._modules, str(2 * l), gf) for l, gf in enumerate(gfs)]
            slots.append((self.actionsMLP._modules, '0', act))
            slots += [(self._modules, nm, self._modules[nm]) for nm in ('ConvLayers', 'compressMLP', 'GFL', 'actionsMLP')]
            d['_enc_tensors'] = t
            d['_enc_slots'] = slots
            d['_mods'] = (gfs, act)
        return t

    def invalidate_packed(self):
        """Rebuild every packed / BN-folded weight copy on the next forward.  Call it after updating
        parameters in a way torch's version counters cannot see (`p.data.mul_()`, writes through
        raw pointers, a captured-graph replay -- training.GraphedTrainStep does it for you)."""
        self._enc_tensors = None
        _native.invalidate_packs()

    def train(self, mode=True):
        if mode != self.training:                 # weights usually changed in between: repack once
            self._enc_tensors = None
            _native.invalidate_packs()
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        self._enc_tensors = None
        self._range_flag = None
        self._out_slot.clear()
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._enc_tensors = None
        return super().load_state_dict(*args, **kwargs)

    # ---- range guard -----------------------------------------------------------------------------
    def _flag(self, dev):
        if self._range_flag is None or self._range_flag.device != dev:
            self._range_flag = torch.zeros(1, dtype=torch.int32, device=dev)
        return self._range_flag

    def _prec(self):
        return _native.precision_code(self.precision)

    def range_exceeded(self):
        """True iff some forward since the last reset handed |activation| >= 65504 to the split-f16
        schedules (precision='split_f16' only; synchronises the device)."""
        return self._range_flag is not None and bool(self._range_flag.item())

    def check_range(self, reset=True):
        """Raise GnnppError if a forward left the f16 range of precision='split_f16' (the default precision has
        no input domain and never raises)."""
        if self.range_exceeded():
            if reset:
                self._range_flag.zero_()
            raise _native.GnnppError(_native.lib().gnnpp_error_string(-4).decode())

    def _pack_encoder(self):
        L = _native.lib()
        ts = [x.detach().contiguous().float() for x in self._encoder_tensors()]
        dev = _native.require_gpu(*ts)
        p = _native.EncoderParams()
        for i in range(5):
            cw, cb, bw, bb, bm, bv = ts[6 * i:6 * i + 6]
            p.conv_w[i], p.conv_b[i] = cw.data_ptr(), cb.data_ptr()
            p.bn_w[i], p.bn_b[i] = bw.data_ptr(), bb.data_ptr()
            p.bn_mean[i], p.bn_var[i] = bm.data_ptr(), bv.data_ptr()
        p.fc_w, p.fc_b = ts[30].data_ptr(), ts[31].data_ptr()
        p.bn_eps = float(self.ConvLayers[_BN_IDX[0]].eps)
        packed = torch.empty(L.gnnpp_encoder_packed_floats(), dtype=torch.float32, device=dev)
        with _native.device_guard(dev):
            _native.check(L.gnnpp_encoder_pack(ctypes.byref(p), _ptr(packed),
                                               _native.stream_ptr(dev)), 'gnnpp_encoder_pack')
        return packed

    def packed_encoder(self):
        return self._enc_cache.get(self._encoder_tensors(), self._pack_encoder)

    def encode(self, inputTensor, precision=None):
        """extractFeatureMap of the reference (decentralplanner.py:283-290), node-major:
        inputTensor [B,N,3,11,11] -> [B,N,128]."""
        B, N = inputTensor.shape[0], inputTensor.shape[1]
        obs = inputTensor.detach().contiguous().float()
        dev = _native.require_gpu(obs, self.compressMLP[0].weight)
        feat = torch.empty(B, N, 128, dtype=torch.float32, device=dev)
        prec = self._prec() if precision is None else _native.precision_code(precision)
        with _native.device_guard(dev):
            _native.check(_native.lib().gnnpp_encoder_fwd(
                _ptr(obs), _ptr(self.packed_encoder()), _ptr(feat), B * N, prec,
                _ptr(self._flag(dev)) if prec == _native.PREC_SPLIT_F16 else None,
                _native.stream_ptr(dev)), 'gnnpp_encoder_fwd')
        return feat

    @staticmethod
    def _head_pointers(gf, act):
        # fp32 kernels read these through raw pointers: cast (a no-op for fp32 modules) and keep
        # the casted tensors alive in the cache entry
        gb = gf.bias.detach().reshape(-1).contiguous().float() if gf.bias is not None else None
        aw, ab = act.weight.detach().contiguous().float(), act.bias.detach().contiguous().float()
        return (gb.data_ptr() if gb is not None else None, aw.data_ptr(), ab.data_ptr(), (gb, aw, ab))

    def policy_pointers(self):
        """(encoder pack, filter taps, GFL bias, head weight, head bias) raw pointers + K for the
        C entry points that run the policy inside a larger kernel (BatchedRollout's one-launch step).
        The tensors behind them are cached on the module and stay alive with it.  None when the
        model is not the single-layer, single-edge-feature planner those kernels implement."""
        if self.L != 1 or self.E != 1 or self.F[-1] != 128:
            return None
        enc = self.packed_encoder()                        # (also refreshes self._mods)
        (gf,), act = self._mods
        taps = gf.packed_taps()
        gb_p, aw_p, ab_p, _keep = self._head_cache.get(
            (gf.bias, act.weight, act.bias) if gf.bias is not None else (act.weight, act.bias),
            lambda: self._head_pointers(gf, act))
        return enc.data_ptr(), taps.data_ptr(), gb_p, aw_p, ab_p, gf.K

    def materialize_packs(self):
        """Build every lazily cached device copy of the weights (packed encoder, packed taps of every graph-filter
        layer, the head's pointer tensors) NOW, on the current stream; returns True when something was (re)built.
        rollout.GroupedRollout calls it before forking onto its group streams."""
        before = (self._enc_cache.key, self._head_cache.key) + tuple(gf._packed.key for gf in self.__dict__.get('_mods', ((), None))[0])
        self.packed_encoder()
        gfs, act = self._mods
        for gf in gfs:
            gf.packed_taps()
        gl = gfs[-1]
        self._head_cache.get((gl.bias, act.weight, act.bias) if gl.bias is not None else (act.weight, act.bias),
                             lambda: self._head_pointers(gl, act))
        after = (self._enc_cache.key, self._head_cache.key) + tuple(gf._packed.key for gf in gfs)
        return before != after

    def pack_state(self):
        """(keys, buffers) of the lazily packed weight copies as the parameters stand NOW: `keys` is what the
        caches WOULD be keyed on (cheap: version counters and ids, no rebuild, no device work), `buffers` the device
        tensors the caches currently hold.  rollout.GraphedPolicyStep bakes the buffers' addresses into a HIP graph:
        it keeps `buffers` alive and compares `keys` before every replay."""
        enc_t = self._encoder_tensors()                    # (also refreshes self._mods)
        gfs, act = self._mods
        gl = gfs[-1]
        head = (gl.bias, act.weight, act.bias) if gl.bias is not None else (act.weight, act.bias)
        K = _native.PackCache.key_of
        keys = (K(enc_t), K(head)) + tuple(K((gf.weight,)) for gf in gfs)
        bufs = (self._enc_cache.buf, self._head_cache.buf) + tuple(gf._packed.buf for gf in gfs)
        return keys, bufs

    def forward_logits(self, inputTensor, _slot=None):
        """One policy step; returns the logits as ONE tensor [N,B,5] (agent-major, each [n] a
        contiguous [B,5] block) -- what forward() unbinds into the reference's list."""
        if self.training:
            return self._forward_train(inputTensor)
        if self.S is None:
            raise TypeError('addGSO() must be called before forward()')
        prec = self._prec()
        logits = self._forward_eval(inputTensor, prec, _slot)
        if prec == _native.PREC_SPLIT_F16 and self.range_policy == 'strict' and self.range_exceeded():
            # an activation left the f16 range: this call again with the fp32-equivalent arithmetic (a per-call
            # argument of the C entry points: nothing process-wide changes, other streams keep their schedule)
            self._range_flag.zero_()
            logits = self._forward_eval(inputTensor, _native.PREC_FP32, _slot)
        return logits

    def _forward_eval(self, inputTensor, prec=_native.PREC_FP32, slot=None):
        B = inputTensor.shape[0]
        N = self.numAgents
        assert inputTensor.shape[1] >= N
        obs = inputTensor.detach()
        if obs.shape[1] != N:
            obs = obs[:, :N]                      # the reference only visits the first numAgents
        if obs.dtype is not torch.float32 or not obs.is_contiguous():
            obs = obs.contiguous().float()
        S = self.S.detach()                       # [B,E,N,N]
        assert S.shape[0] == B
        Ns = S.shape[2]
        assert Ns >= N                            # Nin <= N zero padding (graphML.py:2464-2469)
        if not S.is_contiguous():
            S = S.contiguous()
        if S.dtype not in (torch.float32, torch.float64):
            S = S.float()
        enc = self.packed_encoder()
        gfs, act = self._mods
        dev = _native.require_gpu(obs, S, gfs[0].weight, act.weight)
        if Ns > gml.MAX_NODES:
            # larger graphs than one workgroup's LDS holds: encoder kernel, then every filter layer as dense
            # exact-fp32 GEMMs (graphML._lsigf_large) and the head as one small library GEMM
            x = self.encode(obs, prec)
            if Ns != N:
                x = torch.cat([x, x.new_zeros(B, Ns - N, x.shape[2])], 1)
            for gf in gfs:
                x = gml._lsigf_large(gf.weight, S, x, gf.bias, True, relu=True)
            out = torch.nn.functional.linear(x[:, :N], act.weight.detach().float(), act.bias.detach().float())
            return out.permute(1, 0, 2).contiguous()
        L = _native.lib()
        s64 = int(S.dtype is torch.float64)
        flag = self._flag(dev).data_ptr() if prec == _native.PREC_SPLIT_F16 else None
        gl = gfs[-1]                               # last graph-filter layer: fused with the head
        gb_p, aw_p, ab_p, _keep = self._head_cache.get(
            (gl.bias, act.weight, act.bias) if gl.bias is not None else (act.weight, act.bias),
            lambda: self._head_pointers(gl, act))
        with _native.device_guard(dev):
            st = _native.stream_ptr(dev)
            if self.L == 1 and Ns == N and gl.F == 128:
                # the planner of the reference's configs: ONE C call (one or two kernels)
                # feature workspace between the two kernels: one per (size, stream) -- several batches of a rollout
                # may be in flight on different streams (rollout.GroupedRollout)
                wkey = (B * N, st.value)
                if self._ws is None or self._ws_key != wkey or self._ws.device != dev:
                    if wkey not in self._ws_all or self._ws_all[wkey].device != dev:
                        if len(self._ws_all) > 16:
                            self._ws_all.clear()
                        self._ws_all[wkey] = torch.empty(B * N, 128, dtype=torch.float32, device=dev)
                    self._ws, self._ws_key = self._ws_all[wkey], wkey
                logits = (torch.empty(N, B, 5, dtype=torch.float32, device=dev) if slot is None
                          else slot.acquire(N, B, dev, st.value))
                rc = L.gnnpp_policy_fwd(obs.data_ptr(), S.data_ptr(), enc.data_ptr(),
                                        gl.packed_taps().data_ptr(), gb_p, aw_p, ab_p,
                                        self._ws.data_ptr(), logits.data_ptr(), B, N, gl.K, self.E,
                                        s64, prec, flag, st)
                _native.check(rc, 'gnnpp_policy_fwd')
                return logits
            # general form: encoder kernel, then one filter kernel per layer (node-major in / out, bias
            # + ReLU fused), the last one with the action head.  A GSO larger than numAgents carries
            # zero features on the extra nodes, whose outputs are dropped.
            x = torch.zeros(B, Ns, 128, dtype=torch.float32, device=dev) if Ns != N else None
            feat = self.encode(obs, prec)
            if x is not None:
                x[:, :N] = feat
            else:
                x = feat
            for l, gf in enumerate(gfs):
                last = l == self.L - 1
                bias = gf.bias.detach().reshape(-1).contiguous().float() if gf.bias is not None else None
                if last and gf.F <= 128:
                    logits = torch.empty(Ns, B, 5, dtype=torch.float32, device=dev)
                    rc = L.gnnpp_filter_head_fwd(_ptr(x), _ptr(S), _ptr(gf.packed_taps()), _ptr(bias),
                                                 aw_p, ab_p, _ptr(logits), B, Ns, gf.G, gf.F, gf.K,
                                                 self.E, s64, prec, flag, st)
                    _native.check(rc, 'gnnpp_filter_head_fwd')
                    return logits[:N] if Ns != N else logits
                y = torch.empty(B, Ns, gf.F, dtype=torch.float32, device=dev)
                rc = L.gnnpp_lsigf_fwd(_ptr(x), _ptr(S), _ptr(gf.packed_taps()), _ptr(bias), _ptr(y),
                                       B, Ns, Ns, gf.G, gf.F, gf.K, self.E, s64, 1, 1, 1, 1, 0, prec, flag, st)
                _native.check(rc, 'gnnpp_lsigf_fwd')
                x = y
        # last layer wider than 128 features: the 5-row head is one small library GEMM
        out = torch.nn.functional.linear(x[:, :N], act.weight.detach().float(), act.bias.detach().float())
        return out.permute(1, 0, 2).contiguous()

    def _forward_train(self, inputTensor):
        """Differentiable train-mode forward with the reference's semantics (decentralplanner.py:278-318).
        Hand-written HIP: the per-agent ConvLayers calls (BatchNorm with THAT call's batch statistics, N
        running-statistics updates per forward) as _EncoderTrainFunction over all agents at once
        (csrc/train_encoder.hip, forward and backward); the graph-filter layers on lsigf_kernel
        (graphML._LSIGFFunction: forward with tap dump, input gradient = the transposed filter, tap gradient on
        gnnpp_gemm_kmajor); compressMLP and the action head forward on gnnpp_linear_fwd (bias + ReLU in the launch),
        their backward products on gnnpp_gemm_kmajor_multi; every weight re-ordering of the step in one
        gnnpp_train_pack launch per weight version; both ReLU backward passes around the filter folded into the
        launches that produce the masked gradients.  No library GEMM and no aten kernel is left in the step (r06).  A
        GSO with more nodes than numAgents (graphML.py:2464-2469 zero-pads the signal) is honoured as in eval mode."""
        if self.S is None:
            raise TypeError('addGSO() must be called before forward()')
        _native.require_gpu(inputTensor, self.S, self.compressMLP[0].weight)
        B, N = inputTensor.shape[0], self.numAgents
        obs = inputTensor.detach()
        if obs.shape[1] != N:
            obs = obs[:, :N]
        obs = obs.contiguous().float()
        tensors, buffers = [], []
        bn0 = self.ConvLayers[_BN_IDX[0]]
        track = all(self.ConvLayers[bi].track_running_stats and self.ConvLayers[bi].momentum is not None
                    for bi in _BN_IDX)
        for ci, bi in zip(_CONV_IDX, _BN_IDX):
            conv, bn = self.ConvLayers[ci], self.ConvLayers[bi]
            tensors += [conv.weight, conv.bias, bn.weight, bn.bias]
            buffers += [bn.running_mean, bn.running_var]
        counters = [self.ConvLayers[bi].num_batches_tracked for bi in _BN_IDX] if track else None   # += N each
        packs = self._train_packs(tensors)                                              # one launch per weight version
        feat = _EncoderTrainFunction.apply(obs, buffers if track else None, counters, float(bn0.momentum or 0.0),
                                           float(bn0.eps), packs[0] if packs else None, *tensors)   # [B,N,128]
        fc = self.compressMLP[0]
        if self.S.shape[0] != B:
            raise _native.GnnppError('addGSO() was given %d graphs, the input has %d samples' % (self.S.shape[0], B))
        Ns = self.S.shape[-1]
        if Ns < N:
            raise _native.GnnppError('the GSO has %d nodes, the planner %d agents' % (Ns, N))
        # The two ReLU backward passes around the graph filter are folded into the launches that produce the gradients
        # they mask (r06): compressMLP's into the filter's input-gradient launch (`fold` bit 1 <-> relu = 2), the
        # filter's own into the action head's dx product (`fold` bit 0 <-> mask_dx) -- when the filter's input / output
        # ARE those tensors (no zero-padded nodes in between) and the filter runs on the LDS-resident kernels.
        direct = Ns == N and Ns <= gml.MAX_NODES
        # (`direct`: the head's and the filter's parameter-gradient products wait for the compress layer's backward launch)
        x = _LinearFunction.apply(feat, fc.weight, fc.bias, 2 if direct else 1, False, 2 if direct else 0)   # [B,N,F]
        if Ns != N:                                # Nin < N: zero signal on the extra nodes (graphML.py:2464-2469)
            x = torch.cat([x, x.new_zeros(B, Ns - N, x.shape[2])], 1)
        # every activation stays node-major [B,N,*] (the layout the filter kernel keeps in LDS): no transposing
        # copy between encoder, filter layers and head; each GFL ReLU runs inside its filter launch
        for l in range(self.L):
            gf = self.GFL[2 * l]
            gf.addGSO(self.S)
            fold = ((2 if l == 0 else 0) | (1 if l == self.L - 1 else 0)) if direct else 0
            x = gf.forward_node_major(x, relu=True, packed=packs[1] if packs else None,
                                      packed_T=packs[2] if packs else None, fold=fold | (4 if direct else 0))  # [B,Ns,F_l]
        if Ns != N:
            x = x[:, :N]                           # ... whose outputs are dropped (index_select, :2471-2476)
        act = self.actionsMLP[0]
        return _LinearFunction.apply(x, act.weight, act.bias, 0, direct, 1 if direct else 0).permute(1, 0, 2)   # [N,B,5]

    def _train_packs(self, conv_tensors):
        """(encoder train pack, filter taps forward, filter taps transposed) of the CURRENT weights by ONE
        gnnpp_train_pack launch per weight version (r05: five pack launches + a transposing copy per step), or None for
        planners the one-launch pack does not cover (several filter layers, the opt-in split-f16 forward whose taps
        need the other regions of the tap buffers): those pack per layer as before."""
        if self.L != 1:
            return None
        gf = self.GFL[0]
        if _native.precision_code(gf.precision) == _native.PREC_SPLIT_F16:
            return None
        ws = [conv_tensors[4 * i] for i in range(5)]
        h = gf.weight
        if any(t.dtype is not torch.float32 or not t.is_contiguous() or not t.is_cuda for t in ws + [h]):
            return None

        def build():
            L = _native.lib()
            dev = h.device
            p = _native.EncoderParams()
            for i in range(5):
                p.conv_w[i] = ws[i].data_ptr()
            enc = torch.empty(L.gnnpp_train_pack_floats(), dtype=torch.float32, device=dev)
            fwd = torch.empty(L.gnnpp_filter_packed_floats(gf.G, gf.F, gf.K, gf.E), dtype=torch.float32, device=dev)
            tr = torch.empty(L.gnnpp_filter_packed_floats(gf.F, gf.G, gf.K, gf.E), dtype=torch.float32, device=dev)
            with _native.device_guard(dev):
                _native.check(L.gnnpp_train_pack(ctypes.byref(p), _ptr(enc), _ptr(h.detach()), _ptr(fwd), _ptr(tr),
                                                 gf.G, gf.F, gf.K, gf.E, _native.stream_ptr(dev)), 'gnnpp_train_pack')
            return enc, fwd, tr
        return self._train_pack_cache.get(ws + [h], build)

    def _forward_train_aten(self, inputTensor):
        """The same train-mode forward on stock aten / MIOpen ops (agents as convolution groups).  NOT
        used by forward(): kept as an independent GPU cross-check of the HIP training kernels for the
        tests (tests/test_gpu_training.py)."""
        import torch.nn.functional as tF
        B, N = inputTensor.shape[0], self.numAgents
        x = inputTensor[:, :N].reshape(B, N * 3, 11, 11)
        for ci, bi in zip(_CONV_IDX, _BN_IDX):
            conv, bn = self.ConvLayers[ci], self.ConvLayers[bi]
            C = conv.out_channels
            x = tF.conv2d(x, conv.weight.repeat(N, 1, 1, 1), conv.bias.repeat(N), stride=1,
                          padding=1, groups=N)
            bmean = torch.zeros(N * C, device=x.device, dtype=x.dtype)
            bvar = torch.ones(N * C, device=x.device, dtype=x.dtype)
            x = tF.batch_norm(x, bmean, bvar, bn.weight.repeat(N), bn.bias.repeat(N), training=True,
                              momentum=1.0, eps=bn.eps)
            x = tF.relu(x)
            if isinstance(self.ConvLayers[bi + 2] if bi + 2 < len(self.ConvLayers) else None,
                          nn.MaxPool2d):
                x = tF.max_pool2d(x, 2)
        feat = x.reshape(B, N, self.numFeatures2Share)
        fc = self.compressMLP[0]
        comp = tF.relu(tF.linear(feat, fc.weight, fc.bias))          # B x N x F
        for l in range(self.L):
            self.GFL[2 * l].addGSO(self.S)
        shared = self.GFL(comp.permute(0, 2, 1).contiguous())
        act = self.actionsMLP[0]
        logits = tF.linear(shared.permute(0, 2, 1), act.weight, act.bias)
        return [logits[:, n] for n in range(N)]

    def forward(self, inputTensor):
        """[B,N,3,11,11] -> python list of N tensors [B,5] (decentralplanner.py:278-318)."""
        if self.training:
            return LogitList(self._forward_train(inputTensor))
        slot = self._out_slot
        slot.views = None
        logits = self.forward_logits(inputTensor, slot)
        views = slot.views                         # the views of `logits` when it came out of the slot (_OutputSlot)
        if views is not None and views[0]._base is logits:
            return list(views)
        return list(logits.unbind(0))

    def decode_actions(self, logits):
        """logits [N,B,5] (from forward_logits) -> int32 [B,N] action ids: argmax of the
        LogSoftmax the simulator applies (utils/multirobotsim_dcenlocal.py:589-591)."""
        N, B = logits.shape[0], logits.shape[1]
        dev = _native.require_gpu(logits)
        acts = torch.empty(B, N, dtype=torch.int32, device=dev)
        with _native.device_guard(dev):
            _native.check(_native.lib().gnnpp_decode_actions(
                _ptr(logits.contiguous()), _ptr(acts), B, N, _native.stream_ptr(dev)),
                'gnnpp_decode_actions')
        return acts
