"""Training step and data-parallel gradient exchange for BASELINE config 4.

  train_step()    one optimisation step exactly as agents/decentralplannerlocal.py:287-317:
                  loss = mean over agents of CrossEntropy(predict[n], argmax(target[:, n])).
  FlatBucketDP    data parallelism over the batch dimension, one process per GPU.  The model has
                  206 501 fp32 parameters = 826 KB: the exchange is latency-bound, so ALL gradients
                  travel in ONE flat bucket through ONE all-reduce per step (RCCL over xGMI on the
                  GPU box, gloo in the CPU tests) instead of per-parameter or ring-tuned buckets.
                  BatchNorm running statistics are broadcast from rank 0 at construction and kept
                  local afterwards (each rank sees its own shard, like torch DDP without SyncBN).
The reference has no distributed code at all; this is the build's addition.
"""
import ctypes
import weakref

import torch
import torch.distributed as dist
import torch.nn.functional as tF

from . import _native


class _PolicyLossFunction(torch.autograd.Function):
    """gnnpp_policy_loss: the loss and d loss / d logits in one launch (one workgroup, fixed-order sums)."""

    @staticmethod
    def forward(ctx, logits, target):
        loss, dlogits = _policy_loss_and_grad(logits, target)
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g, None


def _policy_loss_and_grad(logits, target):
    """(loss, d loss / d logits) of logits [N,B,C] by ONE gnnpp_policy_loss launch.  The train-mode forward
    hands out [N,B,C] as a view of a contiguous [B,N,C] tensor: the kernel reads (and writes the gradient in)
    that layout directly."""
    N, B, C = logits.shape
    lg = logits.detach()
    sample_major = lg.permute(1, 0, 2).is_contiguous() and not lg.is_contiguous()
    if sample_major:
        lg = lg.permute(1, 0, 2)                                           # the [B,N,C] storage itself
    if lg.dtype is not torch.float32 or not lg.is_contiguous():
        lg = lg.contiguous().float()
    tg = target.detach().contiguous().float()
    dev = _native.require_gpu(lg, tg)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    dlogits = torch.empty_like(lg)
    with _native.device_guard(dev):
        _native.check(_native.lib().gnnpp_policy_loss(lg.data_ptr(), tg.data_ptr(), loss.data_ptr(),
                                                      dlogits.data_ptr(), B, N, C, int(sample_major),
                                                      _native.stream_ptr(dev)), 'gnnpp_policy_loss')
    return loss, (dlogits.permute(1, 0, 2) if sample_major else dlogits)


def policy_loss(predict, batch_target):
    """predict: list of N tensors [B,5]; batch_target [B,N,5] one-hot expert actions.
    agents/decentralplannerlocal.py:296-312: loss = (1/N) sum_n CrossEntropy(predict[n], argmax target[:, n]).
    Every agent's CrossEntropy is a mean over the same B samples, so the mean over agents of the means is
    the mean over all N*B rows: ONE batched cross-entropy instead of N (10 x fewer launches per step)."""
    logits = getattr(predict, 'stacked', None)                             # [N,B,5] (decentralplanner.LogitList)
    if logits is None:
        logits = torch.stack(list(predict), dim=0)
    labels = batch_target.permute(1, 0, 2).argmax(-1)                      # [N,B]: torch.max(.,1)[1], first maximum
    return tF.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1))


def policy_loss_fused(predict, batch_target):
    """policy_loss() on one HIP launch (forward and backward together); same value, same gradient."""
    logits = getattr(predict, 'stacked', None)
    if logits is None:
        logits = torch.stack(list(predict), dim=0)
    return _PolicyLossFunction.apply(logits, batch_target)


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) (amsgrad=False) on gnnpp_adam_step: ONE launch
    (the device-side step counter is advanced by the launch's last workgroup) for up to 32 parameter tensors instead of the
    eight multi-tensor passes of the stock implementation; graph-capturable (nothing is read on the host).
    The reference builds optim.Adam(lr, weight_decay) (agents/decentralplannerlocal.py:77-79); the update
    rule is the same, the results agree to rounding (tests/test_gpu_training.py)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def load_state_dict(self, state_dict):
        """torch's loader replaces the group dicts and every state tensor: the cached raw-pointer tables die here."""
        self.__dict__.pop('_gnnpp_tables', None)
        return super().load_state_dict(state_dict)

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__.pop('_gnnpp_tables', None)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _native.lib()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group['params'] if p.grad is not None]
            if not ps:
                continue
            dev = _native.require_gpu(*ps)
            # The device-side step counter is optimizer STATE: keyed by the group's INDEX (load_state_dict()
            # rebuilds the group dicts, so id(group) does not survive a checkpoint round trip) and therefore saved
            # and restored by state_dict() / load_state_dict() next to the moments.
            st = self.state.setdefault('gnnpp_group_%d' % gi, {})
            if 'counter' not in st:
                st['counter'] = torch.zeros(8, dtype=torch.float32, device=dev)
            elif st['counter'].device != dev or st['counter'].dtype is not torch.float32:
                st['counter'] = st['counter'].to(device=dev, dtype=torch.float32)      # (loaded with map_location)
            if st['counter'].numel() < 8:                   # a checkpoint written before ABI v330 (three floats)
                st['counter'] = torch.cat([st['counter'].reshape(-1)[:3], st['counter'].new_zeros(5)])
            for p in ps:
                s = self.state[p]
                if 'exp_avg' not in s:
                    assert p.dtype is torch.float32 and p.is_contiguous()
                    s['exp_avg'] = torch.zeros_like(p)
                    s['exp_avg_sq'] = torch.zeros_like(p)
                elif not s['exp_avg'].is_contiguous() or not s['exp_avg_sq'].is_contiguous():
                    s['exp_avg'], s['exp_avg_sq'] = s['exp_avg'].contiguous(), s['exp_avg_sq'].contiguous()
            b1, b2 = group['betas']
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
            # the pointer tables are rebuilt only when a pointer moved (the caching allocator hands the
            # gradients the same blocks step after step; filling a ctypes struct costs more than the launch);
            # the key covers EVERY pointer a table holds -- parameters, gradients and both moments (a
            # load_state_dict() replaces the moment tensors)
            key = tuple(p.data_ptr() for p in ps) + tuple(g.data_ptr() for g in grads) + \
                tuple(self.state[p]['exp_avg'].data_ptr() for p in ps) + \
                tuple(self.state[p]['exp_avg_sq'].data_ptr() for p in ps)
            cache = self.__dict__.setdefault('_gnnpp_tables', {}).setdefault(gi, {})   # (not optimizer state)
            if cache.get('key') != key:
                tables = []
                for i0 in range(0, len(ps), 32):
                    tb = _native.AdamTensors()
                    for i, (p, g) in enumerate(zip(ps[i0:i0 + 32], grads[i0:i0 + 32])):
                        tb.p[i], tb.g[i] = p.data_ptr(), g.data_ptr()
                        tb.m[i], tb.v[i] = self.state[p]['exp_avg'].data_ptr(), self.state[p]['exp_avg_sq'].data_ptr()
                        tb.numel[i] = p.numel()
                    tb.count = min(32, len(ps) - i0)
                    tables.append(tb)
                cache['key'], cache['tables'] = key, tables
            with _native.device_guard(dev):
                for k, tb in enumerate(cache['tables']):
                    _native.check(L.gnnpp_adam_step(ctypes.byref(tb), st['counter'].data_ptr(), group['lr'], b1, b2,
                                                    group['eps'], group['weight_decay'], int(k == 0),
                                                    _native.stream_ptr(dev)), 'gnnpp_adam_step')
        # the parameters changed behind torch's version counters: packed / BN-folded copies are stale
        _native.invalidate_packs()
        return loss


def train_step(model, optimizer, batch_input, batch_target, batch_GSO, dp=None):
    """zero_grad -> addGSO -> forward -> loss -> backward -> (gradient all-reduce) -> step."""
    optimizer.zero_grad()
    model.addGSO(batch_GSO)
    predict = model(batch_input)
    stacked = getattr(predict, 'stacked', None)
    if stacked is not None and stacked.is_cuda:
        # loss and d loss / d logits from one launch; backward starts at the logits (what loss.backward() does,
        # minus the ones_like fill and the multiplication by it)
        loss, dlogits = _policy_loss_and_grad(stacked, batch_target)
        # (this step owns its backward pass -- no hooks read a gradient before the pass has ended --, so the products that
        # only yield parameter gradients may wait for one later launch of the pass: _native.allow_deferred_gemms)
        with _native.allow_deferred_gemms():
            stacked.backward(dlogits)
    else:
        loss = policy_loss(predict, batch_target)
        with _native.allow_deferred_gemms():
            loss.backward()
    if dp is not None:
        dp.reduce_gradients()
    optimizer.step()
    return loss.detach()


class GraphedTrainStep:
    """The whole optimisation step (forward, loss, backward, optimizer) captured once in a HIP graph
    and replayed: at the reference's batch size (64) the step is launch-bound (~250 kernels), and a
    replay costs one launch.  The libgnnpp kernels are enqueued on torch's capture stream through
    the C ABI like any other launch, so they are captured too (including the re-packing of the
    filter taps after each weight update).  Inputs are copied into static buffers before replay.

        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, capturable=True)
        step = GraphedTrainStep(model, opt, batch_input, batch_target, batch_GSO)
        loss = step(batch_input, batch_target, batch_GSO)          # device tensor, no sync

    dp (FlatBucketDP): the one flat-bucket gradient all-reduce is enqueued on the capture stream between
    backward and the optimizer like every other launch, so RCCL's collective becomes a node of the graph
    (RCCL, like NCCL, supports stream capture); every rank must build and replay its graph in step."""

    def __init__(self, model, optimizer, batch_input, batch_target, batch_GSO, warmup=3, dp=None):
        self.inp = batch_input.clone()
        self.tgt = batch_target.clone()
        self.gso = batch_GSO.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                                 # warm allocator, packs, MIOpen
                train_step(model, optimizer, self.inp, self.tgt, self.gso, dp)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = train_step(model, optimizer, self.inp, self.tgt, self.gso, dp)

    def __call__(self, batch_input, batch_target, batch_GSO):
        self.inp.copy_(batch_input)
        self.tgt.copy_(batch_target)
        self.gso.copy_(batch_GSO)
        self.graph.replay()
        # the replayed optimizer step changed the parameters behind torch's version counters: every
        # packed / BN-folded copy (eval forward, rollouts) must be rebuilt before its next use
        _native.invalidate_packs()
        return self.loss


class FlatBucketDP:
    """Data parallelism over the batch dimension with ONE flat gradient bucket (module docstring).

    The gradients LIVE in the bucket: every parameter's slice is registered as its gradient sink
    (_native.register_grad_sink), the backward kernels of this package write d loss / d parameter straight into a
    view of that slice and autograd adopts the view as `.grad`.  reduce_gradients() is then exactly one all-reduce and
    one scale -- no packing, no unpacking (r04: 2 x 26 per-parameter copy launches around the collective, about the
    collective's own cost again on a 0.5 ms step).  Gradients that did not arrive through a sink (a parameter of a
    plain torch module, an accumulation step: _native.grad_out) are copied into their slice first, and `.grad` is
    re-pointed at the slice afterwards; `last_reduce_copies` counts those copies (0 on the planner's training step)."""

    def __init__(self, module, group=None, broadcast_from=0):
        self.module = module
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.bucket = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        self.views, self.offsets, off = [], [], 0
        for p in self.params:
            self.views.append(self.bucket[off:off + p.numel()].view_as(p))
            self.offsets.append(off)
            if p.dtype is torch.float32 and p.is_contiguous():
                _native.register_grad_sink(p, self.bucket, off)
            off += p.numel()
        self.last_reduce_copies = 0
        # the sinks hold the bucket strongly: forget them when this object dies without close()
        self._finalizer = weakref.finalize(self, _native.unregister_grad_sinks, self.bucket)
        if self.world > 1:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t, src=broadcast_from, group=group)

    def close(self):
        """Forget the gradient sinks (the bucket stays alive as long as a `.grad` views it)."""
        self._finalizer()

    def _in_bucket(self, g, v):
        return g is not None and g.data_ptr() == v.data_ptr() and g.stride() == v.stride() and g.dtype is v.dtype

    def reduce_gradients(self):
        """Average the gradients of all ranks: ONE all_reduce of the bucket + one scale.  Launches: 2 when every
        gradient was born in the bucket (+ one copy / zero-fill per gradient that was not)."""
        if self.world == 1:
            return
        with torch.no_grad():
            copies = 0
            for p, v in zip(self.params, self.views):
                g = p.grad
                if self._in_bucket(g, v):
                    continue
                copies += 1
                if g is None:
                    v.zero_()
                else:
                    v.copy_(g)
            self.last_reduce_copies = copies
            dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM, group=self.group)
            self.bucket.mul_(1.0 / self.world)
            if copies:
                for p, v, off in zip(self.params, self.views, self.offsets):
                    if not self._in_bucket(p.grad, v):
                        p.grad = self.bucket[off:off + p.numel()].view_as(p)     # (a fresh view: `.grad` owns it alone)
