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
import torch
import torch.distributed as dist
import torch.nn.functional as tF

from . import _native


def policy_loss(predict, batch_target):
    """predict: list of N tensors [B,5]; batch_target [B,N,5] one-hot expert actions.
    agents/decentralplannerlocal.py:296-312: loss = (1/N) sum_n CrossEntropy(predict[n], argmax target[:, n]).
    Every agent's CrossEntropy is a mean over the same B samples, so the mean over agents of the means is
    the mean over all N*B rows: ONE batched cross-entropy instead of N (10 x fewer launches per step)."""
    logits = torch.stack(predict, dim=0)                                   # [N,B,5]
    labels = batch_target.permute(1, 0, 2).argmax(-1)                      # [N,B]: torch.max(.,1)[1], first maximum
    return tF.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1))


def train_step(model, optimizer, batch_input, batch_target, batch_GSO, dp=None):
    """zero_grad -> addGSO -> forward -> loss -> backward -> (gradient all-reduce) -> step."""
    optimizer.zero_grad()
    model.addGSO(batch_GSO)
    predict = model(batch_input)
    loss = policy_loss(predict, batch_target)
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
    def __init__(self, module, group=None, broadcast_from=0):
        self.module = module
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.bucket = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.bucket[off:off + p.numel()].view_as(p))
            off += p.numel()
        if self.world > 1:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t, src=broadcast_from, group=group)

    def reduce_gradients(self):
        """Average the gradients of all ranks: pack -> one all_reduce -> unpack."""
        if self.world == 1:
            return
        with torch.no_grad():
            for p, v in zip(self.params, self.views):
                if p.grad is None:
                    v.zero_()
                else:
                    v.copy_(p.grad)
            dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM, group=self.group)
            self.bucket.mul_(1.0 / self.world)
            for p, v in zip(self.params, self.views):
                if p.grad is None:
                    p.grad = v.clone()
                else:
                    p.grad.copy_(v)
