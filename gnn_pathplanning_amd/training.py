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


def policy_loss(predict, batch_target):
    """predict: list of N tensors [B,5]; batch_target [B,N,5] one-hot expert actions.
    agents/decentralplannerlocal.py:296-312."""
    tgt = batch_target.permute(1, 0, 2)
    loss = 0
    for n in range(len(predict)):
        loss = loss + tF.cross_entropy(predict[n], torch.max(tgt[n], 1)[1])
    return loss / len(predict)


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
