"""Batch/episode sharding across the GPUs of one node.

The policy forward shards only over the batch dimension: graphs (rollout instances) are independent,
weights are replicated, and agents of one graph are never split.  One process per GPU; there is NO
collective on the data path.  The only collectives are the ones the measurement needs (barrier,
max-over-ranks time, sum of processed units); they run over whatever backend the process group uses
(`nccl` = RCCL on the GPU box, `gloo` in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous slice [lo, hi) of `total` independent units owned by `rank`; sizes differ by at
    most one and cover [0, total) exactly."""
    assert 0 <= rank < world and total >= 0
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(tensors, rank, world):
    """Slice every tensor of a batch-major tuple to this rank's shard (strong-scaling rollouts)."""
    lo, hi = shard_range(tensors[0].shape[0], rank, world)
    return tuple(t[lo:hi] for t in tensors)


def aggregate_throughput(units_local, elapsed_local, device=None, group=None):
    """Whole-job throughput: (sum over ranks of units) / (max over ranks of elapsed seconds).
    Returns (throughput, total_units, max_elapsed).  Works without an initialised group (1 GPU)."""
    if not (dist.is_available() and dist.is_initialized()):
        return units_local / elapsed_local, units_local, elapsed_local
    t = torch.tensor([elapsed_local], dtype=torch.float64, device=device)
    u = torch.tensor([float(units_local)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(u, op=dist.ReduceOp.SUM, group=group)
    return u.item() / t.item(), u.item(), t.item()


def device_identity(device):
    """A string that names the PHYSICAL GPU behind `device`: uuid when this torch exposes it, else PCI bus id, else
    the ordinal.  Two ranks that were meant to own different GPUs must report different identities."""
    if device is None or device.type != 'cuda':
        return 'cpu'
    props = torch.cuda.get_device_properties(device)
    for attr in ('uuid', 'pci_bus_id'):
        v = getattr(props, attr, None)
        if v not in (None, ''):
            extra = getattr(props, 'pci_device_id', '')
            return '%s:%s:%s' % (attr, v, extra)
    return 'ordinal:%d' % (device.index if device.index is not None else torch.cuda.current_device())


def gather_rank_devices(device, group=None):
    """[identity of rank 0's GPU, rank 1's, ...] on every rank (all_gather of strings; one entry without a group)."""
    me = device_identity(device)
    if not (dist.is_available() and dist.is_initialized()):
        return [me]
    world = dist.get_world_size(group)
    out = [None] * world
    try:
        dist.all_gather_object(out, me, group=group)
    except Exception as e:                                 # reporting must never take the measurement down
        out = ['unknown (%s)' % type(e).__name__] * world
        out[dist.get_rank(group)] = me
    return out
