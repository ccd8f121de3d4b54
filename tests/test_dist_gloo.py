"""CPU, world_size 2 over gloo: the N>1 launch path of bench.py (sharding + aggregation).
No GPU and no data-path collective is involved; the GPU box runs the same code over nccl/RCCL."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gnn_pathplanning_amd.sharding import aggregate_throughput, gather_rank_devices, shard_batch, shard_range


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        B = 7
        obs = torch.arange(B * 3, dtype=torch.float32).reshape(B, 3)
        gso = torch.arange(B, dtype=torch.float32)
        o, g = shard_batch((obs, gso), rank, world)
        dist.barrier()
        # rank 1 is "slower": whole-job time is the max, units are summed
        thr, units, t = aggregate_throughput(o.shape[0] * 10, 1.0 + rank)
        gathered = [None] * world
        dist.all_gather_object(gathered, g.tolist())
        devs = gather_rank_devices(torch.device('cpu'))
        assert devs == ['cpu', 'cpu']                    # one entry per rank, same list on every rank
        q.put((rank, thr, units, t, gathered))
    finally:
        dist.destroy_process_group()


def test_shard_range_covers_exactly():
    for total in (0, 1, 5, 8, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_sharding_and_aggregation():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, thr, units, t, gathered in res:
        assert units == 70.0 and t == 2.0 and abs(thr - 35.0) < 1e-9
        assert gathered == [[0.0, 1.0, 2.0, 3.0], [4.0, 5.0, 6.0]]      # disjoint, complete, ordered


class _SinkScale(torch.autograd.Function):
    """y = x * w (elementwise, w a parameter) whose backward writes dw where _native.grad_out says -- what the HIP
    backward entry points of the package do with their parameter gradients."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        ctx.ptr = w.data_ptr()
        return x * w

    @staticmethod
    def backward(ctx, dy):
        from gnn_pathplanning_amd import _native
        x, w = ctx.saved_tensors
        dw = _native.grad_out(ctx.ptr, w.shape, w.device)
        torch.sum(dy * x, dim=0, out=dw)
        return dy * w, dw


def _sink_check(rank, world):
    """gradients are BORN in FlatBucketDP's bucket (gradient sinks), so reduce_gradients() is one
    all-reduce + one scale with NO per-parameter copies; accumulation steps and zero_grad(set_to_none=False) -- where
    handing out the slice again would double the gradient -- stay correct."""
    import torch.nn as nn
    from gnn_pathplanning_amd.training import FlatBucketDP

    class M(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Parameter(torch.arange(3.0) + 1)
            self.b = nn.Parameter(torch.ones(3))

        def forward(self, x):
            return _SinkScale.apply(_SinkScale.apply(x, self.a), self.b)
    m = M()
    dp = FlatBucketDP(m)
    x = torch.full((4, 3), float(rank + 1))
    m(x).sum().backward()
    ok = all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(dp.params, dp.views))      # born in the bucket
    dp.reduce_gradients()
    ok = ok and dp.last_reduce_copies == 0
    # d/da sum(x a b) = sum_rows x * b = 4 (rank + 1): mean over ranks 1, 2 = 6; d/db = 4 (rank + 1) a -> 6 a
    ok = ok and torch.allclose(m.a.grad, torch.full((3,), 6.0)) and torch.allclose(m.b.grad, 6.0 * (torch.arange(3.0) + 1))
    # a second backward WITHOUT zero_grad accumulates (the sink is not handed out again: no doubling)
    before = m.a.grad.clone()
    m(x).sum().backward()
    ok = ok and torch.allclose(m.a.grad, before + 4.0 * (rank + 1))
    # zero_grad(set_to_none=False) keeps the slices as `.grad`: the next step is still right (and needs no copies)
    for p in m.parameters():
        p.grad.zero_()
    m(x).sum().backward()
    dp.reduce_gradients()
    ok = ok and dp.last_reduce_copies == 0 and torch.allclose(m.a.grad, torch.full((3,), 6.0))
    # set_to_none: fresh views again
    for p in m.parameters():
        p.grad = None
    m(x).sum().backward()
    dp.reduce_gradients()
    ok = ok and dp.last_reduce_copies == 0 and torch.allclose(m.b.grad, 6.0 * (torch.arange(3.0) + 1))
    # a parameter consumed by TWO backward nodes of one pass.  `.grad` stays None until AccumulateGrad
    # runs, so the slice used to be handed out twice: the second kernel overwrote the first one's result and autograd
    # then added the two aliased views (2 x the last gradient instead of the sum).  The slice is handed out once at a time.
    for p in m.parameters():
        p.grad = None
    x2 = torch.full((4, 3), 10.0 * (rank + 1))
    (_SinkScale.apply(x, m.a).sum() + _SinkScale.apply(x2, m.a).sum()).backward()
    ok = ok and torch.allclose(m.a.grad, torch.full((3,), 44.0 * (rank + 1)))
    m.a.grad = None
    g1, = torch.autograd.grad(_SinkScale.apply(x, m.a).sum(), [m.a])        # two grad() calls, results both alive
    g2, = torch.autograd.grad(_SinkScale.apply(x2, m.a).sum(), [m.a])
    ok = ok and torch.allclose(g1, torch.full((3,), 4.0 * (rank + 1))) and torch.allclose(g2, torch.full((3,), 40.0 * (rank + 1)))
    del g1, g2
    # inside no_grad_sinks() (what the registered custom op gnnpp::lsigf_backward runs in) nothing comes out of the bucket
    from gnn_pathplanning_amd import _native
    with _native.no_grad_sinks():
        g3 = _native.grad_out(m.a.data_ptr(), m.a.shape, m.a.device)
    lo, hi = dp.bucket.data_ptr(), dp.bucket.data_ptr() + dp.bucket.numel() * 4
    ok = ok and not (lo <= g3.data_ptr() < hi)
    g4 = _native.grad_out(m.a.data_ptr(), m.a.shape, m.a.device)
    ok = ok and lo <= g4.data_ptr() < hi
    del g3, g4
    dp.close()
    m(x).sum().backward()                                   # sinks forgotten: plain tensors again, still correct
    # a FlatBucketDP that dies without close() takes its sinks with it (they held the bucket strongly)
    m2 = M()
    dp2 = FlatBucketDP(m2)
    n_sinks = len(_native._grad_sinks)
    del dp2
    import gc
    gc.collect()
    ok = ok and len(_native._grad_sinks) == n_sinks - 2
    return bool(ok)


def _dp_worker(rank, world, port, q):
    import torch.nn as nn
    from gnn_pathplanning_amd.training import FlatBucketDP
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                       # different init per rank on purpose
        model = nn.Sequential(nn.Linear(6, 5), nn.BatchNorm1d(5), nn.Linear(5, 3))
        dp = FlatBucketDP(model)                            # broadcast params + buffers from rank 0
        w0 = model[0].weight.detach().clone()
        x = torch.full((4, 6), float(rank + 1))
        model(x).sum().backward()
        local = [p.grad.clone() for p in model.parameters()]
        dp.reduce_gradients()
        gathered = [None] * world
        dist.all_gather_object(gathered, [g.tolist() for g in local])
        mean = [(torch.tensor(a) + torch.tensor(b)) / 2 for a, b in zip(*gathered)]
        ok = all(torch.allclose(p.grad, m, atol=1e-6) for p, m in zip(model.parameters(), mean))
        # plain torch modules do not write through the sinks: every gradient was copied in, and `.grad` now IS the slice
        ok = ok and dp.last_reduce_copies == len(dp.params)
        ok = ok and all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(dp.params, dp.views))
        ok = ok and _sink_check(rank, world)
        q.put((rank, ok, w0.tolist(), dp.bucket.numel()))
    finally:
        dist.destroy_process_group()


def test_flat_bucket_dp_two_ranks():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)                             # gradients are the cross-rank mean
    assert res[0][2] == res[1][2]                             # parameters were broadcast from rank 0
    assert res[0][3] == 6 * 5 + 5 + 5 + 5 + 5 * 3 + 3         # ONE bucket holds every gradient


def test_single_process_aggregation_without_group():
    thr, units, t = aggregate_throughput(100, 0.5)
    assert (thr, units, t) == (200.0, 100, 0.5)


def test_rank_devices_without_group_and_scale_checker():
    """tools/check_scale.py rejects a scaling session whose N-GPU line did not use N ranks on N distinct GPUs, or
    whose 1-GPU line disagrees with the committed bench line."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import check_scale
    assert gather_rank_devices(torch.device('cpu')) == ['cpu']
    metric = 'agent-steps/sec (policy fwd)'
    ok = [{'n_gpus': 1, 'ranks_in_group': 1, 'rank_devices': ['uuid:a'], 'value': 100.0, 'metric': metric},
          {'n_gpus': 2, 'ranks_in_group': 2, 'rank_devices': ['uuid:a', 'uuid:b'], 'value': 198.0, 'metric': metric}]
    ref = {'value': 102.0, 'metric': metric}
    errs, vals = check_scale.check(ok, ref)
    assert errs == [] and vals == {1: 100.0, 2: 198.0}
    shared = [ok[0], dict(ok[1], rank_devices=['uuid:a', 'uuid:a'])]
    assert any('share physical devices' in e for e in check_scale.check(shared, ref)[0])
    assert check_scale.check(shared, ref, allow_shared=True)[0] == []
    assert any('ranks_in_group' in e for e in check_scale.check([ok[0], dict(ok[1], ranks_in_group=1)], ref)[0])
    assert any('within 5' in e for e in check_scale.check(ok, {'value': 120.0, 'metric': metric})[0])
    assert any('drops' in e for e in check_scale.check([ok[0], dict(ok[1], value=90.0)], ref)[0])
    assert any('rank_devices reported' in e for e in check_scale.check([dict(ok[0], rank_devices=[])], None)[0])


def _rank_local_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        from gnn_pathplanning_amd.sharding import aggregate_throughput
        refused = None
        if rank == 0:                                           # what bench.py's rank 0 does after the timed regions
            with bench.rank_local_block(active=True):
                try:
                    aggregate_throughput(10, 1.0)               # (r04's hang: a rank-0-only all_reduce)
                    refused = False
                except RuntimeError as e:
                    refused = 'rank-local block' in str(e)
        thr, units, t = aggregate_throughput(10.0 * (rank + 1), 1.0 + rank)      # restored: real collectives work again
        dist.barrier()
        q.put((rank, refused, thr, units, t))
    finally:
        dist.destroy_process_group()


def test_rank_local_block_refuses_unmatched_collectives():
    """r05: the 2-rank check of bench.py found that a rank-0-only secondary record (the C4 training shard, r04) started an
    all-reduce the other rank never matched -- the run hung.  Fixed at the source (tools/train_bench.measure), and
    bench.py now runs everything behind its timed regions inside `rank_local_block`, where a collective RAISES; the
    collectives work again behind the block."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_local_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] is True and res[1][1] is None
    for _, _, thr, units, t in res:
        assert units == 30.0 and t == 2.0 and abs(thr - 15.0) < 1e-9


def test_bench_gpus_n_without_a_launcher_runs_n_ranks():
    """`python bench.py --gpus 2` with no torch.distributed.run around it used to run ONE rank and
    print `n_gpus: 1`.  bench.py now launches the N ranks itself (bench.self_launch); `--launch-check` is the GPU-less
    body of that path: every rank joins a gloo group, rank 0 prints the group's shape."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--launch-check'], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith('{')][-1]
    d = json.loads(line)
    assert d == {'launch_check': True, 'n_gpus': 2, 'ranks_in_group': 2}
    # a launcher whose group does not match --gpus is refused instead of printing a mislabelled line
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--launch-check'],
                       env=dict(env, WORLD_SIZE='1', RANK='0'), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=300, cwd=root)
    assert r.returncode != 0
    # without --launch-check and without GPUs the launcher refuses before starting any rank
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2'], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300, cwd=root)
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        assert r.returncode != 0 and b'refusing' in r.stderr
