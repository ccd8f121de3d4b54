"""Training throughput for BASELINE config 4 (dcp_onlineExpert: 10 agents, K=3, batch 64 per GPU,
Adam lr 1e-3 wd 1e-5), 1 GPU or data parallel:
    python tools/train_bench.py --steps 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        tools/train_bench.py --steps 50
Prints one JSON line on rank 0 (agent-steps/s of full train steps, all ranks).  With --cpu-seconds S (rank 0,
1 GPU) the line carries `cpu_baseline`: the SAME optimisation step -- oracle.policy_forward(training=True) +
policy_loss + autograd backward + torch.optim.Adam, i.e. what agents/decentralplannerlocal.py:301-317 runs on the
reference's CPU path -- timed on this box's host cores on a bounded sample.

`measure()` and `cpu_baseline()` are also what bench.py's `c4_shard` record calls (the per-GPU shard of config 4)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(dev, batch=64, steps=50, warmup=5, graph=False, adam='fused', rank=0, world=1, agents=10, taps=3):
    """Wall-clock seconds per optimisation step (max over ranks) of `batch` graphs of `agents` agents on `dev`;
    returns (seconds_per_step, final_loss)."""
    import torch.distributed as dist
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from gnn_pathplanning_amd.sharding import aggregate_throughput
    from gnn_pathplanning_amd.training import FlatBucketDP, FusedAdam, train_step
    from oracle import policy_oracle as orc                   # synthetic inputs only

    class Cfg:
        num_agents, nGraphFilterTaps, device = agents, taps, dev
    torch.manual_seed(1337)
    net = DecentralPlannerNet(Cfg()).to(dev).train()
    dp = FlatBucketDP(net) if world > 1 else None
    if adam == 'torch':
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5, capturable=graph)
    else:                                                     # the same update on one HIP launch
        opt = FusedAdam(net.parameters(), lr=1e-3, weight_decay=1e-5)
    B, N = batch, agents
    obs = orc.synth_obs(B, N, seed=1337 + rank).to(dev)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, 20, seed=1337 + rank)).float().to(dev)
    g = torch.Generator().manual_seed(rank)
    tgt = torch.nn.functional.one_hot(torch.randint(0, 5, (B, N), generator=g), 5).float().to(dev)
    step = lambda: train_step(net, opt, obs, tgt, S, dp)      # noqa: E731
    if graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                loss = step()
        torch.cuda.current_stream().wait_stream(side)
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            static_loss = step()

        def step():                                           # noqa: F811
            cg.replay()
            return static_loss
    for _ in range(warmup):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:                                             # (world == 1 inside a multi-rank process -- bench.py's rank-0-only
        _, _, el = aggregate_throughput(B * N * steps, el, device=dev)      # shard record -- must not start a collective)
    return el / steps, float(loss.item())


def cpu_baseline(batch=64, seconds=8.0, agents=10, taps=3, threads=None):
    """The reference's training step on the host: train-mode forward of the CPU oracle (per-agent encoder loop with
    batch-statistics BatchNorm, cat-based LSIGF), mean cross-entropy, autograd backward, torch.optim.Adam(lr 1e-3,
    wd 1e-5) -- agents/decentralplannerlocal.py:301-317.  Median of >= 3 steps within `seconds`."""
    from oracle import policy_oracle as orc
    old_threads = torch.get_num_threads()
    sd = orc.init_state_dict(taps, seed=1337, randomize_bn_stats=False)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if v.dtype == torch.float32 and 'running' not in k}
    sd.update(params)
    opt = torch.optim.Adam(list(params.values()), lr=1e-3, weight_decay=1e-5)
    obs = orc.synth_obs(batch, agents, seed=1337)
    S = torch.from_numpy(orc.synth_gso_geometric(batch, agents, 20, seed=1337)).float()
    g = torch.Generator().manual_seed(0)
    tgt = torch.nn.functional.one_hot(torch.randint(0, 5, (batch, agents), generator=g), 5).float()

    def step():
        opt.zero_grad()
        loss = orc.policy_loss(orc.policy_forward(sd, S, obs, training=True), tgt)
        loss.backward()
        opt.step()
        return loss
    # torch's default thread count (= every logical core of the GPU box's host) is catastrophically oversubscribed for
    # these small per-agent convolutions (measured: 508 ms per step on 128 threads, 50 ms on 8): pick the fastest
    if not threads:
        threads, best = 1, float('inf')
        for t in (1, 2, 4, 8, 16, 32, 64):
            if t > (os.cpu_count() or 1):
                break
            torch.set_num_threads(t)
            step()
            t0 = time.perf_counter()
            step()
            dt = time.perf_counter() - t0
            if dt < best:
                threads, best = t, dt
            elif dt > 2.0 * best:
                break
    torch.set_num_threads(threads)
    step()
    times, t_begin = [], time.perf_counter()
    while (time.perf_counter() - t_begin < seconds or len(times) < 3) and len(times) < 200:
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    torch.set_num_threads(old_threads)
    return {'value': batch * agents / med, 'unit': 'agent-steps/s (training step)', 'ms_per_step': 1e3 * med,
            'cores': threads, 'kind': 'port',
            'sample': '%d optimisation steps (median) of the same %d x %d batch: oracle train-mode forward + loss + '
                      'autograd backward + torch.optim.Adam, ~%.0f s of host time' % (len(times), batch, agents,
                                                                                   sum(times))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--graph', action='store_true', help='capture the train step in a HIP graph')
    ap.add_argument('--adam', choices=('fused', 'torch'), default='fused',
                    help='optimizer: training.FusedAdam (gnnpp_adam_step) or torch.optim.Adam')
    ap.add_argument('--cpu-seconds', type=float, default=0.0,
                    help='> 0: also time the CPU oracle\'s training step on the host (rank 0, 1 GPU) for this long')
    ap.add_argument('--no-fork', action='store_true',
                    help='GNNPP_TUNE_TRAIN_FORK = 0: the backward pass of the encoder on ONE stream (A/B of the r05 fork)')
    ap.add_argument('--fork', action='store_true', help='GNNPP_TUNE_TRAIN_FORK = 2: fork whatever the batch size')
    ap.add_argument('--wgrad-wgs', type=int, default=0,
                    help='GNNPP_TUNE_TRAIN_WGRAD_WGS: workgroups per layer of the weight-gradient kernel (default 320)')
    ap.add_argument('--wgrad-per-layer', action='store_true',
                    help='GNNPP_TUNE_TRAIN_WGRAD_MERGED = 0: one weight-gradient launch per layer inside the chain (r05)')
    ap.add_argument('--dist-backend', default='nccl',
                    help='nccl (= RCCL, default); gloo only to exercise the data-parallel code path on a box with '
                         'fewer GPUs than ranks (with GNNPP_BENCH_DEVICE=0; eager mode only)')
    args = ap.parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('GNNPP_BENCH_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)
    from gnn_pathplanning_amd.sharding import gather_rank_devices
    from gnn_pathplanning_amd import _native
    if args.wgrad_per_layer:
        assert _native.lib().gnnpp_set_tuning(18, 0) == 0
    if args.wgrad_wgs:
        assert _native.lib().gnnpp_set_tuning(17, args.wgrad_wgs) == 0
    if args.no_fork or args.fork:
        assert _native.lib().gnnpp_set_tuning(15, 0 if args.no_fork else 2) == 0
    B, N = args.batch, 10
    per_step, loss = measure(dev, B, args.steps, args.warmup, args.graph, args.adam, rank, world)
    rank_devices = gather_rank_devices(dev)
    if rank == 0:
        line = {'metric': 'training agent-steps/s (fwd+bwd+Adam, config 4)', 'value': world * B * N / per_step,
                'n_gpus': world, 'ranks_in_group': dist.get_world_size() if world > 1 else 1,
                'rank_devices': rank_devices,
                'backend': dist.get_backend() if world > 1 else None,
                'batch_per_gpu': B, 'ms_per_step': 1e3 * per_step,
                'final_loss': loss, 'hip_graph': bool(args.graph), 'adam': args.adam,
                'wgrad_wgs': _native.lib().gnnpp_get_tuning(17), 'wgrad_merged': _native.lib().gnnpp_get_tuning(18),
                'backward_fork_knob': _native.lib().gnnpp_get_tuning(15)}
        if args.cpu_seconds > 0 and world == 1:
            cb = cpu_baseline(B, args.cpu_seconds)
            cb['speedup_gpu_over_cpu'] = line['value'] / cb['value']
            line['cpu_baseline'] = cb
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
