"""Training throughput for BASELINE config 4 (dcp_onlineExpert: 10 agents, K=3, batch 64 per GPU,
Adam lr 1e-3 wd 1e-5), 1 GPU or data parallel:
    python tools/train_bench.py --steps 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        tools/train_bench.py --steps 50
Prints one JSON line on rank 0 (agent-steps/s of full train steps, all ranks)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--graph', action='store_true', help='capture the train step in a HIP graph')
    ap.add_argument('--adam', choices=('fused', 'torch'), default='fused',
                    help='optimizer: training.FusedAdam (gnnpp_adam_step) or torch.optim.Adam')
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
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from gnn_pathplanning_amd.sharding import aggregate_throughput, gather_rank_devices
    from gnn_pathplanning_amd.training import FlatBucketDP, FusedAdam, train_step
    from oracle import policy_oracle as orc                   # synthetic inputs only

    class Cfg:
        num_agents, nGraphFilterTaps, device = 10, 3, dev
    torch.manual_seed(1337)
    net = DecentralPlannerNet(Cfg()).to(dev).train()
    dp = FlatBucketDP(net) if world > 1 else None
    if args.adam == 'torch':
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5, capturable=args.graph)
    else:                                                     # the same update on one HIP launch
        opt = FusedAdam(net.parameters(), lr=1e-3, weight_decay=1e-5)
    B, N = args.batch, 10
    obs = orc.synth_obs(B, N, seed=1337 + rank).to(dev)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, 20, seed=1337 + rank)).float().to(dev)
    g = torch.Generator().manual_seed(rank)
    tgt = torch.nn.functional.one_hot(torch.randint(0, 5, (B, N), generator=g), 5).float().to(dev)
    step = lambda: train_step(net, opt, obs, tgt, S, dp)      # noqa: E731
    if args.graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                loss = step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_loss = step()

        def step():                                           # noqa: F811
            graph.replay()
            return static_loss
    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    thr, units, el = aggregate_throughput(B * N * args.steps, el, device=dev)
    rank_devices = gather_rank_devices(dev)
    if rank == 0:
        print(json.dumps({'metric': 'training agent-steps/s (fwd+bwd+Adam, config 4)', 'value': thr,
                          'n_gpus': world, 'ranks_in_group': dist.get_world_size() if world > 1 else 1,
                          'rank_devices': rank_devices,
                          'backend': dist.get_backend() if world > 1 else None,
                          'batch_per_gpu': B, 'ms_per_step': 1e3 * el / args.steps,
                          'final_loss': float(loss.item()), 'hip_graph': bool(args.graph),
                          'adam': args.adam}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
