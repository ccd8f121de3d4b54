"""Headline benchmark: agent-steps/s of the policy forward pass (BASELINE.json metric).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one policy forward (DecentralPlannerNet.addGSO + forward) over one batch of synthetic
input resident in HBM: BASELINE.json configs[1] = 10 agents, K=3, B=512 GSO+observation batch
(seed 1337).  With N GPUs every rank runs its own replica on its own batch
(independent rollout shards, no data-path collective): weak scaling, value = total agent-steps/s.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      dominant kernel (the fused encoder): algorithmic fp32 FLOPs / measured launch time
                vs the matrix-pipe peak of /opt/skills/guides/MI355X_MICROARCH.md for the arithmetic
                the kernel runs (split-f16 schedule: 2500 / 3 TFLOP/s; fp32 schedules: 157.3)
  cpu_baseline  the CPU oracle (op-for-op restatement of the reference's PyTorch path, pinned to
                golden vectors) timed on this box's host cores on a bounded sample of the workload
  parity        max |dlogit| and action-id agreement GPU vs oracle on the bench batch
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (agents N, map W, taps K, batch B)          -- BASELINE.json configs[1], [2], [4]
    'c2': (10, 20, 3, 512),
    'c3': (50, 50, 3, 256),
    'c5': (100, 100, 3, 128),
}
FP32_MFMA_PEAK_TFLOPS = 157.3           # MI355X_MICROARCH.md, chip-level parameters
F16_MFMA_PEAK_TFLOPS = 2500.0           # dense f16/bf16 MFMA peak (~2.5 PF), same table
ENC_MACS_PER_AGENT = 1238112 + 16384    # CNN + compress MLP


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def pick_cpu_threads(orc, sd, N, K, budget_s=6.0):
    """Thread count that makes the CPU oracle fastest on a small sample (B=32) of the workload.
    torch's default (= all logical cores) can be catastrophically oversubscribed on the GPU box."""
    obs = orc.synth_obs(32, N, seed=1)
    S = torch.from_numpy(orc.synth_gso_geometric(32, N, 20, seed=1)).float()
    best_t, best = 1, float('inf')
    t_begin = time.perf_counter()
    for t in (1, 2, 4, 8, 16, 32, 64, 128):
        if t > usable_cores() or time.perf_counter() - t_begin > budget_s:
            break
        torch.set_num_threads(t)
        with torch.no_grad():
            orc.policy_forward(sd, S, obs)
            t0 = time.perf_counter()
            orc.policy_forward(sd, S, obs)
            dt = time.perf_counter() - t0
        if dt < best:
            best_t, best = t, dt
        elif dt > 2.0 * best:
            break
    torch.set_num_threads(best_t)
    return best_t


def policy_flops_per_agent(K, mean_deg):
    return 2.0 * (1238112 + 16384 + K * 128 * 128 + (K - 1) * mean_deg * 128 + 640)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', default='c2', choices=sorted(CONFIGS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--pipeline-streams', type=int, default=3,
                    help='extra measurement: independent batches in flight on this many HIP streams '
                         '(reported under "pipelined", never as "value"); 0 disables')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--prewarm-seconds', type=float, default=0.25,
                    help='untimed device pre-warm before the W warmup steps: the same step in a loop '
                         'until the clocks / host caches have settled (the first few hundred steps '
                         'after start-up run ~8 %% slower, profiles/r01_warmup_sweep.jsonl); 0 disables')
    ap.add_argument('--dist-backend', default='nccl',
                    help='nccl (= RCCL, default); gloo only to exercise the multi-rank code path '
                         'on a box with fewer GPUs than ranks (with GNNPP_BENCH_DEVICE=0)')
    ap.add_argument('--traffic-bytes', type=float, default=None,
                    help='HBM bytes per encoder launch from a separate rocprofv3 --pmc pass')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus or world == 1, 'launch with torch.distributed.run for --gpus > 1'
    assert torch.cuda.is_available(), 'bench.py needs the MI355X'
    dev_index = int(os.environ.get('GNNPP_BENCH_DEVICE', local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    from gnn_pathplanning_amd import _native
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from oracle import policy_oracle as orc           # checker + cpu_baseline leg only
    _native.lib()

    N, W, K, B = CONFIGS[args.config]

    class Cfg:
        num_agents, nGraphFilterTaps, device = N, K, dev

    sd = orc.init_state_dict(K, seed=1337)
    net = DecentralPlannerNet(Cfg()).to(dev).eval()
    net.load_state_dict(sd)
    seed = 1337 + rank
    obs_cpu = orc.synth_obs(B, N, seed=seed)
    S64 = orc.synth_gso_geometric(B, N, W, seed=seed)
    mean_deg = float((S64 != 0).sum() / (B * N))
    S_cpu = torch.from_numpy(S64).float()
    obs, S = obs_cpu.to(dev), S_cpu.to(dev)

    def step():
        net.addGSO(S)
        return net(obs)

    with torch.no_grad():
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < args.prewarm_seconds:     # set-up, not part of W or K
            for _ in range(50):
                step()
            torch.cuda.synchronize()
        for _ in range(args.warmup):
            out = step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    from gnn_pathplanning_amd.sharding import aggregate_throughput
    value, _, elapsed = aggregate_throughput(B * N * args.steps, elapsed, device=dev)

    # Secondary: the same K steps with `pipeline_streams` independent rollout batches in flight
    # (batch i on stream i % S).  Sequentially dependent steps of ONE batch cannot overlap, so this
    # is reported separately; it is what a rollout driver holding S episode batches per GPU gets.
    pipelined = None
    if args.pipeline_streams > 1:
        S_n = args.pipeline_streams
        nets, ins, streams = [net], [(obs, S)], [torch.cuda.Stream() for _ in range(S_n)]
        for i in range(1, S_n):
            ni = DecentralPlannerNet(Cfg()).to(dev).eval()
            ni.load_state_dict(sd)
            nets.append(ni)
            ins.append((orc.synth_obs(B, N, seed=seed + 1000 * i).to(dev),
                        torch.from_numpy(orc.synth_gso_geometric(B, N, W, seed=seed + 1000 * i)).float().to(dev)))

        def run_pipelined(k_steps):
            for k in range(k_steps):
                i = k % S_n
                with torch.cuda.stream(streams[i]):
                    nets[i].addGSO(ins[i][1])
                    nets[i](ins[i][0])
        with torch.no_grad():
            run_pipelined(max(args.warmup, S_n))
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            el_p = float('inf')
            for _ in range(2):                      # best of two: the enqueue side is host-bound-ish
                t0 = time.perf_counter()
                run_pipelined(args.steps)
                torch.cuda.synchronize()
                if dist is not None:
                    dist.barrier()
                el_p = min(el_p, time.perf_counter() - t0)
        v_p, _, el_p = aggregate_throughput(B * N * args.steps, el_p, device=dev)
        pipelined = {'streams': S_n, 'value': v_p, 'ms_per_step': 1e3 * el_p / args.steps,
                     'note': 'K independent batches round-robin on %d HIP streams; not the headline' % S_n}

    result = {
        'metric': 'agent-steps/sec (policy fwd)', 'value': value, 'unit': 'agent-steps/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'prewarm_s': args.prewarm_seconds,
        'config': {'workload': 'policy forward (addGSO+forward), %d agents, %dx%d map GSO, K=%d, '
                               'batch=%d per GPU, eval mode, fp32 GSO resident in HBM'
                               % (N, W, W, K, B),
                   'name': args.config, 'agents': N, 'taps': K, 'batch_per_gpu': B,
                   'mean_degree': round(mean_deg, 3), 'parallelism': 'replicas x%d' % world},
    }

    if pipelined is not None:
        result['pipelined'] = pipelined
    if rank == 0:
        L = _native.lib()
        M = B * N
        enc = net.packed_encoder()
        feat = torch.empty(M, 128, device=dev)
        st = _native.stream_ptr(dev)
        import ctypes
        vp = lambda t: ctypes.c_void_p(t.data_ptr())       # noqa: E731

        def time_kernel(fn, reps=50):
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()                                     # events on the stream the kernels use
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e-3 / reps

        traffic = args.traffic_bytes
        pmc_file = os.path.join(ROOT, 'profiles', 'pmc_encoder_%s.json' % args.config)
        if traffic is None and os.path.exists(pmc_file):
            # HBM bytes per launch from separate rocprofv3 --pmc passes of this same command,
            # gfx950 correction of the microarch guide: FETCH_SIZE counts half of wide reads
            pmc = json.load(open(pmc_file))
            traffic = (2.0 * pmc['FETCH_SIZE_KB_per_dispatch'] + pmc['WRITE_SIZE_KB_per_dispatch']) * 1024.0
        t_enc = time_kernel(lambda: L.gnnpp_encoder_fwd(vp(obs), vp(enc), vp(feat), M, st))
        flops = 2.0 * ENC_MACS_PER_AGENT * M
        achieved = flops / t_enc / 1e12
        tiles = (M + 15) // 16
        variant = L.gnnpp_get_tuning(0)
        if variant == 7:
            # split-f16 schedule: every fp32 MAC is three f16 MACs on the f16 matrix pipe, so the
            # bound for ALGORITHMIC fp32 FLOPs is the dense f16 MFMA peak / 3.  Executed work per
            # 16-agent tile: 4314 v_mfma_f32_16x16x32_f16 of 16384 FLOP (taps that fall on the
            # zero padding and pooled-away positions are never issued).
            peak = F16_MFMA_PEAK_TFLOPS / 3.0
            exe = 4314 * 16384.0 * tiles
            result['roofline'] = {
                'kernel': 'gnnpp::encoder_kernel_h2', 'bound': 'mfma',
                'dtype': 'f16 hi/lo split of fp32 operands (3 f16 MFMAs per fp32 product), fp32 accumulate',
                'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
                'peak_note': 'dense f16 MFMA peak %.1f / 3 products; the exact-fp32 MFMA pipe peaks at '
                             '%.1f TFLOP/s' % (F16_MFMA_PEAK_TFLOPS, FP32_MFMA_PEAK_TFLOPS),
                'vs_fp32_mfma_peak': achieved / FP32_MFMA_PEAK_TFLOPS,
                'traffic': traffic, 'algorithmic_bytes': M * (363 + 128) * 4.0 + 156288 * 4.0,
                'avg_launch_us': t_enc * 1e6, 'flops_per_launch': flops,
                'executed_f16_flops_per_launch': exe,
                'f16_pipe_busy_frac': exe / t_enc / 1e12 / F16_MFMA_PEAK_TFLOPS,
            }
        else:
            result['roofline'] = {
                'kernel': 'gnnpp::encoder_kernel', 'bound': 'mfma', 'achieved': achieved,
                'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / FP32_MFMA_PEAK_TFLOPS,
                'traffic': traffic, 'algorithmic_bytes': M * (363 + 128) * 4.0 + 156288 * 4.0,
                'avg_launch_us': t_enc * 1e6,
                # the kernel skips zero-padding taps and pooled-away positions: MFMA work it executes
                # (v3: 8876 MFMAs of 2048 FLOP per 16-agent tile; v2: 11300; nominal: 12544)
                'executed_flops_per_launch': 8876 * 2048.0 * tiles,
                'executed_frac': 8876 * 2048.0 * tiles / t_enc / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                'flops_per_launch': flops,
            }
        result['encoder_schedule'] = variant
        if variant == 7:
            result['dtype'] = ('f32 operands as f16 hi+lo pairs on the f16 MFMA pipe (encoder, filter contraction), '
                               'f32 MFMA (graph shifts, head); f32 accumulate')
        # Is the step ONE kernel?  (gnnpp_policy_fwd's rule: N <= 16, K = 3, B <= 512 or N >= 13.)  Then the
        # dominant kernel of the step is the fused policy kernel -- encoder + this graph's filter + head in
        # one workgroup per graph -- and the roofline describes THAT launch; the encoder-only figures
        # above move to `encoder_kernel_alone`.
        fused = (variant == 7 and L.gnnpp_get_tuning(6) == 1 and L.gnnpp_get_tuning(5) == 1 and N <= 16
                 and K == 3 and (B <= 512 or N >= 13))
        if fused:
            act = net.actionsMLP[0]
            aw, ab = act.weight.detach().contiguous(), act.bias.detach().contiguous()
            gbias = net.GFL[0].bias.detach().reshape(-1).contiguous()
            lg = torch.empty(N, B, 5, device=dev)
            taps = net.GFL[0].packed_taps()
            t_pol = time_kernel(lambda: L.gnnpp_policy_fwd(vp(obs), vp(S), vp(enc), vp(taps), vp(gbias), vp(aw),
                                                           vp(ab), vp(feat), vp(lg), B, N, K, 0, st), reps=200)
            pflops = policy_flops_per_agent(K, mean_deg) * M
            pmc_pol = os.path.join(ROOT, 'profiles', 'pmc_policy_%s.json' % args.config)
            ptraffic = args.traffic_bytes
            if ptraffic is None and os.path.exists(pmc_pol):
                pp = json.load(open(pmc_pol))
                ptraffic = (2.0 * pp['FETCH_SIZE_KB_per_dispatch'] + pp['WRITE_SIZE_KB_per_dispatch']) * 1024.0
            enc_alone = result['roofline']
            exe = (4314 + 288) * 16384.0 * B          # f16 MFMAs per graph tile: encoder 4314 + filter 288
            result['encoder_kernel_alone'] = enc_alone
            result['roofline'] = {
                'kernel': 'gnnpp::encoder_kernel_h2<true> (fused policy kernel: encoder + graph filter + '
                          'action head, one workgroup per graph)', 'bound': 'mfma',
                'dtype': enc_alone['dtype'], 'achieved': pflops / t_pol / 1e12, 'peak': enc_alone['peak'],
                'unit': 'TFLOP/s', 'frac': pflops / t_pol / 1e12 / enc_alone['peak'],
                'peak_note': enc_alone['peak_note'] + '; a graph of %d agents occupies a 16-lane tile, so at '
                             'most %d/16 of the pipe does algorithmic work' % (N, N),
                'vs_fp32_mfma_peak': pflops / t_pol / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                'traffic': ptraffic,
                'algorithmic_bytes': M * 363 * 4.0 + B * N * N * 4.0 + M * 20.0 + (156288 + 3 * 128 * 128 + 768) * 4.0,
                'avg_launch_us': t_pol * 1e6, 'flops_per_launch': pflops,
                'executed_f16_flops_per_launch': exe,
                'f16_pipe_busy_frac': exe / t_pol / 1e12 / F16_MFMA_PEAK_TFLOPS,
            }
        # secondary: the graph-filter kernel alone (node-major features in, ReLU'd features out)
        gf = net.GFL[0]
        y = torch.empty(M, 128, device=dev)
        gb = gf.bias.detach().reshape(-1)
        t_gf = time_kernel(lambda: L.gnnpp_lsigf_fwd(vp(feat), vp(S), vp(gf.packed_taps()), vp(gb),
                                                     vp(y), B, N, N, 128, 128, K, 1, 0, 1, 1, 1, 1, st))
        gf_bytes = M * (1024 + 4 * N) + 196608.0 * K / 3
        result['filter_kernel'] = {
            'kernel': 'gnnpp::lsigf_kernel', 'avg_launch_us': t_gf * 1e6,
            'agent_steps_per_s': M / t_gf,
            'algorithmic_GBps': gf_bytes / t_gf / 1e9, 'hbm_frac_of_8TBps': gf_bytes / t_gf / 8e12,
            'mfma_TFLOPs': 2.0 * (K * 128 * 128 + (K - 1) * mean_deg * 128) * M / t_gf / 1e12,
        }
        result['step_breakdown_us'] = {'encoder_kernel_alone': t_enc * 1e6, 'filter_kernel_alone': t_gf * 1e6,
                                       'whole_step_wall': 1e6 * elapsed / args.steps}
        if fused:
            result['step_breakdown_us']['fused_policy_kernel'] = t_pol * 1e6
        result['policy_TFLOPs'] = policy_flops_per_agent(K, mean_deg) * value / world / 1e12

        # secondary: one whole rollout step on the device (observation builder + communication GSO
        # + this forward + action decode / collision shielding), B episodes on random maps
        import numpy as np
        from gnn_pathplanning_amd.rollout import BatchedRollout
        rng = np.random.default_rng(1337)
        grids = (rng.random((B, W, W)) < 0.08).astype(np.uint8)
        starts = np.zeros((B, N, 2), np.int64)
        goals = np.zeros((B, N, 2), np.int64)
        for b_i in range(B):
            free = np.argwhere(grids[b_i] == 0)
            pick = rng.choice(len(free), size=2 * N, replace=False)
            starts[b_i], goals[b_i] = free[pick[:N]], free[pick[N:]]
        env = BatchedRollout(grids, starts, goals, 10 ** 6, dev, tie_mode='hashed', seed=1337)
        with torch.no_grad():
            t_roll = time_kernel(lambda: env.step(net), reps=30)
        result['rollout_step'] = {'us': t_roll * 1e6, 'agent_steps_per_s': B * N / t_roll,
                                  'what': 'observe + gso + policy forward + move (collision shielding), '
                                          'all on the device, %d episodes' % B}

        # parity gate on the bench batch + CPU baseline (bounded sample of the same workload)
        threads = pick_cpu_threads(orc, sd, N, K)
        with torch.no_grad():
            want = orc.policy_forward(sd, S_cpu, obs_cpu)
        got = [o.cpu() for o in out]
        err = max((g - w).abs().max().item() for g, w in zip(got, want))
        margin = orc.top2_margin(want)
        ids_w = orc.decode_actions(want)
        ids_g = torch.stack([g.argmax(-1) for g in got], 1)
        clear = margin > 1e-5
        result['parity'] = {'max_abs_dlogit': err, 'tolerance': 1e-4,
                            'argmax_equal_on_clear_rows': bool(torch.equal(ids_g[clear], ids_w[clear])),
                            'near_tie_rows': int((~clear).sum()), 'rows': int(clear.numel())}
        if not args.no_cpu_baseline:
            with torch.no_grad():
                orc.policy_forward(sd, S_cpu, obs_cpu)
                times = []
                t_start = time.perf_counter()
                while time.perf_counter() - t_start < args.cpu_seconds and len(times) < 200:
                    t1 = time.perf_counter()
                    orc.policy_forward(sd, S_cpu, obs_cpu)
                    times.append(time.perf_counter() - t1)
            times.sort()
            med = times[len(times) // 2]
            result['cpu_baseline'] = {
                'value': B * N / med, 'unit': 'agent-steps/s', 'cores': threads, 'usable_cores': usable_cores(),
                'kind': 'port',
                'sample': '%d repetitions (median) of the same %s batch through oracle/policy_oracle.py '
                          '(torch %s CPU, fp32, eval, no_grad), ~%.0f s of host time'
                          % (len(times), args.config, torch.__version__, sum(times)),
                'ms_per_step': med * 1e3, 'speedup_gpu_over_cpu': value / world / (B * N / med)}
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
