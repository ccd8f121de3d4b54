"""Single-process A/B timings of the libgnnpp kernels (HIP events on the launch stream).
Prints one JSON line per measurement; used to pick defaults and to fill DESIGN.md tables."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_pathplanning_amd import _native                      # noqa: E402
from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet   # noqa: E402
from oracle import policy_oracle as orc                       # noqa: E402  (inputs only)

dev = torch.device('cuda:0')
L = _native.lib()
# phase ablation / early exit exist only in the -DGNNPP_MEASURE build (csrc/gnnpp_measure.h)
if len(sys.argv) > 1 and sys.argv[1] in ('encoder_phases', 'filter_ablation'):
    L = _native.measure_lib()
vp = lambda t: ctypes.c_void_p(t.data_ptr())                  # noqa: E731
st = _native.stream_ptr(dev)


def timeit(fn, reps=40, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best                                                # us


class Cfg:
    def __init__(self, n, k):
        self.num_agents, self.nGraphFilterTaps, self.device = n, k, dev


def encoder_phases():
    """Split-f16 encoder truncated after each phase (GNNPP_TUNE_ENCODER_STOP): cumulative us."""
    net = DecentralPlannerNet(Cfg(10, 3)).to(dev).eval()
    net.load_state_dict(orc.init_state_dict(3))
    enc = net.packed_encoder()
    L.gnnpp_set_tuning(0, 7)
    names = {1: 'staging', 2: 'L0', 3: 'L1', 4: 'L2', 5: 'L3', 6: 'L4', 0: 'full'}
    for M in (16, 4096, 5120, 40960):
        obs = (torch.rand(M, 3, 11, 11, device=dev) < 0.1).float()
        feat = torch.empty(M, 128, device=dev)
        row = {'kernel': 'encoder_h2_phases', 'M': M}
        for rep in range(2):
            for stop in (1, 2, 3, 4, 5, 6, 0):
                L.gnnpp_set_tuning(4, stop)
                t = timeit(lambda: L.gnnpp_encoder_fwd(vp(obs), vp(enc), vp(feat), M, None, st))
                row[names[stop]] = min(round(t, 2), row.get(names[stop], 1e9))
        L.gnnpp_set_tuning(4, 0)
        print(json.dumps(row), flush=True)
    L.gnnpp_set_tuning(0, -1)


def fused_ab():
    """Whole policy step (addGSO + forward): fused one-kernel path vs encoder kernel + filter kernel."""
    import time
    for (N, B, K, W) in ((10, 512, 3, 20), (10, 64, 3, 20), (10, 1, 3, 20), (16, 320, 3, 24), (10, 2048, 3, 20)):
        net = DecentralPlannerNet(Cfg(N, K)).to(dev).eval()
        net.load_state_dict(orc.init_state_dict(K))
        obs = orc.synth_obs(B, N, seed=1).to(dev)
        S = torch.from_numpy(orc.synth_gso_geometric(B, N, W, seed=1)).float().to(dev)
        row = {'kernel': 'policy_step', 'N': N, 'B': B}
        for rep in range(3):
            for mode in (1, 0):
                L.gnnpp_set_tuning(6, mode)
                for _ in range(30):
                    net.addGSO(S); net(obs)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(300):
                    net.addGSO(S); net(obs)
                torch.cuda.synchronize()
                us = (time.perf_counter() - t0) / 300 * 1e6
                k = 'fused_us' if mode else 'two_kernels_us'
                row[k] = min(round(us, 2), row.get(k, 1e9))
        L.gnnpp_set_tuning(6, 1)
        print(json.dumps(row), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'fused':
        return fused_ab()
    if len(sys.argv) > 1 and sys.argv[1] == 'encoder_phases':
        return encoder_phases()
    sd = orc.init_state_dict(3)
    net = DecentralPlannerNet(Cfg(10, 3)).to(dev).eval()
    net.load_state_dict(sd)
    enc = net.packed_encoder()
    # ---- encoder variants over M ----
    for M in (16, 4096, 5120, 12800, 40960):
        obs = (torch.rand(M, 3, 11, 11, device=dev) < 0.1).float()
        feat = torch.empty(M, 128, device=dev)
        row = {'kernel': 'encoder', 'M': M}
        L.gnnpp_set_tuning(0, 5)
        L.gnnpp_encoder_fwd(vp(obs), vp(enc), vp(feat), M, None, st)
        ref = feat.clone()
        L.gnnpp_set_tuning(0, 7)
        L.gnnpp_encoder_fwd(vp(obs), vp(enc), vp(feat), M, None, st)
        row['max_abs_diff_v7_vs_v5'] = float((feat - ref).abs().max())
        for v in (7, 5, 7, 5):
            L.gnnpp_set_tuning(0, v)
            t = timeit(lambda: L.gnnpp_encoder_fwd(vp(obs), vp(enc), vp(feat), M, None, st))
            row['v%d_us' % v] = min(round(t, 2), row.get('v%d_us' % v, 1e9))
        L.gnnpp_set_tuning(0, -1)
        print(json.dumps(row), flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == 'encoder':
        return
    # ---- filter: graphs-per-workgroup sweep ----
    for (N, B, K, W) in ((10, 512, 3, 20), (50, 256, 3, 50), (100, 128, 3, 100), (10, 4096, 3, 20)):
        gf = DecentralPlannerNet(Cfg(N, K)).to(dev).GFL[0]
        x = torch.relu(torch.randn(B * N, 128, device=dev))
        S = torch.from_numpy(orc.synth_gso_geometric(B, N, W, seed=1)).float().to(dev)
        y = torch.empty(B * N, 128, device=dev)
        gb = gf.bias.detach().reshape(-1)
        taps = gf.packed_taps()
        row = {'kernel': 'lsigf', 'N': N, 'B': B, 'K': K}
        for waves in (0, 8, 16):
            L.gnnpp_set_tuning(2, waves)
            for gpw in (0, 1, 2, 3, 4, 6, 8):
                if gpw * N > 112 or (waves == 0 and gpw != 0):
                    continue
                L.gnnpp_set_tuning(1, gpw)
                t = timeit(lambda: L.gnnpp_lsigf_fwd(vp(x), vp(S), vp(taps), vp(gb), vp(y), B, N, N,
                                                     128, 128, K, 1, 0, 1, 1, 1, 1, 0, None, st))
                row['w%d_gpw%d_us' % (waves, gpw)] = round(t, 2)
        L.gnnpp_set_tuning(1, 0)
        L.gnnpp_set_tuning(2, 0)
        # module-API layout (feature-major in/out), heuristic gpw
        xf = x.reshape(B, N, 128).permute(0, 2, 1).contiguous()
        yf = torch.empty(B, 128, N, device=dev)
        S4 = S.unsqueeze(1).contiguous()
        t = timeit(lambda: L.gnnpp_lsigf_fwd(vp(xf), vp(S4), vp(taps), vp(gb), vp(yf), B, N, N, 128, 128,
                                             K, 1, 0, 1, 0, 0, 0, 0, None, st))
        row['feature_major_us'] = round(t, 2)
        print(json.dumps(row), flush=True)
    # ---- whole policy step (python call included in the stream time) ----
    for (N, B, W) in ((10, 1, 20), (10, 64, 20), (10, 512, 20), (50, 256, 50), (100, 128, 100)):
        net = DecentralPlannerNet(Cfg(N, 3)).to(dev).eval()
        net.load_state_dict(sd)
        obs = orc.synth_obs(B, N).to(dev)
        S = torch.from_numpy(orc.synth_gso_geometric(B, N, W, seed=2)).float().to(dev)

        def step():
            net.addGSO(S)
            return net(obs)
        t = timeit(step, reps=30)
        print(json.dumps({'kernel': 'policy_step', 'N': N, 'B': B, 'us': round(t, 2),
                          'agent_steps_per_s': round(B * N / t * 1e6)}), flush=True)


def rollout_bench():
    import numpy as np
    from gnn_pathplanning_amd.rollout import BatchedRollout
    sd = orc.init_state_dict(3)
    for (N, B, W) in ((10, 512, 20), (50, 256, 50), (100, 128, 100)):
        rng = np.random.default_rng(N)
        grids = (rng.random((B, W, W)) < 0.08).astype(np.uint8)
        starts = np.zeros((B, N, 2), np.int64); goals = np.zeros((B, N, 2), np.int64)
        for b in range(B):
            free = np.argwhere(grids[b] == 0)
            idx = rng.choice(len(free), size=2 * N, replace=False)
            starts[b], goals[b] = free[idx[:N]], free[idx[N:]]
        net = DecentralPlannerNet(Cfg(N, 3)).to(dev).eval()
        net.load_state_dict(sd)
        env = BatchedRollout(grids, starts, goals, 10 ** 6, dev, tie_mode='hashed')
        env.step(net)
        row = {'kernel': 'rollout_step', 'N': N, 'B': B}
        row['observe_us'] = round(timeit(env.observe, reps=20), 2)
        row['gso_us'] = round(timeit(lambda: env.gso(5), reps=20), 2)
        net.addGSO(env.S)
        lg = net.forward_logits(env.obs)
        row['move_us'] = round(timeit(lambda: env.move(logits=lg), reps=20), 2)
        row['sim_step_us'] = round(timeit(lambda: env.move_and_observe(logits=lg), reps=20), 2)   # one launch
        row['move_then_pair_us'] = round(timeit(lambda: (env.move(logits=lg), env.gso_observe()), reps=20), 2)
        row['policy_us'] = round(timeit(lambda: (net.addGSO(env.S), net.forward_logits(env.obs)), reps=20), 2)
        row['step_us_one_per_call'] = round(timeit(lambda: env.step(net), reps=20), 2)
        t = timeit(lambda: env.steps(net, 8), reps=6) / 8       # eight steps enqueued per host call
        row['step_us'] = round(t, 2)
        row['agent_steps_per_s'] = round(B * N / t * 1e6)
        print(json.dumps(row), flush=True)


def filter_ablation():
    """Phase timing of the filter kernel by ablation (results of ablated runs are meaningless)."""
    for (N, B, K, W) in ((10, 512, 3, 20), (50, 256, 3, 50), (100, 128, 3, 100)):
        gf = DecentralPlannerNet(Cfg(N, K)).to(dev).GFL[0]
        x = torch.relu(torch.randn(B * N, 128, device=dev))
        S = torch.from_numpy(orc.synth_gso_geometric(B, N, W, seed=1)).float().to(dev)
        y = torch.empty(B * N, 128, device=dev)
        gb = gf.bias.detach().reshape(-1)
        taps = gf.packed_taps()
        row = {'kernel': 'lsigf_ablation', 'N': N, 'B': B}
        for name, mask in (('full', 0), ('no_shift', 1), ('no_mfma', 2), ('no_S', 4), ('no_epilogue', 8),
                           ('no_shift_no_mfma', 3), ('staging_only', 15), ('no_shift_mfma_S', 7)):
            L.gnnpp_set_tuning(3, mask)
            t = timeit(lambda: L.gnnpp_lsigf_fwd(vp(x), vp(S), vp(taps), vp(gb), vp(y), B, N, N, 128,
                                                 128, K, 1, 0, 1, 1, 1, 1, 0, None, st))
            row[name] = round(t, 2)
        L.gnnpp_set_tuning(3, 0)
        print(json.dumps(row), flush=True)


def pipelined_bench():
    """Two independent batches in flight: step i of batch A overlaps step i of batch B on a second
    stream (what a rollout driver with two episode batches per GPU does)."""
    import time
    sd = orc.init_state_dict(3)
    for (N, B, W) in ((10, 512, 20), (50, 256, 50)):
        nets, obs, gso, streams = [], [], [], []
        for i in range(3):
            net = DecentralPlannerNet(Cfg(N, 3)).to(dev).eval()
            net.load_state_dict(sd)
            nets.append(net)
            obs.append(orc.synth_obs(B, N, seed=i).to(dev))
            gso.append(torch.from_numpy(orc.synth_gso_geometric(B, N, W, seed=i)).float().to(dev))
            streams.append(torch.cuda.Stream())
        row = {'kernel': 'pipelined_steps', 'N': N, 'B': B}
        for ns in (1, 2, 3):
            def go(steps):
                for k in range(steps):
                    i = k % ns
                    with torch.cuda.stream(streams[i]):
                        nets[i].addGSO(gso[i])
                        nets[i](obs[i])
            go(12)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            go(120)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / 120
            row['streams%d_us_per_step' % ns] = round(t * 1e6, 2)
            row['streams%d_agent_steps_per_s' % ns] = round(B * N / t)
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'filter_ablation':
        filter_ablation()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'pipelined':
        pipelined_bench()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'rollout':
        rollout_bench()
        sys.exit(0)
    main()
