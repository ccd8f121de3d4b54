"""Where the time goes INSIDE the rollout kernels: phase time stamps of the -DGNNPP_MEASURE build
(csrc/gnnpp_common.h GNNPP_STAMP, 100 MHz wall clock).  Prints median microseconds per phase over the
workgroups of one launch, for the standalone simulator step kernel and for the one-launch policy step."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_pathplanning_amd import _native                      # noqa: E402
from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet   # noqa: E402
from gnn_pathplanning_amd.rollout import BatchedRollout       # noqa: E402
from oracle import policy_oracle as orc                       # noqa: E402  (inputs only)

M = _native.measure_lib()
M.gnnpp_measure_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device('cuda:0')
SLOTS = {'move:entry': 0, 'move:state_loaded': 1, 'move:proposed': 2, 'move:pass1': 3, 'move:passes': 4,
         'move:final_pass': 5, 'move:stored': 6, 'sim:move_done': 7, 'sim:gso_done': 8, 'sim:observe_done': 9,   # gso_done = graph built + goal cells marked
         'policy:head_done': 10, 'kernel:start': 11}


def stamps(nwg):
    buf = np.zeros(1024 * 32, np.uint64)
    assert M.gnnpp_measure_read_stamps(buf.ctypes.data, buf.size) == 0
    rows = buf.reshape(1024, 32)[:nwg].astype(np.float64)
    stamps.cycles = rows[:, 16:]                                          # shader clock cycles
    return rows[:, :16] * 0.01                                            # microseconds (100 MHz wall clock)


def report(tag, st, order):
    row = {'what': tag}
    for a, b in zip(order[:-1], order[1:]):
        d = st[:, SLOTS[b]] - st[:, SLOTS[a]]
        row['%s -> %s' % (a, b)] = round(float(np.median(d)), 2)
    row['total'] = round(float(np.median(st[:, SLOTS[order[-1]]] - st[:, SLOTS[order[0]]])), 2)
    cyc = stamps.cycles[:, SLOTS[order[-1]]] - stamps.cycles[:, SLOTS[order[0]]]
    row['engine_clock_GHz'] = round(float(np.median(cyc / (st[:, SLOTS[order[-1]]] - st[:, SLOTS[order[0]]]))) * 1e-3, 3)
    print(json.dumps(row), flush=True)


for (N, B, W) in ((10, 512, 20), (16, 512, 20), (100, 128, 100)):
    class Cfg:
        num_agents, nGraphFilterTaps, device = N, 3, dev
    rng = np.random.default_rng(N)
    grids = (rng.random((B, W, W)) < 0.08).astype(np.uint8)
    starts = np.zeros((B, N, 2), np.int64); goals = np.zeros((B, N, 2), np.int64)
    for b in range(B):
        free = np.argwhere(grids[b] == 0)
        idx = rng.choice(len(free), size=2 * N, replace=False)
        starts[b], goals[b] = free[idx[:N]], free[idx[N:]]
    net = DecentralPlannerNet(Cfg()).to(dev).eval()
    net.load_state_dict(orc.init_state_dict(3))
    env = BatchedRollout(grids, starts, goals, 10 ** 6, dev, tie_mode='hashed')
    for _ in range(5):
        env.step(net)                                         # product library: warm state, real positions
    torch.cuda.synchronize()
    # one launch of each kernel through the MEASURE library on the same episode state
    r = env._r
    net.addGSO(env.S)
    lg = net.forward_logits(env.obs)
    r.logits, r.actions, r.grow, r.currentstep = lg.data_ptr(), None, 0, env.t + 1
    st_ptr = _native.stream_ptr(dev)
    for _ in range(3):
        assert M.gnnpp_rollout_step(ctypes.byref(r), st_ptr) == 0
        torch.cuda.synchronize()
    report('rollout_step_kernel N=%d B=%d' % (N, B), stamps(min(B, 1024)),
           ['move:entry', 'move:state_loaded', 'move:proposed', 'move:pass1', 'move:passes', 'move:final_pass',
            'move:stored', 'sim:move_done', 'sim:gso_done', 'sim:observe_done'])
    if N > 16:                                                # (the one-launch policy step is for N <= 16)
        continue
    ptrs = net.policy_pointers()
    enc, taps, gb, aw, ab, K = ptrs
    M.gnnpp_rollout_policy_step.argtypes = [ctypes.POINTER(_native.RolloutStruct)] + [ctypes.c_void_p] * 5 + \
        [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    for _ in range(3):
        r.currentstep += 1
        assert M.gnnpp_rollout_policy_step(ctypes.byref(r), enc, taps, gb, aw, ab, K, 0, st_ptr) == 0
        torch.cuda.synchronize()
    report('policy_step (one launch) N=%d B=%d' % (N, B), stamps(min(B, 1024)),
           ['kernel:start', 'policy:head_done', 'move:entry', 'move:state_loaded', 'move:proposed', 'move:pass1',
            'move:passes', 'move:final_pass', 'move:stored', 'sim:move_done', 'sim:gso_done', 'sim:observe_done'])


# ---- where the 40 us of the one-launch POLICY kernel go (C2: 512 graphs of 10 agents, two workgroups per CU;
# 256 graphs: one workgroup per CU, the kernel's latency without a neighbour on the CU)
class Cfg2:
    num_agents, nGraphFilterTaps, device = 10, 3, dev
net = DecentralPlannerNet(Cfg2()).to(dev).eval()
net.load_state_dict(orc.init_state_dict(3))
M.gnnpp_policy_fwd.argtypes = [ctypes.c_void_p] * 9 + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_void_p]
SLOTS.update({'enc:staged': 0, 'enc:L0': 1, 'enc:L1': 2, 'enc:L2': 3, 'enc:L3': 4, 'enc:L4': 5, 'enc:FC(z0)': 12,
              'filter:shifts': 13, 'filter:contraction': 14})
N = 10
for B in (512, 256):
    obs = orc.synth_obs(B, N, seed=1337).to(dev)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, 20, seed=1337)).float().to(dev)
    net.addGSO(S)
    for _ in range(20):
        net(obs)
    enc, taps, gb, aw, ab, K = net.policy_pointers()
    ws = torch.empty(B * N, 128, device=dev)
    lg = torch.empty(N, B, 5, device=dev)
    for _ in range(5):
        assert M.gnnpp_policy_fwd(obs.data_ptr(), S.data_ptr(), enc, taps, gb, aw, ab, ws.data_ptr(), lg.data_ptr(),
                                  B, N, 3, 1, 0, 0, None, _native.stream_ptr(dev)) == 0
        torch.cuda.synchronize()
    report('policy kernel (B=%d, N=10: %d workgroup(s) per CU)' % (B, B // 256), stamps(B),
           ['kernel:start', 'enc:staged', 'enc:L0', 'enc:L1', 'enc:L2', 'enc:L3', 'enc:L4', 'enc:FC(z0)',
            'filter:shifts', 'filter:contraction'])


# (the filter kernels' phases: tools/b3_stamps.py filter / small)
