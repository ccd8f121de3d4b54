"""When do the workgroups of one C2 launch of the one-launch policy kernel start and end?  (measure build, stamps 11 = entry,
14 = last phase done.)  Prints percentiles of start / end times relative to the earliest start, for the first and second
half of the grid (the second half takes the CUs' second slots)."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_pathplanning_amd import _native                      # noqa: E402
from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet   # noqa: E402
from oracle import policy_oracle as orc                       # noqa: E402  (inputs only)

dev = torch.device('cuda:0')
st = _native.stream_ptr(dev)
M = _native.measure_lib()
M.gnnpp_measure_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
for (B, N, K) in [(512, 10, 3), (256, 10, 3), (1024, 10, 3)]:
    class Cfg:
        num_agents, nGraphFilterTaps, device = N, K, dev
    net = DecentralPlannerNet(Cfg()).to(dev).eval()
    net.load_state_dict(orc.init_state_dict(K))
    obs = orc.synth_obs(B, N, seed=1337).to(dev)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, 20, seed=1337)).float().to(dev)
    enc, taps, gb, aw, ab, _ = net.policy_pointers()
    ws = torch.empty(B * N, 128, device=dev)
    lg = torch.empty(N, B, 5, device=dev)
    assert M.gnnpp_set_tuning(6, 2) == 0
    args = (obs.data_ptr(), S.data_ptr(), enc, taps, gb, aw, ab, ws.data_ptr(), lg.data_ptr(), B, N, K, 1, 0, 0, None, st)
    for _ in range(8):
        assert M.gnnpp_policy_fwd(*args) == 0
        torch.cuda.synchronize()
    buf = np.zeros(1024 * 32, np.uint64)
    assert M.gnnpp_measure_read_stamps(buf.ctypes.data, buf.size) == 0
    rows = buf.reshape(1024, 32)[:min(B, 1024)].astype(np.float64)
    t0 = rows[:, 11].min()
    start, end = (rows[:, 11] - t0) * 0.01, (rows[:, 14] - t0) * 0.01
    pct = lambda a: [round(float(np.percentile(a, q)), 2) for q in (0, 10, 50, 90, 100)]
    rec = {'B': B, 'N': N, 'K': K, 'start_us_pct_0_10_50_90_100': pct(start), 'end_us': pct(end),
           'duration_us': pct(end - start)}
    h = B // 2
    if B >= 512:
        rec['first_half'] = {'start': pct(start[:h]), 'end': pct(end[:h]), 'duration': pct((end - start)[:h])}
        rec['second_half'] = {'start': pct(start[h:]), 'end': pct(end[h:]), 'duration': pct((end - start)[h:])}
    print(json.dumps(rec), flush=True)
M.gnnpp_set_tuning(6, 1)
