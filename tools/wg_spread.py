"""Per-workgroup start / duration structure of one launch of the fused policy kernel (measure build stamps)."""
import ctypes, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_pathplanning_amd import _native
from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
from oracle import policy_oracle as orc
M = _native.measure_lib()
M.gnnpp_measure_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device('cuda:0'); st = _native.stream_ptr(dev)
class Cfg2:
    num_agents, nGraphFilterTaps, device = 10, 3, dev
net = DecentralPlannerNet(Cfg2()).to(dev).eval(); net.load_state_dict(orc.init_state_dict(3))
N = 10
for prec in (0, 2):
    B = 512
    obs = orc.synth_obs(B, N, seed=1337).to(dev)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, 20, seed=1337)).float().to(dev)
    enc, taps, gb, aw, ab, K = net.policy_pointers()
    ws = torch.empty(B * N, 128, device=dev); lg = torch.empty(N, B, 5, device=dev)
    for _ in range(8):
        assert M.gnnpp_policy_fwd(obs.data_ptr(), S.data_ptr(), enc, taps, gb, aw, ab, ws.data_ptr(), lg.data_ptr(), B, N, 3, 1, 0, prec, None, st) == 0
        torch.cuda.synchronize()
    buf = np.zeros(1024 * 32, np.uint64)
    assert M.gnnpp_measure_read_stamps(buf.ctypes.data, buf.size) == 0
    us = buf.reshape(1024, 32)[:B, :16].astype(np.float64) * 0.01
    start = us[:, 11] - us[:, 11].min(); end = us[:, 14] - us[:, 11].min(); tot = us[:, 14] - us[:, 11]
    phases = [11, 0, 1, 2, 3, 4, 5, 12, 13, 14]
    names = ['stage', 'L0', 'L1', 'L2', 'L3', 'L4', 'FC', 'shift', 'contr']
    order = np.argsort(tot)
    def prof(idx):
        return {n: round(float(np.mean(us[idx, b] - us[idx, a])), 2) for n, a, b in zip(names, phases[:-1], phases[1:])}
    out = {'prec': prec, 'start_pct': [round(float(np.percentile(start, p)), 2) for p in (0, 25, 50, 75, 100)],
           'total_pct': [round(float(np.percentile(tot, p)), 2) for p in (0, 10, 25, 50, 75, 90, 100)],
           'end_pct': [round(float(np.percentile(end, p)), 2) for p in (0, 25, 50, 75, 100)],
           'corr_total_start': round(float(np.corrcoef(tot, start)[0, 1]), 3),
           'fastest_64_phases': prof(order[:64]), 'slowest_64_phases': prof(order[-64:]),
           'slowest_wg_ids_mod8': np.bincount(order[-64:] % 8, minlength=8).tolist(),
           'slowest_wg_ids_div256': np.bincount(order[-64:] // 256, minlength=2).tolist(),
           'mean_total_first256_second256': [round(float(tot[:256].mean()), 2), round(float(tot[256:].mean()), 2)],
           'mean_start_first256_second256': [round(float(start[:256].mean()), 2), round(float(start[256:].mean()), 2)]}
    print(json.dumps(out), flush=True)
