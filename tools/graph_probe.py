"""Probe: policy step (encoder + filter launches) replayed from a HIP graph vs launched directly."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet   # noqa: E402
from oracle import policy_oracle as orc                                  # noqa: E402

dev = torch.device('cuda:0')


class Cfg:
    def __init__(self, n, k):
        self.num_agents, self.nGraphFilterTaps, self.device = n, k, dev


for (N, B, K, W) in ((10, 512, 3, 20), (10, 1, 3, 20), (50, 256, 3, 50)):
    net = DecentralPlannerNet(Cfg(N, K)).to(dev).eval()
    net.load_state_dict(orc.init_state_dict(K))
    obs = orc.synth_obs(B, N, seed=1).to(dev)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, W, seed=1)).float().to(dev)
    with torch.no_grad():
        for _ in range(5):
            net.addGSO(S); ref = net.forward_logits(obs)
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                net.addGSO(S); out = net.forward_logits(obs)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                net.addGSO(S); out = net.forward_logits(obs)
        torch.cuda.synchronize()
        row = {'probe': 'graph replay', 'N': N, 'B': B}
        for name, fn in (('direct_us', lambda: (net.addGSO(S), net.forward_logits(obs))), ('graph_us', g.replay)):
            best = 1e9
            for rep in range(3):
                for _ in range(20):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(300):
                    fn()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 300 * 1e6)
            row[name] = round(best, 2)
        g.replay(); torch.cuda.synchronize()
        row['equal'] = bool(torch.equal(out, ref))
        print(json.dumps(row), flush=True)
