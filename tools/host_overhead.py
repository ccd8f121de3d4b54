"""Host-side cost of one policy step (addGSO + forward): cProfile of the enqueue path."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet   # noqa: E402
from oracle import policy_oracle as orc                                  # noqa: E402

dev = torch.device('cuda:0')


class Cfg:
    num_agents, nGraphFilterTaps, device = 10, 3, dev


class Cfg100:
    num_agents, nGraphFilterTaps, device = 100, 3, dev


# the two-launch path (teams of 17 .. 100 agents: encoder kernel + policy_filter_kernel), tiny batch
net100 = DecentralPlannerNet(Cfg100()).to(dev).eval()
net100.load_state_dict(orc.init_state_dict(3))
obs100 = orc.synth_obs(1, 100, seed=1).to(dev)
S100 = torch.from_numpy(orc.synth_gso_geometric(1, 100, 100, seed=1)).float().to(dev)
with torch.no_grad():
    for _ in range(200):
        net100.addGSO(S100); net100(obs100)
    torch.cuda.synchronize()
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):                # (short bursts: the HIP queue must not fill up and block the host)
            net100.addGSO(S100); net100(obs100)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print('N = 100 (two launches): %.2f us of host time per step (enqueue only), %.2f us per step until the '
              'device is done' % ((t1 - t0) / 300 * 1e6, (t2 - t0) / 300 * 1e6))

net = DecentralPlannerNet(Cfg()).to(dev).eval()
net.load_state_dict(orc.init_state_dict(3))
B = 8                                   # tiny batch: the GPU is never the bottleneck
obs = orc.synth_obs(B, 10, seed=1).to(dev)
S = torch.from_numpy(orc.synth_gso_geometric(B, 10, 20, seed=1)).float().to(dev)
with torch.no_grad():
    for _ in range(200):
        net.addGSO(S); net(obs)
    torch.cuda.synchronize()
    for name, fn in (('addGSO+forward', lambda: (net.addGSO(S), net(obs))),
                     ('addGSO+forward_logits', lambda: (net.addGSO(S), net.forward_logits(obs)))):
        t0 = time.perf_counter()
        for _ in range(2000):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print('%s: %.2f us of host time per step (enqueue only)' % (name, (t1 - t0) / 2000 * 1e6))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        net.addGSO(S); net(obs)
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18)
    print(s.getvalue()[:3500])
