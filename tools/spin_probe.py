"""Probe: does a single spinning wave on a second stream slow the encoder kernel down?"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnn_pathplanning_amd import _native                                 # noqa: E402
from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet   # noqa: E402
from oracle import policy_oracle as orc                                  # noqa: E402

dev = torch.device('cuda:0')
L = _native.lib()
P = ctypes.CDLL(os.path.join(ROOT, 'tools', 'probe', 'libprobe.so'))
P.probe_spin.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
vp = lambda t: ctypes.c_void_p(t.data_ptr())                  # noqa: E731


class Cfg:
    num_agents, nGraphFilterTaps, device = 10, 3, dev


net = DecentralPlannerNet(Cfg()).to(dev).eval()
net.load_state_dict(orc.init_state_dict(3))
enc = net.packed_encoder()
M = 5120
obs = (torch.rand(M, 3, 11, 11, device=dev) < 0.1).float()
feat = torch.empty(M, 128, device=dev)
flag = torch.zeros(2, dtype=torch.int64, device=dev)
sink = torch.zeros(2, dtype=torch.int64, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
for _ in range(20):
    L.gnnpp_encoder_fwd(vp(obs), vp(enc), vp(feat), M, None, ctypes.c_void_p(sA.cuda_stream))
torch.cuda.synchronize()
row = {'probe': 'encoder next to one spinning wave', 'M': M}
for name, mode in (('alone', None), ('s_sleep_only', 0), ('s_sleep_and_poll', 1), ('busy_clock_loop', 2)):
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if mode is not None:
            P.probe_spin(vp(flag), 400, mode, vp(sink), ctypes.c_void_p(sB.cuda_stream))
        with torch.cuda.stream(sA):
            e0.record()
            for _ in range(5):
                L.gnnpp_encoder_fwd(vp(obs), vp(enc), vp(feat), M, None, ctypes.c_void_p(sA.cuda_stream))
            e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 5)
    row[name + '_us'] = round(best, 2)
print(json.dumps(row))
