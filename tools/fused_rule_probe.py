"""Where does the one-launch policy kernel stop paying?  gnnpp_policy_fwd with the default rule (GNNPP_TUNE_FUSED_POLICY = 1:
one launch for B <= 512 or N >= 13) against the forced one-launch kernel (= 2) and the forced two-kernel path (= 0),
N = 10, K = 3, HIP events around 100 back-to-back calls.  One JSON line per batch size."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_pathplanning_amd import _native                      # noqa: E402
from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet   # noqa: E402
from oracle import policy_oracle as orc                       # noqa: E402  (inputs only)

dev = torch.device('cuda:0')
st = _native.stream_ptr(dev)
L = _native.lib()


def main():
    N, K = 10, 3

    class Cfg:
        num_agents, nGraphFilterTaps, device = N, K, dev
    net = DecentralPlannerNet(Cfg()).to(dev).eval()
    net.load_state_dict(orc.init_state_dict(K))
    enc, taps, gb, aw, ab, _ = net.policy_pointers()
    for B in (512, 640, 768, 1024, 1280, 1536, 2048):
        obs = orc.synth_obs(min(B, 1024), N, seed=1337).to(dev)
        obs = obs.repeat((B + obs.shape[0] - 1) // obs.shape[0], 1, 1, 1, 1)[:B].contiguous()
        S = torch.from_numpy(orc.synth_gso_geometric(512, N, 20, seed=1337)).float().to(dev)
        S = S.repeat((B + 511) // 512, 1, 1)[:B].contiguous()
        ws = torch.empty(B * N, 128, device=dev)
        lg = torch.empty(N, B, 5, device=dev)
        args = (obs.data_ptr(), S.data_ptr(), enc, taps, gb, aw, ab, ws.data_ptr(), lg.data_ptr(), B, N, K, 1, 0, 0, None, st)
        row = {'B': B, 'N': N}
        for knob, name in ((0, 'two_kernels_us'), (2, 'one_launch_us'), (1, 'default_rule_us')):
            assert L.gnnpp_set_tuning(6, knob) == 0
            for _ in range(10):
                assert L.gnnpp_policy_fwd(*args) == 0
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(100):
                    L.gnnpp_policy_fwd(*args)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 10.0)
            row[name] = round(sorted(ts)[2], 2)
        L.gnnpp_set_tuning(6, 1)
        row['M_agent_steps_per_s_best'] = round(B * N / min(row['two_kernels_us'], row['one_launch_us']), 1)
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
