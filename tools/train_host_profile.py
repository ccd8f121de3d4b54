"""Where the HOST time of an eager training step goes (cProfile over the steady-state loop only): the eager
step at B = 64 is host-bound (the same step replayed from a HIP graph takes half the time), so this is the
profile that matters for it.  Prints the top functions by own time."""
import cProfile
import io
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet   # noqa: E402
from gnn_pathplanning_amd.training import FusedAdam, train_step        # noqa: E402
from oracle import policy_oracle as orc                                 # noqa: E402  (synthetic inputs only)


def main():
    dev = torch.device('cuda:0')

    class Cfg:
        num_agents, nGraphFilterTaps, device = 10, 3, dev
    torch.manual_seed(1337)
    net = DecentralPlannerNet(Cfg()).to(dev).train()
    opt = FusedAdam(net.parameters(), lr=1e-3, weight_decay=1e-5)
    B, N = 64, 10
    obs = orc.synth_obs(B, N, seed=1337).to(dev)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, 20, seed=1337)).float().to(dev)
    tgt = torch.nn.functional.one_hot(torch.randint(0, 5, (B, N)), 5).float().to(dev)
    for _ in range(30):
        train_step(net, opt, obs, tgt, S)
    torch.cuda.synchronize()
    if len(sys.argv) > 1 and sys.argv[1] == 'ops':
        # which aten / autograd op launches which device kernel (the stray fill / copy launches of the step)
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
            for _ in range(5):
                train_step(net, opt, obs, tgt, S)
            torch.cuda.synchronize()
        rows = [e for e in prof.events() if e.device_type is not None]
        seen = {}
        for e in prof.events():
            nm = e.name
            if any(k in nm for k in ('fill', 'copy', 'Memcpy', 'zero', 'clamp', 'threshold', 'contiguous', 'clone', 'empty_like')):
                key = (nm, tuple((e.stack or [])[:6]))
                seen[key] = seen.get(key, 0) + 1
        for (nm, stack), c in sorted(seen.items(), key=lambda kv: -kv[1])[:24]:
            print('%4d  %s' % (c, nm))
            for fr in stack:
                if 'site-packages/torch' not in fr:
                    print('        ', fr)
        return
    steps = 300
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        train_step(net, opt, obs, tgt, S)
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s)
    st.sort_stats('tottime').print_stats(32)
    out = s.getvalue()
    print('per-step host time under the profiler: %.3f ms' % (st.total_tt / steps * 1e3))
    print(out[out.index('ncalls'):][:6000])


if __name__ == '__main__':
    main()
