"""A/B timings of gnnpp::lsigf_kernel at the BASELINE shapes (HIP events, node-major in/out, bias + ReLU):
graphs per workgroup x waves per workgroup x two-workgroups-per-graph.  One JSON line per shape; every
variant's output is compared with the default's (must be identical: only the schedule changes)."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_pathplanning_amd import _native                      # noqa: E402
from gnn_pathplanning_amd.graphML import pack_filter_taps     # noqa: E402
from oracle import policy_oracle as orc                       # noqa: E402  (inputs only)

dev = torch.device('cuda:0')
L = _native.lib()
vp = lambda t: ctypes.c_void_p(t.data_ptr())                  # noqa: E731
st = _native.stream_ptr(dev)


def timeit(fn, reps=60, warm=8):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(4):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return round(best, 2)


for (B, N, W, K) in ((512, 10, 20, 3), (256, 50, 50, 3), (128, 100, 100, 2), (128, 100, 100, 3),
                     (128, 100, 100, 4), (64, 100, 100, 3), (1024, 50, 50, 3), (16, 64, 40, 3)):
    g = torch.Generator().manual_seed(N + K)
    h = ((torch.rand(128, 1, K, 128, generator=g) * 2 - 1) / (128 * K) ** 0.5).to(dev)
    gb = (torch.randn(128, generator=g) * 0.1).to(dev)
    x = torch.relu(torch.randn(B * N, 128, generator=g)).to(dev)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, W, seed=N)).float().to(dev)
    taps = pack_filter_taps(h)
    y = torch.empty(B * N, 128, device=dev)

    def run():
        rc = L.gnnpp_lsigf_fwd(vp(x), vp(S), vp(taps), vp(gb), vp(y), B, N, N, 128, 128, K, 1, 0, 1, 1, 1, 1,
                               0, None, st)
        assert rc == 0, rc
    row = {'B': B, 'N': N, 'K': K, 'mean_degree': round(float((S != 0).sum() / (B * N)), 2)}
    run()
    ref = y.clone()
    row['default_us'] = timeit(run)
    for split in (1, 2):
        for waves in (8, 16):
            for gpw in (1, 2, 3, 4, 8):
                if gpw * N > 112 or (split == 2 and gpw != 1):
                    continue
                L.gnnpp_set_tuning(7, split); L.gnnpp_set_tuning(2, waves); L.gnnpp_set_tuning(1, gpw)
                y.zero_()
                run()
                assert torch.equal(y, ref), (split, waves, gpw)
                row['s%d_w%d_g%d' % (split, waves, gpw)] = timeit(run)
    L.gnnpp_set_tuning(7, 0); L.gnnpp_set_tuning(2, 0); L.gnnpp_set_tuning(1, 0)
    # exact-fp32 contraction for reference
    L.gnnpp_set_tuning(5, 0)
    row['fp32_contraction_us'] = timeit(run)
    L.gnnpp_set_tuning(5, 1)
    # float64 check of the default
    want = torch.relu(torch.from_numpy(orc.lsigf_f64(h.cpu().numpy(), S.cpu().unsqueeze(1).numpy(),
                                                     x.cpu().reshape(B, N, 128).permute(0, 2, 1).numpy(),
                                                     gb.cpu().reshape(128, 1).numpy()))).permute(0, 2, 1)
    row['max_err_vs_f64'] = float((ref.cpu().reshape(B, N, 128) - want).abs().max())
    print(json.dumps(row), flush=True)
