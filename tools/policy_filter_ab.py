"""A/B of the policy step's filter + head launch (gnnpp_filter_head_fwd): policy_filter_kernel vs the general filter
kernel, one vs two workgroups per graph.  Microseconds per launch (HIP events, back-to-back launches)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_pathplanning_amd import _native
from oracle import policy_oracle as orc                      # (inputs only)

dev = torch.device('cuda:0')
L = _native.lib()
for (N, B, K) in ((50, 256, 3), (100, 128, 3), (100, 128, 2), (100, 128, 4), (100, 256, 3), (64, 128, 3), (20, 512, 3)):
    g = torch.Generator().manual_seed(N + B)
    h = (torch.randn(128, 1, K, 128, generator=g) / (128 * K) ** 0.5).to(dev)
    x = torch.relu(torch.randn(B, N, 128, generator=g)).to(dev)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, N, seed=1)).float().to(dev)
    bias, aw, ab = (torch.randn(128, generator=g) / 4).to(dev), (torch.randn(5, 128, generator=g) / 8).to(dev), \
        torch.randn(5, generator=g).to(dev)
    packed = torch.empty(L.gnnpp_filter_packed_floats(128, 128, K, 1), dtype=torch.float32, device=dev)
    assert L.gnnpp_filter_pack(h.data_ptr(), packed.data_ptr(), 128, 128, K, 1, None) == 0
    lg = torch.empty(N, B, 5, device=dev)
    st = _native.stream_ptr(dev)

    def launch():
        assert L.gnnpp_filter_head_fwd(x.data_ptr(), S.data_ptr(), packed.data_ptr(), bias.data_ptr(), aw.data_ptr(),
                                       ab.data_ptr(), lg.data_ptr(), B, N, 128, 128, K, 1, 0, None, st) == 0

    def timeit(reps=300):
        for _ in range(30):
            launch()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                launch()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
        return round(sorted(ts)[2], 2)

    row = {'N': N, 'B': B, 'K': K}
    try:
        for kern in (1, 0):
            for split in (0, 1, 2):
                L.gnnpp_set_tuning(9, kern); L.gnnpp_set_tuning(7, split)
                row[('policy_filter' if kern else 'general') + '_split%s' % ('auto', '1', '2')[split]] = timeit()
    finally:
        L.gnnpp_set_tuning(9, 1); L.gnnpp_set_tuning(7, 0)
    print(json.dumps(row), flush=True)
