"""Filter-only throughput sweep (the north star's "achieved-HBM-fraction on synthetic random GSO+feature batches"):
gnnpp_lsigf_fwd on B graphs of N nodes, G = F = 128, K taps, node-major rows, bias + ReLU fused, default precision;
the throughput kernel (lsigf_small_b3_kernel) against the general filter kernel.  One JSON line per (B, kernel).
    python tools/filter_sweep.py [--pmc-target B]     (the second form: a few launches at one batch, for rocprofv3 --pmc)"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_pathplanning_amd import _native                      # noqa: E402
from gnn_pathplanning_amd.graphML import pack_filter_taps     # noqa: E402
from oracle import policy_oracle as orc                       # noqa: E402  (inputs only)

L = _native.lib()
dev = torch.device('cuda:0')
st = _native.stream_ptr(dev)
N, K = 10, 3
g = torch.Generator().manual_seed(1337)
h = (torch.randn(128, 1, K, 128, generator=g) / (128 * K) ** 0.5).to(dev)
taps = pack_filter_taps(h)
bias = (torch.randn(128, generator=g) / 4).to(dev)
S0 = torch.from_numpy(orc.synth_gso_geometric(512, N, 20, seed=1337)).float().to(dev)
mean_deg = float((S0 != 0).sum() / (512 * N))


def run(B, mode, reps, rows=0):
    S = S0.repeat((B + 511) // 512, 1, 1)[:B].contiguous()
    x = torch.relu(torch.randn(B * N, 128, generator=torch.Generator().manual_seed(B))).to(dev)
    y = torch.empty_like(x)
    assert L.gnnpp_set_tuning(10, mode) == 0 and L.gnnpp_set_tuning(11, rows) == 0
    call = lambda: L.gnnpp_lsigf_fwd(x.data_ptr(), S.data_ptr(), taps.data_ptr(), bias.data_ptr(), y.data_ptr(), B, N, N,  # noqa: E731
                                     128, 128, K, 1, 0, 1, 1, 1, 1, 0, 0, None, st)
    for _ in range(5):
        assert call() == 0
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3 / reps)
    L.gnnpp_set_tuning(10, 1)
    L.gnnpp_set_tuning(11, 0)
    return sorted(ts)[2], y


if len(sys.argv) > 2 and sys.argv[1] == '--pmc-target':
    rows = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    if len(sys.argv) > 4:                                 # access-ablation mask of the pipeline kernel (measure build)
        L = _native.measure_lib()
        assert L.gnnpp_set_tuning(3, int(sys.argv[4], 0)) == 0
    run(int(sys.argv[2]), (3 if rows == 64 else 2) if rows else 1, 4, 0 if rows == 64 else rows)
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[1] == '--ablate-times':
    # launch time of the pipeline kernel with single LDS accesses removed (results are wrong by construction; what is
    # measured is what each access costs): 0x10 plane writes of the shift, 0x20 z write-back, 0x40 A-operand reads of the
    # shift, 0x80 the consumers' plane reads, 0x100 / 0x200 staging writes (fp32 rows / planes)
    L = _native.measure_lib()
    B = int(sys.argv[2])
    for mask in (0, 0x10, 0x20, 0x40, 0x80, 0x100, 0x200, 0x30, 0x70, 0x3f0):
        assert L.gnnpp_set_tuning(3, mask) == 0
        t, _ = run(B, 3, 20)
        print(json.dumps({'batch': B, 'kernel': 'lsigf_pipe_b3_kernel (measure build)', 'ablate_mask': hex(mask),
                          'us': round(t * 1e6, 2)}), flush=True)
    L.gnnpp_set_tuning(3, 0)
    sys.exit(0)
for B in (512, 2048, 8192, 32768, 131072):
    ref = None
    for mode, rows, name in ((1, 0, 'default (heuristic: small-graph kernel; pipeline kernel from 4096 groups on)'),
                             (2, 32, 'small-graph kernel, 32 rows'), (2, 48, 'small-graph kernel, 48 rows'),
                             (3, 0, 'pipeline kernel (8-wave producer / consumer, 64-row groups)'),
                             (0, 0, 'general filter kernel')):
        reps = max(3, min(200, int(4e6 / (B * N))))
        t, y = run(B, mode, reps, rows)
        fb = B * N * (1024 + 4 * N) + 196608.0 * K / 3
        ffl = 2.0 * (K * 128 * 128 + (K - 1) * mean_deg * 128) * B * N
        rec = {'batch': B, 'agents': N, 'taps': K, 'kernel': name, 'us': round(t * 1e6, 2),
               'agent_steps_per_s': B * N / t, 'algorithmic_GBps': fb / t / 1e9, 'hbm_frac_of_8TBps': fb / t / 8e12,
               'algorithmic_TFLOPs': ffl / t / 1e12, 'frac_of_bf16x3_ceiling_417TF': ffl / t / 1e12 / (2500.0 / 6)}
        if ref is None:
            ref = y.clone()
        else:
            rec['max_abs_diff_vs_default'] = float((y - ref).abs().max())
        print(json.dumps(rec), flush=True)
