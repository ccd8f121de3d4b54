"""Where does the EAGER two-launch policy step of a small batch spend its time?  (r05: the 16 x 100 shard of config 5 runs
41.8 us of kernels, 47.5 us as a HIP-graph replay and 59-63 us eager.)

  python tools/shard_gap_probe.py            host-side cost of the pieces, enqueue only (short bursts, no queue back-pressure)
  rocprofv3 --kernel-trace --output-format csv -d DIR -o trace -- python tools/shard_gap_probe.py trace
  python tools/shard_gap_probe.py gaps DIR   device-side gaps between consecutive kernels of that trace
"""
import csv
import ctypes
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def gaps(d):
    rows = []
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:48]))
    rows.sort()
    rows = [r for r in rows if 'encoder_kernel_b3' in r[2] or 'policy_filter_kernel' in r[2]]
    rows = rows[len(rows) // 2:]                              # the steady second half
    e2f, f2e, enc, fil = [], [], [], []
    for a, b in zip(rows[:-1], rows[1:]):
        gap = (b[0] - a[1]) * 1e-3
        (e2f if 'encoder' in a[2] else f2e).append(gap)
    for r in rows:
        (enc if 'encoder' in r[2] else fil).append((r[1] - r[0]) * 1e-3)
    med = lambda v: sorted(v)[len(v) // 2] if v else None     # noqa: E731
    print(json.dumps({'what': 'device timeline of the eager 16 x 100 step (rocprofv3 kernel trace, us, medians)',
                      'encoder': med(enc), 'filter': med(fil), 'gap_encoder_to_filter': med(e2f),
                      'gap_filter_to_next_encoder': med(f2e), 'kernels': len(rows)}))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == 'gaps':
        return gaps(sys.argv[2])
    import torch
    from gnn_pathplanning_amd import _native
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from oracle import policy_oracle as orc                  # synthetic inputs only
    dev = torch.device('cuda:0')
    B, N, K = 16, 100, 3

    class Cfg:
        num_agents, nGraphFilterTaps, device = N, K, dev
    net = DecentralPlannerNet(Cfg()).to(dev).eval()
    net.load_state_dict(orc.init_state_dict(K))
    obs = orc.synth_obs(B, N, seed=1).to(dev)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, 100, seed=1)).float().to(dev)
    L = _native.lib()
    vp = lambda t: ctypes.c_void_p(t.data_ptr())             # noqa: E731
    with torch.no_grad():
        for _ in range(100):
            net.addGSO(S)
            net(obs)
        torch.cuda.synchronize()
        if len(sys.argv) > 1 and sys.argv[1] == 'trace':
            for _ in range(400):
                net.addGSO(S)
                net(obs)
            torch.cuda.synchronize()
            return
        enc, taps, gb, aw, ab, _ = net.policy_pointers()
        st = _native.stream_ptr(dev)
        feat = torch.empty(B * N, 128, device=dev)
        lg = torch.empty(N, B, 5, device=dev)
        pieces = {
            'python addGSO + forward (the bench step)': lambda: (net.addGSO(S), net(obs)),
            'C call gnnpp_policy_fwd (both launches)': lambda: L.gnnpp_policy_fwd(
                vp(obs), vp(S), enc, taps, gb, aw, ab, vp(feat), vp(lg), B, N, K, 1, 0, 0, None, st),
            'C call gnnpp_encoder_fwd': lambda: L.gnnpp_encoder_fwd(vp(obs), enc, vp(feat), B * N, 0, None, st),
            'C call gnnpp_filter_head_fwd': lambda: L.gnnpp_filter_head_fwd(
                vp(feat), vp(S), taps, gb, aw, ab, vp(lg), B, N, 128, 128, K, 1, 0, 0, None, st),
            'torch.empty(N, B, 5)': lambda: torch.empty(N, B, 5, device=dev),
        }
        out = {}
        for name, fn in pieces.items():
            best_host, best_total = 1e9, 1e9
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(24):                           # a burst short enough not to fill the HIP queue
                    fn()
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                best_host, best_total = min(best_host, (t1 - t0) / 24), min(best_total, (t2 - t0) / 24)
            out[name] = {'host_us_per_call': round(best_host * 1e6, 2), 'until_device_done_us': round(best_total * 1e6, 2)}
        print(json.dumps({'what': 'host cost of the pieces of the eager 16 x 100 policy step (bursts of 24 calls)', **out}))


if __name__ == '__main__':
    main()
