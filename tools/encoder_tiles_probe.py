"""Encoder launch time against the number of 16-agent tiles (one workgroup each): separates per-workgroup
latency (flat up to 256 / 512 tiles) from throughput.  tools/ only; prints one JSON line per size."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_pathplanning_amd import _native                      # noqa: E402
from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet  # noqa: E402


def main():
    dev = torch.device('cuda:0')

    class Cfg:
        num_agents, nGraphFilterTaps, device = 10, 3, dev

    net = DecentralPlannerNet(Cfg()).to(dev).eval()
    lib = _native.lib()
    packed = net.packed_encoder()
    for tiles in (64, 128, 256, 384, 512, 640, 768, 800, 1024, 1536, 2048):
        M = tiles * 16
        obs = torch.rand(M, 3, 11, 11, device=dev)
        feat = torch.empty(M, 128, device=dev)
        st = torch.cuda.current_stream().cuda_stream

        def run():
            rc = lib.gnnpp_encoder_fwd(ctypes.c_void_p(obs.data_ptr()), ctypes.c_void_p(packed.data_ptr()),
                                       ctypes.c_void_p(feat.data_ptr()), M, None, ctypes.c_void_p(st))
            assert rc == 0, rc
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        best = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                run()
            e1.record()
            torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1) * 10.0)
        best.sort()
        print(json.dumps({'tiles': tiles, 'us_per_launch': round(best[2], 2),
                          'ns_per_tile': round(best[2] * 1e3 / tiles, 1)}), flush=True)


if __name__ == '__main__':
    main()
