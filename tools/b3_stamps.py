"""Where the time goes INSIDE the policy kernels, per arithmetic (precision 0 = bf16x3 default, 2 = split-f16):
phase time stamps of the -DGNNPP_MEASURE build (csrc/gnnpp_common.h GNNPP_STAMP: 100 MHz wall clock + shader cycle
counter), median over the workgroups of one launch.  Prints one JSON line per (kernel, shape, precision)."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_pathplanning_amd import _native                      # noqa: E402
from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet   # noqa: E402
from oracle import policy_oracle as orc                       # noqa: E402  (inputs only)

M = _native.measure_lib()
M.gnnpp_measure_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device('cuda:0')
st = _native.stream_ptr(dev)


def stamps(nwg):
    buf = np.zeros(1024 * 32, np.uint64)
    assert M.gnnpp_measure_read_stamps(buf.ctypes.data, buf.size) == 0
    rows = buf.reshape(1024, 32)[:min(nwg, 1024)].astype(np.float64)
    return rows[:, :16] * 0.01, rows[:, 16:]                  # microseconds (100 MHz wall clock), shader cycles


def report(tag, nwg, order):
    us, cyc = stamps(nwg)
    row = {'what': tag}
    for (na, a), (nb, b) in zip(order[:-1], order[1:]):
        row['%s->%s' % (na, nb)] = round(float(np.median(us[:, b] - us[:, a])), 2)
    a, b = order[0][1], order[-1][1]
    tot = us[:, b] - us[:, a]
    row['total_median'] = round(float(np.median(tot)), 2)
    row['total_max'] = round(float(np.max(tot)), 2)
    row['span_first_start_to_last_end'] = round(float(us[:, b].max() - us[:, a].min()), 2)
    ok = tot > 0
    row['engine_clock_GHz'] = round(float(np.median((cyc[:, b] - cyc[:, a])[ok] / tot[ok])) * 1e-3, 3)
    print(json.dumps(row), flush=True)


ENC = [('start', 11), ('staged', 0), ('L0', 1), ('L1', 2), ('L2', 3), ('L3', 4), ('L4', 5)]
which = sys.argv[1:] or ['fused', 'encoder', 'filter']
if 'fused' in which:
    class Cfg2:
        num_agents, nGraphFilterTaps, device = 10, 3, dev
    net = DecentralPlannerNet(Cfg2()).to(dev).eval()
    net.load_state_dict(orc.init_state_dict(3))
    N = 10
    for prec in (0, 2):
        for B in (512, 256):
            obs = orc.synth_obs(B, N, seed=1337).to(dev)
            S = torch.from_numpy(orc.synth_gso_geometric(B, N, 20, seed=1337)).float().to(dev)
            enc, taps, gb, aw, ab, K = net.policy_pointers()
            ws = torch.empty(B * N, 128, device=dev)
            lg = torch.empty(N, B, 5, device=dev)
            for _ in range(8):
                assert M.gnnpp_policy_fwd(obs.data_ptr(), S.data_ptr(), enc, taps, gb, aw, ab, ws.data_ptr(),
                                          lg.data_ptr(), B, N, 3, 1, 0, prec, None, st) == 0
                torch.cuda.synchronize()
            report('fused policy kernel prec=%d B=%d N=10 (%d wg/CU)' % (prec, B, B // 256), B,
                   ENC + [('FC(z0)', 12), ('shifts', 13), ('contraction', 14)])
if 'encoder' in which:
    class Cfg3:
        num_agents, nGraphFilterTaps, device = 50, 3, dev
    net = DecentralPlannerNet(Cfg3()).to(dev).eval()
    net.load_state_dict(orc.init_state_dict(3))
    enc = net.packed_encoder()
    for prec in (0,):
        for Mag in (16 * 256, 16 * 512, 12800, 16 * 1024):
            obs = orc.synth_obs(Mag // 50 + 1, 50, seed=3).reshape(-1, 3, 11, 11)[:Mag].contiguous().to(dev)
            feat = torch.empty(Mag, 128, device=dev)
            for _ in range(8):
                assert M.gnnpp_encoder_fwd(obs.data_ptr(), enc.data_ptr(), feat.data_ptr(), Mag, prec, None, st) == 0
                torch.cuda.synchronize()
            report('encoder kernel prec=%d M=%d (%d tiles)' % (prec, Mag, (Mag + 15) // 16), (Mag + 15) // 16,
                   ENC + [('FC', 6)])
if 'filter' in which:
    # (N, graphs, forced GNNPP_TUNE_FILTER_SPLIT; 0 = the heuristic): the full batches of configs 3 / 5 and the 16-graph
    # shard one GPU of eight holds of config 5, as two parts (v310's rule) and as the heuristic's one part per row tile
    for (N, B, split) in ((50, 256, 0), (100, 128, 0), (100, 16, 2), (100, 16, 0)):
        class Cfg4:
            num_agents, nGraphFilterTaps, device = N, 3, dev
        net = DecentralPlannerNet(Cfg4()).to(dev).eval()
        net.load_state_dict(orc.init_state_dict(3))
        S = torch.from_numpy(orc.synth_gso_geometric(B, N, N, seed=1337)).float().to(dev)
        x = torch.relu(torch.randn(B * N, 128, device=dev))
        enc, taps, gb, aw, ab, K = net.policy_pointers()
        lg = torch.empty(N, B, 5, device=dev)
        for prec in (0, 1, 2):
            M.gnnpp_set_tuning(7, split)
            for _ in range(8):
                assert M.gnnpp_filter_head_fwd(x.data_ptr(), S.data_ptr(), taps, gb, aw, ab, lg.data_ptr(), B, N, 128,
                                               128, 3, 1, 0, prec, None, st) == 0
                torch.cuda.synchronize()
            M.gnnpp_set_tuning(7, 0)
            if prec == 2:
                order = [('entry', 0), ('staged', 1), ('lists', 2), ('tap0', 3), ('barrier1', 4), ('shift1', 5),
                         ('barrier2', 6), ('shift2', 7), ('split1', 8), ('tap1', 9), ('tap2', 12), ('partial', 13),
                         ('stored', 14)]
            else:
                order = [('entry', 0), ('staged', 1), ('lists', 2), ('tap0', 3), ('barrier1', 4), ('shift1+barrier', 5),
                         ('tap1 issued', 6), ('barrier(k=2)', 7), ('shift2', 8), ('barrier', 9),
                         ('tap2 issued', 12), ('partial', 13), ('stored', 14)]
            rt, groups = (N + 15) // 16, (B + 7) // 8        # csrc/lsigf_kernel.hip lsigf_plan's rule
            ns = min(split, rt) if split else (max(2, min(256 // (8 * groups), rt)) if B <= 128 and rt >= 4 else 1)
            report('policy_filter_kernel prec=%d B=%d N=%d parts=%d' % (prec, B, N, ns),
                   min(1024, B if ns == 1 else groups * 8 * ns), order)

if 'small' in which:
    from gnn_pathplanning_amd.graphML import pack_filter_taps
    N, K = 10, 3
    h = (torch.randn(128, 1, K, 128) / (128 * K) ** 0.5).to(dev)
    taps = pack_filter_taps(h)
    bias = torch.zeros(128, device=dev)
    for B in (1024, 2048, 8192):
        S = torch.from_numpy(orc.synth_gso_geometric(512, N, 20, seed=1337)).float().to(dev).repeat(B // 512, 1, 1).contiguous()
        x = torch.relu(torch.randn(B * N, 128, device=dev))
        y = torch.empty_like(x)
        M.gnnpp_set_tuning(10, 2)
        for _ in range(6):
            assert M.gnnpp_lsigf_fwd(x.data_ptr(), S.data_ptr(), taps.data_ptr(), bias.data_ptr(), y.data_ptr(), B, N, N,
                                     128, 128, K, 1, 0, 1, 1, 1, 1, 0, 0, None, st) == 0
            torch.cuda.synchronize()
        M.gnnpp_set_tuning(10, 1)
        report('lsigf_small_b3_kernel B=%d N=10 K=3 (%d workgroups; stamps of the first 1024)' % (B, (B + 3) // 4),
               min(1024, (B + 3) // 4),
               [('entry', 0), ('staged', 1), ('tap0', 2), ('barrier', 3), ('shift1', 4), ('tap1', 5), ('barrier', 6),
                ('shift2', 7), ('tap2', 8), ('y in lds', 12), ('stored', 13)])

if 'pipe' in which:
    from gnn_pathplanning_amd.graphML import pack_filter_taps
    N, K = 10, 3
    h = (torch.randn(128, 1, K, 128) / (128 * K) ** 0.5).to(dev)
    taps = pack_filter_taps(h)
    bias = torch.zeros(128, device=dev)
    for B in (3072, 8192, 32768):
        S = torch.from_numpy(orc.synth_gso_geometric(512, N, 20, seed=1337)).float().to(dev).repeat(B // 512, 1, 1).contiguous()
        x = torch.relu(torch.randn(B * N, 128, device=dev))
        y = torch.empty_like(x)
        M.gnnpp_set_tuning(10, 3)
        for _ in range(6):
            assert M.gnnpp_lsigf_fwd(x.data_ptr(), S.data_ptr(), taps.data_ptr(), bias.data_ptr(), y.data_ptr(), B, N, N,
                                     128, 128, K, 1, 0, 1, 1, 1, 1, 0, 0, None, st) == 0
            torch.cuda.synchronize()
        M.gnnpp_set_tuning(10, 1)
        report('lsigf_pipe_b3_kernel B=%d N=10 K=3, a SHIFT stage (tap 0 of the last group): producer start -> shift done | '
               'consumer contraction done | barrier passed' % B, 256,
               [('stage start', 0), ('producer done', 1), ('barrier', 3)])
        report('   same stage, consumer', 256, [('stage start', 0), ('contraction done', 2), ('barrier', 3)])
        report('   same stage, producer wave 4 inside shift_group', 256, [('stage start', 0), ('loop entry', 8), ('unit 0: MFMAs done', 9), ('unit 0: written', 10), ('all units', 1)])
        report('lsigf_pipe_b3_kernel B=%d: a STAGING stage (last tap of the group before)' % B, 256,
               [('stage start', 4), ('producer done', 5), ('barrier', 7)])
        report('   same stage, consumer', 256, [('stage start', 4), ('contraction done', 6), ('barrier', 7)])
