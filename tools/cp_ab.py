"""A/B of the column-packed one-launch policy kernel (GNNPP_TUNE_POLICY_CP = 1, teams of <= 12 agents) against the
agents-on-columns schedule (= 0) on the same inputs: launch time from HIP events on the launch stream (median of 5
regions of `reps` back-to-back launches of gnnpp_policy_fwd), bit-identity of the logits, and -- with `stamps` -- the
phase time stamps of the -DGNNPP_MEASURE build.  One JSON line per shape.
    python tools/cp_ab.py [stamps]"""
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

dev = torch.device('cuda:0')
st = _native.stream_ptr(dev)
L = _native.lib()
SHAPES = [(512, 10, 3), (256, 10, 3), (512, 12, 3), (512, 8, 3), (512, 6, 3), (512, 10, 2), (512, 10, 4), (64, 10, 3),
          (1, 10, 3), (1024, 10, 3), (2048, 10, 3)]


def time_launches(lib, args, reps):
    for _ in range(20):
        assert lib.gnnpp_policy_fwd(*args) == 0
    out = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            lib.gnnpp_policy_fwd(*args)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / reps)
    return sorted(out)[2]


def main():
    want_stamps = 'stamps' in sys.argv[1:]
    global M
    M = None
    if want_stamps:
        M = _native.measure_lib()
        M.gnnpp_measure_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    if 'tiles' in sys.argv[1:]:
        return encoder_tiles(M)
    for (B, N, K) in SHAPES:
        class Cfg:
            num_agents, nGraphFilterTaps, device = N, K, dev
        net = DecentralPlannerNet(Cfg()).to(dev).eval()
        net.load_state_dict(orc.init_state_dict(K))
        obs = orc.synth_obs(B, N, seed=1337).to(dev)
        S = torch.from_numpy(orc.synth_gso_geometric(B, N, 20, seed=1337)).float().to(dev)
        enc, taps, gb, aw, ab, _ = net.policy_pointers()
        ws = torch.empty(B * N, 128, device=dev)
        row = {'B': B, 'N': N, 'K': K}
        logits = {}
        for cp in (0, 1, 0, 1):
            lg = torch.full((N, B, 5), float('nan'), device=dev)
            args = (obs.data_ptr(), S.data_ptr(), enc, taps, gb, aw, ab, ws.data_ptr(), lg.data_ptr(), B, N, K, 1, 0, 0,
                    None, st)
            assert L.gnnpp_set_tuning(6, 2) == 0 and L.gnnpp_set_tuning(13, cp) == 0
            # (knob 6 = 2: the one-launch kernel whatever the batch size -- the default rule sends B > 512 at N < 13 to
            # the two-kernel path)
            us = time_launches(L, args, 200 if B <= 1024 else 60)
            row.setdefault('cp%d_us' % cp, []).append(round(us, 2))
            logits[cp] = lg.clone()
        row['bit_identical'] = bool(torch.equal(logits[0], logits[1]))
        row['finite'] = bool(torch.isfinite(logits[1]).all())
        row['speedup'] = round(min(row['cp0_us']) / min(row['cp1_us']), 4)
        row['M_agent_steps_per_s_cp1'] = round(B * N / min(row['cp1_us']), 2)
        if M is not None and B <= 1024:
            for cp in (0, 1):
                assert M.gnnpp_set_tuning(13, cp) == 0 and M.gnnpp_set_tuning(6, 2) == 0
                lg = torch.empty(N, B, 5, device=dev)
                args = (obs.data_ptr(), S.data_ptr(), enc, taps, gb, aw, ab, ws.data_ptr(), lg.data_ptr(), B, N, K, 1, 0,
                        0, None, st)
                for _ in range(6):
                    assert M.gnnpp_policy_fwd(*args) == 0
                    torch.cuda.synchronize()
                buf = np.zeros(1024 * 32, np.uint64)
                assert M.gnnpp_measure_read_stamps(buf.ctypes.data, buf.size) == 0
                rows = buf.reshape(1024, 32)[:min(B, 1024)].astype(np.float64)
                us_, cyc = rows[:, :16] * 0.01, rows[:, 16:]
                order = [('start', 11), ('staged', 0), ('L0', 1), ('L1', 2), ('L2', 3), ('L3', 4), ('L4', 5), ('FC', 12),
                         ('shifts', 13), ('contr', 14)]
                ph = {}
                for (na, a), (nb, b) in zip(order[:-1], order[1:]):
                    ph[nb] = round(float(np.median(us_[:, b] - us_[:, a])), 2)
                tot = us_[:, 14] - us_[:, 11]
                ph['total_median'] = round(float(np.median(tot)), 2)
                ph['total_max'] = round(float(np.max(tot)), 2)
                ok = tot > 0
                ph['clock_GHz'] = round(float(np.median((cyc[:, 14] - cyc[:, 11])[ok] / tot[ok])) * 1e-3, 3)
                if cp:
                    ph['L1_addr'] = round(float(np.median(us_[:, 7] - us_[:, 1])), 2)
                    ph['L1_stream_wave0'] = round(float(np.median(us_[:, 8] - us_[:, 7])), 2)
                    ph['L1_barrier'] = round(float(np.median(us_[:, 9] - us_[:, 8])), 2)
                    ph['L1_epilogue'] = round(float(np.median(us_[:, 2] - us_[:, 9])), 2)
                    ph['L2_addr'] = round(float(np.median(us_[:, 15] - us_[:, 2])), 2)
                row['stamps_cp%d' % cp] = ph
            M.gnnpp_set_tuning(13, 1)
        print(json.dumps(row), flush=True)
    L.gnnpp_set_tuning(13, 1)
    L.gnnpp_set_tuning(6, 1)


M = None


def encoder_tiles(M):
    """The unfused encoder in the latency regime (the per-GPU shard of configs 3 / 5: 1 600 agents): 16-agent tiles
    against column-packed tiles of 4 / 7 (= the heuristic) / 10 / 12 agents -- launch time and phase stamps."""
    class Cfg:
        num_agents, nGraphFilterTaps, device = 10, 3, dev
    net = DecentralPlannerNet(Cfg()).to(dev).eval()
    net.load_state_dict(orc.init_state_dict(3))
    enc = net.packed_encoder()
    for Mag in (1600, 640, 3000):
        obs = orc.synth_obs(Mag // 10 + 1, 10, seed=3).reshape(-1, 3, 11, 11)[:Mag].contiguous().to(dev)
        feat = torch.empty(Mag, 128, device=dev)
        row = {'encoder_M': Mag}
        ref = None
        for tile in (16, 0, 4, 7, 10, 12):
            assert L.gnnpp_set_tuning(14, tile) == 0
            args = (obs.data_ptr(), enc.data_ptr(), feat.data_ptr(), Mag, 0, None, st)
            for _ in range(20):
                assert L.gnnpp_encoder_fwd(*args) == 0
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(200):
                    L.gnnpp_encoder_fwd(*args)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / 200)
            row['tile%d_us' % tile] = round(sorted(ts)[2], 2)
            if ref is None:
                ref = feat.clone()
            else:
                row['tile%d_bit_identical' % tile] = bool(torch.equal(ref, feat))
            if M is not None:
                assert M.gnnpp_set_tuning(14, tile) == 0
                for _ in range(6):
                    assert M.gnnpp_encoder_fwd(*args) == 0
                    torch.cuda.synchronize()
                tl = tile if tile not in (0, 16) else (16 if tile == 16 else max(1, (Mag + 255) // 256))
                nwg = min((Mag + tl - 1) // tl, 1024)
                buf = np.zeros(1024 * 32, np.uint64)
                assert M.gnnpp_measure_read_stamps(buf.ctypes.data, buf.size) == 0
                rows = buf.reshape(1024, 32)[:nwg].astype(np.float64)
                us_ = rows[:, :16] * 0.01
                order = [('start', 11), ('staged', 0), ('L0', 1), ('L1', 2), ('L2', 3), ('L3', 4), ('L4', 5), ('FC', 6)]
                ph = {nb: round(float(np.median(us_[:, b] - us_[:, a])), 2) for (na, a), (nb, b) in zip(order[:-1], order[1:])}
                ph['total_median'] = round(float(np.median(us_[:, 6] - us_[:, 11])), 2)
                ph['span'] = round(float(us_[:, 6].max() - us_[:, 11].min()), 2)
                row['stamps_tile%d' % tile] = ph
                M.gnnpp_set_tuning(14, 0)
        L.gnnpp_set_tuning(14, 0)
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
