"""Headline benchmark: agent-steps/s of the policy forward pass (BASELINE.json metric).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...         (no launcher: bench.py starts the N ranks itself, self_launch())

A "step" is one policy forward (DecentralPlannerNet.addGSO + forward) over one batch of synthetic
input resident in HBM: BASELINE.json configs[1] = 10 agents, K=3, B=512 GSO+observation batch
(seed 1337).  With N GPUs every rank runs its own replica on its own batch
(independent rollout shards, no data-path collective): weak scaling, value = total agent-steps/s.

Timing: W warm-up steps, then `--repeats` timed regions of EXACTLY K steps each, every region
bracketed by barrier + torch.cuda.synchronize() on both sides, max over ranks per region; the
reported region is the MEDIAN one (all of them are listed under `regions_ms`).  HIP events recorded
on the launch stream around the same regions give the device-side time the roofline uses, so the
kernel time can never exceed the step time it is part of.

Rank 0 prints ONE JSON line of at most 6 KB (driver_line(): the contract keys, `roofline`, `cpu_baseline`, `parity`,
`summary`, and `details_file` = the side file holding everything else: `secondary`, sweeps, shard records, notes).  `value`, `ms_per_step`, `dtype`, `roofline` and `parity` all describe the DEFAULT
precision ("fp32": every fp32 operand exactly as three bf16 planes, six plane products on the bf16 MFMA, fp32
accumulate -- no input domain, nothing narrower than the reference's fp32).  Besides the contract fields:
  roofline      dominant kernel: ALGORITHMIC fp32 FLOPs per launch / launch time (HIP events on the launch stream,
                inside the timed regions for the one-kernel step) / the dense peak of the instruction actually
                issued (MI355X_MICROARCH.md); the number of MFMA products one fp32 MAC costs is stated beside it
                (`mfma_products_per_fp32_mac`), never folded into `peak`; `traffic` = HBM bytes per launch from
                rocprofv3 --pmc passes run by THIS process (separate FETCH_SIZE / WRITE_SIZE passes, gfx950
                correction), or null when rocprofv3 is not usable
  cpu_baseline  the CPU oracle (op-for-op restatement of the reference's PyTorch path, pinned to golden vectors)
                timed on this box's host cores (bounded sample)
  parity        max |dlogit| and action-id agreement GPU vs oracle on the bench batch, near-tie rows listed;
                `range_flag` as read back from the device
  secondary     (side file only) the other two precisions (exact fp32 MFMA; opt-in split-f16, labelled narrower than fp32) with their
                own roofline blocks, the remaining single-GPU configurations (C3, C5 at K = 2, 3, 4), a
                rotating-batch variant of C2 (64 distinct batches, > 256 MB: not cache-resident), batch sweeps,
                the filter-only HBM fraction, the argmax-D2H-inclusive rate, the rollout step
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (agents N, map W, taps K, batch B)          -- BASELINE.json configs[1], [2], [4]
    'c2': (10, 20, 3, 512),
    'c3': (50, 50, 3, 256),
    'c5': (100, 100, 3, 128),
}
FP32_MFMA_PEAK_TFLOPS = 157.3           # MI355X_MICROARCH.md, chip-level parameters
F16_MFMA_PEAK_TFLOPS = 2500.0           # dense f16/bf16 MFMA peak (~2.5 PF), same table
HBM_PEAK_TBPS = 8.0
NOMINAL_CLOCK_GHZ = 2.4                 # engine clock the peaks above are quoted at
ENC_MACS_PER_AGENT = 1238112 + 16384    # CNN + compress MLP
ENC_WEIGHT_FLOATS = 156288              # conv + BN-folded scale/shift + FC, read once per launch


# The three arithmetics of the matrix-pipe contractions (include/gnnpp.h GNNPP_PREC_*): what the roofline prices.
# `peak` is the guide's dense peak of the MFMA instruction actually issued; `products` = MFMA products issued per
# algorithmic fp32 multiply-add (NOT folded into the peak): the ceiling of `frac` for algorithmic fp32 FLOPs on that
# instruction is 1 / products.
PRECISION_INFO = {
    0: {'name': 'fp32', 'instr': 'v_mfma_f32_16x16x32_bf16', 'peak': F16_MFMA_PEAK_TFLOPS, 'products': 6,
        'dtype': 'f32 (fp32-equivalent: every operand exactly as three bf16 planes, 6 of 9 plane products on the bf16 '
                 'MFMA pipe, f32 accumulate; graph shifts / epilogues / head in exact f32)'},
    1: {'name': 'fp32_mfma', 'instr': 'v_mfma_f32_16x16x4_f32', 'peak': FP32_MFMA_PEAK_TFLOPS, 'products': 1,
        'dtype': 'f32 (exact fp32 MFMA)'},
    2: {'name': 'split_f16', 'instr': 'v_mfma_f32_16x16x32_f16', 'peak': F16_MFMA_PEAK_TFLOPS, 'products': 3,
        'dtype': 'f32 operands as f16 hi+lo pairs (22 significand bits, |x| < 65504): NARROWER than f32'},
}


def roofline_block(kernel, info, flops, t_launch, exe_flops, lanes='', column_fill=1.0):
    """frac = algorithmic fp32 FLOPs per launch / launch time / the guide's dense peak of the instruction issued.
    `executed_over_algorithmic_flops` and `column_fill` say why frac sits where it does: MFMA FLOPs issued per
    algorithmic FLOP (plane products x tile padding x skipped structural zeros), and the share of the MFMA tiles'
    16 columns that carry an agent (or an (agent, position) pair)."""
    ach = flops / t_launch / 1e12
    return {'kernel': kernel, 'bound': 'mfma', 'dtype': info['dtype'], 'instruction': info['instr'],
            'achieved': ach, 'peak': info['peak'], 'unit': 'TFLOP/s', 'frac': ach / info['peak'],
            'mfma_products_per_fp32_mac': info['products'],
            'frac_ceiling_for_this_arithmetic': 1.0 / info['products'],
            'frac_of_arithmetic_ceiling': ach / (info['peak'] / info['products']),
            'peak_note': ('dense %s peak of MI355X_MICROARCH.md; algorithmic fp32 FLOPs cost %d MFMA product(s) '
                          'each, which is stated here and NOT folded into the peak' % (info['instr'], info['products']))
                         + lanes,
            'vs_fp32_mfma_peak': ach / FP32_MFMA_PEAK_TFLOPS,
            'avg_launch_us': t_launch * 1e6, 'flops_per_launch': flops,
            'executed_mfma_flops_per_launch': exe_flops,
            'executed_over_algorithmic_flops': exe_flops / flops, 'column_fill': column_fill,
            'pipe_busy_frac': exe_flops / t_launch / 1e12 / info['peak']}


CP_MAX_AGENTS = 12                      # csrc/encoder_kernel_b3.hip kCpMaxAgents


def fused_mfma_per_graph(N, K, cp):
    """bf16 MFMAs the one-launch policy kernel issues per graph ({0, 1} observations: L0 issues 3 of its 6 plane
    products) and the share of those MFMAs' columns that carry an agent / an (agent, position) pair.  Agents on the
    columns: 8028 + 192 K whatever N.  Column-packed (N <= 12, GNNPP_TUNE_POLICY_CP): L1 = ceil(25 N / 16) tiles x 9
    taps x 12, L2 = N tiles x 9 taps x 24 (no tap can be skipped at compile time any more)."""
    l0, late = 600, 768 + 1536 + 192 + 192 * K
    if cp and N <= CP_MAX_AGENTS:
        t1 = (25 * N + 15) // 16
        l1, l2 = t1 * 9 * 12, N * 9 * 24
        fill = (l0 * N / 16.0 + l1 * (25.0 * N) / (16 * t1) + l2 + late * N / 16.0) / (l0 + l1 + l2 + late)
    else:
        l1, l2, fill = 2028, 2904, N / 16.0
    return l0 + l1 + l2 + late, fill


def fused_rule(L, B, N, K, prec):
    """gnnpp_policy_fwd's rule for the one-launch policy kernel (csrc/gnnpp_api.hip fused_policy_applies)."""
    knob = L.gnnpp_get_tuning(6)                             # 0 = never, 1 = the rule, 2 = whenever the kernel applies
    return bool(knob != 0 and prec != 1 and N <= 16 and 2 <= K <= 4 and (B <= 512 or N >= 13 or knob == 2))


FILTER_HEAD_MODES = {0: 'split-f16 planes', 1: 'exact fp32 MFMA', 2: 'bf16x3 planes (own LDS buffer)',
                     3: 'bf16x3 planes aliased onto the dead z buffer', -1: 'general filter kernels'}


def filter_head_roofline(L, N, K, B, mean_deg, prec, t):
    """Roofline figures of the policy step's SECOND launch (features -> logits: K-tap filter + bias + ReLU + action
    head; csrc/policy_filter_kernel.hip for teams of 17 .. 100 agents).  Which matrix instruction contracts the taps is
    ASKED of the library (gnnpp_filter_head_mode: it depends on the team size AND, through the workgroups-per-graph
    heuristic, on the batch -- ADVICE r05): bf16x3 planes (six v_mfma_f32_16x16x32_bf16 per 32 channels) in a buffer of
    their own (mode 2) or aliased onto the dead z buffer (mode 3, r06: split teams of 65 .. 100 agents), the exact
    v_mfma_f32_16x16x4_f32 (mode 1: 'fp32_mfma', and one-workgroup teams of 65 .. 100 agents), split-f16 (mode 0).
    `frac` = algorithmic FLOP/s over the dense peak of THAT instruction; `pipe_busy_frac` = executed MFMA FLOP/s (plane
    products, rows padded to 16-row tiles) over the same peak."""
    alg = 2.0 * (K * 128 * 128 + (K - 1) * mean_deg * 128 + 640) * B * N
    tiles = (N + 15) // 16
    mode = L.gnnpp_filter_head_mode(B, N, K, prec)
    if mode == 0 or (mode < 0 and prec == 2):
        instr, peak, products = 'v_mfma_f32_16x16x32_f16', 2500.0, 3
    elif mode in (2, 3) or (mode < 0 and prec == 0 and N <= 16):
        instr, peak, products = 'v_mfma_f32_16x16x32_bf16', 2500.0, 6
    else:
        instr, peak, products = 'v_mfma_f32_16x16x4_f32', 157.3, 1
    exe = 2.0 * products * K * 128 * 128 * 16 * tiles * B + 2.0 * 640 * 16 * tiles * B * (16 / 5.0)
    return {'us': t * 1e6, 'instruction': instr, 'peak_TFLOPs': peak, 'algorithmic_TFLOPs': alg / t / 1e12,
            'frac': alg / t / 1e12 / peak, 'pipe_busy_frac': exe / t / 1e12 / peak,
            'mode': mode, 'schedule': FILTER_HEAD_MODES.get(mode, '?'),
            'note': 'latency-scheduled kernel: one or a few workgroups per graph, every global load issued up front'}


def quick_config(orc, L, _native, name, k_over, dev, timed_regions, time_kernel, vp, st, batch=None):
    """Compact record of another BASELINE config inside the C2 line: value, ms/step, dominant kernel us, frac,
    parity (max |dlogit| vs the oracle, near-ties).  Default precision.  batch: run only that many graphs of the
    config (the per-GPU shard of a strong-scaling run: sharding.shard_batch of the global batch, shard 0)."""
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from gnn_pathplanning_amd.sharding import shard_batch
    N, W, K, B = CONFIGS[name]
    K = k_over or K

    class Cfg:
        num_agents, nGraphFilterTaps, device = N, K, dev
    sd = orc.init_state_dict(K, seed=1337)
    net = DecentralPlannerNet(Cfg()).to(dev).eval()
    net.load_state_dict(sd)
    obs_cpu = orc.synth_obs(B, N, seed=1337)
    S64 = orc.synth_gso_geometric(B, N, W, seed=1337)
    S_cpu = torch.from_numpy(S64).float()
    if batch is not None and batch < B:                      # shard 0 of B / batch ranks of the SAME global batch
        obs_cpu, S_cpu = (t.contiguous() for t in shard_batch((obs_cpu, S_cpu), 0, B // batch))
        B = obs_cpu.shape[0]
    mean_deg = float((S_cpu != 0).sum() / (B * N))
    obs, S = obs_cpu.to(dev), S_cpu.to(dev)
    M = B * N

    def step():
        net.addGSO(S)
        return net(obs)
    for _ in range(15):
        out = step()
    # median of 5 regions of 40 steps (r04: 3 x 30 -- one slow region made `c5_shard_K4` read 17 M where two builder
    # sessions read 24-25 M: VERDICT r04 item 4)
    nst, nreg = 40, 5
    r = sorted(x[0] for x in timed_regions(step, nst, nreg, collective=False))[nreg // 2] / nst
    # the same step replayed from a HIP graph (what a rollout harness with static input buffers can do): without the host's
    # enqueue cost, which a two-launch step of ~55 us of device time does not hide
    graphed = None
    if batch is not None:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg):
                out_g = step()
            for _ in range(5):
                cg.replay()
            rg = sorted(x[0] for x in timed_regions(cg.replay, nst, nreg, collective=False))[nreg // 2] / nst
            same = all(torch.equal(a_, b_) for a_, b_ in zip(out_g, out))
            graphed = {'value': M / rg, 'ms_per_step': 1e3 * rg, 'bit_identical_to_eager': bool(same)}
        except Exception as e:                             # an extra: never break the record
            graphed = {'error': '%s: %s' % (type(e).__name__, e)}
    # The reference's return type is a python LIST of N tensors [B,5] (decentralplanner.py:304-318): at N = 100 building
    # those 100 view objects costs the host ~30 us per step -- more than the shard's kernels leave room for (r05:
    # profiles/r05_shard_gap_probe.jsonl: the C call enqueues in 8.8 us and the device needs 42.6 us per step, the python
    # step 58 us).  forward_logits() hands out the same logits as ONE [N,B,5] tensor (what forward() unbinds, and what
    # rollout.BatchedRollout consumes): the eager rate of a caller that does not need the list.
    stacked = None
    if batch is not None:
        def step_stacked():
            net.addGSO(S)
            return net.forward_logits(obs)
        for _ in range(10):
            out_s = step_stacked()
        rs = sorted(x[0] for x in timed_regions(step_stacked, nst, nreg, collective=False))[nreg // 2] / nst
        stacked = {'value': M / rs, 'ms_per_step': 1e3 * rs,
                   'bit_identical_to_forward': bool(all(torch.equal(a_, b_) for a_, b_ in zip(out_s.unbind(0), out))),
                   'what': 'addGSO + forward_logits(): one [N,B,5] tensor instead of the list of N views'}
    prec = net._prec()
    enc = net.packed_encoder()
    feat = torch.empty(M, 128, device=dev)
    t_enc = time_kernel(lambda: L.gnnpp_encoder_fwd(vp(obs), vp(enc), vp(feat), M, prec, None, st), 30)
    gf, act = net.GFL[0], net.actionsMLP[0]
    aw, ab = act.weight.detach().contiguous(), act.bias.detach().contiguous()
    gbias, taps = gf.bias.detach().reshape(-1).contiguous(), gf.packed_taps()
    lg = torch.empty(N, B, 5, device=dev)
    t_fh = time_kernel(lambda: L.gnnpp_filter_head_fwd(vp(feat), vp(S), vp(taps), vp(gbias), vp(aw), vp(ab), vp(lg),
                                                       B, N, 128, 128, K, 1, 0, prec, None, st), 30)
    tiles = (M + 15) // 16
    info = PRECISION_INFO[prec]
    rl = roofline_block('gnnpp::encoder_kernel_b3<false, 3>', info, 2.0 * ENC_MACS_PER_AGENT * M, t_enc,
                        8028 * 16384.0 * tiles)       # ({0, 1} observations: L0 issues 3 of 6 plane products)
    fh = filter_head_roofline(L, N, K, B, mean_deg, prec, t_fh)
    with torch.no_grad():
        want = orc.policy_forward(sd, S_cpu, obs_cpu)
    got = [o.cpu() for o in out]
    err = max((g - w).abs().max().item() for g, w in zip(got, want))
    margin = orc.top2_margin(want)
    clear = margin > 1e-5
    ids_w = orc.decode_actions(want)
    ids_g = torch.stack([g.argmax(-1) for g in got], 1)
    rec_extra = {'hip_graph_replay': graphed} if graphed is not None else {}
    if stacked is not None:
        rec_extra['forward_logits_eager'] = stacked
    rec_extra['filter_and_head'] = fh
    return {**rec_extra, 'agents': N, 'taps': K, 'batch': B, 'mean_degree': round(mean_deg, 3),
            'value': M / r, 'unit': 'agent-steps/s', 'ms_per_step': 1e3 * r,
            'dominant_kernel': rl['kernel'], 'dominant_kernel_us': t_enc * 1e6, 'frac': rl['frac'],
            'frac_of_arithmetic_ceiling': rl['frac_of_arithmetic_ceiling'], 'pipe_busy_frac': rl['pipe_busy_frac'],
            'filter_and_head_us': 1e6 * t_fh,
            'how': 'whole step: wall clock, median of %d regions of %d steps; the two kernels: HIP events around '
                   'back-to-back launches' % (nreg, nst),
            'parity_max_abs_dlogit': err, 'near_tie_rows': int((~clear).sum()),
            'argmax_equal_on_clear_rows': bool(torch.equal(ids_g[clear], ids_w[clear]))}


def rotating_batches(orc, net, dev, N, W, B, timed_regions, resident_value, nb=64):
    """C2 with `nb` DISTINCT resident batches visited round-robin (> 256 MB in total: neither L2 nor the Infinity
    Cache can hold them), against the headline's one L2-warm batch."""
    base_o = orc.synth_obs(B, N, seed=4242).to(dev)
    base_S = torch.from_numpy(orc.synth_gso_geometric(B, N, W, seed=4242)).float().to(dev)
    g = torch.Generator(device='cpu').manual_seed(99)
    obs_all = torch.empty(nb, *base_o.shape, device=dev)
    S_all = torch.empty(nb, *base_S.shape, device=dev)
    for i in range(nb):                                      # distinct contents: a per-batch permutation of the graphs
        perm = torch.randperm(B, generator=g).to(dev)
        obs_all[i] = base_o[perm]
        S_all[i] = base_S[perm]
    total_mb = (obs_all.numel() + S_all.numel()) * 4 / 1e6
    state = {'i': 0}

    def step():
        i = state['i'] = (state['i'] + 1) % nb
        net.addGSO(S_all[i])
        return net(obs_all[i])
    for _ in range(nb):
        step()
    nst = 2 * nb
    r = sorted(x[0] for x in timed_regions(step, nst, 3, collective=False))[1] / nst
    return {'batches': nb, 'resident_MB': round(total_mb, 1), 'agent_steps_per_s': B * N / r, 'ms_per_step': 1e3 * r,
            'vs_single_resident_batch': (B * N / r) / resident_value,
            'what': '%d distinct batches (%.0f MB of observations + GSOs) round-robin: every step reads its inputs '
                    'from HBM' % (nb, total_mb)}



def c4_shard_record(orc, dev, cpu_seconds):
    """The per-GPU shard of BASELINE config 4 (dcp_onlineExpert training, 64 graphs x 10 agents per GPU, Adam): one
    optimisation step eager and as a HIP-graph replay (tools/train_bench.measure), the CPU oracle's training step
    beside it, and the first-step loss / logits against the oracle on the same inputs."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import train_bench
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from gnn_pathplanning_amd.training import policy_loss
    B, N, K = 64, 10, 3
    with torch.enable_grad():                                # (bench.py's secondary block runs under no_grad)
        return _c4_shard_record(orc, dev, cpu_seconds, train_bench, DecentralPlannerNet, policy_loss, B, N, K)


def _c4_shard_record(orc, dev, cpu_seconds, train_bench, DecentralPlannerNet, policy_loss, B, N, K):
    t_eager, _ = train_bench.measure(dev, B, steps=40, warmup=5, graph=False)
    t_graph, _ = train_bench.measure(dev, B, steps=40, warmup=5, graph=True)

    class Cfg:
        num_agents, nGraphFilterTaps, device = N, K, dev
    torch.manual_seed(1337)
    net = DecentralPlannerNet(Cfg()).to(dev).train()
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    obs = orc.synth_obs(B, N, seed=1337)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, 20, seed=1337)).float()
    tgt = torch.nn.functional.one_hot(torch.randint(0, 5, (B, N), generator=torch.Generator().manual_seed(0)), 5).float()
    net.addGSO(S.to(dev))
    got = net(obs.to(dev))
    loss_g = float(policy_loss(got, tgt.to(dev)).item())
    with torch.no_grad():
        want = orc.policy_forward(sd, S, obs, training=True)
        loss_w = float(orc.policy_loss(want, tgt).item())
    err = max((g.detach().cpu() - w).abs().max().item() for g, w in zip(got, want))
    rec = {'agents': N, 'taps': K, 'batch': B, 'what': 'fwd + loss + bwd + FusedAdam, train-mode BatchNorm; the shard '
           'every one of the 8 GPUs of config 4 runs (the gradient all-reduce of 826 KB is the only collective)',
           'value': B * N / t_graph, 'unit': 'agent-steps/s (training step)', 'ms_per_step': 1e3 * t_graph,
           'how': 'HIP-graph replay of the whole step (GraphedTrainStep), wall clock over 40 steps',
           'eager_value': B * N / t_eager, 'eager_ms_per_step': 1e3 * t_eager,
           'dominant_kernel': 'none: ~50 launches of 4-24 us each (profiles/r06_train_kernel_stats.csv); the step is the '
                              'sum of latency-bound kernels at 640 agent-samples',
           'predicted_8gpu_value_without_allreduce': 8 * B * N / t_graph,
           'parity_train_mode_max_abs_dlogit': err, 'parity_loss_gpu': loss_g, 'parity_loss_oracle': loss_w}
    if cpu_seconds > 0:
        cb = train_bench.cpu_baseline(B, cpu_seconds)
        cb['speedup_gpu_over_cpu'] = rec['value'] / cb['value']
        rec['cpu_baseline'] = cb
    return rec


def compact_summary(d):
    """The per-config figures of the line once more, compactly, as its LAST key (a log that keeps only the tail of
    the line still holds them).  Records are [M agent-steps/s, ms per step, dominant-kernel us, frac, max |dlogit|]."""
    def r(x, n=4):
        return None if x is None else float('%.*g' % (n, x))
    rl, par = d.get('roofline', {}), d.get('parity', {})
    out = {'legend': '[M agent-steps/s, ms/step, dominant kernel us, roofline frac, max |dlogit| vs oracle; other '
                     'configs: + filter-and-head us, its frac, its pipe-busy frac]',
           d['config']['name']: [r(d['value'] / d['n_gpus'] / 1e6), r(d['ms_per_step']), r(rl.get('avg_launch_us')),
                                 r(rl.get('frac')), r(par.get('max_abs_dlogit'), 2)],
           'executed_over_algorithmic_flops': r(rl.get('executed_over_algorithmic_flops')),
           'column_fill': r(rl.get('column_fill')), 'scaling': d.get('scaling'), 'n_gpus': d.get('n_gpus')}
    sec = d.get('secondary') or {}

    def rec(v):
        if 'error' in v:
            return 'error'
        fh = v.get('filter_and_head') or {}
        return [r(v['value'] / 1e6), r(v['ms_per_step']), r(v.get('dominant_kernel_us')), r(v.get('frac')),
                r(v.get('parity_max_abs_dlogit'), 2), r(fh.get('us')), r(fh.get('frac'), 3), r(fh.get('pipe_busy_frac'), 3)]
    for k, v in (sec.get('other_configs') or {}).items():
        out[k] = rec(v)
    for k, v in (sec.get('shards_of_8gpu_configs') or {}).items():
        if k == 'c4_shard' and 'error' not in v:
            out['c4_shard_train'] = {'M_per_s': r(v['value'] / 1e6), 'ms': r(v['ms_per_step']),
                                     'eager_ms': r(v['eager_ms_per_step']),
                                     'cpu_M_per_s': r(v.get('cpu_baseline', {}).get('value', 0) / 1e6) or None,
                                     'dlogit': r(v['parity_train_mode_max_abs_dlogit'], 2)}
        elif k.startswith('c5_shard'):
            out[k] = rec(v)
            g_ = v.get('hip_graph_replay') or {}
            if 'value' in g_:
                out[k + '_graphed_M_per_s'] = r(g_['value'] / 1e6)
            s_ = v.get('forward_logits_eager') or {}
            if 'value' in s_:
                out[k + '_stacked_M_per_s'] = r(s_['value'] / 1e6)
    rv = sec.get('real_valued_observations') or {}
    if 'agent_steps_per_s' in rv:
        out['real_valued_obs_M_per_s'] = r(rv['agent_steps_per_s'] / 1e6)
    rot = sec.get('c2_rotating_batches') or {}
    if 'agent_steps_per_s' in rot:
        out['c2_rotating_M_per_s'] = r(rot['agent_steps_per_s'] / 1e6)
    bb = sec.get('c2_best_batch') or {}
    if 'value' in bb:
        out['c2_best_batch'] = [r(bb['value'] / 1e6), 'B=%d' % bb['batch'], bb['path']]
    dr = sec.get('dispatch_rule') or {}
    if 'chosen' in dr:
        out['dispatch_rule'] = {'chosen': dr['chosen'], 'ms': r(dr['chosen_ms']), 'alt_ms': r(dr['alternative_ms'])}
    fs = sec.get('batch_sweep_filter_only') or []
    if fs:
        best = max(fs, key=lambda x: x['hbm_frac_of_8TBps'])
        out['filter_hbm_frac'] = [r(best['hbm_frac_of_8TBps']), 'B=%d' % best['batch']]
    cb = d.get('cpu_baseline') or {}
    if 'value' in cb:
        out['cpu_M_per_s'] = r(cb['value'] / 1e6)
    return out


DRIVER_LINE_MAX_BYTES = 6144            # a 23 KB line was not recovered by the driver (BENCH_r05 parsed: null)


def _sig(x, n=6):
    """Floats of the printed line to n significant digits (the side file keeps full precision)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    return float('%.*g' % (n, x))


def _pick(d, keys):
    return {k: _sig(d[k]) for k in keys if k in d}


def driver_line(result, details_file=None):
    """The ONE line rank 0 prints: the driver contract's keys + `roofline` + `cpu_baseline` + `parity` + `summary`,
    at most DRIVER_LINE_MAX_BYTES characters.  Everything else of `result` (secondary records, sweeps, shard records,
    prose notes, near-tie list) goes to the side file named by `details_file`, whose path the line carries.  A pure
    function of `result`: tests/test_host_logic.py feeds it canned results."""
    out = _pick(result, ('metric', 'value', 'unit', 'n_gpus', 'ranks_in_group', 'rank_devices', 'dist_backend', 'steps',
                         'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
                         'precision', 'timing', 'rank_devices_distinct', 'device_override'))
    if 'regions_ms' in result:
        out['regions_ms'] = [_sig(x, 5) for x in result['regions_ms']]
    out['config'] = _pick(result['config'], ('workload', 'name', 'agents', 'taps', 'batch_per_gpu', 'global_batch',
                                             'mean_degree', 'parallelism'))
    rl = result.get('roofline')
    if rl is not None:
        o = _pick(rl, ('kernel', 'bound', 'instruction', 'achieved', 'peak', 'unit', 'frac', 'mfma_products_per_fp32_mac',
                       'frac_of_arithmetic_ceiling', 'avg_launch_us', 'flops_per_launch', 'executed_over_algorithmic_flops',
                       'column_fill', 'pipe_busy_frac', 'traffic', 'algorithmic_bytes', 'effective_clock_GHz'))
        o['kernel'] = o.get('kernel', '').split(' (')[0]
        td = rl.get('traffic_detail') or {}
        if 'note' in td:
            o['traffic_note'] = str(td['note'])[:120]
        out['roofline'] = o
    if 'step_breakdown_us' in result:
        out['step_breakdown_us'] = _pick(result['step_breakdown_us'], tuple(result['step_breakdown_us']))
    par = result.get('parity')
    if par is not None:
        out['parity'] = _pick(par, ('max_abs_dlogit', 'tolerance', 'argmax_equal_on_clear_rows', 'near_tie_rows', 'rows',
                                    'range_flag'))
    cb = result.get('cpu_baseline')
    if cb is not None:
        o = _pick(cb, ('value', 'unit', 'cores', 'usable_cores', 'kind', 'ms_per_step', 'speedup_gpu_over_cpu'))
        o['sample'] = str(cb.get('sample', ''))[:200]
        if 'c1_b1' in cb:
            o['c1_b1'] = _pick(cb['c1_b1'], ('cpu_ms_per_step', 'gpu_ms_per_step', 'speedup_gpu_over_cpu', 'cores'))
        if 'one_thread' in cb:
            o['one_thread_value'] = _sig(cb['one_thread'].get('agent_steps_per_s'))
        out['cpu_baseline'] = o
    if 'summary' in result:
        out['summary'] = result['summary']
    out['details_file'] = details_file
    line = json.dumps(out)
    if len(line) > DRIVER_LINE_MAX_BYTES:                 # never print an unparseable line: drop the optional blocks
        for k in ('regions_ms', 'timing', 'step_breakdown_us', 'summary'):
            out.pop(k, None)
            line = json.dumps(out)
            if len(line) <= DRIVER_LINE_MAX_BYTES:
                break
    assert len(line) <= DRIVER_LINE_MAX_BYTES, len(line)
    return line


def write_details(result, path):
    """The full result (what earlier rounds printed as one 20 KB line) as a side file; returns the path written, relative
    to the repo root where possible, or None when no location is writable."""
    cands = [path] if path else []
    cands += [os.path.join(ROOT, 'gpurun_out', 'bench', 'bench_%s_full.json' % result['config']['name']),
              os.path.join(tempfile.gettempdir(), 'gnnpp_bench_%s_full.json' % result['config']['name'])]
    for p in cands:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(p)), exist_ok=True)
            with open(p, 'w') as f:
                json.dump(result, f)
                f.write('\n')
            ap = os.path.abspath(p)
            return os.path.relpath(ap, ROOT) if ap.startswith(ROOT + os.sep) else ap
        except OSError:
            continue
    return None


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): run the N ranks ourselves,
    exactly as the driver's torch.distributed.run line would (one process per GPU, rendezvous on 127.0.0.1), and hand
    back its exit code.  Refuses loudly when the box has fewer GPUs than ranks -- unless GNNPP_BENCH_DEVICE pins all
    ranks to one device (the gloo code-path check on a one-GPU box)."""
    import socket
    if not args.launch_check and 'GNNPP_BENCH_DEVICE' not in os.environ:
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            raise SystemExit('bench.py --gpus %d: this box has %d GPU(s); refusing to print a mislabelled line'
                             % (args.gpus, n_dev))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def launch_check(args):
    """Hidden mode (--launch-check): every rank joins a gloo group and rank 0 prints the group's shape -- what the CPU
    test of the launcher-less `--gpus N` path runs (no GPU, no kernels)."""
    import torch.distributed as dist
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if world > 1:
        dist.init_process_group('gloo')
        t = torch.tensor([rank + 1.0])
        dist.all_reduce(t)
        assert t.item() == world * (world + 1) / 2
    assert world == args.gpus, (world, args.gpus)
    if rank == 0:
        print(json.dumps({'launch_check': True, 'n_gpus': world,
                          'ranks_in_group': dist.get_world_size() if world > 1 else 1}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


class rank_local_block:
    """Everything rank 0 does AFTER the timed regions (roofline probes, secondary records, parity gate, CPU baseline) is
    rank-local: the other ranks are already waiting in the final barrier.  A collective started in here would never be
    matched -- the job would hang until the process-group timeout (r04's C4 shard record did exactly that through
    tools/train_bench.measure -> sharding.aggregate_throughput: found by the 2-rank check of r05).  Inside this context
    every torch.distributed collective raises instead, so such a mistake costs one `error` record, not the run."""
    NAMES = ('all_reduce', 'barrier', 'all_gather', 'all_gather_object', 'all_gather_into_tensor', 'broadcast',
             'broadcast_object_list', 'reduce', 'reduce_scatter', 'reduce_scatter_tensor', 'all_to_all',
             'all_to_all_single', 'gather', 'scatter', 'send', 'recv')

    def __init__(self, active=True):
        self.active, self.saved = active, {}

    def __enter__(self):
        if self.active:
            import torch.distributed as d

            def refuse(name):
                def f(*a, **k):
                    raise RuntimeError('torch.distributed.%s() inside the rank-local block of bench.py: the other '
                                       'ranks are in the final barrier and would never match it' % name)
                return f
            for n in self.NAMES:
                if hasattr(d, n):
                    self.saved[n] = getattr(d, n)
                    setattr(d, n, refuse(n))
        return self

    def __exit__(self, *exc):
        import torch.distributed as d
        for n, f in self.saved.items():
            setattr(d, n, f)
        self.saved = {}
        return False


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def pick_cpu_threads(orc, sd, N, K, budget_s=6.0):
    """Thread count that makes the CPU oracle fastest on a sample (B=128, best of three) of the workload.
    torch's default (= all logical cores) can be catastrophically oversubscribed on the GPU box."""
    obs = orc.synth_obs(128, N, seed=1)
    S = torch.from_numpy(orc.synth_gso_geometric(128, N, 20, seed=1)).float()
    best_t, best = 1, float('inf')
    t_begin = time.perf_counter()
    for t in (1, 2, 4, 8, 16, 32, 64, 128):
        if t > usable_cores() or time.perf_counter() - t_begin > budget_s:
            break
        torch.set_num_threads(t)
        with torch.no_grad():
            orc.policy_forward(sd, S, obs)
            dt = float('inf')
            for _ in range(3):                   # (one sample per thread count picked 8 of 16 cores on a noisy box)
                t0 = time.perf_counter()
                orc.policy_forward(sd, S, obs)
                dt = min(dt, time.perf_counter() - t0)
        if dt < best:
            best_t, best = t, dt
        elif dt > 2.0 * best:
            break
    torch.set_num_threads(best_t)
    return best_t


def time_cpu(orc, sd, S_cpu, obs_cpu, seconds, max_reps=200):
    """Median wall time of oracle.policy_forward (>= 3 warm repetitions, bounded by `seconds`)."""
    with torch.no_grad():
        orc.policy_forward(sd, S_cpu, obs_cpu)
        times = []
        t_start = time.perf_counter()
        while (time.perf_counter() - t_start < seconds or len(times) < 3) and len(times) < max_reps:
            t1 = time.perf_counter()
            orc.policy_forward(sd, S_cpu, obs_cpu)
            times.append(time.perf_counter() - t1)
    times.sort()
    return times[len(times) // 2], len(times), sum(times)


def policy_flops_per_agent(K, mean_deg):
    return 2.0 * (1238112 + 16384 + K * 128 * 128 + (K - 1) * mean_deg * 128 + 640)


def build_case(orc, name, dev, seed, shard=None):
    """shard = (rank, world): strong scaling -- the config's GLOBAL batch (same seed on every rank), of which this rank
    keeps its sharding.shard_batch slice; None: the whole batch (weak scaling: one such batch per GPU)."""
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from gnn_pathplanning_amd.sharding import shard_batch
    N, W, K, B = CONFIGS[name]

    class Cfg:
        num_agents, nGraphFilterTaps, device = N, K, dev
    sd = orc.init_state_dict(K, seed=1337)
    net = DecentralPlannerNet(Cfg()).to(dev).eval()
    net.load_state_dict(sd)
    obs_cpu = orc.synth_obs(B, N, seed=seed)
    S64 = orc.synth_gso_geometric(B, N, W, seed=seed)
    S_cpu = torch.from_numpy(S64).float()
    if shard is not None:
        obs_cpu, S_cpu = (t.contiguous() for t in shard_batch((obs_cpu, S_cpu), *shard))
    return net, sd, Cfg, obs_cpu, S_cpu, float((S_cpu != 0).sum() / max(1, obs_cpu.shape[0] * N))


def pmc_target(args):
    """Hidden mode run UNDER rocprofv3 --pmc by measure_traffic(): a handful of policy steps."""
    from oracle import policy_oracle as orc           # synthetic inputs only
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    net, _, _, obs_cpu, S_cpu, _ = build_case(orc, args.config, dev, 1337)
    obs, S = obs_cpu.to(dev), S_cpu.to(dev)
    with torch.no_grad():
        for _ in range(12):
            net.addGSO(S)
            net(obs)
    torch.cuda.synchronize()


def measure_traffic(config, kernel_substr, timeout_s=150):
    """HBM bytes per launch of the dominant kernel: two rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE do not fit one pass on gfx950) over `bench.py --pmc-target`, mean counter value per
    dispatch of the kernel, combined as the microarch guide prescribes for gfx950:
    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE counts half of wide coalesced reads;
    both counters are in KB).  Returns (bytes | None, detail dict)."""
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None, {'note': 'rocprofv3 not found'}
    vals, detail = {}, {}
    env = dict(os.environ, TMPDIR='/tmp')
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='gnnpp_pmc_', dir='/tmp')
        try:
            cmd = [exe, '--pmc', ctr, '--output-format', 'csv', '-d', d, '-o', 'pmc', '--',
                   sys.executable, os.path.join(ROOT, 'bench.py'), '--pmc-target', '--config', config]
            r = subprocess.run(cmd, cwd='/tmp', env=env, timeout=timeout_s, stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT)
            tot, n, ns = 0.0, 0, 0.0
            for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get('Counter_Name') == ctr and kernel_substr in row.get('Kernel_Name', ''):
                        tot += float(row.get('Counter_Value', 0) or 0)
                        ns += float(row.get('End_Timestamp', 0) or 0) - float(row.get('Start_Timestamp', 0) or 0)
                        n += 1
            if n == 0:
                return None, {'note': '%s pass produced no rows for %s (rc=%d)' % (ctr, kernel_substr, r.returncode)}
            vals[ctr] = tot / n
            detail[ctr + '_KB_per_dispatch'] = tot / n
            detail[ctr + '_dispatches'] = n
        except (subprocess.TimeoutExpired, OSError) as e:
            return None, {'note': '%s pass failed: %s' % (ctr, type(e).__name__)}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    detail['formula'] = '(2*FETCH_SIZE + WRITE_SIZE) * 1024 B (gfx950 correction, MI355X_MICROARCH.md HBM section)'
    return (2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024.0, detail


def measure_clock(fused, args_fused, args_enc, nwg):
    """Engine clock the dominant kernel really runs at: the -DGNNPP_MEASURE build of the SAME kernel stamps
    the 100 MHz wall clock and the shader-cycle counter at its phase boundaries; cycles / wall time between
    kernel start and the end of the encoder, median over the workgroups of the last of three launches.
    (DVFS: a dense-MFMA kernel is power-limited below the 2.4 GHz the peak is quoted at --
    MI355X_MICROARCH.md, "DVFS give-back".)  libgnnpp_measure.so is a profiling build; nothing else of this
    bench touches it."""
    import ctypes
    import numpy as np
    from gnn_pathplanning_amd import _native
    M = _native.measure_lib()
    M.gnnpp_measure_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for _ in range(3):
        rc = M.gnnpp_policy_fwd(*args_fused) if fused else M.gnnpp_encoder_fwd(*args_enc)
        assert rc == 0, rc
        torch.cuda.synchronize()
    buf = np.zeros(1024 * 32, np.uint64)
    assert M.gnnpp_measure_read_stamps(buf.ctypes.data, buf.size) == 0
    rows = buf.reshape(1024, 32)[:min(nwg, 1024)].astype(np.float64)
    wall_ns = (rows[:, 5] - rows[:, 11]) * 10.0           # slot 11 = kernel start, 5 = last convolution done
    cyc = rows[:, 16 + 5] - rows[:, 16 + 11]
    ok = wall_ns > 0
    return float(np.median(cyc[ok] / wall_ns[ok]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', default='c2', choices=sorted(CONFIGS))
    ap.add_argument('--scaling', default='weak', choices=('weak', 'strong'),
                    help='weak (default): every rank runs its own batch of the config\'s size; strong: the '
                         'config\'s batch is the GLOBAL batch, sharded over the ranks with sharding.shard_batch '
                         '(SURVEY.md section 8e: C5 = 128 graphs -> 16 per GPU at N = 8)')
    ap.add_argument('--repeats', type=int, default=7,
                    help='timed regions of exactly --steps steps each; the median one is reported')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip sweeps / rollout / exact-fp32 extras')
    ap.add_argument('--pipeline-streams', type=int, default=3,
                    help='extra measurement: independent batches in flight on this many HIP streams '
                         '(reported under "pipelined", never as "value"); 0 disables')
    ap.add_argument('--cpu-seconds', type=float, default=10.0)
    ap.add_argument('--prewarm-seconds', type=float, default=0.25,
                    help='untimed device pre-warm before the W warmup steps: the same step in a loop '
                         'until the clocks / host caches have settled (the first few hundred steps '
                         'after start-up run ~8 %% slower, profiles/r01_warmup_sweep.jsonl); 0 disables')
    ap.add_argument('--dist-backend', default='nccl',
                    help='nccl (= RCCL, default); gloo only to exercise the multi-rank code path '
                         'on a box with fewer GPUs than ranks (with GNNPP_BENCH_DEVICE=0)')
    ap.add_argument('--pmc', default='auto', choices=('auto', 'off'),
                    help='auto: rank 0 at N=1 measures roofline.traffic with two rocprofv3 --pmc passes')
    ap.add_argument('--pmc-target', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--launch-check', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--details-file', default=None,
                    help='where the full result (secondary records, sweeps, notes) is written; default '
                         'gpurun_out/bench/bench_<config>_full.json.  The printed line stays <= %d bytes and names it'
                         % DRIVER_LINE_MAX_BYTES)
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:   # no launcher around us: run the N ranks ourselves
        sys.exit(self_launch(args, sys.argv[1:]))
    if args.launch_check:
        return launch_check(args)
    assert torch.cuda.is_available(), 'bench.py needs the MI355X'
    if args.pmc_target:
        return pmc_target(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py --gpus %d inside a group of WORLD_SIZE=%d ranks: the line would be mislabelled'
                         % (args.gpus, world))
    dev_index = int(os.environ.get('GNNPP_BENCH_DEVICE', local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    from gnn_pathplanning_amd import _native
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from gnn_pathplanning_amd.sharding import aggregate_throughput, gather_rank_devices
    from oracle import policy_oracle as orc           # checker + cpu_baseline leg + synthetic inputs only
    L = _native.lib()

    N, W, K, B_global = CONFIGS[args.config]
    strong = args.scaling == 'strong'
    seed = 1337 if strong else 1337 + rank                # strong: every rank generates the SAME global batch
    net, sd, Cfg, obs_cpu, S_cpu, mean_deg = build_case(orc, args.config, dev, seed,
                                                        shard=(rank, world) if strong else None)
    B = obs_cpu.shape[0]                                   # graphs THIS rank steps (strong: its shard)
    assert B > 0, 'strong scaling: more ranks than graphs'
    obs, S = obs_cpu.to(dev), S_cpu.to(dev)
    M = B * N

    def step():
        net.addGSO(S)
        return net(obs)

    def timed_regions(fn, steps, repeats, collective=True):
        """`repeats` regions of exactly `steps` calls; per region (wall seconds max over ranks,
        device seconds from HIP events on the launch stream).  collective=False: rank-local (the
        secondary measurements run on rank 0 only)."""
        dist = dist_all if collective else None
        out = []
        for _ in range(repeats):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0.record()
            for _ in range(steps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            if dist is not None:
                _, _, wall = aggregate_throughput(1, wall, device=dev)    # max over ranks
            out.append((wall, e0.elapsed_time(e1) * 1e-3))
        return out

    dist_all = dist
    with torch.no_grad():
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < args.prewarm_seconds:     # set-up, not part of W or K
            for _ in range(50):
                step()
            torch.cuda.synchronize()
        for _ in range(args.warmup):
            out = step()
        regions = timed_regions(step, args.steps, max(1, args.repeats))
    dist_all = None                              # everything below is rank-local
    order = sorted(range(len(regions)), key=lambda i: regions[i][0])
    elapsed, dev_elapsed = regions[order[len(order) // 2]]              # the median region
    # whole job: the graphs ALL ranks stepped / the slowest rank's time (weak: world batches; strong: the one
    # global batch, whose shards differ by at most one graph)
    value = (B_global if strong else world * B) * N * args.steps / elapsed
    value_rank = B * N * args.steps / elapsed             # this rank's own rate (what one GPU does)

    # Secondary: the same K steps with `pipeline_streams` independent rollout batches in flight
    # (batch i on stream i % S).  Sequentially dependent steps of ONE batch cannot overlap, so this
    # is reported separately; it is what a rollout driver holding S episode batches per GPU gets.
    pipelined = None
    if args.pipeline_streams > 1 and not args.no_secondary:
        S_n = args.pipeline_streams
        nets, ins, streams = [net], [(obs, S)], [torch.cuda.Stream() for _ in range(S_n)]
        for i in range(1, S_n):
            ni = DecentralPlannerNet(Cfg()).to(dev).eval()
            ni.load_state_dict(sd)
            nets.append(ni)
            ins.append((orc.synth_obs(B, N, seed=seed + 1000 * i).to(dev),
                        torch.from_numpy(orc.synth_gso_geometric(B, N, W, seed=seed + 1000 * i)).float().to(dev)))

        def run_pipelined(k_steps):
            for k in range(k_steps):
                i = k % S_n
                with torch.cuda.stream(streams[i]):
                    nets[i].addGSO(ins[i][1])
                    nets[i](ins[i][0])
        with torch.no_grad():
            run_pipelined(max(args.warmup, S_n))
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            el_p = float('inf')
            for _ in range(2):                      # best of two: the enqueue side is host-bound-ish
                t0 = time.perf_counter()
                run_pipelined(args.steps)
                torch.cuda.synchronize()
                if dist is not None:
                    dist.barrier()
                el_p = min(el_p, time.perf_counter() - t0)
        v_p, _, el_p = aggregate_throughput(B * N * args.steps, el_p, device=dev)
        pipelined = {'streams': S_n, 'value': v_p, 'ms_per_step': 1e3 * el_p / args.steps,
                     'note': 'K independent batches round-robin on %d HIP streams; not the headline' % S_n}

    rank_devices = gather_rank_devices(dev)                # which physical GPU every rank ran on (tools/check_scale.py)
    rank_ordinals = [dev_index]
    if dist is not None:
        rank_ordinals = [None] * world
        dist.all_gather_object(rank_ordinals, dev_index)
    result = {
        'metric': 'agent-steps/sec (policy fwd)', 'value': value, 'unit': 'agent-steps/s',
        'n_gpus': world, 'ranks_in_group': dist.get_world_size() if dist is not None else 1,
        'rank_devices': rank_devices,
        'dist_backend': dist.get_backend() if dist is not None else None,
        'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': args.scaling,
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'prewarm_s': args.prewarm_seconds,
        'regions_ms': [round(1e3 * r[0], 4) for r in regions],
        'timing': 'median of %d timed regions of exactly %d steps each (barrier + synchronize on both '
                  'sides of every region, max over ranks)' % (len(regions), args.steps),
        'config': {'workload': 'policy forward (addGSO+forward), %d agents, %dx%d map GSO, K=%d, '
                               'batch=%d per GPU, eval mode, fp32 GSO resident in HBM; the SAME resident '
                               'batch every step (inputs are L2 / Infinity-Cache warm by construction)'
                               % (N, W, W, K, B),
                   'name': args.config, 'agents': N, 'taps': K, 'batch_per_gpu': B,
                   'global_batch': B_global if strong else world * B,
                   'mean_degree': round(mean_deg, 3),
                   'parallelism': ('%d-graph batch sharded over %d ranks (sharding.shard_batch), no data-path '
                                   'collective' % (B_global, world)) if strong else 'replicas x%d' % world},
    }
    if pipelined is not None:
        result['pipelined'] = pipelined

    if rank == 0:
      with rank_local_block(active=dist is not None):
          import ctypes
          enc = net.packed_encoder()
          feat = torch.empty(M, 128, device=dev)
          st = _native.stream_ptr(dev)
          vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None       # noqa: E731
          gf, act = net.GFL[0], net.actionsMLP[0]
          aw, ab = act.weight.detach().contiguous(), act.bias.detach().contiguous()
          gbias = gf.bias.detach().reshape(-1).contiguous()
          taps = gf.packed_taps()
          lg = torch.empty(N, B, 5, device=dev)

          def time_kernel(fn, reps=None):
              """Back-to-back launches of ONE C entry point, timed like the main regions (median of 5)."""
              reps = reps or max(50, args.steps)
              for _ in range(10):
                  fn()
              return sorted(r[1] for r in timed_regions(fn, reps, 5))[2] / reps

          # ---- roofline of the dominant kernel, in the arithmetic the HEADLINE runs: the model's default precision
          # 'fp32' = bf16x3 operand split (fp32-equivalent; include/gnnpp.h GNNPP_PREC_FP32)
          prec = net._prec()
          assert prec == _native.PREC_FP32, 'the headline is measured in the default (fp32-equivalent) arithmetic'
          info = PRECISION_INFO[prec]
          tiles = (M + 15) // 16
          fused = fused_rule(L, B, N, K, prec)
          t_enc = time_kernel(lambda: L.gnnpp_encoder_fwd(vp(obs), vp(enc), vp(feat), M, prec, None, st))
          y = torch.empty(M, 128, device=dev)
          t_gf = time_kernel(lambda: L.gnnpp_lsigf_fwd(vp(feat), vp(S), vp(taps), vp(gbias), vp(y), B, N, N,
                                                       128, 128, K, 1, 0, 1, 1, 1, 1, 0, prec, None, st))
          # the filter launch of the policy step itself: features -> logits (filter + ReLU + action head; for teams
          # of 17..100 agents that is policy_filter_kernel, else lsigf_kernel with the fused head)
          lg_fh = torch.empty(N, B, 5, device=dev)
          t_fh = time_kernel(lambda: L.gnnpp_filter_head_fwd(vp(feat), vp(S), vp(taps), vp(gbias), vp(aw), vp(ab),
                                                             vp(lg_fh), B, N, 128, 128, K, 1, 0, prec, None, st))
          enc_flops = 2.0 * ENC_MACS_PER_AGENT * M
          pol_flops = policy_flops_per_agent(K, mean_deg) * M
          if fused:
              kernel = ('gnnpp::encoder_kernel_b3<true, K=%d> (fused policy kernel: encoder + graph filter + action ' % K +
                        'head, one workgroup per graph)')
              kname = 'encoder_kernel_b3<true'                    # (<true, K>: the fused instantiation)
              t_dom = dev_elapsed / args.steps                   # HIP events around the reported region
              how = ('HIP events on the launch stream around the reported timed region / its %d launches '
                     '(the step is this one kernel; includes the inter-launch gap)' % args.steps)
              flops = pol_flops
              alg_bytes = M * 363 * 4.0 + B * N * N * 4.0 + M * 20.0 + (ENC_WEIGHT_FLOATS + K * 128 * 128 + 768) * 4.0
              # bf16 MFMAs per graph tile: encoder 8028 (L1..FC: six plane products where split-f16 issues three = 7428;
              # L0: 600 -- the bench's {0, 1} observations are ONE bf16 plane, so L0 issues three of its six products:
              # PLANE SKIPPING, csrc/encoder_kernel_b3.hip) + filter contraction 192 K.  PMC agrees:
              # SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 / 16384 = 8606 per workgroup at K = 3 (profiles/r03_c2_pmc_summary.txt)
              cp_on = L.gnnpp_get_tuning(13) == 1
              mf, col_fill = fused_mfma_per_graph(N, K, cp_on)
              exe = mf * 16384.0 * B
              if cp_on and N <= CP_MAX_AGENTS:
                  kernel = kernel.replace('<true, K=%d>' % K, '<true, K=%d, CP>' % K)
                  lanes = ('; teams of <= 12 agents: (agent, position) pairs on the MFMA columns of the two 5x5 layers '
                           '(%d instead of %d MFMAs per graph), agents on the columns elsewhere' % (mf, 8028 + 192 * K))
              else:
                  lanes = '; a graph of %d agents occupies a 16-lane tile, so at most %d/16 of the pipe does ' \
                          'algorithmic work' % (N, N)
          else:
              kernel, kname = 'gnnpp::encoder_kernel_b3<false, 3>', 'encoder_kernel_b3<false'
              t_dom = t_enc
              how = ('HIP events around back-to-back launches of the kernel (median of 5 regions); the step is '
                     'this kernel followed by the filter + head kernel')
              flops = enc_flops
              alg_bytes = M * (363 + 128) * 4.0 + ENC_WEIGHT_FLOATS * 4.0
              exe = 8028 * 16384.0 * tiles                        # MFMAs per 16-agent tile ({0, 1} observations) x FLOP each
              lanes = ''
              col_fill = M / (16.0 * tiles)
          traffic, traffic_detail = None, {'note': 'not measured (--pmc off, N > 1, or not rank 0)'}
          if args.pmc == 'auto' and world == 1:
              try:
                  traffic, traffic_detail = measure_traffic(args.config, kname)
              except Exception as e:                              # never let the profiler break the bench line
                  traffic, traffic_detail = None, {'note': 'pmc pass raised %s' % type(e).__name__}
          result['roofline'] = roofline_block(kernel, info, flops, t_dom, exe, lanes, col_fill)
          result['roofline'].update({'traffic': traffic, 'traffic_detail': traffic_detail,
                                     'algorithmic_bytes': alg_bytes, 'avg_launch_how': how})
          clk = None
          if world == 1 and args.pmc == 'auto':
              try:
                  clk = measure_clock(
                      fused,
                      (vp(obs), vp(S), vp(enc), vp(taps), vp(gbias), vp(aw), vp(ab), vp(feat), vp(lg), B, N, K, 1,
                       int(S.dtype is torch.float64), prec, None, st),
                      (vp(obs), vp(enc), vp(feat), M, prec, None, st), B if fused else tiles)
              except Exception as e:                              # a measurement extra: never break the line
                  result['roofline']['effective_clock_note'] = 'not measured: %s' % type(e).__name__
          if clk:
              # the same ratios against the pipe's rate at the clock the kernel was measured to run at (the
              # peak above assumes 2.4 GHz); `frac` stays the contract's figure
              rl = result['roofline']
              rl['effective_clock_GHz'] = clk
              rl['effective_clock_how'] = ('shader cycles / wall time inside the kernel (measure build of the same '
                                           'kernel, median over its workgroups)')
              rl['frac_at_effective_clock'] = rl['frac'] * NOMINAL_CLOCK_GHZ / clk
              rl['pipe_busy_frac_at_effective_clock'] = rl['pipe_busy_frac'] * NOMINAL_CLOCK_GHZ / clk
          result['precision'] = info['name']
          result['dtype'] = 'f32 (bf16x3 exact operand split, f32 accumulate)'
          result['dtype_detail'] = info['dtype']
          gf_bytes = M * (1024 + 4 * N) + 196608.0 * K / 3
          result['step_breakdown_us'] = {
              'whole_step_wall': 1e6 * elapsed / args.steps, 'whole_step_device': 1e6 * dev_elapsed / args.steps,
              'encoder_kernel_alone': t_enc * 1e6, 'filter_kernel_alone': t_gf * 1e6,
              'filter_and_head_alone': t_fh * 1e6, 'one_kernel_step': bool(fused)}
          result['policy_filter'] = {
              'kernel': ('gnnpp::policy_filter_kernel' if 17 <= N <= 100 and L.gnnpp_get_tuning(9) == 1 else
                         'gnnpp::lsigf_kernel') + ' (features -> logits: filter + bias + ReLU + action head, the second '
                        'launch of the two-kernel policy step)',
              'avg_launch_us': t_fh * 1e6, 'agent_steps_per_s': M / t_fh,
              'algorithmic_GBps': (gf_bytes - 512.0 * M + 20.0 * M) / t_fh / 1e9}
          result['filter_kernel'] = {
              'kernel': 'gnnpp::lsigf_kernel (node-major features in, bias + ReLU fused)', 'avg_launch_us': t_gf * 1e6,
              'agent_steps_per_s': M / t_gf, 'algorithmic_GBps': gf_bytes / t_gf / 1e9,
              'hbm_frac_of_8TBps': gf_bytes / t_gf / (HBM_PEAK_TBPS * 1e12),
              'mfma_TFLOPs': 2.0 * (K * 128 * 128 + (K - 1) * mean_deg * 128) * M / t_gf / 1e12,
              'regime': 'one launch over a %.1f MB working set: launch-latency bound, not HBM bound' % (gf_bytes / 1e6)}
          result['policy_TFLOPs'] = policy_flops_per_agent(K, mean_deg) * value_rank / 1e12

          if not args.no_secondary:
              sec = {}
              with torch.no_grad():
                  # (1) what the rollout loop needs on the host: forward + argmax decode + D2H of the ids
                  def step_d2h():
                      net.addGSO(S)
                      return net.decode_actions(net.forward_logits(obs)).cpu()
                  for _ in range(10):
                      step_d2h()
                  r = sorted(x[0] for x in timed_regions(step_d2h, max(20, args.steps // 4), 5))[2]
                  sec['with_argmax_d2h'] = {'agent_steps_per_s': M * max(20, args.steps // 4) / r,
                                            'ms_per_step': 1e3 * r / max(20, args.steps // 4),
                                            'what': 'addGSO + forward + decode_actions kernel + .cpu() of the [B,N] int32 '
                                                    'ids, synchronous every step (multirobotsim_dcenlocal.py:589-599)'}
                  # (1b) the same step on REAL-VALUED observations: the simulator's observations are {0, 1}, one bf16 plane,
                  # and L0 then issues three of its six plane products (plane skipping: bit-identical, exact zeros); any
                  # other observation takes the general L0 (all six products, compiler-scheduled) -- its cost, stated
                  obs_real = (obs * torch.randn(obs.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(7)))

                  def step_real():
                      net.addGSO(S)
                      return net(obs_real)
                  for _ in range(10):
                      out_real = step_real()
                  nst_r = max(100, args.steps // 2)     # (>= 100 steps: 20-step regions read 0.79x where 100-step ones read 0.89x)
                  r = sorted(x[0] for x in timed_regions(step_real, nst_r, 5))[2] / nst_r
                  want_real = orc.policy_forward(sd, S_cpu, obs_real.cpu())
                  sec['real_valued_observations'] = {
                      'agent_steps_per_s': M / r, 'ms_per_step': 1e3 * r, 'vs_binary_observations': (M / r) / value_rank,
                      'parity_max_abs_dlogit': max((g_.cpu() - w_).abs().max().item() for g_, w_ in zip(out_real, want_real)),
                      'what': 'observations x N(0, 1): three bf16 planes per pixel, L0 issues all six plane products '
                              '(b3_l0_generic); everything behind L0 is unchanged'}
                  # (2) the other two arithmetics on the same step, each with its own roofline block: the exact fp32
                  # MFMA, and the opt-in split-f16 mode (NARROWER than fp32: a labelled secondary, never the headline)
                  for pname, key in (('fp32_mfma', 'exact_fp32_mfma_schedule'), ('split_f16', 'split_f16_fast_mode')):
                      pc = _native.precision_code(pname)
                      pinfo = PRECISION_INFO[pc]
                      net.precision = pname
                      old_policy, net.range_policy = net.range_policy, 'flag'     # (the guard is read once, below: the
                      try:                                                       # default 'strict' syncs every forward)
                          for _ in range(10):
                              outp = step()
                          nst = max(100, args.steps // 2)
                          regs = timed_regions(step, nst, 5)
                          r = sorted(x[0] for x in regs)[2]
                          rdev = sorted(x[1] for x in regs)[2]
                          tp = r / nst
                          t_encp = time_kernel(lambda: L.gnnpp_encoder_fwd(vp(obs), vp(enc), vp(feat), M, pc, None, st), 50)
                          flag_p = int(net.range_exceeded()) if pname == 'split_f16' else 0
                      finally:
                          net.precision = 'fp32'
                          net.range_policy = old_policy
                      fused_p = fused_rule(L, B, N, K, pc)
                      if pname == 'split_f16':
                          exe_p = ((4314 + 96 * K) * 16384.0 * B) if fused_p else 4314 * 16384.0 * tiles
                          kern_p = 'gnnpp::encoder_kernel_h2<%s>' % ('true, K=%d' % K if fused_p else 'false, 3')
                      else:
                          exe_p, kern_p = 8876 * 2048.0 * tiles, 'gnnpp::encoder_kernel_f32'
                      rec = {'precision': pname, 'dtype': pinfo['dtype'], 'agent_steps_per_s': M / tp,
                             'ms_per_step': 1e3 * tp, 'encoder_kernel_us': t_encp * 1e6,
                             'max_abs_dlogit_vs_default': max((a - b).abs().max().item() for a, b in zip(out, outp)),
                             'roofline': roofline_block(kern_p, pinfo, pol_flops if fused_p else enc_flops,
                                                        rdev / nst if fused_p else t_encp, exe_p, ''),
                             'range_flag': flag_p}
                      if pname == 'split_f16':
                          rec['what'] = ('precision="split_f16" (opt-in): operands as f16 hi+lo pairs, 22 significand '
                                         'bits, valid for |activation| < 65504 -- narrower than the reference\'s fp32, '
                                         'hence never the headline')
                      else:
                          rec['what'] = 'precision="fp32_mfma": every MFMA is v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain)'
                      sec[key] = rec
                  # (3) batch sweep: many graphs per launch (throughput regime) -- policy step and filter alone
                  sweep, fsweep = [], []
                  for Bs in (B, 4 * B, 16 * B, 64 * B):
                      o = orc.synth_obs(min(Bs, 2048), N, seed=7).to(dev)
                      o = o.repeat((Bs + o.shape[0] - 1) // o.shape[0], 1, 1, 1, 1)[:Bs].contiguous()
                      Ss = S.repeat((Bs + B - 1) // B, 1, 1)[:Bs].contiguous()

                      def sstep():
                          net.addGSO(Ss)
                          return net(o)
                      for _ in range(3):
                          sstep()
                      reps = max(5, min(args.steps, int(2e6 / (Bs * N)) + 1))
                      r = sorted(x[0] for x in timed_regions(sstep, reps, 3))[1]
                      sweep.append({'batch': Bs, 'agent_steps_per_s': Bs * N * reps / r, 'ms_per_step': 1e3 * r / reps})
                      xf = torch.relu(torch.randn(Bs * N, 128, device=dev))
                      yf = torch.empty_like(xf)
                      tf = time_kernel(lambda: L.gnnpp_lsigf_fwd(vp(xf), vp(Ss), vp(taps), vp(gbias), vp(yf), Bs, N, N,
                                                                 128, 128, K, 1, 0, 1, 1, 1, 1, 0, prec, None, st), reps)
                      fb = Bs * N * (1024 + 4 * N) + 196608.0 * K / 3
                      ffl = 2.0 * (K * 128 * 128 + (K - 1) * mean_deg * 128) * Bs * N
                      fsweep.append({'batch': Bs, 'us': tf * 1e6, 'agent_steps_per_s': Bs * N / tf,
                                     'algorithmic_GBps': fb / tf / 1e9, 'hbm_frac_of_8TBps': fb / tf / (HBM_PEAK_TBPS * 1e12),
                                     'algorithmic_TFLOPs': ffl / tf / 1e12})
                      del o, Ss, xf, yf
                  for row in sweep:
                      row['path'] = 'one launch' if fused_rule(L, row['batch'], N, K, prec) else 'encoder + filter launches'
                  sec['batch_sweep_policy'] = sweep
                  best_row = max(sweep, key=lambda x: x['agent_steps_per_s'])
                  sec['c2_best_batch'] = {
                      'value': best_row['agent_steps_per_s'], 'batch': best_row['batch'], 'path': best_row['path'],
                      'vs_headline_batch': best_row['agent_steps_per_s'] / value_rank,
                      'what': 'best row of batch_sweep_policy: the headline batch (%d graphs = two one-graph workgroups '
                              'per CU, ONE round of the chip) is a granularity corner -- larger batches run 16-agent '
                              'tiles with every MFMA column filled' % B}
                  # what gnnpp_policy_fwd's rule chose at the headline shape, and the alternative it rejected, measured
                  # here on the same inputs (GNNPP_TUNE_FUSED_POLICY: 2 = always the one kernel, 0 = never)
                  knob_old = L.gnnpp_get_tuning(6)
                  alt = {}
                  try:
                      for nm_, kv in (('one_launch', 2), ('two_launches', 0)):
                          L.gnnpp_set_tuning(6, kv)
                          for _ in range(5):
                              out_alt = step()
                          nst_a = max(100, args.steps // 2)
                          ra = sorted(x[0] for x in timed_regions(step, nst_a, 5))[2] / nst_a
                          alt[nm_] = {'ms_per_step': 1e3 * ra, 'agent_steps_per_s': M / ra,
                                      'max_abs_dlogit_vs_headline': max((a_ - b_).abs().max().item()
                                                                        for a_, b_ in zip(out, out_alt))}
                  finally:
                      L.gnnpp_set_tuning(6, knob_old)
                  chosen = 'one_launch' if fused else 'two_launches'
                  other = 'two_launches' if fused else 'one_launch'
                  sec['dispatch_rule'] = {
                      'rule': 'csrc/gnnpp_api.hip fused_policy_applies: one launch iff N <= 16, K in 2..4 and '
                              '(B <= 512 or N >= 13)', 'shape': {'batch': B, 'agents': N, 'taps': K},
                      'chosen': chosen, 'chosen_ms': alt[chosen]['ms_per_step'],
                      'alternative': other, 'alternative_ms': alt[other]['ms_per_step'],
                      'chosen_over_alternative': alt[other]['ms_per_step'] / alt[chosen]['ms_per_step'], 'measured': alt}
                  sec['batch_sweep_filter_only'] = fsweep
                  sec['batch_sweep_note'] = ('larger batches amortise launch latency and the per-launch weight stream; '
                                             'the filter-only rows are the HBM-fraction figure of SURVEY.md section 8d '
                                             '(default precision: fp32-equivalent contraction)')
                  # (4) the remaining single-GPU configs of BASELINE.json and a NON-RESIDENT variant of this one:
                  # compact records (value, ms/step, dominant kernel, frac, parity) inside the driver-run line
                  if args.config == 'c2':
                      others = {}
                      for nm, k_over in (('c3', None), ('c5', 2), ('c5', 3), ('c5', 4)):
                          try:
                              others['%s_K%d' % (nm, k_over or CONFIGS[nm][2])] = quick_config(
                                  orc, L, _native, nm, k_over, dev, timed_regions, time_kernel, vp, st)
                          except Exception as e:                  # an extra: never break the line
                              others['%s_K%s' % (nm, k_over)] = {'error': '%s: %s' % (type(e).__name__, e)}
                      sec['other_configs'] = others
                      # (5) the per-GPU SHARDS of the 8-GPU configs, measured on this one GPU: what
                      # each of the 8 ranks of `--scaling strong` executes.  C5: 128 graphs -> 16 per GPU
                      # (sharding.shard_batch, shard 0); C4: 64 graphs per GPU, one optimisation step.
                      # Only in the ONE-GPU line: the records capture HIP graphs (the step, the training step) on rank 0 while
                      # the other ranks sit in a collective -- with RCCL's watchdog thread alive a capture can be invalidated,
                      # and an N-GPU run has no use for a prediction of itself.
                      shards = {}
                      for kk in ((2, 3, 4) if world == 1 else ()):
                          try:
                              rec = quick_config(orc, L, _native, 'c5', kk, dev, timed_regions, time_kernel, vp, st, batch=16)
                              rec['predicted_8gpu_value'] = 8.0 * rec['value']
                              if 'value' in (rec.get('hip_graph_replay') or {}):
                                  rec['predicted_8gpu_value_graph_replay'] = 8.0 * rec['hip_graph_replay']['value']
                              if 'value' in (rec.get('forward_logits_eager') or {}):
                                  rec['predicted_8gpu_value_forward_logits'] = 8.0 * rec['forward_logits_eager']['value']
                              rec['vs_full_batch_rate'] = rec['value'] / others['c5_K%d' % kk]['value']
                              shards['c5_shard_K%d' % kk] = rec
                          except Exception as e:
                              shards['c5_shard_K%d' % kk] = {'error': '%s: %s' % (type(e).__name__, e)}
                      try:
                          if world == 1:
                              shards['c4_shard'] = c4_shard_record(orc, dev, 0.0 if args.no_cpu_baseline
                                                                   else min(4.0, args.cpu_seconds))
                      except Exception as e:
                          shards['c4_shard'] = {'error': '%s: %s' % (type(e).__name__, e)}
                      shards['note'] = ('rollout shards are independent (no data-path collective): the 8-GPU whole-job '
                                        'value of a strong-scaling run is 8 x the shard rate measured here')
                      sec['shards_of_8gpu_configs'] = shards
                      try:
                          sec['c2_rotating_batches'] = rotating_batches(orc, net, dev, N, W, B, timed_regions, value_rank)
                      except Exception as e:
                          sec['c2_rotating_batches'] = {'error': '%s: %s' % (type(e).__name__, e)}
              result['secondary'] = sec

              # one whole rollout step on the device (observation builder + communication GSO + this
              # forward + action decode / collision shielding), B episodes on random maps
              import numpy as np
              from gnn_pathplanning_amd.rollout import BatchedRollout
              rng = np.random.default_rng(1337)
              grids = (rng.random((B, W, W)) < 0.08).astype(np.uint8)
              starts = np.zeros((B, N, 2), np.int64)
              goals = np.zeros((B, N, 2), np.int64)
              for b_i in range(B):
                  free = np.argwhere(grids[b_i] == 0)
                  pick = rng.choice(len(free), size=2 * N, replace=False)
                  starts[b_i], goals[b_i] = free[pick[:N]], free[pick[N:]]
              env = BatchedRollout(grids, starts, goals, 10 ** 6, dev, tie_mode='hashed', seed=1337)
              with torch.no_grad():
                  t_roll1 = time_kernel(lambda: env.step(net), reps=40)
                  t_roll = time_kernel(lambda: env.steps(net, 8), reps=10) / 8
              # the same episodes as two slices on two HIP streams (episodes are independent: one slice's kernel
              # tail and launch gap overlap the other's kernel); reported beside the single-stream figure
              from gnn_pathplanning_amd.rollout import GroupedRollout
              genv = GroupedRollout(grids, starts, goals, 10 ** 6, dev, groups=2, tie_mode='hashed', seed=1337)

              with torch.no_grad():
                  genv.steps(net, 8)
                  # (wall clock between device synchronisations: an event on one stream does not see the other)
                  t_roll2 = sorted(r[0] for r in timed_regions(lambda: genv.steps(net, 8, wait_caller=False), 20, 5,
                                                               collective=False))[2] / 160
              result['rollout_step'] = {'us': t_roll * 1e6, 'agent_steps_per_s': B * N / t_roll,
                                        'us_one_step_per_call': t_roll1 * 1e6,
                                        'us_two_streams': t_roll2 * 1e6,
                                        'agent_steps_per_s_two_streams': B * N / t_roll2,
                                        'how': 'BatchedRollout.steps(model, 8): eight steps enqueued per host call; '
                                               'two_streams: GroupedRollout(groups=2), wall clock over 160 steps',
                                        'what': 'observe + gso + policy forward + move (collision shielding), '
                                                'all on the device, %d episodes' % B}

          # parity gate on the bench batch + CPU baseline (bounded sample of the same workload)
          threads = pick_cpu_threads(orc, sd, N, K)
          with torch.no_grad():
              want = orc.policy_forward(sd, S_cpu, obs_cpu)
          got = [o.cpu() for o in out]
          err = max((g - w).abs().max().item() for g, w in zip(got, want))
          margin = orc.top2_margin(want)
          ids_w = orc.decode_actions(want)
          ids_g = torch.stack([g.argmax(-1) for g in got], 1)
          clear = margin > 1e-5
          near = [{'graph': int(b_), 'agent': int(n_), 'margin': float(margin[b_, n_]),
                   'same_action': bool(ids_g[b_, n_] == ids_w[b_, n_])}
                  for b_, n_ in (~clear).nonzero().tolist()][:16]
          result['parity'] = {'max_abs_dlogit': err, 'tolerance': 1e-4,
                              'argmax_equal_on_clear_rows': bool(torch.equal(ids_g[clear], ids_w[clear])),
                              'near_tie_rows': int((~clear).sum()), 'near_tie_list': near,
                              'rows': int(clear.numel()),
                              # the default arithmetic has no input domain: no guard exists to be read; the flag of
                              # the split-f16 secondary is reported with it
                              'range_flag': int(net.range_exceeded())}
          if not args.no_cpu_baseline:
              med, reps, spent = time_cpu(orc, sd, S_cpu, obs_cpu, args.cpu_seconds)
              cb = {'value': B * N / med, 'unit': 'agent-steps/s', 'cores': threads, 'usable_cores': usable_cores(),
                    'kind': 'port',
                    'sample': '%d repetitions (median) of the same %s batch through oracle/policy_oracle.py '
                              '(torch %s CPU, fp32, eval, no_grad), ~%.0f s of host time'
                              % (reps, args.config, torch.__version__, spent),
                    'ms_per_step': med * 1e3, 'speedup_gpu_over_cpu': value_rank / (B * N / med)}
              # the reference's own rollout step: B = 1 (BASELINE.json configs[0]), same thread count
              o1, S1 = obs_cpu[:1].contiguous(), S_cpu[:1].contiguous()
              med1, reps1, _ = time_cpu(orc, sd, S1, o1, min(3.0, args.cpu_seconds))
              net1 = DecentralPlannerNet(Cfg()).to(dev).eval()
              net1.load_state_dict(sd)
              o1d, S1d = o1.to(dev), S1.to(dev)

              def step1():
                  net1.addGSO(S1d)
                  return net1(o1d)
              with torch.no_grad():
                  for _ in range(20):
                      step1()
                  g1 = sorted(x[0] for x in timed_regions(step1, 100, 5))[2] / 100
              cb['c1_b1'] = {'cpu_agent_steps_per_s': N / med1, 'cpu_ms_per_step': med1 * 1e3, 'cores': threads,
                             'gpu_agent_steps_per_s': N / g1, 'gpu_ms_per_step': g1 * 1e3,
                             'speedup_gpu_over_cpu': med1 / g1, 'repetitions': reps1,
                             'what': 'one 10-agent case per step (the rollout loop of agents/decentralplannerlocal.py:560-599)'}
              torch.set_num_threads(1)
              medt, repst, _ = time_cpu(orc, sd, S_cpu, obs_cpu, min(6.0, args.cpu_seconds), max_reps=20)
              medt1, _, _ = time_cpu(orc, sd, S1, o1, 2.0)
              torch.set_num_threads(threads)
              cb['one_thread'] = {'agent_steps_per_s': B * N / medt, 'ms_per_step': medt * 1e3, 'repetitions': repst,
                                  'c1_b1_agent_steps_per_s': N / medt1}
              result['cpu_baseline'] = cb
          result['summary'] = compact_summary(result)
          # one process per GPU, really: as many ranks in the group as the line claims, each on its own device (unless
          # GNNPP_BENCH_DEVICE pinned them to one on purpose: the gloo code-path check)
          assert result['ranks_in_group'] == result['n_gpus'] == args.gpus, (result['ranks_in_group'], args.gpus)
          if 'GNNPP_BENCH_DEVICE' not in os.environ:
              # every rank on its OWN device ordinal of this node (what the launch controls); the physical identities
              # (uuid / PCI bus id, as torch reports them) are stated beside it, not asserted: a runtime that reports
              # one uuid for every GPU must not cost the line
              assert len(set(rank_ordinals)) == args.gpus, rank_ordinals
              result['rank_devices_distinct'] = len(set(map(str, rank_devices))) == args.gpus
          else:
              result['device_override'] = os.environ['GNNPP_BENCH_DEVICE']
          # the line the driver parses: <= 6 KB (contract keys, roofline, cpu_baseline, parity, summary); the rest
          # (secondary records, sweeps, notes) in the side file it names
          print(driver_line(result, write_details(result, args.details_file)), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
