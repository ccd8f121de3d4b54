"""CPU: host-side logic of the package -- module tree / state_dict contract, init rule, shape
asserts, loud failure without a GPU, and the C-ABI surface of the built library."""
import ctypes
import math
import os
import re

import pytest
import torch

from conftest import ROOT, golden_state_dict
from oracle import policy_oracle as orc


class Cfg:
    def __init__(self, n=10, k=3):
        self.num_agents, self.nGraphFilterTaps, self.device = n, k, torch.device('cpu')


@pytest.fixture(scope='module')
def built_lib():
    from gnn_pathplanning_amd import _native
    if not os.path.exists('/opt/rocm/bin/hipcc') and not os.path.exists(_native.LIB_PATH):
        pytest.skip('hipcc not available and libgnnpp.so not prebuilt')
    _native.build()
    return _native


def test_cabi_exports_every_declared_symbol(built_lib):
    header = open(os.path.join(ROOT, 'include', 'gnnpp.h')).read()
    declared = set(re.findall(r'\b(gnnpp_[a-z_]+)\s*\(', header))
    assert declared == set(built_lib.EXPORTS), declared ^ set(built_lib.EXPORTS)
    L = ctypes.CDLL(built_lib.LIB_PATH)
    for sym in declared:
        assert hasattr(L, sym), sym
    lib = built_lib.lib()
    assert lib.gnnpp_version() >= 100
    assert b'not supported' in lib.gnnpp_error_string(-2)
    # fp32 fragments + split-f16 fragments + scale + bf16x3 fragments (three 16-byte planes per block of 32 channels)
    assert lib.gnnpp_filter_packed_floats(128, 128, 3, 1) == 2 * (3 * 8 * 8 * 256) + 4 + 3 * 8 * 4 * 768
    assert lib.gnnpp_filter_packed_floats(5, 3, 2, 1) == 2 * 256 + 2 * 512 + 4 + 2 * 768
    assert lib.gnnpp_encoder_packed_floats() > 555000 // 4
    # argument validation happens before any HIP call, so it is checkable without a GPU
    assert lib.gnnpp_encoder_fwd(None, None, None, 16, 0, None, None) == -1
    assert lib.gnnpp_version() == 330                        # ABI 300: per-call `precision`; 320: n-way graph split; 330: one-launch training packs, folded masks
    assert lib.gnnpp_set_tuning(0, 7) == -1 and lib.gnnpp_set_tuning(5, 0) == -1
    assert lib.gnnpp_decode_actions(None, None, 1, 1, None) == -1
    # gnnpp_lsigf_fits: 1 = the LDS-resident kernels take the graph, 0 = take the dense form (F > 8192 too)
    assert lib.gnnpp_lsigf_fits(10, 128, 128, 3, 1) == 1 and lib.gnnpp_lsigf_fits(100, 128, 8192, 3, 1) == 1
    assert lib.gnnpp_lsigf_fits(10, 128, 8193, 3, 1) == 0 and lib.gnnpp_lsigf_fits(113, 128, 128, 3, 1) == 0
    assert lib.gnnpp_lsigf_fits(0, 128, 128, 3, 1) == -1
    # tuning keys: n-way graph split (v320), the removed MODE 3 value of GNNPP_TUNE_POLICY_FILTER
    assert lib.gnnpp_set_tuning(7, 7) == 0 and lib.gnnpp_set_tuning(7, 8) == -1 and lib.gnnpp_set_tuning(7, 0) == 0
    assert lib.gnnpp_set_tuning(9, 2) == -1 and lib.gnnpp_get_tuning(9) == 1


def test_unvalidated_toolchain_switches_the_asm_scheduled_layers_off(built_lib, tmp_path, monkeypatch):
    """the column-packed layers rest on hand-placed MFMA wait states; a library built by another hipcc than
    the validated one loads with the compiler-scheduled forms (same results) and a warning, until re-validated."""
    import warnings
    L = built_lib.lib()
    assert built_lib.built_with() in (None, built_lib.VALIDATED_HIPCC)           # this container IS the validated toolchain
    assert L.gnnpp_get_tuning(13) == 1 and L.gnnpp_get_tuning(14) == 0
    stamp = tmp_path / 'toolchain.txt'
    stamp.write_text('HIP version: 9.9.99999-deadbeef\n')
    monkeypatch.setattr(built_lib, 'TOOLCHAIN_STAMP', str(stamp))
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            built_lib._apply_toolchain_policy(L)
        assert sum('column-packed' in str(x.message) for x in w) == 1
        assert L.gnnpp_get_tuning(13) == 0 and L.gnnpp_get_tuning(14) == 16
        L.gnnpp_set_tuning(13, 1); L.gnnpp_set_tuning(14, 0)
        monkeypatch.setenv('GNNPP_TRUST_TOOLCHAIN', '1')
        built_lib._apply_toolchain_policy(L)
        assert L.gnnpp_get_tuning(13) == 1 and L.gnnpp_get_tuning(14) == 0
    finally:
        L.gnnpp_set_tuning(13, 1); L.gnnpp_set_tuning(14, 0)


def test_split_f16_is_refused_when_its_isa_check_failed(built_lib, tmp_path, monkeypatch):
    """the opt-in split-f16 encoder is outside the build's must-pass ISA list; a violation there
    leaves a marker and the PRECISION is refused with a message, the default path keeps working."""
    mark = tmp_path / 'split_f16_disabled.txt'
    monkeypatch.setattr(built_lib, 'H2_UNSAFE_MARK', str(mark))
    assert built_lib.precision_code('split_f16') == 2
    mark.write_text('encoder_kernel_h2ILb0ELi3E: ring slot read before its load was waited for')
    with pytest.raises(built_lib.GnnppError, match='split_f16'):
        built_lib.precision_code('split_f16')
    with pytest.raises(built_lib.GnnppError):
        built_lib.precision_code(2)
    assert built_lib.precision_code('fp32') == 0 and built_lib.precision_code('fp32_mfma') == 1
    assert all(k[0].startswith('encoder_kernel_b3') for k in built_lib.RING_KERNELS)
    assert all(k[0].startswith('encoder_kernel_h2') for k in built_lib.RING_KERNELS_OPTIONAL)


def test_module_tree_matches_reference_state_dict(policy_golden):
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    z, _ = policy_golden
    sd = golden_state_dict(z, 3)
    net = DecentralPlannerNet(Cfg())
    mine = net.state_dict()
    assert list(mine.keys()) == list(sd.keys())
    for k in sd:
        assert tuple(mine[k].shape) == tuple(sd[k].shape) and mine[k].dtype == sd[k].dtype, k
    net.load_state_dict(sd)                                  # strict load of a reference checkpoint
    assert sum(p.numel() for p in net.parameters()) == 206501
    assert net.numAgents == 10 and net.L == 1 and net.K == [3] and net.E == 1 and net.F == [128, 128]
    assert net.numFeatures2Share == 128 and net.bias is True
    # transfer-learning freeze pattern of agents/decentralplannerlocal.py:172-179 still matches
    import fnmatch
    names = [n for n, _ in net.named_parameters()]
    assert any(fnmatch.fnmatch(n, '*GFL*') for n in names)
    assert any(fnmatch.fnmatch(n, '*actions*') for n in names)
    for K in (1, 2, 4):
        assert DecentralPlannerNet(Cfg(7, K)).GFL[0].weight.shape == (128, 1, K, 128)


def test_init_rule_statistics():
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    torch.manual_seed(0)
    net = DecentralPlannerNet(Cfg())
    w = net.ConvLayers[14].weight                           # xavier normal: std = sqrt(2/(fan_in+fan_out))
    assert abs(w.std().item() - math.sqrt(2.0 / (64 * 9 + 128 * 9))) < 2e-3
    assert net.compressMLP[0].bias.abs().max().item() == 0.0
    assert abs(net.ConvLayers[15].weight.mean().item() - 1.0) < 0.01
    gw = net.GFL[0].weight
    bound = 1.0 / math.sqrt(128 * 3)
    assert gw.abs().max().item() <= bound and gw.abs().max().item() > 0.9 * bound
    assert net.GFL[0].bias.shape == (128, 1)


def test_addgso_and_graphfilter_contracts():
    import gnn_pathplanning_amd.graphML as gml
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    net = DecentralPlannerNet(Cfg())
    with pytest.raises(AssertionError):
        net.addGSO(torch.zeros(10, 10))
    net.addGSO(torch.zeros(2, 10, 10))
    assert net.S.shape == (2, 1, 10, 10)
    gf = gml.GraphFilter(4, 6, 3, 2, bias=False)
    assert gf.bias is None and gf.weight.shape == (6, 2, 3, 4)
    assert 'no GSO stored' in gf.extra_repr() and 'filter_taps=3' in gf.extra_repr()
    with pytest.raises(AssertionError):
        gf.addGSO(torch.zeros(1, 5, 5))                      # E mismatch
    gf.addGSO(torch.zeros(2, 5, 5))
    assert gf.N == 5 and 'GSO stored' in gf.extra_repr()
    gfb = gml.GraphFilterBatch(4, 6, 3)
    with pytest.raises(AssertionError):
        gfb.addGSO(torch.zeros(2, 5, 5))
    with pytest.raises(TypeError):
        gml.GraphFilterBatch(4, 6, 3)(torch.zeros(1, 4, 5))


def test_no_cpu_fallback(built_lib):
    """CPU tensors must raise -- the product has exactly one compute path."""
    import gnn_pathplanning_amd.graphML as gml
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    net = DecentralPlannerNet(Cfg()).eval()
    net.addGSO(torch.zeros(1, 10, 10))
    with pytest.raises(built_lib.GnnppError):
        net(torch.zeros(1, 10, 3, 11, 11))
    with pytest.raises(built_lib.GnnppError):
        gml.BatchLSIGF(torch.zeros(4, 1, 2, 4), torch.zeros(1, 1, 5, 5), torch.zeros(1, 4, 5))
    gf = gml.GraphFilter(4, 4, 2)
    gf.addGSO(torch.zeros(1, 5, 5))
    with pytest.raises(built_lib.GnnppError):
        gf(torch.zeros(1, 4, 5))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'gnn_pathplanning_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, os.path.join(dirpath, f)
                assert 'emu' not in src.lower().replace('enumerate', ''), os.path.join(dirpath, f)


def test_pack_cache_keying():
    from gnn_pathplanning_amd._native import PackCache
    calls = []
    t = torch.zeros(4)
    c = PackCache()
    c.get((t,), lambda: calls.append(1) or 'a')
    c.get((t,), lambda: calls.append(1) or 'b')
    assert len(calls) == 1
    t.add_(1)                                                # in-place update bumps _version
    assert c.get((t,), lambda: calls.append(1) or 'c') == 'c' and len(calls) == 2


def test_prepowered_gso_api_surface():
    import gnn_pathplanning_amd.graphML as gml
    m = gml.GraphFilterBatchGSO(4, 6, 3, 2, bias=False)
    assert isinstance(m, gml.GraphFilter) and m.weight.shape == (6, 2, 3, 4)
    SK = gml.matrixPowersBatch(torch.rand(2, 5, 5), 3)
    assert SK.shape == (2, 3, 5, 5) and torch.equal(SK[:, 0], torch.eye(5).repeat(2, 1, 1))
    S = torch.rand(2, 2, 5, 5)
    SK = gml.matrixPowersBatch(S, 3)
    assert SK.shape == (2, 2, 3, 5, 5) and torch.allclose(SK[:, :, 2], S @ S, atol=1e-6)
    m.addGSO(S)
    assert m.N == 5 and m.B == 2 and 'number_nodes=5' in m.extra_repr()


def _check_r04_extras(f, d, summary_last=True):
    """r04: the line ends with a compact `summary` (so that a 2 000-character tail keeps the per-config figures), the
    roofline says how many MFMA FLOPs were issued per algorithmic FLOP and how full the tiles' columns were, and the
    C2 line holds the per-GPU SHARDS of the 8-GPU configs (C5: 16 graphs x 100 agents, K = 2, 3, 4; C4: a 64 x 10
    optimisation step with the CPU oracle's training step beside it)."""
    import json
    assert list(d)[-1] == 'summary' or not summary_last, f     # (r04 / r05: a 2 000-character tail had to keep it)
    sm, rl = d['summary'], d['roofline']
    assert len(json.dumps(sm)) <= 1900, (f, len(json.dumps(sm)))
    name = d['config']['name']
    assert abs(sm[name][0] * 1e6 - d['value'] / d['n_gpus']) <= 2e-3 * d['value']
    assert abs(sm[name][1] - d['ms_per_step']) <= 2e-3 * d['ms_per_step']
    assert abs(rl['executed_over_algorithmic_flops'] - rl['executed_mfma_flops_per_launch'] / rl['flops_per_launch']) <= 1e-9
    assert rl['executed_over_algorithmic_flops'] > 1 and 0 < rl['column_fill'] <= 1
    assert abs(sm['executed_over_algorithmic_flops'] - rl['executed_over_algorithmic_flops']) <= 1e-3 * rl['executed_over_algorithmic_flops']
    if name != 'c2':
        return
    sh = d['secondary']['shards_of_8gpu_configs']
    for k in ('c5_shard_K2', 'c5_shard_K3', 'c5_shard_K4'):
        v = sh[k]
        assert 'error' not in v, (k, v)
        assert v['batch'] == 16 and v['agents'] == 100
        assert abs(v['value'] - v['batch'] * v['agents'] / (v['ms_per_step'] * 1e-3)) <= 1e-3 * v['value']
        assert abs(v['predicted_8gpu_value'] - 8 * v['value']) <= 1e-6 * v['value']
        assert v['parity_max_abs_dlogit'] <= 1e-4 and v['argmax_equal_on_clear_rows'] is True
        assert abs(sm[k][0] * 1e6 - v['value']) <= 2e-3 * v['value']
    c4 = sh['c4_shard']
    assert 'error' not in c4, c4
    assert c4['batch'] == 64 and c4['agents'] == 10
    assert abs(c4['value'] - 640 / (c4['ms_per_step'] * 1e-3)) <= 1e-3 * c4['value']
    assert c4['parity_train_mode_max_abs_dlogit'] <= 1e-4
    assert abs(c4['parity_loss_gpu'] - c4['parity_loss_oracle']) <= 1e-5
    assert c4['cpu_baseline']['value'] > 0 and c4['cpu_baseline']['kind'] == 'port'
    for k in ('c3_K3', 'c5_K2', 'c5_K3', 'c5_K4', 'c2_rotating_M_per_s', 'filter_hbm_frac', 'c4_shard_train'):
        assert k in sm, k


def _check_r05_extras(f, d):
    """r05: the C2 line states the granularity corner -- `c2_best_batch` (best row of the
    batch sweep, with its batch and the path gnnpp_policy_fwd took) and `dispatch_rule` (what fused_policy_applies
    chose at the headline shape and the alternative it rejected, both measured) -- inside the 2 000-character tail,
    and the records of the other configs / shards carry the filter-and-head launch's roofline figures."""
    import json
    _check_r04_extras(f, d)
    if d['config']['name'] != 'c2':
        return
    sm, sec = d['summary'], d['secondary']
    tail = json.dumps(d)[-2000:]
    assert '"c2_best_batch"' in tail and '"dispatch_rule"' in tail, f
    bb, dr = sec['c2_best_batch'], sec['dispatch_rule']
    rows = sec['batch_sweep_policy']
    assert bb['value'] == max(x['agent_steps_per_s'] for x in rows) and bb['batch'] in [x['batch'] for x in rows]
    assert bb['path'] in ('one launch', 'encoder + filter launches')
    assert abs(sm['c2_best_batch'][0] * 1e6 - bb['value']) <= 2e-3 * bb['value']
    assert dr['chosen'] in ('one_launch', 'two_launches') and dr['alternative'] != dr['chosen']
    assert dr['shape'] == {'batch': d['config']['batch_per_gpu'], 'agents': d['config']['agents'], 'taps': d['config']['taps']}
    assert abs(dr['chosen_over_alternative'] - dr['alternative_ms'] / dr['chosen_ms']) <= 1e-9
    assert abs(dr['chosen_ms'] - d['ms_per_step']) <= 0.15 * d['ms_per_step']      # (the same step, re-timed)
    assert all(m['max_abs_dlogit_vs_headline'] <= 1e-4 for m in dr['measured'].values())
    assert sm['dispatch_rule']['chosen'] == dr['chosen']
    recs = dict(sec['other_configs'])
    recs.update({k: v for k, v in sec['shards_of_8gpu_configs'].items() if k.startswith('c5_shard')})
    for k, v in recs.items():
        fh = v['filter_and_head']
        assert fh['instruction'] in ('v_mfma_f32_16x16x32_bf16', 'v_mfma_f32_16x16x4_f32'), k
        assert abs(fh['frac'] - fh['algorithmic_TFLOPs'] / fh['peak_TFLOPs']) <= 1e-9 and 0 < fh['frac'] < 1
        assert fh['frac'] <= fh['pipe_busy_frac'] < 1 and abs(fh['us'] - v['filter_and_head_us']) <= 1e-6
        assert len(sm[k]) == 8 and abs(sm[k][5] - fh['us']) <= 2e-3 * fh['us']
        fl = v.get('forward_logits_eager')                      # (shard records of lines written after the gap probe)
        if fl is not None:
            assert fl['bit_identical_to_forward'] is True and fl['value'] >= 0.95 * v['value']
            assert abs(sm[k + '_stacked_M_per_s'] * 1e6 - fl['value']) <= 2e-3 * fl['value']


def test_committed_bench_lines_follow_the_contract():
    """The bench lines committed under profiles/ (copied from GPU sessions) carry every field of the driver's
    contract, with consistent arithmetic: value = batch x agents / ms_per_step, roofline.frac = achieved / peak with
    the guide's dense peak of the MFMA instruction actually issued (the product count is stated, not folded into the
    peak), the headline in fp32-equivalent arithmetic, and -- in the C2 line -- the compact records of the other
    single-GPU configs and of the non-resident (rotating-batch) workload."""
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, 'profiles', 'r03_bench_c*.json')) +
                   glob.glob(os.path.join(root, 'profiles', 'r04_bench_c*.json')) +
                   glob.glob(os.path.join(root, 'profiles', 'r05_bench_c*.json')) +
                   [x for x in glob.glob(os.path.join(root, 'profiles', 'r06_bench_c*.json')) if not x.endswith('_full.json')])
    assert len(files) >= 3
    assert any(os.path.basename(x).startswith('r06_bench_c2') for x in files), 'the r06 bench lines are missing'
    for f in files:
        raw = open(f).read().strip().splitlines()[-1]
        d = json.loads(raw)
        if os.path.basename(f).startswith('r04'):
            _check_r04_extras(f, d)
        if os.path.basename(f).startswith('r05'):
            _check_r05_extras(f, d)
        if os.path.basename(f).startswith('r06'):
            # r06: the file holds the line AS PRINTED -- at most 6 KB, every contract block in it --
            # and names the side file with everything else; the printed line is bench.driver_line() of that side file
            import sys
            sys.path.insert(0, root)
            import bench
            assert len(raw) <= bench.DRIVER_LINE_MAX_BYTES == 6144, (f, len(raw))
            for key in ('roofline', 'cpu_baseline', 'parity', 'summary', 'details_file'):
                assert key in d, (f, key)
            assert 'secondary' not in d and d['details_file'].endswith('_full.json')
            full = json.loads(open(f[:-len('.json')] + '_full.json').read())
            assert bench.driver_line(full, d['details_file']) == raw, f
            if full['config']['name'] == 'c2':
                for key in ('c3_K3', 'c5_K2', 'c5_K3', 'c5_K4', 'c5_shard_K3', 'c4_shard_train', 'c2_best_batch',
                            'dispatch_rule', 'filter_hbm_frac', 'cpu_M_per_s'):
                    assert key in d['summary'], (f, key)
                # the filter-and-head records state the schedule the LIBRARY reported (gnnpp_filter_head_mode), and the
                # split 100-agent teams run bf16x3 planes aliased onto the dead z buffer (mode 3), not the fp32 MFMA
                oc = full['secondary']['other_configs']
                assert oc['c3_K3']['filter_and_head']['mode'] == 2 and oc['c5_K3']['filter_and_head']['mode'] == 3
                assert oc['c5_K3']['filter_and_head']['instruction'] == 'v_mfma_f32_16x16x32_bf16'
            d = full                                          # (the checks below: on the full record)
            _check_r04_extras(f, d, summary_last=False)
        for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                    'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
            assert key in d, (f, key)
        assert d['unit'] == 'agent-steps/s' and d['higher_is_better'] is True and d['scaling'] in ('weak', 'strong')
        assert d['vs_baseline'] is None and d['data'] == 'synthetic' and 'workload' in d['config']
        assert d['dtype'].startswith('f32') and d['precision'] == 'fp32'      # the headline: fp32-equivalent arithmetic
        cfg = d['config']
        per_step = cfg['batch_per_gpu'] * cfg['agents'] * d['n_gpus']
        assert abs(d['value'] - per_step / (d['ms_per_step'] * 1e-3)) <= 1e-3 * d['value']
        rl = d['roofline']
        for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'instruction', 'mfma_products_per_fp32_mac'):
            assert key in rl, (f, key)
        assert rl['bound'] in ('hbm', 'mfma') and abs(rl['frac'] - rl['achieved'] / rl['peak']) <= 1e-6
        assert rl['instruction'] == 'v_mfma_f32_16x16x32_bf16' and rl['peak'] == 2500.0
        assert rl['mfma_products_per_fp32_mac'] == 6
        assert abs(rl['achieved'] - rl['flops_per_launch'] / (rl['avg_launch_us'] * 1e-6) / 1e12) <= 1e-6 * rl['achieved']
        assert rl['avg_launch_us'] <= d['ms_per_step'] * 1e3 * 1.001 or not d['step_breakdown_us']['one_kernel_step']
        assert rl['traffic'] is None or rl['traffic'] > 0
        cb = d['cpu_baseline']
        for key in ('value', 'unit', 'cores', 'kind', 'sample'):
            assert key in cb, (f, key)
        assert cb['kind'] in ('port', 'reference') and cb['value'] > 0
        assert d['parity']['max_abs_dlogit'] <= d['parity']['tolerance'] and d['parity']['range_flag'] == 0
        sec = d['secondary']
        # the narrower split-f16 mode and the exact fp32 MFMA are labelled secondaries with their own roofline blocks
        for key, instr in (('split_f16_fast_mode', 'v_mfma_f32_16x16x32_f16'),
                           ('exact_fp32_mfma_schedule', 'v_mfma_f32_16x16x4_f32')):
            r2 = sec[key]['roofline']
            assert r2['instruction'] == instr and abs(r2['frac'] - r2['achieved'] / r2['peak']) <= 1e-6
            assert sec[key]['max_abs_dlogit_vs_default'] <= 1e-4
        if cfg['name'] == 'c2':
            oc = sec['other_configs']
            assert set(oc) == {'c3_K3', 'c5_K2', 'c5_K3', 'c5_K4'}
            for k, v in oc.items():
                assert 'error' not in v, (k, v)
                assert abs(v['value'] - v['batch'] * v['agents'] / (v['ms_per_step'] * 1e-3)) <= 1e-3 * v['value']
                assert v['parity_max_abs_dlogit'] <= 1e-4 and v['argmax_equal_on_clear_rows'] is True
                assert 0 < v['frac'] < 1 / 6 and v['dominant_kernel_us'] < v['ms_per_step'] * 1e3
            rot = sec['c2_rotating_batches']
            assert rot['batches'] >= 64 and rot['resident_MB'] > 256 and 0.5 < rot['vs_single_resident_batch'] < 1.1
            assert any(x['batch'] >= 8192 for x in sec['batch_sweep_filter_only'])


def test_output_slot_recycles_only_what_nobody_can_observe():
    """eval-mode forward() hands the previous step's [N,B,5] buffer and its N views out again
    (decentralplanner._OutputSlot) -- only when the caller has dropped the previous list and everything derived from
    it: a kept list, one kept view, a slice of a view, a tensor autograd saved, a different stream or shape all force
    fresh memory."""
    import torch
    from gnn_pathplanning_amd.decentralplanner import _OutputSlot
    s, dev = _OutputSlot(), torch.device('cpu')

    def step(stream=0, N=12, B=4):
        t = s.acquire(N, B, dev, stream)
        assert s.views is not None and len(s.views) == N and s.views[0]._base is t
        return t, list(s.views)

    t1, l1 = step()
    p1 = t1.data_ptr()
    assert isinstance(l1, list) and l1[3].shape == (4, 5) and l1[3].data_ptr() == p1 + 3 * 4 * 5 * 4
    t2, l2 = step()                                       # l1 is alive: its memory must not be written again
    assert t2.data_ptr() != p1 and (s.fresh, s.recycled) == (2, 0)
    p2 = t2.data_ptr()
    del t1, l1, t2, l2
    t3, l3 = step()                                       # everything dropped: same buffer, same view objects
    assert t3.data_ptr() == p2 and (s.fresh, s.recycled) == (2, 1)
    ids = [id(v) for v in l3]
    l3.append(None)                                       # the caller's list is its own: mutating it changes nothing
    del t3, l3
    t4, l4 = step()
    assert [id(v) for v in l4] == ids and len(l4) == 12 and s.recycled == 2
    keep = l4[5]                                          # ONE view kept
    del t4, l4
    t5, l5 = step()
    assert t5.data_ptr() != keep._base.data_ptr() and s.fresh == 3
    sub = l5[2][:, :2]                                    # a slice of a view kept (storage use count)
    del t5, l5
    t6, l6 = step()
    assert t6.data_ptr() != sub._base.data_ptr() if sub._base is not None else True
    assert s.fresh == 4
    w = torch.ones(4, 5, requires_grad=True)
    z = (w * l6[0]).sum()                                 # autograd saved l6[0] (TensorImpl use count)
    del t6, l6
    t7, l7 = step()
    assert s.fresh == 5
    z.backward()
    del t7, l7, z
    n_fresh = s.fresh
    step(stream=7)                                        # another stream / another shape: their own entries
    step(N=3)
    assert s.fresh == n_fresh + 2
    for _ in range(5):
        step()
    assert s.fresh == n_fresh + 2 and s.recycled >= 7
    s.clear()
    step()
    assert s.fresh == n_fresh + 3


def test_deferred_gemm_queue_writes_to_whatever_is_the_gradient(monkeypatch):
    """r06b: products that only yield parameter gradients wait in _native's queue until a later launch of the same
    backward pass (_native.defer_gemms / flush_deferred_gemms).  The queue holds a product's OUTPUT weakly -- a strong
    reference would make autograd's AccumulateGrad clone the still-uncomputed buffer -- and at flush time writes to the
    tensor that then IS the gradient: the adopted `.grad` (same storage, new tensor object), the tensor captured by
    torch.autograd.grad, or nothing when nobody holds the result.  The autograd engine runs the flush as a final callback
    of the pass.  (GPU twin with the real GEMM: tests/test_gpu_training.py::test_deferred_parameter_gradient_products.)"""
    import torch
    from gnn_pathplanning_amd import _native
    launches = []

    def fake_multi(specs):                                   # C = A^T-ish products of this test: out <- A @ B
        launches.append(len(specs))
        for sp in specs:
            sp[4].copy_(sp[0] @ sp[2])
    monkeypatch.setattr(_native, 'gemm_kmajor_multi', fake_multi)

    class Lin(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, W, flush):
            ctx.save_for_backward(x, W)
            ctx.flush = flush
            return x @ W.t()

        @staticmethod
        def backward(ctx, g):
            x, W = ctx.saved_tensors
            dW = torch.full_like(W, float('nan'))            # "uncomputed"
            spec = (g.t().contiguous(), None, x, None, dW, None, 1, 0, 0, 0)
            if ctx.flush:
                _native.flush_deferred_gemms([spec])
            elif W.grad is None:
                _native.defer_gemms([spec], [W])
            else:
                _native.gemm_kmajor_multi([spec])
            return g @ W, dW, None
    x = torch.arange(12.0).reshape(4, 3)
    W1, W2 = torch.nn.Parameter(torch.ones(2, 3)), torch.nn.Parameter(torch.ones(3, 3))
    want1 = (torch.ones(4, 2).t() @ (x @ W2.detach().t()))
    # (1) the queued product of the LAST layer rides with the flushing (first) layer's launch: one launch, two products
    Lin.apply(Lin.apply(x, W2, True), W1, False).sum().backward()
    assert launches == [2] and not _native._deferred_gemms
    assert torch.equal(W1.grad, want1) and torch.isfinite(W2.grad).all()
    # (2) accumulation: `.grad` exists -> the caller computes at once (nothing queued), autograd adds
    launches.clear()
    Lin.apply(Lin.apply(x, W2, True), W1, False).sum().backward()
    assert launches == [1, 1] and torch.equal(W1.grad, 2 * want1)
    # (3) a pass that never reaches a flushing node: the engine's final callback launches the queue
    W1.grad = None
    launches.clear()
    g1, = torch.autograd.grad(Lin.apply(x @ W2.detach().t(), W1, False).sum(), [W1])
    assert launches == [1] and not _native._deferred_gemms and torch.equal(g1, want1)
    W1.grad = None
    launches.clear()
    Lin.apply(x @ W2.detach().t(), W1, False).sum().backward()
    assert launches == [1] and torch.equal(W1.grad, want1)
    # (4) nobody holds the result any more at flush time: skipped, not written into freed memory
    launches.clear()
    t = torch.empty(2, 3)
    _native._deferred_gemms.append(((torch.ones(2, 4), None, torch.ones(4, 3), None), __import__('weakref').ref(t), (2, 3),
                                    (None, 1, 0, 0, 0), torch.nn.Parameter(torch.ones(2, 3))))
    del t
    _native.flush_deferred_gemms()
    assert launches == [] and not _native._deferred_gemms


def test_driver_line_is_bounded_and_round_trips():
    """the driver could not parse the 23 KB line of round 5 (BENCH_r05.json parsed: null).  The line
    bench.py prints is now bench.driver_line(result): the contract keys + roofline + cpu_baseline + parity + summary in
    at most 6 KB, everything else in the side file it names.  Canned results: every full line committed in r04 / r05."""
    import glob
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    assert bench.DRIVER_LINE_MAX_BYTES == 6144
    files = sorted(glob.glob(os.path.join(root, 'profiles', 'r0[45]_bench_c*.json')))
    assert len(files) >= 6
    for f in files:
        full = json.loads(open(f).read().strip().splitlines()[-1])
        line = bench.driver_line(full, 'gpurun_out/bench/bench_%s_full.json' % full['config']['name'])
        assert len(line) <= 6144 and '\n' not in line, (f, len(line))
        d = json.loads(line)
        assert json.loads(json.dumps(d)) == d
        for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                    'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'parity', 'details_file'):
            assert key in d, (f, key)
        assert ('summary' in d) == ('summary' in full)
        assert 'workload' in d['config'] and 'secondary' not in d and 'near_tie_list' not in d['parity']
        for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'instruction', 'avg_launch_us',
                    'algorithmic_bytes', 'pipe_busy_frac', 'executed_over_algorithmic_flops'):
            assert key in d['roofline'], (f, key)
        assert 'peak_note' not in d['roofline']
        for key in ('value', 'unit', 'cores', 'kind', 'sample', 'ms_per_step', 'speedup_gpu_over_cpu', 'c1_b1'):
            assert key in d['cpu_baseline'], (f, key)
        # six significant digits: the arithmetic of the contract still closes on the printed numbers
        assert abs(d['value'] - full['value']) <= 1e-5 * full['value']
        assert abs(d['roofline']['frac'] - d['roofline']['achieved'] / d['roofline']['peak']) <= 1e-5
    # a pathological result (huge summary) is cut down rather than printed oversize
    fat = dict(full, summary={'x': 'y' * 9000})
    line = bench.driver_line(fat, None)
    assert len(line) <= 6144 and 'summary' not in json.loads(line) and 'roofline' in json.loads(line)


def test_torch_ops_registration_shapes_and_loud_cpu_failure():
    """torch.ops.gnnpp.* (gnn_pathplanning_amd/ops.py): registered with fake implementations (output shapes / dtypes
    under FakeTensorMode, which is what torch.compile / export use) and -- like every entry of the package -- no CPU
    fallback: a real call with CPU tensors raises GnnppError."""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode
    import gnn_pathplanning_amd.ops  # noqa: F401
    from gnn_pathplanning_amd._native import GnnppError
    with FakeTensorMode():
        h, S, x = torch.empty(150, 2, 3, 64), torch.empty(7, 2, 10, 10), torch.empty(7, 64, 10)
        y = torch.ops.gnnpp.lsigf(h, S, x, torch.empty(150, 1), True, 0)
        assert tuple(y.shape) == (7, 150, 10) and y.dtype == torch.float32
        lg = torch.ops.gnnpp.policy_logits(torch.empty(7, 10, 3, 11, 11), torch.empty(7, 1, 10, 10, dtype=torch.float64),
                                           torch.empty(10), torch.empty(10), None, torch.empty(5, 128), torch.empty(5), 3)
        assert tuple(lg.shape) == (10, 7, 5)
        ids = torch.ops.gnnpp.decode_actions(lg)
        assert tuple(ids.shape) == (7, 10) and ids.dtype == torch.int32
    with pytest.raises(GnnppError):
        torch.ops.gnnpp.lsigf(torch.zeros(128, 1, 3, 128), torch.zeros(2, 1, 4, 4), torch.zeros(2, 128, 4), None)
    with pytest.raises(GnnppError):
        torch.ops.gnnpp.decode_actions(torch.zeros(4, 2, 5))
    # r04: lsigf carries a registered autograd formula (gnnpp::lsigf_backward): under FakeTensorMode -- what AOTAutograd
    # traces with -- the gradients of the taps, the signal and the bias have the inputs' shapes, the GSO gets none
    with FakeTensorMode():
        h = torch.empty(150, 2, 3, 64, requires_grad=True)
        x = torch.empty(7, 64, 10, requires_grad=True)
        b = torch.empty(150, 1, requires_grad=True)
        S = torch.empty(7, 2, 10, 10, requires_grad=True)
        y = torch.ops.gnnpp.lsigf(h, S, x, b, True, 0)
        assert y.requires_grad
        dh, dx, db, dS = torch.autograd.grad(y, [h, x, b, S], torch.empty_like(y), allow_unused=True)
        assert dh.shape == h.shape and dx.shape == x.shape and db.shape == b.shape and dS is None
        g3 = torch.ops.gnnpp.lsigf_backward(h, S, x, None, torch.empty_like(y), False, 0)
        assert g3[0].shape == h.shape and g3[1].shape == x.shape and g3[2].numel() == 0
        # r05: `needs` -- a gradient nobody asked for is an empty tensor (and is not computed)
        g4 = torch.ops.gnnpp.lsigf_backward(h, S, x, b, torch.empty_like(y), False, 0, 2)
        assert g4[0].numel() == 0 and g4[1].shape == x.shape and g4[2].numel() == 0
        xo = torch.empty(7, 64, 10, requires_grad=True)
        yo = torch.ops.gnnpp.lsigf(h.detach(), S, xo, b.detach(), False, 0)
        (dxo,) = torch.autograd.grad(yo, [xo], torch.empty_like(yo))
        assert dxo.shape == xo.shape
