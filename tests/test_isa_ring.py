"""Static safety check of the split-f16 encoder's inline-asm weight ring (CPU only: hipcc
cross-compiles gfx950 without a GPU).  The ring lives in v[208:255], registers the compiler must
never touch; tools/check_ring_isa.py walks the kernel's control-flow graph in the generated ISA and
verifies that, plus the load -> wait -> take discipline and the register/occupancy budget."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc')
def test_ring_registers_are_private_to_the_asm(tmp_path):
    import check_ring_isa
    # The ISA of the very compilation that produced libgnnpp.so is kept by _native.build() (-save-temps); when it is
    # at least as new as every source it IS the code that ships: check that.  Otherwise compile here (two minutes).
    csrc = os.path.join(ROOT, 'gnn_pathplanning_amd', 'csrc')
    kept = os.path.join(ROOT, 'gnn_pathplanning_amd', 'build', 'product', 'gnnpp_api-hip-amdgcn-amd-amdhsa-gfx950.s')
    newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc))
    newest = max(newest, os.path.getmtime(os.path.join(ROOT, 'include', 'gnnpp.h')))
    if os.path.exists(kept) and os.path.getmtime(kept) >= newest:
        out = kept
    else:
        out = str(tmp_path / 'gnnpp.s')
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only',
                               '-Wno-unused-result', '-w', os.path.join(csrc, 'gnnpp_api.hip'), '-o', out])
    # encoder only (196 stream items) and the fused policy kernels (+ 16 filter-tap fragments per tap, K = 2, 3, 4)
    from gnn_pathplanning_amd._native import RING_KERNELS as MUST, RING_KERNELS_OPTIONAL as OPT
    RING_KERNELS = OPT + MUST                              # (the shipped toolchain passes the opt-in kernels too)
    # split-f16: 196 + 16 K; bf16x3 (default): 294 + 24 K; r04: the column-packed forms of the bf16x3 kernels (fused
    # K = 2, 3, 4 for teams of <= 12 agents; the encoder's latency form) stream exactly the same items
    assert [k[1] for k in RING_KERNELS] == [196, 228, 244, 260, 294, 342, 366, 390, 342, 366, 390, 294]
    for kern, items, scratch in RING_KERNELS:
        errors, stats, meta = check_ring_isa.check(out, kern, scratch)
        assert not errors, (kern, errors[:10])
        assert stats['loads'] == items and stats['takes'] == items, (kern, stats)   # each item exactly once
        assert meta['NumVgprs'] == 256 and meta['Occupancy'] == 2 and meta['ScratchSize'] <= scratch, (kern, meta)


def test_checker_catches_violations(tmp_path):
    """The checker itself: a compiler-looking instruction that touches a ring register, a take
    with too weak a wait, and a reload of a pending slot must all be reported."""
    import check_ring_isa
    good = '''_ZN5gnnpp17encoder_kernel_h2ILb0EEEvPKfS2_Pfii:
\tglobal_load_dwordx4 v[208:211], v1, s[0:1] ; RINGLOAD 0
\tglobal_load_dwordx4 v[212:215], v1, s[0:1] ; RINGLOAD 1
\ts_waitcnt vmcnt(1) ; RINGWAIT
\tv_mov_b64 v[2:3], v[208:209] ; RINGTAKE 0
\tv_mov_b64 v[4:5], v[210:211] ; RINGTAKE 0
\ts_waitcnt vmcnt(0) ; RINGWAIT
\tv_mov_b64 v[2:3], v[212:213] ; RINGTAKE 1
\tv_mov_b64 v[4:5], v[214:215] ; RINGTAKE 1
\ts_endpgm
.Lfunc_end0:
; NumVgprs: 256
; ScratchSize: 0
; Occupancy: 2
'''
    f = tmp_path / 'a.s'
    f.write_text(good)
    assert check_ring_isa.check(str(f))[0] == []
    for bad in (good.replace('\ts_waitcnt vmcnt(1) ; RINGWAIT', '\tv_add_f32 v220, v209, v209\n\ts_waitcnt vmcnt(1) ; RINGWAIT'),
                good.replace('vmcnt(1) ; RINGWAIT', 'vmcnt(2) ; RINGWAIT'),
                good.replace('\ts_waitcnt vmcnt(1) ; RINGWAIT',
                             '\tglobal_load_dwordx4 v[208:211], v1, s[0:1] ; RINGLOAD 0\n\ts_waitcnt vmcnt(1) ; RINGWAIT'),
                good.replace('; ScratchSize: 0', '; ScratchSize: 16')):
        f.write_text(bad)
        assert check_ring_isa.check(str(f))[0], bad


def test_checker_follows_per_wave_switch_arms(tmp_path):
    """The bf16x3 kernel puts a whole layer's ring traffic inside each case of a per-wave switch: every arm has its own
    copy of the loads / waits / takes.  The walk is path sensitive: equal arms pass; an arm that consumes a different
    number of items, or takes with too weak a wait, is reported even though the other arm is fine."""
    import check_ring_isa
    arm = '''\tglobal_load_dwordx4 v[212:215], v1, s[0:1] ; RINGLOAD 1
\ts_waitcnt vmcnt(1) ; RINGWAIT
\tv_mov_b64 v[2:3], v[208:209] ; RINGTAKE 0
\tv_mov_b64 v[4:5], v[210:211] ; RINGTAKE 0
\ts_waitcnt vmcnt(0) ; RINGWAIT
\tv_mov_b64 v[2:3], v[212:213] ; RINGTAKE 1
\tv_mov_b64 v[4:5], v[214:215] ; RINGTAKE 1
'''
    text = ('_ZN5gnnpp17encoder_kernel_b3ILb0ELi3EEEvPKfS2_Pfii:\n'
            '\tglobal_load_dwordx4 v[208:211], v1, s[0:1] ; RINGLOAD 0\n'
            '\ts_cmp_eq_u32 s4, 0\n'
            '\ts_cbranch_scc1 .LBB0_2\n'
            + arm.replace('v[2:3]', 'v[6:7]') +
            '\ts_branch .LBB0_3\n'
            '.LBB0_2:\n' + arm +
            '.LBB0_3:\n'
            '\ts_endpgm\n'
            '.Lfunc_end0:\n; NumVgprs: 256\n; ScratchSize: 0\n; Occupancy: 2\n')
    f = tmp_path / 'sw.s'
    f.write_text(text)
    errors, stats, _ = check_ring_isa.check(str(f), 'encoder_kernel_b3ILb0')
    assert errors == [] and stats == {'loads': 2, 'takes': 2}
    second = text.rindex('\tglobal_load_dwordx4 v[212:215]')
    # (a) the second arm takes slot 0 behind vmcnt(1) only AFTER a second younger load was issued: fine; with
    #     vmcnt(2) it is not
    weak = text[:second] + text[second:].replace('vmcnt(1) ; RINGWAIT', 'vmcnt(2) ; RINGWAIT')
    f.write_text(weak)
    assert any('younger loads' in e for e in check_ring_isa.check(str(f), 'encoder_kernel_b3ILb0')[0])
    # (b) the second arm forgets its own load of slot 1 (the first arm has it): reported on that path only
    short = text[:second] + text[second:].replace('\tglobal_load_dwordx4 v[212:215], v1, s[0:1] ; RINGLOAD 1\n', '')
    f.write_text(short)
    errs = check_ring_isa.check(str(f), 'encoder_kernel_b3ILb0')[0]
    assert any('taken but not loaded' in e for e in errs), errs


def test_checker_follows_the_direct_form(tmp_path):
    """r04: the column-packed layers read a ring slot as the A operand of an inline-asm MFMA (`; RINGUSE s`) and refill it
    behind its last reader.  Accepted: load -> sufficient wait -> uses -> reload -> wait -> take.  Reported: a use before
    any sufficient wait, a use of a slot whose reload has not been waited for, a reload before the slot was read, an MFMA
    whose OTHER operands touch the ring, a use that names the wrong slot."""
    import check_ring_isa
    good = '''_ZN5gnnpp17encoder_kernel_b3ILb1ELi3ELb1EEEvPKfS2_Pfii:
\tglobal_load_dwordx4 v[208:211], v1, s[0:1] ; RINGLOAD 0
\tglobal_load_dwordx4 v[212:215], v1, s[0:1] ; RINGLOAD 1
\ts_waitcnt vmcnt(1) ; RINGWAIT
\tv_mfma_f32_16x16x32_bf16 v[0:3], v[208:211], v[4:7], 0 ; RINGUSE 0
\tglobal_load_dwordx4 v[216:219], v1, s[0:1] ; RINGLOAD 2
\tv_mfma_f32_16x16x32_bf16 v[0:3], v[208:211], v[4:7], v[0:3] ; RINGUSE 0
\tglobal_load_dwordx4 v[208:211], v1, s[0:1] ; RINGLOAD 0
\ts_waitcnt vmcnt(2) ; RINGWAIT
\tv_mfma_f32_16x16x32_bf16 v[0:3], v[212:215], v[4:7], v[0:3] ; RINGUSE 1
\ts_waitcnt vmcnt(1) ; RINGWAIT
\tv_mov_b64 v[8:9], v[216:217] ; RINGTAKE 2
\tv_mov_b64 v[10:11], v[218:219] ; RINGTAKE 2
\ts_waitcnt vmcnt(0) ; RINGWAIT
\tv_mfma_f32_16x16x32_bf16 v[0:3], v[208:211], v[4:7], v[0:3] ; RINGUSE 0
\ts_endpgm
.Lfunc_end0:
; NumVgprs: 256
; ScratchSize: 0
; Occupancy: 2
'''
    f = tmp_path / 'd.s'
    f.write_text(good)
    errors, stats, _ = check_ring_isa.check(str(f), 'encoder_kernel_b3ILb1')
    assert errors == [] and stats == {'loads': 4, 'takes': 4}, (errors, stats)
    cases = {
        'before a sufficient wait': good.replace('\ts_waitcnt vmcnt(1) ; RINGWAIT\n\tv_mfma_f32_16x16x32_bf16 v[0:3], v[208:211], v[4:7], 0',
                                                 '\ts_waitcnt vmcnt(2) ; RINGWAIT\n\tv_mfma_f32_16x16x32_bf16 v[0:3], v[208:211], v[4:7], 0'),
        # the last use of slot 0 reads the RELOADED item: vmcnt(0) removed -> its load is not known to have landed
        'before a sufficient wait ': good.replace('\ts_waitcnt vmcnt(0) ; RINGWAIT\n', ''),
        'reloaded before it was taken / used': good.replace('\tv_mfma_f32_16x16x32_bf16 v[0:3], v[212:215], v[4:7], v[0:3] ; RINGUSE 1\n',
                                                            '\tglobal_load_dwordx4 v[212:215], v1, s[0:1] ; RINGLOAD 1\n'),
        'wrong registers': good.replace('v[0:3], v[212:215], v[4:7], v[0:3] ; RINGUSE 1', 'v[0:3], v[212:215], v[216:219], v[0:3] ; RINGUSE 1'),
        'wrong registers ': good.replace('v[0:3], v[212:215], v[4:7], v[0:3] ; RINGUSE 1', 'v[0:3], v[212:215], v[4:7], v[0:3] ; RINGUSE 2'),
    }
    for what, bad in cases.items():
        assert bad != good, what
        f.write_text(bad)
        errs = check_ring_isa.check(str(f), 'encoder_kernel_b3ILb1')[0]
        assert any(what.strip() in e for e in errs), (what, errs)
