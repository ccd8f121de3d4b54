"""Static check of the split-f16 encoder's weight ring in the generated gfx950 ISA.

encoder_kernel_h2.hip keeps its weight ring in v[208:255], registers the compiler is told not to
allocate (amdgpu_num_vgpr(104), doubled by the backend); inline asm loads them (`global_load_dwordx4 ... ; RINGLOAD s`),
waits (`s_waitcnt vmcnt(n) ; RINGWAIT`) and copies a fragment out (`v_mov_b64 ... ; RINGTAKE s`).
This script verifies on the ISA hipcc produced that
  1. no other instruction of the kernel mentions v208..v255,
  2. along every control-flow path, a RINGTAKE of slot s follows a load of that slot and a RINGWAIT
     whose vmcnt is <= the number of ring loads issued after that load (loads return in order), and
     a slot is never reloaded before it was taken,
  3. the kernel allocates 256 VGPRs, spills nothing and keeps two waves per SIMD.

Every instantiation is checked: encoder_kernel_h2<false, 3> (196 stream items) and the fused policy
kernels encoder_kernel_h2<true, K> for K = 2, 3, 4 filter taps (196 + 16 K items).

    python tools/check_ring_isa.py file.s [mangled-name-substring]   (exit status 1 on any violation)
"""
import re
import sys

REG = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')
RING_LO, NSLOT = 208, 12


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def kernel_text(path, name):
    """name: a substring of the mangled kernel symbol, e.g. 'encoder_kernel_h2ILb0' (first match)."""
    lines, on, meta = [], False, {}
    for ln in open(path):
        if not on and re.match(r'^_Z\w*%s\w*:' % re.escape(name), ln):
            on = True
            continue
        if on:
            lines.append(ln.rstrip('\n'))
            for key in ('NumVgprs', 'ScratchSize', 'Occupancy'):
                m = re.match(r'^; %s: (\d+)' % key, ln)
                if m:
                    meta[key] = int(m.group(1))
            if 'Occupancy' in meta:
                break
    return lines, meta


def check(path, name='encoder_kernel_h2ILb0', max_scratch=0):
    """max_scratch: bytes of scratch per lane tolerated.  Scratch traffic cannot break the ring (loads
    return in order, so `vmcnt <= younger ring loads` can only hold once the awaited load is back,
    whatever stores are outstanding); it is a performance guard: 0 for the encoder, a few bytes for the
    fused policy kernel whose simulator tail keeps small indexed arrays."""
    lines, meta = kernel_text(path, name)
    if not lines:
        raise SystemExit('kernel %s not found in %s' % (name, path))
    errors = []
    if meta.get('NumVgprs') != 256 or meta.get('ScratchSize', 1 << 30) > max_scratch or meta.get('Occupancy') != 2:
        errors.append('resource usage %r (want NumVgprs 256, ScratchSize <= %d, Occupancy 2)' % (meta, max_scratch))
    blocks, cur, label_of = [], {'label': None, 'ins': []}, {}
    for ln in lines:
        t = ln.strip()
        if t.startswith('.Lfunc_end'):
            break
        if not t or t.startswith(';') or (t.startswith('.') and not t.startswith('.LBB')):
            continue
        m = re.match(r'^(\.LBB\w+):', t)
        if m:
            if cur['ins'] or cur['label']:
                blocks.append(cur)
            cur = {'label': m.group(1), 'ins': []}
            continue
        cur['ins'].append(t)
        if t.startswith(('s_cbranch', 's_branch', 's_endpgm')):
            blocks.append(cur)
            cur = {'label': None, 'ins': []}
    if cur['ins'] or cur['label']:
        blocks.append(cur)
    for i, b in enumerate(blocks):
        if b['label']:
            label_of[b['label']] = i
    succ = []
    for i, b in enumerate(blocks):
        last = b['ins'][-1] if b['ins'] else ''
        s = []
        if last.startswith('s_branch'):
            s.append(label_of[last.split()[1]])
        elif last.startswith('s_cbranch'):
            s.append(label_of[last.split()[1]])
            if i + 1 < len(blocks):
                s.append(i + 1)
        elif not last.startswith('s_endpgm') and i + 1 < len(blocks):
            s.append(i + 1)
        succ.append(s)
    # state per slot: age = ring loads issued after this slot's load (None = slot empty / taken)
    EMPTY = -1
    entry = [None] * len(blocks)
    entry[0] = tuple([EMPTY] * NSLOT)
    work, reported = [0], set()
    stats = {'loads': 0, 'takes': 0}
    counted = set()

    def err(i, k, msg):
        if (i, k) not in reported:
            reported.add((i, k))
            errors.append('block %d: %s' % (i, msg))

    while work:
        i = work.pop()
        age = list(entry[i])
        last_wait = None
        for k, t in enumerate(blocks[i]['ins']):
            code = t.split(';')[0]
            if 'RINGLOAD' in t:
                s = int(t.split('RINGLOAD')[1])
                if regs_of(code.split(',')[0]) != set(range(RING_LO + 4 * s, RING_LO + 4 * s + 4)):
                    err(i, k, 'load into the wrong registers: ' + t)
                if age[s] != EMPTY:
                    err(i, k, 'slot %d reloaded before it was taken: %s' % (s, t))
                age = [a + 1 if a != EMPTY else a for a in age]
                age[s] = 0
                if (i, k) not in counted:
                    counted.add((i, k)); stats['loads'] += 1
            elif 'RINGWAIT' in t:
                last_wait = int(re.search(r'vmcnt\((\d+)\)', code).group(1))
            elif 'RINGTAKE' in t:
                s = int(t.split('RINGTAKE')[1])
                if not regs_of(code.split(',')[1]) <= set(range(RING_LO + 4 * s, RING_LO + 4 * s + 4)):
                    err(i, k, 'take from the wrong registers: ' + t)
                if age[s] == EMPTY and not blocks[i]['ins'][k - 1].endswith('RINGTAKE %d' % s):
                    err(i, k, 'slot %d taken but not loaded: %s' % (s, t))
                elif age[s] != EMPTY and (last_wait is None or last_wait > age[s]):
                    err(i, k, 'slot %d taken after vmcnt(%s) but only %d younger loads: %s'
                        % (s, last_wait, age[s], t))
                if blocks[i]['ins'][k - 1].endswith('RINGTAKE %d' % s):
                    age[s] = EMPTY                      # second half of the fragment: slot is free
                    if (i, k) not in counted:
                        counted.add((i, k)); stats['takes'] += 1
            else:
                bad = [r for r in regs_of(code) if r >= RING_LO]
                if bad:
                    err(i, k, 'compiler code touches ring registers v%s: %s' % (sorted(bad), t))
                if not code.startswith(('v_', 'ds_', 's_nop', 's_mov', 's_add', 's_lshl', 's_and', 's_cmp',
                                        's_mul', 's_sub', 's_or', 's_cselect')):
                    if code.startswith('s_waitcnt') and 'vmcnt' in code:
                        m = re.search(r'vmcnt\((\d+)\)', code)
                        last_wait = min(last_wait, int(m.group(1))) if last_wait is not None else int(m.group(1))
                    elif code.startswith(('s_barrier', 's_cbranch', 's_branch', 's_endpgm', 's_waitcnt',
                                          's_load', 'global_', 'buffer_', 'scratch_', 's_')):
                        pass
        out = tuple(age)
        for j in succ[i]:
            if entry[j] is None:
                new = out
            else:       # merge: a slot must agree on being empty; keep the smaller age (stricter)
                new = tuple(EMPTY if (a == EMPTY and b == EMPTY) else
                            (min(a, b) if a != EMPTY and b != EMPTY else -2) for a, b in zip(entry[j], out))
                exit_only = any(t.startswith('s_endpgm') for t in blocks[j]['ins']) and \
                    not any('RING' in t for t in blocks[j]['ins'])
                if -2 in new and not exit_only:      # (early returns may leave loads pending)
                    err(j, -1, 'paths disagree on which ring slots are pending at block %d' % j)
                    new = tuple(EMPTY if x == -2 else x for x in new)
            if new != entry[j]:
                entry[j] = new
                work.append(j)
    return errors, stats, meta


if __name__ == '__main__':
    bad = 0
    for kern in (sys.argv[2:3] or ['encoder_kernel_h2ILb0ELi3E', 'encoder_kernel_h2ILb1ELi2E',
                                   'encoder_kernel_h2ILb1ELi3E', 'encoder_kernel_h2ILb1ELi4E']):
        errs, st, meta = check(sys.argv[1], kern, 0 if 'ILb0' in kern else 32)
        print('%s: ring loads: %d, fragments taken: %d, %r, violations: %d'
              % (kern, st['loads'], st['takes'], meta, len(errs)))
        for e in errs[:40]:
            print('  ' + e)
        bad += len(errs)
    sys.exit(1 if bad else 0)
