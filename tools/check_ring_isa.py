"""Static check of the split-f16 encoder's weight ring in the generated gfx950 ISA.

encoder_kernel_h2.hip keeps its weight ring in v[208:255], registers the compiler is told not to
allocate (amdgpu_num_vgpr(104), doubled by the backend); inline asm loads them (`global_load_dwordx4 ... ; RINGLOAD s`),
waits (`s_waitcnt vmcnt(n) ; RINGWAIT`) and copies a fragment out (`v_mov_b64 ... ; RINGTAKE s`).
This script verifies on the ISA hipcc produced that
  1. no other instruction of the kernel mentions v208..v255,
  2. along every control-flow path, a RINGTAKE of slot s follows a load of that slot and a RINGWAIT
     whose vmcnt is <= the number of ring loads issued after that load (loads return in order), and
     a slot is never reloaded before it was taken -- or, in the DIRECT form of encoder_kernel_b3.hip's column-packed
     layers (`v_mfma ... ; RINGUSE s`: the slot is the MFMA's A operand), before it was read, every such read behind a
     wait that covers the slot's load,
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
    # PATH-SENSITIVE walk of the control-flow graph.  A state = (age of every ring slot: ring loads issued after that
    # slot's load, EMPTY = slot empty / taken; ring loads and fragments taken along the path; an over-estimate of the
    # EXEC narrowing depth; SGPR pairs known to hold 0 / -1; what is known about vcc).  States are NOT merged at
    # joins: a block is re-walked for every distinct state that reaches it (a few per block -- the arms of a
    # per-wave switch each carry their own copy of a layer's ring traffic and must leave identical states behind).
    EMPTY = -1
    start = (tuple([EMPTY] * NSLOT), 0, 0, 0, frozenset(), None, 0, 0)
    seen = [set() for _ in blocks]
    seen[0].add(start)
    work, reported = [(0, start)], set()
    stats = {'loads': 0, 'takes': 0}
    exits = set()

    def err(i, k, msg):
        if (i, k) not in reported:
            reported.add((i, k))
            errors.append('block %d: %s' % (i, msg))

    steps = 0
    while work:
        i, st0 = work.pop()
        steps += 1
        if steps > 200000:
            errors.append('analysis did not converge (ring traffic inside a loop?)')
            break
        age, nl, nt, depth = list(st0[0]), st0[1], st0[2], st0[3]
        consts, vcc = dict(st0[4]), st0[5]
        ready, used = st0[6], st0[7]                       # bit masks over the slots: load known to have landed | read by a
        last_wait = None                                   # RINGUSE (direct form) since its load
        exec_written = False
        for k, t in enumerate(blocks[i]['ins']):
            code = t.split(';')[0]
            if 'RINGLOAD' in t:
                s = int(t.split('RINGLOAD')[1])
                if regs_of(code.split(',')[0]) != set(range(RING_LO + 4 * s, RING_LO + 4 * s + 4)):
                    err(i, k, 'load into the wrong registers: ' + t)
                if age[s] != EMPTY and not (used >> s) & 1:
                    err(i, k, 'slot %d reloaded before it was taken / used: %s' % (s, t))
                age = [a + 1 if a != EMPTY else a for a in age]
                age[s] = 0
                ready &= ~(1 << s)
                used &= ~(1 << s)
                nl += 1
            elif 'RINGWAIT' in t:
                last_wait = int(re.search(r'vmcnt\((\d+)\)', code).group(1))
                for s2 in range(NSLOT):                    # loads return in order: everything with >= n younger loads is back
                    if age[s2] != EMPTY and age[s2] >= last_wait:
                        ready |= 1 << s2
            elif 'RINGUSE' in t:
                # direct form: a v_mfma whose A operand is the slot (the accumulator / B operands are the compiler's)
                s = int(t.split('RINGUSE')[1])
                ops = [o.strip() for o in code.split(None, 1)[1].split(',')]
                if not code.strip().startswith('v_mfma') or regs_of(ops[1]) != set(range(RING_LO + 4 * s, RING_LO + 4 * s + 4)) \
                        or any(r >= RING_LO for o in (ops[0], ops[2], ops[3]) for r in regs_of(o)):
                    err(i, k, 'use of the wrong registers: ' + t)
                if age[s] == EMPTY:
                    err(i, k, 'slot %d used but not loaded: %s' % (s, t))
                elif not (ready >> s) & 1:
                    err(i, k, 'slot %d used before a sufficient wait (%d younger loads): %s' % (s, age[s], t))
                if age[s] != EMPTY and not (used >> s) & 1:
                    used |= 1 << s
                    nt += 1
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
                    ready &= ~(1 << s)
                    nt += 1
            else:
                # EXEC narrowing: `depth` over-estimates how many exec-narrowing writes are unmatched on this path
                # (capped, so loops converge); `exec_written` = this block wrote EXEC before its terminator
                if re.match(r'^(s_and_saveexec_b64|s_andn2_saveexec_b64|s_and_b64\s+exec|s_andn2_b64\s+exec|'
                            r's_xor_b64\s+exec|v_cmpx_)', code):
                    depth = min(depth + 1, 4)
                    exec_written = True
                elif re.match(r'^(s_or_b64\s+exec,\s*exec,|s_mov_b64\s+exec,)', code):
                    depth = max(0, depth - 1)
                    exec_written = True
                elif re.match(r'^s_or_saveexec_b64', code):
                    exec_written = True
                elif code.startswith('s_barrier'):
                    depth = 0                              # __syncthreads() is only ever reached with every lane
                                                           # active (workgroup-uniform code): the estimate restarts
                # The structuriser lowers a wave-uniform if / else as a flag in an SGPR pair (s_mov_b64 sN, 0 | -1) and
                # `s_and[n2]_b64 vcc, exec, sN; s_cbranch_vcc[n]z`: follow the flag, so that the two arms are not
                # strung into one (infeasible) path.  Anything else that writes the pair or vcc forgets it.
                cs = code.strip()
                mm = re.match(r'^s_mov_b64\s+s\[(\d+):(\d+)\],\s*(0|-1)\s*$', cs)
                ma = re.match(r'^s_(and|andn2)_b64\s+vcc,\s*exec,\s*s\[(\d+):(\d+)\]\s*$', cs)
                first = re.match(r'^[sv]_\w+\s+(vcc|s\[(\d+):(\d+)\]|s(\d+))', cs)
                if mm:
                    consts[(int(mm.group(1)), int(mm.group(2)))] = int(mm.group(3))
                elif ma and (int(ma.group(2)), int(ma.group(3))) in consts and depth == 0:
                    c = consts[(int(ma.group(2)), int(ma.group(3)))]
                    vcc = ('nz' if c == -1 else 'z') if ma.group(1) == 'and' else ('z' if c == -1 else 'nz')
                elif first and not cs.startswith(('s_cmp', 's_cbranch', 's_branch', 's_waitcnt', 's_nop', 's_barrier',
                                                  's_endpgm')):
                    if first.group(1) == 'vcc':
                        vcc = None
                    else:
                        lo = int(first.group(2) if first.group(2) is not None else first.group(4))
                        hi = int(first.group(3) if first.group(3) is not None else first.group(4))
                        for key in [k2 for k2 in consts if not (k2[1] < lo or k2[0] > hi)]:
                            del consts[key]
                bad = [r for r in regs_of(code) if r >= RING_LO]
                if bad:
                    err(i, k, 'compiler code touches ring registers v%s: %s' % (sorted(bad), t))
                if cs.startswith('s_waitcnt') and 'vmcnt' in cs:
                    m = re.search(r'vmcnt\((\d+)\)', cs)
                    last_wait = min(last_wait, int(m.group(1))) if last_wait is not None else int(m.group(1))
                    for s2 in range(NSLOT):
                        if age[s2] != EMPTY and age[s2] >= int(m.group(1)):
                            ready |= 1 << s2
        last_ins = blocks[i]['ins'][-1] if blocks[i]['ins'] else ''
        if last_ins.startswith('s_endpgm'):
            exits.add((nl, nt))
        # With every lane active (depth 0: the kernels run full 64-lane waves and the ring traffic sits in
        # wave-uniform code) EXEC is not zero, so `s_cbranch_execnz` -- which hipcc emits as the jump out of a
        # uniform switch's case -- is always taken and `s_cbranch_execz` never: following their other edge would
        # string several cases of a per-wave switch into one (infeasible) path.
        nexts = succ[i]
        if depth == 0 and not exec_written and last_ins.startswith('s_cbranch_execnz'):
            nexts = succ[i][:1]
        elif depth == 0 and not exec_written and last_ins.startswith('s_cbranch_execz'):
            nexts = succ[i][1:]
        elif last_ins.startswith('s_cbranch_vccnz') and vcc is not None:
            nexts = succ[i][:1] if vcc == 'nz' else succ[i][1:]
        elif last_ins.startswith('s_cbranch_vccz') and vcc is not None:
            nexts = succ[i][:1] if vcc == 'z' else succ[i][1:]
        if all(a == EMPTY for a in age):
            # no ring load in flight (before the prologue / behind the last take: the simulator tail's divergent code):
            # what is known about EXEC and the flags cannot matter to the ring any more -- collapse the states
            depth, consts, vcc = 4, {}, None
        out = (tuple(age), nl, nt, depth, frozenset(consts.items()), vcc, ready, used)
        for j in nexts:
            if out not in seen[j]:
                if len(seen[j]) > 64:
                    err(j, -1, 'more than 64 distinct ring states reach block %d' % j)
                    continue
                seen[j].add(out)
                work.append((j, out))
    # every path to the end of the kernel that consumed the whole stream did it exactly once; early exits (the
    # measurement build's phase stops, nothing in the product) may leave loads pending
    if exits:
        stats['loads'], stats['takes'] = max(e[0] for e in exits), max(e[1] for e in exits)
        full = {e for e in exits if e[1] == stats['takes']}
        if len(full) != 1:
            errors.append('paths through the whole stream disagree on the ring traffic: %r' % sorted(exits))
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
