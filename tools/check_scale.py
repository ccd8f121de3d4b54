"""Fail LOUDLY when a scaling session did not measure what it claims (tools/scale_session.sh runs this on its own
output):  python tools/check_scale.py scale_policy.jsonl [--reference profiles/r03_bench_c2.json] [--allow-shared]
  * every line of an N-GPU run reports n_gpus == ranks_in_group == N and N DISTINCT physical devices
    (`rank_devices`: uuid / PCI bus id per rank) -- two ranks on one GPU are not a 2-GPU measurement;
  * the N = 1 line agrees with the committed 1-GPU bench line within 5 % (same metric, same config);
  * values are positive and the weak-scaling whole-job value does not DEcrease with N.
Exit status 1 with the reasons on stderr; prints the per-N values (no efficiency: the driver computes that)."""
import json
import sys


def check(lines, reference=None, allow_shared=False):
    errs, seen = [], {}
    for d in lines:
        n = d.get('n_gpus')
        if n is None or d.get('ranks_in_group') != n:
            errs.append('N=%s: ranks_in_group=%s' % (n, d.get('ranks_in_group')))
        devs = d.get('rank_devices') or []
        if len(devs) != n:
            errs.append('N=%s: %d rank_devices reported' % (n, len(devs)))
        elif len(set(devs)) != n and not allow_shared:
            errs.append('N=%s: ranks share physical devices: %s' % (n, devs))
        if not d.get('value', 0) > 0:
            errs.append('N=%s: value=%s' % (n, d.get('value')))
        if n in seen:
            errs.append('N=%s measured twice' % n)
        seen[n] = d
    order = sorted(seen)
    for a, b in zip(order[:-1], order[1:]):
        if seen[b]['value'] < seen[a]['value']:
            errs.append('whole-job value drops from N=%d (%.4g) to N=%d (%.4g)' % (a, seen[a]['value'], b, seen[b]['value']))
    if reference is not None and 1 in seen:
        r, v = reference['value'], seen[1]['value']
        if reference.get('metric') == seen[1].get('metric') and abs(v - r) > 0.05 * r:
            errs.append('N=1 value %.4g is not within 5 %% of the committed line (%.4g)' % (v, r))
    return errs, {n: seen[n]['value'] for n in order}


if __name__ == '__main__':
    args = sys.argv[1:]
    allow = '--allow-shared' in args
    ref = None
    if '--reference' in args:
        ref = json.loads(open(args[args.index('--reference') + 1]).read().strip().splitlines()[-1])
    lines = [json.loads(l) for l in open(args[0]) if l.strip().startswith('{')]
    if not lines:
        sys.exit('no JSON lines in %s' % args[0])
    errs, vals = check(lines, ref, allow)
    print(json.dumps({'per_n_value': vals}))
    if errs:
        sys.stderr.write('SCALING SESSION INVALID:\n  ' + '\n  '.join(errs) + '\n')
        sys.exit(1)
