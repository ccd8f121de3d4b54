"""Record rollout traces from the REAL reference simulator (utils/multirobotsim_dcenlocal.py) into
tests/golden/rollout_traces.npz.  Build-container only (needs /root/reference); only data is stored.

    python oracle/gen_golden_rollout.py          # rollout_traces.npz (+ the .mat / checkpoint fixtures)
    python oracle/gen_golden_rollout.py large    # rollout_traces_large.npz: teams of 50 and 100 agents

Every case is a hand-made MAPF instance (random obstacle map, distinct start/goal cells) driven
through setup / getCurrentState / getGSO / move exactly like agents/decentralplannerlocal.py:534-592.
The action logits come either from the reference DecentralPlannerNet (golden parameters) or from a
scripted noisy-greedy policy that provokes vertex collisions, swaps and obstacle bumps.  The outcome
of every random.choice inside the collision shielding is recorded so the step can be replayed.
"""
import json
import os
import random
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, 'tests', 'golden')


def import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import matplotlib
    matplotlib.use('Agg')
    stub = types.ModuleType('torchsummaryX')
    stub.summary = lambda *a, **k: None
    sys.modules['torchsummaryX'] = stub
    for name, path in (('utils', REF + '/utils'), ('utils.graphUtils', REF + '/utils/graphUtils'),
                       ('dataloader', REF + '/dataloader')):
        pkg = types.ModuleType(name)
        pkg.__path__ = [path]
        sys.modules[name] = pkg
    import utils.multirobotsim_dcenlocal as simmod
    from graphs.models.decentralplanner import DecentralPlannerNet
    return simmod, DecentralPlannerNet


class Cfg:
    def __init__(self, n, k=3, commR=6, rate=2):
        self.num_agents, self.nGraphFilterTaps = n, k
        self.device = torch.device('cpu')
        self.rate_maxstep, self.commR = rate, commR


def make_case(rng, N, W, density):
    grid = (rng.random((W, W)) < density).astype(np.float32)
    free = np.argwhere(grid == 0)
    idx = rng.choice(len(free), size=2 * N, replace=False)
    return grid, free[idx[:N]], free[idx[N:]]


def greedy_logits(rng, cur, goal, noise):
    """[N,5] logits: step that reduces the Manhattan distance (row first), random with prob `noise`."""
    N = len(cur)
    out = np.zeros((N, 5), dtype=np.float32)
    for n in range(N):
        dx, dy = goal[n][0] - cur[n][0], goal[n][1] - cur[n][1]
        if rng.random() < noise:
            a = int(rng.integers(0, 5))
        elif dx != 0 and (dy == 0 or rng.random() < 0.5):
            a = 2 if dx > 0 else 0
        elif dy != 0:
            a = 3 if dy > 0 else 1
        else:
            a = 4
        out[n] = rng.normal(0, 0.1, 5)
        out[n, a] += 2.0
    return out


def run_case(simmod, net, cfg, grid, starts, goals, makespan, policy, rng, noise):
    N = cfg.num_agents
    inp = torch.tensor(np.stack([goals, starts])[None].astype(np.float32))      # [1,2,N,2]
    tgt = torch.zeros(1, N, makespan, 5)
    tgt[..., 4] = 1                                                             # expert "stays"
    sim = simmod.multiRobotSim(cfg)
    choices = []
    orig = random.choice

    def recording_choice(seq):
        r = orig(seq)
        choices.append(seq.index(r))
        return r
    simmod.random.choice = recording_choice
    sim.setup(inp, tgt, torch.tensor([makespan]), torch.tensor(grid[None]), 0)
    rec = {k: [] for k in ('pos', 'obs', 'gso', 'radius', 'logits', 'actions', 'flags', 'reached',
                           'choices', 'nchoices')}

    def positions():
        return np.array([[int(v) for v in sim.status_MultiAgent['agent%d' % i]['currentState'][0]]
                         for i in range(N)], dtype=np.int16)
    rec['pos'].append(positions())
    for step in range(sim.getMaxstep()):
        state = sim.getCurrentState()
        gso = sim.getGSO(step)
        if policy == 'model':
            with torch.no_grad():
                net.addGSO(gso)
                av = net(state)
            logits = torch.stack(av, 1)[0].numpy()
        else:
            logits = greedy_logits(rng, positions(), goals, noise)
            av = [torch.from_numpy(logits[n:n + 1]) for n in range(N)]
        n0 = len(choices)
        all_reach, mv, pr = sim.move(av, step + 1)
        rec['obs'].append(state[0].numpy().astype(np.uint8))
        rec['gso'].append(gso[0].numpy())
        rec['radius'].append(sim.communicationRadius)
        rec['logits'].append(logits.astype(np.float32))
        rec['actions'].append(np.array([int(np.argmax(logits[n])) for n in range(N)], dtype=np.int8))
        rec['flags'].append([int(all_reach), int(mv), int(pr)])
        rec['reached'].append(np.array(sim.count_reachgoal, dtype=np.uint8))
        rec['choices'] += choices[n0:]
        rec['nchoices'].append(len(choices) - n0)
        rec['pos'].append(positions())
        if all_reach:
            break
    simmod.random.choice = orig
    end = [sim.status_MultiAgent['agent%d' % i]['endStep_action_predict'] for i in range(N)]
    start = [sim.status_MultiAgent['agent%d' % i]['startStep_action_predict'] for i in range(N)]
    fin = {'makespan': int(sim.makespanPredict), 'flowtime': int(sim.flowtimePredict),
           'maxstep': int(sim.getMaxstep()),
           'end_step': [-1 if e is None else int(e) for e in end],
           'start_step': [-1 if s is None else int(s) for s in start]}
    return rec, fin


def pack_case(store, meta, ci, cfg, grid, goals, rec, fin, policy):
    T = len(rec['obs'])
    store['t%d_grid' % ci] = grid.astype(np.uint8)
    store['t%d_goal' % ci] = goals.astype(np.int16)
    store['t%d_pos' % ci] = np.stack(rec['pos'])                 # [T+1,N,2]
    store['t%d_obs' % ci] = np.stack(rec['obs'])                 # [T,N,3,11,11] uint8
    store['t%d_gso' % ci] = np.stack(rec['gso'])                 # [T,N,N] float64
    store['t%d_radius' % ci] = np.array(rec['radius'])
    store['t%d_logits' % ci] = np.stack(rec['logits'])
    store['t%d_actions' % ci] = np.stack(rec['actions'])
    store['t%d_flags' % ci] = np.array(rec['flags'], dtype=np.uint8)
    store['t%d_reached' % ci] = np.stack(rec['reached'])
    store['t%d_choices' % ci] = np.array(rec['choices'], dtype=np.int16)
    store['t%d_nchoices' % ci] = np.array(rec['nchoices'], dtype=np.int16)
    m = {'N': cfg.num_agents, 'W': int(grid.shape[0]), 'policy': policy, 'T': T, 'commR': cfg.commR,
         'collisions': int(sum(rec['nchoices']))}
    m.update(fin)
    meta.append(m)
    print(m)


def main_large():
    """Teams of 50 and 100 agents on 50 x 50 / 100 x 100 maps (the rollouts of BASELINE configs 3 and 5) through the
    reference simulator, scripted noisy-greedy policy -> rollout_traces_large.npz."""
    simmod, _ = import_reference()
    rng = np.random.default_rng(20260927)
    random.seed(4242)
    store, meta = {}, []
    for ci, (N, W, dens, mk, noise) in enumerate(((50, 50, 0.05, 10, 0.2), (100, 100, 0.03, 8, 0.2))):
        cfg = Cfg(N)
        grid, starts, goals = make_case(rng, N, W, dens)
        rec, fin = run_case(simmod, None, cfg, grid, starts, goals, mk, 'greedy', rng, noise)
        pack_case(store, meta, ci, cfg, grid, goals, rec, fin, 'greedy')
    store['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, 'rollout_traces_large.npz'), **store)


def main():
    simmod, Net = import_reference()
    z = np.load(os.path.join(OUT, 'policy_model.npz'))
    sd = {k[3:]: torch.from_numpy(np.array(z[k])) for k in z.files if k.startswith('sd/')}
    rng = np.random.default_rng(20260926)
    random.seed(1337)
    specs = [  # (N, W, density, makespan, policy, noise)
        (10, 20, 0.10, 7, 'model', 0.0),
        (10, 20, 0.10, 16, 'greedy', 0.25),
        (6, 8, 0.15, 12, 'greedy', 0.35),
        (23, 30, 0.08, 12, 'greedy', 0.20),
        (12, 10, 0.05, 10, 'greedy', 0.30),
        (2, 6, 0.0, 6, 'greedy', 0.5),
    ]
    store, meta = {}, []
    for ci, (N, W, dens, mk, policy, noise) in enumerate(specs):
        cfg = Cfg(N)
        net = None
        if policy == 'model':
            net = Net(cfg).eval()
            net.load_state_dict(sd)
        grid, starts, goals = make_case(rng, N, W, dens)
        rec, fin = run_case(simmod, net, cfg, grid, starts, goals, mk, policy, rng, noise)
        T = len(rec['obs'])
        store['t%d_grid' % ci] = grid.astype(np.uint8)
        store['t%d_goal' % ci] = goals.astype(np.int16)
        store['t%d_pos' % ci] = np.stack(rec['pos'])                 # [T+1,N,2]
        store['t%d_obs' % ci] = np.stack(rec['obs'])                 # [T,N,3,11,11] uint8
        store['t%d_gso' % ci] = np.stack(rec['gso'])                 # [T,N,N] float64
        store['t%d_radius' % ci] = np.array(rec['radius'])
        store['t%d_logits' % ci] = np.stack(rec['logits'])
        store['t%d_actions' % ci] = np.stack(rec['actions'])
        store['t%d_flags' % ci] = np.array(rec['flags'], dtype=np.uint8)
        store['t%d_reached' % ci] = np.stack(rec['reached'])
        store['t%d_choices' % ci] = np.array(rec['choices'], dtype=np.int16)
        store['t%d_nchoices' % ci] = np.array(rec['nchoices'], dtype=np.int16)
        m = {'N': N, 'W': W, 'policy': policy, 'T': T, 'commR': cfg.commR,
             'collisions': int(sum(rec['nchoices']))}
        m.update(fin)
        meta.append(m)
        print(m)
    store['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, 'rollout_traces.npz'), **store)

    # ---- a dataset case in the reference's .mat format, read back by the REFERENCE's loader ------
    import scipy.io as sio
    from dataloader.statetransformer import AgentState
    ed = types.ModuleType('easydict')
    ed.EasyDict = dict
    sys.modules.setdefault('easydict', ed)             # imported by the loader module, unused here
    import dataloader.Dataloader_dcplocal_notTF_onlineExpert as dl
    ci = 2                                                 # the 6-agent 8x8 trace
    T = meta[ci]['T']
    acts = store['t%d_actions' % ci].astype(np.int64)      # [T,N]
    case = {'map': store['t%d_grid' % ci].astype(np.float64), 'goal': store['t%d_goal' % ci].astype(np.int64),
            'inputState': store['t%d_pos' % ci][:T].astype(np.int64),
            'inputTensor': store['t%d_obs' % ci].astype(np.float64),
            'target': np.eye(5, dtype=np.int64)[acts], 'GSO': store['t%d_gso' % ci],
            'makespan': T}
    mat = os.path.join(OUT, 'case_fixture.mat')
    sio.savemat(mat, case, do_compression=True)

    class FakeLoader:
        pass
    fl = FakeLoader()
    fl.AgentState = AgentState(meta[ci]['N'])
    tr = dl.CreateDataset.load_train_data(fl, mat, 3)
    te = dl.CreateDataset.load_data_during_training(fl, mat, 0)
    np.savez_compressed(os.path.join(OUT, 'case_fixture_expected.npz'),
                        train_input=tr[0].numpy(), train_target=tr[1].numpy(), train_gso=tr[2].numpy(),
                        train_map=tr[3].numpy(), test_input=te[0].numpy(), test_target=te[1].numpy(),
                        test_map=te[3].numpy())
    print('case fixture written', os.path.getsize(mat))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'large':
        main_large()
    else:
        main()
