"""GPU (-m gpu): the batched rollout step through the C ABI / BatchedRollout against (i) the traces
of the real simulator, (ii) the CPU rollout oracle on fresh random episodes, (iii) the policy oracle
inside a closed-loop rollout.  Integer/boolean/fp64 stages are bit-exact; logits within 1e-4."""
import numpy as np
import pytest
import torch

from oracle import policy_oracle as orc
from oracle import rollout_oracle as ro

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    from gnn_pathplanning_amd import _native
    _native.lib()
    return torch.device('cuda:0')


def random_episodes(rng, B, N, W, density):
    grids, starts, goals = [], [], []
    for _ in range(B):
        g = (rng.random((W, W)) < density).astype(np.uint8)
        free = np.argwhere(g == 0)
        idx = rng.choice(len(free), size=2 * N, replace=False)
        grids.append(g); starts.append(free[idx[:N]]); goals.append(free[idx[N:]])
    return np.stack(grids), np.stack(starts), np.stack(goals)


def test_replay_simulator_traces(dev, rollout_golden):
    _replay_traces(dev, *rollout_golden)


def test_replay_simulator_traces_large_teams(dev, rollout_large_golden):
    """The reference simulator's own traces for 50 agents / 50 x 50 and 100 agents / 100 x 100 (the rollouts of
    BASELINE configs 3 and 5): observations, GSO, radius, collision shielding, statistics -- bit for bit."""
    z, meta = rollout_large_golden
    assert [m['N'] for m in meta] == [50, 100]
    _replay_traces(dev, z, meta)


def _replay_traces(dev, z, meta):
    from gnn_pathplanning_amd.rollout import BatchedRollout
    for ci, m in enumerate(meta):
        pos_all = z['t%d_pos' % ci].astype(np.int64)
        env = BatchedRollout(z['t%d_grid' % ci], pos_all[0][None], z['t%d_goal' % ci][None],
                             m['maxstep'], dev, commR=m['commR'], tie_mode='replay')
        used = 0
        for t in range(m['T']):
            assert (env.pos[0].cpu().numpy() == pos_all[t]).all(), (ci, t)
            assert (env.observe()[0].cpu().numpy() == z['t%d_obs' % ci][t]).all(), (ci, t)
            S = env.gso(t)[0].cpu().numpy()
            assert (S == z['t%d_gso' % ci][t].astype(np.float32)).all(), (ci, t)
            assert env.radius[0].item() == z['t%d_radius' % ci][t]
            nch = int(z['t%d_nchoices' % ci][t])
            ch = torch.zeros(1, max(nch, 1), dtype=torch.int16)
            ch[0, :nch] = torch.from_numpy(z['t%d_choices' % ci][used:used + nch].astype(np.int16))
            used += nch
            logits = torch.from_numpy(z['t%d_logits' % ci][t]).unsqueeze(1).to(dev)     # [N,1,5]
            flags = env.move(logits=logits, choices=ch)
            assert flags[0].cpu().tolist() == [int(v) for v in z['t%d_flags' % ci][t]], (ci, t)
            assert env.choice_count[0].item() == nch
            assert (env.reached[0].cpu().numpy() == z['t%d_reached' % ci][t]).all()
        res = env.results()
        assert (res['positions'][0].numpy() == pos_all[m['T']]).all()
        assert res['makespan'][0].item() == m['makespan'] and res['flowtime'][0].item() == m['flowtime']
        assert res['end_step'][0].tolist() == m['end_step']
        assert res['start_step'][0].tolist() == m['start_step']


@pytest.mark.parametrize('B,N,W,dens', [(64, 10, 20, 0.1), (16, 40, 24, 0.05), (8, 100, 40, 0.05),
                                        (4, 100, 100, 0.02)])        # C5's map size: goal offsets up to 99
def test_batched_steps_vs_oracle_random_actions(dev, B, N, W, dens):
    """Many episodes at once, random joint actions (lots of collisions), lowest-index tie-break."""
    from gnn_pathplanning_amd.rollout import BatchedRollout
    rng = np.random.default_rng(B + N)
    grids, starts, goals = random_episodes(rng, B, N, W, dens)
    maxstep = 12
    # mixed per-episode limits (maxstep = rate * makespan[b] in the reference): an episode whose own
    # loop has ended must stay frozen while the rest of the batch goes on
    limits = [maxstep - (b % 4) * 3 for b in range(B)]
    env = BatchedRollout(grids, starts, goals, limits, dev, tie_mode='lowest')
    eps = [ro.EpisodeState(grids[b], goals[b], starts[b], limits[b]) for b in range(B)]
    radius = [6.0] * B
    lowest = lambda c: c[0]                                              # noqa: E731
    for t in range(maxstep + 1):
        obs = env.observe().cpu().numpy()
        S = env.gso(t).cpu().numpy()
        rad = env.radius.cpu().numpy()
        acts = rng.integers(0, 5, size=(B, N))
        flags = env.move(actions=torch.from_numpy(acts).to(dev)).cpu().numpy()
        pos = env.pos.cpu().numpy()
        for b in range(B):
            assert (obs[b] == ro.build_observations(grids[b], goals[b], eps[b].cur)).all(), (t, b)
            Sb, radius[b], _ = ro.communication_gso(eps[b].cur, radius[b], grow=(t == 0))
            assert radius[b] == rad[b] and (S[b] == Sb.astype(np.float32)).all(), (t, b)
            f = ro.loop_step(eps[b], acts[b], t + 1, lowest)
            assert [int(v) for v in f] == flags[b].tolist(), (t, b)
            assert (pos[b] == eps[b].cur).all(), (t, b)
    res = env.results()
    for b in range(B):
        assert res['makespan'][b].item() == eps[b].makespan
        assert res['flowtime'][b].item() == eps[b].flowtime
        assert bool(res['done'][b]) == eps[b].done and eps[b].done
        assert res['reached'][b].tolist() == eps[b].reached
        assert res['end_step'][b].tolist() == eps[b].end_step


def test_run_respects_per_episode_maxstep(dev):
    """BatchedRollout.run() with mixed limits == the reference's case loop run episode by episode
    (agents/decentralplannerlocal.py:560-605): an episode that hits its own maxstep reports failure
    and makespan = maxstep even though the batch keeps stepping."""
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from gnn_pathplanning_amd.rollout import BatchedRollout
    B, N, W = 12, 10, 20
    rng = np.random.default_rng(77)
    grids, starts, goals = random_episodes(rng, B, N, W, 0.08)

    class Cfg:
        num_agents, nGraphFilterTaps, device = N, 3, dev
    net = DecentralPlannerNet(Cfg()).to(dev).eval()
    net.load_state_dict(orc.init_state_dict(3, seed=4))
    limits = [2 + 3 * (b % 5) for b in range(B)]
    out = BatchedRollout(grids, starts, goals, limits, dev, tie_mode='lowest').run(net, check_every=3)
    assert out['done'].all()
    # episode by episode, alone, with the same limit: identical metrics
    for b in range(B):
        solo = BatchedRollout(grids[b:b + 1], starts[b:b + 1], goals[b:b + 1], limits[b], dev,
                              tie_mode='lowest').run(net, check_every=1)
        assert solo['steps'] <= limits[b]
        for k in ('makespan', 'flowtime', 'reached', 'end_step', 'start_step', 'positions'):
            assert torch.equal(solo[k][0], out[k][b]), (b, k)
        if not bool(out['success'][b]):
            assert out['makespan'][b].item() <= limits[b] and out['end_step'][b].max().item() == limits[b]


@pytest.mark.parametrize('B,N,W,dens', [(64, 10, 20, 0.1), (16, 40, 24, 0.05), (8, 100, 40, 0.05)])
def test_fused_step_equals_separate_kernels(dev, B, N, W, dens):
    """move_and_observe (gnnpp_rollout_step: move -> gso -> observe in one launch) against the three
    separate launches on a twin environment: every state tensor identical after every step."""
    from gnn_pathplanning_amd.rollout import BatchedRollout
    rng = np.random.default_rng(7 * B + N)
    grids, starts, goals = random_episodes(rng, B, N, W, dens)
    maxstep = 10
    a = BatchedRollout(grids, starts, goals, maxstep, dev, tie_mode='hashed', seed=3)
    b = BatchedRollout(grids, starts, goals, maxstep, dev, tie_mode='hashed', seed=3)
    a.observe(); a.gso(0)
    b.observe(); b.gso(0)
    for t in range(maxstep + 2):
        acts = torch.from_numpy(rng.integers(0, 5, size=(B, N))).to(dev)
        fa = a.move_and_observe(actions=acts).clone()
        fb = b.move(actions=acts).clone()
        b.observe(); b.gso()
        for name in ('pos', 'obs', 'S', 'radius', 'reached', 'start_step', 'end_step', 'stats', 'choice_count'):
            assert torch.equal(getattr(a, name), getattr(b, name)), (t, name)
        assert torch.equal(fa, fb), t


@pytest.mark.parametrize('B,N,W', [(64, 10, 20), (7, 16, 24), (3, 1, 8)])
def test_one_launch_rollout_step_equals_separate_launches(dev, B, N, W):
    """BatchedRollout.step() for small teams = gnnpp_rollout_policy_step (policy + move + gso + observe in
    one kernel).  A twin environment is stepped with forward_logits + move + gso + observe; logits and
    every state tensor must be identical after every step."""
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from gnn_pathplanning_amd.rollout import BatchedRollout
    rng = np.random.default_rng(11 * B + N)
    grids, starts, goals = random_episodes(rng, B, N, W, 0.08)

    class Cfg:
        num_agents, nGraphFilterTaps, device = N, 3, dev
    net = DecentralPlannerNet(Cfg()).to(dev).eval()
    net.load_state_dict(orc.init_state_dict(3, seed=9))
    a = BatchedRollout(grids, starts, goals, 9, dev, tie_mode='hashed', seed=5)
    b = BatchedRollout(grids, starts, goals, 9, dev, tie_mode='hashed', seed=5)
    for t in range(11):
        a.step(net)
        assert a._logits is not None and a._state_step == a.t     # the one-launch path was taken
        if t == 0:
            b.observe(); b.gso(0)
        net.addGSO(b.S)
        lg = net.forward_logits(b.obs)
        b.move(logits=lg)
        b.observe(); b.gso()
        assert torch.equal(a._logits, lg), (t, (a._logits - lg).abs().max().item())
        for name in ('pos', 'obs', 'S', 'radius', 'reached', 'start_step', 'end_step', 'stats', 'flags',
                     'choice_count'):
            assert torch.equal(getattr(a, name), getattr(b, name)), (t, name)


def test_steps_burst_equals_single_steps(dev):
    """BatchedRollout.steps(model, n) (gnnpp_rollout_policy_steps: n launches enqueued by one C call) leaves
    exactly the state that n single step() calls leave, for mixed per-episode step limits and both tie rules
    that carry state on the device."""
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from gnn_pathplanning_amd.rollout import BatchedRollout
    B, N, W = 24, 10, 20
    rng = np.random.default_rng(77)
    grids, starts, goals = random_episodes(rng, B, N, W, 0.08)

    class Cfg:
        num_agents, nGraphFilterTaps, device = N, 3, dev
    net = DecentralPlannerNet(Cfg()).to(dev).eval()
    net.load_state_dict(orc.init_state_dict(3, seed=9))
    maxstep = rng.integers(3, 20, size=B)
    for tie in ('hashed', 'mt19937'):
        a = BatchedRollout(grids, starts, goals, maxstep, dev, tie_mode=tie, seed=5)
        b = BatchedRollout(grids, starts, goals, maxstep, dev, tie_mode=tie, seed=5)
        for n in (1, 5, 8, 3):
            a.steps(net, n)
            for _ in range(n):
                b.step(net)
            assert a.t == b.t
            for name in ('pos', 'obs', 'S', 'radius', 'reached', 'start_step', 'end_step', 'stats', 'flags',
                         'done'):
                assert torch.equal(getattr(a, name), getattr(b, name)), (tie, n, name)
            assert torch.equal(a._logits, b._logits)


@pytest.mark.parametrize('N,groups', [(10, 2), (10, 3), (24, 2)])
def test_grouped_rollout_equals_one_batch(dev, N, groups):
    """GroupedRollout (episode slices on their own HIP streams, overlapping one another) ends every episode in
    exactly the state the single-stream BatchedRollout gives it.  N = 24 takes the two-kernel policy, whose
    feature workspace is per stream."""
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from gnn_pathplanning_amd.rollout import BatchedRollout, GroupedRollout
    B, W = 50, 20
    rng = np.random.default_rng(N + groups)
    grids, starts, goals = random_episodes(rng, B, N, W, 0.08)

    class Cfg:
        num_agents, nGraphFilterTaps, device = N, 3, dev
    net = DecentralPlannerNet(Cfg()).to(dev).eval()
    net.load_state_dict(orc.init_state_dict(3, seed=13))
    maxstep = rng.integers(10, 40, size=B)
    for tie in ('lowest', 'mt19937'):
        one = BatchedRollout(grids, starts, goals, maxstep, dev, tie_mode=tie, seed=3).run(net)
        env = GroupedRollout(grids, starts, goals, maxstep, dev, groups=groups, tie_mode=tie, seed=3)
        many = env.run(net)
        assert [hi - lo for lo, hi in env.slices] and sum(hi - lo for lo, hi in env.slices) == B
        assert set(one) == set(many)
        for k in one:
            if k == 'steps':
                assert one[k] == many[k]
            else:
                assert torch.equal(one[k], many[k]), (tie, k)


def test_closed_loop_rollout_with_policy(dev):
    """observe -> gso -> forward -> move on the GPU; every stage checked against the CPU oracles
    fed with the GPU's own state, so a near-tie in the logits cannot make the trajectories drift."""
    from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
    from gnn_pathplanning_amd.rollout import BatchedRollout
    B, N, W = 6, 10, 20
    rng = np.random.default_rng(5)
    grids, starts, goals = random_episodes(rng, B, N, W, 0.1)

    class Cfg:
        num_agents, nGraphFilterTaps, device = N, 3, dev
    sd = orc.init_state_dict(3, seed=21)
    net = DecentralPlannerNet(Cfg()).to(dev).eval()
    net.load_state_dict(sd)
    env = BatchedRollout(grids, starts, goals, 10, dev, tie_mode='lowest')
    eps = [ro.EpisodeState(grids[b], goals[b], starts[b], 10) for b in range(B)]
    for t in range(10):
        obs = env.observe()
        S = env.gso()
        net.addGSO(S)
        logits = net.forward_logits(obs)                                  # [N,B,5]
        with torch.no_grad():
            want = torch.stack(orc.policy_forward(sd, S.cpu(), obs.cpu()), 0)   # [N,B,5]
        assert (logits.cpu() - want).abs().max().item() <= 1e-4
        acts = net.decode_actions(logits).cpu().numpy()                   # [B,N]
        margin = torch.topk(want, 2, dim=-1).values
        clear = ((margin[..., 0] - margin[..., 1]) > 1e-5).numpy().T      # [B,N]
        assert (acts[clear] == want.argmax(-1).numpy().T[clear]).all()
        env.move(logits=logits)
        pos = env.pos.cpu().numpy()
        for b in range(B):
            ro.loop_step(eps[b], acts[b], t + 1, lambda c: c[0])
            assert (pos[b] == eps[b].cur).all(), (t, b)
    out = BatchedRollout(grids, starts, goals, 24, dev).run(net, check_every=4)
    assert out['steps'] <= 24 and out['reached'].shape == (B, N)


def test_rollout_needs_gpu():
    from gnn_pathplanning_amd import _native
    from gnn_pathplanning_amd.rollout import BatchedRollout
    with pytest.raises(_native.GnnppError):
        BatchedRollout(np.zeros((4, 4)), np.zeros((1, 2, 2)), np.ones((1, 2, 2)), 4, 'cpu')


def test_rollout_from_reference_case_files(dev, rollout_golden):
    """A dataset case in the reference's .mat format feeds the batched rollout directly."""
    import os
    from conftest import GOLDEN
    from gnn_pathplanning_amd.formats import rollout_from_cases
    z, meta = rollout_golden
    mat = os.path.join(GOLDEN, 'case_fixture.mat')                   # = trace 2 (6 agents, 8x8)
    env = rollout_from_cases([mat, mat], dev, tie_mode='lowest')
    assert (env.B, env.N, env.H, env.W) == (2, 6, 8, 8)
    assert env.maxstep.tolist() == [2 * meta[2]['T']] * 2
    obs = env.observe().cpu().numpy()
    assert (obs[0] == z['t2_obs'][0]).all() and (obs[1] == z['t2_obs'][0]).all()
    assert (env.gso(0)[1].cpu().numpy() == z['t2_gso'][0].astype(np.float32)).all()


def test_observe_kernel_projected_goals_full_grid(dev):
    """Every goal offset with |dx|, |dy| <= 120 around a common goal on a 241x241 map through the REAL
    observation kernel: channel 1 is the oracle's one-hot cell (atan2 / np.round rule of
    dataloader/statetransformer.py:47-66 vs the kernel's integer rule)."""
    from gnn_pathplanning_amd.rollout import BatchedRollout
    W, c, N = 241, 120, 128
    cells = [(x, y) for x in range(W) for y in range(W) if (x, y) != (c, c)]
    B = (len(cells) + N - 1) // N
    cells += cells[:B * N - len(cells)]
    pos = np.array(cells, np.int64).reshape(B, N, 2)
    goal = np.full((B, N, 2), c, np.int64)
    env = BatchedRollout(np.zeros((W, W), np.uint8), pos, goal, 4, dev)
    ch1 = env.observe()[:, :, 1].cpu().numpy()
    assert (ch1.sum(axis=(2, 3)) == 1).all()
    want = np.zeros((B, N, 2), np.int64)
    for b in range(B):
        for n in range(N):
            dx, dy = c - pos[b, n, 0], c - pos[b, n, 1]
            want[b, n] = (dx + 5, dy + 5) if (abs(dx) <= 4 and abs(dy) <= 4) else ro.projected_goal(dx, dy)
    bi, ni = np.meshgrid(np.arange(B), np.arange(N), indexing='ij')
    assert (ch1[bi, ni, want[..., 0], want[..., 1]] == 1.0).all()


def test_mt19937_tie_break_matches_python_random_choice(dev):
    """tie_mode='mt19937': episode b resolves collisions exactly like the reference's random.choice
    (utils/multirobotsim_dcenlocal.py:489) after random.seed(seed + b)."""
    import random
    from gnn_pathplanning_amd.rollout import BatchedRollout
    rng = np.random.default_rng(31)
    B, N, W = 48, 12, 5
    grids, starts, goals = random_episodes(rng, B, N, W, 0.0)
    env = BatchedRollout(grids, starts, goals, 50, dev, tie_mode='mt19937', seed=500)
    eps = [ro.EpisodeState(grids[b], goals[b], starts[b], 50) for b in range(B)]
    gens = [random.Random(500 + b) for b in range(B)]
    draws = 0
    for t in range(6):
        acts = rng.integers(0, 5, size=(B, N))
        flags = env.move(actions=torch.from_numpy(acts).to(dev)).cpu().numpy()
        pos = env.pos.cpu().numpy()
        for b in range(B):
            f = ro.move_step(eps[b], acts[b], t + 1, gens[b].choice)
            assert [int(v) for v in f] == flags[b].tolist(), (t, b)
            assert (pos[b] == eps[b].cur).all(), (t, b)
        draws += int(env.choice_count.sum().item())
    assert draws > 100


def test_mt19937_word_stream_overflow_is_an_error(dev):
    """an episode that consumes more Mersenne-Twister words than it was given (the kernel substitutes 0
    from there on) no longer follows the reference's random.choice stream; results() / check_rng() must raise instead
    of reporting a silently different rollout."""
    from gnn_pathplanning_amd import _native
    from gnn_pathplanning_amd.rollout import BatchedRollout
    rng = np.random.default_rng(32)
    B, N, W = 16, 12, 5
    grids, starts, goals = random_episodes(rng, B, N, W, 0.0)
    env = BatchedRollout(grids, starts, goals, 50, dev, tie_mode='mt19937', seed=7, rng_words=4)
    for t in range(8):                                            # a crowded 5 x 5 map: dozens of tie-breaks
        env.move(actions=torch.from_numpy(rng.integers(0, 5, size=(B, N))).to(dev))
    assert int(env.choice_count.sum().item()) >= 0
    assert int((env.rng_cursor > 4).sum().item()) > 0
    with pytest.raises(_native.GnnppError, match='random words'):
        env.results()
    ok = BatchedRollout(grids, starts, goals, 50, dev, tie_mode='mt19937', seed=7)
    for t in range(8):
        ok.move(actions=torch.from_numpy(rng.integers(0, 5, size=(B, N))).to(dev))
    ok.results()                                                  # the default 2048 words: fine
