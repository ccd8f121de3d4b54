"""CPU: the HIP rollout kernels (observe / gso / move), compiled unmodified for the host emulation,
replay the traces recorded from the real simulator bit-exactly (tie-breaks replayed)."""
import ctypes
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.skipif(not os.path.exists('/opt/rocm/lib/llvm/bin/clang++'),
                                reason='host clang++ from ROCm not present')


def replay_case(lib, Struct, z, ci, m, fused=False):
    N, W, T = m['N'], m['W'], m['T']
    grid = np.ascontiguousarray(z['t%d_grid' % ci].astype(np.uint8))
    goal = np.ascontiguousarray(z['t%d_goal' % ci].astype(np.int32))[None]
    pos_all = z['t%d_pos' % ci].astype(np.int32)
    pos = np.ascontiguousarray(pos_all[0][None])
    obs = np.zeros((1, N, 3, 11, 11), np.float32)
    radius = np.array([float(m['commR'])], np.float64)
    S = np.zeros((1, N, N), np.float32)
    conn = np.zeros(1, np.int32)
    reached = np.zeros((1, N), np.int32)
    start = np.full((1, N), -1, np.int32)
    end = np.full((1, N), -1, np.int32)
    maxstep = np.array([m['maxstep']], np.int32)
    flags = np.zeros((1, 3), np.int32)
    stats = np.zeros((1, 2), np.int32)
    ccount = np.zeros(1, np.int32)
    all_choices = z['t%d_choices' % ci]
    used = 0
    r = Struct()
    r.grid, r.grid_batched, r.goal, r.pos = grid.ctypes.data, 0, goal.ctypes.data, pos.ctypes.data
    r.B, r.N, r.H, r.W = 1, N, W, W
    r.obs, r.radius, r.S, r.connected = obs.ctypes.data, radius.ctypes.data, S.ctypes.data, conn.ctypes.data
    r.reached, r.start_step, r.end_step = reached.ctypes.data, start.ctypes.data, end.ctypes.data
    r.maxstep, r.flags, r.stats = maxstep.ctypes.data, flags.ctypes.data, stats.ctypes.data
    r.tie_mode, r.choice_count = 2, ccount.ctypes.data
    for t in range(T):
        assert (pos[0] == pos_all[t]).all(), (ci, t)
        pair = fused == 'pair' and t > 0       # gnnpp_rollout_gso_observe: graph + observations in one launch
        if pair:
            r.grow = 0
            assert lib.gnnpp_rollout_gso_observe(ctypes.byref(r), None) == 0
        if (not fused or t == 0) or (fused == 'pair' and not pair):   # fused step: already produced obs and S
            assert lib.gnnpp_rollout_observe(ctypes.byref(r), None) == 0
        assert (obs[0] == z['t%d_obs' % ci][t].astype(np.float32)).all(), (ci, t)
        r.grow = int(t == 0)
        if (not fused or t == 0) or (fused == 'pair' and not pair):
            assert lib.gnnpp_rollout_gso(ctypes.byref(r), None) == 0
        assert radius[0] == z['t%d_radius' % ci][t], (ci, t)
        assert (S[0] == z['t%d_gso' % ci][t].astype(np.float32)).all(), (ci, t)
        assert conn[0] in (0, 1)
        # move: logits in the [N,B,5] layout of gnnpp_policy_fwd; recorded tie-breaks for this step
        logits = np.ascontiguousarray(z['t%d_logits' % ci][t][:, None, :])
        nch = int(z['t%d_nchoices' % ci][t])
        ch = np.zeros((1, max(nch, 1)), np.int16)
        ch[0, :nch] = all_choices[used:used + nch]
        used += nch
        r.logits, r.actions, r.currentstep = logits.ctypes.data, None, t + 1
        r.choices, r.max_choices = ch.ctypes.data, ch.shape[1]
        r.grow = 0
        assert (lib.gnnpp_rollout_step if fused is True else lib.gnnpp_rollout_move)(ctypes.byref(r), None) == 0
        assert ccount[0] == nch, (ci, t, ccount[0], nch)
        assert list(flags[0]) == [int(v) for v in z['t%d_flags' % ci][t]], (ci, t)
        assert (reached[0] == z['t%d_reached' % ci][t]).all(), (ci, t)
    assert (pos[0] == pos_all[T]).all()
    assert list(stats[0]) == [m['makespan'], m['flowtime']], (ci, list(stats[0]), m)
    assert list(end[0]) == m['end_step'] and list(start[0]) == m['start_step']


def test_emu_rollout_traces(rollout_golden):
    import emu_lib
    from gnn_pathplanning_amd._native import RolloutStruct
    lib = emu_lib.load()
    z, meta = rollout_golden
    for ci, m in enumerate(meta):
        replay_case(lib, RolloutStruct, z, ci, m)


def test_emu_rollout_traces_fused_step(rollout_golden):
    """gnnpp_rollout_step (move -> gso -> observe in one launch) replays the same traces."""
    import emu_lib
    from gnn_pathplanning_amd._native import RolloutStruct
    lib = emu_lib.load()
    z, meta = rollout_golden
    for ci, m in enumerate(meta):
        replay_case(lib, RolloutStruct, z, ci, m, fused=True)


def test_emu_rollout_traces_gso_observe_pair(rollout_golden):
    """gnnpp_rollout_gso_observe (the graph's and the observations' workgroups in one launch, what large teams
    run between two moves) replays the same traces."""
    import emu_lib
    from gnn_pathplanning_amd._native import RolloutStruct
    lib = emu_lib.load()
    z, meta = rollout_golden
    for ci, m in enumerate(meta):
        replay_case(lib, RolloutStruct, z, ci, m, fused='pair')


def test_emu_rollout_traces_large_teams(rollout_large_golden):
    """The reference simulator's traces for 50 agents / 50 x 50 and 100 agents / 100 x 100 maps through the HIP
    kernels (separate launches and the graph + observations pair large teams run): two-word agent masks, the
    cell-count map on big grids, goal offsets up to 99."""
    import emu_lib
    from gnn_pathplanning_amd._native import RolloutStruct
    lib = emu_lib.load()
    z, meta = rollout_large_golden
    for ci, m in enumerate(meta):
        replay_case(lib, RolloutStruct, z, ci, m)
        replay_case(lib, RolloutStruct, z, ci, m, fused='pair')


def test_emu_rollout_argument_checks():
    import emu_lib
    from gnn_pathplanning_amd._native import RolloutStruct
    lib = emu_lib.load()
    r = RolloutStruct()
    assert lib.gnnpp_rollout_observe(ctypes.byref(r), None) == -1
    assert lib.gnnpp_rollout_gso(None, None) == -1
    r.B, r.N = 1, 200
    assert lib.gnnpp_rollout_move(ctypes.byref(r), None) == -1


def test_emu_mixed_maxstep_freezes_finished_episodes():
    """Batch with different per-episode limits through the emulated move kernel vs the oracle's case
    loop (oracle.rollout_oracle.loop_step): an episode past its own maxstep, or whose loop broke
    after allReachGoal, is left untouched (success/makespan were inflated)."""
    import emu_lib
    from gnn_pathplanning_amd._native import RolloutStruct
    from oracle import rollout_oracle as ro
    lib = emu_lib.load()
    rng = np.random.default_rng(3)
    B, N, W = 6, 5, 7
    grids = (rng.random((B, W, W)) < 0.05).astype(np.uint8)
    starts = np.zeros((B, N, 2), np.int32); goals = np.zeros((B, N, 2), np.int32)
    for b in range(B):
        free = np.argwhere(grids[b] == 0)
        pick = rng.choice(len(free), 2 * N, replace=False)
        starts[b], goals[b] = free[pick[:N]], free[pick[N:]]
    grids[0] = 0                              # episode 0: every agent one step (action 3) from its goal
    starts[0] = [[i, 0] for i in range(N)]
    goals[0] = [[i, 1] for i in range(N)]
    limits = np.array([6, 2, 3, 8, 1, 5], np.int32)
    pos = np.ascontiguousarray(starts.copy())
    reached = np.zeros((B, N), np.int32)
    start = np.full((B, N), -1, np.int32); end = np.full((B, N), -1, np.int32)
    flags = np.zeros((B, 3), np.int32); stats = np.zeros((B, 2), np.int32)
    done = np.zeros(B, np.int32)
    r = RolloutStruct()
    r.grid, r.grid_batched, r.goal, r.pos = grids.ctypes.data, 1, goals.ctypes.data, pos.ctypes.data
    r.B, r.N, r.H, r.W = B, N, W, W
    r.reached, r.start_step, r.end_step = reached.ctypes.data, start.ctypes.data, end.ctypes.data
    r.maxstep, r.flags, r.stats, r.done = limits.ctypes.data, flags.ctypes.data, stats.ctypes.data, done.ctypes.data
    r.tie_mode = 0
    eps = [ro.EpisodeState(grids[b], goals[b], starts[b], limits[b]) for b in range(B)]
    for t in range(10):
        acts = np.ascontiguousarray(rng.integers(0, 5, size=(B, N)).astype(np.int32))
        if t == 0:
            acts[0] = 3                        # all arrive at call 1; call 2 sees allReachGoal and ends the loop
        r.logits, r.actions, r.currentstep = None, acts.ctypes.data, t + 1
        assert lib.gnnpp_rollout_move(ctypes.byref(r), None) == 0
        for b in range(B):
            f = ro.loop_step(eps[b], acts[b], t + 1, lambda c: c[0])
            assert [int(v) for v in f] == list(flags[b]), (t, b)
            assert (pos[b] == eps[b].cur).all(), (t, b)
            assert bool(done[b]) == eps[b].done, (t, b)
    for b in range(B):
        assert eps[b].done and done[b] == 1
        assert list(stats[b]) == [eps[b].makespan, eps[b].flowtime], b
        assert list(reached[b]) == [int(v) for v in eps[b].reached]
        assert list(end[b]) == eps[b].end_step
    assert list(stats[0]) == [1, N] and limits[0] == 6   # ended by allReachGoal, long before its maxstep


def test_emu_move_dense_conflicts_vs_oracle():
    """Crowded little maps, random joint actions: lots of vertex conflicts, chains of fall-backs and
    swaps per step.  The kernel's candidate-driven collision passes must reproduce the oracle's
    agent-by-agent loops exactly (positions, flags, tie-break counts), lowest-index tie-break."""
    import emu_lib
    from gnn_pathplanning_amd._native import RolloutStruct
    from oracle import rollout_oracle as ro
    lib = emu_lib.load()
    rng = np.random.default_rng(21)
    # (the last case: a map of more than 32 768 cells has no LDS cell-count map -- the collision candidates come
    # from the all-pairs scan; its agents start in one 5 x 5 corner so that they still collide)
    for (B, N, W) in ((24, 9, 4), (12, 14, 5), (6, 70, 10), (3, 14, 182)):
        grids = (rng.random((B, W, W)) < 0.04).astype(np.uint8)
        starts = np.zeros((B, N, 2), np.int32); goals = np.zeros((B, N, 2), np.int32)
        for b in range(B):
            if W > 100:
                grids[b, :5, :5] = 0
            free = np.argwhere(grids[b] == 0)
            if W > 100:
                free = free[(free[:, 0] < 5) & (free[:, 1] < 5)]
            while len(free) < N:
                grids[b] = 0
                free = np.argwhere(grids[b] == 0)
            starts[b] = free[rng.choice(len(free), N, replace=False)]
            goals[b] = free[rng.choice(len(free), N, replace=False)]
        pos = np.ascontiguousarray(starts.copy())
        reached = np.zeros((B, N), np.int32)
        start = np.full((B, N), -1, np.int32); end = np.full((B, N), -1, np.int32)
        flags = np.zeros((B, 3), np.int32); stats = np.zeros((B, 2), np.int32)
        ccount = np.zeros(B, np.int32)
        limits = np.full(B, 50, np.int32)
        r = RolloutStruct()
        r.grid, r.grid_batched, r.goal, r.pos = grids.ctypes.data, 1, goals.ctypes.data, pos.ctypes.data
        r.B, r.N, r.H, r.W = B, N, W, W
        r.reached, r.start_step, r.end_step = reached.ctypes.data, start.ctypes.data, end.ctypes.data
        r.maxstep, r.flags, r.stats = limits.ctypes.data, flags.ctypes.data, stats.ctypes.data
        r.tie_mode, r.choice_count = 0, ccount.ctypes.data
        eps = [ro.EpisodeState(grids[b], goals[b], starts[b], 50) for b in range(B)]
        collisions = 0
        for t in range(6):
            acts = np.ascontiguousarray(rng.integers(0, 5, size=(B, N)).astype(np.int32))
            r.logits, r.actions, r.currentstep = None, acts.ctypes.data, t + 1
            assert lib.gnnpp_rollout_move(ctypes.byref(r), None) == 0
            for b in range(B):
                calls = [0]

                def lowest(c, calls=calls):
                    calls[0] += 1
                    return c[0]
                f = ro.move_step(eps[b], acts[b], t + 1, lowest)
                assert [int(v) for v in f] == list(flags[b]), (N, t, b)
                assert (pos[b] == eps[b].cur).all(), (N, t, b)
                assert ccount[b] == calls[0], (N, t, b)
                collisions += calls[0]
        assert collisions > 20 * B // 6


def test_emu_mt19937_tie_break_is_random_choice():
    """tie_mode = GNNPP_TIE_MT19937: the kernel applies CPython's random.choice (bit_length / getrandbits /
    rejection) to a Mersenne-Twister word stream -- episode b moves exactly as the oracle does with
    random.Random(seed_b).choice as the tie-break, i.e. as the reference after random.seed(seed_b)."""
    import random
    import emu_lib
    from gnn_pathplanning_amd._native import RolloutStruct
    from oracle import rollout_oracle as ro
    lib = emu_lib.load()
    rng = np.random.default_rng(8)
    B, N, W, NW = 10, 12, 4, 256
    grids = np.zeros((B, W, W), np.uint8)
    starts = np.zeros((B, N, 2), np.int32); goals = np.zeros((B, N, 2), np.int32)
    for b in range(B):
        free = np.argwhere(grids[b] == 0)
        starts[b] = free[rng.choice(len(free), N, replace=False)]
        goals[b] = free[rng.choice(len(free), N, replace=False)]
    words = np.array([[g.getrandbits(32) for _ in range(NW)] for g in (random.Random(100 + b) for b in range(B))],
                     dtype=np.uint32)
    cursor = np.zeros(B, np.int32)
    pos = np.ascontiguousarray(starts.copy())
    reached = np.zeros((B, N), np.int32)
    start = np.full((B, N), -1, np.int32); end = np.full((B, N), -1, np.int32)
    flags = np.zeros((B, 3), np.int32); stats = np.zeros((B, 2), np.int32)
    ccount = np.zeros(B, np.int32); limits = np.full(B, 99, np.int32)
    r = RolloutStruct()
    r.grid, r.grid_batched, r.goal, r.pos = grids.ctypes.data, 1, goals.ctypes.data, pos.ctypes.data
    r.B, r.N, r.H, r.W = B, N, W, W
    r.reached, r.start_step, r.end_step = reached.ctypes.data, start.ctypes.data, end.ctypes.data
    r.maxstep, r.flags, r.stats = limits.ctypes.data, flags.ctypes.data, stats.ctypes.data
    r.tie_mode, r.choice_count = 3, ccount.ctypes.data
    r.rng_words, r.rng_cursor, r.rng_max = words.ctypes.data, cursor.ctypes.data, NW
    eps = [ro.EpisodeState(grids[b], goals[b], starts[b], 99) for b in range(B)]
    gens = [random.Random(100 + b) for b in range(B)]
    draws = 0
    for t in range(5):
        acts = np.ascontiguousarray(rng.integers(0, 5, size=(B, N)).astype(np.int32))
        r.logits, r.actions, r.currentstep = None, acts.ctypes.data, t + 1
        assert lib.gnnpp_rollout_move(ctypes.byref(r), None) == 0
        for b in range(B):
            f = ro.move_step(eps[b], acts[b], t + 1, gens[b].choice)
            assert [int(v) for v in f] == list(flags[b]), (t, b)
            assert (pos[b] == eps[b].cur).all(), (t, b)
        draws += int(ccount.sum())
    assert draws > 30 and (cursor >= 0).all() and cursor.sum() >= draws      # rejections consume extra words
