"""CPU: pin oracle/rollout_oracle.py (observation builder, communication GSO, move + collision
shielding) to traces recorded from the REAL reference simulator (oracle/gen_golden_rollout.py).
Everything here is integer / boolean / fp64 work: the bar is bit-exact."""
import numpy as np

from oracle import rollout_oracle as ro


class Replay:
    """random.choice replaced by the recorded outcomes (index into the collided list)."""

    def __init__(self, seq):
        self.seq, self.i = list(seq), 0

    def __call__(self, collided):
        k = self.seq[self.i]
        self.i += 1
        return collided[k]


def test_rollout_traces_replay_bit_exact(rollout_golden):
    z, meta = rollout_golden
    assert sum(m['collisions'] for m in meta) >= 30          # the traces do exercise the shielding
    _replay(z, meta)


def test_rollout_traces_large_teams_replay_bit_exact(rollout_large_golden):
    """50 agents on a 50 x 50 map, 100 agents on a 100 x 100 map (the rollouts of BASELINE configs 3 and 5)."""
    z, meta = rollout_large_golden
    assert [m['N'] for m in meta] == [50, 100] and sum(m['collisions'] for m in meta) >= 20
    _replay(z, meta)


def _replay(z, meta):
    for ci, m in enumerate(meta):
        grid, goal = z['t%d_grid' % ci], z['t%d_goal' % ci]
        pos, T = z['t%d_pos' % ci], m['T']
        ep = ro.EpisodeState(grid, goal, pos[0], m['maxstep'])
        chooser = Replay(z['t%d_choices' % ci])
        radius = float(m['commR'])
        for t in range(T):
            assert (ep.cur == pos[t]).all(), (ci, t)
            obs = ro.build_observations(grid, goal, ep.cur)
            assert (obs == z['t%d_obs' % ci][t].astype(np.float32)).all(), (ci, t)
            S, radius, _ = ro.communication_gso(ep.cur, radius, grow=(t == 0))
            assert radius == z['t%d_radius' % ci][t], (ci, t)
            assert (S == z['t%d_gso' % ci][t]).all(), (ci, t)
            acts = np.argmax(z['t%d_logits' % ci][t], axis=-1)
            assert (acts == z['t%d_actions' % ci][t]).all()
            used = chooser.i
            flags = ro.move_step(ep, acts, t + 1, chooser)
            assert chooser.i - used == z['t%d_nchoices' % ci][t], (ci, t)
            assert [int(f) for f in flags] == list(z['t%d_flags' % ci][t]), (ci, t)
            assert (np.array(ep.reached, dtype=np.uint8) == z['t%d_reached' % ci][t]).all(), (ci, t)
        assert (ep.cur == pos[T]).all()
        assert ep.makespan == m['makespan'] and ep.flowtime == m['flowtime'], (ci, m)
        assert [(-1 if e is None else e) for e in ep.end_step] == m['end_step']
        assert [(-1 if s is None else s) for s in ep.start_step] == m['start_step']


def test_projected_goal_matches_reference_rule_on_a_grid():
    """Spot-check the angle rule against a direct evaluation (ties at 45 degrees included)."""
    for dx in range(-30, 31):
        for dy in range(-30, 31):
            if abs(dx) <= 4 and abs(dy) <= 4:
                continue
            px, py = ro.projected_goal(dx, dy)
            assert 0 <= px <= 10 and 0 <= py <= 10
            assert px in (0, 10) or py in (0, 10)               # always on the border ring


def _kernel_rule(dx, dy):
    """The integer form rollout_kernels.hip::projected_goal uses instead of atan2 + np.round:
    vertical branch iff |dy| >= |dx| and dy != 0; round-half-even of 5 * d / |D|."""
    def rhe_div(num, den):
        q, rem = divmod(num, den)                       # floor division, 0 <= rem < den
        if 2 * rem > den or (2 * rem == den and (q & 1)):
            q += 1
        return q
    adx, ady = abs(dx), abs(dy)
    if ady >= adx and dy != 0:
        return 5 + rhe_div(5 * dx, ady), 10 if dy > 0 else 0
    return (10 if dx > 0 else (0 if dx < 0 else 5)), 5 + rhe_div(5 * dy, adx)


def test_integer_projected_goal_rule_exhaustive():
    """EXHAUSTIVE for |dx|, |dy| <= 150 (the largest map of BASELINE.json is 100x100, offsets <= 99):
    the kernel's integer rule equals the reference's atan2 / np.round rule (statetransformer.py:47-66)
    on every offset outside the field of view -- the claim rollout_kernels.hip makes."""
    bad = []
    for dx in range(-150, 151):
        for dy in range(-150, 151):
            if abs(dx) <= 4 and abs(dy) <= 4:
                continue
            if _kernel_rule(dx, dy) != ro.projected_goal(dx, dy):
                bad.append((dx, dy))
    assert not bad, bad[:10]


def test_emulated_observe_kernel_projected_goals():
    """The HIP observation kernel itself (host emulation) on every goal offset with |d| <= 40 around a
    common goal: channel 1 must be the oracle's one-hot cell."""
    import ctypes
    import os
    import sys
    import pytest
    if not os.path.exists('/opt/rocm/lib/llvm/bin/clang++'):
        pytest.skip('host clang++ from ROCm not present')
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu'))
    import emu_lib
    from gnn_pathplanning_amd._native import RolloutStruct
    lib = emu_lib.load()
    W = 81
    cells = [(x, y) for x in range(W) for y in range(W) if (x, y) != (40, 40)]
    N = 128
    B = (len(cells) + N - 1) // N
    cells += cells[:B * N - len(cells)]                     # pad the last episode with repeats of others
    pos = np.array(cells, np.int32).reshape(B, N, 2)
    # (agents of one episode must be distinct cells: consecutive cells are)
    goal = np.full((B, N, 2), 40, np.int32)
    grid = np.zeros((W, W), np.uint8)
    obs = np.zeros((B, N, 3, 11, 11), np.float32)
    r = RolloutStruct()
    r.grid, r.grid_batched, r.goal, r.pos = grid.ctypes.data, 0, goal.ctypes.data, pos.ctypes.data
    r.B, r.N, r.H, r.W, r.obs = B, N, W, W, obs.ctypes.data
    assert lib.gnnpp_rollout_observe(ctypes.byref(r), None) == 0
    for b in range(0, B, 7):                                # the oracle's python loops are slow: sample
        want = ro.build_observations(grid, goal[b], pos[b])
        assert (obs[b] == want).all(), b
    ch1 = obs[:, :, 1]
    assert (ch1.sum(axis=(2, 3)) == 1).all()                # exactly one goal cell per agent
    for b in range(B):
        for n in range(N):
            dx, dy = 40 - pos[b, n, 0], 40 - pos[b, n, 1]
            px, py = (dx + 5, dy + 5) if (abs(dx) <= 4 and abs(dy) <= 4) else ro.projected_goal(dx, dy)
            assert ch1[b, n, px, py] == 1.0, (b, n, dx, dy)
