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
