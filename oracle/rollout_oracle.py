"""CPU oracle for one rollout step around the policy forward.  TEST INFRASTRUCTURE ONLY.

Plain numpy / python restatement of the three host-side pieces the reference runs every simulated
timestep; checker for the HIP rollout kernels, never imported by the
product.  Parity status: PINNED -- tests/golden/rollout_traces.npz is recorded from the REAL
simulator (oracle/gen_golden_rollout.py) and tests/test_rollout_oracle.py replays every step.

  build_observations()  <- AgentState.toInputTensor   dataloader/statetransformer.py:82-130
                           (projected goal :47-66, map padding :32-34, agent channel :36-45)
  communication_gso()   <- multiRobotSim.computeAdjacencyMatrix  utils/multirobotsim_dcenlocal.py:320-365
                           (+ initCommunicationRadius :240-241, getGSO :367-394)
  move_step()           <- multiRobotSim.move :562-723 and interRobotCollision :462-555

Positions are integer (row, col) pairs; the reference keeps them in float tensors.  The only
non-deterministic ingredient of the reference, random.choice among colliding agents (:489), is an
injected callable so recorded choices can be replayed.
"""
import math

import numpy as np

FOV = 9
FOV_W = 4           # int(FOV / 2)
OBS_HW = 11         # FOV + 2 * border
DIST = 5            # floor(W / 2) = centre index
STOP = 4
DELTA = ((-1, 0), (0, -1), (1, 0), (0, 1), (0, 0))     # multirobotsim_dcenlocal.py:22-27


def projected_goal(dx, dy):
    """Border cell that stands for a goal outside the field of view (statetransformer.py:47-66).
    dx, dy = goal - state along the first / second coordinate.  Returns (row, col) in the 11x11
    channel.  np.round rounds half to even, python's round() does the same."""
    angle = math.atan2(float(dy), float(dx))
    if (angle >= math.pi / 4 and angle <= math.pi * 3 / 4) or \
            (angle >= -math.pi * (3 / 4) and angle <= -math.pi / 4):
        gy = int(DIST * (np.sign(dy) + 1))
        gx = int(DIST + np.round(DIST * dx / abs(dy)))
    else:
        gx = int(DIST * (np.sign(dx) + 1))
        gy = int(DIST + np.round(DIST * dy / abs(dx)))
    return gx, gy


def build_observations(grid, goals, states):
    """grid [H,W] {0,1}; goals, states [N,2] ints -> [N,3,11,11] float32.
    channel 0 obstacles (outside the map = 1), channel 1 goal or its projection, channel 2 agents."""
    H, W = grid.shape
    N = len(states)
    occ = np.zeros((H, W), dtype=np.int64)
    for n in range(N):
        occ[int(states[n][0]), int(states[n][1])] = 1
    out = np.zeros((N, 3, OBS_HW, OBS_HW), dtype=np.float32)
    for n in range(N):
        cx, cy = int(states[n][0]), int(states[n][1])
        gx, gy = int(goals[n][0]), int(goals[n][1])
        for i in range(FOV):
            for j in range(FOV):
                x, y = cx - FOV_W + i, cy - FOV_W + j
                inside = 0 <= x < H and 0 <= y < W
                out[n, 0, i + 1, j + 1] = grid[x, y] if inside else 1.0
                out[n, 2, i + 1, j + 1] = occ[x, y] if inside else 0.0
        if abs(gx - cx) <= FOV_W and abs(gy - cy) <= FOV_W:
            out[n, 1, gx - cx + FOV_W + 1, gy - cy + FOV_W + 1] = 1.0
        else:
            px, py = projected_goal(gx - cx, gy - cy)
            out[n, 1, px, py] = 1.0
    return out


def _connected(adj):
    n = adj.shape[0]
    seen = np.zeros(n, dtype=bool)
    seen[0] = True
    stack = [0]
    while stack:
        i = stack.pop()
        for j in np.nonzero(adj[i])[0]:
            if not seen[j]:
                seen[j] = True
                stack.append(j)
    return bool(seen.all())


def communication_gso(states, radius, grow):
    """states [N,2]; radius: current communication radius; grow: True at step 0 (the radius is
    divided by 1.1 once and multiplied by 1.1 until the graph is connected, :342-348).
    Returns (S [N,N] float64, radius, connected).  Connectivity by graph search -- the same
    boolean as the reference's Laplacian-spectrum test (graphTools.py:396-423)."""
    pos = np.asarray(states, dtype=np.float64)
    N = pos.shape[0]
    d = np.sqrt(((pos[:, None, :] - pos[None, :, :]) ** 2).sum(-1))

    def adjacency(r):
        A = (d < r).astype(np.float64)
        np.fill_diagonal(A, 0.0)
        return A

    if grow:
        radius = radius / 1.1
        connected = False
        while connected is False:
            radius = radius * 1.1
            A = adjacency(radius)
            connected = _connected(A)
    else:
        A = adjacency(radius)
        connected = _connected(A)
    deg = A.sum(axis=1)
    zero = np.abs(deg) < 1e-9
    deg[zero] = 1.0
    inv = np.sqrt(1.0 / deg)
    inv[zero] = 0.0
    S = (inv[:, None] * A) * inv[None, :]
    return S, radius, connected


class EpisodeState:
    """Mutable per-episode state of the reference simulator that move() touches."""

    def __init__(self, grid, goals, starts, maxstep):
        self.grid = np.asarray(grid)
        self.goal = np.asarray(goals, dtype=np.int64).copy()
        self.cur = np.asarray(starts, dtype=np.int64).copy()
        self.nxt = self.cur.copy()
        self.N = len(self.cur)
        self.maxstep = int(maxstep)
        self.reached = [False] * self.N
        self.start_step = [None] * self.N
        self.end_step = [None] * self.N
        self.last_action = [STOP] * self.N
        self.makespan = self.maxstep
        self.flowtime = self.maxstep * self.N
        self.done = False          # the case loop of the agent has ended (loop_step below)


def _inter_robot_collision(ep, choose):
    """interRobotCollision, utils/multirobotsim_dcenlocal.py:462-555."""
    N = ep.N
    collision = False
    snapshot = [tuple(ep.nxt[i]) for i in range(N)]        # allagents_pos: never updated (:470-475)
    list_pos = list(snapshot)                              # list_pos: updated as agents are stopped
    for i in range(N):
        pos = list_pos[i]
        if list_pos.count(pos) > 1:
            collision = True
            collided = [j for j in range(N) if snapshot[j] == pos]
            mover = choose(collided)
            for j in collided:
                if ep.last_action[j] == STOP:
                    for k in collided:                     # one of them stands still: all stop
                        ep.last_action[k] = STOP
                        ep.nxt[k] = ep.cur[k]
                        list_pos[k] = tuple(ep.nxt[k])
                elif j != mover:
                    ep.last_action[j] = STOP
                    ep.nxt[j] = ep.cur[j]
                    list_pos[j] = tuple(ep.nxt[j])
    # position swaps (:524-553)
    list_next = [tuple(ep.nxt[i]) for i in range(N)]
    for i in range(N):
        cur = tuple(ep.cur[i])
        if cur in list_next:
            s = list_next.index(cur)
            if s != i and tuple(ep.cur[s]) == tuple(ep.nxt[i]):
                ep.nxt[i] = ep.cur[i]
                ep.nxt[s] = ep.cur[s]
                ep.last_action[i] = STOP
                ep.last_action[s] = STOP
                collision = True
    return collision


def move_step(ep, action_ids, currentstep, choose):
    """multiRobotSim.move, :562-723, given the decoded action ids (argmax of LogSoftmax, :589-591).
    Returns (allReachGoal at entry, check_moveCollision, check_predictCollision)."""
    H, W = ep.grid.shape
    all_reached = all(ep.reached)
    predict_collision = False
    move_collision = False
    if (not all_reached) or (currentstep < ep.maxstep):
        for i in range(ep.N):
            key = int(action_ids[i])
            if key != STOP and ep.start_step[i] is None:
                ep.start_step[i] = currentstep - 1
            nx, ny = ep.cur[i][0] + DELTA[key][0], ep.cur[i][1] + DELTA[key][1]
            edge = nx >= H or nx < 0 or ny >= W or ny < 0
            if edge or ep.grid[nx, ny] == 1:
                predict_collision = True
                ep.last_action[i] = STOP
                ep.nxt[i] = ep.cur[i]
            else:
                ep.last_action[i] = key
                ep.nxt[i] = (nx, ny)
        detect = _inter_robot_collision(ep, choose)
        for _ in range(ep.N):
            if detect:
                detect = _inter_robot_collision(ep, choose)
                predict_collision = True
            else:
                break
        move_collision = _inter_robot_collision(ep, choose)
        for i in range(ep.N):
            ep.cur[i] = ep.nxt[i]
            if tuple(ep.nxt[i]) == tuple(ep.goal[i]) and not ep.reached[i]:
                ep.reached[i] = True
                ep.end_step[i] = currentstep
            if currentstep >= ep.maxstep and not ep.reached[i]:
                ep.end_step[i] = currentstep
                if ep.start_step[i] is None:
                    ep.start_step[i] = 0
    if all_reached or currentstep >= ep.maxstep:
        ep.flowtime = sum(ep.end_step[i] - ep.start_step[i] for i in range(ep.N))
        ep.makespan = max(ep.end_step) - min(ep.start_step)
    return all_reached, move_collision, predict_collision


def loop_step(ep, action_ids, currentstep, choose):
    """One iteration of the agent's per-case loop around move()
    (agents/decentralplannerlocal.py:560-605): `for step in range(maxstep)` with currentStep =
    step + 1, `break` after the call that returned allReachGoal or ran at currentStep >= maxstep.
    Once the loop has ended the simulator is never called again, so a batch that keeps stepping its
    other episodes must leave this one untouched: returns (allReachGoal, False, False) and changes
    nothing."""
    if ep.done or currentstep > ep.maxstep:
        return all(ep.reached), False, False
    out = move_step(ep, action_ids, currentstep, choose)
    if out[0] or currentstep >= ep.maxstep:
        ep.done = True
    return out
