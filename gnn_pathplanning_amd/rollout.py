"""Batched on-device rollouts: B independent MAPF episodes stepped together on one MI355X.

Mirrors what the reference does per test case on the host (agents/decentralplannerlocal.py:534-592
driving utils/multirobotsim_dcenlocal.py), with the episode state resident in HBM:

    sim.getCurrentState()  -> BatchedRollout.observe()   AgentState.toInputTensor (statetransformer.py:82-130)
    sim.getGSO(step)       -> BatchedRollout.gso(step)   computeAdjacencyMatrix (multirobotsim_dcenlocal.py:320-394)
    model.addGSO / model() -> DecentralPlannerNet.forward_logits
    sim.move(actionVec, t) -> BatchedRollout.move(...)   move + interRobotCollision (:462-723)

A step is ONE kernel launch for teams of up to 16 agents (policy forward, move, next GSO and
observations in the same workgroup per episode), otherwise encoder, filter + head and the simulator
kernels, and never a host synchronisation; `run()` only reads back a "finished"
flag every few steps.  The reference's random.choice tie-break among colliding agents (:489) is
replaced by a deterministic rule (`tie_mode`): 'lowest' index, 'hashed' counter-based RNG, 'replay' of
recorded choices (parity tests), or 'mt19937' = CPython's random.choice itself on a per-episode
Mersenne-Twister stream (episode b behaves as the reference does after random.seed(seed[b])).
Positions are (row, col) integers.
"""
import ctypes

import torch

from . import _native

_TIE = {'lowest': 0, 'hashed': 1, 'replay': 2, 'mt19937': 3}


def _p(t):
    return t.data_ptr() if t is not None else None


class BatchedRollout:
    def __init__(self, grid, starts, goals, maxstep, device, commR=6.0, tie_mode='lowest', seed=0,
                 rng_words=2048):
        """grid [B,H,W] or [H,W] (1 = obstacle); starts, goals [B,N,2]; maxstep int or [B]."""
        dev = torch.device(device)
        if dev.type != 'cuda':
            raise _native.GnnppError('BatchedRollout needs a HIP device (no CPU fallback)')
        _native.lib()
        self.device = dev
        g = torch.as_tensor(grid)
        self.grid_batched = int(g.dim() == 3)
        self.grid = g.to(torch.uint8).contiguous().to(dev)
        self.H, self.W = int(g.shape[-2]), int(g.shape[-1])
        self.pos = torch.as_tensor(starts).to(torch.int32).contiguous().to(dev)
        self.goal = torch.as_tensor(goals).to(torch.int32).contiguous().to(dev)
        self.B, self.N = int(self.pos.shape[0]), int(self.pos.shape[1])
        assert self.goal.shape == self.pos.shape and self.pos.shape[2] == 2
        assert not self.grid_batched or self.grid.shape[0] == self.B
        if self.N > 128:
            raise _native.GnnppError('at most 128 agents per episode')
        ms = torch.as_tensor(maxstep, dtype=torch.int32)
        self.maxstep = (ms if ms.dim() else ms.repeat(self.B)).contiguous().to(dev)
        B, N = self.B, self.N
        self.obs = torch.empty(B, N, 3, 11, 11, dtype=torch.float32, device=dev)
        self.radius = torch.full((B,), float(commR), dtype=torch.float64, device=dev)
        self.S = torch.empty(B, N, N, dtype=torch.float32, device=dev)
        self.connected = torch.zeros(B, dtype=torch.int32, device=dev)
        self.reached = torch.zeros(B, N, dtype=torch.int32, device=dev)
        self.start_step = torch.full((B, N), -1, dtype=torch.int32, device=dev)
        self.end_step = torch.full((B, N), -1, dtype=torch.int32, device=dev)
        self.flags = torch.zeros(B, 3, dtype=torch.int32, device=dev)
        self.done = torch.zeros(B, dtype=torch.int32, device=dev)   # the case's loop has ended: frozen
        self.stats = torch.zeros(B, 2, dtype=torch.int32, device=dev)
        self.choice_count = torch.zeros(B, dtype=torch.int32, device=dev)
        self.tie_mode = _TIE[tie_mode]
        self.rng_words = self.rng_cursor = None
        if self.tie_mode == 3:
            # raw genrand_uint32() outputs of random.Random(seed_b): the kernel applies random.choice's
            # own rejection sampling to them (csrc/rollout_kernels.hip::choose_mover)
            import random
            seeds = list(seed) if hasattr(seed, '__len__') else [int(seed) + b for b in range(self.B)]
            assert len(seeds) == self.B
            import numpy as np
            rows = np.array([[g.getrandbits(32) for _ in range(int(rng_words))]
                             for g in (random.Random(int(sd)) for sd in seeds)], dtype=np.uint32)
            self.rng_words = torch.from_numpy(rows.view(np.int32)).to(dev)
            self.rng_cursor = torch.zeros(self.B, dtype=torch.int32, device=dev)
            seed = 0
        self.seed = int(seed) & 0xffffffff
        self.t = 0                                            # steps taken so far
        self._state_step = -1                                 # step whose positions obs / S describe
        # teams up to this size run move -> graph -> observations as ONE launch (gnnpp_rollout_step, one workgroup
        # per episode); larger ones as move, then graph || observations in a second launch (gnnpp_rollout_gso_observe:
        # more workgroups than episodes).  Measured per simulator step: N = 10 16.1 (one launch) vs 17.3 us,
        # N = 50 33.2 vs 28.5, N = 100 52.4 vs 42.4
        self.fused_sim_max_agents = 32
        self._logits = None                                   # [N,B,5] of the one-launch step
        r = _native.RolloutStruct()
        r.grid, r.grid_batched, r.goal, r.pos = _p(self.grid), self.grid_batched, _p(self.goal), _p(self.pos)
        r.B, r.N, r.H, r.W = B, N, self.H, self.W
        r.obs, r.radius, r.S, r.connected = _p(self.obs), _p(self.radius), _p(self.S), _p(self.connected)
        r.reached, r.start_step, r.end_step = _p(self.reached), _p(self.start_step), _p(self.end_step)
        r.maxstep, r.flags, r.stats = _p(self.maxstep), _p(self.flags), _p(self.stats)
        r.done = _p(self.done)
        r.tie_mode, r.seed, r.choice_count = self.tie_mode, self.seed, _p(self.choice_count)
        if self.tie_mode == 3:
            r.rng_words, r.rng_cursor, r.rng_max = _p(self.rng_words), _p(self.rng_cursor), int(rng_words)
        self._r = r

    # -- the three simulator calls -------------------------------------------------------------
    def _call(self, fn, what):
        with _native.device_guard(self.device):
            _native.check(fn(ctypes.byref(self._r), _native.stream_ptr(self.device)), what)

    def observe(self):
        """[B,N,3,11,11] float32 observations of the current positions (buffer reused each step)."""
        self._call(_native.lib().gnnpp_rollout_observe, 'gnnpp_rollout_observe')
        return self.obs

    def gso(self, step=None):
        """[B,N,N] float32 GSO of the current positions.  step 0 grows the radius until connected."""
        step = self.t if step is None else step
        self._r.grow = int(step == 0)
        self._call(_native.lib().gnnpp_rollout_gso, 'gnnpp_rollout_gso')
        return self.S

    def gso_observe(self):
        """gso() (no radius growth: not for step 0) and observe() of the current positions as ONE launch
        (gnnpp_rollout_gso_observe): the two are independent given the positions."""
        self._r.grow = 0
        self._call(_native.lib().gnnpp_rollout_gso_observe, 'gnnpp_rollout_gso_observe')
        self._state_step = self.t
        return self.obs, self.S

    def move(self, logits=None, actions=None, choices=None, currentstep=None):
        """Apply one joint action.  logits [N,B,5] (DecentralPlannerNet.forward_logits) or action
        ids [B,N] int32; `choices` [B,C] int16 only for tie_mode='replay'.  Returns flags [B,3]
        (allReachGoal at entry, moveCollision, predictCollision) -- a device tensor."""
        self._prepare_move(logits, actions, choices, currentstep)
        self._call(_native.lib().gnnpp_rollout_move, 'gnnpp_rollout_move')
        return self.flags

    def _prepare_move(self, logits, actions, choices, currentstep):
        r = self._r
        keep = []
        if logits is not None:
            lg = logits.detach().contiguous().float()
            assert lg.shape == (self.N, self.B, 5)
            keep.append(lg)
            r.logits, r.actions = _p(lg), None
        else:
            ac = actions.to(torch.int32).contiguous()
            assert ac.shape == (self.B, self.N)
            keep.append(ac)
            r.logits, r.actions = None, _p(ac)
        if self.tie_mode == 2:
            ch = choices.to(torch.int16).contiguous().to(self.device)
            keep.append(ch)
            r.choices, r.max_choices = _p(ch), int(ch.shape[1])
        self.t += 1
        r.currentstep = self.t if currentstep is None else int(currentstep)
        self._keep = keep                                   # alive until the launch has been enqueued

    # -- whole episodes ------------------------------------------------------------------------------
    def move_and_observe(self, logits=None, actions=None, choices=None, currentstep=None):
        """move(), then the next step's gso() and observe(), as ONE kernel launch
        (gnnpp_rollout_step).  Same results as the three calls in sequence."""
        self._prepare_move(logits, actions, choices, currentstep)
        self._r.grow = 0
        self._call(_native.lib().gnnpp_rollout_step, 'gnnpp_rollout_step')
        self._state_step = self.t                          # obs / S describe the positions after step t
        return self.flags

    def _policy_step(self, model, nsteps=1):
        """gnnpp_rollout_policy_step(s): policy forward + move + next gso/observe in ONE kernel per step, when
        the model and the team qualify (eval mode, N = model.numAgents <= 16, K = 2..4, not 'replay'); nsteps
        launches are enqueued by one C call."""
        if (self.N > 16 or self.tie_mode == 2 or getattr(model, 'training', True)
                or getattr(model, 'numAgents', -1) != self.N or not hasattr(model, 'policy_pointers')):
            return False
        ptrs = model.policy_pointers()
        if ptrs is None or not 2 <= ptrs[5] <= 4:            # (the fused kernel exists for K = 2, 3, 4 taps)
            return False
        enc, taps, gb, aw, ab, K = ptrs
        if self._logits is None:
            self._logits = torch.empty(self.N, self.B, 5, dtype=torch.float32, device=self.device)
        r = self._r
        r.logits, r.actions, r.grow = _p(self._logits), None, 0
        prec = model._prec() if hasattr(model, '_prec') else _native.PREC_FP32
        # range guard: only the opt-in split-f16 arithmetic has an input domain
        r.range_flag = _p(model._flag(self.device)) if prec == _native.PREC_SPLIT_F16 else None
        r.currentstep = self.t + 1
        with _native.device_guard(self.device):
            rc = _native.lib().gnnpp_rollout_policy_steps(ctypes.byref(r), enc, taps, gb, aw, ab, K, nsteps, prec,
                                                          _native.stream_ptr(self.device))
        if rc == -2:                                         # shape not supported by the fused kernel
            return False
        _native.check(rc, 'gnnpp_rollout_policy_steps')
        self.t += nsteps
        self._state_step = self.t
        return True

    def step(self, model):
        """One rollout step of all episodes: observe -> gso -> policy forward -> move.  From the
        second step on the observation and the GSO were already produced by the previous step's
        fused move kernel."""
        if self.t == 0:                                      # step 0 may grow the radius
            self.observe()
            self.gso()
        elif self._state_step != self.t:                     # stale state: graph and observations side by side
            self.gso_observe()
        if self._policy_step(model):                         # small teams: the whole step is one launch
            return self.flags
        model.addGSO(self.S)
        logits = model.forward_logits(self.obs)
        # one launch for move -> graph -> observations (see fused_sim_max_agents); beyond it: move now, the
        # graph and the observations side by side at the start of the next step
        return self.move_and_observe(logits=logits) if self.N <= self.fused_sim_max_agents else self.move(logits=logits)

    def steps(self, model, n):
        """n rollout steps.  Small teams (the one-launch step): the launches of all n steps are enqueued by ONE
        C call (gnnpp_rollout_policy_steps) -- no interpreter between two steps."""
        done = 0
        if n > 1 and (self._state_step != self.t or self.t == 0):
            self.step(model)                                 # (the first step may grow the radius)
            done = 1
        if n - done > 1 and self._policy_step(model, n - done):
            return self.flags
        for _ in range(n - done):
            self.step(model)
        return self.flags

    def run(self, model, max_steps=None, check_every=8):
        """Step until every episode's loop has ended: the call after its last agent arrived (that call
        writes its statistics, like the reference loop agents/decentralplannerlocal.py:560-605), or
        the call at its OWN maxstep.  An ended episode is frozen by the move kernel (`done`), so mixed
        per-episode limits (maxstep = rate * makespan[b]) report exactly what the reference reports
        case by case."""
        limit = int(self.maxstep.max().item()) if max_steps is None else int(max_steps)
        steps = 0
        while steps < limit:
            burst = min(check_every - steps % check_every, limit - steps)
            self.steps(model, burst)
            steps += burst
            if steps % check_every == 0:
                if hasattr(model, 'check_range'):
                    model.check_range()                      # an activation left the f16 range: error
                if bool((self.done != 0).all().item()):
                    break
        if hasattr(model, 'check_range'):
            model.check_range()
        return self.results()

    def check_rng(self):
        """tie_mode 'mt19937': every episode gets `rng_words` words of its Mersenne-Twister stream; the kernel
        substitutes 0 for words past the end (and keeps counting them in rng_cursor).  An episode that consumed
        more than it was given no longer follows the reference's random.choice stream: that is an error, not a
        silent deviation."""
        if self.rng_cursor is not None:
            over = (self.rng_cursor > int(self.rng_words.shape[1])).nonzero().flatten()
            if over.numel():
                raise _native.GnnppError(
                    'tie_mode mt19937: episode(s) %s consumed more than the %d random words they were given; '
                    'construct the rollout with a larger rng_words' % (over[:8].tolist(), int(self.rng_words.shape[1])))

    def results(self):
        torch.cuda.synchronize(self.device)
        self.check_rng()
        reached = self.reached.bool()
        return {'steps': self.t, 'reached': reached.cpu(), 'success': reached.all(dim=1).cpu(),
                'makespan': self.stats[:, 0].cpu(), 'flowtime': self.stats[:, 1].cpu(),
                'end_step': self.end_step.cpu(), 'start_step': self.start_step.cpu(),
                'positions': self.pos.cpu(), 'radius': self.radius.cpu(), 'done': self.done.bool().cpu()}


class GroupedRollout:
    """The same batch of episodes as `groups` BatchedRollouts over contiguous slices, each on its OWN HIP stream.
    Episodes are independent, so nothing orders one group's step t + 1 after another group's step t: the launch
    gap and the tail of one group's kernel (workgroups finish at different times, the observation stores drain)
    overlap with the other group's kernel.  Measured at C2 (512 episodes of 10 agents, one-launch steps): 44-49 us
    per step of all 512 episodes with two groups against 52-55 us on one stream.

    Same interface as BatchedRollout for what a rollout driver needs: steps(), run(), results().  tie_mode
    'lowest' and 'mt19937' give exactly the single-batch results (episode b keeps its generator seed + b);
    'hashed' hashes the episode's index inside its group, i.e. draws a different -- equally arbitrary -- stream."""

    def __init__(self, grid, starts, goals, maxstep, device, groups=2, seed=0, **kw):
        starts, goals = torch.as_tensor(starts), torch.as_tensor(goals)
        g = torch.as_tensor(grid)
        B = int(starts.shape[0])
        ms = torch.as_tensor(maxstep)
        groups = max(1, min(int(groups), B))
        bounds = [round(i * B / groups) for i in range(groups + 1)]
        self.envs, self.streams, self.slices = [], [], []
        dev = torch.device(device)
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            sd = seed[lo:hi] if hasattr(seed, '__len__') else int(seed) + lo
            self.envs.append(BatchedRollout(g[lo:hi] if g.dim() == 3 else g, starts[lo:hi], goals[lo:hi],
                                            ms[lo:hi] if ms.dim() else ms, dev, seed=sd, **kw))
            self.streams.append(torch.cuda.Stream(device=dev))
            self.slices.append((lo, hi))
        self.device, self.B, self.N = dev, B, self.envs[0].N

    def _each(self, fn, wait_caller=True, model=None):
        cur = torch.cuda.current_stream(self.device)
        if model is not None and hasattr(model, 'materialize_packs'):
            # The model's lazily built device caches (packed encoder, packed taps, head pointers) are rebuilt on
            # whichever stream touches them first after a weight change.  Build them HERE, on the caller's stream,
            # which every group stream then waits for -- otherwise the second group could launch on a pack the
            # first group's stream is still writing (a cross-stream read-before-write race).
            if model.materialize_packs():
                wait_caller = True                           # (the groups must see the packs just enqueued)
        for env, st in zip(self.envs, self.streams):
            if wait_caller:
                st.wait_stream(cur)                          # whatever the caller's stream prepared (the episodes'
            with torch.cuda.stream(st):                      # state, new weights) is visible to the group
                fn(env)

    @property
    def t(self):
        return self.envs[0].t                                # (all groups step together)

    def join(self):
        """Make the caller's stream wait for every group's work."""
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            cur.wait_stream(st)

    def steps(self, model, n, wait_caller=True):
        """n steps of every group.  wait_caller=False: the groups do not wait for the caller's stream first --
        for back-to-back bursts with nothing in between (after a join() a wait would be a barrier across the
        groups: the faster group would idle until the slower one has finished the previous burst)."""
        self._each(lambda env: env.steps(model, n), wait_caller, model)

    def run(self, model, max_steps=None, check_every=8):
        limit = max(int(e.maxstep.max().item()) for e in self.envs) if max_steps is None else int(max_steps)
        steps = 0
        while steps < limit:
            burst = min(check_every - steps % check_every, limit - steps)
            self.steps(model, burst, wait_caller=steps == 0)
            steps += burst
            if steps % check_every == 0:
                self.join()
                if hasattr(model, 'check_range'):
                    model.check_range()
                if all(bool((e.done != 0).all().item()) for e in self.envs):
                    break
        self.join()
        if hasattr(model, 'check_range'):
            model.check_range()
        return self.results()

    def results(self):
        self.join()
        parts = [e.results() for e in self.envs]
        out = {'steps': max(p['steps'] for p in parts)}
        for k in parts[0]:
            if k != 'steps':
                out[k] = torch.cat([p[k] for p in parts], 0)
        return out


class GraphedPolicyStep:
    """`model.addGSO(S); model(obs)` of a fixed batch shape as ONE HIP-graph replay.

    For small batches -- one test case per step as in the reference's rollout loop
    (agents/decentralplannerlocal.py:560-599), or the 16-graph shard a rank holds when a 128-graph batch of
    100-agent teams is split over 8 GPUs -- the policy step is a chain of one to three kernels that fill a
    fraction of the chip, and the launch boundaries between them are a tenth of the step.  The libgnnpp
    kernels are enqueued on torch's capture stream through the C ABI like any other launch, so the whole
    step (GSO hand-over, encoder, filter + head) becomes one graph; inputs are copied into static buffers
    before a replay, the returned logits are the graph's own output tensors (valid until the next call).

        step = GraphedPolicyStep(model, obs, S)        # eval mode; captures on the current device
        logits = step(obs_t, S_t)                      # list of N tensors [B, 5], as model(obs) returns

    The weights are FROZEN at capture: the graph holds the device addresses of the model's packed weight copies
    (encoder pack, filter taps, head constants).  The step keeps those buffers alive and, before every replay,
    compares the caches' keys (parameter version counters / identities: host-only, a few us) with the captured ones;
    after load_state_dict / an optimizer step / .to() it RE-CAPTURES (`recaptures` counts them) instead of replaying
    freed or stale buffers.  precision='split_f16' with range_policy='strict' is refused: its range check reads a
    device flag on the host, which cannot happen inside a capture (use range_policy='flag' + check_range()).

    Bit-identical to the eager call (tests/test_gpu_parity.py::test_graphed_policy_step_equals_eager)."""

    def __init__(self, model, obs, S, warmup=3):
        if model.training:
            raise ValueError('GraphedPolicyStep captures the eval-mode forward (BatchNorm folded into the packed weights)')
        if getattr(model, 'precision', 'fp32') == 'split_f16' and getattr(model, 'range_policy', None) == 'strict':
            raise ValueError("GraphedPolicyStep cannot capture precision='split_f16' with range_policy='strict' (the "
                             "range check synchronises with the host); use range_policy='flag' and check_range()")
        self.model = model
        self.obs = obs.clone()
        self.S = S.clone()
        self.warmup = warmup
        self.recaptures = -1
        self._retired = []                                 # graphs (and their output pools) replaced by a re-capture
        self._capture()

    def _capture(self):
        model = self.model
        if self.recaptures >= 0:
            # a re-capture used to release the old graph's private pool while logits returned by earlier
            # calls still viewed it, and ran warm-up + capture inside a latency-critical step without a word.  The
            # replaced graph (and with it the memory its outputs live in) is kept -- the last few of them -- and the
            # caller is told.
            import warnings
            self._retired = (self._retired + [(self.graph, self.out)])[-4:]
            warnings.warn('GraphedPolicyStep: the model\'s weights changed since the capture -- re-capturing (warm-up + '
                          'capture inside this call; outputs of earlier calls keep their old values)', RuntimeWarning,
                          stacklevel=3)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):                   # packs, workspaces, allocator
                model.addGSO(self.S)
                model(self.obs)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            model.addGSO(self.S)
            self.out = model(self.obs)
        # what the graph's kernels read through raw pointers: keep it alive, remember what it was built from
        self._keys, self._held = model.pack_state()
        self.recaptures += 1

    def __call__(self, obs, S):
        if obs.shape != self.obs.shape or S.shape != self.S.shape:
            raise ValueError('GraphedPolicyStep was captured for observations %s and GSOs %s'
                             % (tuple(self.obs.shape), tuple(self.S.shape)))
        if self.model.training:
            raise ValueError('GraphedPolicyStep replays the eval-mode forward: call model.eval() first')
        if self.model.pack_state()[0] != self._keys:       # weights changed since the capture
            self._capture()
        self.obs.copy_(obs)
        self.S.copy_(S)
        self.graph.replay()
        return self.out
