"""Probe: ONE policy step (B graphs) as two half batches on two HIP streams, driven at the C-ABI level
(gnnpp_policy_fwd per half, fork / join with events) -- would an internal two-stream split of the two-kernel policy
step pay?  Prints microseconds per whole step."""
import ctypes, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_pathplanning_amd import _native
from gnn_pathplanning_amd.decentralplanner import DecentralPlannerNet
from oracle import policy_oracle as orc                      # (inputs only)

dev = torch.device('cuda:0')
L = _native.lib()
for (N, B) in ((50, 256), (100, 128)):
    class Cfg:
        num_agents, nGraphFilterTaps, device = N, 3, dev
    net = DecentralPlannerNet(Cfg()).to(dev).eval()
    net.load_state_dict(orc.init_state_dict(3))
    obs = orc.synth_obs(B, N, seed=1).to(dev)
    S = torch.from_numpy(orc.synth_gso_geometric(B, N, N, seed=1)).float().to(dev)
    net.addGSO(S)
    net(obs)
    enc, taps, gb, aw, ab, K = net.policy_pointers()
    h = B // 2
    ws = torch.empty(B * N, 128, device=dev)
    lg = torch.empty(N, B, 5, device=dev)
    lgh = [torch.empty(N, h, 5, device=dev), torch.empty(N, B - h, 5, device=dev)]
    obs_h = [obs[:h].contiguous(), obs[h:].contiguous()]
    S_h = [S[:h].contiguous(), S[h:].contiguous()]
    ws_h = [ws[:h * N], ws[h * N:]]
    cur = torch.cuda.current_stream(dev)

    def call(o, s, w, l, b, st):
        rc = L.gnnpp_policy_fwd(o.data_ptr(), s.data_ptr(), enc, taps, gb, aw, ab, w.data_ptr(), l.data_ptr(),
                                b, N, 3, 1, 0, None, ctypes.c_void_p(st.cuda_stream))
        assert rc == 0, rc

    def whole():
        call(obs, S, ws, lg, B, cur)

    def timeit(fn, reps=300):
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / reps)
        return sorted(ts)[2] * 1e6

    row = {'N': N, 'B': B, 'one_call_us': round(timeit(whole), 2)}
    for tag, prios in (('two_streams', (0, 0)), ('two_streams_first_high', (-1, 0))):
        sts = [torch.cuda.Stream(device=dev, priority=p) for p in prios]
        evs = [torch.cuda.Event() for _ in range(3)]

        def split():
            evs[0].record(cur)
            for i, st in enumerate(sts):
                st.wait_event(evs[0])
                call(obs_h[i], S_h[i], ws_h[i], lgh[i], lgh[i].shape[1], st)
                evs[1 + i].record(st)
            cur.wait_event(evs[1]); cur.wait_event(evs[2])
        row[tag + '_us'] = round(timeit(split), 2)
    # the same two half calls one after the other on ONE stream (what the split costs without any overlap)
    def serial():
        for i in range(2):
            call(obs_h[i], S_h[i], ws_h[i], lgh[i], lgh[i].shape[1], cur)
    row['two_calls_one_stream_us'] = round(timeit(serial), 2)
    print(json.dumps(row), flush=True)
