// Batched rollout step around the policy forward: the three pieces
// the reference runs on the host, in numpy / python loops, once per simulated timestep and case:
//
//   observe  <- AgentState.toInputTensor            dataloader/statetransformer.py:82-130
//   gso      <- multiRobotSim.computeAdjacencyMatrix utils/multirobotsim_dcenlocal.py:320-365
//   move     <- multiRobotSim.move :562-723 + interRobotCollision :462-555 (collision shielding)
//
// Here B episodes live on the device; one workgroup owns one episode per kernel, so a whole
// rollout step is observe -> gso -> policy_fwd -> move with no host round trip.  This is integer /
// boolean / fp64 work: results are bit-exact against traces of the real simulator.
// Positions are int32 (row, col) pairs (the reference keeps integers in float tensors).
#include "../../include/gnnpp.h"
#include "gnnpp_common.h"

namespace gnnpp {

constexpr int kMaxAgents = GNNPP_ROLLOUT_MAX_AGENTS;

typedef ::gnnpp_rollout RolloutArgs;      // the C-ABI struct itself (include/gnnpp.h)

// ---- observation builder ----------------------------------------------------------------------
// Border cell standing for a goal outside the 9x9 field of view (statetransformer.py:47-66).  The
// reference decides with atan2 against +-pi/4, +-3pi/4 and np.round (half to even); for integer
// offsets that is exactly: "vertical" branch iff |dy| >= |dx| and dy != 0, and round-half-even of
// 5*dx/|dy| (verified exhaustively for |dx|,|dy| <= 150 in tests/test_rollout_oracle.py).
__device__ __forceinline__ int round_half_even_div(int num, int den) {      // den > 0
    int q = num / den, rem = num - q * den;
    if (rem < 0) { rem += den; q -= 1; }                                     // floor division
    const int twice = 2 * rem;
    if (twice > den || (twice == den && (q & 1))) q += 1;
    return q;
}

__device__ __forceinline__ void projected_goal(int dx, int dy, int& px, int& py) {
    const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
    if (ady >= adx && dy != 0) {
        py = dy > 0 ? 10 : 0;
        px = 5 + round_half_even_div(5 * dx, ady);
    } else {
        px = dx > 0 ? 10 : (dx < 0 ? 0 : 5);
        py = 5 + round_half_even_div(5 * dy, adx);
    }
}

__global__ __launch_bounds__(256) void rollout_observe_kernel(const RolloutArgs p) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    unsigned char* occ = reinterpret_cast<unsigned char*>(gnnpp_smem);     // [H*W] agents present
    const int b = blockIdx.x, tid = threadIdx.x;
    const int HW = p.H * p.W;
    const unsigned char* grid = p.grid + (p.grid_batched ? (size_t)b * HW : 0);
    const int* pos = p.pos + (size_t)b * p.N * 2;
    const int* goal = p.goal + (size_t)b * p.N * 2;
    for (int i = tid; i < HW; i += 256) occ[i] = 0;
    __syncthreads();
    for (int n = tid; n < p.N; n += 256) occ[pos[2 * n] * p.W + pos[2 * n + 1]] = 1;
    __syncthreads();
    float* out = p.obs + (size_t)b * p.N * 363;
    for (int e = tid; e < p.N * 363; e += 256) {
        const int n = e / 363, r = e - n * 363;
        const int ch = r / 121, r2 = r - ch * 121;
        const int i = r2 / 11, j = r2 - i * 11;
        const int cx = pos[2 * n], cy = pos[2 * n + 1];
        float v = 0.f;
        if (ch == 1) {
            const int dx = goal[2 * n] - cx, dy = goal[2 * n + 1] - cy;
            int px, py;
            if (dx >= -4 && dx <= 4 && dy >= -4 && dy <= 4) { px = dx + 5; py = dy + 5; }
            else projected_goal(dx, dy, px, py);
            v = (i == px && j == py) ? 1.f : 0.f;
        } else if (i >= 1 && i <= 9 && j >= 1 && j <= 9) {
            const int x = cx + i - 5, y = cy + j - 5;
            const bool inside = x >= 0 && x < p.H && y >= 0 && y < p.W;
            if (ch == 0) v = inside ? (float)grid[x * p.W + y] : 1.f;
            else v = inside ? (float)occ[x * p.W + y] : 0.f;
        }
        out[e] = v;
    }
}

// ---- communication GSO ---------------------------------------------------------------------------
// A = (pdist < R) with zero diagonal; at step 0 R is divided by 1.1 once and multiplied by 1.1
// until the graph is connected; S = D^-1/2 A D^-1/2 in fp64 (isolated nodes -> 0), rounded to fp32
// (what `S.float()` does to the simulator's float64 GSO).  Connectivity by graph search on
// adjacency bit masks -- the same boolean as the reference's Laplacian-spectrum test.
__global__ __launch_bounds__(64) void rollout_gso_kernel(const RolloutArgs p) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    unsigned long long* adj = reinterpret_cast<unsigned long long*>(gnnpp_smem);   // [N][2]
    double* inv = reinterpret_cast<double*>(adj + 2 * kMaxAgents);                  // [N]
    double* shared_r = inv + kMaxAgents;                                            // [1]
    int* shared_flag = reinterpret_cast<int*>(shared_r + 1);                        // [1]
    const int b = blockIdx.x, lane = threadIdx.x, N = p.N;
    const int* pos = p.pos + (size_t)b * N * 2;
    if (lane == 0) {
        double r = p.radius[b];
        if (p.grow) r = r / 1.1;
        *shared_r = r;
        *shared_flag = 0;
    }
    __syncthreads();
    for (;;) {
        if (lane == 0 && p.grow) *shared_r = *shared_r * 1.1;
        __syncthreads();
        const double R = *shared_r;
        for (int i = lane; i < N; i += 64) {
            unsigned long long w0 = 0, w1 = 0;
            const int xi = pos[2 * i], yi = pos[2 * i + 1];
            for (int j = 0; j < N; ++j) {
                const int dx = xi - pos[2 * j], dy = yi - pos[2 * j + 1];
                const bool e = (j != i) && (sqrt((double)(dx * dx + dy * dy)) < R);
                if (e) { if (j < 64) w0 |= 1ull << j; else w1 |= 1ull << (j - 64); }
            }
            adj[2 * i] = w0; adj[2 * i + 1] = w1;
        }
        __syncthreads();
        if (lane == 0) {
            unsigned long long r0 = 1ull, r1 = 0, d0 = 0, d1 = 0;      // reached / expanded
            for (;;) {
                const unsigned long long f0 = r0 & ~d0, f1 = r1 & ~d1;
                if (!(f0 | f1)) break;
                int m;
                if (f0) { m = __ffsll((long long)f0) - 1; d0 |= 1ull << m; }
                else { m = 64 + __ffsll((long long)f1) - 1; d1 |= 1ull << (m - 64); }
                r0 |= adj[2 * m]; r1 |= adj[2 * m + 1];
            }
            const int cnt = __popcll(r0) + __popcll(r1);
            *shared_flag = (cnt == N);
        }
        __syncthreads();
        if (*shared_flag || !p.grow) break;
        __syncthreads();
    }
    for (int i = lane; i < N; i += 64) {
        const int deg = __popcll(adj[2 * i]) + __popcll(adj[2 * i + 1]);
        inv[i] = deg ? sqrt(1.0 / (double)deg) : 0.0;
    }
    __syncthreads();
    float* S = p.S + (size_t)b * N * N;
    for (int e = lane; e < N * N; e += 64) {
        const int i = e / N, j = e - i * N;
        const bool on = j < 64 ? (adj[2 * i] >> j) & 1ull : (adj[2 * i + 1] >> (j - 64)) & 1ull;
        S[e] = on ? (float)(inv[i] * inv[j]) : 0.f;
    }
    if (lane == 0) {
        p.radius[b] = *shared_r;
        if (p.connected) p.connected[b] = *shared_flag;
    }
}

// ---- move + collision shielding -------------------------------------------------------------------
struct MoveScratch {             // all in LDS, N <= kMaxAgents
    int cur[kMaxAgents][2];
    int nxt[kMaxAgents][2];
    int snap[kMaxAgents][2];     // allagents_pos: snapshot, never updated inside one call
    int lpos[kMaxAgents][2];     // list_pos: updated as agents are stopped
    int last_action[kMaxAgents];
    int collided[kMaxAgents];
};

__device__ __forceinline__ unsigned hash_u32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// random.choice(collided_agents) of the reference (:489): index into the collided list.
__device__ int choose_mover(const RolloutArgs& p, int b, int ncol, int& calls) {
    int k = 0;
    if (p.tie_mode == 1) {
        k = (int)(hash_u32(p.seed ^ hash_u32((unsigned)b * 0x9E3779B9u + (unsigned)p.currentstep * 0x85EBCA6Bu +
                                             (unsigned)calls)) % (unsigned)ncol);
    } else if (p.tie_mode == 2) {
        const int c = calls < p.max_choices ? (int)p.choices[(size_t)b * p.max_choices + calls] : 0;
        k = c < ncol ? c : 0;
    }
    ++calls;
    return k;
}

__device__ bool inter_robot_collision(const RolloutArgs& p, MoveScratch& s, int b, int N, int& calls) {
    bool collision = false;
    for (int i = 0; i < N; ++i) {
        s.snap[i][0] = s.nxt[i][0]; s.snap[i][1] = s.nxt[i][1];
        s.lpos[i][0] = s.nxt[i][0]; s.lpos[i][1] = s.nxt[i][1];
    }
    for (int i = 0; i < N; ++i) {
        const int px = s.lpos[i][0], py = s.lpos[i][1];
        int count = 0;
        for (int j = 0; j < N; ++j) count += (s.lpos[j][0] == px && s.lpos[j][1] == py);
        if (count > 1) {
            collision = true;
            int ncol = 0;
            for (int j = 0; j < N; ++j)
                if (s.snap[j][0] == px && s.snap[j][1] == py) s.collided[ncol++] = j;
            const int mover = s.collided[choose_mover(p, b, ncol, calls)];
            for (int c = 0; c < ncol; ++c) {
                const int j = s.collided[c];
                if (s.last_action[j] == 4) {
                    for (int c2 = 0; c2 < ncol; ++c2) {           // one stands still: all stop
                        const int k = s.collided[c2];
                        s.last_action[k] = 4;
                        s.nxt[k][0] = s.cur[k][0]; s.nxt[k][1] = s.cur[k][1];
                        s.lpos[k][0] = s.nxt[k][0]; s.lpos[k][1] = s.nxt[k][1];
                    }
                } else if (j != mover) {
                    s.last_action[j] = 4;
                    s.nxt[j][0] = s.cur[j][0]; s.nxt[j][1] = s.cur[j][1];
                    s.lpos[j][0] = s.nxt[j][0]; s.lpos[j][1] = s.nxt[j][1];
                }
            }
        }
    }
    // position swaps (:524-553): list_nextpos is a snapshot taken here
    for (int i = 0; i < N; ++i) { s.snap[i][0] = s.nxt[i][0]; s.snap[i][1] = s.nxt[i][1]; }
    for (int i = 0; i < N; ++i) {
        int sidx = -1;
        for (int j = 0; j < N; ++j)
            if (s.snap[j][0] == s.cur[i][0] && s.snap[j][1] == s.cur[i][1]) { sidx = j; break; }
        if (sidx >= 0 && sidx != i && s.cur[sidx][0] == s.nxt[i][0] && s.cur[sidx][1] == s.nxt[i][1]) {
            s.nxt[i][0] = s.cur[i][0]; s.nxt[i][1] = s.cur[i][1];
            s.nxt[sidx][0] = s.cur[sidx][0]; s.nxt[sidx][1] = s.cur[sidx][1];
            s.last_action[i] = 4; s.last_action[sidx] = 4;
            collision = true;
        }
    }
    return collision;
}

__global__ __launch_bounds__(64) void rollout_move_kernel(const RolloutArgs p) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    MoveScratch& s = *reinterpret_cast<MoveScratch*>(gnnpp_smem);
    const int b = blockIdx.x, lane = threadIdx.x, N = p.N;
    int* pos = p.pos + (size_t)b * N * 2;
    // decode the actions in parallel (argmax of the logits == argmax of LogSoftmax, first max wins)
    for (int n = lane; n < N; n += 64) {
        int key;
        if (p.logits) {
            const float* l = p.logits + ((size_t)n * p.B + b) * 5;
            key = 0;
            float best = l[0];
#pragma unroll
            for (int k = 1; k < 5; ++k)
                if (l[k] > best) { best = l[k]; key = k; }
        } else {
            key = p.actions[(size_t)b * N + n];
        }
        s.collided[n] = key;                                    // staging slot for the decoded key
        s.cur[n][0] = pos[2 * n]; s.cur[n][1] = pos[2 * n + 1];
    }
    __syncthreads();
    if (lane != 0) return;                                      // the shielding logic is sequential

    const unsigned char* grid = p.grid + (p.grid_batched ? (size_t)b * p.H * p.W : 0);
    const int* goal = p.goal + (size_t)b * N * 2;
    int* reached = p.reached + (size_t)b * N;
    int* start_step = p.start_step + (size_t)b * N;
    int* end_step = p.end_step + (size_t)b * N;
    const int step = p.currentstep, maxstep = p.maxstep[b];
    const int dxs[5] = {-1, 0, 1, 0, 0}, dys[5] = {0, -1, 0, 1, 0};
    bool all_reached = true;
    for (int n = 0; n < N; ++n) all_reached = all_reached && reached[n];
    bool predict_collision = false, move_collision = false;
    int calls = 0;
    if (!all_reached || step < maxstep) {
        for (int n = 0; n < N; ++n) {
            const int key = s.collided[n];
            if (key != 4 && start_step[n] < 0) start_step[n] = step - 1;
            const int nx = s.cur[n][0] + dxs[key], ny = s.cur[n][1] + dys[key];
            const bool edge = nx >= p.H || nx < 0 || ny >= p.W || ny < 0;
            if (edge || grid[nx * p.W + ny] == 1) {
                predict_collision = true;
                s.last_action[n] = 4;
                s.nxt[n][0] = s.cur[n][0]; s.nxt[n][1] = s.cur[n][1];
            } else {
                s.last_action[n] = key;
                s.nxt[n][0] = nx; s.nxt[n][1] = ny;
            }
        }
        bool detect = inter_robot_collision(p, s, b, N, calls);
        for (int it = 0; it < N; ++it) {
            if (!detect) break;
            detect = inter_robot_collision(p, s, b, N, calls);
            predict_collision = true;
        }
        move_collision = inter_robot_collision(p, s, b, N, calls);
        for (int n = 0; n < N; ++n) {
            pos[2 * n] = s.nxt[n][0]; pos[2 * n + 1] = s.nxt[n][1];
            if (s.nxt[n][0] == goal[2 * n] && s.nxt[n][1] == goal[2 * n + 1] && !reached[n]) {
                reached[n] = 1;
                end_step[n] = step;
            }
            if (step >= maxstep && !reached[n]) {
                end_step[n] = step;
                if (start_step[n] < 0) start_step[n] = 0;
            }
        }
    }
    if (all_reached || step >= maxstep) {
        // An agent that reached its goal without ever issuing a move keeps start_step = None in
        // the reference (which then raises TypeError on `end - None`); we count it from step 0.
        int flow = 0, emax = -(1 << 30), smin = 1 << 30;
        for (int n = 0; n < N; ++n) {
            const int st = start_step[n] < 0 ? 0 : start_step[n];
            flow += end_step[n] - st;
            emax = end_step[n] > emax ? end_step[n] : emax;
            smin = st < smin ? st : smin;
        }
        p.stats[2 * b] = emax - smin;
        p.stats[2 * b + 1] = flow;
    }
    p.flags[3 * b] = all_reached;
    p.flags[3 * b + 1] = move_collision;
    p.flags[3 * b + 2] = predict_collision;
    if (p.choice_count) p.choice_count[b] = calls;
}

// ---- launchers -----------------------------------------------------------------------------------
int rollout_observe_launch(const RolloutArgs& a, hipStream_t st) {
    const size_t smem = ((size_t)a.H * a.W + 15) & ~(size_t)15;
    if (smem > 64 * 1024) return -2;
    hipLaunchKernelGGL(rollout_observe_kernel, dim3(a.B), dim3(256), smem, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int rollout_gso_launch(const RolloutArgs& a, hipStream_t st) {
    const size_t smem = 2 * kMaxAgents * 8 + kMaxAgents * 8 + 16;
    hipLaunchKernelGGL(rollout_gso_kernel, dim3(a.B), dim3(64), smem, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int rollout_move_launch(const RolloutArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(rollout_move_kernel, dim3(a.B), dim3(64), sizeof(MoveScratch), st, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace gnnpp
