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

// ---- wave-level helpers: lane l of a wave holds agents l and l + 64 (teams of up to 128) -------------
__device__ __forceinline__ int lane_get(const int (&v)[2], int agent) {
    return __builtin_amdgcn_readlane(agent < 64 ? v[0] : v[1], agent & 63);
}

struct MaskPair { unsigned long long lo, hi; };

__device__ __forceinline__ MaskPair ballot2(bool p0, bool p1) {
    MaskPair m;
    m.lo = __ballot(p0);
    m.hi = __ballot(p1);
    return m;
}


// ---- observation builder ----------------------------------------------------------------------
// Border cell standing for a goal outside the 9x9 field of view (statetransformer.py:47-66).  The
// reference decides with atan2 against +-pi/4, +-3pi/4 and np.round (half to even); for integer
// offsets that is exactly: "vertical" branch iff |dy| >= |dx| and dy != 0, and round-half-even of
// 5*dx/|dy| (tests/test_rollout_oracle.py::test_integer_projected_goal_rule_exhaustive compares the
// two rules on every offset with |dx|, |dy| <= 150; this kernel itself is checked on full grids of
// offsets by tests/test_rollout_oracle.py and tests/test_gpu_rollout.py).
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

constexpr int kObsAgentsPerWg = 16;
constexpr size_t kObsStageBytes = kObsAgentsPerWg * 363 * sizeof(float);   // LDS stage of a workgroup's output rows

// Observation builder, in two parts so that the static half can run early (in the fused kernels the
// waves that do not move stage it while wave 0 runs the collision shielding):
//   observe_stage   cell[H*W] bytes of LDS <- the episode's map (bit 0 = obstacle, fetched once,
//                   coalesced), goal_l[2N] ints of LDS <- goals; threads [t0, t0+nt)
//   observe_prep    after a barrier: bit 1 of cell = an agent stands there; per agent the goal's cell in
//                   the 11x11 channel (in view, or projected onto the border ring) is computed ONCE;
//   observe_rows    after another barrier: one thread per (agent, channel, row), 11 outputs sharing all
//                   their index arithmetic.  Everything in the loops comes from LDS.
__device__ __forceinline__ void observe_stage(const RolloutArgs& p, int b, unsigned char* cell, int* goal_l,
                                              int t0, int nt) {
    if (t0 < 0) return;
    const int HW = p.H * p.W;
    const unsigned char* grid = p.grid + (p.grid_batched ? (size_t)b * HW : 0);
    const int* goal = p.goal + (size_t)b * p.N * 2;
    if ((HW & 15) == 0 && ((reinterpret_cast<size_t>(grid) | reinterpret_cast<size_t>(cell)) & 15) == 0) {
        // 16 cells per load (a 100 x 100 map is 40 byte loads per thread otherwise); "cell != 0" per byte: OR the
        // eight bits of every byte into its bit 0 (the shifts only carry foreign bits into bits >= 4)
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        const v4u* g4 = reinterpret_cast<const v4u*>(grid);
        v4u* c4 = reinterpret_cast<v4u*>(cell);
        for (int i = t0; i < (HW >> 4); i += nt) {
            v4u w = g4[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                unsigned t = w[k] | (w[k] >> 4);
                t |= t >> 2;
                t |= t >> 1;
                w[k] = t & 0x01010101u;
            }
            c4[i] = w;
        }
    } else if ((HW & 3) == 0 && ((reinterpret_cast<size_t>(grid) | reinterpret_cast<size_t>(cell)) & 3) == 0) {
        const unsigned* g1 = reinterpret_cast<const unsigned*>(grid);       // (50 x 50: episodes 4-byte aligned)
        unsigned* c1 = reinterpret_cast<unsigned*>(cell);
        for (int i = t0; i < (HW >> 2); i += nt) {
            unsigned t = g1[i];
            t |= t >> 4;
            t |= t >> 2;
            t |= t >> 1;
            c1[i] = t & 0x01010101u;
        }
    } else {
        for (int i = t0; i < HW; i += nt) cell[i] = grid[i] ? 1 : 0;
    }
    for (int i = t0; i < 2 * p.N; i += nt) goal_l[i] = goal[i];
}

// agents' cells (bit 1 of cell) and, per agent, the goal's cell in channel 1 (in view, or projected onto
// the border ring) -- computed ONCE per agent; threads [t0, t0 + nt)
__device__ __forceinline__ void observe_prep(const RolloutArgs& p, const int* pos, unsigned char* cell,
                                             int* goal_l, int t0, int nt) {
    if (t0 < 0) return;
    for (int n = t0; n < p.N; n += nt) {
        const int cx = pos[2 * n], cy = pos[2 * n + 1];
        cell[cx * p.W + cy] |= 2;                        // agents stand on distinct cells
        const int dx = goal_l[2 * n] - cx, dy = goal_l[2 * n + 1] - cy;
        int px, py;
        if (dx >= -4 && dx <= 4 && dy >= -4 && dy <= 4) { px = dx + 5; py = dy + 5; }
        else projected_goal(dx, dy, px, py);
        goal_l[2 * n] = px; goal_l[2 * n + 1] = py;      // from here on: the goal's cell in channel 1
    }
}

// one thread per (agent, channel, row): 11 outputs sharing all their index arithmetic; LDS reads only.
// stage (optional): (n1 - n0) * 363 floats of LDS that receive the rows instead of p.obs -- a thread's 11 floats
// are 44 bytes apart from its neighbour's, so written straight to memory every store instruction scatters 4-byte
// pieces; observe_flush() then moves the block out with consecutive lanes on consecutive floats.
__device__ __forceinline__ void observe_rows(const RolloutArgs& p, int b, const int* pos, int n0, int n1,
                                             const unsigned char* cell, const int* goal_l, int tid, int nt,
                                             float* stage = nullptr) {
    float* out = stage ? stage : p.obs + ((size_t)b * p.N + n0) * 363;
    const int rows = (n1 - n0) * 33;                     // (agent, channel, row i) triples
    for (int row = tid; row < rows; row += nt) {
        const int na = row / 33, rr = row - na * 33;
        const int ch = rr / 11, i = rr - ch * 11;
        const int n = n0 + na;
        float* o = out + na * 363 + ch * 121 + i * 11;
        if (ch == 1) {
            const int px = goal_l[2 * n], py = goal_l[2 * n + 1];
#pragma unroll
            for (int j = 0; j < 11; ++j) o[j] = (i == px && j == py) ? 1.f : 0.f;
        } else {
            const int x = pos[2 * n] + i - 5, y0 = pos[2 * n + 1] - 5;
            const bool rowin = i >= 1 && i <= 9 && x >= 0 && x < p.H;
            const bool border = i < 1 || i > 9;          // rows 0 and 10 of the 11x11 frame stay 0
            const unsigned char* crow = cell + (rowin ? x : 0) * p.W;
            o[0] = 0.f; o[10] = 0.f;                     // columns 0 and 10 of the frame
#pragma unroll
            for (int j = 1; j <= 9; ++j) {
                const int y = y0 + j;
                const bool inside = rowin && y >= 0 && y < p.W;
                const int c = inside ? (int)crow[y] : 1; // outside the map = obstacle
                const float v = ch == 0 ? (float)(c & 1) : (inside ? (float)(c >> 1) : 0.f);
                o[j] = border ? 0.f : v;
            }
        }
    }
}

__device__ __forceinline__ void observe_flush(const RolloutArgs& p, int b, int n0, int n1, const float* stage,
                                              int tid, int nt) {
    float* out = p.obs + ((size_t)b * p.N + n0) * 363;
    for (int i = tid; i < (n1 - n0) * 363; i += nt) out[i] = stage[i];
}

// grid = (ceil(N / 16), B): a workgroup builds the episode's occupancy grid in LDS and writes the
// observations of 16 agents.
__global__ __launch_bounds__(256) void rollout_observe_kernel(const RolloutArgs p, int staged) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    const int b = blockIdx.y;
    const int n0 = blockIdx.x * kObsAgentsPerWg;
    int* goal_l = reinterpret_cast<int*>(gnnpp_smem);                       // [2 kMaxAgents]
    unsigned char* cell = reinterpret_cast<unsigned char*>(goal_l + 2 * kMaxAgents);
    const int* pos = p.pos + (size_t)b * p.N * 2;
    observe_stage(p, b, cell, goal_l, threadIdx.x, 256);
    __syncthreads();
    observe_prep(p, pos, cell, goal_l, threadIdx.x, 256);
    __syncthreads();
    const size_t occ_bytes = ((size_t)p.H * p.W + 15) & ~(size_t)15;
    float* stage = staged ? reinterpret_cast<float*>(cell + occ_bytes) : nullptr;   // (maps too large for it: direct)
    const int n1 = min(p.N, n0 + kObsAgentsPerWg);
    observe_rows(p, b, pos, n0, n1, cell, goal_l, threadIdx.x, 256, stage);
    if (stage) {                                         // (workgroup-uniform)
        __syncthreads();
        observe_flush(p, b, n0, n1, stage, threadIdx.x, 256);
    }
}

// ---- communication GSO ---------------------------------------------------------------------------
// A = (pdist < R) with zero diagonal; at step 0 R is divided by 1.1 once and multiplied by 1.1
// until the graph is connected; S = D^-1/2 A D^-1/2 in fp64 (isolated nodes -> 0), rounded to fp32
// (what `S.float()` does to the simulator's float64 GSO).  Connectivity by graph search on
// adjacency bit masks -- the same boolean as the reference's Laplacian-spectrum test.
// Largest integer d2 with sqrt((double)d2) < R, i.e. the exact integer form of the reference's
// `distance < R` test on integer positions (-1 if none).  Evaluated with the same correctly rounded
// fp64 sqrt the reference's pdist uses, once per radius instead of once per agent pair.
__device__ __forceinline__ long long dist2_threshold(double R) {
    if (!(R > 0.0)) return -1;
    // start at ceil(fl(R*R)) >= the answer (the answer is < R^2 <= fl(R*R) (1 + 2^-53)) and walk down:
    // two square roots in the common case instead of four
    long long t = (long long)ceil(R * R);
    while (t >= 0 && !(sqrt((double)t) < R)) --t;
    return t;
}

constexpr int kGsoSmemBytes = 2 * kMaxAgents * 8 + kMaxAgents * 8 + 32;   // adj [N][2] | inv [N] | radius, flag

// Communication graph of one episode on ONE wavefront (lane l: agents l and l + 64), registers only:
//   * adjacency row i = ballot over the lanes of "dist(i, me) < R" with agent i's cell broadcast by
//     readlane (N ballots instead of N^2 / 64 lane-loops; the relation is symmetric);
//   * connectivity by LEVEL-synchronous search from node 0: a node joins when its row meets the
//     frontier -- one ballot per level (graph diameter) instead of one LDS round trip per node;
//   * at step 0 (grow) the radius is divided by 1.1 once and multiplied by 1.1 until connected, the
//     same fp64 sequence as the host loop (multirobotsim_dcenlocal.py:342-348).
// Leaves adj [N][2] (128-bit rows), inv [N] (fp64 D^-1/2, 0 for isolated nodes), the flag and the final
// radius in LDS for gso_store.
__device__ __forceinline__ void gso_wave0(const RolloutArgs& p, const int* pos, bool grow, double r,
                                          char* smem, int lane) {
    unsigned long long* adj = reinterpret_cast<unsigned long long*>(smem);          // [N][2]
    double* inv = reinterpret_cast<double*>(adj + 2 * kMaxAgents);                  // [N]
    double* shared_r = inv + kMaxAgents;                                            // [1]
    int* shared_flag = reinterpret_cast<int*>(shared_r + 1);                        // [1]
    const int N = p.N;
    const bool two = N > 64;
    int px[2], py[2];
    bool live[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = lane + 64 * h;
        live[h] = n < N;
        px[h] = live[h] ? pos[2 * n] : -(1 << 20);            // dead lanes: far away from everything
        py[h] = live[h] ? pos[2 * n + 1] : -(1 << 20);
    }
    unsigned long long a0[2] = {0ull, 0ull}, a1[2] = {0ull, 0ull};                 // my agents' rows
    if (grow) r = r / 1.1;
    int connected = 0;
    for (;;) {
        if (grow) r = r * 1.1;
        const long long T = dist2_threshold(r);
        const int Ti = T > 0x7fffffffLL ? 0x7fffffff : (int)T;       // d2 < 2^18 on a 256 x 256 map
        for (int i = 0; i < N; ++i) {
            const int bx = lane_get(px, i), by = lane_get(py, i);
            bool e[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int dx = px[h] - bx, dy = py[h] - by;
                e[h] = live[h] && lane + 64 * h != i && dx * dx + dy * dy <= Ti;
            }
            const MaskPair row = ballot2(e[0], two && e[1]);
            if (lane == (i & 63)) {
                a0[i >> 6] = row.lo;
                a1[i >> 6] = row.hi;
            }
        }
        MaskPair R = {1ull, 0ull}, F = R;                              // reached set, frontier
        while (F.lo | F.hi) {
            const MaskPair nb = ballot2(live[0] && ((a0[0] & F.lo) | (a1[0] & F.hi)) != 0ull,
                                        two && live[1] && ((a0[1] & F.lo) | (a1[1] & F.hi)) != 0ull);
            F.lo = nb.lo & ~R.lo; F.hi = nb.hi & ~R.hi;
            R.lo |= nb.lo; R.hi |= nb.hi;
        }
        connected = __popcll(R.lo) + __popcll(R.hi) == N;
        if (connected || !grow) break;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = lane + 64 * h;
        if (live[h]) {
            const int deg = __popcll(a0[h]) + __popcll(a1[h]);
            adj[2 * n] = a0[h]; adj[2 * n + 1] = a1[h];
            inv[n] = deg ? sqrt(1.0 / (double)deg) : 0.0;
        }
    }
    if (lane == 0) { *shared_r = r; *shared_flag = connected; }
}

// The same graph built by ALL waves of the workgroup (no radius growth): the N adjacency rows are independent,
// so wave w computes rows w, w + nw, ... (every wave holds all positions: lane l = agents l and l + 64) and puts
// them into adj; after a barrier wave 0 reads its agents' rows back (the relation is symmetric: row n is
// also agent n's neighbour mask), runs the level-synchronous search and writes inv / flag / radius.  On one
// wave the row loop is ~30 dependent instructions x N: 17 us at N = 100; split over 16 waves it is the search
// that remains.  Contains TWO workgroup barriers: every thread of the workgroup must call it.
__device__ __forceinline__ void gso_all_waves(const RolloutArgs& p, const int* pos, double r, char* smem, int tid,
                                              int nt) {
    unsigned long long* adj = reinterpret_cast<unsigned long long*>(smem);          // [N][2]
    double* inv = reinterpret_cast<double*>(adj + 2 * kMaxAgents);                  // [N]
    double* shared_r = inv + kMaxAgents;                                            // [1]
    int* shared_flag = reinterpret_cast<int*>(shared_r + 1);                        // [1]
    const int N = p.N, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const bool two = N > 64;
    int px[2], py[2];
    bool live[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = lane + 64 * h;
        live[h] = n < N;
        px[h] = live[h] ? pos[2 * n] : -(1 << 20);
        py[h] = live[h] ? pos[2 * n + 1] : -(1 << 20);
    }
    const long long T = dist2_threshold(r);
    const int Ti = T > 0x7fffffffLL ? 0x7fffffff : (int)T;
    for (int i = wave; i < N; i += nw) {
        const int bx = lane_get(px, i), by = lane_get(py, i);
        bool e[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int dx = px[h] - bx, dy = py[h] - by;
            e[h] = live[h] && lane + 64 * h != i && dx * dx + dy * dy <= Ti;
        }
        const MaskPair row = ballot2(e[0], two && e[1]);
        if (lane == 0) { adj[2 * i] = row.lo; adj[2 * i + 1] = row.hi; }
    }
    __syncthreads();
    if (wave == 0) {
        unsigned long long a0[2] = {0ull, 0ull}, a1[2] = {0ull, 0ull};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = lane + 64 * h;
            if (live[h]) { a0[h] = adj[2 * n]; a1[h] = adj[2 * n + 1]; }
        }
        MaskPair R = {1ull, 0ull}, F = R;                              // reached set, frontier
        while (F.lo | F.hi) {
            const MaskPair nb = ballot2(live[0] && ((a0[0] & F.lo) | (a1[0] & F.hi)) != 0ull,
                                        two && live[1] && ((a0[1] & F.lo) | (a1[1] & F.hi)) != 0ull);
            F.lo = nb.lo & ~R.lo; F.hi = nb.hi & ~R.hi;
            R.lo |= nb.lo; R.hi |= nb.hi;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = lane + 64 * h;
            if (live[h]) {
                const int deg = __popcll(a0[h]) + __popcll(a1[h]);
                inv[n] = deg ? sqrt(1.0 / (double)deg) : 0.0;
            }
        }
        if (lane == 0) { *shared_r = r; *shared_flag = __popcll(R.lo) + __popcll(R.hi) == N; }
    }
    __syncthreads();
}

// S = float(D^-1/2 A D^-1/2) to HBM by all threads, radius / connected by thread 0 (after a barrier)
__device__ __forceinline__ void gso_store(const RolloutArgs& p, int b, const char* smem, int tid, int nt) {
    const unsigned long long* adj = reinterpret_cast<const unsigned long long*>(smem);
    const double* inv = reinterpret_cast<const double*>(adj + 2 * kMaxAgents);
    const double* shared_r = inv + kMaxAgents;
    const int* shared_flag = reinterpret_cast<const int*>(shared_r + 1);
    const int N = p.N;
    float* S = p.S + (size_t)b * N * N;
    for (int i = tid / 16; i < N; i += nt / 16) {               // nt/16 rows in flight, 16 lanes per row
        const unsigned long long w0 = adj[2 * i], w1 = adj[2 * i + 1];
        const double ii = inv[i];
        for (int j = tid & 15; j < N; j += 16) {
            const bool on = j < 64 ? (w0 >> j) & 1ull : (w1 >> (j - 64)) & 1ull;
            S[i * N + j] = on ? (float)(ii * inv[j]) : 0.f;
        }
    }
    if (tid == 0) {
        p.radius[b] = *shared_r;
        if (p.connected) p.connected[b] = *shared_flag;
    }
}

__device__ __forceinline__ void gso_body(const RolloutArgs& p, int b, const int* pos, bool grow,
                                         char* smem, int tid, int nt) {
    if (grow) {                                          // step 0: the radius search is a sequential loop
        if (tid < 64) gso_wave0(p, pos, true, p.radius[b], smem, tid);
        __syncthreads();
    } else {
        gso_all_waves(p, pos, p.radius[b], smem, tid, nt);
    }
    gso_store(p, b, smem, tid, nt);
}

// The simulator work between two policy forwards of a rollout (move -> GSO -> observations of the new
// positions) inside one workgroup, two barriers in all:
//   wave 0 moves                              | the other waves fetch the episode's map and goals
//   ------------------------------------------- barrier
//   wave 0 builds the communication graph     | the other waves mark the agents' cells, goal cells
//   ------------------------------------------- barrier
//   everybody stores S, then the observation rows
// red [4 kMaxAgents] + spos [2 kMaxAgents] + goal_l [2 kMaxAgents] ints, gso_smem [kGsoSmemBytes],
// occ [H*W] bytes of LDS.
__device__ void move_body(const RolloutArgs& p, int b, int lane, int* red, int* spos, unsigned* cellcnt = nullptr,
                          const float* lds_logits = nullptr);
__device__ __forceinline__ void sim_tail(const RolloutArgs& p, int b, int* spos, int* red, int* goal_l,
                                         char* gso_smem, unsigned char* occ, int tid, int nt,
                                         unsigned* cellcnt = nullptr, const float* lds_logits = nullptr,
                                         float* obs_stage = nullptr) {
    // obs_stage: nullptr, or N * 363 floats of LDS: the observation rows go there first and leave with consecutive
    // lanes on consecutive floats (see observe_rows); one more barrier
    const double radius = p.radius[b];                   // (in flight while wave 0 moves)
    if (tid < 64) move_body(p, b, tid, red, spos, cellcnt, lds_logits);
    else observe_stage(p, b, occ, goal_l, tid - 64, nt - 64);
    __syncthreads();
    GNNPP_STAMP(b, 7, tid == 0);
    if (tid < 64) gso_wave0(p, spos, false, radius, gso_smem, tid);
    else observe_prep(p, spos, occ, goal_l, tid - 64, nt - 64);
    __syncthreads();
    GNNPP_STAMP(b, 8, tid == 0);
    gso_store(p, b, gso_smem, tid, nt);
    observe_rows(p, b, spos, 0, p.N, occ, goal_l, tid, nt, obs_stage);
    if (obs_stage) {                                     // (workgroup-uniform)
        __syncthreads();
        observe_flush(p, b, 0, p.N, obs_stage, tid, nt);
    }
    GNNPP_STAMP(b, 9, tid == 0);
}

__global__ __launch_bounds__(256) void rollout_gso_kernel(const RolloutArgs p) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    const int b = blockIdx.x;
    gso_body(p, b, p.pos + (size_t)b * p.N * 2, p.grow != 0, gnnpp_smem, threadIdx.x, 256);
}

// ---- move + collision shielding -------------------------------------------------------------------
// One wavefront per episode; lane l holds agents l and l + 64 in registers (N <= 128).  The
// reference's python is sequential, but only its outer `for i in range(N)` loops carry a true
// dependence; everything inside them is a search or a set update that a ballot does in one step:
//   * list_pos.count(pos) > 1            -> popcount(ballot(lpos == pos_i))
//   * collided = [j : allagents_pos[j] == pos] (ascending) -> ballot mask; random.choice -> k-th bit
//   * `for name in collided: if last action is STOP: everybody in collided stops
//                            elif name != chosen: it stops`
//     == "if ANY collided agent already stands still, all of them (the chosen one too) stop,
//        otherwise all but the chosen one stop"  (an agent's own flag can only flip through the
//        stop-everybody branch, which is idempotent) -- checked against the simulator's traces;
//   * list_nextpos.index(cur_i)          -> ffs(ballot(snapshot == cur_i)).
__device__ __forceinline__ unsigned hash_u32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// random.choice(collided_agents) of the reference (:489): index into the (ascending) collided list.
// tie_mode 3 reproduces CPython's random.choice on a Mersenne-Twister stream the host hands over as raw
// 32-bit outputs: _randbelow(n) = { k = n.bit_length(); do r = getrandbits(k) while r >= n }, and
// getrandbits(k <= 32) = genrand_uint32() >> (32 - k).  One word per draw, cursor kept per episode.
__device__ __forceinline__ int choose_mover(const RolloutArgs& p, int b, int ncol, int& calls) {
    int k = 0;
    if (p.tie_mode == 3) {
        const int bits = 32 - __builtin_clz((unsigned)ncol);
        int cur = p.rng_cursor[b];                          // (one wave per episode: uniform, no race)
        for (;;) {
            const unsigned w = cur < p.rng_max ? p.rng_words[(size_t)b * p.rng_max + cur] : 0u;
            ++cur;
            k = (int)(w >> (32 - bits));
            if (k < ncol) break;
        }
        __builtin_amdgcn_wave_barrier();
        if ((threadIdx.x & 63) == 0) p.rng_cursor[b] = cur;
        __builtin_amdgcn_wave_barrier();
        ++calls;
        return k;
    }
    if (p.tie_mode == 1) {
        k = (int)(hash_u32(p.seed ^ hash_u32((unsigned)b * 0x9E3779B9u +
                                             (unsigned)p.currentstep * 0x85EBCA6Bu + (unsigned)calls)) %
                  (unsigned)ncol);
    } else if (p.tie_mode == 2) {
        const int c = calls < p.max_choices ? (int)p.choices[(size_t)b * p.max_choices + calls] : 0;
        k = c < ncol ? c : 0;
    }
    ++calls;
    return k;
}

// index of the k-th set bit (k < popcount) of a 128-bit mask
__device__ __forceinline__ int kth_set_bit(MaskPair m, int k) {
    const int nlo = __popcll(m.lo);
    unsigned long long w = m.lo;
    int base = 0;
    if (k >= nlo) { w = m.hi; k -= nlo; base = 64; }
    for (int i = 0; i < k; ++i) w &= w - 1;
    return base + __ffsll((long long)w) - 1;
}

struct AgentRegs {               // per lane: agents `lane` and `lane + 64`
    int curx[2], cury[2], nxtx[2], nxty[2], last[2];
};

// lowest set bit of a 128-bit mask at or above `from` (-1 if none)
__device__ __forceinline__ int next_set_bit(MaskPair m, int from) {
    if (from < 64) {
        const unsigned long long lo = m.lo & (~0ull << from);
        if (lo) return __ffsll((long long)lo) - 1;
        from = 64;
    }
    if (from < 128) {
        const unsigned long long hi = m.hi & (~0ull << (from - 64));
        if (hi) return 64 + __ffsll((long long)hi) - 1;
    }
    return -1;
}

// interRobotCollision (utils/multirobotsim_dcenlocal.py:462-555).  The python loops visit every agent
// i = 0..N-1 in order, but an agent only DOES something when (loop 1) its planned cell is planned by
// somebody else too, or (loop 2) it swaps cells with another agent.  One all-pairs scan (each agent's
// planned cell broadcast with readlane) marks the agents for which that can be true;
// the loops then jump from marked agent to marked agent, re-checking each with the LIVE state exactly
// as the reference does when it reaches it.  The marks are conservative supersets:
//   loop 1: list_pos only ever changes to the CURRENT cell of an agent that is stopped, so new
//           duplicates can only appear at those cells -- a ballot per stopped agent adds them;
//   loop 2: list_nextpos is a snapshot and nxt only ever changes to cur, which cannot create a swap.
// When nothing is marked the call is the no-op the reference's would be, and returns False.
// (row, col) of a cell as one comparable word; dead lanes carry distinct negative dummies, maps are at
// most 256 x 256 (the occupancy grid lives in LDS), so keys never collide
__device__ __forceinline__ int cell_key(int x, int y) { return x * 65536 + y; }

template <bool two>
__device__ __forceinline__ MaskPair ballot_eq(const int (&key)[2], int value) {
    MaskPair m;
    m.lo = __ballot(key[0] == value);
    m.hi = two ? __ballot(key[1] == value) : 0ull;
    return m;
}
template <bool two>
__device__ __forceinline__ int lane_get_t(const int (&v)[2], int agent) {
    return two ? lane_get(v, agent) : __builtin_amdgcn_readlane(v[0], agent);
}

// two = lanes carry a second agent (N > 64): a compile-time switch, so that teams of up to 64 agents
// run straight-line scalar code in the scan below
// cellcnt (optional, large teams): an all-zero LDS map of one BYTE per grid cell (packed four to a word).  The
// two candidate sets -- "somebody else plans my planned cell" and "somebody else plans the cell I stand on" --
// are then three LDS operations per agent, all agents at once (count the plans per cell, read two counts, put
// the zeros back), instead of the scan's N dependent broadcast-and-ballot steps (6 us per call at N = 100).
template <bool two>
__device__ bool inter_robot_collision_t(const RolloutArgs& p, AgentRegs& r, int b, int N, int lane,
                                        int& calls, unsigned* cellcnt) {
    int ckey[2], nkey[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        ckey[h] = cell_key(r.curx[h], r.cury[h]);
        nkey[h] = cell_key(r.nxtx[h], r.nxty[h]);
    }
    MaskPair todo = {0ull, 0ull}, todo2 = {0ull, 0ull};
    if (cellcnt) {
        int ni[2], ci[2];
        bool lv[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            lv[h] = lane + 64 * h < N;
            ni[h] = lv[h] ? r.nxtx[h] * p.W + r.nxty[h] : 0;
            ci[h] = lv[h] ? r.curx[h] * p.W + r.cury[h] : 0;
            if (lv[h]) __hip_atomic_fetch_add(cellcnt + (ni[h] >> 2), 1u << (8 * (ni[h] & 3)), __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // one wave: its LDS instructions complete in order, so every lane's increment precedes every lane's
        // read below -- as long as the compiler keeps the instruction order (a relaxed atomic orders nothing)
        __builtin_amdgcn_wave_barrier();
        bool dup[2], stood[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned cn = lv[h] ? (cellcnt[ni[h] >> 2] >> (8 * (ni[h] & 3))) & 255u : 0u;
            const unsigned cc = lv[h] ? (cellcnt[ci[h] >> 2] >> (8 * (ci[h] & 3))) & 255u : 0u;
            dup[h] = cn > 1u;                                        // another agent plans my planned cell
            stood[h] = cc > (ni[h] == ci[h] ? 1u : 0u);              // another agent plans the cell I stand on
        }
        todo = ballot2(dup[0], two && dup[1]);
        todo2 = ballot2(stood[0], two && stood[1]);
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (lv[h]) cellcnt[ni[h] >> 2] = 0u;                     // the map is all zero again
    } else {
    // ---- all-pairs scan, branch-free: agent j's planned cell is broadcast with readlane; one ballot
    // marks everybody else planning the same cell, one everybody standing on it ------------------------
    const int n_lo = two ? 64 : N;
    for (int j = 0; j < n_lo; ++j) {
        const int nj = __builtin_amdgcn_readlane(nkey[0], j);
        const MaskPair same = ballot_eq<two>(nkey, nj), stand = ballot_eq<two>(ckey, nj);
        const unsigned long long self = ~(1ull << j);
        todo.lo |= same.lo & self; todo.hi |= same.hi;
        todo2.lo |= stand.lo & self; todo2.hi |= stand.hi;
    }
    if (two) {
        for (int j = 64; j < N; ++j) {
            const int nj = __builtin_amdgcn_readlane(nkey[1], j - 64);
            const MaskPair same = ballot_eq<two>(nkey, nj), stand = ballot_eq<two>(ckey, nj);
            const unsigned long long self = ~(1ull << (j - 64));
            todo.lo |= same.lo; todo.hi |= same.hi & self;
            todo2.lo |= stand.lo; todo2.hi |= stand.hi & self;
        }
    }
    }
    if (!(todo.lo | todo.hi | todo2.lo | todo2.hi)) return false;

    bool collision = false;
    int skey[2], lkey[2];                            // allagents_pos (never updated), list_pos (updated)
#pragma unroll
    for (int h = 0; h < 2; ++h) { skey[h] = nkey[h]; lkey[h] = nkey[h]; }
    for (int i = next_set_bit(todo, 0); i >= 0; i = next_set_bit(todo, i + 1)) {
        const int pk = lane_get_t<two>(lkey, i);
        const MaskPair same = ballot_eq<two>(lkey, pk);
        if (__popcll(same.lo) + __popcll(same.hi) > 1) {
            collision = true;
            const bool in[2] = {skey[0] == pk, skey[1] == pk};
            const MaskPair col = ballot2(in[0], two && in[1]);
            const int ncol = __popcll(col.lo) + __popcll(col.hi);
            const int mover = kth_set_bit(col, choose_mover(p, b, ncol, calls));
            const MaskPair still = ballot2(in[0] && r.last[0] == 4, two && in[1] && r.last[1] == 4);
            const bool all_stop = (still.lo | still.hi) != 0;
            bool moved_back[2] = {false, false};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (in[h] && (all_stop || lane + 64 * h != mover)) {
                    moved_back[h] = lkey[h] != ckey[h];
                    r.last[h] = 4;
                    r.nxtx[h] = r.curx[h]; r.nxty[h] = r.cury[h];
                    lkey[h] = ckey[h];
                }
            }
            // the cells the stopped agents fell back to may now be claimed twice: mark their claimants
            const MaskPair back = ballot2(moved_back[0], two && moved_back[1]);
            for (int s2 = next_set_bit(back, 0); s2 >= 0; s2 = next_set_bit(back, s2 + 1)) {
                const MaskPair claim = ballot_eq<two>(lkey, lane_get_t<two>(lkey, s2));
                if (__popcll(claim.lo) + __popcll(claim.hi) > 1) {
                    todo.lo |= claim.lo;
                    todo.hi |= claim.hi;
                }
            }
        }
    }
    // position swaps (:524-553); list_nextpos is a snapshot taken here.  Candidates: agents whose
    // current cell somebody planned at entry -- plans only ever change to current cells, and two
    // agents never share a current cell, so no other agent can become part of a swap.
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        nkey[h] = cell_key(r.nxtx[h], r.nxty[h]);    // live plans
        skey[h] = nkey[h];                           // the snapshot
    }
    for (int i = next_set_bit(todo2, 0); i >= 0; i = next_set_bit(todo2, i + 1)) {
        const MaskPair hit = ballot_eq<two>(skey, lane_get_t<two>(ckey, i));
        if (hit.lo | hit.hi) {
            const int sidx = hit.lo ? __ffsll((long long)hit.lo) - 1 : 64 + __ffsll((long long)hit.hi) - 1;
            if (sidx != i && lane_get_t<two>(ckey, sidx) == lane_get_t<two>(nkey, i)) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int me = lane + 64 * h;
                    if (me == i || me == sidx) {
                        r.nxtx[h] = r.curx[h]; r.nxty[h] = r.cury[h];
                        nkey[h] = ckey[h];
                        r.last[h] = 4;
                    }
                }
                collision = true;
            }
        }
    }
    return collision;
}

__device__ bool inter_robot_collision(const RolloutArgs& p, AgentRegs& r, int b, int N, int lane,
                                      int& calls, unsigned* cellcnt) {
    return N > 64 ? inter_robot_collision_t<true>(p, r, b, N, lane, calls, cellcnt)
                  : inter_robot_collision_t<false>(p, r, b, N, lane, calls, cellcnt);
}

// One episode's move by ONE wavefront (lane = threadIdx.x & 63; no workgroup barrier inside, so it can
// run as wave 0 of a larger workgroup).  red = [4][kMaxAgents] ints of LDS (conflict test / statistics
// scratch); spos (optional) receives the positions after the move ([N][2], LDS) for the fused step kernel.
// cellcnt: nullptr, or ceil(H*W / 4) words of LDS for inter_robot_collision's cell-count map (zeroed here).
// lds_logits: nullptr, or this episode's logits [N][5] left in LDS by THIS wave (the one-launch policy kernel's
// head: no trip through memory, no workgroup barrier between head and move).
__device__ void move_body(const RolloutArgs& p, int b, int lane, int* red, int* spos, unsigned* cellcnt,
                          const float* lds_logits) {
    const int N = p.N;
    int* pos = p.pos + (size_t)b * N * 2;
    const unsigned char* grid = p.grid + (p.grid_batched ? (size_t)b * p.H * p.W : 0);
    const int* goal = p.goal + (size_t)b * N * 2;
    int* reached = p.reached + (size_t)b * N;
    int* start_step = p.start_step + (size_t)b * N;
    int* end_step = p.end_step + (size_t)b * N;
    const int step = p.currentstep, maxstep = p.maxstep[b];

    GNNPP_STAMP(b, 0, lane == 0);
    AgentRegs r;
    int key[2], rch[2], sst[2], est[2];
    bool live[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = lane + 64 * h;
        live[h] = n < N;
        key[h] = 4; rch[h] = 1; sst[h] = -1; est[h] = -1;
        r.curx[h] = r.cury[h] = r.nxtx[h] = r.nxty[h] = -1 - n;          // distinct dummies
        r.last[h] = 4;
        if (live[h]) {
            if (lds_logits || p.logits) {   // argmax of the logits == argmax of LogSoftmax, first max wins
                const float* l = lds_logits ? lds_logits + n * 5 : p.logits + ((size_t)n * p.B + b) * 5;
                int kk = 0;
                float best = l[0];
#pragma unroll
                for (int k = 1; k < 5; ++k)
                    if (l[k] > best) { best = l[k]; kk = k; }
                key[h] = kk;
            } else {
                key[h] = p.actions[(size_t)b * N + n];
            }
            r.curx[h] = pos[2 * n]; r.cury[h] = pos[2 * n + 1];
            rch[h] = reached[n]; sst[h] = start_step[n]; est[h] = end_step[n];
        }
    }
    GNNPP_STAMP(b, 1, lane == 0);
    const MaskPair not_reached = ballot2(live[0] && !rch[0], live[1] && !rch[1]);
    const bool all_reached = !(not_reached.lo | not_reached.hi);
    bool predict_collision = false, move_collision = false;
    int calls = 0;
    // The reference never calls move() again for a case whose loop has ended
    // (agents/decentralplannerlocal.py:560-605: `for step in range(maxstep)`, break after the call
    // that saw allReachGoal).  In a batch the other episodes go on, so such an episode is frozen:
    // nothing of its state changes, its flags read (allReachGoal, 0, 0).
    if ((p.done && p.done[b] != 0) || step > maxstep) {
        if (lane == 0) {
            p.flags[3 * b] = all_reached;
            p.flags[3 * b + 1] = 0;
            p.flags[3 * b + 2] = 0;
            if (p.choice_count) p.choice_count[b] = 0;
        }
        if (spos) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (live[h]) {
                    spos[2 * (lane + 64 * h)] = r.curx[h];
                    spos[2 * (lane + 64 * h) + 1] = r.cury[h];
                }
        }
        return;
    }
    if (!all_reached || step < maxstep) {
        bool bumped[2] = {false, false};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (live[h]) {
                if (key[h] != 4 && sst[h] < 0) sst[h] = step - 1;
                const int dx = key[h] == 0 ? -1 : key[h] == 2 ? 1 : 0;
                const int dy = key[h] == 1 ? -1 : key[h] == 3 ? 1 : 0;
                const int nx = r.curx[h] + dx, ny = r.cury[h] + dy;
                const bool edge = nx >= p.H || nx < 0 || ny >= p.W || ny < 0;
                if (edge || grid[nx * p.W + ny] == 1) {
                    bumped[h] = true;
                    r.last[h] = 4;
                    r.nxtx[h] = r.curx[h]; r.nxty[h] = r.cury[h];
                } else {
                    r.last[h] = key[h];
                    r.nxtx[h] = nx; r.nxty[h] = ny;
                }
            }
        }
        const MaskPair bm = ballot2(bumped[0], bumped[1]);
        predict_collision = (bm.lo | bm.hi) != 0;
        GNNPP_STAMP(b, 2, lane == 0);
        if (cellcnt) {                                       // (the allocation is a multiple of 16 bytes)
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            const v4u zero4 = {0u, 0u, 0u, 0u};
            for (int i = lane; i < (p.H * p.W + 15) / 16; i += 64) reinterpret_cast<v4u*>(cellcnt)[i] = zero4;
            __builtin_amdgcn_wave_barrier();                 // (all of the map is zero before the first count)
        }
        bool detect = inter_robot_collision(p, r, b, N, lane, calls, cellcnt);
        GNNPP_STAMP(b, 3, lane == 0);
        for (int it = 0; it < N; ++it) {
            if (!detect) break;
            detect = inter_robot_collision(p, r, b, N, lane, calls, cellcnt);
            predict_collision = true;
        }
        GNNPP_STAMP(b, 4, lane == 0);
        move_collision = inter_robot_collision(p, r, b, N, lane, calls, cellcnt);
        GNNPP_STAMP(b, 5, lane == 0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (live[h]) {
                const int n = lane + 64 * h;
                pos[2 * n] = r.nxtx[h]; pos[2 * n + 1] = r.nxty[h];
                r.curx[h] = r.nxtx[h]; r.cury[h] = r.nxty[h];
                if (r.nxtx[h] == goal[2 * n] && r.nxty[h] == goal[2 * n + 1] && !rch[h]) {
                    rch[h] = 1;
                    est[h] = step;
                }
                if (step >= maxstep && !rch[h]) {
                    est[h] = step;
                    if (sst[h] < 0) sst[h] = 0;
                }
                reached[n] = rch[h]; start_step[n] = sst[h]; end_step[n] = est[h];
            }
        }
    }
    if (all_reached || step >= maxstep) {
        // An agent that reached its goal without ever issuing a move keeps start_step = None in
        // the reference (which then raises TypeError on `end - None`); we count it from step 0.
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (live[h]) {
                const int n = lane + 64 * h;
                red[n] = est[h];
                red[kMaxAgents + n] = sst[h] < 0 ? 0 : sst[h];
            }
        }
        __builtin_amdgcn_wave_barrier();                 // one wave: its LDS writes precede lane 0's reads
        if (lane == 0) {
            int flow = 0, emax = -(1 << 30), smin = 1 << 30;
            for (int n = 0; n < N; ++n) {
                const int e = red[n], st = red[kMaxAgents + n];
                flow += e - st;
                emax = e > emax ? e : emax;
                smin = st < smin ? st : smin;
            }
            p.stats[2 * b] = emax - smin;
            p.stats[2 * b + 1] = flow;
        }
    }
    GNNPP_STAMP(b, 6, lane == 0);
    if (lane == 0) {
        p.flags[3 * b] = all_reached;
        p.flags[3 * b + 1] = move_collision;
        p.flags[3 * b + 2] = predict_collision;
        if (p.choice_count) p.choice_count[b] = calls;
        if (p.done && (all_reached || step >= maxstep)) p.done[b] = 1;   // the reference's loop breaks here
    }
    if (spos) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (live[h]) {
                spos[2 * (lane + 64 * h)] = r.curx[h];
                spos[2 * (lane + 64 * h) + 1] = r.cury[h];
            }
    }
}

// Maps of at most kCellMapMaxCells cells get the cell-count map of inter_robot_collision_t (one byte per cell of
// LDS behind the other scratch; it pays at every team size: N = 16 1.5 us per step, N = 100 18 us); the
// launchers size the allocation with the same rule.
constexpr int kCellMapMaxCells = 32 * 1024;
__host__ __device__ inline size_t cell_map_bytes(int N, int H, int W) {
    (void)N;
    return (long)H * W <= kCellMapMaxCells ? (((size_t)H * W + 15) & ~(size_t)15) : 0;
}

__global__ __launch_bounds__(64) void rollout_move_kernel(const RolloutArgs p) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    int* red = reinterpret_cast<int*>(gnnpp_smem);
    unsigned* cellcnt = cell_map_bytes(p.N, p.H, p.W) ? reinterpret_cast<unsigned*>(red + 4 * kMaxAgents) : nullptr;
    move_body(p, blockIdx.x, threadIdx.x, red, nullptr, cellcnt);
}

// Fused simulator step between two policy forwards: move (wave 0) -> communication GSO ->
// observations of the NEW positions, one workgroup per episode, positions handed over in LDS.
// Same results as the three kernels in sequence (gso never grows the radius here: that only
// happens at step 0, which runs the separate kernels).
__global__ __launch_bounds__(1024) void rollout_step_kernel(const RolloutArgs p, int staged) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    int* spos = reinterpret_cast<int*>(gnnpp_smem);                        // [N][2]
    int* red = spos + 2 * kMaxAgents;                                      // [4][kMaxAgents]
    int* goal_l = red + 4 * kMaxAgents;                                    // [2][kMaxAgents]
    char* gso_smem = reinterpret_cast<char*>(goal_l + 2 * kMaxAgents);
    unsigned char* occ = reinterpret_cast<unsigned char*>(gso_smem + kGsoSmemBytes);
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const size_t occ_bytes = ((size_t)p.H * p.W + 15) & ~(size_t)15;
    const size_t map_bytes = cell_map_bytes(p.N, p.H, p.W);
    unsigned* cellcnt = map_bytes ? reinterpret_cast<unsigned*>(occ + occ_bytes) : nullptr;
    float* stage = staged ? reinterpret_cast<float*>(occ + occ_bytes + map_bytes) : nullptr;
    sim_tail(p, b, spos, red, goal_l, gso_smem, occ, tid, nt, cellcnt, nullptr, stage);
}

// Communication GSO and observations of the CURRENT positions in one launch, for teams too large for one workgroup
// per episode (rollout_step_kernel): both depend only on the positions, so their workgroups run side by side --
// per episode `groups` observation workgroups (16 agents each) and one GSO workgroup.  The two kernels in
// sequence leave most of the chip idle twice (B graph workgroups, then B * groups observation workgroups).
__global__ __launch_bounds__(256) void rollout_gso_observe_kernel(const RolloutArgs p, int groups, int staged) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    const int per = groups + 1;
    const int b = blockIdx.x / per, r = blockIdx.x - b * per;
    const int* pos = p.pos + (size_t)b * p.N * 2;
    if (r == groups) {                                   // (workgroup-uniform)
        gso_body(p, b, pos, false, gnnpp_smem, threadIdx.x, 256);
        return;
    }
    const int n0 = r * kObsAgentsPerWg;
    int* goal_l = reinterpret_cast<int*>(gnnpp_smem);                       // [2 kMaxAgents]
    unsigned char* cell = reinterpret_cast<unsigned char*>(goal_l + 2 * kMaxAgents);
    observe_stage(p, b, cell, goal_l, threadIdx.x, 256);
    __syncthreads();
    observe_prep(p, pos, cell, goal_l, threadIdx.x, 256);
    __syncthreads();
    const size_t occ_bytes = ((size_t)p.H * p.W + 15) & ~(size_t)15;
    float* stage = staged ? reinterpret_cast<float*>(cell + occ_bytes) : nullptr;   // (maps too large for it: direct)
    const int n1 = min(p.N, n0 + kObsAgentsPerWg);
    observe_rows(p, b, pos, n0, n1, cell, goal_l, threadIdx.x, 256, stage);
    if (stage) {                                         // (workgroup-uniform)
        __syncthreads();
        observe_flush(p, b, n0, n1, stage, threadIdx.x, 256);
    }
}

// ---- launchers -----------------------------------------------------------------------------------
int rollout_gso_observe_launch(const RolloutArgs& a, hipStream_t st) {
    const size_t occ = ((size_t)a.H * a.W + 15) & ~(size_t)15;
    if (occ > 64 * 1024) return -2;
    size_t smem = occ + 2 * kMaxAgents * sizeof(int);
    const int staged = smem + kObsStageBytes <= 64 * 1024;      // the output stage, while the default LDS limit allows
    if (staged) smem += kObsStageBytes;
    if (smem < (size_t)kGsoSmemBytes) smem = kGsoSmemBytes;
    const int groups = (a.N + kObsAgentsPerWg - 1) / kObsAgentsPerWg;
    hipLaunchKernelGGL(rollout_gso_observe_kernel, dim3(a.B * (groups + 1)), dim3(256), smem, st, a, groups, staged);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int rollout_observe_launch(const RolloutArgs& a, hipStream_t st) {
    const size_t occ = ((size_t)a.H * a.W + 15) & ~(size_t)15;
    if (occ > 64 * 1024) return -2;
    size_t smem = occ + 2 * kMaxAgents * sizeof(int);
    const int staged = smem + kObsStageBytes <= 64 * 1024;
    if (staged) smem += kObsStageBytes;
    hipLaunchKernelGGL(rollout_observe_kernel,
                       dim3((a.N + kObsAgentsPerWg - 1) / kObsAgentsPerWg, a.B), dim3(256), smem, st,
                       a, staged);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int rollout_gso_launch(const RolloutArgs& a, hipStream_t st) {
    const size_t smem = kGsoSmemBytes;
    hipLaunchKernelGGL(rollout_gso_kernel, dim3(a.B), dim3(256), smem, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int rollout_move_launch(const RolloutArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(rollout_move_kernel, dim3(a.B), dim3(64),
                       4 * kMaxAgents * sizeof(int) + cell_map_bytes(a.N, a.H, a.W), st, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int rollout_step_launch(const RolloutArgs& a, hipStream_t st) {
    const size_t occ = ((size_t)a.H * a.W + 15) & ~(size_t)15;
    if (occ > 64 * 1024) return -2;
    size_t smem = 8 * kMaxAgents * sizeof(int) + kGsoSmemBytes + occ + cell_map_bytes(a.N, a.H, a.W);
    const size_t stage_bytes = (size_t)a.N * 363 * sizeof(float);
    const int staged = smem + stage_bytes <= 64 * 1024;  // the observations' output stage, while the LDS limit allows
    if (staged) smem += stage_bytes;
    const int nt = a.N > 32 ? 1024 : 256;               // enough threads for N * 363 observation cells
    hipLaunchKernelGGL(rollout_step_kernel, dim3(a.B), dim3(nt), smem, st, a, staged);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace gnnpp
