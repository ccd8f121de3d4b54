// The rest of the optimisation step of BASELINE config 4 around the encoder and the graph filter
// (agents/decentralplannerlocal.py:287-317: forward, CrossEntropy per agent, loss.backward(), optimizer.step()):
//
//   gemm_kmajor_kernel    weight gradients that are "tall contraction, small output" GEMMs: the graph filter's
//                         dh[f,e,k,g] = sum over (b,n) dy[f,(b,n)] z_k[(b,n),g] (graphML.py:2345-2352 run
//                         backwards) and the 128x128 compress layer's dW = dY^T X.  Output 128 x 128, contraction
//                         640..5120: a library GEMM puts ONE 128x128 macro tile = one workgroup on it; here the
//                         contraction is split over workgroups (fp32 MFMA 16x16x4, partials summed in order).
//   policy_loss_kernel    the loss of the training loop in one launch, forward and backward at once:
//                         mean over agents of CrossEntropy(predict[n], argmax(target[:, n])) and d loss / d logits.
//   adam_kernel           torch.optim.Adam's update (L2 weight decay folded into the gradient, bias correction)
//                         for up to 32 parameter tensors per launch; the step counter lives on the device so that
//                         the launch can sit in a HIP graph.
// All reductions have a fixed association (no atomics): the step is deterministic.
#include "gnnpp_common.h"

namespace gnnpp {

// ---- C[b][m][n] = sum_k A_b(m,k) * B_b(k,n), several independent products per launch ------------------------------
// A_b(m,k) at A + b*a_sb + m*a_sm + k*a_sk;  B_b(k,n) at B + b*b_sb + k*b_sk + n;  C_b(m,n) at C + b*c_sb + m*c_sm + n.
// One launch serves up to kGemmMax products (a Linear's dx, dW and db are three): the 1-D grid is the
// concatenation of every product's (column tiles x row tiles x batch*ksplit) workgroups; a workgroup finds its
// product by its index.  block = 256: wave w owns rows [m0 + 16w, +16) x 64 columns (four 16x16 accumulators) over
// the K range of its split.  Lane (i = lane & 15, q = lane >> 4): A value (row i, k = 4s + q), B values (k = 4s + q,
// column 16t + i).  Eight k-steps of operands (8 + 32 loads) are in flight before their 32 MFMAs.
// out: ksplit == 1 -> C directly; else the partial of (split, b) -> ws + ws_off + ((split*batch + b)*M + m)*N + n,
// summed in split order by gemm_reduce_kernel (one launch for all products that were split).
constexpr int kGemmMax = 8;
struct GemmOne {
    const float* A; long a_sb, a_sm, a_sk;
    const float* B; long b_sb, b_sk;
    float* C; long c_sb, c_sm;
    const float* mask;                // optional, addressed like C: C(m,n) is written as 0 where mask(m,n) <= 0 (the
                                      // ReLU backward of the layer below, folded into the product that feeds it)
    int batch, M, N, K, ksplit, kper, nx, ny;
    long ws_off;
};
struct GemmTable {
    GemmOne g[kGemmMax];
    int first[kGemmMax + 1];          // first workgroup of product i (multiply kernel)
    int rfirst[kGemmMax + 1];         // first workgroup of product i (reduce kernel; empty range when ksplit == 1)
    int count;
};

__global__ __launch_bounds__(256) void gemm_kmajor_kernel(const GemmTable tb, float* __restrict__ ws) {
    int gi = 0;
    while (gi + 1 < tb.count && (int)blockIdx.x >= tb.first[gi + 1]) ++gi;   // (scalar: at most 7 steps)
    const GemmOne& g = tb.g[gi];
    const int local = (int)blockIdx.x - tb.first[gi];
    const int bx = local % g.nx, by = (local / g.nx) % g.ny, bz = local / (g.nx * g.ny);
    const int M = g.M, N = g.N, K = g.K, batch = g.batch;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    const int b = bz % batch, split = bz / batch;
    const int m = by * 64 + wave * 16 + i16;
    const int n0 = bx * 64;
    const bool mv = m < M;
    const long a_sk = g.a_sk, b_sk = g.b_sk;
    const float* a = g.A + b * g.a_sb + (long)(mv ? m : 0) * g.a_sm;
    const float* bp[4];
    bool nv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int n = n0 + 16 * t + i16;
        nv[t] = n < N;
        bp[t] = g.B + b * g.b_sb + (nv[t] ? n : 0);
    }
    v4f acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = vzero();
    const int k0 = split * g.kper, k1 = min(K, k0 + g.kper);
    constexpr int U = 8;          // 32 contraction indices = a whole default K slice in ONE round of loads (r05: U = 4,
                                  // two dependent rounds of ~2 us each in a kernel of 11 us)
    for (int ks = k0; ks < k1; ks += 4 * U) {
        float av[U], bv[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = ks + 4 * u + q;
            const bool kv = k < k1;
            const int kc = kv ? k : k0;
            const float a0 = a[(long)kc * a_sk];
            av[u] = kv && mv ? a0 : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float b0 = bp[t][(long)kc * b_sk];
                bv[u][t] = kv && nv[t] ? b0 : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma16(av[u], bv[u][t], acc[t]);
    }
    // D register r of lane l: D[i = 4 q + r][j = l & 15]
    const int mr = by * 64 + wave * 16 + 4 * q;
    const bool direct = g.ksplit == 1;
    float* o = direct ? g.C + b * g.c_sb : ws + g.ws_off + ((long)split * batch + b) * M * N;
    const long o_sm = direct ? g.c_sm : (long)N;
    const float* mk = direct && g.mask ? g.mask + b * g.c_sb : nullptr;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int n = n0 + 16 * t + i16;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (mr + r < M && n < N) {
                float v = acc[t][r];
                if (mk && !(mk[(long)(mr + r) * o_sm + n] > 0.f)) v = 0.f;
                o[(long)(mr + r) * o_sm + n] = v;
            }
    }
}

// C_b(m,n) = sum over the splits of the partials: 64 outputs per workgroup, four threads per output -- thread (sub, e)
// adds the splits sub, sub + 4, .. of output e in order, then the four sums are added in sub order: a fixed
// association (deterministic), every load of a wave a contiguous 256-byte run, four times the loads in flight of
// the one-thread-per-output form (r05: 6.9 us per launch for 4 MB of partials).  Products concatenated.
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const GemmTable tb, const float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    float* red = reinterpret_cast<float*>(gnnpp_smem);                    // [256]
    int gi = 0;
    while (gi + 1 < tb.count && (int)blockIdx.x >= tb.rfirst[gi + 1]) ++gi;
    const GemmOne& g = tb.g[gi];
    const long total = (long)g.batch * g.M * g.N;
    const int e = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const long i = (long)((int)blockIdx.x - tb.rfirst[gi]) * 64 + e;
    const float* part = ws + g.ws_off;
    float s = 0.f;
    if (i < total) {
#pragma unroll 4
        for (int k = sub; k < g.ksplit; k += 4) s += part[(long)k * total + i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (sub != 0 || i >= total) return;
    s = (red[e] + red[64 + e]) + (red[128 + e] + red[192 + e]);
    const int n = (int)(i % g.N);
    const long bm = i / g.N;
    const int m = (int)(bm % g.M), b = (int)(bm / g.M);
    const long at = b * g.c_sb + (long)m * g.c_sm + n;
    if (g.mask && !(g.mask[at] > 0.f)) s = 0.f;
    g.C[at] = s;
}

struct GemmPlan { int ksplit, kper; };
inline GemmPlan gemm_plan(int batch, int M, int N, int K) {
    const int tiles = ((M + 63) / 64) * ((N + 63) / 64) * batch;
    int ks = (256 + tiles - 1) / tiles;                 // ~one workgroup per CU
    const int kmax = (K + 31) / 32;                     // at least 32 contraction steps per split
    if (ks > kmax) ks = kmax;
    if (ks < 1) ks = 1;
    GemmPlan p;
    p.kper = (((K + ks - 1) / ks) + 3) / 4 * 4;
    p.ksplit = (K + p.kper - 1) / p.kper;
    return p;
}

inline size_t gemm_workspace_floats(int batch, int M, int N, int K) {
    const GemmPlan p = gemm_plan(batch, M, N, K);
    return p.ksplit > 1 ? ((size_t)p.ksplit * batch * M * N + 3) & ~(size_t)3 : 0;
}

// fills plan / grid fields of tb.g[0..count) (A..K set by the caller), launches the multiply and, when any
// product was split, ONE reduce
inline int gemm_multi_launch(GemmTable& tb, float* ws, hipStream_t st) {
    int blocks = 0, rblocks = 0;
    long off = 0;
    for (int i = 0; i < tb.count; ++i) {
        GemmOne& g = tb.g[i];
        const GemmPlan p = gemm_plan(g.batch, g.M, g.N, g.K);
        g.ksplit = p.ksplit; g.kper = p.kper;
        g.nx = (g.N + 63) / 64; g.ny = (g.M + 63) / 64;
        g.ws_off = off;
        off += (long)gemm_workspace_floats(g.batch, g.M, g.N, g.K);
        tb.first[i] = blocks;
        blocks += g.nx * g.ny * g.batch * g.ksplit;
        tb.rfirst[i] = rblocks;
        if (g.ksplit > 1) rblocks += (int)(((long)g.batch * g.M * g.N + 63) / 64);
    }
    tb.first[tb.count] = blocks;
    tb.rfirst[tb.count] = rblocks;
    hipLaunchKernelGGL(gemm_kmajor_kernel, dim3(blocks), dim3(256), 0, st, tb, ws);
    if (rblocks > 0) hipLaunchKernelGGL(gemm_reduce_kernel, dim3(rblocks), dim3(256), 256 * sizeof(float), st, tb, ws);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---- loss of the training loop, forward and backward in one launch ------------------------------------------------
// logits [N][B][C] (agent-major, the LogitList's stacked tensor) or, with logits_bn, [B][N][C] (the same tensor
// as the train-mode forward produces it); target [B][N][C] one-hot expert actions.
//   label(n,b) = first maximum of target[b][n][:]           (torch.max(batchTarget[:, n], 1)[1])
//   loss = (1 / (N*B)) sum_{n,b} ( logsumexp(logits[n][b]) - logits[n][b][label] )
//        = mean over agents of CrossEntropyLoss(predict[n], label[:, n])   (every agent averages the same B rows)
//   dlogits[n][b][c] = (softmax(logits[n][b])[c] - [c == label]) / (N*B)
// One workgroup of 1024 threads walks the rows; the row losses are summed in doubles by a fixed tree.
__global__ __launch_bounds__(1024) void policy_loss_kernel(const float* __restrict__ logits,
                                                           const float* __restrict__ target,
                                                           float* __restrict__ loss, float* __restrict__ dlogits,
                                                           int B, int N, int C, int logits_bn) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    double* red = reinterpret_cast<double*>(gnnpp_smem);             // [1024]
    const int R = N * B;
    const float inv = 1.f / (float)R;
    double part = 0.0;
    for (int r = threadIdx.x; r < R; r += 1024) {
        const int n = r / B, b = r - n * B;
        const long row = logits_bn ? (long)b * N + n : (long)r;    // logits / dlogits [B][N][C] or [N][B][C]
        const float* lg = logits + row * C;
        const float* tg = target + ((long)b * N + n) * C;
        int label = 0;
        float tbest = tg[0], mx = lg[0];
        for (int c = 1; c < C; ++c) {
            if (tg[c] > tbest) { tbest = tg[c]; label = c; }
            mx = fmaxf(mx, lg[c]);
        }
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(lg[c] - mx);
        const float lse = mx + logf(se);
        part += (double)(lse - lg[label]);
        if (dlogits)
            for (int c = 0; c < C; ++c)
                dlogits[row * C + c] = (expf(lg[c] - lse) - (c == label ? 1.f : 0.f)) * inv;
    }
    red[threadIdx.x] = part;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(red[0] / (double)R);
}

// ---- Adam -----------------------------------------------------------------------------------------------------------
// torch.optim.Adam (amsgrad = False, maximize = False):  g' = g + wd * p;  m = m + (1 - b1)(g' - m);
// v = b2 v + (1 - b2) g'^2;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps),  t = step count.
// state (8 floats): [0] = steps taken so far, [1] = lr / (1 - b1^t), [2] = 1 / sqrt(1 - b2^t) of the LAST tick, [3] =
// arrival counter (an unsigned; zero between launches), [4], [5] = the betas and [6], [7] = the two bias-correction
// factors of the NEXT step, precomputed by the last workgroup of the tick.  A launch with tick != 0 is the first table of a step: every
// workgroup reads t = state[0] + 1 and derives the two bias-correction factors itself; the workgroup that arrives
// LAST (a ticket from the arrival counter, taken after the workgroup's own read of state[0]) stores t and the factors
// and re-arms the counter -- r05 spent a launch of one thread on that.  Further tables of the same step (tick == 0)
// read the stored factors.  adam_kernel applies the update to up to kAdamTensors tensors: workgroup w serves elements
// [1024 (w - first[i]), +1024) of tensor i, first[i] <= w < first[i+1].
constexpr int kAdamTensors = 32;
struct AdamTable {
    float* p[kAdamTensors];
    const float* g[kAdamTensors];
    float* m[kAdamTensors];
    float* v[kAdamTensors];
    long numel[kAdamTensors];
    int first[kAdamTensors + 1];
    int count;
};

__global__ __launch_bounds__(256) void adam_kernel(const AdamTable tb, float* __restrict__ state, float lr, float b1,
                                                   float b2, float eps, float wd, int tick) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    float* sh = reinterpret_cast<float*>(gnnpp_smem);                     // [4]
    if (threadIdx.x == 0) {
        if (tick) {
            const float t = state[0] + 1.f;
            // 1 / (1 - b1^t) and 1 / sqrt(1 - b2^t): left behind for THIS step by the previous step's last workgroup
            // (state[6], [7], valid for the betas in state[4], [5]); two double-precision pow() per workgroup on the
            // critical path of every workgroup cost 3 us of an 8.7 us launch
            float c1, c2;
            if (state[4] == b1 && state[5] == b2 && state[6] != 0.f) {
                c1 = state[6];
                c2 = state[7];
            } else {
                c1 = (float)(1.0 / (1.0 - pow((double)b1, (double)t)));
                c2 = (float)(1.0 / sqrt(1.0 - pow((double)b2, (double)t)));
            }
            sh[0] = lr * c1;
            sh[1] = c2;
            sh[2] = t;
        } else {
            sh[0] = state[1];
            sh[1] = state[2];
        }
    }
    __syncthreads();
    int i = 0;
    while (i + 1 < tb.count && (int)blockIdx.x >= tb.first[i + 1]) ++i;     // (scalar: at most 31 steps)
    const long base = (long)((int)blockIdx.x - tb.first[i]) * 1024;
    const float step_size = sh[0], inv_bc2 = sh[1];
    float* p = tb.p[i];
    const float* g = tb.g[i];
    float* m = tb.m[i];
    float* v = tb.v[i];
    const long n = tb.numel[i];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long e = base + u * 256 + threadIdx.x;
        if (e < n) {
            const float pe = p[e];
            const float ge = fmaf(wd, pe, g[e]);
            const float me = fmaf(1.f - b1, ge - m[e], m[e]);
            const float ve = fmaf(b2, v[e], (1.f - b2) * ge * ge);
            m[e] = me;
            v[e] = ve;
            p[e] = pe - step_size * (me / (sqrtf(ve) * inv_bc2 + eps));
        }
    }
    if (tick && threadIdx.x == 0) {
        unsigned* cnt = reinterpret_cast<unsigned*>(state + 3);
        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gridDim.x - 1) {                         // every workgroup has read the state: advance it
            const float t = sh[2];
            state[0] = t;
            state[1] = sh[0];
            state[2] = sh[1];
            state[4] = b1;
            state[5] = b2;
            state[6] = (float)(1.0 / (1.0 - pow((double)b1, (double)t + 1.0)));
            state[7] = (float)(1.0 / sqrt(1.0 - pow((double)b2, (double)t + 1.0)));
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- y = x W^T + b (+ ReLU): the forward product of the two small Linear layers of the training step -------------------
// x [R][I], W [O][I] (nn.Linear's layout: BOTH operands have the contraction index contiguous -- the shape
// gnnpp_gemm_kmajor cannot take without a transposed copy of W per step), y [R][O].  R is a few hundred rows, I = 128,
// O = 128 (compressMLP, decentralplanner.py:187-195, :289-290) or 5 (actionsMLP, :232-243, :304-315): r05 ran them as
// library GEMMs followed by an aten ReLU.  One wave owns a 16 (features) x 16 (rows) tile on v_mfma_f32_16x16x4_f32:
// lane (i = lane & 15, q = lane >> 4) supplies W[f0 + i][k] and x[r0 + i][k] for the I / 4 contraction indices
// k in [q I / 4, (q + 1) I / 4) -- the pipe's k-slot (4 s + q) is fed index q I / 4 + s for both operands, a
// permutation of the sum's terms that makes every lane's reads contiguous 16-byte loads.  D register r = feature
// f0 + 4 q + r of row r0 + i: one 16-byte store per lane.  Exact fp32, fixed order.  Needs I % 16 == 0.
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ y, int R,
                                                         int I, int O, int relu) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, q = lane >> 4;
    const int FT = (O + 15) / 16, RT = (R + 15) / 16;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= FT * RT) return;
    const int rt = tile / FT, ft = tile - rt * FT;
    const int f = ft * 16 + i16, r = rt * 16 + i16;
    const int kq = I / 4;
    const float* wr = W + (long)(f < O ? f : 0) * I + q * kq;
    const float* xr = x + (long)(r < R ? r : 0) * I + q * kq;
    const bool fv = f < O, rv = r < R;
    v4f acc = vzero();
    for (int k = 0; k < kq; k += 16) {                       // four 16-byte loads per operand in flight
        v4f a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const v4f*>(wr + k + 4 * u);
            b[u] = *reinterpret_cast<const v4f*>(xr + k + 4 * u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = mfma16(fv ? a[u][c] : 0.f, rv ? b[u][c] : 0.f, acc);
    }
    if (!rv) return;
    const int fo = ft * 16 + 4 * q;
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v = acc[c] + (bias && fo + c < O ? bias[fo + c] : 0.f);
        o[c] = relu ? fmaxf(v, 0.f) : v;
    }
    float* yr = y + (long)r * O + fo;
    if (fo + 3 < O && (O & 3) == 0) {
        v4f ov = {o[0], o[1], o[2], o[3]};
        *reinterpret_cast<v4f*>(yr) = ov;
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (fo + c < O) yr[c] = o[c];
    }
}

}  // namespace gnnpp
