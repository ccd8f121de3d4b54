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
// column 16t + i).  Four k-steps of operands (4 + 16 loads) are in flight before their 16 MFMAs.
// out: ksplit == 1 -> C directly; else the partial of (split, b) -> ws + ws_off + ((split*batch + b)*M + m)*N + n,
// summed in split order by gemm_reduce_kernel (one launch for all products that were split).
constexpr int kGemmMax = 8;
struct GemmOne {
    const float* A; long a_sb, a_sm, a_sk;
    const float* B; long b_sb, b_sk;
    float* C; long c_sb, c_sm;
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
    constexpr int U = 4;
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
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int n = n0 + 16 * t + i16;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (mr + r < M && n < N) o[(long)(mr + r) * o_sm + n] = acc[t][r];
    }
}

// C_b(m,n) = sum over splits (in order) of the partials; 256 outputs per workgroup, products concatenated
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const GemmTable tb, const float* __restrict__ ws) {
    int gi = 0;
    while (gi + 1 < tb.count && (int)blockIdx.x >= tb.rfirst[gi + 1]) ++gi;
    const GemmOne& g = tb.g[gi];
    const long total = (long)g.batch * g.M * g.N;
    const long i = (long)((int)blockIdx.x - tb.rfirst[gi]) * 256 + threadIdx.x;
    if (i >= total) return;
    const float* part = ws + g.ws_off;
    float s = 0.f;
    for (int k = 0; k < g.ksplit; ++k) s += part[(long)k * total + i];
    const int n = (int)(i % g.N);
    const long bm = i / g.N;
    const int m = (int)(bm % g.M), b = (int)(bm / g.M);
    g.C[b * g.c_sb + (long)m * g.c_sm + n] = s;
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
        if (g.ksplit > 1) rblocks += (int)(((long)g.batch * g.M * g.N + 255) / 256);
    }
    tb.first[tb.count] = blocks;
    tb.rfirst[tb.count] = rblocks;
    hipLaunchKernelGGL(gemm_kmajor_kernel, dim3(blocks), dim3(256), 0, st, tb, ws);
    if (rblocks > 0) hipLaunchKernelGGL(gemm_reduce_kernel, dim3(rblocks), dim3(256), 0, st, tb, ws);
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
// adam_tick_kernel advances the device-side step counter and leaves the two bias-correction factors next to it
// (state[0] = t, state[1] = lr / (1 - b1^t), state[2] = 1 / sqrt(1 - b2^t)); adam_kernel applies the update to up to
// kAdamTensors tensors: workgroup w serves elements [1024 (w - first[i]), +1024) of tensor i, first[i] <= w < first[i+1].
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

__global__ void adam_tick_kernel(float* __restrict__ state, float lr, float b1, float b2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float t = state[0] + 1.f;
    state[0] = t;
    state[1] = (float)((double)lr / (1.0 - pow((double)b1, (double)t)));
    state[2] = (float)(1.0 / sqrt(1.0 - pow((double)b2, (double)t)));
}

__global__ __launch_bounds__(256) void adam_kernel(const AdamTable tb, const float* __restrict__ state, float b1,
                                                   float b2, float eps, float wd) {
    int i = 0;
    while (i + 1 < tb.count && (int)blockIdx.x >= tb.first[i + 1]) ++i;     // (scalar: at most 31 steps)
    const long base = (long)((int)blockIdx.x - tb.first[i]) * 1024;
    const float step_size = state[1], inv_bc2 = state[2];
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
}

}  // namespace gnnpp
