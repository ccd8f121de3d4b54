// Train-mode CNN encoder, forward AND backward, for gfx950 (BASELINE config 4: dcp_onlineExpert training,
// loss.backward() at agents/decentralplannerlocal.py:314 over graphs/models/decentralplanner.py:284-290).
//
// What makes train mode different from the fused inference kernels: the reference runs ConvLayers once
// PER AGENT on that agent's mini-batch [B,3,11,11], so every BatchNorm2d normalises with the statistics
// of (agent n, channel c) over the B samples and the H x W positions of THAT call, and updates its
// running statistics N times per forward, in agent order.  That is a reduction across the whole batch
// between every convolution and its ReLU -- a layer-by-layer schedule with the activations in HBM
// (they are needed again by the backward pass anyway): at B = 64, N = 10 a layer is a few MB.
//
// Layout of every activation tensor: [N][B][C][P] (agent-major, P = H*W row-major), fp32.  All kernels
// are plain fp32 arithmetic (fmaf chains / fp32 MFMA), deterministic: every reduction has a fixed order
// (per-wave partial sums in LDS, partials summed in index order by a finalize kernel; no atomics).
//
//   forward, per conv layer l (3->32 @11x11 pool, 32->32 @5x5, 32->64 @5x5 pool, 64->64 @2x2, 64->128 @2x2 pool)
//     conv_cols_kernel        y = conv3x3(x) + bias: one lane = one output column (b, y, x) of agent n,
//                             16 output channels in registers; the weights of the wave's channel tile are
//                             WAVE-UNIFORM, so they come through scalar loads and enter the FMAs as SGPR
//                             operands ([ci][co][tap] copy made by pack_train_weights_kernel); per-wave
//                             partial (sum, sum of squares) per channel for the BatchNorm statistics
//     bn_stats_kernel         partials -> mean, 1/sqrt(var + eps) per (agent, channel) (+ unbiased variance
//                             for the running statistics)
//     bn_relu_pool_kernel     x_{l+1} = maxpool2x2?( relu( (y - mean) * invstd * gamma + beta ) )
//   then bn_running_kernel    the N sequential momentum updates of every layer's running statistics
//   backward, per layer from the last to the first
//     bn_bwd_reduce_kernel    dz = relu'(a) * unpool(d x_{l+1})  (a and the pool's arg-max recomputed from y:
//                             first maximum in scan order, as torch's max_pool2d backward), written out,
//                             with per-wave partial sums of dz and dz * yhat
//     bn_bwd_coef_kernel      partials -> per (agent, channel) coefficients; d gamma, d beta summed over agents
//     bn_bwd_apply_kernel     dy = gamma * invstd * (dz - mean(dz) - yhat * mean(dz * yhat)), in place
//     conv_cols_kernel        dx = conv3x3(dy) with the transposed, flipped kernel (skipped for layer 0)
//     conv_wgrad_kernel       dW[co][ci][tap] (and d bias) = sum over all columns of dy x patch(x): a GEMM with
//                             the columns as the contraction index, on the fp32 MFMA 16x16x4, split over
//                             column ranges; conv_wgrad_reduce_kernel sums the splits in order.
// The 128 -> 128 compress MLP, the graph filter and the action head are not in here: the MLP is one library
// GEMM each way (torch), the graph filter runs on lsigf_kernel (graphML._LSIGFFunction).
#include "gnnpp_common.h"

namespace gnnpp {

constexpr int kTrainLayers = 5;
struct TrainLayerDims { int Cin, Cout, H, W, pool; };
__host__ __device__ inline TrainLayerDims train_layer(int l) {
    const TrainLayerDims d[kTrainLayers] = {{3, 32, 11, 11, 1}, {32, 32, 5, 5, 0}, {32, 64, 5, 5, 1},
                                            {64, 64, 2, 2, 0}, {64, 128, 2, 2, 1}};
    return d[l];
}
constexpr int kWgSplitMax = 640;     // column-range splits of conv_wgrad_kernel

// ---- weights in the order conv_cols_kernel consumes them: [input channel][tap][output channel], so that a
// wave's channel tile of one (input channel, tap) is a run of consecutive floats -- scalar loads deliver
// aligned SGPR pairs for v_pk_fma_f32 with no re-shuffling.  Two copies per layer, one launch for all:
//   b (forward):        wf[ci][tap][co]     = W[co][ci][tap]
//   c (input gradient): wb[co][tap][ci]     = W[co][ci][8 - tap]   (transposed + flipped kernel)
struct TrainPtrs5 { const float* a[kTrainLayers]; float* b[kTrainLayers]; float* c[kTrainLayers]; };
__global__ void pack_train_weights_kernel(const TrainPtrs5 p) {
    const int l = blockIdx.y;
    const TrainLayerDims d = train_layer(l);
    const float* w = p.a[l];
    const int total = d.Cin * d.Cout * 9;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int tap = i % 9, ci = (i / 9) % d.Cin, co = i / (9 * d.Cin);
        const float v = w[i];                                        // W[co][ci][tap]
        p.b[l][((long)ci * 9 + tap) * d.Cout + co] = v;
        p.c[l][((long)co * 9 + (8 - tap)) * d.Cin + ci] = v;
    }
}

// ---- convolution over columns ---------------------------------------------------------------------------
// out[n,b,co,p] = bias[co] + sum_ci sum_tap wk[(ci*9 + tap)*Cout + co] * in[n,b,ci,p + off(tap)]   (zero padding)
//   forward: wk = the layer's wf copy; input gradient: wk = its wb copy with the roles of Cin / Cout swapped
// grid = (N * chunks, Cout / 16), block = 64: wave (n, chunk) x channel tile; lane = column chunk*64 + lane
// of agent n (columns = B*P).  x image (n, b) starts at x + n*x_sn + b*x_sb (the observations arrive
// sample-major [B][N]...; every other tensor is agent-major).  part != nullptr: per-wave (sum, sum sq) of
// the outputs per channel -> part[((n*chunks + chunk)*Cout + co)*2 + {0,1}].
template <int CT>
__global__ __launch_bounds__(64) void conv_cols_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ wk,
                                                       const float* __restrict__ bias,
                                                       float* __restrict__ y, float* __restrict__ part,
                                                       int B, int Cin, int Cout, int H, int W, long x_sn,
                                                       long x_sb, int chunks) {
    const int lane = threadIdx.x;
    const int n = blockIdx.x / chunks, chunk = blockIdx.x - n * chunks;
    const int co0 = blockIdx.y * CT;
    const int P = H * W;
    const int col = chunk * 64 + lane;
    const bool active = col < B * P;
    const int colc = active ? col : 0;
    const int b = colc / P, pos = colc - b * P;
    const int py = pos / W, px = pos - py * W;
    int off[9];
    float msk[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
        const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
        off[t] = in ? yy * W + xx : pos;                 // (a valid address; the value is masked)
        msk[t] = in ? 1.f : 0.f;
    }
    float acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = bias ? bias[co0 + c] : 0.f;
    const float* xi = x + n * x_sn + b * x_sb;
    // input channels U at a time: all 9 U patch loads of a group are issued before its FMAs, so one
    // memory round trip feeds U * 9 * CT FMAs (Cin is 3 or a multiple of 8)
    constexpr int U = CT >= 16 ? 4 : 8;          // narrower channel tiles leave registers for deeper prefetch
    for (int ci0 = 0; ci0 < Cin; ci0 += U) {
        float patch[U][9];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ci = ci0 + u < Cin ? ci0 + u : Cin - 1;
            const float live = ci0 + u < Cin ? 1.f : 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) patch[u][t] = xi[(long)ci * P + off[t]] * (msk[t] * live);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ci = ci0 + u < Cin ? ci0 + u : Cin - 1;
            const float* wrow = wk + (long)ci * 9 * Cout + co0;        // wave-uniform: scalar loads
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[c] = fmaf(wrow[(long)t * Cout + c], patch[u][t], acc[c]);
        }
    }
    if (active) {
        float* yo = y + (((long)n * B + b) * Cout + co0) * P + pos;
#pragma unroll
        for (int c = 0; c < CT; ++c) yo[(long)c * P] = acc[c];
    }
    if (part) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const float v = active ? acc[c] : 0.f;
            const float s1 = wave_sum(v), s2 = wave_sum(v * v);
            if (lane == 0) {
                float* o = part + (((long)n * chunks + chunk) * Cout + co0 + c) * 2;
                o[0] = s1;
                o[1] = s2;
            }
        }
    }
}

// ---- convolution with one lane per OUTPUT CHANNEL (layers 1..4, forward and input gradient) ----------------
// The roles that suit the hardware when the images are tiny (5x5, 2x2): a lane owns an output channel, so
//   * the WEIGHTS differ per lane and arrive by coalesced vector loads (256 bytes per (input channel, tap),
//     in-order returns: the next input channel's nine are in flight during this one's FMAs),
//   * the ACTIVATIONS of the wave's IMG images are the same for every lane: scalar loads into SGPRs, used as
//     the scalar operand of v_fma -- fetched one input channel AHEAD (two register sets), so the
//     "wait for all scalar loads" that out-of-order SMEM returns force never waits for a load just issued,
//   * zero padding is resolved at compile time (H, W are template parameters: taps that fall outside the
//     image are not issued),
//   * the BatchNorm partial sums are lane-local (a lane = a channel): no cross-lane reduction at all.
// grid = (N * chunks, ceil(Cout / 64)), block = 64; wave = images [chunk*IMG, +IMG) of agent n.
// wk = [ci][tap][Cout] (forward: wf; input gradient: wb with the layer's Cin / Cout swapped).
template <int H, int W, int IMG>
__global__ __launch_bounds__(64) void conv_ch_kernel(const float* __restrict__ x, const float* __restrict__ wk,
                                                     const float* __restrict__ bias, float* __restrict__ y,
                                                     float* __restrict__ part, int B, int Cin, int Cout,
                                                     long x_sn, long x_sb, int chunks) {
    constexpr int P = H * W;
    const int lane = threadIdx.x;
    const int n = blockIdx.x / chunks, chunk = blockIdx.x - n * chunks;
    const int b0 = chunk * IMG;
    const int co = blockIdx.y * 64 + lane;
    const bool cv = co < Cout;
    const int coc = cv ? co : Cout - 1;
    const float bv = bias ? bias[coc] : 0.f;
    float acc[IMG][P];
#pragma unroll
    for (int i = 0; i < IMG; ++i)
#pragma unroll
        for (int p = 0; p < P; ++p) acc[i][p] = bv;
    const float* xb[IMG];                                   // wave-uniform image bases (clamped: the extra
#pragma unroll                                              // images of a ragged tail are computed, not stored)
    for (int i = 0; i < IMG; ++i) xb[i] = x + n * x_sn + (long)(b0 + i < B ? b0 + i : B - 1) * x_sb;
    const float* wl = wk + coc;

    auto load_x = [&](float (&xs)[IMG][P], int ci) {
#pragma unroll
        for (int i = 0; i < IMG; ++i)
#pragma unroll
            for (int p = 0; p < P; ++p) xs[i][p] = xb[i][(long)ci * P + p];
    };
    auto fma_all = [&](const float (&xs)[IMG][P], int ci) {
        float w[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) w[t] = wl[((long)ci * 9 + t) * Cout];
#pragma unroll
        for (int i = 0; i < IMG; ++i)
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int yy = p / W + t / 3 - 1, xx = p % W + t % 3 - 1;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W)
                        acc[i][p] = fmaf(w[t], xs[i][yy * W + xx], acc[i][p]);
                }
    };
    float xa[IMG][P], xc[IMG][P];
    load_x(xa, 0);
    for (int ci = 0; ci < Cin; ci += 2) {                   // Cin is even for every layer this kernel serves
        load_x(xc, ci + 1);
        fma_all(xa, ci);
        if (ci + 2 < Cin) load_x(xa, ci + 2);
        fma_all(xc, ci + 1);
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < IMG; ++i) {
        if (b0 + i < B) {                                   // (wave-uniform)
            float* yo = y + (((long)n * B + b0 + i) * Cout + coc) * P;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (cv) yo[p] = acc[i][p];
                s1 += acc[i][p];
                s2 += acc[i][p] * acc[i][p];
            }
        }
    }
    if (part && cv) {
        float* o = part + (((long)n * chunks + chunk) * Cout + co) * 2;
        o[0] = s1;
        o[1] = s2;
    }
}

// ---- BatchNorm statistics of one layer: grid = N blocks, Cout threads -------------------------------------
// stat[(n*C + c)*4 + {0: mean, 1: invstd, 2: unbiased variance (running stats), 3: unused}]
__global__ void bn_stats_kernel(const float* __restrict__ part, float* __restrict__ stat, int chunks, int C,
                                int m, float eps) {
    const int n = blockIdx.x, c = threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < chunks; ++k) {
        const float* p = part + (((long)n * chunks + k) * C + c) * 2;
        s1 += (double)p[0];
        s2 += (double)p[1];
    }
    const double mean = s1 / m;
    double var = s2 / m - mean * mean;
    if (var < 0.0) var = 0.0;
    float* o = stat + ((long)n * C + c) * 4;
    o[0] = (float)mean;
    o[1] = (float)(1.0 / sqrt(var + (double)eps));
    o[2] = (float)(m > 1 ? var * m / (m - 1) : var);
    o[3] = 0.f;
}

// ---- x_next = maxpool2x2?(relu(bn(y))): one thread per output element ---------------------------------------
__device__ __forceinline__ float bn_act(float yv, float mean, float invstd, float g, float be) {
    return fmaxf(fmaf((yv - mean) * invstd, g, be), 0.f);
}

// grid-stride over the elements of x_next [N][B][C][Po]
__global__ void bn_relu_pool_kernel(const float* __restrict__ y, const float* __restrict__ stat,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    float* __restrict__ xn, long total, int B, int C, int H, int W,
                                    int pool) {
    const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W, Po = Ho * Wo, P = H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int po = (int)(i % Po);
        const long ic = i / Po;                          // (n*B + b)*C + c
        const int c = (int)(ic % C);
        const int n = (int)((ic / C) / B);
        const float* st = stat + ((long)n * C + c) * 4;
        const float mean = st[0], invstd = st[1], g = gamma[c], be = beta[c];
        const float* yc = y + ic * P;
        float v;
        if (pool) {
            const int oy = po / Wo, ox = po - oy * Wo;
            const float* q = yc + (2 * oy) * W + 2 * ox;
            v = fmaxf(fmaxf(bn_act(q[0], mean, invstd, g, be), bn_act(q[1], mean, invstd, g, be)),
                      fmaxf(bn_act(q[W], mean, invstd, g, be), bn_act(q[W + 1], mean, invstd, g, be)));
        } else {
            v = bn_act(yc[po], mean, invstd, g, be);
        }
        xn[i] = v;
    }
}

// ---- running statistics: the N per-agent-call updates in agent order (one thread per channel) -------------
// nn.BatchNorm2d in train mode: r <- (1 - momentum) r + momentum * batch statistic (unbiased variance)
// all five layers in one launch: blockIdx.y = layer (a = stat, b = running_mean, c = running_var)
__global__ void bn_running_kernel(const TrainPtrs5 p, int N, float momentum) {
    const int l = blockIdx.y;
    const int C = train_layer(l).Cout;
    const float* stat = p.a[l];
    float* rmean = p.b[l];
    float* rvar = p.c[l];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C || !rmean || !rvar) return;
    float m = rmean[c], v = rvar[c];
    for (int n = 0; n < N; ++n) {
        const float* st = stat + ((long)n * C + c) * 4;
        m = (1.f - momentum) * m + momentum * st[0];
        v = (1.f - momentum) * v + momentum * st[2];
    }
    rmean[c] = m;
    rvar[c] = v;
}

// ---- backward through pool / ReLU / BatchNorm, pass 1 --------------------------------------------------------
// One lane = one column (b, pos) of agent n, looping over ALL channels: dz[c] = relu'(a) * d a, where
// d a = dxn[window] if this position is the FIRST maximum of its 2x2 window (scan order, like torch's
// max_pool2d backward; a recomputed from y), 0 for positions the pool never reads; without pool d a = dxn.
// Writes dz [N][B][C][P] and per-wave partial sums of dz and dz * yhat -> part[((n*chunks+chunk)*C + c)*2].
constexpr int kBnBwdCG = 8;          // channels per wave of bn_bwd_reduce_kernel (grid.y = C / 8)
__global__ __launch_bounds__(64) void bn_bwd_reduce_kernel(const float* __restrict__ y,
                                                           const float* __restrict__ stat,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ dxn,
                                                           float* __restrict__ dz, float* __restrict__ part,
                                                           int B, int C, int H, int W, int pool, int chunks) {
    const int lane = threadIdx.x;
    const int n = blockIdx.x / chunks, chunk = blockIdx.x - n * chunks;
    const int c0 = blockIdx.y * kBnBwdCG;
    const int P = H * W, Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W, Po = Ho * Wo;
    const int col = chunk * 64 + lane;
    const bool active = col < B * P;
    const int colc = active ? col : 0;
    const int b = colc / P, pos = colc - b * P;
    const int py = pos / W, px = pos - py * W;
    const bool pooled_in = !pool || (py < 2 * Ho && px < 2 * Wo);       // inside the region the pool reads
    const int oy = pool ? py / 2 : py, ox = pool ? px / 2 : px;
    const int wpos = pool && pooled_in ? (2 * oy) * W + 2 * ox : pos;   // window origin (a valid address)
    const int me = pool ? (py - 2 * oy) * 2 + (px - 2 * ox) : 0;        // my slot in the window
    const int wo[4] = {0, 1, W, W + 1};
#pragma unroll
    for (int cc = 0; cc < kBnBwdCG; ++cc) {
        const int c = c0 + cc;
        const float* st = stat + ((long)n * C + c) * 4;
        const float mean = st[0], invstd = st[1], g = gamma[c], be = beta[c];
        const long ic = ((long)n * B + b) * C + c;
        const float* yc = y + ic * P;
        const float yv = yc[pos];
        const float yhat = (yv - mean) * invstd;
        const float a = fmaxf(fmaf(yhat, g, be), 0.f);
        bool mine = active && pooled_in;
        if (pool) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float ak = bn_act(yc[pooled_in ? wpos + wo[k] : pos], mean, invstd, g, be);
                mine = mine && (k < me ? ak < a : ak <= a);           // strictly greater than the earlier ones
            }
        }
        const float da = dxn[ic * Po + (pooled_in ? oy * Wo + ox : 0)];
        const float d = (mine && a > 0.f) ? da : 0.f;
        if (active) dz[ic * P + pos] = d;
        const float s1 = wave_sum(d), s2 = wave_sum(d * yhat);
        if (lane == 0) {
            float* o = part + (((long)n * chunks + chunk) * C + c) * 2;
            o[0] = s1;
            o[1] = s2;
        }
    }
}

// pass 2a: coefficients per (agent, channel): coef[(n*C + c)*4 + {k1, k2, k3}] with
//   dy = k1 * (dz - k2 - yhat * k3),  k1 = gamma * invstd, k2 = mean(dz), k3 = mean(dz * yhat);
// grid = N blocks, C threads; the per-agent sums go to pn[(n*C + c)*2] for bn_bwd_dparam_kernel
__global__ void bn_bwd_coef_kernel(const float* __restrict__ part, const float* __restrict__ stat,
                                   const float* __restrict__ gamma, float* __restrict__ coef,
                                   float* __restrict__ pn, int chunks, int C, int m) {
    const int n = blockIdx.x, c = threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < chunks; ++k) {
        const float* p = part + (((long)n * chunks + k) * C + c) * 2;
        s1 += (double)p[0];
        s2 += (double)p[1];
    }
    float* o = coef + ((long)n * C + c) * 4;
    o[0] = gamma[c] * stat[((long)n * C + c) * 4 + 1];
    o[1] = (float)(s1 / m);
    o[2] = (float)(s2 / m);
    o[3] = 0.f;
    pn[((long)n * C + c) * 2] = (float)s1;
    pn[((long)n * C + c) * 2 + 1] = (float)s2;
}

// d gamma[c] = sum_n sum(dz * yhat), d beta[c] = sum_n sum(dz): agents in order, one thread per channel
__global__ void bn_bwd_dparam_kernel(const float* __restrict__ pn, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta, int N, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double tb = 0.0, tg = 0.0;
    for (int n = 0; n < N; ++n) {
        tb += (double)pn[((long)n * C + c) * 2];
        tg += (double)pn[((long)n * C + c) * 2 + 1];
    }
    dgamma[c] = (float)tg;
    dbeta[c] = (float)tb;
}

// pass 2b: dz -> dy in place
__global__ void bn_bwd_apply_kernel(const float* __restrict__ y, const float* __restrict__ stat,
                                    const float* __restrict__ coef, float* __restrict__ dz, long total, int B,
                                    int C, int P) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long ic = i / P;
        const int c = (int)(ic % C);
        const int n = (int)((ic / C) / B);
        const float* st = stat + ((long)n * C + c) * 4;
        const float* k = coef + ((long)n * C + c) * 4;
        const float yhat = (y[i] - st[0]) * st[1];
        dz[i] = k[0] * (dz[i] - k[1] - yhat * k[2]);
    }
}

// ---- weight gradient: dW[co][j], j = ci*9 + tap (j = Cin*9: the bias column) -------------------------------
//   dW[co][j] = sum over columns (n, b, p) of dy[n,b,co,p] * X[j][(n,b,p)],  X = x[n,b,ci,p + off(tap)] or 1
// A GEMM with the columns as the contraction index on v_mfma_f32_16x16x4_f32: a wave owns the 16 x 16 tile
// (co tile, j tile) and a range of images; A[i][k] = dy[co0+i][col k], B[k][j] = X[j0+j][col k], four
// columns per MFMA.  grid = (co tiles, j tiles, splits); partial results -> wpart[split][Cout][J16].
__global__ __launch_bounds__(64) void conv_wgrad_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ dy,
                                                        float* __restrict__ wpart, int NB, int Cin, int Cout,
                                                        int H, int W, long x_sn, long x_sb, int B,
                                                        int imgs_per_split) {
    const int lane = threadIdx.x;
    const int i16 = lane & 15, q = lane >> 4;
    const int co0 = blockIdx.x * 16, j0 = blockIdx.y * 16, split = blockIdx.z;
    const int P = H * W, J = Cin * 9 + 1, J16 = gridDim.y * 16;
    const int j = j0 + i16;                              // this lane's B column
    const bool jb = j == J - 1, jv = j < J - 1;
    const int ci = jv ? j / 9 : 0, tap = jv ? j - ci * 9 : 4;
    const int dyy = tap / 3 - 1, dxx = tap % 3 - 1;
    v4f acc = vzero();
    const int img0 = split * imgs_per_split, img1 = min(NB, img0 + imgs_per_split);
    // the K loop runs over (image, group of 4 positions) steps; four steps' operands are fetched before
    // their MFMAs so that one memory round trip feeds four of them
    const int spi = (P + 3) >> 2;                        // steps per image
    const int nsteps = (img1 - img0) * spi;
    constexpr int U = 4;
    for (int s0 = 0; s0 < nsteps; s0 += U) {
        float av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int sidx = s0 + u;
            const bool sv = sidx < nsteps;
            const int sc = sv ? sidx : 0;
            const int img = img0 + sc / spi, p = (sc - (sc / spi) * spi) * 4 + q;   // this lane's column
            const bool pv = sv && p < P;
            const int pc = pv ? p : 0;
            const int n = img / B, b = img - n * B;
            const int py = pc / W, px = pc - py * W;
            const int yy = py + dyy, xx = px + dxx;
            const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
            const float a0 = dy[((long)img * Cout + co0 + i16) * P + pc];                 // A: channel co0 + i16
            const float x0 = x[n * x_sn + b * x_sb + (long)ci * P + (in ? yy * W + xx : pc)];
            av[u] = pv ? a0 : 0.f;
            bv[u] = pv ? (jb ? 1.f : (jv && in ? x0 : 0.f)) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = mfma16(av[u], bv[u], acc);
    }
    // D register r of lane l: D[i = 4 q + r][j = l & 15]
    float* o = wpart + ((long)split * Cout + co0 + 4 * q) * J16 + j0 + i16;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[(long)r * J16] = acc[r];
}

// sum the splits: 8 lanes per output element each add every 8th split (in order), then the 8 partial sums
// are added in lane order -- a fixed association, deterministic.  dw [Cout][Cin][9], db [Cout]
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ wpart,
                                                                float* __restrict__ dw, float* __restrict__ db,
                                                                int nsplit, int Cin, int Cout, int J16) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    float* red = reinterpret_cast<float*>(gnnpp_smem);                 // [256]
    const int J = Cin * 9 + 1;
    const int total = Cout * J;
    const int sub = threadIdx.x & 7;
    const int i = blockIdx.x * 32 + (threadIdx.x >> 3);
    float s = 0.f;
    if (i < total) {
        const int co = i / J, jj = i - co * J;
        for (int k = sub; k < nsplit; k += 8) s += wpart[((long)k * Cout + co) * J16 + jj];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (sub == 0 && i < total) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[threadIdx.x + k];
        const int co = i / J, jj = i - co * J;
        if (jj == J - 1) db[co] = t;
        else dw[(long)co * (J - 1) + jj] = t;
    }
}

// ---- host side: workspace layout and the two entry points -----------------------------------------------------
// Workspace (floats), for N agents and B samples per agent:
//   per layer l: y_l [N*B*Cout*P] | x_{l+1} [N*B*Cout*Po] | stat_l [N*Cout*4]
//   scratch: wt (packed forward weights, all layers) | part (partial sums) | dz (largest y) | dxa, dxb
//            (gradients w.r.t. layer inputs, ping-pong) | coef [N*128*4] | wpart
struct TrainWs {
    size_t y[kTrainLayers], xn[kTrainLayers], stat[kTrainLayers], wt[kTrainLayers], wtb[kTrainLayers];
    size_t part, dz, dxa, dxb, coef, wpart, total;
    int chunks[kTrainLayers], nsplit[kTrainLayers], ips[kTrainLayers], jt[kTrainLayers];
};

inline TrainWs train_ws_layout(int N, int B) {
    TrainWs w;
    size_t o = 0, max_y = 0, max_part = 0, max_x = 0, max_wp = 0;
    const size_t NB = (size_t)N * B;
    for (int l = 0; l < kTrainLayers; ++l) {
        const TrainLayerDims d = train_layer(l);
        const int P = d.H * d.W, Po = d.pool ? (d.H / 2) * (d.W / 2) : P;
        w.y[l] = o; o += NB * d.Cout * P;
        w.xn[l] = o; o += NB * d.Cout * Po;
        w.stat[l] = o; o += (size_t)N * d.Cout * 4;
        w.wt[l] = o; o += (size_t)d.Cin * d.Cout * 9;
        w.wtb[l] = o; o += (size_t)d.Cin * d.Cout * 9;
        w.chunks[l] = (B * P + 63) / 64;
        max_y = max_y > NB * d.Cout * P ? max_y : NB * d.Cout * P;
        const int cmax = w.chunks[l] > B ? w.chunks[l] : B;          // column chunks (BN backward) vs image chunks (conv)
        const size_t pp = (size_t)N * cmax * d.Cout * 2;
        max_part = max_part > pp ? max_part : pp;
        max_x = max_x > NB * d.Cin * P ? max_x : NB * d.Cin * P;
        // weight-gradient splits: enough waves to fill the chip, at least one image per split
        w.jt[l] = (d.Cin * 9 + 1 + 15) / 16;
        const int tiles = (d.Cout / 16) * w.jt[l];
        int ns = (8192 + tiles - 1) / tiles;               // ~8 waves per SIMD: the K loop is latency bound
        if (ns > kWgSplitMax) ns = kWgSplitMax;
        if ((size_t)ns > NB) ns = (int)NB;
        w.ips[l] = (int)((NB + ns - 1) / ns);
        w.nsplit[l] = (int)((NB + w.ips[l] - 1) / w.ips[l]);
        const size_t wp = (size_t)w.nsplit[l] * d.Cout * w.jt[l] * 16;
        max_wp = max_wp > wp ? max_wp : wp;
    }
    w.part = o; o += max_part;
    w.dz = o; o += max_y;
    w.dxa = o; o += max_x;
    w.dxb = o; o += max_x;
    w.coef = o; o += (size_t)N * 128 * 6;            // coefficients [N][128][4] + per-agent sums [N][128][2]
    w.wpart = o; o += max_wp;
    w.total = o;
    return w;
}

static inline bool launched_ok() { return hipGetLastError() == hipSuccess; }

// Which convolution kernel serves a layer: the 11x11 first layer (3 input channels, 77 440 columns at B = 64,
// N = 10) keeps one lane per COLUMN with 16 channels in registers; the 5x5 and 2x2 layers (and their input
// gradients) put one lane per CHANNEL.  Returns the number of per-agent partial-sum chunks it writes.
static int conv_chunks(int l, int B) {
    const TrainLayerDims d = train_layer(l);
    if (l == 0) return (B * d.H * d.W + 63) / 64;
    return d.H == 5 ? B : (B + 3) / 4;
}

static void conv_launch(int l, bool input_grad, const float* x, const float* wk, const float* bias, float* y,
                        float* part, int N, int B, long sn, long sb, hipStream_t st) {
    const TrainLayerDims d = train_layer(l);
    const int Cin = input_grad ? d.Cout : d.Cin, Cout = input_grad ? d.Cin : d.Cout;
    const int chunks = conv_chunks(l, B);
    if (l == 0) {
        hipLaunchKernelGGL((conv_cols_kernel<16>), dim3(N * chunks, Cout / 16), dim3(64), 0, st, x, wk, bias, y,
                           part, B, Cin, Cout, d.H, d.W, sn, sb, chunks);
    } else if (d.H == 5) {
        hipLaunchKernelGGL((conv_ch_kernel<5, 5, 1>), dim3(N * chunks, (Cout + 63) / 64), dim3(64), 0, st, x, wk,
                           bias, y, part, B, Cin, Cout, sn, sb, chunks);
    } else {
        hipLaunchKernelGGL((conv_ch_kernel<2, 2, 4>), dim3(N * chunks, (Cout + 63) / 64), dim3(64), 0, st, x, wk,
                           bias, y, part, B, Cin, Cout, sn, sb, chunks);
    }
}

// obs: [B][N][3][11][11] (the reference's inputTensor, decentralplanner.py:278-286); feat = x_5 [N][B][128]
int train_encoder_fwd(const EncRawParams& rp, float* const* rmean, float* const* rvar, float momentum,
                      const float* obs, float* ws, float* feat, int N, int B, hipStream_t st) {
    const TrainWs L = train_ws_layout(N, B);
    const long NB = (long)N * B;
    TrainPtrs5 pk = {}, run = {};
    for (int l = 0; l < kTrainLayers; ++l) {
        pk.a[l] = rp.conv_w[l]; pk.b[l] = ws + L.wt[l]; pk.c[l] = ws + L.wtb[l];
        run.a[l] = ws + L.stat[l]; run.b[l] = rmean ? rmean[l] : nullptr; run.c[l] = rvar ? rvar[l] : nullptr;
    }
    hipLaunchKernelGGL(pack_train_weights_kernel, dim3(32, kTrainLayers), dim3(256), 0, st, pk);
    for (int l = 0; l < kTrainLayers; ++l) {
        const TrainLayerDims d = train_layer(l);
        const int P = d.H * d.W, Po = d.pool ? (d.H / 2) * (d.W / 2) : P;
        const float* xin = l == 0 ? obs : ws + L.xn[l - 1];
        const long sn = l == 0 ? (long)d.Cin * P : (long)B * d.Cin * P;        // obs is [B][N]: n is the inner index
        const long sb = l == 0 ? (long)N * d.Cin * P : (long)d.Cin * P;
        conv_launch(l, false, xin, ws + L.wt[l], rp.conv_b[l], ws + L.y[l], ws + L.part, N, B, sn, sb, st);
        hipLaunchKernelGGL(bn_stats_kernel, dim3(N), dim3(128), 0, st, ws + L.part, ws + L.stat[l],
                           conv_chunks(l, B), d.Cout, B * P, rp.bn_eps);
        const long tot = NB * d.Cout * Po;
        hipLaunchKernelGGL(bn_relu_pool_kernel, dim3((unsigned)((tot + 255) / 256 < 2048 ? (tot + 255) / 256 : 2048)),
                           dim3(256), 0, st, ws + L.y[l], ws + L.stat[l], rp.bn_w[l], rp.bn_b[l],
                           l == kTrainLayers - 1 ? feat : ws + L.xn[l], tot, B, d.Cout, d.H, d.W, d.pool);
    }
    if (rmean && rvar)
        hipLaunchKernelGGL(bn_running_kernel, dim3(1, kTrainLayers), dim3(128), 0, st, run, N, momentum);
    return launched_ok() ? 0 : -3;
}

// dfeat: gradient w.r.t. x_5 [N][B][128]; writes d conv_w / d conv_b / d bn_w / d bn_b of every layer
int train_encoder_bwd(const EncRawParams& rp, const float* obs, float* ws, const float* dfeat,
                      float* const* dconv_w, float* const* dconv_b, float* const* dbn_w, float* const* dbn_b,
                      int N, int B, hipStream_t st) {
    const TrainWs L = train_ws_layout(N, B);
    const long NB = (long)N * B;
    const float* dxn = dfeat;
    float* dx_buf[2] = {ws + L.dxa, ws + L.dxb};
    for (int l = kTrainLayers - 1; l >= 0; --l) {
        const TrainLayerDims d = train_layer(l);
        const int P = d.H * d.W;
        float* dz = ws + L.dz;
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(N * L.chunks[l], d.Cout / kBnBwdCG), dim3(64), 0, st,
                           ws + L.y[l], ws + L.stat[l], rp.bn_w[l], rp.bn_b[l], dxn, dz, ws + L.part, B,
                           d.Cout, d.H, d.W, d.pool, L.chunks[l]);
        hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3(N), dim3(128), 0, st, ws + L.part, ws + L.stat[l],
                           rp.bn_w[l], ws + L.coef, ws + L.coef + (size_t)N * 128 * 4, L.chunks[l], d.Cout, B * P);
        hipLaunchKernelGGL(bn_bwd_dparam_kernel, dim3(1), dim3(128), 0, st, ws + L.coef + (size_t)N * 128 * 4,
                           dbn_w[l], dbn_b[l], N, d.Cout);
        const long tot = NB * d.Cout * P;
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((tot + 255) / 256 < 2048 ? (tot + 255) / 256 : 2048)),
                           dim3(256), 0, st, ws + L.y[l], ws + L.stat[l], ws + L.coef, dz, tot, B, d.Cout, P);
        const float* xin = l == 0 ? obs : ws + L.xn[l - 1];
        const long sn = l == 0 ? (long)d.Cin * P : (long)B * d.Cin * P;
        const long sb = l == 0 ? (long)N * d.Cin * P : (long)d.Cin * P;
        hipLaunchKernelGGL(conv_wgrad_kernel, dim3(d.Cout / 16, L.jt[l], L.nsplit[l]), dim3(64), 0, st, xin, dz,
                           ws + L.wpart, (int)NB, d.Cin, d.Cout, d.H, d.W, sn, sb, B, L.ips[l]);
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((d.Cout * (d.Cin * 9 + 1) + 31) / 32), dim3(256),
                           256 * sizeof(float), st, ws + L.wpart, dconv_w[l], dconv_b[l], L.nsplit[l], d.Cin,
                           d.Cout, L.jt[l] * 16);
        if (l > 0) {
            // dx [N][B][Cin][P] = conv(dy) with the flipped kernel; here "Cin" of the call = Cout of the layer
            float* dx = dx_buf[l & 1];
            // output channels of this call = d.Cin (a multiple of 16 for l >= 1)
            conv_launch(l, true, dz, ws + L.wtb[l], nullptr, dx, nullptr, N, B, (long)B * d.Cout * P,
                        (long)d.Cout * P, st);
            dxn = dx;
        }
    }
    return launched_ok() ? 0 : -3;
}

}  // namespace gnnpp
