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
// are plain fp32 arithmetic (exact fp32 MFMA 16x16x4 / fmaf), deterministic: every reduction has a fixed
// order (per-wave partial sums, partials summed in index order by the kernel that consumes them; no atomics).
//
//   forward: pack_train_weights_kernel (the ten weight packs of the step -- and the graph filter's two -- in one
//   launch; once per weight version when the caller keeps the pack: gnnpp_train_pack), then per conv layer l
//   (3->32 @11x11 pool, 32->32 @5x5, 32->64 @5x5 pool, 64->64 @2x2, 64->128 @2x2 pool)
//     conv_mfma_kernel        y = conv3x3(x) + bias as a GEMM on the fp32 MFMA: rows = output channels,
//                             columns = (image, position), contraction = (tap, input channel); images
//                             zero-bordered in LDS, A fragments from the pack; the 2x2 layers run as DENSE
//                             maps of the flattened image (no multiplications spent on the border); per-workgroup
//                             partial (sum, sum of squares) per channel for the BatchNorm statistics
//     bn_relu_pool_kernel     partials -> mean, 1/sqrt(var + eps) per (agent, channel) (+ unbiased variance for
//                             the running statistics) in the workgroup's prologue, then
//                             x_{l+1} = maxpool2x2?( relu( (y - mean) * invstd * gamma + beta ) )
//   then bn_running_kernel    the N sequential momentum updates of every layer's running statistics
//   backward, per layer from the last to the first
//     bn_bwd_reduce_kernel    dz = relu'(a) * unpool(d x_{l+1})  (a and the pool's arg-max recomputed from y:
//                             first maximum in scan order, as torch's max_pool2d backward), written out,
//                             with per-wave partial sums of dz and dz * yhat
//     bn_bwd_apply_kernel     partials -> per (agent, channel) coefficients in the workgroup's prologue, then
//                             dy = gamma * invstd * (dz - mean(dz) - yhat * mean(dz * yhat)), in place
//     (after the first layer: ONE conv_wgrad_reduce_kernel launch sums the splits of all five layers, and d gamma / d beta)
//     conv_wgrad_kernel       dW[co][ci][tap] (and d bias) = sum over all columns of dy x patch(x): a GEMM with
//                             the columns as the contraction index, on the fp32 MFMA 16x16x4, operands staged
//                             through LDS, split over image ranges; conv_wgrad_reduce_kernel sums the splits in order
//     conv_mfma_kernel        dx = conv3x3(dy) with the transposed, flipped kernel (skipped for layer 0)
// The 128 -> 128 compress MLP, the graph filter and the action head are not in here: the MLP is one library
// GEMM each way (torch), the graph filter runs on lsigf_kernel (graphML._LSIGFFunction).
#include <mutex>

#include "gnnpp_common.h"

namespace gnnpp {

constexpr int kTrainLayers = 5;
struct TrainLayerDims { int Cin, Cout, H, W, pool; };
__host__ __device__ inline TrainLayerDims train_layer(int l) {
    const TrainLayerDims d[kTrainLayers] = {{3, 32, 11, 11, 1}, {32, 32, 5, 5, 0}, {32, 64, 5, 5, 1},
                                            {64, 64, 2, 2, 0}, {64, 128, 2, 2, 1}};
    return d[l];
}

// ---- convolution geometry: every layer, forward and input gradient, as ONE GEMM shape ------------------------------
// out[img][m][p] = bias[m / RPC] + sum over (tap, c) A[m][(tap, c)] * in[img][c][p + off(tap)]   (zero padding)
//   11x11 and 5x5 layers: the 3x3 convolution itself (TAPS = 9; rows m = output channels; RPC = 1);
//   2x2 layers: every output position sees every input position through exactly one tap, so the layer is a
//   DENSE map of the flattened image: rows m = (channel, position) (RPC = 4 rows per channel), contraction
//   index c = (input channel, input position), TAPS = 1, "image" of one position.  The 3x3 form would spend
//   5/9 of its multiplications on the zero border.
//   input gradient: the same with the transposed, flipped kernel (rows = the layer's input channels).
// IB images per workgroup, CC contraction channels per LDS stage.
constexpr int kDenseIB = 32;                // images per workgroup of the dense (2x2) layers (r06: 64 -> 32: the 32 row-tile
                                            // workgroups of an image chunk each re-stage the chunk -- per-workgroup staging latency is
                                            // the launch; conv_mfma_kernel<1, 1> 11.4 -> 9.2 us per launch, 16 images: 10.5)
struct ConvGeom { int H, W, P, TAPS, RPC, IB, CC, M, Ck, nchunk, SPC, chunks_per_agent; };
__host__ __device__ inline ConvGeom conv_geom(int l, bool input_grad, int B) {
    const TrainLayerDims d = train_layer(l);
    ConvGeom g;
    const bool dense = d.H == 2;
    g.H = dense ? 1 : d.H; g.W = dense ? 1 : d.W; g.P = g.H * g.W;
    g.TAPS = dense ? 1 : 9;
    g.RPC = dense ? 4 : 1;
    g.IB = dense ? kDenseIB : d.H == 5 ? 4 : 2;
    g.CC = dense ? 256 : d.H == 5 ? 32 : 4;
    const int rows = input_grad ? d.Cin : d.Cout, ck = input_grad ? d.Cout : d.Cin;
    g.M = rows * g.RPC;
    g.Ck = ck * g.RPC;
    g.nchunk = (g.Ck + g.CC - 1) / g.CC;
    g.SPC = g.TAPS * g.CC / 4;
    g.chunks_per_agent = (B + g.IB - 1) / g.IB;
    return g;
}
__host__ __device__ constexpr int conv_row_stride(int ibpp, bool dense) {
    return dense ? ibpp + 1 : (ibpp - 16 + 63) / 64 * 64 + 16;
}
__host__ __device__ inline size_t conv_pack_floats(int l, bool input_grad) {
    const ConvGeom g = conv_geom(l, input_grad, 1);
    return (size_t)g.M * g.nchunk * g.SPC * 4;
}

// ---- weights as MFMA A fragments: pack[((mt*nchunk + ch)*SPC + s)*64 + lane] = A[m = mt*16 + (lane & 15)]
// [k-step s of stage ch: tap = s / (CC/4), c = ch*CC + 4*(s % (CC/4)) + (lane >> 4)], 0 for c >= Ck.  One launch for
// the ten packs of a step (blockIdx.y = layer*2 + {0: forward -> b, 1: input gradient -> c}).
struct TrainPtrs5 { const float* a[kTrainLayers]; float* b[kTrainLayers]; float* c[kTrainLayers]; };
// The graph filter's taps of the same step (r06; graphML.py:2434 `weight` [F,E,K,G]): the fp32 MFMA fragments of the
// forward filter (-> fwd, the first region of a gnnpp_filter_pack buffer of h) and of the input-gradient filter = the
// filter of h^T [G,E,K,F] (-> tr), read straight from h: blockIdx.y = 2 * kTrainLayers + {0, 1}.  The training step's
// filter launches contract in exact fp32 and read nothing else of those buffers; r05 packed every region of both
// (split-f16 scale + fragments, bf16x3 planes) with two launches each, behind a transposing copy of h.
struct TrainFilterPack { const float* h; float* fwd; float* tr; int G, F, K, E; };
__device__ __forceinline__ void pack_train_filter(const TrainFilterPack f, bool transposed) {
    const int Fo = transposed ? f.G : f.F, Gi = transposed ? f.F : f.G;   // output / input features of THIS filter
    const int NG = (Gi + 15) / 16, MT = (Fo + 15) / 16;
    const long total = (long)f.E * f.K * MT * NG * 256;
    float* out = transposed ? f.tr : f.fwd;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int s = idx & 3, l = (idx >> 2) & 63;
        long blk = idx >> 8;
        const int gg = blk % NG; blk /= NG;
        const int mt = blk % MT; blk /= MT;
        const int k = blk % f.K;
        const int e = (int)(blk / f.K);
        const int fo = mt * 16 + (l & 15), gi = gg * 16 + (l >> 4) * 4 + s;
        float v = 0.f;
        if (fo < Fo && gi < Gi) {
            const int hf = transposed ? gi : fo, hg = transposed ? fo : gi;          // h[f][e][k][g]
            v = f.h[(((long)hf * f.E + e) * f.K + k) * f.G + hg];
        }
        out[idx] = v;
    }
}

__global__ void pack_train_weights_kernel(const TrainPtrs5 p, const TrainFilterPack fp) {
    if (blockIdx.y >= 2 * kTrainLayers) {
        if (fp.h) pack_train_filter(fp, blockIdx.y == 2 * kTrainLayers + 1);
        return;
    }
    const int l = blockIdx.y >> 1;
    const bool ig = blockIdx.y & 1;
    if (!p.a[l]) return;                                             // (a filter-only call)
    const TrainLayerDims d = train_layer(l);
    const ConvGeom g = conv_geom(l, ig, 1);
    const float* w = p.a[l];
    float* out = ig ? p.c[l] : p.b[l];
    if (ig && l == 0) return;                                        // (no input gradient of the observations)
    const int total = g.M * g.nchunk * g.SPC * 4, spm = g.nchunk * g.SPC;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int lane = e & 63, sg = (e >> 6) % spm, mt = (e >> 6) / spm;
        const int m = mt * 16 + (lane & 15);
        const int ch = sg / g.SPC, s0 = sg - ch * g.SPC;
        const int tap = s0 / (g.CC / 4), cig = s0 - tap * (g.CC / 4);
        const int c = ch * g.CC + 4 * cig + (lane >> 4);
        float v = 0.f;
        if (c < g.Ck) {
            int co, ci, t;
            if (g.RPC == 1) {                                        // the 3x3 convolution
                co = ig ? c : m; ci = ig ? m : c; t = ig ? 8 - tap : tap;
            } else {                                                 // dense 2x2: positions po (output), pi (input)
                const int row_ch = m >> 2, row_p = m & 3, k_ch = c >> 2, k_p = c & 3;
                co = ig ? k_ch : row_ch; ci = ig ? row_ch : k_ch;
                const int po = ig ? k_p : row_p, pi = ig ? row_p : k_p;
                t = ((pi >> 1) - (po >> 1) + 1) * 3 + ((pi & 1) - (po & 1) + 1);
            }
            v = w[((long)co * d.Cin + ci) * 9 + t];
        }
        out[e] = v;
    }
}

// The ten weight packs of a step as ONE caller-owned buffer (gnnpp_train_pack; r06): packed once per weight version
// together with the graph filter's taps instead of once per forward call inside the workspace.
struct TrainPackLayout { size_t wt[kTrainLayers], wtb[kTrainLayers], total; };
inline TrainPackLayout train_pack_layout() {
    TrainPackLayout w;
    size_t o = 0;
    for (int l = 0; l < kTrainLayers; ++l) {
        w.wt[l] = o;  o += (conv_pack_floats(l, false) + 3) & ~(size_t)3;
        w.wtb[l] = o; o += (conv_pack_floats(l, true) + 3) & ~(size_t)3;
    }
    w.total = o;
    return w;
}

// ---- the convolution on v_mfma_f32_16x16x4_f32 -------------------------------------------------------------------
// grid = (N * chunks_per_agent, M / 16), block = 256.  A workgroup owns 16 output rows and the IB images
// [b0, b0 + IB) of agent n, i.e. IB*P columns in tiles of 16; wave w holds the accumulators of column tiles
// w, w + 4, ... (TN of them).  Per stage of CC contraction channels: the A fragments of the stage (one
// coalesced copy of the pack) and the raw images go to LDS -- the images ZERO-BORDERED, [c][img][(H+2)(W+2)],
// so the operand of any (tap, c) is the lane's column offset plus a compile-time constant: the inner loop is
// ds_read (immediate offsets) + MFMA and nothing else.  Lane (i = lane & 15, q = lane >> 4): A value (row i,
// k = 4s + q), B value (k = 4s + q, column i);  D register r: (row 4q + r, column i).
// Epilogue: + bias, store, and (part != nullptr) the BatchNorm partial sums of the workgroup's channels over
// its valid columns -> part[((n*chunks + chunk)*C + c)*2 + {sum, sum of squares}] (16-lane butterflies, then
// the four waves in order: deterministic).
template <int H, int W>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        float* __restrict__ part, int B, int Ck, int M, long x_sn,
                                                        long x_sb, int chunks, int nchunk) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    constexpr bool kDense = H == 1 && W == 1;
    constexpr int P = H * W, TAPS = kDense ? 1 : 9, RPC = kDense ? 4 : 1;
    constexpr int PP = kDense ? 1 : (H + 2) * (W + 2);
    constexpr int IB = kDense ? kDenseIB : H == 5 ? 4 : 2, CC = kDense ? 256 : H == 5 ? 32 : 4;
    constexpr int SPC = TAPS * CC / 4, NT = (IB * P + 15) / 16, TN = (NT + 3) / 4;
    // floats between two channels' rows in LDS, padded against bank conflicts (64 banks): the B read's four
    // channel groups q sit 16 banks apart (stride = 16 mod 64); the dense layers' 64 one-float "images" get a
    // stride of 65 (a 16-byte load holds four CHANNELS of one image: with 64 every lane's scatter would hit
    // one bank)
    constexpr int RS = conv_row_stride(IB * PP, kDense);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    const int n = blockIdx.x / chunks, chunk = blockIdx.x - n * chunks;
    const int b0 = chunk * IB, nimg = min(IB, B - b0);
    const int m0 = blockIdx.y * 16;
    float* wsm = reinterpret_cast<float*>(gnnpp_smem);          // [SPC][64]
    float* xs = wsm + SPC * 64;                                  // [CC][RS >= IB*PP]
    int cb[TN], cimg[TN], cp[TN];
    bool cv[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int col = (wave + 4 * t) * 16 + i16;
        cv[t] = col < nimg * P;
        cimg[t] = cv[t] ? col / P : 0;
        cp[t] = cv[t] ? col - cimg[t] * P : 0;
        cb[t] = cimg[t] * PP + (kDense ? 0 : (cp[t] / W + 1) * (W + 2) + cp[t] % W + 1) + q * RS;
    }
    v4f acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[t] = vzero();

    // Staging = two phases so that global latency is paid once per stage, not once per element: issue() puts
    // every load of a stage in flight (16-byte loads: five for the A fragments, up to eight for the images; the
    // 11x11 layer's odd-sized observations go element-wise), commit() scatters the registers into LDS.  The
    // next stage's issue() runs BEFORE this stage's MFMAs, its commit() after them.
    const float* xa = x + n * x_sn + (long)b0 * x_sb;
    constexpr bool kVec = H != 11;                              // 16-byte image loads (runs and bases are aligned)
    constexpr int NWV = (SPC * 16 + 255) / 256;                 // float4 A-fragment loads per thread
    constexpr int XE = CC * IB * P;                             // image elements of a full stage
    constexpr int NXV = kVec ? (XE / 4 + 255) / 256 : (XE + 255) / 256;
    v4f wv[NWV];
    v4f xv[kVec ? NXV : 1];
    float xsc[kVec ? 1 : NXV];
    auto issue = [&](int ch) {
        const v4f* wsrc = reinterpret_cast<const v4f*>(wp + ((long)(blockIdx.y * nchunk + ch) * SPC) * 64);
#pragma unroll
        for (int u = 0; u < NWV; ++u) {
            const int i = tid + u * 256;
            wv[u] = wsrc[i < SPC * 16 ? i : 0];
        }
        const int c0 = ch * CC;
        if constexpr (kVec) {
#pragma unroll
            for (int u = 0; u < NXV; ++u) {
                const int e = (tid + u * 256) * 4;
                const int im = e / (CC * P), r = e - im * (CC * P);
                const bool ok = e < XE && im < nimg;
                xv[u] = *reinterpret_cast<const v4f*>(xa + (ok ? (long)im * x_sb + (long)c0 * P + r : 0));
            }
        } else {
#pragma unroll
            for (int u = 0; u < NXV; ++u) {
                const int e = tid + u * 256;
                const int im = e / (CC * P), r = e - im * (CC * P);
                const bool ok = e < XE && im < nimg && c0 + r / P < Ck;
                xsc[u] = xa[ok ? (long)im * x_sb + (long)c0 * P + r : 0];
            }
        }
    };
    auto xs_at = [&](int im, int r) {                           // LDS slot of element r = cl*P + pp of image im
        const int cl = r / P, pp = r - cl * P;
        return cl * RS + im * PP + (kDense ? 0 : (pp / W + 1) * (W + 2) + pp % W + 1);
    };
    auto commit = [&](int ch) {
        v4f* wdst = reinterpret_cast<v4f*>(wsm);
#pragma unroll
        for (int u = 0; u < NWV; ++u) {
            const int i = tid + u * 256;
            if (i < SPC * 16) wdst[i] = wv[u];
        }
        if constexpr (kVec) {
#pragma unroll
            for (int u = 0; u < NXV; ++u) {
                const int e = (tid + u * 256) * 4;
                const int im = e / (CC * P), r = e - im * (CC * P);
                if (e < XE && im < nimg) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) xs[xs_at(im, r + k)] = xv[u][k];
                }
            }
        } else {
            const int c0 = ch * CC;
#pragma unroll
            for (int u = 0; u < NXV; ++u) {
                const int e = tid + u * 256;
                const int im = e / (CC * P), r = e - im * (CC * P);
                if (e < XE && im < nimg) xs[xs_at(im, r)] = c0 + r / P < Ck ? xsc[u] : 0.f;
            }
        }
    };
    issue(0);
    for (int i = tid; i < CC * RS; i += 256) xs[i] = 0.f;        // borders / missing images stay zero
    __syncthreads();
    commit(0);
    __syncthreads();
    for (int ch = 0; ch < nchunk; ++ch) {
        if (ch + 1 < nchunk) issue(ch + 1);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int toff = kDense ? 0 : (tap / 3 - 1) * (W + 2) + (tap % 3 - 1);
#pragma unroll
            for (int cig = 0; cig < CC / 4; ++cig) {
                const float a = wsm[(tap * (CC / 4) + cig) * 64 + lane];
#pragma unroll
                for (int t = 0; t < TN; ++t)
                    acc[t] = mfma16(a, xs[cb[t] + toff + cig * 4 * RS], acc[t]);
            }
        }
        if (ch + 1 < nchunk) {
            __syncthreads();                                     // every wave is done reading this stage
            commit(ch + 1);
            __syncthreads();
        }
    }

    // ---- epilogue
    const int C = M / RPC;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 4 * q + r;
        const float bv = bias ? bias[m / RPC] : 0.f;
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const float v = acc[t][r] + bv;
            if (cv[t]) {
                y[(((long)n * B + b0 + cimg[t]) * M + m) * P + cp[t]] = v;
                s1[RPC == 1 ? r : 0] += v;
                s2[RPC == 1 ? r : 0] += v * v;
            }
        }
    }
    if (part) {
        constexpr int NS = RPC == 1 ? 4 : 1;                     // channel slots per lane
        __syncthreads();                                         // the A fragments are dead: reuse their LDS
        float* red = wsm;                                        // [4 waves][16 / RPC channels][2]
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            float a1 = s1[k], a2 = s2[k];
#pragma unroll
            for (int msk = 1; msk < 16; msk <<= 1) {
                a1 += __shfl_xor(a1, msk);
                a2 += __shfl_xor(a2, msk);
            }
            if (i16 == 0) {
                const int slot = RPC == 1 ? 4 * q + k : q;
                red[(wave * 16 + slot) * 2] = a1;
                red[(wave * 16 + slot) * 2 + 1] = a2;
            }
        }
        __syncthreads();
        if (tid < 16 / RPC) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                t1 += red[(w * 16 + tid) * 2];
                t2 += red[(w * 16 + tid) * 2 + 1];
            }
            float* o = part + (((long)n * chunks + chunk) * C + m0 / RPC + tid) * 2;
            o[0] = t1;
            o[1] = t2;
        }
    }
}

// ---- sums of per-chunk partial pairs, inside the kernel that consumes them -----------------------------------
// A workgroup that needs the (sum, sum) pairs of CG channels of agent n adds the `chunks` partial pairs of each
// channel itself: 256 / CG threads per channel take every (256/CG)-th chunk in order (doubles), then a fixed
// binary tree in LDS.  The association is a function of (chunks, CG) only: deterministic, and every workgroup
// that reduces the same channel obtains the same bits.  Costs a few hundred loads per workgroup instead of
// a launch with N blocks walking the chunks serially.  red: LDS, 512 doubles.  Result of channel cc (all
// threads, after the call): red[(cc * (256 / CG)) * 2 + {0, 1}].
__device__ __forceinline__ void reduce_partials(const float* __restrict__ part, int n, int chunks, int C, int c0,
                                                int CG, double* red) {
    const int tpc = 256 / CG;                               // threads per channel (CG is a power of two <= 32)
    const int cc = threadIdx.x / tpc, r = threadIdx.x - cc * tpc;
    double s1 = 0.0, s2 = 0.0;
    for (int k = r; k < chunks; k += tpc) {
        const float* p = part + (((long)n * chunks + k) * C + c0 + cc) * 2;
        s1 += (double)p[0];
        s2 += (double)p[1];
    }
    red[threadIdx.x * 2] = s1;
    red[threadIdx.x * 2 + 1] = s2;
    __syncthreads();
    for (int off = tpc >> 1; off > 0; off >>= 1) {
        if (r < off) {
            red[threadIdx.x * 2] += red[(threadIdx.x + off) * 2];
            red[threadIdx.x * 2 + 1] += red[(threadIdx.x + off) * 2 + 1];
        }
        __syncthreads();
    }
}

// Workgroup shape of the two BatchNorm element-wise kernels: (channels per workgroup, images per workgroup)
// chosen so that a workgroup's slice of one image is a contiguous run of >= 64 floats and a layer has a
// few hundred workgroups.
constexpr size_t kBnSmem = 512 * sizeof(double) + 32 * 5 * sizeof(float);
struct BnTile { int CG, BR; };
__host__ __device__ inline BnTile bn_tile(int l) {
    const BnTile t[kTrainLayers] = {{2, 8}, {8, 8}, {8, 8}, {32, 16}, {32, 16}};
    return t[l];
}

// ---- x_next = maxpool2x2?(relu(bn(y))) ----------------------------------------------------------------------
__device__ __forceinline__ float bn_act(float yv, float mean, float invstd, float g, float be) {
    return fmaxf(fmaf((yv - mean) * invstd, g, be), 0.f);
}

// The running-statistics update (nn.BatchNorm2d in train mode: r <- (1 - momentum) r + momentum * batch statistic,
// unbiased variance; the N per-agent-call updates in agent order) INSIDE the forward's last bn_relu_pool launch (r06b;
// r05 / r06a: bn_running_kernel, a launch of its own).  mode 1 (the first layer's launch): zero the arrival counters.
// mode 2 (the last layer's launch, grid z = N + 1): the extra z plane updates layers 0 .. L-2, whose statistics earlier
// launches wrote (workgroup x = layer); the last layer's statistics are written by THIS launch's workgroups (x, 0, n),
// n = 0 .. N-1, so the N workgroups of a channel tile x publish them with the release / ticket / acquire hand-off of
// gnnpp_common.h and the one that arrives last updates the tile's channels.
struct BnRunningFuse {
    const float* stat[kTrainLayers];
    float* rmean[kTrainLayers];
    float* rvar[kTrainLayers];
    long long* nb[kTrainLayers];                   // num_batches_tracked (+= N) or nullptr
    unsigned* tick;                                // [C_last / CG_last <= 8] arrival counters
    float momentum;
    int mode, N;
};
__device__ __forceinline__ void bn_running_update(const float* __restrict__ stat, float* __restrict__ rmean,
                                                  float* __restrict__ rvar, int C, int c, int N, float momentum) {
    float m = rmean[c], v = rvar[c];
    for (int n = 0; n < N; ++n) {
        const float* st = stat + ((long)n * C + c) * 4;
        m = (1.f - momentum) * m + momentum * st[0];
        v = (1.f - momentum) * v + momentum * st[2];
    }
    rmean[c] = m;
    rvar[c] = v;
}

// grid = (C / CG, ceil(B / BR), N), block = 256: BatchNorm statistics of the workgroup's channels from the
// convolution's partial sums (m = B * P values each), then the element-wise part over images [b0, b0 + BR).
// stat[(n*C + c)*4 + {0: mean, 1: invstd, 2: unbiased variance (running stats), 3: unused}] is written by
// the workgroups of the first image range (the backward pass and the running-statistics update read it).
__global__ __launch_bounds__(256) void bn_relu_pool_kernel(const float* __restrict__ y,
                                                           const float* __restrict__ part,
                                                           float* __restrict__ stat,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           float* __restrict__ xn, int B, int C, int H, int W,
                                                           int pool, int chunks, int CG, int BR, float eps,
                                                           int out_bn, const BnRunningFuse rf) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    double* red = reinterpret_cast<double*>(gnnpp_smem);                               // [512]
    float (*sm)[5] = reinterpret_cast<float (*)[5]>(gnnpp_smem + 512 * sizeof(double));  // mean, invstd, gamma, beta
    const int N = rf.mode == 2 ? rf.N : (int)gridDim.z;
    if (rf.mode == 1 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 8)
        rf.tick[threadIdx.x] = 0u;                               // (a later launch of this stream takes the tickets)
    if (rf.mode == 2 && (int)blockIdx.z == N) {                  // the extra plane: layers 0 .. L-2
        const int l = blockIdx.x;
        if (blockIdx.y == 0 && l < kTrainLayers - 1) {
            const int Cl = train_layer(l).Cout;
            if ((int)threadIdx.x < Cl)
                bn_running_update(rf.stat[l], rf.rmean[l], rf.rvar[l], Cl, threadIdx.x, N, rf.momentum);
            if (threadIdx.x == 0 && rf.nb[l]) *rf.nb[l] += N;
        }
        return;
    }
    const int n = blockIdx.z, c0 = blockIdx.x * CG, b0 = blockIdx.y * BR;
    const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W, Po = Ho * Wo, P = H * W;
    reduce_partials(part, n, chunks, C, c0, CG, red);
    if ((int)threadIdx.x < CG) {
        const int cc = threadIdx.x, c = c0 + cc;
        const int m = B * P;
        const double s1 = red[cc * (256 / CG) * 2], s2 = red[cc * (256 / CG) * 2 + 1];
        const double mean = s1 / m;
        double var = s2 / m - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        sm[cc][0] = (float)mean;
        sm[cc][1] = invstd;
        sm[cc][2] = gamma[c];
        sm[cc][3] = beta[c];
        if (blockIdx.y == 0) {
            float* o = stat + ((long)n * C + c) * 4;
            o[0] = (float)mean;
            o[1] = invstd;
            o[2] = (float)(m > 1 ? var * m / (m - 1) : var);
            o[3] = 0.f;
        }
    }
    __syncthreads();
    const int nb = min(BR, B - b0);
    const int per_img = CG * Po;
    for (int i = threadIdx.x; i < nb * per_img; i += 256) {
        const int bi = i / per_img, rem = i - bi * per_img;
        const int cc = rem / Po, po = rem - cc * Po;
        const float mean = sm[cc][0], invstd = sm[cc][1], g = sm[cc][2], be = sm[cc][3];
        const long ic = ((long)n * B + b0 + bi) * C + c0 + cc;
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
        // out_bn: image (n, b) of the OUTPUT sits at b*N + n (sample-major, what the graph filter reads
        // node-major) instead of n*B + b
        const long io = out_bn ? ((long)(b0 + bi) * N + n) * C + c0 + cc : ic;
        xn[io * Po + po] = v;
    }
    if (rf.mode == 2 && blockIdx.y == 0) {
        // this workgroup wrote the statistics of (agent n, channels c0 .. c0 + CG): publish them; the last of the tile's N
        // workgroups to arrive runs the N sequential updates of those channels
        const int L = kTrainLayers - 1;
        handoff_drain_stores();
        __syncthreads();
        unsigned* flag = reinterpret_cast<unsigned*>(&sm[0][0]);        // (the statistics table is dead now)
        if (threadIdx.x == 0) {
            handoff_release();
            const unsigned t = __hip_atomic_fetch_add(rf.tick + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = t == (unsigned)N - 1u;
            if (last) handoff_acquire();
            *flag = last ? 1u : 0u;
        }
        __syncthreads();
        if (*flag) {
            if ((int)threadIdx.x < CG)
                bn_running_update(rf.stat[L], rf.rmean[L], rf.rvar[L], C, c0 + threadIdx.x, N, rf.momentum);
            if (threadIdx.x == 0 && blockIdx.x == 0 && rf.nb[L]) *rf.nb[L] += N;
        }
    }
}

// ---- running statistics: the N per-agent-call updates in agent order (one thread per channel) -------------
// nn.BatchNorm2d in train mode: r <- (1 - momentum) r + momentum * batch statistic (unbiased variance)
// all five layers in one launch: blockIdx.y = layer (a = stat, b = running_mean, c = running_var)
struct TrainCounters { long long* c[kTrainLayers]; };
__global__ void bn_running_kernel(const TrainPtrs5 p, const TrainCounters nb, int N, float momentum) {
    const int l = blockIdx.y;
    if (blockIdx.x == 0 && threadIdx.x == 0 && nb.c[l]) *nb.c[l] += N;   // num_batches_tracked: N forward calls
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
// dz[c] = relu'(a) * d a, where d a = dxn[window] if this position is the FIRST maximum of its 2x2 window (scan
// order, like torch's max_pool2d backward; a recomputed from y), 0 for positions the pool never reads; without
// pool d a = dxn.  grid = (N * splits, C / 4), block = 256: wave w serves channel 4*blockIdx.y + w of agent n over
// the columns (b, pos) of split s, 64 at a time; the sums of dz and dz * yhat stay in the lanes until ONE
// butterfly per wave at the end -> part[((n*splits + s)*C + c)*2].  Writes dz [N][B][C][P].
template <int H, int W, bool pool>                               // (constant geometry: no runtime divisions)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ y,
                                                            const float* __restrict__ stat,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ dxn,
                                                            float* __restrict__ dz, float* __restrict__ part,
                                                            int B, int C, int splits, int N, int dxn_bn) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x / splits, sp = blockIdx.x - n * splits;
    const int c = blockIdx.y * 4 + wave;
    constexpr int P = H * W, Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W, Po = Ho * Wo;
    const int cols = B * P, per = ((cols + splits - 1) / splits + 63) / 64 * 64;
    const int col0 = sp * per, col1 = min(cols, col0 + per);
    const float* st = stat + ((long)n * C + c) * 4;
    const float mean = st[0], invstd = st[1], g = gamma[c], be = beta[c];
    const int wo[4] = {0, 1, W, W + 1};
    float s1 = 0.f, s2 = 0.f;
    for (int col = col0 + lane; col < col1; col += 64) {
        const int b = col / P, pos = col - b * P;
        const int py = pos / W, px = pos - py * W;
        const bool pooled_in = !pool || (py < 2 * Ho && px < 2 * Wo);   // inside the region the pool reads
        const int oy = pool ? py / 2 : py, ox = pool ? px / 2 : px;
        const long ic = ((long)n * B + b) * C + c;
        const float* yc = y + ic * P;
        const float yv = yc[pos];
        const float yhat = (yv - mean) * invstd;
        const float a = fmaxf(fmaf(yhat, g, be), 0.f);
        bool mine = pooled_in;
        if (pool && pooled_in) {
            const int wpos = (2 * oy) * W + 2 * ox;                    // window origin
            const int me = (py - 2 * oy) * 2 + (px - 2 * ox);          // my slot in the window
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float ak = bn_act(yc[wpos + wo[k]], mean, invstd, g, be);
                mine = mine && (k < me ? ak < a : ak <= a);            // strictly greater than the earlier ones
            }
        }
        const long icx = dxn_bn ? ((long)b * N + n) * C + c : ic;       // (sample-major d x_next: see bn_relu_pool)
        const float da = pooled_in ? dxn[icx * Po + oy * Wo + ox] : 0.f;
        const float d = (mine && a > 0.f) ? da : 0.f;
        dz[ic * P + pos] = d;
        s1 += d;
        s2 += d * yhat;
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
        float* o = part + (((long)n * splits + sp) * C + c) * 2;
        o[0] = s1;
        o[1] = s2;
    }
}

// pass 2: dz -> dy in place,  dy = k1 * (dz - k2 - yhat * k3),  k1 = gamma * invstd, k2 = mean(dz),
// k3 = mean(dz * yhat) over the m = B * P values of (agent, channel).  Same workgroup shape as
// bn_relu_pool_kernel: the workgroup sums pass 1's partial pairs of its channels itself; the workgroups of the
// first image range leave the per-agent sums in pn[(n*C + c)*2] for bn_bwd_dparam_kernel.
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ y,
                                                           const float* __restrict__ stat,
                                                           const float* __restrict__ part,
                                                           const float* __restrict__ gamma,
                                                           float* __restrict__ dz, float* __restrict__ pn, int B,
                                                           int C, int P, int chunks, int CG, int BR) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    double* red = reinterpret_cast<double*>(gnnpp_smem);                               // [512]
    float (*sm)[5] = reinterpret_cast<float (*)[5]>(gnnpp_smem + 512 * sizeof(double));  // mean, invstd, k1, k2, k3
    const int n = blockIdx.z, c0 = blockIdx.x * CG, b0 = blockIdx.y * BR;
    reduce_partials(part, n, chunks, C, c0, CG, red);
    if ((int)threadIdx.x < CG) {
        const int cc = threadIdx.x, c = c0 + cc;
        const int m = B * P;
        const double s1 = red[cc * (256 / CG) * 2], s2 = red[cc * (256 / CG) * 2 + 1];
        const float* st = stat + ((long)n * C + c) * 4;
        sm[cc][0] = st[0];
        sm[cc][1] = st[1];
        sm[cc][2] = gamma[c] * st[1];
        sm[cc][3] = (float)(s1 / m);
        sm[cc][4] = (float)(s2 / m);
        if (blockIdx.y == 0) {
            pn[((long)n * C + c) * 2] = (float)s1;
            pn[((long)n * C + c) * 2 + 1] = (float)s2;
        }
    }
    __syncthreads();
    const int nb = min(BR, B - b0);
    const int per_img = CG * P;                              // a contiguous run of y / dz per image
    for (int i = threadIdx.x; i < nb * per_img; i += 256) {
        const int bi = i / per_img, rem = i - bi * per_img;
        const int cc = rem / P;
        const long at = (((long)n * B + b0 + bi) * C + c0) * P + rem;
        const float yhat = (y[at] - sm[cc][0]) * sm[cc][1];
        dz[at] = sm[cc][2] * (dz[at] - sm[cc][3] - yhat * sm[cc][4]);
    }
}

// d gamma[c] = sum_n sum(dz * yhat), d beta[c] = sum_n sum(dz): agents in order, one thread per channel;
// all five layers in one launch at the end of the backward pass (blockIdx.y = layer; a = pn of the layer,
// b = d gamma, c = d beta)
__global__ void bn_bwd_dparam_kernel(const TrainPtrs5 p, int N) {
    const int l = blockIdx.y;
    const int C = train_layer(l).Cout;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float* pn = p.a[l];
    double tb = 0.0, tg = 0.0;
    for (int n = 0; n < N; ++n) {
        tb += (double)pn[((long)n * C + c) * 2];
        tg += (double)pn[((long)n * C + c) * 2 + 1];
    }
    p.b[l][c] = (float)tg;
    p.c[l][c] = (float)tb;
}

// ---- weight gradient: dW[co][j], j = ci*9 + tap (j = Cin*9: the bias column) -------------------------------
//   dW[co][j] = sum over columns (n, b, p) of dy[n,b,co,p] * X[j][(n,b,p)],  X = x[n,b,ci,p + off(tap)] or 1
// A GEMM with the columns as the contraction index on v_mfma_f32_16x16x4_f32 (A[i][k] = dy[co0+i][col k],
// B[k][j] = X[j0+j][col k], four columns per MFMA), operands staged through LDS:
//   grid = (Cout / 16, splits), block = 256.  A workgroup owns one tile of 16 output channels and a range of
//   images, which it walks IB images at a time: the raw x images go to LDS once, ZERO-BORDERED
//   ([ci][img][(H+2)*(W+2)]), so the im2col operand of ANY (ci, tap) is the same per-lane position offset plus
//   a per-j constant -- no bounds tests, no address arithmetic in the loop; dy goes to LDS as [16][img][P
//   rounded up to 4] (zero tail: a short last step contributes nothing).  Global reads are contiguous runs
//   (an image's Cin*P floats, a tile's 16*P floats) and happen once per workgroup.
//   Wave w holds the accumulators of j tiles  (w % JW) + JW*t, t < TJ  at once: one A read feeds TJ MFMAs.
//   KW > 1 (layer 0: only two j tiles): the waves also split the images of a batch KW ways, each K group
//   writing its own partial.  Partials -> wpart[split*KW + kgroup][Cout][J16], summed by conv_wgrad_reduce_kernel.
__host__ __device__ constexpr int wgrad_ib(int H) { return H == 11 ? 4 : H == 5 ? 6 : 8; }   // images per LDS batch

// (a __device__ body: one layer per launch -- conv_wgrad_kernel -- or all five layers in ONE launch --
// conv_wgrad_all_kernel; bx = output-channel tile, by = image split of the workgroup)
template <int H, int W, int Cin, int TJ, int KW>
__device__ __forceinline__ void conv_wgrad_body(const float* __restrict__ x, const float* __restrict__ dy,
                                                float* __restrict__ wpart, int NB, int Cout, long x_sn, long x_sb,
                                                int B, int imgs_per_split, int JT, int bx, int by) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    constexpr int P = H * W, PP = (H + 2) * (W + 2), SPI = (P + 3) / 4, P4 = SPI * 4, JW = 4 / KW;
    constexpr int IB = wgrad_ib(H);
    constexpr int RS = (IB * PP) | 1;                            // odd channel-row stride: the scatter of a 16-byte
                                                                 // load (2x2: four positions of consecutive channels)
                                                                 // and the B reads spread over the banks
    constexpr int dstride = ((IB * P4 + 63) / 64) * 64 + 4;      // rows 4 banks apart: conflict-free A reads
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    const int jw = wave % JW, kw = wave / JW;
    const int co0 = bx * 16, split = by;
    const int J = Cin * 9 + 1, J16 = JT * 16;
    float* xs = reinterpret_cast<float*>(gnnpp_smem);            // [Cin][RS >= IB*PP]
    float* dys = xs + Cin * RS;                                  // [16][dstride]

    int joff[TJ];
    bool jbias[TJ];
#pragma unroll
    for (int t = 0; t < TJ; ++t) {
        const int j = min(jw + JW * t, JT - 1) * 16 + i16;  // (a slot past the last tile recomputes it; not stored)
        const bool jv = j < J - 1;
        const int ci = jv ? j / 9 : 0, tap = jv ? j - ci * 9 : 4;
        joff[t] = ci * RS + (tap / 3 - 1) * (W + 2) + (tap % 3 - 1);
        jbias[t] = j == J - 1;
    }
    int ppos[SPI];
#pragma unroll
    for (int sidx = 0; sidx < SPI; ++sidx) {
        const int p = sidx * 4 + q;
        ppos[sidx] = p < P ? (p / W + 1) * (W + 2) + p % W + 1 : (W + 2) + 1;
    }
    v4f acc[TJ];
#pragma unroll
    for (int t = 0; t < TJ; ++t) acc[t] = vzero();

    // Staging in two phases (as conv_mfma_kernel): issue() puts all the loads of a batch of IB images in flight
    // -- 16-byte loads on the contiguous runs (an image's Cin*P floats, a tile's 16*P floats); the 11x11 layer's
    // observations are neither contiguous across images nor 16-byte aligned and go element-wise -- and
    // commit() scatters them into LDS.  The next batch is issued before this batch's MFMAs.
    constexpr bool kVecX = H != 11;
    constexpr int XE = IB * Cin * P, DE = IB * 16 * P;           // elements of a full batch (DE is a multiple of 4)
    constexpr int NX = kVecX ? (XE / 4 + 255) / 256 : (XE + 255) / 256, ND = (DE / 4 + 255) / 256;
    v4f xv[kVecX ? NX : 1], dv[ND];
    float xsc[kVecX ? 1 : NX];
    auto issue = [&](int base, int nimg) {
        if constexpr (kVecX) {
#pragma unroll
            for (int u = 0; u < NX; ++u) {
                const int e = (tid + u * 256) * 4;
                const int im = e / (Cin * P), r = e - im * (Cin * P);
                const bool ok = e < XE && im < nimg;
                xv[u] = *reinterpret_cast<const v4f*>(x + (ok ? (long)(base + im) * (Cin * P) + r : 0));
            }
        } else {
#pragma unroll
            for (int u = 0; u < NX; ++u) {
                const int e = tid + u * 256;
                const int im = e / (Cin * P), r = e - im * (Cin * P);
                const bool ok = e < XE && im < nimg;
                const int img = base + (ok ? im : 0), n = img / B, b = img - n * B;
                xsc[u] = x[n * x_sn + b * x_sb + (ok ? r : 0)];
            }
        }
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int e = (tid + u * 256) * 4;
            const int im = e / (16 * P), r = e - im * (16 * P);
            const bool ok = e < DE && im < nimg;
            dv[u] = *reinterpret_cast<const v4f*>(dy + (ok ? ((long)(base + im) * Cout + co0) * P + r : 0));
        }
    };
    auto commit = [&](int nimg) {
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int e = kVecX ? (tid + u * 256) * 4 : tid + u * 256;
            const int im = e / (Cin * P), r = e - im * (Cin * P);
            if (e < XE && im < nimg) {
#pragma unroll
                for (int k = 0; k < (kVecX ? 4 : 1); ++k) {
                    const int ci = (r + k) / P, p = (r + k) - ci * P;
                    float v;
                    if constexpr (kVecX) v = xv[u][k];
                    else v = xsc[u];
                    xs[ci * RS + im * PP + (p / W + 1) * (W + 2) + p % W + 1] = v;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int e = (tid + u * 256) * 4;
            const int im = e / (16 * P), r = e - im * (16 * P);
            if (e < DE && im < nimg) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = (r + k) / P, p = (r + k) - i * P;
                    dys[i * dstride + im * P4 + p] = dv[u][k];
                }
            }
        }
    };

    const int img0 = split * imgs_per_split, img1 = min(NB, img0 + imgs_per_split);
    issue(img0, min(IB, img1 - img0));
    for (int i = tid; i < Cin * RS + 16 * dstride; i += 256) xs[i] = 0.f;   // borders and tails stay zero
    __syncthreads();
    commit(min(IB, img1 - img0));
    __syncthreads();
    for (int base = img0; base < img1; base += IB) {
        const int nimg = min(IB, img1 - base);
        const int nnext = min(IB, img1 - base - IB);             // (<= 0: this is the last batch)
        if (nnext > 0) issue(base + IB, nnext);
        for (int im = kw; im < nimg; im += KW) {                 // no branches inside: the reads batch up
            const float* xi = xs + im * PP;
            const float* di = dys + i16 * dstride + im * P4 + q;
#pragma unroll
            for (int sidx = 0; sidx < SPI; ++sidx) {
                const float a = di[sidx * 4];
#pragma unroll
                for (int t = 0; t < TJ; ++t) {
                    const float bx = xi[joff[t] + ppos[sidx]];
                    acc[t] = mfma16(a, jbias[t] ? 1.f : bx, acc[t]);
                }
            }
        }
        if (nnext > 0) {
            __syncthreads();                                     // every wave is done reading this batch
            commit(nnext);
            __syncthreads();
        }
    }
    // D register r of lane l: D[i = 4 q + r][j = l & 15]
#pragma unroll
    for (int t = 0; t < TJ; ++t) {
        const int jt = jw + JW * t;
        if (jt < JT) {
            float* o = wpart + ((long)(split * KW + kw) * Cout + co0 + 4 * q) * J16 + jt * 16 + i16;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[(long)r * J16] = acc[t][r];
        }
    }
}

template <int H, int W, int Cin, int TJ, int KW>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ dy,
                                                         float* __restrict__ wpart, int NB, int Cout,
                                                         long x_sn, long x_sb, int B, int imgs_per_split, int JT) {
    conv_wgrad_body<H, W, Cin, TJ, KW>(x, dy, wpart, NB, Cout, x_sn, x_sb, B, imgs_per_split, JT, blockIdx.x, blockIdx.y);
}

// The weight gradients of ALL five layers in one launch (r06b).  Nothing downstream of a layer's weight gradient is
// needed before the optimizer, and at the reference's batch (64 x 10) each per-layer launch is ~256 workgroups of 9 .. 19 us
// of latency that fill a fraction of the chip: run one after the other inside the backward chain they cost 65 us, the fp32
// MFMAs of all five together are 14 us of the pipe.  So the backward chain only runs R -> A -> D per layer (every layer
// keeps its dy in a buffer of its own) and the five weight gradients run as ONE grid behind it: workgroup w belongs to
// layer l with first[l] <= w < first[l + 1], its (output-channel tile, image split) = ((w - first[l]) % nx[l], / nx[l]).
struct WgradAllTable {
    const float* x[kTrainLayers];
    const float* dy[kTrainLayers];
    float* wpart[kTrainLayers];
    long x_sn[kTrainLayers], x_sb[kTrainLayers];
    int ips[kTrainLayers], jt[kTrainLayers], nx[kTrainLayers], first[kTrainLayers + 1];
    int order[kTrainLayers];       // slot -> layer: the grid's slots hold the layers LONGEST workgroups first (below)
    int NB, B;
};
__global__ __launch_bounds__(256) void conv_wgrad_all_kernel(const WgradAllTable tb) {
    int sl = 0;
    while (sl + 1 < kTrainLayers && (int)blockIdx.x >= tb.first[sl + 1]) ++sl;    // (scalar: at most 4 steps)
    const int l = tb.order[sl];
    const int local = (int)blockIdx.x - tb.first[sl];
    const int bx = local % tb.nx[l], by = local / tb.nx[l];
    const int Cout = tb.nx[l] * 16;
    if (l == 0)
        conv_wgrad_body<11, 11, 3, 1, 2>(tb.x[0], tb.dy[0], tb.wpart[0], tb.NB, Cout, tb.x_sn[0], tb.x_sb[0], tb.B,
                                         tb.ips[0], tb.jt[0], bx, by);
    else if (l <= 2)
        conv_wgrad_body<5, 5, 32, 5, 1>(tb.x[l], tb.dy[l], tb.wpart[l], tb.NB, Cout, tb.x_sn[l], tb.x_sb[l], tb.B,
                                        tb.ips[l], tb.jt[l], bx, by);
    else
        conv_wgrad_body<2, 2, 64, 10, 1>(tb.x[l], tb.dy[l], tb.wpart[l], tb.NB, Cout, tb.x_sn[l], tb.x_sb[l], tb.B,
                                         tb.ips[l], tb.jt[l], bx, by);
}

// sum the splits: 8 threads per output element each add every 8th split (in order), then the 8 partial sums are
// added in order -- a fixed association, deterministic.  Thread (sub = tid >> 5, e = tid & 31): the 32 lanes of a
// half wave read 32 CONSECUTIVE outputs of one split (r05 had the 8 splits on neighbouring lanes: every lane its own
// cache line, 8.1 us per launch; same association, same bits).  dw [Cout][Cin][9], db [Cout].
// ONE launch for all five layers (r06b; r05 / r06a: one per layer, 5 us each inside the backward chain): every layer's
// partial slabs have a region of their own in the workspace, nothing downstream of a layer's weight gradient is needed
// before the optimizer, so the five reductions run once, behind the last weight-gradient kernel.  A workgroup finds its
// layer by its index (first[l] <= blockIdx.x < first[l + 1]).
struct WgradReduceTable {
    const float* wpart[kTrainLayers];
    float* dw[kTrainLayers];
    float* db[kTrainLayers];
    const float* pn[kTrainLayers];                 // bn_bwd_apply_kernel's per-agent sums of the layer
    float* dgamma[kTrainLayers];
    float* dbeta[kTrainLayers];
    int nsplit[kTrainLayers], J16[kTrainLayers], first[kTrainLayers + 1];
    int N;
};
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const WgradReduceTable tb) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    float* red = reinterpret_cast<float*>(gnnpp_smem);                 // [8][32]
    int l = 0;
    while (l + 1 < kTrainLayers && (int)blockIdx.x >= tb.first[l + 1]) ++l;       // (scalar: at most 4 steps)
    const TrainLayerDims d = train_layer(l);
    const int Cin = d.Cin, Cout = d.Cout, J16 = tb.J16[l], nsplit = tb.nsplit[l];
    const int blk = (int)blockIdx.x - tb.first[l];
    // The layer's BatchNorm parameter gradients ride along (r06; r05: a launch of their own behind the last layer):
    // d gamma[c] = sum_n sum(dz * yhat), d beta[c] = sum_n sum(dz) from bn_bwd_apply_kernel's per-agent sums pn (complete:
    // that kernel precedes this launch in stream order), agents in order, one thread per channel of the layer's LAST
    // workgroup (the one with the fewest outputs to sum).
    if (tb.pn[l] && (int)blockIdx.x == tb.first[l + 1] - 1 && (int)threadIdx.x < Cout) {
        const int c = threadIdx.x;
        const float* pn = tb.pn[l];
        double sb = 0.0, sg = 0.0;
        for (int n = 0; n < tb.N; ++n) {
            sb += (double)pn[((long)n * Cout + c) * 2];
            sg += (double)pn[((long)n * Cout + c) * 2 + 1];
        }
        tb.dgamma[l][c] = (float)sg;
        tb.dbeta[l][c] = (float)sb;
    }
    const int J = Cin * 9 + 1;
    const int total = Cout * J;
    const int e = threadIdx.x & 31, sub = threadIdx.x >> 5;
    const int i = blk * 32 + e;
    float s = 0.f;
    if (i < total) {
        const int co = i / J, jj = i - co * J;
        const float* src = tb.wpart[l] + (long)co * J16 + jj;
        const long stride = (long)Cout * J16;
#pragma unroll 4
        for (int k = sub; k < nsplit; k += 8) s += src[k * stride];
    }
    red[sub * 32 + e] = s;
    __syncthreads();
    if (sub == 0 && i < total) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k * 32 + e];
        const int co = i / J, jj = i - co * J;
        if (jj == J - 1) tb.db[l][co] = t;
        else tb.dw[l][(long)co * (J - 1) + jj] = t;
    }
}

// ---- host side: workspace layout and the two entry points -----------------------------------------------------
// Workspace (floats), for N agents and B samples per agent:
//   per layer l: y_l [N*B*Cout*P] | x_{l+1} [N*B*Cout*Po] | stat_l [N*Cout*4]
//   scratch: wt (packed forward weights, all layers) | part (partial sums) | dz x 5 (the gradient w.r.t. y_l of every
//            layer: all of them are read again by the ONE weight-gradient launch at the end) | dxa, dxb
//            (gradients w.r.t. layer inputs, ping-pong) | coef (per-agent BN-backward sums) | wpart x 5 (per layer)
struct TrainWs {
    size_t y[kTrainLayers], xn[kTrainLayers], stat[kTrainLayers], wt[kTrainLayers], wtb[kTrainLayers];
    size_t part, dz[kTrainLayers], dxa, dxb, coef, wpart[kTrainLayers], tick, total;
    int chunks[kTrainLayers];
    // conv_wgrad_kernel: image splits, images per split, images per LDS batch, j tiles, K groups per workgroup
    int nsplit[kTrainLayers], ips[kTrainLayers], ib[kTrainLayers], jt[kTrainLayers], kw[kTrainLayers];
};

// GNNPP_TUNE_TRAIN_WGRAD_WGS: workgroups per layer of the weight-gradient kernel (image splits x output-channel tiles).
// More splits = shorter workgroups but more partial slabs for conv_wgrad_reduce_kernel to sum.  0 (default) = by the
// batch: 128 up to 1 280 agent-samples, 256 beyond -- with the five layers in ONE launch the parallelism comes from
// the layers (r06b, graphed step, `profiles/r06_train_wgrad_sweep.jsonl`: 64 x 10: 64: 0.370, 96: 0.358, 128: 0.350,
// 192: 0.354, 256: 0.359, 320: 0.363, 448: 0.373 ms; 512 x 10: 128: 1.335, 256: 1.317, 512: 1.332, 1024: 1.377 ms;
// per-layer launches, r06a: 256 was best at 64 x 10).  Set it BEFORE a forward call: the workspace size depends on it.
std::atomic<int> g_train_wgrad_wgs{0};
std::atomic<int> g_train_running_fused{1};      // GNNPP_TUNE_TRAIN_RUNNING_FUSED

inline TrainWs train_ws_layout(int N, int B) {
    TrainWs w;
    size_t o = 0, max_y = 0, max_part = 0, max_x = 0, wp[kTrainLayers];
    // every region starts on a 16-byte boundary (the kernels use 16-byte loads on image runs and packs)
    auto take = [&o](size_t n) { const size_t at = o; o += (n + 3) & ~(size_t)3; return at; };
    const size_t NB = (size_t)N * B;
    for (int l = 0; l < kTrainLayers; ++l) {
        const TrainLayerDims d = train_layer(l);
        const int P = d.H * d.W, Po = d.pool ? (d.H / 2) * (d.W / 2) : P;
        w.y[l] = take(NB * d.Cout * P);
        w.xn[l] = take(NB * d.Cout * Po);
        w.stat[l] = take((size_t)N * d.Cout * 4);
        w.wt[l] = take(conv_pack_floats(l, false));
        w.wtb[l] = take(conv_pack_floats(l, true));
        {   // column splits of bn_bwd_reduce_kernel: >= ~2 500 waves per layer and at most two (r05: four) 64-column trips
            // per wave (each trip is one dependent round of loads), at least 64 columns each
            const int by_waves = (2560 + N * d.Cout - 1) / (N * d.Cout), by_trips = (B * P + 127) / 128;
            const int sp = by_waves > by_trips ? by_waves : by_trips, spmax = (B * P + 63) / 64;
            w.chunks[l] = sp < 1 ? 1 : sp > spmax ? spmax : sp > 64 ? 64 : sp;
        }
        max_y = max_y > NB * d.Cout * P ? max_y : NB * d.Cout * P;
        const int cmax = w.chunks[l] > B ? w.chunks[l] : B;          // column chunks (BN backward) vs image chunks (conv)
        const size_t pp = (size_t)N * cmax * d.Cout * 2;
        max_part = max_part > pp ? max_part : pp;
        max_x = max_x > NB * d.Cin * P ? max_x : NB * d.Cin * P;
        // weight-gradient splits: ~320 workgroups per layer (one per 16 output channels and image range)
        w.jt[l] = (d.Cin * 9 + 1 + 15) / 16;
        w.kw[l] = w.jt[l] <= 2 ? 2 : 1;
        const int knob = g_train_wgrad_wgs.load(std::memory_order_relaxed);
        const int wgs = knob > 0 ? knob : NB <= 1280 ? 128 : 256;
        int ns = (wgs + d.Cout / 16 - 1) / (d.Cout / 16);
        if ((size_t)ns > NB) ns = (int)NB;
        w.ips[l] = (int)((NB + ns - 1) / ns);
        w.nsplit[l] = (int)((NB + w.ips[l] - 1) / w.ips[l]);
        w.ib[l] = wgrad_ib(d.H);                           // LDS: [Cin][IB][(H+2)(W+2)] + [16][IB*P4] floats <= 64 KB
        wp[l] = (size_t)w.nsplit[l] * w.kw[l] * d.Cout * w.jt[l] * 16;
    }
    w.part = take(max_part);
    for (int l = 0; l < kTrainLayers; ++l) {               // dz / dy of every layer in a buffer of its own: the weight
        const TrainLayerDims d = train_layer(l);           // gradients of all layers read them in ONE launch at the end
        w.dz[l] = take(NB * d.Cout * d.H * d.W);
    }
    w.dxa = take(max_x);
    w.dxb = take(max_x);
    w.coef = take((size_t)kTrainLayers * N * 128 * 2);   // per-agent sums of the BN backward, [layer][N][128][2]
    w.tick = take(8);                                      // arrival counters of the fused running-statistics update
    for (int l = 0; l < kTrainLayers; ++l) w.wpart[l] = take(wp[l]);   // (one per layer: summed by ONE launch at the end)
    w.total = o;
    return w;
}

static inline bool launched_ok() { return hipGetLastError() == hipSuccess; }

// per-agent partial-sum chunks the forward convolution of layer l writes (= its workgroups per agent)
static int conv_chunks(int l, int B) { return conv_geom(l, false, B).chunks_per_agent; }

static void conv_launch(int l, bool input_grad, const float* x, const float* wpack, const float* bias, float* y,
                        float* part, int N, int B, long sn, long sb, hipStream_t st) {
    const ConvGeom g = conv_geom(l, input_grad, B);
    const dim3 grid(N * g.chunks_per_agent, g.M / 16);
    const size_t smem = ((size_t)g.SPC * 64 +
                         (size_t)g.CC * conv_row_stride(g.TAPS == 1 ? g.IB : g.IB * (g.H + 2) * (g.W + 2),
                                                        g.TAPS == 1)) * sizeof(float);
#define GNNPP_CONV(HH, WW)                                                                                     \
    do {                                                                                                       \
        static LdsAttrOnce once;                               /* (the dense stages use 80 KB of LDS) */     \
        set_lds_attr_once(once, reinterpret_cast<const void*>(&conv_mfma_kernel<HH, WW>), (int)smem);          \
        hipLaunchKernelGGL((conv_mfma_kernel<HH, WW>), grid, dim3(256), smem, st, x, wpack, bias, y, part, B,  \
                           g.Ck, g.M, sn, sb, g.chunks_per_agent, g.nchunk);                                   \
    } while (0)
    if (g.H == 11) GNNPP_CONV(11, 11);
    else if (g.H == 5) GNNPP_CONV(5, 5);
    else GNNPP_CONV(1, 1);
#undef GNNPP_CONV
}

static void wgrad_launch(int l, const TrainWs& L, const float* x, const float* dy, float* wpart, int NB, long sn,
                         long sb, int B, hipStream_t st) {
    const TrainLayerDims d = train_layer(l);
    const int P = d.H * d.W, PP = (d.H + 2) * (d.W + 2), P4 = (P + 3) / 4 * 4;
    const int IB = L.ib[l], dstride = ((IB * P4 + 63) / 64) * 64 + 4;
    const size_t smem = ((size_t)d.Cin * ((IB * PP) | 1) + 16 * (size_t)dstride) * sizeof(float);
    const dim3 grid(d.Cout / 16, L.nsplit[l]);
#define GNNPP_WGRAD(HH, WW, CI, TJ, KW)                                                                       \
    hipLaunchKernelGGL((conv_wgrad_kernel<HH, WW, CI, TJ, KW>), grid, dim3(256), smem, st, x, dy, wpart, NB,    \
                       d.Cout, sn, sb, B, L.ips[l], L.jt[l])
    if (l == 0) GNNPP_WGRAD(11, 11, 3, 1, 2);
    else if (d.H == 5) GNNPP_WGRAD(5, 5, 32, 5, 1);
    else GNNPP_WGRAD(2, 2, 64, 10, 1);
#undef GNNPP_WGRAD
}

// obs: [B][N][3][11][11] (the reference's inputTensor, decentralplanner.py:278-286); feat = x_5 [N][B][128], or
// [B][N][128] with feat_bn (sample-major: node-major rows for the graph filter, no transposing copy)
// the ten packs of the encoder (+ the graph filter's two, fp.h != nullptr) in one launch
int train_pack_launch(const float* const* conv_w, float* enc_pack, const TrainFilterPack& fp, hipStream_t st) {
    const TrainPackLayout PL = train_pack_layout();
    TrainPtrs5 pk = {};
    for (int l = 0; l < kTrainLayers; ++l) {
        pk.a[l] = conv_w ? conv_w[l] : nullptr;
        pk.b[l] = enc_pack ? enc_pack + PL.wt[l] : nullptr;
        pk.c[l] = enc_pack ? enc_pack + PL.wtb[l] : nullptr;
    }
    hipLaunchKernelGGL(pack_train_weights_kernel, dim3(128, 2 * kTrainLayers + (fp.h ? 2 : 0)), dim3(256), 0, st, pk, fp);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// enc_pack: the caller's pack of the CURRENT weights (gnnpp_train_pack), or nullptr: packed here, into the workspace
int train_encoder_fwd(const EncRawParams& rp, float* const* rmean, float* const* rvar,
                      long long* const* num_batches, float momentum,
                      const float* obs, float* ws, float* feat, int N, int B, int feat_bn, hipStream_t st,
                      const float* enc_pack = nullptr) {
    const TrainWs L = train_ws_layout(N, B);
    const TrainPackLayout PL = train_pack_layout();
    TrainPtrs5 pk = {}, run = {};
    const float* wt[kTrainLayers];
    for (int l = 0; l < kTrainLayers; ++l) {
        pk.a[l] = rp.conv_w[l]; pk.b[l] = ws + L.wt[l]; pk.c[l] = ws + L.wtb[l];
        run.a[l] = ws + L.stat[l]; run.b[l] = rmean ? rmean[l] : nullptr; run.c[l] = rvar ? rvar[l] : nullptr;
        wt[l] = enc_pack ? enc_pack + PL.wt[l] : ws + L.wt[l];
    }
    if (!enc_pack)
        hipLaunchKernelGGL(pack_train_weights_kernel, dim3(128, 2 * kTrainLayers), dim3(256), 0, st, pk, TrainFilterPack{});
    // GNNPP_TUNE_TRAIN_RUNNING_FUSED (default 1): the running-statistics update inside the last bn_relu_pool launch
    const BnTile tl = bn_tile(kTrainLayers - 1);
    const int last_tiles = train_layer(kTrainLayers - 1).Cout / tl.CG;
    bool have_running = rmean && rvar;                        // (update_running == 0 arrives as arrays of null pointers)
    for (int l = 0; l < kTrainLayers && have_running; ++l) have_running = rmean[l] && rvar[l];
    const bool fuse_running = have_running && g_train_running_fused.load(std::memory_order_relaxed) != 0 &&
                              last_tiles >= kTrainLayers - 1 && last_tiles <= 8;
    BnRunningFuse rf = {};
    for (int l = 0; l < kTrainLayers; ++l) {
        rf.stat[l] = ws + L.stat[l];
        rf.rmean[l] = rmean ? rmean[l] : nullptr; rf.rvar[l] = rvar ? rvar[l] : nullptr;
        rf.nb[l] = num_batches ? num_batches[l] : nullptr;
    }
    rf.tick = reinterpret_cast<unsigned*>(ws + L.tick);
    rf.momentum = momentum; rf.N = N;
    for (int l = 0; l < kTrainLayers; ++l) {
        const TrainLayerDims d = train_layer(l);
        const int P = d.H * d.W;
        const float* xin = l == 0 ? obs : ws + L.xn[l - 1];
        const long sn = l == 0 ? (long)d.Cin * P : (long)B * d.Cin * P;        // obs is [B][N]: n is the inner index
        const long sb = l == 0 ? (long)N * d.Cin * P : (long)d.Cin * P;
        conv_launch(l, false, xin, wt[l], rp.conv_b[l], ws + L.y[l], ws + L.part, N, B, sn, sb, st);
        const BnTile t = bn_tile(l);
        BnRunningFuse f = {};
        const bool last = l == kTrainLayers - 1;
        if (fuse_running && (l == 0 || last)) {
            f = rf;
            f.mode = last ? 2 : 1;
        }
        hipLaunchKernelGGL(bn_relu_pool_kernel, dim3(d.Cout / t.CG, (B + t.BR - 1) / t.BR, N + (f.mode == 2 ? 1 : 0)),
                           dim3(256), kBnSmem, st, ws + L.y[l], ws + L.part, ws + L.stat[l], rp.bn_w[l], rp.bn_b[l],
                           last ? feat : ws + L.xn[l], B, d.Cout, d.H, d.W, d.pool,
                           conv_chunks(l, B), t.CG, t.BR, rp.bn_eps, (last && feat_bn) ? 1 : 0, f);
    }
    if (have_running && !fuse_running) {
        TrainCounters nb = {};
        for (int l = 0; l < kTrainLayers; ++l) nb.c[l] = num_batches ? num_batches[l] : nullptr;
        hipLaunchKernelGGL(bn_running_kernel, dim3(1, kTrainLayers), dim3(128), 0, st, run, nb, N, momentum);
    }
    return launched_ok() ? 0 : -3;
}

// ---- the weight-gradient branch of the backward pass on a second stream (r05) ------------------------------------
// Per layer the backward chain is  R (bn_bwd_reduce) -> A (bn_bwd_apply) -> { D: input gradient -> layer l - 1 }
//                                                                          { W: weight gradient -> Wr: its reduction }
// and nothing downstream of W / Wr is needed before the optimizer.  On one stream the ~20 launches run back to back
// (257 us at 64 x 10, most of them 5 .. 20 us kernels that fill a fraction of the chip for a fraction of their time:
// profiles/r05_train_kernel_stats.csv); here W / Wr of every layer go to a SIDE stream that forks behind A(l) and
// joins in front of the parameter-gradient kernel -- also inside a stream capture, where the fork / join events
// become edges of the HIP graph.  dz is double-buffered so that R(l - 1) does not wait for W(l); R(l - 2), which
// re-uses W(l)'s buffer, waits for W(l)'s event.  Measured (profiles/r05_train_fork_ab.jsonl, same kernels, bit-identical
// gradients): 512 x 10: eager 1.626 -> 1.514 ms, graphed 1.633 -> 1.612 ms; 64 x 10 (the per-GPU shard of config 4): graphed
// 0.486 -> 0.566 ms, eager 1.158 -> 1.270 ms -- eleven cross-stream edges cost more than the overlap of 5 .. 20 us kernels
// buys.  GNNPP_TUNE_TRAIN_FORK: 1 (default) = fork from kTrainForkMinRows agent-samples on, 0 = never, 2 = always.
std::atomic<int> g_train_fork{1};
std::atomic<int> g_train_wgrad_merged{1};       // GNNPP_TUNE_TRAIN_WGRAD_MERGED
constexpr long kTrainForkMinRows = 4096;

struct BwdFork {
    hipStream_t side = nullptr;
    hipEvent_t dz_ready[kTrainLayers] = {}, w_done[kTrainLayers] = {}, join = nullptr;
    bool ok = false;
    // The side stream and its events are ONE set per device.  Two backward calls enqueued at the same time from two
    // host threads (two models training on two streams) would interleave their record / wait pairs on the shared
    // events -- a wait would bind to the other call's record --, so a call holds this lock while it ENQUEUES (host
    // side only: microseconds); the calls' side work then simply queues up on the one side stream.
    std::mutex enqueue;
};

// one per device, created on first use OUTSIDE a stream capture (resource creation is not a capturable operation);
// a call that arrives while `st` is capturing before any eager call has run stays on one stream
static BwdFork* bwd_fork(hipStream_t st, long rows) {
    static std::mutex mu;
    static BwdFork forks[16];
    const int knob = g_train_fork.load(std::memory_order_relaxed);
    if (knob == 0 || (knob == 1 && rows < kTrainForkMinRows)) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    {   // the side stream / events are per DEVICE: they must be the ones of the device `st` belongs to; a
        // stream of another device than the current one stays on one stream
        hipDevice_t sdev;
        if (st != nullptr && hipStreamGetDevice(st, &sdev) == hipSuccess && (int)sdev != dev) return nullptr;
        (void)hipGetLastError();
    }
    std::lock_guard<std::mutex> lock(mu);
    BwdFork& f = forks[dev];
    if (!f.ok) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
        bool good = hipStreamCreateWithFlags(&f.side, hipStreamNonBlocking) == hipSuccess;
        for (int l = 0; l < kTrainLayers && good; ++l)
            good = hipEventCreateWithFlags(&f.dz_ready[l], hipEventDisableTiming) == hipSuccess &&
                   hipEventCreateWithFlags(&f.w_done[l], hipEventDisableTiming) == hipSuccess;
        good = good && hipEventCreateWithFlags(&f.join, hipEventDisableTiming) == hipSuccess;
        if (!good) { (void)hipGetLastError(); return nullptr; }
        f.ok = true;
    }
    return &f;
}

// dfeat: gradient w.r.t. x_5 [N][B][128]; writes d conv_w / d conv_b / d bn_w / d bn_b of every layer
int train_encoder_bwd(const EncRawParams& rp, const float* obs, float* ws, const float* dfeat,
                      float* const* dconv_w, float* const* dconv_b, float* const* dbn_w, float* const* dbn_b,
                      int N, int B, int feat_bn, hipStream_t st, const float* enc_pack = nullptr) {
    const TrainWs L = train_ws_layout(N, B);
    const TrainPackLayout PL = train_pack_layout();
    const long NB = (long)N * B;
    const float* dxn = dfeat;
    float* dx_buf[2] = {ws + L.dxa, ws + L.dxb};
    TrainPtrs5 dp = {};
    // GNNPP_TUNE_TRAIN_WGRAD_MERGED (default 1): the weight gradients of all five layers as ONE launch behind the chain
    // (conv_wgrad_all_kernel); 0: one launch per layer inside the chain, where the fork rule below can put them on a
    // side stream (r05).  Same kernels' bodies, same partial layout, same reduction: bit-identical gradients.
    const bool merged = g_train_wgrad_merged.load(std::memory_order_relaxed) != 0;
    BwdFork* const fk = merged ? nullptr : bwd_fork(st, NB);
    bool fork_ok = true;
    WgradReduceTable rt = {};
    WgradAllTable wa = {};
    hipStream_t const sw = fk ? fk->side : st;              // where the weight-gradient branch runs
    std::unique_lock<std::mutex> enqueue_lock;
    if (fk) enqueue_lock = std::unique_lock<std::mutex>(fk->enqueue);
    for (int l = kTrainLayers - 1; l >= 0; --l) {
        const TrainLayerDims d = train_layer(l);
        const int P = d.H * d.W;
        float* dz = ws + L.dz[l];                            // (every layer its own: no buffer is re-used inside the call)
#define GNNPP_BNR(HH, WW, PL)                                                                                    \
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<HH, WW, PL>), dim3(N * L.chunks[l], d.Cout / 4), dim3(256), 0, st,  \
                       ws + L.y[l], ws + L.stat[l], rp.bn_w[l], rp.bn_b[l], dxn, dz, ws + L.part, B, d.Cout,    \
                       L.chunks[l], N, (l == kTrainLayers - 1 && feat_bn) ? 1 : 0)
        if (d.H == 11) GNNPP_BNR(11, 11, true);
        else if (d.H == 5 && d.pool) GNNPP_BNR(5, 5, true);
        else if (d.H == 5) GNNPP_BNR(5, 5, false);
        else if (d.pool) GNNPP_BNR(2, 2, true);
        else GNNPP_BNR(2, 2, false);
#undef GNNPP_BNR
        const BnTile t = bn_tile(l);
        float* pn = ws + L.coef + (size_t)l * N * 128 * 2;
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(d.Cout / t.CG, (B + t.BR - 1) / t.BR, N), dim3(256), kBnSmem, st,
                           ws + L.y[l], ws + L.stat[l], ws + L.part, rp.bn_w[l], dz, pn, B, d.Cout, P, L.chunks[l],
                           t.CG, t.BR);
        dp.a[l] = pn; dp.b[l] = dbn_w[l]; dp.c[l] = dbn_b[l];
        const float* xin = l == 0 ? obs : ws + L.xn[l - 1];
        const long sn = l == 0 ? (long)d.Cin * P : (long)B * d.Cin * P;
        const long sb = l == 0 ? (long)N * d.Cin * P : (long)d.Cin * P;
        if (fk) {                                            // fork: the side stream waits for dz of this layer
            fork_ok &= hipEventRecord(fk->dz_ready[l], st) == hipSuccess;
            fork_ok &= hipStreamWaitEvent(sw, fk->dz_ready[l], 0) == hipSuccess;
        }
        if (merged) {
            wa.x[l] = xin; wa.dy[l] = dz; wa.wpart[l] = ws + L.wpart[l]; wa.x_sn[l] = sn; wa.x_sb[l] = sb;
            wa.ips[l] = L.ips[l]; wa.jt[l] = L.jt[l]; wa.nx[l] = d.Cout / 16;
        } else {
            wgrad_launch(l, L, xin, dz, ws + L.wpart[l], (int)NB, sn, sb, B, sw);
        }
        rt.wpart[l] = ws + L.wpart[l]; rt.dw[l] = dconv_w[l]; rt.db[l] = dconv_b[l];
        rt.pn[l] = pn; rt.dgamma[l] = dbn_w[l]; rt.dbeta[l] = dbn_b[l];
        rt.nsplit[l] = L.nsplit[l] * L.kw[l]; rt.J16[l] = L.jt[l] * 16;
        if (l > 0) {
            // dx [N][B][Cin][P] = conv(dy) with the flipped kernel; here "Cin" of the call = Cout of the layer
            float* dx = dx_buf[l & 1];
            // output channels of this call = d.Cin (a multiple of 16 for l >= 1)
            conv_launch(l, true, dz, enc_pack ? enc_pack + PL.wtb[l] : ws + L.wtb[l], nullptr, dx, nullptr, N, B, (long)B * d.Cout * P,
                        (long)d.Cout * P, st);
            dxn = dx;
        }
    }
    if (merged) {                                            // the five weight gradients: ONE grid
        int blocks = 0;
        size_t smem = 0;
        // slots in the order of the workgroups' durations, longest first (measured per-layer launches at 64 x 10: layer 2
        // 19 us, 4: 13, 1: 12, 0: 11, 3: 9): the grid is ~2 rounds of the chip and the dispatcher hands out workgroups in
        // index order -- short workgroups at the end fill the tail instead of long ones starting last
        const int order[kTrainLayers] = {2, 4, 1, 0, 3};
        for (int sl = 0; sl < kTrainLayers; ++sl) {
            const int l = order[sl];
            const TrainLayerDims d = train_layer(l);
            const int P = d.H * d.W, PP = (d.H + 2) * (d.W + 2), P4 = (P + 3) / 4 * 4;
            const int IB = L.ib[l], dstride = ((IB * P4 + 63) / 64) * 64 + 4;
            const size_t sm = ((size_t)d.Cin * ((IB * PP) | 1) + 16 * (size_t)dstride) * sizeof(float);
            smem = smem > sm ? smem : sm;
            wa.order[sl] = l;
            wa.first[sl] = blocks;
            blocks += wa.nx[l] * L.nsplit[l];
        }
        wa.first[kTrainLayers] = blocks;
        wa.NB = (int)NB; wa.B = B;
        hipLaunchKernelGGL(conv_wgrad_all_kernel, dim3(blocks), dim3(256), smem, st, wa);
    }
    {   // the five layers' split sums (+ d gamma / d beta) in ONE launch, behind the last weight-gradient kernel
        int blocks = 0;
        for (int l = 0; l < kTrainLayers; ++l) {
            const TrainLayerDims d = train_layer(l);
            rt.first[l] = blocks;
            blocks += (d.Cout * (d.Cin * 9 + 1) + 31) / 32;
        }
        rt.first[kTrainLayers] = blocks;
        rt.N = N;
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(blocks), dim3(256), 256 * sizeof(float), sw, rt);
    }
    if (fk) {                                                // join: everything behind this call sees every gradient
        fork_ok &= hipEventRecord(fk->join, sw) == hipSuccess;
        fork_ok &= hipStreamWaitEvent(st, fk->join, 0) == hipSuccess;
        // a failed record / wait (a capture-mode restriction, a stream of another device) would silently drop
        // the dz_ready / w_done ordering the ping-pong dz buffers rely on: the call fails instead of racing
        if (!fork_ok) { (void)hipGetLastError(); return -3; }
    }
    (void)dp;                                                // (d gamma / d beta: inside conv_wgrad_reduce_kernel since r06)
    return launched_ok() ? 0 : -3;
}

}  // namespace gnnpp
