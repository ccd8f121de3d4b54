// Per-agent CNN encoder + compress MLP for gfx950: shared geometry, weight packing and dispatch.
//
// Replaces ConvLayers (5 x [conv3x3 pad1 -> BatchNorm(eval) -> ReLU], MaxPool2d(2) after layers
// 0, 2, 4) and compressMLP (Linear 128->128 + ReLU) as the reference runs them once per agent
// (graphs/models/decentralplanner.py:284-290, layers built at :155-195).  In eval mode every
// agent of every sample is independent, so all M = B*N agents are folded into one batch.
//
// All three schedules (encoder_kernel_b3.hip = bf16x3, the fp32-equivalent default; encoder_kernel_f32.hip = exact
// fp32 MFMA; encoder_kernel_h2.hip = split-f16, opt-in) share this plan: one workgroup (4 waves) owns a tile of 16 agents and carries them through all
// six layers with the activations never leaving LDS:
//
//   obs   [16][3][12][12]  zero-padded on top/left only                 (27.7 KB, buffer Y)
//   L0    3 -> 32  @ 11x11 (only the 10x10 the pool reads) -> pool -> 32 @ 5x5     (buffer X)
//   L1   32 -> 32  @ 5x5                                                            (X, in place)
//   L2   32 -> 64  @ 5x5  (only the 4x4 the pool reads)    -> pool -> 64 @ 2x2
//   L3   64 -> 64  @ 2x2
//   L4   64 -> 128 @ 2x2                                   -> pool -> 128 @ 1x1
//   FC  128 -> 128 + ReLU  -> feat[agent][128] in HBM (node-major, what the filter kernel reads)
//
// LDS per workgroup is 50 KB + 27.7 KB = 78.9 KB, so TWO workgroups share a CU (the second hides
// the first's barrier, staging and weight-latency bubbles).
//
// Every layer is an implicit GEMM on the MFMA (gnnpp_common.h): output channels on the MFMA i axis
// (weights = A operand, pre-packed fragments streamed from L2), the 16 agents on the j axis
// (activations = B operand from LDS), one MFMA tile per OUTPUT POSITION.  Because a tile is a single
// spatial position, zero padding is resolved at compile time: taps that fall outside the image are
// simply not issued (25 % of L1/L2's and 56 % of L3/L4's nominal MACs), and positions the following
// MaxPool discards are never computed.  Results are unchanged (x + 0*w == x); the algorithmic FLOP
// count used for the roofline is the reference's nominal one.
// BatchNorm(eval) is folded into a per-channel scale/shift applied in the epilogue.
#include "gnnpp_common.h"

namespace gnnpp {

constexpr int kTileAgents = 16;
constexpr int kObsFloats = 3 * 11 * 11;          // 363
constexpr int kPadHW = 12;                       // 11 + one zero row/column on top/left only:
                                                 // outputs 0..9 never read below/right of row 10
constexpr int kAgentStride = 433;                // 3*12*12 = 432 -> 433 (odd: 16 lanes, 16 banks)
constexpr int kBufFloats = 25 * 2 * 256;         // largest activation: 25 positions x 32 channels
constexpr int kObsFloatsLds = kTileAgents * kAgentStride;            // 6928 floats = 27712 B
static_assert(kObsFloatsLds % 4 == 0 && kObsFloatsLds <= kBufFloats, "obs staging buffer");

// ---- weight packing (device side; inputs are the reference's state_dict tensors) --------------
struct EncRawParams {
    const float* conv_w[5];
    const float* conv_b[5];
    const float* bn_w[5];
    const float* bn_b[5];
    const float* bn_mean[5];
    const float* bn_var[5];
    const float* fc_w;
    const float* fc_b;
    float bn_eps;
};

__device__ __forceinline__ int enc_w_off(int layer) {
    return layer == 0 ? EncLayout::kW0 : layer == 1 ? EncLayout::kW1 : layer == 2 ? EncLayout::kW2
         : layer == 3 ? EncLayout::kW3 : layer == 4 ? EncLayout::kW4 : EncLayout::kWfc;
}
__device__ __forceinline__ int enc_ss_off(int layer) {
    return layer == 0 ? EncLayout::kSS0 : layer == 1 ? EncLayout::kSS1 : layer == 2 ? EncLayout::kSS2
         : layer == 3 ? EncLayout::kSS3 : EncLayout::kSS4;
}

// Winograd F(2x2,3x3) weight transform, element (a,b) of U = G g G^T with
// G = [[1,0,0],[1/2,1/2,1/2],[1/2,-1/2,1/2],[0,0,1]];  g = 3x3 kernel (row-major, cross-correlation).
__device__ __forceinline__ float winograd_u(const float* __restrict__ g, int a, int b) {
    float t[3];                                                  // row a of (G g)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float g0 = g[c], g1 = g[3 + c], g2 = g[6 + c];
        t[c] = a == 0 ? g0 : a == 1 ? 0.5f * (g0 + g1 + g2) : a == 2 ? 0.5f * (g0 - g1 + g2) : g2;
    }
    return b == 0 ? t[0] : b == 1 ? 0.5f * (t[0] + t[1] + t[2]) : b == 2 ? 0.5f * (t[0] - t[1] + t[2])
                                                                        : t[2];
}

// ---- split-f16 path: per-layer power-of-two weight scale ---------------------------------------
// Block b handles layer b + 1 (conv 1..4; b = 4: the FC; b = 5: conv 0).  2^k is chosen so that max|w| * 2^k lies in
// [512, 1024): the hi halves stay far from f16 overflow and the lo halves (w*2^k - hi, ~2^-11 of
// hi) of all but negligible weights are normal f16 numbers.  The kernel undoes 2^k exactly in its
// epilogue (folded into the BatchNorm scale).
__device__ __forceinline__ void enc_h2_layer(int b, const EncRawParams& rp, const float*& w, int& n) {
    w = b < 4 ? rp.conv_w[b + 1] : b == 4 ? rp.fc_w : rp.conv_w[0];
    n = b == 0 ? 32 * 32 * 9 : b == 1 ? 64 * 32 * 9 : b == 2 ? 64 * 64 * 9 : b == 3 ? 128 * 64 * 9
      : b == 4 ? 128 * 128 : 32 * 3 * 9;
}

__global__ void enc_layer_scale_kernel(const EncRawParams rp, float* __restrict__ packed) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    float* red = reinterpret_cast<float*>(gnnpp_smem);
    const float* w;
    int n;
    enc_h2_layer(blockIdx.x, rp, w, n);
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, fabsf(w[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {          // tree reduction (blockDim.x = 2^k)
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        m = red[0];
        int k = 0;
        if (m > 0.f && m < 3.0e38f) {
            int e;
            (void)frexpf(m, &e);                       // m = f * 2^e, f in [0.5, 1)
            k = min(max(10 - e, -60), 60);
        }
        packed[EncLayout::kHscale + blockIdx.x] = ldexpf(1.f, k);
        packed[EncLayout::kHinv + blockIdx.x] = ldexpf(1.f, -k);
    }
}

// One split-f16 A fragment element pair: packed[...] holds two halves per float slot.
// Layout of a layer: [group][kb][tap][mt_local][hi/lo][lane 64][e 8];  mt = group * NMTL + mt_local,
// half e of lane (q, i) = W[cout = 16 mt + i][cin = 32 kb + 16 (e >> 2) + 4 q + (e & 3)][tap] * 2^k.
__device__ __forceinline__ void enc_h2_pack_layer(const float* __restrict__ w, float scale, int cin,
                                                  int ntap, int ngroup, int nmtl, int nkb,
                                                  float* __restrict__ dst, int t0, int stride) {
    _Float16* out = reinterpret_cast<_Float16*>(dst);
    const int total = ngroup * nkb * ntap * nmtl * 512;           // (lane, e) pairs per hi/lo
    for (int idx = t0; idx < total; idx += stride) {
        const int e = idx & 7, l = (idx >> 3) & 63;
        int blk = idx >> 9;
        const int ml = blk % nmtl; blk /= nmtl;
        const int tap = blk % ntap; blk /= ntap;
        const int kb = blk % nkb;
        const int grp = blk / nkb;
        const int co = (grp * nmtl + ml) * 16 + (l & 15);
        const int ci = 32 * kb + 16 * (e >> 2) + 4 * (l >> 4) + (e & 3);
        const float v = w[((size_t)co * cin + ci) * ntap + tap] * scale;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        const size_t item = (size_t)(idx >> 9) * 2;               // hi item, lo item follows
        out[(item * 64 + l) * 8 + e] = hi;
        out[((item + 1) * 64 + l) * 8 + e] = lo;
    }
}

// bf16x3 A fragments (gnnpp_common.h, "b3"): w = h + m + l exactly, no scale.  Layout of a layer:
// [group][kb][tap][mt_local][plane 3][lane 64][e 8], the same channel order as the split-f16 fragments.
__device__ __forceinline__ void enc_b3_pack_layer(const float* __restrict__ w, int cin, int ntap, int ngroup,
                                                  int nmtl, int nkb, float* __restrict__ dst, int t0, int stride) {
    unsigned short* out = reinterpret_cast<unsigned short*>(dst);
    const int total = ngroup * nkb * ntap * nmtl * 512;           // (lane, e) pairs per plane
    for (int idx = t0; idx < total; idx += stride) {
        const int e = idx & 7, l = (idx >> 3) & 63;
        int blk = idx >> 9;
        const int ml = blk % nmtl; blk /= nmtl;
        const int tap = blk % ntap; blk /= ntap;
        const int kb = blk % nkb;
        const int grp = blk / nkb;
        const int co = (grp * nmtl + ml) * 16 + (l & 15);
        const int ci = 32 * kb + 16 * (e >> 2) + 4 * (l >> 4) + (e & 3);
        const float v = w[((size_t)co * cin + ci) * ntap + tap];
        unsigned h, m, lo;
        b3_split2(v, 0.f, h, m, lo);
        const size_t item = (size_t)(idx >> 9) * 3;               // h item, then m, then l
        out[(item * 64 + l) * 8 + e] = (unsigned short)(h & 0xffffu);
        out[((item + 1) * 64 + l) * 8 + e] = (unsigned short)(m & 0xffffu);
        out[((item + 2) * 64 + l) * 8 + e] = (unsigned short)(lo & 0xffffu);
    }
}

__global__ void pack_encoder_kernel(const EncRawParams rp, float* __restrict__ packed) {
    const int stride = gridDim.x * blockDim.x;
    const int t0 = blockIdx.x * blockDim.x + threadIdx.x;
    // bf16x3 fragments of the default (fp32-equivalent) schedule
    enc_b3_pack_layer(rp.conv_w[1], 32, 9, 1, 2, 1, packed + EncLayout::kB1, t0, stride);
    enc_b3_pack_layer(rp.conv_w[2], 32, 9, 2, 2, 1, packed + EncLayout::kB2, t0, stride);
    enc_b3_pack_layer(rp.conv_w[3], 64, 9, 4, 1, 2, packed + EncLayout::kB3, t0, stride);
    enc_b3_pack_layer(rp.conv_w[4], 64, 9, 4, 2, 2, packed + EncLayout::kB4, t0, stride);
    enc_b3_pack_layer(rp.fc_w, 128, 1, 4, 2, 4, packed + EncLayout::kBfcw, t0, stride);
    {   // L0: [mt 2][plane 3][lane 64][e 8], k-slots as the split-f16 L0 block below
        unsigned short* out = reinterpret_cast<unsigned short*>(packed + EncLayout::kB0);
        for (int idx = t0; idx < 2 * 512; idx += stride) {
            const int e = idx & 7, l = (idx >> 3) & 63, mt = idx >> 9, qq = l >> 4;
            const int co = mt * 16 + (l & 15);
            float v = 0.f;
            if (qq < 3) v = rp.conv_w[0][(co * 3 + qq) * 9 + e];
            else if (e < 3) v = rp.conv_w[0][(co * 3 + e) * 9 + 8];
            // L0 is pooled on its RAW accumulators: max_i (s c_i + b) = |s| max_i (sgn(s) c_i) + b (fma is monotone
            // in c for s >= 0), so the sign of the folded BatchNorm scale goes into the weights (an exact negation)
            // and the table holds |s| -- the same value to the bit, three VALU operations per output fewer
            if (rp.bn_w[0][co] / sqrtf(rp.bn_var[0][co] + rp.bn_eps) < 0.f) v = -v;
            unsigned h, m, lo;
            b3_split2(v, 0.f, h, m, lo);
            out[((mt * 3 + 0) * 64 + l) * 8 + e] = (unsigned short)(h & 0xffffu);
            out[((mt * 3 + 1) * 64 + l) * 8 + e] = (unsigned short)(m & 0xffffu);
            out[((mt * 3 + 2) * 64 + l) * 8 + e] = (unsigned short)(lo & 0xffffu);
        }
    }
    // split-f16 fragments (scales were written by enc_layer_scale_kernel, earlier on this stream)
    enc_h2_pack_layer(rp.conv_w[1], packed[EncLayout::kHscale + 0], 32, 9, 1, 2, 1,
                      packed + EncLayout::kH1, t0, stride);
    enc_h2_pack_layer(rp.conv_w[2], packed[EncLayout::kHscale + 1], 32, 9, 2, 2, 1,
                      packed + EncLayout::kH2, t0, stride);
    enc_h2_pack_layer(rp.conv_w[3], packed[EncLayout::kHscale + 2], 64, 9, 4, 1, 2,
                      packed + EncLayout::kH3, t0, stride);
    enc_h2_pack_layer(rp.conv_w[4], packed[EncLayout::kHscale + 3], 64, 9, 4, 2, 2,
                      packed + EncLayout::kH4, t0, stride);
    enc_h2_pack_layer(rp.fc_w, packed[EncLayout::kHscale + 4], 128, 1, 4, 2, 4,
                      packed + EncLayout::kHfc, t0, stride);
    {   // L0: [mt 2][hi/lo][lane 64][e 8]
        _Float16* out = reinterpret_cast<_Float16*>(packed + EncLayout::kH0);
        const float scale = packed[EncLayout::kHscale + 5];
        for (int idx = t0; idx < 2 * 512; idx += stride) {
            const int e = idx & 7, l = (idx >> 3) & 63, mt = idx >> 9, qq = l >> 4;
            const int co = mt * 16 + (l & 15);
            float v = 0.f;
            if (qq < 3) v = rp.conv_w[0][(co * 3 + qq) * 9 + e];
            else if (e < 3) v = rp.conv_w[0][(co * 3 + e) * 9 + 8];
            v *= scale;
            const _Float16 hi = (_Float16)v;
            out[((mt * 2 + 0) * 64 + l) * 8 + e] = hi;
            out[((mt * 2 + 1) * 64 + l) * 8 + e] = (_Float16)(v - (float)hi);
        }
    }
    // Winograd L0: [mt 2][wpos 16][lane 64] = U[cout = mt*16+i][cin = q][wpos], 0 for q = 3
    for (int idx = t0; idx < 2 * 16 * 64; idx += stride) {
        const int l = idx & 63, wp = (idx >> 6) & 15, mt = idx >> 10;
        const int co = mt * 16 + (l & 15), ci = l >> 4;
        packed[EncLayout::kU0 + idx] =
            ci < 3 ? winograd_u(rp.conv_w[0] + (co * 3 + ci) * 9, wp >> 2, wp & 3) : 0.f;
    }
    // Winograd L2: [mt 4][g 2][wpos 16][lane 64][s 4] = U[cout][cin = g*16 + q*4 + s][wpos]
    for (int idx = t0; idx < 4 * 2 * 16 * 256; idx += stride) {
        const int s = idx & 3, l = (idx >> 2) & 63, wp = (idx >> 8) & 15, g = (idx >> 12) & 1,
                  mt = idx >> 13;
        const int co = mt * 16 + (l & 15), ci = g * 16 + (l >> 4) * 4 + s;
        packed[EncLayout::kU2 + idx] = winograd_u(rp.conv_w[2] + (co * 32 + ci) * 9, wp >> 2, wp & 3);
    }
    // L0: [mt 2][s 7][lane 64];  k = 4*s + q  ->  (c, ky, kx) = (k/9, (k%9)/3, k%3);  k = 27 -> 0
    for (int idx = t0; idx < 2 * 7 * 64; idx += stride) {
        const int l = idx & 63, s = (idx >> 6) % 7, mt = idx / (7 * 64);
        const int k = 4 * s + (l >> 4);
        const int cout = mt * 16 + (l & 15);
        packed[EncLayout::kW0 + idx] = (k < 27) ? rp.conv_w[0][cout * 27 + k] : 0.f;
    }
    // L1..L4: [mt][tap 9][g][lane 64][s 4] = W[cout = mt*16+i][cin = g*16 + q*4 + s][tap]
    for (int layer = 1; layer < 5; ++layer) {
        const int cin = layer <= 2 ? 32 : 64;
        const int cout_n = layer == 1 ? 32 : layer <= 3 ? 64 : 128;
        const int NG = cin / 16, total = cout_n * cin * 9;
        const int off = enc_w_off(layer);
        for (int idx = t0; idx < total; idx += stride) {
            const int s = idx & 3, l = (idx >> 2) & 63;
            int blk = idx >> 8;
            const int g = blk % NG; blk /= NG;
            const int tap = blk % 9;
            const int mt = blk / 9;
            const int co = mt * 16 + (l & 15);
            const int ci = g * 16 + (l >> 4) * 4 + s;
            packed[off + idx] = rp.conv_w[layer][(co * cin + ci) * 9 + tap];
        }
    }
    // FC: [mt 8][g 8][lane 64][s 4] = W[f = mt*16+i][c = g*16 + q*4 + s]
    for (int idx = t0; idx < 128 * 128; idx += stride) {
        const int s = idx & 3, l = (idx >> 2) & 63;
        const int blk = idx >> 8;
        const int g = blk & 7, mt = blk >> 3;
        packed[EncLayout::kWfc + idx] =
            rp.fc_w[(mt * 16 + (l & 15)) * 128 + g * 16 + (l >> 4) * 4 + s];
    }
    for (int idx = t0; idx < 128; idx += stride) packed[EncLayout::kBfc + idx] = rp.fc_b[idx];
    // folded BatchNorm: y = conv_nobias * scale + shift,
    //   scale = gamma / sqrt(var + eps),  shift = beta + (conv_bias - mean) * scale
    for (int layer = 0; layer < 5; ++layer) {
        const int c_n = layer <= 1 ? 32 : layer <= 3 ? 64 : 128;
        const int off = enc_ss_off(layer);
        for (int c = t0; c < c_n; c += stride) {
            const float sc = rp.bn_w[layer][c] / sqrtf(rp.bn_var[layer][c] + rp.bn_eps);
            const float shf = rp.bn_b[layer][c] + (rp.conv_b[layer][c] - rp.bn_mean[layer][c]) * sc;
            packed[off + c] = sc;
            packed[off + c_n + c] = shf;
            // split-f16 path: the accumulators carry the weight scale 2^k of their layer
            const float inv = packed[EncLayout::kHinv + (layer == 0 ? 5 : layer - 1)];
            const int hoff = layer == 0 ? EncLayout::kHss0
                           : EncLayout::kHss + (layer == 1 ? EncLayout::kHssL1 : layer == 2 ? EncLayout::kHssL2
                                                : layer == 3 ? EncLayout::kHssL3 : EncLayout::kHssL4);
            packed[hoff + c] = sc * inv;
            packed[hoff + c_n + c] = shf;
            // bf16x3 path: unscaled, one contiguous table
            const int boff = EncLayout::kBss + (layer == 0 ? EncLayout::kBssL0 : layer == 1 ? EncLayout::kBssL1
                                                : layer == 2 ? EncLayout::kBssL2 : layer == 3 ? EncLayout::kBssL3
                                                : EncLayout::kBssL4);
            packed[boff + c] = layer == 0 ? fabsf(sc) : sc;     // (L0: the sign lives in the bf16x3 weights, see above)
            packed[boff + c_n + c] = shf;
        }
    }
}

// L1: 25 positions split in two halves (13 + 12) so that 2 channel tiles x 2 halves = 4 waves.
template <int PART>
struct PosL1 {
    static __device__ __forceinline__ bool get(int j, int& y, int& x) {
        const int p = PART * 13 + j;
        y = p / 5; x = p % 5;
        return p < 25;
    }
};
// L2: the 4x4 block the pool reads, slot = window*4 + (py*2+px).
struct PosL2 {
    static __device__ __forceinline__ bool get(int j, int& y, int& x) {
        const int w = j >> 2, i = j & 3;
        y = 2 * (w >> 1) + (i >> 1); x = 2 * (w & 1) + (i & 1);
        return true;
    }
};
struct Pos2x2 {
    static __device__ __forceinline__ bool get(int j, int& y, int& x) {
        y = j >> 1; x = j & 1;
        return true;
    }
};

__device__ __forceinline__ void load_ss(const float* __restrict__ ss, int cn, int mt, int q,
                                        v4f& sc, v4f& sh) {
    sc = *reinterpret_cast<const v4f*>(ss + mt * 16 + q * 4);
    sh = *reinterpret_cast<const v4f*>(ss + cn + mt * 16 + q * 4);
}

// ---- host-side launchers ----------------------------------------------------------------------
int encoder_pack_launch(const EncRawParams& rp, float* packed, hipStream_t st) {
    hipLaunchKernelGGL(enc_layer_scale_kernel, dim3(6), dim3(1024), 1024 * sizeof(float), st, rp, packed);
    hipLaunchKernelGGL(pack_encoder_kernel, dim3(128), dim3(256), 0, st, rp, packed);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// Arithmetic of a call (include/gnnpp.h GNNPP_PREC_*): 0 = bf16x3 (encoder_kernel_b3.hip, fp32-equivalent, default),
// 1 = exact fp32 MFMA (encoder_kernel_f32.hip, Winograd F(2x2,3x3) in L0 and L2), 2 = split-f16
// (encoder_kernel_h2.hip, |x| < 65504).  Chosen per call: no process-wide state.
constexpr int kPrecFp32 = 0, kPrecFp32Mfma = 1, kPrecSplitF16 = 2;
int encoder_launch_f32(const float* obs, const float* packed, float* feat, int M, hipStream_t st);
int encoder_launch_h2(const float* obs, const float* packed, float* feat, int M, int* range_flag,
                      hipStream_t st);
int encoder_launch_b3(const float* obs, const float* packed, float* feat, int M, hipStream_t st);

// range_flag (optional device int): raised by the split-f16 schedule when an activation leaves the
// f16 range; the other schedules have no such limit and never touch it.
int encoder_launch(const float* obs, const float* packed, float* feat, int M, int* range_flag, int prec,
                   hipStream_t st) {
    switch (prec) {
        case kPrecFp32: return encoder_launch_b3(obs, packed, feat, M, st);
        case kPrecFp32Mfma: return encoder_launch_f32(obs, packed, feat, M, st);
        case kPrecSplitF16: return encoder_launch_h2(obs, packed, feat, M, range_flag, st);
        default: return -1;
    }
}

}  // namespace gnnpp
