// Encoder kernel, exact-fp32 schedule (tuning value 5): fp32 MFMA 16x16x4 with the weight-fragment
// register ring of encoder_ring_f32.hip and Winograd F(2x2,3x3) for the two
// layers whose 2x2 output tiles coincide with a MaxPool window, L0 (3 -> 32 @ 11x11) and
// L2 (32 -> 64 @ 5x5).  The encoder is bound by the fp32 matrix pipe for the MFMAs it issues, so the
// remaining lever is to issue fewer:
//
//      Y = A^T [ (G g G^T) (.) (B^T d B) ] A            16 multiplies per 2x2 outputs instead of 36
//
//   L2: 4 tiles x 16 Winograd positions x 8 k-steps = 512 MFMAs per wave (direct + tap skipping: 968)
//   L0: 16 MFMAs per channel tile per pool window (direct: 28)             => -21 % MFMAs per tile
//
// Why it is cheap HERE: in the fragment layout a lane holds 4 input channels of ONE agent, and the
// input transform B^T d B only mixes spatial positions -- it is lane-local arithmetic on the sixteen
// v4f the lane reads from LDS (no staging buffer, no shuffles), and its result already IS the MFMA
// B fragment.  Likewise the 16 products of a tile sit in 16 accumulators of the same lane, so the
// output transform, BatchNorm, ReLU and the 2x2 max are lane-local too.  G g G^T is precomputed by
// gnnpp_encoder_pack.  Still exact fp32 arithmetic (a different summation order: |dlogit| ~1e-7).
#include "gnnpp_common.h"

namespace gnnpp {

// per-wave weight stream of v3: L1 (18) | L2 Winograd (4 tiles x 2 groups x 16 positions = 128) |
// L3 (36) | L4 tile a (36) | L4 tile b (36) | FC a (8) | FC b (8)
constexpr int k3_L1 = 0, k3_L2 = 18, k3_L3 = 146, k3_L4A = 182, k3_L4B = 218, k3_FCA = 254,
              k3_FCB = 262, k3_END = 270;

struct WStream3 {
    const float* l1;
    const float* u2;
    const float* l3;
    const float* l4a;
    const float* l4b;
    const float* fca;
    const float* fcb;
};

struct ItemsV3 {
    typedef WStream3 Stream;
    static constexpr int kEnd = k3_END;
    static __device__ __forceinline__ const float* ptr(const WStream3& ws, int idx) {
        if (idx < k3_L2) { const int j = idx - k3_L1; return ws.l1 + ((j % 9) * 2 + j / 9) * 256; }
        if (idx < k3_L3) return ws.u2 + ((idx - k3_L2) % 32) * 256;     // [g 2][wpos 16], per tile
        if (idx < k3_L4A) { const int j = idx - k3_L3; return ws.l3 + ((j % 9) * 4 + j / 9) * 256; }
        if (idx < k3_L4B) { const int j = idx - k3_L4A; return ws.l4a + ((j % 9) * 4 + j / 9) * 256; }
        if (idx < k3_FCA) { const int j = idx - k3_L4B; return ws.l4b + ((j % 9) * 4 + j / 9) * 256; }
        if (idx < k3_FCB) return ws.fca + (idx - k3_FCA) * 256;
        return ws.fcb + (idx - k3_FCB) * 256;
    }
};

// B^T d B on a 4x4 patch, element type T (float or v4f); in place
template <class T>
__device__ __forceinline__ void winograd_input(T (&d)[16]) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {                       // rows: B^T d
        const T d0 = d[v], d1 = d[4 + v], d2 = d[8 + v], d3 = d[12 + v];
        d[v] = d0 - d2; d[4 + v] = d1 + d2; d[8 + v] = d2 - d1; d[12 + v] = d1 - d3;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {                       // columns: (.) B
        const T t0 = d[4 * a], t1 = d[4 * a + 1], t2 = d[4 * a + 2], t3 = d[4 * a + 3];
        d[4 * a] = t0 - t2; d[4 * a + 1] = t1 + t2; d[4 * a + 2] = t2 - t1; d[4 * a + 3] = t1 - t3;
    }
}

// A^T M A: 16 products -> the 2x2 outputs y[0..3] = (0,0), (0,1), (1,0), (1,1)
__device__ __forceinline__ void winograd_output(const v4f (&m)[16], v4f (&y)[4]) {
    v4f r0[4], r1[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        r0[b] = m[b] + m[4 + b] + m[8 + b];
        r1[b] = m[4 + b] - m[8 + b] - m[12 + b];
    }
    y[0] = r0[0] + r0[1] + r0[2];
    y[1] = r0[1] - r0[2] - r0[3];
    y[2] = r1[0] + r1[1] + r1[2];
    y[3] = r1[1] - r1[2] - r1[3];
}

__global__ __launch_bounds__(kThreads, 2) void encoder_kernel_f32(const float* __restrict__ obs,
                                                                 const float* __restrict__ pk,
                                                                 float* __restrict__ feat, int M) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    float* const X = reinterpret_cast<float*>(gnnpp_smem);          // activations, in place
    float* const bufObs = X + kBufFloats;                            // padded observations
    v4f* const X4 = reinterpret_cast<v4f*>(X);
    // late layers run in place: a wave holds its outputs in registers across a barrier (measured
    // faster than ping-ponging through the dead observation buffer, profiles/r01_ab_encoder_variants.jsonl)
    constexpr bool LATE_Y = false;
    v4f* const Y4 = X4;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int a = lane & 15;
    const int q = lane >> 4;
    const int agent0 = blockIdx.x * kTileAgents;

    WStream3 ws;
    ws.l1 = pk + EncLayout::kW1 + (wave & 1) * (9 * 2 * 256) + lane * 4;
    ws.u2 = pk + EncLayout::kU2 + wave * (2 * 16 * 256) + lane * 4;
    ws.l3 = pk + EncLayout::kW3 + wave * (9 * 4 * 256) + lane * 4;
    ws.l4a = pk + EncLayout::kW4 + wave * (9 * 4 * 256) + lane * 4;
    ws.l4b = pk + EncLayout::kW4 + (wave + kWaves) * (9 * 4 * 256) + lane * 4;
    ws.fca = pk + EncLayout::kWfc + wave * (8 * 256) + lane * 4;
    ws.fcb = pk + EncLayout::kWfc + (wave + kWaves) * (8 * 256) + lane * 4;
    v4f ring[kRing];
#pragma unroll
    for (int i = 0; i < kRing; ++i) ring_load<ItemsV3>(ws, ring, i);

    // ---- observations: all loads first, zero-fill while they fly, then scatter (as v2) ----------
    {
        constexpr int NV4 = kTileAgents * kObsFloats / 4;
        constexpr int PER = (NV4 + kThreads - 1) / kThreads;
        const int n_agents = min(kTileAgents, M - agent0);
        const int valid = n_agents * kObsFloats;
        const float* src = obs + (size_t)agent0 * kObsFloats;
        v4f v[PER];
        if (n_agents == kTileAgents) {
#pragma unroll
            for (int k = 0; k < PER; ++k)
                v[k] = *reinterpret_cast<const v4f*>(src + 4 * min(tid + k * kThreads, NV4 - 1));
        } else {
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int e0 = (tid + k * kThreads) * 4;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[k][c] = src[min(e0 + c, valid - 1)];
            }
        }
        v4f* z = reinterpret_cast<v4f*>(bufObs);
        for (int i = tid; i < kObsFloatsLds / 4; i += kThreads) z[i] = vzero();
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e0 = (tid + k * kThreads) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int e = e0 + c;
                if (e < valid) {
                    const int ag = e / kObsFloats, rem = e - ag * kObsFloats;
                    const int ch = rem / 121, r2 = rem - ch * 121;
                    const int y = r2 / 11, x = r2 - y * 11;
                    bufObs[ag * kAgentStride + ch * (kPadHW * kPadHW) + (y + 1) * kPadHW + x + 1] =
                        v[k][c];
                }
            }
        }
    }
    __syncthreads();

    // ---- L0 (Winograd): one 4x4 input patch per pool window, 16 MFMAs per channel tile ------------
    {
        float U0[2][16];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int w = 0; w < 16; ++w) U0[i][w] = pk[EncLayout::kU0 + (i * 16 + w) * 64 + lane];
        v4f sc[2], sh[2];
        load_ss(pk + EncLayout::kSS0, 32, 0, q, sc[0], sh[0]);
        load_ss(pk + EncLayout::kSS0, 32, 1, q, sc[1], sh[1]);
        // k-slot q = input channel (slot 3 has zero weights: any finite operand will do)
        const float* chan = bufObs + a * kAgentStride + (q < 3 ? q : 2) * (kPadHW * kPadHW);
        float dc[16], dn[16];
        auto load_patch = [&](float (&d)[16], int win) {
            const int wy = win / 5, wx = win - wy * 5;
            const float* base = chan + (2 * wy) * kPadHW + 2 * wx;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) d[4 * u + v] = base[u * kPadHW + v];
        };
        load_patch(dc, wave);
        for (int win = wave; win < 25; win += kWaves) {
            if (win + kWaves < 25) load_patch(dn, win + kWaves);
            winograd_input(dc);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                v4f m[16];
#pragma unroll
                for (int w = 0; w < 16; ++w) m[w] = mfma16(U0[i][w], dc[w], vzero());
                v4f y[4];
                winograd_output(m, y);
                v4f r = vrelu(vfma(y[0], sc[i], sh[i]));
#pragma unroll
                for (int pp = 1; pp < 4; ++pp) r = vmax(r, vfma(y[pp], sc[i], sh[i]));
                X4[(win * 2 + i) * 64 + lane] = r;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) dc[i] = dn[i];
        }
    }
    __syncthreads();

    // ---- L1: 32 -> 32 @ 5x5, in place (direct, as v2) ----------------------------------------------
    {
        const int mt = wave & 1, half = wave >> 1;
        v4f sc, sh;
        load_ss(pk + EncLayout::kSS1, 32, mt, q, sc, sh);
        v4f acc[13];
#pragma unroll
        for (int j = 0; j < 13; ++j) acc[j] = vzero();
        if (half == 0)
            conv_tile_ring<ItemsV3, k3_L1, 32, 5, 5, 13, PosL1<0>>(ws, ring, X4, acc, lane);
        else
            conv_tile_ring<ItemsV3, k3_L1, 32, 5, 5, 13, PosL1<1>>(ws, ring, X4, acc, lane);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 13; ++j)
            if (half * 13 + j < 25)
                X4[((half * 13 + j) * 2 + mt) * 64 + lane] = vrelu(vfma(acc[j], sc, sh));
    }
    __syncthreads();

    // ---- L2 (Winograd): 32 -> 64 @ 5x5, the four 2x2 output tiles == the four pool windows ---------
    {
        const int mt = wave;
        v4f sc, sh;
        load_ss(pk + EncLayout::kSS2, 64, mt, q, sc, sh);
        v4f res[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ty = t >> 1, tx = t & 1;
            v4f acc[16];
#pragma unroll
            for (int w = 0; w < 16; ++w) acc[w] = vzero();
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                v4f d[16];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int r = 2 * ty - 1 + u, c = 2 * tx - 1 + v;      // compile-time
                        d[4 * u + v] = (r >= 0 && r < 5 && c >= 0 && c < 5)
                                           ? X4[((r * 5 + c) * 2 + g) * 64 + lane]
                                           : vzero();
                    }
                winograd_input(d);
#pragma unroll
                for (int w2 = 0; w2 < 8; ++w2) {          // two Winograd positions per step: two
                    const int idx = k3_L2 + t * 32 + g * 16 + 2 * w2;   // independent MFMA chains
                    __builtin_amdgcn_sched_barrier(kSchedItemMask);
                    const v4f A0 = ring[idx % kRing];
                    ring_load<ItemsV3>(ws, ring, idx + kRing);
                    const v4f A1 = ring[(idx + 1) % kRing];
                    ring_load<ItemsV3>(ws, ring, idx + 1 + kRing);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc[2 * w2] = mfma16(A0[s], d[2 * w2][s], acc[2 * w2]);
                        acc[2 * w2 + 1] = mfma16(A1[s], d[2 * w2 + 1][s], acc[2 * w2 + 1]);
                    }
                }
            }
            v4f y[4];
            winograd_output(acc, y);
            v4f r = vrelu(vfma(y[0], sc, sh));
#pragma unroll
            for (int pp = 1; pp < 4; ++pp) r = vmax(r, vfma(y[pp], sc, sh));
            if (LATE_Y) Y4[(t * 4 + mt) * 64 + lane] = r;
            else res[t] = r;
        }
        if (!LATE_Y) {
            __syncthreads();                           // everyone is done reading L1's output
#pragma unroll
            for (int t = 0; t < 4; ++t) X4[(t * 4 + mt) * 64 + lane] = res[t];
        }
    }
    __syncthreads();

    // ---- L3: 64 -> 64 @ 2x2 : Y -> X -----------------------------------------------------------------
    {
        const int mt = wave;
        v4f sc, sh;
        load_ss(pk + EncLayout::kSS3, 64, mt, q, sc, sh);
        v4f acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = vzero();
        conv_tile_ring<ItemsV3, k3_L3, 64, 2, 2, 4, Pos2x2>(ws, ring, Y4, acc, lane);
        if (!LATE_Y) __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) X4[(j * 4 + mt) * 64 + lane] = vrelu(vfma(acc[j], sc, sh));
    }
    __syncthreads();

    // ---- L4: 64 -> 128 @ 2x2, pool -> [1][8][64], two channel tiles per wave : X -> Y -------------
    {
        v4f sc0, sh0, sc1, sh1;
        load_ss(pk + EncLayout::kSS4, 128, wave, q, sc0, sh0);
        load_ss(pk + EncLayout::kSS4, 128, wave + kWaves, q, sc1, sh1);
        v4f acc0[4], acc1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc0[j] = vzero(); acc1[j] = vzero(); }
        conv_tile_ring<ItemsV3, k3_L4A, 64, 2, 2, 4, Pos2x2>(ws, ring, X4, acc0, lane);
        conv_tile_ring<ItemsV3, k3_L4B, 64, 2, 2, 4, Pos2x2>(ws, ring, X4, acc1, lane);
        v4f m0 = vrelu(vfma(acc0[0], sc0, sh0)), m1 = vrelu(vfma(acc1[0], sc1, sh1));
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            m0 = vmax(m0, vfma(acc0[j], sc0, sh0));
            m1 = vmax(m1, vfma(acc1[j], sc1, sh1));
        }
        if (!LATE_Y) __syncthreads();
        Y4[wave * 64 + lane] = m0;
        Y4[(wave + kWaves) * 64 + lane] = m1;
    }
    __syncthreads();

    // ---- FC 128 -> 128 + ReLU -> feat[agent][128] ----------------------------------------------
    {
        v4f Bf[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) Bf[g] = Y4[g * 64 + lane];
        v4f acc[2][2] = {{vzero(), vzero()}, {vzero(), vzero()}};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int idx = (t == 0 ? k3_FCA : k3_FCB) + g;
                __builtin_amdgcn_sched_barrier(kSchedItemMask);
                const v4f A = ring[idx % kRing];
                ring_load<ItemsV3>(ws, ring, idx + kRing);
                acc[t][g & 1] = mfma16x4(A, Bf[g], acc[t][g & 1]);
            }
        }
        if (agent0 + a < M) {
            const int mt0 = wave, mt1 = wave + kWaves;
            float* dst = feat + (size_t)(agent0 + a) * 128 + q * 4;
            const v4f b0 = *reinterpret_cast<const v4f*>(pk + EncLayout::kBfc + mt0 * 16 + q * 4);
            const v4f b1 = *reinterpret_cast<const v4f*>(pk + EncLayout::kBfc + mt1 * 16 + q * 4);
            *reinterpret_cast<v4f*>(dst + mt0 * 16) = vrelu(acc[0][0] + acc[0][1] + b0);
            *reinterpret_cast<v4f*>(dst + mt1 * 16) = vrelu(acc[1][0] + acc[1][1] + b1);
        }
    }
}

int encoder_launch_f32(const float* obs, const float* packed, float* feat, int M, hipStream_t st) {
    static LdsAttrOnce once;
    constexpr size_t smem = (kBufFloats + kObsFloatsLds) * sizeof(float);
    set_lds_attr_once(once, reinterpret_cast<const void*>(&encoder_kernel_f32), (int)smem);
    const int grid = (M + kTileAgents - 1) / kTileAgents;
    hipLaunchKernelGGL((encoder_kernel_f32), dim3(grid), dim3(kThreads), smem, st, obs, packed,
                       feat, M);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace gnnpp
