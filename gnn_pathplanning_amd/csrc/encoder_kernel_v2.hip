// Encoder kernel, schedule v2 (default).  Same math, layouts, LDS plan and packed weights as
// encoder_kernel<true> (encoder_kernel.hip: in-place layers, 78.9 KB LDS, two workgroups per CU);
// what changes is how operands reach the MFMA pipe.  Motivated by the r01 ISA trace + PMC data
// (DESIGN.md section 4.1):
//
//  * observations: the 16-agent tile is one contiguous, 16-byte aligned 23 KB run of HBM.  v1 walked
//    it with a load -> wait -> ds_write loop (27 exposed memory round trips per tile).  v2 issues
//    all of a thread's float4 loads up front, zero-fills the padded LDS image while they fly, then
//    scatters.
//  * weights: every wave consumes a fixed sequence of 160 packed A fragments (1 KiB each, L2
//    resident) from L1 to the FC.  In the 2x2 layers a fragment feeds only 4..16 MFMAs, far less
//    than the L2 latency, and the compiler kept 1-2 loads in flight.  v2 streams them through a
//    12-deep register ring (48 VGPRs) that runs ahead of the MFMAs ACROSS layer boundaries: the
//    next layer's first fragments are already in registers when its barrier opens.
//  * traversal is group-major (g, then tap) so consecutive fragments of the 2x2 layers hit
//    different accumulators (no dependent MFMA chains), and the FC uses two accumulators per tile.
//  * L0 operands (28 ds_read_b32 per 2x2 pool window) are double-buffered one window ahead.
#include "gnnpp_common.h"

namespace gnnpp {

constexpr int kRing = 12;
// sched_barrier mask: ALU | VALU | SALU | DS | DS read | DS write | transcendental may cross
constexpr int kSchedItemMask = 0x1 | 0x2 | 0x4 | 0x80 | 0x100 | 0x200 | 0x400;
// item index space of the per-wave weight stream
constexpr int kI_L1 = 0, kI_L2 = 18, kI_L3 = 36, kI_L4A = 72, kI_L4B = 108, kI_FCA = 144,
              kI_FCB = 152, kI_END = 160;

struct WStream {                 // per-wave segment bases, already offset by lane * 4 floats
    const float* l1;
    const float* l2;
    const float* l3;
    const float* l4a;
    const float* l4b;
    const float* fca;
    const float* fcb;
};

// Address of stream item `idx` (compile-time after unrolling).  Conv segments are traversed
// group-major: local item j -> (g = j / 9, tap = j % 9), stored at [(tap * NG + g)].
struct ItemsV2 {
    typedef WStream Stream;
    static constexpr int kEnd = kI_END;
    static __device__ __forceinline__ const float* ptr(const WStream& ws, int idx);
};
__device__ __forceinline__ const float* ItemsV2::ptr(const WStream& ws, int idx) {
    if (idx < kI_L2) { const int j = idx - kI_L1; return ws.l1 + ((j % 9) * 2 + j / 9) * 256; }
    if (idx < kI_L3) { const int j = idx - kI_L2; return ws.l2 + ((j % 9) * 2 + j / 9) * 256; }
    if (idx < kI_L4A) { const int j = idx - kI_L3; return ws.l3 + ((j % 9) * 4 + j / 9) * 256; }
    if (idx < kI_L4B) { const int j = idx - kI_L4A; return ws.l4a + ((j % 9) * 4 + j / 9) * 256; }
    if (idx < kI_FCA) { const int j = idx - kI_L4B; return ws.l4b + ((j % 9) * 4 + j / 9) * 256; }
    if (idx < kI_FCB) return ws.fca + (idx - kI_FCA) * 256;
    return ws.fcb + (idx - kI_FCB) * 256;
}

// Ring loads are relaxed wavefront-scope ATOMIC loads (two 8-byte halves): same instruction and
// cache policy as a plain global_load, but "ordered" for the compiler, so they are issued where
// the source puts them instead of being sunk next to their first use 12 items later.
template <class Items>
__device__ __forceinline__ void ring_load(const typename Items::Stream& ws, v4f (&ring)[kRing],
                                          int idx) {
    if (idx < Items::kEnd) {
        typedef unsigned long long u64;
        u64* p = reinterpret_cast<u64*>(const_cast<float*>(Items::ptr(ws, idx)));
        const u64 lo = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const u64 hi = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        v4f r;
        r[0] = __int_as_float((int)(lo & 0xffffffffu));
        r[1] = __int_as_float((int)(lo >> 32));
        r[2] = __int_as_float((int)(hi & 0xffffffffu));
        r[3] = __int_as_float((int)(hi >> 32));
        ring[idx % kRing] = r;
    }
}

// One output-channel tile over a compile-time position set, weights from the ring.
template <class Items, int START, int CIN, int H, int W, int NSLOT, class PosFn>
__device__ __forceinline__ void conv_tile_ring(const typename Items::Stream& ws,
                                               v4f (&ring)[kRing], const v4f* in,
                                               v4f (&acc)[NSLOT], int lane) {
    constexpr int NG = CIN / 16;
#pragma unroll
    for (int it = 0; it < 9 * NG; ++it) {
        const int g = it / 9, tap = it % 9;
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        // Pin the stream: MFMAs and global loads may not cross an item boundary (ALU and LDS
        // operations may), otherwise the scheduler sinks each refill next to its consumer.
        __builtin_amdgcn_sched_barrier(kSchedItemMask);
        const v4f A = ring[(START + it) % kRing];
        ring_load<Items>(ws, ring, START + it + kRing);   // refill the slot just consumed
        v4f Bf[NSLOT];
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            int y = 0, x = 0;
            const bool used = PosFn::get(j, y, x);
            const int iy = y + dy, ix = x + dx;
            if (used && iy >= 0 && iy < H && ix >= 0 && ix < W)
                Bf[j] = in[((iy * W + ix) * NG + g) * 64 + lane];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int j = 0; j < NSLOT; ++j) {
                int y = 0, x = 0;
                const bool used = PosFn::get(j, y, x);
                const int iy = y + dy, ix = x + dx;
                if (used && iy >= 0 && iy < H && ix >= 0 && ix < W)
                    acc[j] = mfma16(A[s], Bf[j][s], acc[j]);
            }
        }
    }
}

__global__ __launch_bounds__(kThreads, 2) void encoder_kernel_v2(const float* __restrict__ obs,
                                                                 const float* __restrict__ pk,
                                                                 float* __restrict__ feat, int M) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    float* const X = reinterpret_cast<float*>(gnnpp_smem);          // activations, in place
    float* const bufObs = X + kBufFloats;                            // padded observations
    v4f* const X4 = reinterpret_cast<v4f*>(X);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int a = lane & 15;
    const int q = lane >> 4;
    const int agent0 = blockIdx.x * kTileAgents;

    // ---- weight stream of this wave; the ring starts filling right away ------------------------
    WStream ws;
    ws.l1 = pk + EncLayout::kW1 + (wave & 1) * (9 * 2 * 256) + lane * 4;
    ws.l2 = pk + EncLayout::kW2 + wave * (9 * 2 * 256) + lane * 4;
    ws.l3 = pk + EncLayout::kW3 + wave * (9 * 4 * 256) + lane * 4;
    ws.l4a = pk + EncLayout::kW4 + wave * (9 * 4 * 256) + lane * 4;
    ws.l4b = pk + EncLayout::kW4 + (wave + kWaves) * (9 * 4 * 256) + lane * 4;
    ws.fca = pk + EncLayout::kWfc + wave * (8 * 256) + lane * 4;
    ws.fcb = pk + EncLayout::kWfc + (wave + kWaves) * (8 * 256) + lane * 4;
    v4f ring[kRing];
#pragma unroll
    for (int i = 0; i < kRing; ++i) ring_load<ItemsV2>(ws, ring, i);

    // ---- observations: all loads first, zero-fill while they fly, then scatter ------------------
    {
        constexpr int NV4 = kTileAgents * kObsFloats / 4;            // 1452 float4 per full tile
        constexpr int PER = (NV4 + kThreads - 1) / kThreads;         // 6
        const int n_agents = min(kTileAgents, M - agent0);
        const int valid = n_agents * kObsFloats;                     // floats present in HBM
        const float* src = obs + (size_t)agent0 * kObsFloats;
        v4f v[PER];
        if (n_agents == kTileAgents) {
            // full tile (block-uniform branch): unconditional 16-byte loads, clamped index
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

    // ---- L0: 3 -> 32 @ 11x11 (10x10 used), BN, ReLU, pool -> [25][2][64] v4f in X --------------
    {
        float A0[2][7];
        int offB[7];
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            A0[0][s] = pk[EncLayout::kW0 + (0 * 7 + s) * 64 + lane];
            A0[1][s] = pk[EncLayout::kW0 + (1 * 7 + s) * 64 + lane];
            int k = 4 * s + q;
            if (k >= 27) k = 0;                        // weight is zero there; any finite operand
            const int c = k / 9, ky = (k % 9) / 3, kx = k % 3;
            offB[s] = a * kAgentStride + c * (kPadHW * kPadHW) + ky * kPadHW + kx;
        }
        v4f sc[2], sh[2];
        load_ss(pk + EncLayout::kSS0, 32, 0, q, sc[0], sh[0]);
        load_ss(pk + EncLayout::kSS0, 32, 1, q, sc[1], sh[1]);
        float Bc[28], Bn[28];
        auto load_window = [&](float (&B)[28], int win) {
            const int wy = win / 5, wx = win - wy * 5;
            const float* base = bufObs + (2 * wy) * kPadHW + 2 * wx;
#pragma unroll
            for (int s = 0; s < 7; ++s)
#pragma unroll
                for (int pp = 0; pp < 4; ++pp)
                    B[s * 4 + pp] = base[offB[s] + (pp >> 1) * kPadHW + (pp & 1)];
        };
        load_window(Bc, wave);
        for (int win = wave; win < 25; win += kWaves) {
            if (win + kWaves < 25) load_window(Bn, win + kWaves);   // next window, one trip ahead
            v4f acc[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) acc[i][pp] = vzero();
#pragma unroll
            for (int s = 0; s < 7; ++s) {
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) {
                    acc[0][pp] = mfma16(A0[0][s], Bc[s * 4 + pp], acc[0][pp]);
                    acc[1][pp] = mfma16(A0[1][s], Bc[s * 4 + pp], acc[1][pp]);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                v4f m = vrelu(vfma(acc[i][0], sc[i], sh[i]));
#pragma unroll
                for (int pp = 1; pp < 4; ++pp) m = vmax(m, vfma(acc[i][pp], sc[i], sh[i]));
                X4[(win * 2 + i) * 64 + lane] = m;
            }
#pragma unroll
            for (int i = 0; i < 28; ++i) Bc[i] = Bn[i];
        }
    }
    __syncthreads();

    // ---- L1: 32 -> 32 @ 5x5, in place ------------------------------------------------------------
    {
        const int mt = wave & 1, half = wave >> 1;
        v4f sc, sh;
        load_ss(pk + EncLayout::kSS1, 32, mt, q, sc, sh);
        v4f acc[13];
#pragma unroll
        for (int j = 0; j < 13; ++j) acc[j] = vzero();
        if (half == 0)
            conv_tile_ring<ItemsV2, kI_L1, 32, 5, 5, 13, PosL1<0>>(ws, ring, X4, acc, lane);
        else
            conv_tile_ring<ItemsV2, kI_L1, 32, 5, 5, 13, PosL1<1>>(ws, ring, X4, acc, lane);
        __syncthreads();                               // everyone is done reading L0's output
#pragma unroll
        for (int j = 0; j < 13; ++j)
            if (half * 13 + j < 25)
                X4[((half * 13 + j) * 2 + mt) * 64 + lane] = vrelu(vfma(acc[j], sc, sh));
    }
    __syncthreads();

    // ---- L2: 32 -> 64 @ 5x5 (4x4 used), pool -> [4][4][64], in place ----------------------------
    {
        const int mt = wave;
        v4f sc, sh;
        load_ss(pk + EncLayout::kSS2, 64, mt, q, sc, sh);
        v4f acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = vzero();
        conv_tile_ring<ItemsV2, kI_L2, 32, 5, 5, 16, PosL2>(ws, ring, X4, acc, lane);
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            v4f m = vrelu(vfma(acc[4 * w], sc, sh));
#pragma unroll
            for (int i = 1; i < 4; ++i) m = vmax(m, vfma(acc[4 * w + i], sc, sh));
            X4[(w * 4 + mt) * 64 + lane] = m;
        }
    }
    __syncthreads();

    // ---- L3: 64 -> 64 @ 2x2, in place --------------------------------------------------------------
    {
        const int mt = wave;
        v4f sc, sh;
        load_ss(pk + EncLayout::kSS3, 64, mt, q, sc, sh);
        v4f acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = vzero();
        conv_tile_ring<ItemsV2, kI_L3, 64, 2, 2, 4, Pos2x2>(ws, ring, X4, acc, lane);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) X4[(j * 4 + mt) * 64 + lane] = vrelu(vfma(acc[j], sc, sh));
    }
    __syncthreads();

    // ---- L4: 64 -> 128 @ 2x2, pool -> [1][8][64], two channel tiles per wave, in place ---------
    {
        v4f sc0, sh0, sc1, sh1;
        load_ss(pk + EncLayout::kSS4, 128, wave, q, sc0, sh0);
        load_ss(pk + EncLayout::kSS4, 128, wave + kWaves, q, sc1, sh1);
        v4f acc0[4], acc1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc0[j] = vzero(); acc1[j] = vzero(); }
        conv_tile_ring<ItemsV2, kI_L4A, 64, 2, 2, 4, Pos2x2>(ws, ring, X4, acc0, lane);
        conv_tile_ring<ItemsV2, kI_L4B, 64, 2, 2, 4, Pos2x2>(ws, ring, X4, acc1, lane);
        v4f m0 = vrelu(vfma(acc0[0], sc0, sh0)), m1 = vrelu(vfma(acc1[0], sc1, sh1));
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            m0 = vmax(m0, vfma(acc0[j], sc0, sh0));
            m1 = vmax(m1, vfma(acc1[j], sc1, sh1));
        }
        __syncthreads();
        X4[wave * 64 + lane] = m0;
        X4[(wave + kWaves) * 64 + lane] = m1;
    }
    __syncthreads();

    // ---- FC 128 -> 128 + ReLU -> feat[agent][128] ----------------------------------------------
    {
        v4f Bf[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) Bf[g] = X4[g * 64 + lane];
        v4f acc[2][2] = {{vzero(), vzero()}, {vzero(), vzero()}};   // [tile][g parity]
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int idx = (t == 0 ? kI_FCA : kI_FCB) + g;
                __builtin_amdgcn_sched_barrier(kSchedItemMask);
                const v4f A = ring[idx % kRing];
                ring_load<ItemsV2>(ws, ring, idx + kRing);
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

int encoder_launch_v2(const float* obs, const float* packed, float* feat, int M, hipStream_t st) {
    static bool attr_set = false;
    constexpr size_t smem = (kBufFloats + kObsFloatsLds) * sizeof(float);
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&encoder_kernel_v2),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    const int grid = (M + kTileAgents - 1) / kTileAgents;
    hipLaunchKernelGGL(encoder_kernel_v2, dim3(grid), dim3(kThreads), smem, st, obs, packed, feat, M);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace gnnpp
