// K-tap graph filter for MANY SMALL graphs (N <= 16 nodes, G = F = 128, node-major rows): the throughput form of
// BatchLSIGF (utils/graphUtils/graphML.py:2273-2367) -- thousands of 10-agent graphs per launch, the regime of the
// north star's "achieved-HBM-fraction on synthetic random GSO+feature batches" -- in the default, fp32-equivalent
// arithmetic (bf16x3 planes, gnnpp_common.h).
//
// lsigf_kernel serves every layout the filter API has from ONE >= 120 KB-LDS workgroup per CU (two fp32 z buffers
// for up to 112 rows): staging, list building, shifts and contractions serialise behind barriers with nothing
// else resident (profiles/r02_c3_filter_pmc.txt: 55 % of the wave time parked, the matrix pipe 18 % busy), and it
// has no room for bf16x3 planes, so its fp32-equivalent contraction runs on the fp32 MFMA (2.7x the pipe time).
// Here a workgroup of four waves owns at most 48 rows = floor(48 / N) whole graphs and keeps
//      z   [48][136] fp32    the CURRENT tap signal, shifted IN PLACE: a wave owns whole graphs, and a graph's shift
//                            z_{k+1} = S^T z_k is ONE dense 16 x 16 product per 16-feature tile on the exact fp32
//                            MFMA (the GSO block zero-padded to 16 x 16; all m ascending: bit-identical to the sparse
//                            gather, fmaf(0, z, acc) == acc) -- 32 MFMAs per graph, all eight result tiles in
//                            registers before any row is written back
//      PB  [48][800 B]       the same rows as three bf16 planes, written by whoever produces z_k (the staging loop,
//                            the shift's write-back): no conversion pass
//      S   [graphs][16][17]  the GSO blocks, zero-padded (the shift's B operand)
// <= 80 KB: TWO workgroups per CU, so one's staging / shift / barrier phases run beside the other's MFMAs.  A wave
// contracts two 16-feature output tiles over all rows (six MFMAs per 32 channels and row tile, hh products and the
// five smaller terms in separate accumulators); the tap's 24 A fragments per wave are requested one tap ahead.
// Epilogue: bias (+ ReLU) -> fp32 rows in z -> fully coalesced 512-byte row stores, or the 128 -> 5 action head.
// HBM traffic is the algorithmic minimum (x and S in, y out, once); the taps stream from L2 (288 KB per workgroup
// at K = 3, 7.2 KB per agent-step at N = 10 -- the reason a workgroup takes 48 rows and not 16).
#include "gnnpp_common.h"

namespace gnnpp {

constexpr int kSmMaxNodes = 16;
constexpr int kSmZs = 136;                                 // fp32 row stride (floats)
constexpr int kSmPRow = 3 * 256 + 32;                      // plane row stride (bytes)
constexpr int kSmMaxGraphs = 12;                           // graphs per workgroup (tiny graphs: 12, not ROWS / N)
constexpr int kSmSBlock = 16 * 17;
// ROWS = rows (graphs x nodes) of one workgroup: 48 (three row tiles, two workgroups per CU: the fewest tap bytes
// per agent-step) or 32 (two row tiles, 49 KB: THREE workgroups per CU -- more phases of different workgroups side
// by side; at N = 10 its 30 rows also fill their tiles better than 40 of 48)
template <int ROWS> struct SmLayout {
    static constexpr int kZBytes = ROWS * kSmZs * 4;
    static constexpr int kPBytes = ROWS * kSmPRow;
    static constexpr int kCOff = kZBytes + kPBytes;        // constants: bias [128] | act_w [5][128] | act_b [5] (+ pad)
    static constexpr int kSOff = kCOff + 776 * 4;          // GSO blocks: graphs x 16 x 17 floats
    static constexpr int smem(int gpw) { return kSOff + gpw * kSmSBlock * 4; }
};
static_assert(2 * SmLayout<48>::smem(kSmMaxGraphs) <= kLdsBytes && 3 * SmLayout<32>::smem(3) <= kLdsBytes, "LDS budget");

// NS4 = ceil(N / 4): the k-steps of the dense shift that can meet a non-zero weight (rows m >= N of a graph's 16 x 16 GSO
// block are zero: skipping their MFMAs is bit-identical; as a compile-time bound -- a run-time guard broke the unrolled
// schedule and cost 20 %)
template <int ROWS, int NS4>
__global__ __launch_bounds__(256, 2) void lsigf_small_b3_kernel(const LsigfArgs p) {
    typedef SmLayout<ROWS> LY;
    constexpr int RT = ROWS / 16, XV = ROWS * 32 / 256;
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    float* const z = reinterpret_cast<float*>(gnnpp_smem);
    char* const PB = gnnpp_smem + LY::kZBytes;
    float* const Ssm = reinterpret_cast<float*>(gnnpp_smem + LY::kSOff);
    float* const cb = reinterpret_cast<float*>(gnnpp_smem + LY::kCOff);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int a = lane & 15, q = lane >> 4;
    const int N = p.N, K = p.K;
    const int g0 = blockIdx.x * p.gpw;
    const int ng = min(p.gpw, p.B - g0);
    const int rows = ng * N;                               // <= ROWS
    const int ntile = (rows + 15) >> 4;                    // row tiles in use (wave-uniform)

    GNNPP_STAMP(blockIdx.x, 0, tid == 0);
    // ---- every global load of the kernel's front, issued now ------------------------------------------------
    const v4f* xs = reinterpret_cast<const v4f*>(p.x + (size_t)g0 * N * 128);
    v4f xv[XV];
#pragma unroll
    for (int u = 0; u < XV; ++u) {
        const int i = tid + u * 256;
        if (i < rows * 32) xv[u] = xs[i];
    }
    // the GSO blocks, gathered element by element of the PADDED [graph][16][17] layout (each LDS word is written
    // exactly once: zeros outside the graph's N x N block); four loads in flight per thread and trip
    const int nSb = ng * kSmSBlock;
    float sv[4] = {0.f, 0.f, 0.f, 0.f};
    auto s_load = [&](int idx) -> float {
        const int j = idx / kSmSBlock, e = idx - j * kSmSBlock;
        const int m = e / 17, n = e - m * 17;
        if (m >= N || n >= N) return 0.f;
        const size_t gi = ((size_t)(g0 + j) * N + m) * N + n;
        return p.s_is_f64 ? (float)reinterpret_cast<const double*>(p.S)[gi] : reinterpret_cast<const float*>(p.S)[gi];
    };
    if (K > 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (tid + u * 256 < nSb) sv[u] = s_load(tid + u * 256);
    }
    float cpre[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = tid + u * 256;                       // bias [128] | act_w [640] | act_b [5]
        if (c < 128) cpre[u] = p.bias ? p.bias[c] : 0.f;
        else if (c < 768) cpre[u] = p.act_w ? p.act_w[c - 128] : 0.f;
        else if (c < 773) cpre[u] = p.act_w ? p.act_b[c - 768] : 0.f;
    }
    // A fragments of a tap: this wave's channel tiles 2w, 2w+1, block (k, mt, kb) = three 16-byte planes
    constexpr size_t tap_stride = (size_t)8 * 4 * 768;
    v4f A[2][12];
    auto load_tap = [&](int tap) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const float* wt = p.wpk_b + tap * tap_stride + ((size_t)(2 * wave + m) * 12 * 64 + lane) * 4;
#pragma unroll
            for (int i = 0; i < 12; ++i) A[m][i] = *reinterpret_cast<const v4f*>(wt + i * 256);
        }
    };

    // ---- stage: z_0 as fp32 rows and as planes, the GSO column-major, the constants ----------------------------
#pragma unroll
    for (int u = 0; u < XV; ++u) {
        const int i = tid + u * 256;
        const bool ok = i < rows * 32;
        const v4f v = ok ? xv[u] : vzero();
        v2f pl[3];
        b3_split4(v, pl);
        // (every row of z is written, zeros past the group's rows: the dense shift multiplies the rows behind a graph by
        // zero weights, and 0 x stale-NaN would still be NaN)
        if (K > 1 && i < ROWS * 32) *reinterpret_cast<v4f*>(z + (i >> 5) * kSmZs + 4 * (i & 31)) = v;
        if (ok) {
            const int r = i >> 5, c4 = i & 31;
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) *reinterpret_cast<v2f*>(PB + r * kSmPRow + pp * 256 + 8 * c4) = pl[pp];
        }
    }
    if (K > 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (tid + u * 256 < nSb) Ssm[tid + u * 256] = sv[u];
        for (int idx = tid + 1024; idx < nSb; idx += 256) Ssm[idx] = s_load(idx);      // (more than 3 graphs: N <= 12)
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (tid + u * 256 < 773) cb[tid + u * 256] = cpre[u];
    load_tap(0);                                           // (behind the staged data in the memory queue)
    __syncthreads();
    GNNPP_STAMP(blockIdx.x, 1, tid == 0);

    v4f acc[RT][2], acc2[RT][2];                           // [row tile][channel tile]: hh | the five smaller terms
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m) { acc[t][m] = vzero(); acc2[t][m] = vzero(); }

    for (int k = 0; k < K; ++k) {
        // ---- contraction of tap k: D[f, row] += W_k[f, g] z_k[row, g] from the planes ---------------------------
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            if (t < ntile) {
                const int pr = min(t * 16 + a, rows - 1);   // (rows >= `rows`: copies, never stored)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const char* zrow = PB + pr * kSmPRow + kb * 64 + q * 16;
                    v8b Bp[3];
#pragma unroll
                    for (int pp = 0; pp < 3; ++pp) Bp[pp] = as_b8(*reinterpret_cast<const v4f*>(zrow + pp * 256));
#pragma unroll
                    for (int term = 0; term < kB3Terms; ++term)
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const v8b Ap = as_b8(A[m][3 * kb + b3_term_a(term)]);
                            if (term == kB3Terms - 1) acc[t][m] = mfma16b(Ap, Bp[0], acc[t][m]);
                            else acc2[t][m] = mfma16b(Ap, Bp[b3_term_b(term)], acc2[t][m]);
                        }
                }
            }
        }
        GNNPP_STAMP(blockIdx.x, 2 + 3 * k, tid == 0 && k < 3);          // this wave's part of tap k contracted
        if (k + 1 == K) break;
        load_tap(k + 1);                                   // in flight during the shift
        __syncthreads();                                   // every wave is done with the planes of z_k
        GNNPP_STAMP(blockIdx.x, 3 + 3 * k, tid == 0 && k < 3);
        // ---- shift z_k -> z_{k+1} in place: wave w owns graphs w, w + 4, ..; one graph = eight 16-feature tiles of
        // D[feature][node n] = sum_m z_k[m][feature] S[m][n] on the fp32 MFMA (rows past the graph meet zero weights)
        {
            const bool last = k + 2 == K;                  // z_{K-1} is only needed as planes
            for (int j = wave; j < ng; j += 4) {
                const int r0 = j * N;
                float Sb[NS4];
#pragma unroll
                for (int s4 = 0; s4 < NS4; ++s4) Sb[s4] = Ssm[j * kSmSBlock + (4 * s4 + q) * 17 + a];
                v4f d[8];
#pragma unroll
                for (int ft = 0; ft < 8; ++ft) {
                    d[ft] = vzero();
#pragma unroll
                    for (int s4 = 0; s4 < NS4; ++s4) {
                        // k-slots past the graph's N rows meet zero weights; they re-read the graph's OWN last row,
                        // never a neighbour's (0 x Inf = NaN must stay inside the sample, as in the reference)
                        const int m = r0 + min(4 * s4 + q, N - 1);
                        d[ft] = mfma16(z[m * kSmZs + 16 * ft + a], Sb[s4], d[ft]);
                    }
                }
                __builtin_amdgcn_wave_barrier();           // every row of the graph has been read
#pragma unroll
                for (int ft = 0; ft < 8; ++ft) {           // lane (q, a): node a, features 16 ft + 4 q ..
                    v2f pl[3];
                    b3_split4(d[ft], pl);
                    if (a < N) {
                        const int r = r0 + a;
                        if (!last) *reinterpret_cast<v4f*>(z + r * kSmZs + 16 * ft + 4 * q) = d[ft];
#pragma unroll
                        for (int pp = 0; pp < 3; ++pp)
                            *reinterpret_cast<v2f*>(PB + r * kSmPRow + pp * 256 + (16 * ft + 4 * q) * 2) = pl[pp];
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();                                   // planes of z_{k+1} visible
        GNNPP_STAMP(blockIdx.x, 4 + 3 * k, tid == 0 && k < 3);
    }

    // ---- epilogue: bias (+ ReLU) -> fp32 rows in z -> coalesced store / action head ------------------------------
    __syncthreads();                                       // (K == 1: nobody reads z; K > 1: the last shift is done)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int f0 = (2 * wave + m) * 16 + 4 * q;
        const v4f bv = *reinterpret_cast<const v4f*>(cb + f0);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int row = t * 16 + a;
            if (t < ntile && row < rows) {
                v4f v = (acc[t][m] + acc2[t][m]) + bv;
                if (p.relu) v = vrelu(v);
                *reinterpret_cast<v4f*>(z + row * kSmZs + f0) = v;
            }
        }
    }
    __syncthreads();
    GNNPP_STAMP(blockIdx.x, 12, tid == 0);
    if (p.y) {
        v4f* yd = reinterpret_cast<v4f*>(p.y + (size_t)g0 * N * 128);
        for (int i = tid; i < rows * 32; i += 256) {
            const int r = i >> 5, c4 = i & 31;
            yd[i] = *reinterpret_cast<const v4f*>(z + r * kSmZs + 4 * c4);
        }
    }
    if (p.act_w) {
        // logits[n, b, c] = act_b[c] + sum_f act_w[c, f] y[row, f]: one thread per (row, action), ascending f
        for (int i = tid; i < rows * 5; i += 256) {
            const int r = i / 5, c = i - r * 5;
            const float* yr = z + r * kSmZs;
            const float* w = cb + 128 + c * 128;
            float s = 0.f;
#pragma unroll 8
            for (int f = 0; f < 128; ++f) s = fmaf(w[f], yr[f], s);
            const int j = r / N, n = r - j * N;
            p.logits[((size_t)n * p.B + (g0 + j)) * 5 + c] = s + cb[768 + c];
        }
    }
    GNNPP_STAMP(blockIdx.x, 13, tid == 0);
}

// ---------------------------------------------------------------------------------------------------------------
// lsigf_pipe_b3_kernel: the same filter as a PRODUCER / CONSUMER pipeline -- one persistent workgroup of EIGHT waves per
// CU that loops over groups of <= 64 rows (whole graphs).  Waves 0..3 (consumers) only contract: tap k of group i from
// plane buffer PB[t & 1], stage t = i K + k; waves 4..7 (producers) prepare what the NEXT stage contracts, into
// PB[(t + 1) & 1]: the shifted signal z_{k+1} = S^T z_k of the same group (dense fp32 MFMA, in place in z, planes written
// by the producer of the value), or -- during a group's last tap -- z_0 of the next group (its rows and GSO blocks were
// requested one group ahead and wait in the producers' registers).  One workgroup barrier per stage.  In
// lsigf_small_b3_kernel the phases of ONE group serialise (shift -> barrier -> contraction -> barrier ..) and two
// workgroups per CU overlap them only by chance: 48.6 % matrix-pipe busy; here the matrix pipe always has a
// contraction to run while shifts, conversions, staging and global-memory latency happen on the other four waves.
//   LDS: z [64][136] fp32 (producers only) | PB [2][64][800 B] | GSO blocks [<= 12][16][17] | bias -- 151 KB, one
//   workgroup per CU (256 registers per wave for both roles).  Output: 16-byte stores straight from the accumulators
//   (no LDS room for a transposition; the stores overlap the producers' work of the next stage).
constexpr int kPipeRows = 64;
constexpr int kPipeZBytes = kPipeRows * kSmZs * 4;                 // 34 816
constexpr int kPipePBytes = kPipeRows * kSmPRow;                   // 51 200 per buffer
constexpr int kPipeSOff = kPipeZBytes + 2 * kPipePBytes;           // GSO blocks
constexpr int kPipeCOff = kPipeSOff + kSmMaxGraphs * kSmSBlock * 4;   // bias [128]
constexpr int kPipeSmem = kPipeCOff + 128 * 4;
static_assert(kPipeSmem <= kLdsBytes, "LDS budget");

// of nu shift units (two per graph) the producer waves take the first 2/3 (rounded up to their four waves)
__host__ __device__ inline int pipe_producer_units(int nu) { const int pu = ((2 * nu + 2) / 3 + 3) & ~3; return pu < nu ? pu : nu; }

template <int NS4>
__global__ __launch_bounds__(512, 1) void lsigf_pipe_b3_kernel(const LsigfArgs p) {
    constexpr int RT = kPipeRows / 16, XV = kPipeRows * 32 / 256;  // row tiles; 16-byte pieces of x per producer thread
    constexpr int SV = (kSmMaxGraphs * kSmSBlock + 255) / 256;     // GSO words per producer thread
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    float* const z = reinterpret_cast<float*>(gnnpp_smem);
    float* const Ssm = reinterpret_cast<float*>(gnnpp_smem + kPipeSOff);
    float* const cb = reinterpret_cast<float*>(gnnpp_smem + kPipeCOff);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave8 >= 4;                              // (wave-uniform)
    const int wave = wave8 & 3, ptid = tid & 255;
    const int a = lane & 15, q = lane >> 4;
    const int N = p.N, K = p.K;
    const int ngroups = (p.B + p.gpw - 1) / p.gpw;
    const int mine = (ngroups - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // groups of this workgroup
    const int T = mine * K;                                        // stages
    auto group_of = [&](int i) { return (int)blockIdx.x + i * (int)gridDim.x; };
    auto rows_of = [&](int grp) { return min(p.gpw, p.B - grp * p.gpw) * N; };

    if (tid < 128) cb[tid] = p.bias ? p.bias[tid] : 0.f;

    // ---- producers' state: the rows / GSO words of ONE group in flight in registers
    v4f xv[XV];
    float sv[SV];
    auto s_load = [&](int g0, int ng, int idx) -> float {
        const int j = idx / kSmSBlock, e = idx - j * kSmSBlock;
        const int m = e / 17, n = e - m * 17;
        if (j >= ng || m >= N || n >= N) return 0.f;
        const size_t gi = ((size_t)(g0 + j) * N + m) * N + n;
        return p.s_is_f64 ? (float)reinterpret_cast<const double*>(p.S)[gi] : reinterpret_cast<const float*>(p.S)[gi];
    };
    auto request = [&](int grp) {
        const int g0 = grp * p.gpw, ng = min(p.gpw, p.B - g0), rows = ng * N;
        const v4f* xs = reinterpret_cast<const v4f*>(p.x + (size_t)g0 * N * 128);
#pragma unroll
        for (int u = 0; u < XV; ++u) {
            const int i = ptid + u * 256;
            xv[u] = i < rows * 32 ? xs[i] : vzero();
        }
        if (K > 1) {
#pragma unroll
            for (int u = 0; u < SV; ++u) sv[u] = s_load(g0, ng, ptid + u * 256);
        }
    };
    auto stage_group = [&](int grp, char* PBn) {                   // registers -> z (fp32), planes of z_0, GSO blocks
        const int rows = rows_of(grp);
#pragma unroll
        for (int u = 0; u < XV; ++u) {
            const int i = ptid + u * 256;
            v2f pl[3];
            b3_split4(xv[u], pl);
            const int r = i >> 5, c4 = i & 31;
            // (every row of z is written, zeros past the group's rows: the dense shift multiplies the rows behind a
            // graph by zero weights, and 0 x stale-NaN would still be NaN)
            if (K > 1 && !GNNPP_ABLATE(p, 0x100)) *reinterpret_cast<v4f*>(z + r * kSmZs + 4 * c4) = xv[u];
            if (i < rows * 32 && !GNNPP_ABLATE(p, 0x200)) {
#pragma unroll
                for (int pp = 0; pp < 3; ++pp) *reinterpret_cast<v2f*>(PBn + r * kSmPRow + pp * 256 + 8 * c4) = pl[pp];
            }
        }
        if (K > 1) {
#pragma unroll
            for (int u = 0; u < SV; ++u)
                if (ptid + u * 256 < kSmMaxGraphs * kSmSBlock) Ssm[ptid + u * 256] = sv[u];
        }
    };
    // shift z_k -> z_{k+1} of the staged group, in place; planes into PBn.  Unit = (graph j, half hf): four 16-feature
    // tiles of D[feature][node n] = sum_m z_k[m][feature] S[m][n] on the exact fp32 MFMA (all m ascending: bit-identical to
    // the sparse gather); units are dealt round-robin to the four producer waves (the two halves of a graph touch disjoint
    // feature columns: no cross-wave hazard in the in-place update).  (A VALU fmaf-chain form of the same product --
    // bit-identical, no matrix-pipe traffic -- was measured at 7 us per stage: ten dependent LDS round trips per graph.)
    // (units [u0, u1) with stride 4 from `wave`: the producers take the first two thirds, the consumers -- idle once their
    // contraction is done -- the rest)
    auto shift_group = [&](int grp, char* PBn, int u0, int u1, bool stamp = false) {
        GNNPP_STAMP(blockIdx.x, 8, tid == 256 && stamp);
        for (int unit = u0 + wave; unit < u1; unit += 4) {
            const int j = unit >> 1, hf = unit & 1, r0 = j * N;
            float Sb[NS4];
#pragma unroll
            for (int s4 = 0; s4 < NS4; ++s4) Sb[s4] = Ssm[j * kSmSBlock + (4 * s4 + q) * 17 + a];
            v4f d[4];
#pragma unroll
            for (int f4 = 0; f4 < 4; ++f4) {
                const int ft = 4 * hf + f4;
                d[f4] = vzero();
#pragma unroll
                for (int s4 = 0; s4 < NS4; ++s4) {
                    const int m = r0 + min(4 * s4 + q, N - 1);     // (k-slots past the graph: its OWN last row x zero weight)
                    d[f4] = mfma16(GNNPP_ABLATE(p, 0x40) ? Sb[s4] : z[m * kSmZs + 16 * ft + a], Sb[s4], d[f4]);
                }
            }
            __builtin_amdgcn_wave_barrier();                       // every row of these columns has been read
            GNNPP_STAMP(blockIdx.x, 9, tid == 256 && stamp && unit == u0 + wave && d[3][0] != 12345.f);
#pragma unroll
            for (int f4 = 0; f4 < 4; ++f4) {                       // lane (q, a): node a, features 16 ft + 4 q ..
                const int ft = 4 * hf + f4;
                v2f pl[3];
                b3_split4(d[f4], pl);
                if (a < N) {
                    const int r = r0 + a;
                    if (!GNNPP_ABLATE(p, 0x20)) *reinterpret_cast<v4f*>(z + r * kSmZs + 16 * ft + 4 * q) = d[f4];
                    if (!GNNPP_ABLATE(p, 0x10)) {
#pragma unroll
                        for (int pp = 0; pp < 3; ++pp)
                            *reinterpret_cast<v2f*>(PBn + r * kSmPRow + pp * 256 + (16 * ft + 4 * q) * 2) = pl[pp];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            GNNPP_STAMP(blockIdx.x, 10, tid == 256 && stamp && unit == u0 + wave);
        }
    };

    // One workgroup barrier per stage, LDS-only: s_waitcnt lgkmcnt(0) + s_barrier.  __syncthreads() also waits for every
    // outstanding GLOBAL access (vmcnt(0)): the consumers' next-tap fragments and output stores, the producers' prefetch of
    // the group after next -- 1.7 us per stage measured.  What crosses the barrier is LDS data only.
    auto stage_barrier = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
        __syncthreads();
#endif
    };

    // The two roles run SEPARATE loops with the same number of workgroup barriers (one after the prologue, one per stage):
    // in one shared loop the compiler keeps the registers of both roles alive for every wave (A fragments + accumulators
    // + the group in flight = 1.2 KB of scratch per lane); a wave-uniform branch around a loop costs nothing.
    if (producer) {
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_s_setprio(3);                             // the producers' few MFMAs and LDS operations go first: the
#endif                                                             // stage ends when THEY are done, the consumers have slack
        // ---- prologue: the first group staged into PB[0]; the second one requested
        request(group_of(0));
        stage_group(group_of(0), gnnpp_smem + kPipeZBytes);
        request(group_of(min(1, mine - 1)));
        __syncthreads();
        // (nested loops, and the prefetch of the group after next UNCONDITIONAL -- a clamped index re-requests the last
        // group for nothing: as a conditional, loop-carried update the registers of the group in flight became a PHI that
        // the compiler resolved with vmcnt(0) + register copies in front of the barrier, i.e. the producers sat out the
        // whole HBM latency of the loads they had just issued, 1.6 us per group)
        int t = 0;
        for (int i = 0; i < mine; ++i) {
            for (int k = 0; k + 1 < K; ++k, ++t) {
                const int nu = 2 * min(p.gpw, p.B - group_of(i) * p.gpw);
                const bool stamp_s = i == mine - 1 && k == 0;
                GNNPP_STAMP(blockIdx.x, 0, tid == 256 && stamp_s);
                shift_group(group_of(i), gnnpp_smem + kPipeZBytes + ((t + 1) & 1) * kPipePBytes, 0,
                            pipe_producer_units(nu), stamp_s);
                GNNPP_STAMP(blockIdx.x, 1, tid == 256 && stamp_s);
                stage_barrier();                                   // PB[(t+1)&1] complete, PB[t&1] free
                GNNPP_STAMP(blockIdx.x, 3, tid == 256 && stamp_s);
            }
            const bool stamp_g = i == mine - 2;
            GNNPP_STAMP(blockIdx.x, 4, tid == 256 && stamp_g);
            if (i + 1 < mine) stage_group(group_of(i + 1), gnnpp_smem + kPipeZBytes + ((t + 1) & 1) * kPipePBytes);
            request(group_of(min(i + 2, mine - 1)));               // (its data is staged a whole group later)
            GNNPP_STAMP(blockIdx.x, 5, tid == 256 && stamp_g);
            stage_barrier();
            GNNPP_STAMP(blockIdx.x, 7, tid == 256 && stamp_g);
            ++t;
        }
        return;
    }

    // ---- consumers: this wave's two channel tiles of the current tap (24 plane fragments), the accumulators
    constexpr size_t tap_stride = (size_t)8 * 4 * 768;
    v4f A[2][12];
    auto load_tap = [&](int tap) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            // (uniform base + 32-bit lane index: scalar-base addressing, no 64-bit address registers per load -- with
            // per-lane 64-bit pointers the loads were issued in three batches with waits in between)
            const v4f* wt = reinterpret_cast<const v4f*>(p.wpk_b + tap * tap_stride + (size_t)(2 * wave + m) * 12 * 256);
#pragma unroll
            for (int i = 0; i < 12; ++i) A[m][i] = wt[i * 64 + (unsigned)lane];
        }
    };
    v4f acc[RT][2], acc2[RT][2];                                   // [row tile][channel tile]: hh | the five smaller terms
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m) { acc[t][m] = vzero(); acc2[t][m] = vzero(); }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const int i = t / K, k = t - i * K;
        const int grp = group_of(i);
        const char* const PBc = gnnpp_smem + kPipeZBytes + (t & 1) * kPipePBytes;
        const int rows = rows_of(grp);
        const int ntile = (rows + 15) >> 4;
        // This stage's fragments are requested HERE and used below -- not carried around the loop: a loop-carried fragment
        // array (requested at the end of the previous stage) made the compiler insert vmcnt(0) + 48 register copies in
        // front of every barrier, the whole L2 latency exposed per stage (1.7 us measured).  The latency is covered by the
        // consumers' share of the shift (they would otherwise wait for the producers at the barrier).
        load_tap(k);
        if (k + 1 < K) {
            const int nu = 2 * min(p.gpw, p.B - grp * p.gpw);
            shift_group(grp, gnnpp_smem + kPipeZBytes + ((t + 1) & 1) * kPipePBytes, pipe_producer_units(nu), nu);
        }
#pragma unroll
        for (int tl = 0; tl < RT; ++tl) {
            if (tl < ntile) {
                const int pr = min(tl * 16 + a, rows - 1);         // (rows >= `rows`: copies, never stored)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const char* zrow = PBc + pr * kSmPRow + kb * 64 + q * 16;
                    v8b Bp[3];
#pragma unroll
                    for (int pp = 0; pp < 3; ++pp)
                        Bp[pp] = GNNPP_ABLATE(p, 0x80) ? as_b8(A[0][pp]) : as_b8(*reinterpret_cast<const v4f*>(zrow + pp * 256));
#pragma unroll
                    for (int term = 0; term < kB3Terms; ++term)
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const v8b Ap = as_b8(A[m][3 * kb + b3_term_a(term)]);
                            if (term == kB3Terms - 1) acc[tl][m] = mfma16b(Ap, Bp[0], acc[tl][m]);
                            else acc2[tl][m] = mfma16b(Ap, Bp[b3_term_b(term)], acc2[tl][m]);
                        }
                }
            }
        }
        GNNPP_STAMP(blockIdx.x, (i == mine - 1 && k == 0 && K > 1) ? 2 : 6,
                    tid == 0 && ((i == mine - 1 && k == 0 && K > 1) || (i == mine - 2 && k == K - 1)));
        if (k + 1 == K) {
            // epilogue: bias (+ ReLU), 16-byte stores from the accumulators (a wave's two tiles = 128 contiguous bytes
            // of a row; the other consumers fill the rest of the 512-byte row), accumulators cleared
            const int g0 = grp * p.gpw;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int f0 = (2 * wave + m) * 16 + 4 * q;
                const v4f bv = *reinterpret_cast<const v4f*>(cb + f0);
#pragma unroll
                for (int tl = 0; tl < RT; ++tl) {
                    const int row = tl * 16 + a;
                    if (tl < ntile && row < rows) {
                        v4f v = (acc[tl][m] + acc2[tl][m]) + bv;
                        if (p.relu) v = vrelu(v);
                        *reinterpret_cast<v4f*>(p.y + ((size_t)g0 * N + row) * 128 + f0) = v;
                    }
                    acc[tl][m] = vzero();
                    acc2[tl][m] = vzero();
                }
            }
        }
        stage_barrier();                                           // PB[(t+1)&1] complete, PB[t&1] free
    }
}

std::atomic<int> g_filter_pipe_grid{0};                    // GNNPP_TUNE_FILTER_PIPE_GRID: persistent workgroups of the pipeline
                                                           // kernel, 0 = one per CU (256)
std::atomic<int> g_filter_small_rows{0};                   // GNNPP_TUNE_FILTER_SMALL_ROWS: 0 = heuristic, 32 or 48
std::atomic<int> g_filter_small_kernel{1};                 // GNNPP_TUNE_FILTER_SMALL: 0 = never, 1 = heuristic, 2 = whenever the shape fits

template <int ROWS, int NS4>
static void sm_launch_small(const LsigfArgs& a, int grid, hipStream_t st) {
    static LdsAttrOnce once;
    set_lds_attr_once(once, reinterpret_cast<const void*>(&lsigf_small_b3_kernel<ROWS, NS4>), SmLayout<ROWS>::smem(kSmMaxGraphs));
    hipLaunchKernelGGL((lsigf_small_b3_kernel<ROWS, NS4>), dim3(grid), dim3(256), SmLayout<ROWS>::smem(a.gpw), st, a);
}
template <int NS4>
static void sm_launch_pipe(const LsigfArgs& a, int grid, hipStream_t st) {
    static LdsAttrOnce once;
    set_lds_attr_once(once, reinterpret_cast<const void*>(&lsigf_pipe_b3_kernel<NS4>), kPipeSmem);
    hipLaunchKernelGGL((lsigf_pipe_b3_kernel<NS4>), dim3(grid), dim3(512), kPipeSmem, st, a);
}

// Does the planned launch have this kernel's shape?  Returns 1 when not (the caller goes on), 0 / -3 after a launch.
static int lsigf_small_dispatch(LsigfArgs a, hipStream_t st) {
    const int mode = g_filter_small_kernel.load(std::memory_order_relaxed);
    if (!mode || a.prec != kPrecFp32 || a.G != 128 || a.F != 128 ||
        a.F_all != 128 || a.E != 1 || !a.x_node_major || !a.s_batched || a.s_transposed || a.zs || a.bias_per_node ||
        a.Nin != a.N || a.N > kSmMaxNodes || a.K < 1 || (a.y && !a.y_node_major) || (!a.y && !a.act_w) ||
        (reinterpret_cast<uintptr_t>(a.x) & 15) || (a.y && (reinterpret_cast<uintptr_t>(a.y) & 15))
#ifdef GNNPP_MEASURE
        || (a.ablate & 0xf)            // (bits 0x10 .. 0x200 ablate accesses of the pipeline kernel below)
#endif
    )
        return 1;
    // Measured (profiles/r03_filter_sweep.jsonl, N = 10, K = 3): 48-row workgroups win once they fill the chip twice
    // over (B = 2048: 18.6 vs 23.8 us; B = 8192 .. 131072: 1.19 - 1.26 G agent-steps/s vs 1.12 - 1.21 -- fewer tap
    // bytes per agent-step); below that 32-row workgroups spread the graphs over more CUs (B = 512: 11.5 us vs 14.3,
    // and vs 15.2 for the general kernel); a handful of graphs stay on the general kernel's wider workgroups.
    // Throughput regime (>= 16 groups of 64 rows per CU): the producer / consumer pipeline.  Measured (N = 10, K = 3,
    // profiles/r03_filter_sweep.jsonl): 1.03 / 1.16 / 1.25 G agent-steps/s at B = 8 192 / 32 768 / 131 072 against 1.08 /
    // 1.10 / 1.16 for the 48-row kernel.  (FILTER_SMALL = 3 forces it, FILTER_SMALL_ROWS = 64 likewise; no head epilogue)
    {
        const int per64 = kPipeRows / a.N < kSmMaxGraphs ? kPipeRows / a.N : kSmMaxGraphs;
        const int groups = (a.B + per64 - 1) / per64;
        const int rows_knob = g_filter_small_rows.load(std::memory_order_relaxed);
        if (a.y && !a.act_w && (mode == 3 || rows_knob == 64 || (mode == 1 && rows_knob == 0 && groups >= 16 * 256))) {
            a.gpw = per64;
            const int knob = g_filter_pipe_grid.load(std::memory_order_relaxed);
            const int resident = knob ? knob : 256;
            const int grid = groups < resident ? groups : resident;
            switch ((a.N + 3) >> 2) {
                case 1: sm_launch_pipe<1>(a, grid, st); break;
                case 2: sm_launch_pipe<2>(a, grid, st); break;
                case 3: sm_launch_pipe<3>(a, grid, st); break;
                default: sm_launch_pipe<4>(a, grid, st); break;
            }
            return hipGetLastError() == hipSuccess ? 0 : -3;
        }
        if (mode == 3 || rows_knob == 64) return 1;                // (the shape has no pipeline form: the general kernel)
    }
    const int forced = g_filter_small_rows.load(std::memory_order_relaxed);
    const int per48 = 48 / a.N < kSmMaxGraphs ? 48 / a.N : kSmMaxGraphs;
    const int rows = forced ? forced : ((a.B + per48 - 1) / per48 >= 512 ? 48 : 32);
    const int per = rows / a.N;
    a.gpw = per < kSmMaxGraphs ? per : kSmMaxGraphs;
    const int grid = (a.B + a.gpw - 1) / a.gpw;
    if (grid < 64 && mode != 2) return 1;
    const int ns4 = (a.N + 3) >> 2;
    if (rows == 48) {
        switch (ns4) {
            case 1: sm_launch_small<48, 1>(a, grid, st); break;
            case 2: sm_launch_small<48, 2>(a, grid, st); break;
            case 3: sm_launch_small<48, 3>(a, grid, st); break;
            default: sm_launch_small<48, 4>(a, grid, st); break;
        }
    } else {
        switch (ns4) {
            case 1: sm_launch_small<32, 1>(a, grid, st); break;
            case 2: sm_launch_small<32, 2>(a, grid, st); break;
            case 3: sm_launch_small<32, 3>(a, grid, st); break;
            default: sm_launch_small<32, 4>(a, grid, st); break;
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace gnnpp
