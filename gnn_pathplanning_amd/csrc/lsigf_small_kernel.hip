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
//      z   [48][136] fp32    the CURRENT tap signal, shifted IN PLACE: a wave owns whole graphs; the rows of a graph
//                            (N <= 16) are all read into registers -- quarter wave per row, dense ascending-m fmaf
//                            chain over the graph's N rows, bit-identical to the sparse gather (fmaf(0, z, acc) ==
//                            acc) -- before any of them is written back
//      PB  [48][800 B]       the same rows as three bf16 planes, written by whoever produces z_k (the staging loop,
//                            the shift's write-back): no conversion pass
//      S   [graphs][N][N+1]  the GSO slabs, column-major (row n = the weights node n gathers with)
// = 69 KB: TWO workgroups per CU, so one's staging / shift / barrier phases run beside the other's MFMAs.  A wave
// contracts two 16-feature output tiles over all rows (six MFMAs per 32 channels and row tile, hh products and the
// five smaller terms in separate accumulators); the tap's 24 A fragments per wave are requested one tap ahead.
// Epilogue: bias (+ ReLU) -> fp32 rows in z -> fully coalesced 512-byte row stores, or the 128 -> 5 action head.
// HBM traffic is the algorithmic minimum (x and S in, y out, once); the taps stream from L2 (288 KB per workgroup
// at K = 3, 7.2 KB per agent-step at N = 10 -- the reason a workgroup takes 48 rows and not 16).
#include "gnnpp_common.h"

namespace gnnpp {

constexpr int kSmRows = 48;                                // rows (graphs x nodes) of one workgroup = 3 row tiles
constexpr int kSmMaxNodes = 16;
constexpr int kSmZs = 136;                                 // fp32 row stride (floats)
constexpr int kSmPRow = 3 * 256 + 32;                      // plane row stride (bytes)
constexpr int kSmZBytes = kSmRows * kSmZs * 4;             // 26 112
constexpr int kSmPBytes = kSmRows * kSmPRow;               // 38 400
constexpr int kSmSOff = kSmZBytes + kSmPBytes;             // GSO slabs: up to 48 rows x 17 floats
constexpr int kSmSBytes = kSmRows * (kSmMaxNodes + 1) * 4;
constexpr int kSmCOff = kSmSOff + kSmSBytes;               // constants: bias [128] | act_w [5][128] | act_b [5] (+ pad)
constexpr int kSmSmem = kSmCOff + 776 * 4;                 // 70 880 B
static_assert(2 * kSmSmem <= kLdsBytes, "two workgroups per CU");

__global__ __launch_bounds__(256, 2) void lsigf_small_b3_kernel(const LsigfArgs p) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    float* const z = reinterpret_cast<float*>(gnnpp_smem);
    char* const PB = gnnpp_smem + kSmZBytes;
    float* const Ssm = reinterpret_cast<float*>(gnnpp_smem + kSmSOff);
    float* const cb = reinterpret_cast<float*>(gnnpp_smem + kSmCOff);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int a = lane & 15, q = lane >> 4;
    const int N = p.N, K = p.K, Np = N + 1;
    const int g0 = blockIdx.x * p.gpw;
    const int ng = min(p.gpw, p.B - g0);
    const int rows = ng * N;                               // <= 48
    const int ntile = (rows + 15) >> 4;                    // row tiles in use (wave-uniform)

    // ---- every global load of the kernel's front, issued now ------------------------------------------------
    const v4f* xs = reinterpret_cast<const v4f*>(p.x + (size_t)g0 * N * 128);
    v4f xv[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int i = tid + u * 256;
        if (i < rows * 32) xv[u] = xs[i];
    }
    float sv[4];
    const int nS = ng * N * N;                             // <= 4 * 256
    if (K > 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + u * 256;
            if (i < nS) sv[u] = p.s_is_f64 ? (float)(reinterpret_cast<const double*>(p.S)[(size_t)g0 * N * N + i])
                                           : reinterpret_cast<const float*>(p.S)[(size_t)g0 * N * N + i];
        }
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
    for (int u = 0; u < 6; ++u) {
        const int i = tid + u * 256;
        const bool ok = i < rows * 32;
        const v4f v = ok ? xv[u] : vzero();
        v2f pl[3];
        b3_split4(v, pl);
        if (ok) {
            const int r = i >> 5, c4 = i & 31;
            if (K > 1) *reinterpret_cast<v4f*>(z + r * kSmZs + 4 * c4) = v;
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) *reinterpret_cast<v2f*>(PB + r * kSmPRow + pp * 256 + 8 * c4) = pl[pp];
        }
    }
    if (K > 1) {
        const float inv_nn = 1.0f / (float)(N * N), inv_n = 1.0f / (float)N;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + u * 256;
            if (i < nS) {                                  // element (graph j, m, n) -> Ssm[j][n][m]  (exact float
                const int j = (int)(((float)i + 0.5f) * inv_nn);          // reciprocal: i < 2^12)
                const int e = i - j * N * N;
                const int m = (int)(((float)e + 0.5f) * inv_n), n = e - m * N;
                Ssm[(j * N + n) * Np + m] = sv[u];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (tid + u * 256 < 773) cb[tid + u * 256] = cpre[u];
    load_tap(0);                                           // (behind the staged data in the memory queue)
    __syncthreads();

    v4f acc[3][2], acc2[3][2];                             // [row tile][channel tile]: hh | the five smaller terms
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m) { acc[t][m] = vzero(); acc2[t][m] = vzero(); }

    for (int k = 0; k < K; ++k) {
        // ---- contraction of tap k: D[f, row] += W_k[f, g] z_k[row, g] from the planes ---------------------------
#pragma unroll
        for (int t = 0; t < 3; ++t) {
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
        if (k + 1 == K) break;
        load_tap(k + 1);                                   // in flight during the shift
        __syncthreads();                                   // every wave is done with the planes of z_k
        // ---- shift z_k -> z_{k+1} in place: wave w owns graphs w, w + 4, .. -------------------------------------
        {
            const int quarter = lane >> 4, ql = lane & 15;
            const bool last = k + 2 == K;                  // z_{K-1} is only needed as planes
            for (int j = wave; j < ng; j += 4) {
                const float* zg = z + j * N * kSmZs + 4 * ql;
                v4f r0[4], r1[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int n = it * 4 + quarter;
                    r0[it] = vzero(); r1[it] = vzero();
                    if (it * 4 < N) {                      // (wave-uniform)
                        const float* wl = Ssm + (j * N + min(n, N - 1)) * Np;
                        for (int m = 0; m < N; ++m) {
                            const float w = wl[m];
                            const v4f za = *reinterpret_cast<const v4f*>(zg + m * kSmZs);
                            const v4f zb = *reinterpret_cast<const v4f*>(zg + m * kSmZs + 64);
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                r0[it][c] = fmaf(w, za[c], r0[it][c]);
                                r1[it][c] = fmaf(w, zb[c], r1[it][c]);
                            }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();           // every row of the graph is in registers
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int n = it * 4 + quarter;
                    if (it * 4 < N) {
                        v2f p0[3], p1[3];
                        b3_split4(r0[it], p0);
                        b3_split4(r1[it], p1);
                        if (n < N) {
                            const int r = j * N + n;
                            if (!last) {
                                *reinterpret_cast<v4f*>(z + r * kSmZs + 4 * ql) = r0[it];
                                *reinterpret_cast<v4f*>(z + r * kSmZs + 64 + 4 * ql) = r1[it];
                            }
                            char* row = PB + r * kSmPRow;
#pragma unroll
                            for (int pp = 0; pp < 3; ++pp) {
                                *reinterpret_cast<v2f*>(row + pp * 256 + 8 * ql) = p0[pp];
                                *reinterpret_cast<v2f*>(row + pp * 256 + 128 + 8 * ql) = p1[pp];
                            }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();                                   // planes of z_{k+1} visible
    }

    // ---- epilogue: bias (+ ReLU) -> fp32 rows in z -> coalesced store / action head ------------------------------
    __syncthreads();                                       // (K == 1: nobody reads z; K > 1: the last shift is done)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int f0 = (2 * wave + m) * 16 + 4 * q;
        const v4f bv = *reinterpret_cast<const v4f*>(cb + f0);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int row = t * 16 + a;
            if (t < ntile && row < rows) {
                v4f v = (acc[t][m] + acc2[t][m]) + bv;
                if (p.relu) v = vrelu(v);
                *reinterpret_cast<v4f*>(z + row * kSmZs + f0) = v;
            }
        }
    }
    __syncthreads();
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
}

std::atomic<int> g_filter_small_kernel{1};                 // GNNPP_TUNE_FILTER_SMALL: 0 = never, 1 = heuristic, 2 = whenever the shape fits

// Does the planned launch have this kernel's shape?  Returns 1 when not (the caller goes on), 0 / -3 after a launch.
static int lsigf_small_dispatch(LsigfArgs a, hipStream_t st) {
    const int mode = g_filter_small_kernel.load(std::memory_order_relaxed);
    if (!mode || a.prec != kPrecFp32 || a.G != 128 || a.F != 128 ||
        a.F_all != 128 || a.E != 1 || !a.x_node_major || !a.s_batched || a.s_transposed || a.zs || a.bias_per_node ||
        a.Nin != a.N || a.N > kSmMaxNodes || a.K < 1 || (a.y && !a.y_node_major) || (!a.y && !a.act_w) ||
        (reinterpret_cast<uintptr_t>(a.x) & 15) || (a.y && (reinterpret_cast<uintptr_t>(a.y) & 15))
#ifdef GNNPP_MEASURE
        || a.ablate
#endif
    )
        return 1;
    // the throughput regime: enough graphs that every CU gets at least two workgroups; below that the general
    // kernel's wider workgroups win on latency
    a.gpw = kSmRows / a.N;
    const int grid = (a.B + a.gpw - 1) / a.gpw;
    if (grid < 512 && mode != 2) return 1;
    static LdsAttrOnce once;
    set_lds_attr_once(once, reinterpret_cast<const void*>(&lsigf_small_b3_kernel), kSmSmem);
    hipLaunchKernelGGL(lsigf_small_b3_kernel, dim3(grid), dim3(256), kSmSmem, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace gnnpp
