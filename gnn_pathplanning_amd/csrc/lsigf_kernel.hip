// K-tap graph-shift filter (LSIGF / BatchLSIGF) for gfx950.
//
//   y[b,n,:] = bias + sum_e sum_k W[:,e,k,:] . z_{e,k}[b,n,:],   z_{e,0} = x,
//   z_{e,k}[b,n,:] = sum_m S[b,e,m,n] * z_{e,k-1}[b,m,:]          (node n gathers COLUMN n of S)
//
// Follows utils/graphUtils/graphML.py:48-141 (LSIGF) and :2273-2367 (BatchLSIGF) of the
// reference, which build the taps with dense batched matmuls, K-1 torch.cat re-copies and a
// materialised permute.  Here one workgroup owns `gpw` whole graphs (R = gpw*N rows <= 112):
//
//   1. the feature rows x[b] are staged ONCE into LDS, node-major, row stride G+8 floats
//      (coalesced 512-byte row reads; the +8 makes the MFMA B-fragment ds_read_b128 conflict free);
//   2. the dense S slabs are staged into LDS (fp64 -> fp32 on the fly, like `S.float()`);
//   3. shift k: ONE WAVEFRONT PER NODE.  Lane m reads S[m,n], a ballot compacts the column to its
//      non-zeros (exact: structural zeros contribute nothing), and for each neighbour the wave
//      reads that neighbour's 512-byte feature row from LDS (ds_read_b64 per lane, conflict free)
//      and accumulates 2 features per lane.  z ping-pongs between two LDS buffers;
//   4. contraction of tap k right after its shift: D[f, row] += W_k[f, g] z_k[row, g] on fp32 MFMA
//      16x16x4 with the accumulators living in registers across all taps; W_k fragments stream
//      from L2 in a pre-packed order (one 16-byte load per lane per four MFMAs);
//   5. epilogue: + bias, optional ReLU, staged through LDS so the store is coalesced in either
//      output layout; optionally the 128 -> 5 action head (decentralplanner.py:304-315) is fused.
//
// HBM traffic per launch is the algorithmic minimum: x, S and y once (+ the packed taps, L2 hits).
#include "gnnpp_common.h"

namespace gnnpp {

struct LsigfArgs {
    const float* x;
    const void* S;
    const float* wpk;      // packed taps, see pack_filter_kernel
    const float* bias;     // [F] or nullptr
    float* y;              // may be nullptr when only the action head is wanted
    const float* act_w;    // [5,F] or nullptr: fused action head
    const float* act_b;    // [5]
    float* logits;         // [N,B,5]
    int B, N, Nin, G, F, K, E;
    int NG, MT;            // ceil(G/16), ceil(F/16)   (F <= 128 per launch -> MT <= 8)
    int zstride;           // LDS row stride in floats = 16*max(NG,MT) + 8
    int gpw;               // graphs per workgroup
    int Ns;                // LDS row stride of an S slab (odd)
    int s_is_f64, s_batched, x_node_major, y_node_major, relu;
};

// Re-order h[F,E,K,G] into MFMA A fragments: block (e,k,mt,gg) holds, for lane l = q*16 + i and
// k-step s, h[f = mt*16 + i][e][k][g = gg*16 + q*4 + s]  (0 outside F x G).
__global__ void pack_filter_kernel(const float* __restrict__ h, float* __restrict__ packed,
                                   int G, int F, int K, int E) {
    const int NG = (G + 15) / 16, MT = (F + 15) / 16;
    const size_t total = (size_t)E * K * MT * NG * 256;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int s = idx & 3;
        const int l = (idx >> 2) & 63;
        size_t blk = idx >> 8;
        const int gg = blk % NG; blk /= NG;
        const int mt = blk % MT; blk /= MT;
        const int k = blk % K;
        const int e = blk / K;
        const int f = mt * 16 + (l & 15);
        const int g = gg * 16 + (l >> 4) * 4 + s;
        float v = 0.f;
        if (f < F && g < G) v = h[(((size_t)f * E + e) * K + k) * G + g];
        packed[idx] = v;
    }
}

// One shift for the rows owned by this wave: z_cur[r,:] = sum_m S[m, n(r)] * z_prev[m,:].
// Lane m reads S[m,n]; the ballot is the column's sparsity pattern; neighbours are consumed four
// at a time so four independent LDS row reads are in flight (ILP), 2 features per lane.
__device__ __forceinline__ void gather_rows(const LsigfArgs& p, const float* __restrict__ Sl,
                                            const float* __restrict__ zprev,
                                            float* __restrict__ zcur, int R, int wave, int lane) {
    const int N = p.N, zs = p.zstride, GP = p.NG * 16;
    for (int r = wave; r < R; r += kWaves) {
        const int j = r / N, n = r - j * N;
        const float* Scol = Sl + j * N * p.Ns + n;
        const float* zg = zprev + j * N * zs;
        for (int c0 = 0; c0 < GP; c0 += 128) {
            const int col = c0 + 2 * lane;
            const bool live = col < GP;
            const int colc = live ? col : 0;
            v2f s2 = {0.f, 0.f};
            for (int m0 = 0; m0 < N; m0 += 64) {
                const int m = m0 + lane;
                const float sv = (m < N) ? Scol[m * p.Ns] : 0.f;
                unsigned long long mask = __ballot(sv != 0.f);
                while (mask) {
                    int mm[4];
                    float sc[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (mask) {
                            mm[u] = __ffsll((long long)mask) - 1;
                            mask &= mask - 1;
                            sc[u] = wave_read_lane(sv, mm[u]);
                        } else {
                            mm[u] = mm[0];             // harmless re-read, weight 0
                            sc[u] = 0.f;
                        }
                    }
                    v2f zv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        zv[u] = *reinterpret_cast<const v2f*>(zg + (m0 + mm[u]) * zs + colc);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        s2[0] = fmaf(sc[u], zv[u][0], s2[0]);
                        s2[1] = fmaf(sc[u], zv[u][1], s2[1]);
                    }
                }
            }
            if (live) *reinterpret_cast<v2f*>(zcur + r * zs + col) = s2;
        }
    }
}

__device__ __forceinline__ void stage_x(const LsigfArgs& p, float* __restrict__ z0, int g0, int ng,
                                        int tid, bool rezero) {
    const int N = p.N, zs = p.zstride, R = ng * N;
    if (p.x_node_major) {
        const float* xs = p.x + (size_t)g0 * N * p.G;
        if ((p.G & 3) == 0) {
            const int G4 = p.G >> 2;
            for (int i = tid; i < R * G4; i += kThreads) {
                const int r = i / G4, c = i - r * G4;
                *reinterpret_cast<v4f*>(z0 + r * zs + 4 * c) =
                    *reinterpret_cast<const v4f*>(xs + (size_t)r * p.G + 4 * c);
            }
        } else {
            for (int i = tid; i < R * p.G; i += kThreads) {
                const int r = i / p.G, c = i - r * p.G;
                z0[r * zs + c] = xs[(size_t)r * p.G + c];
            }
        }
    } else {
        // x[b][g][n], n < Nin; linear (coalesced) walk over each graph's G x Nin slab
        const int slab = p.G * p.Nin;
        for (int j = 0; j < ng; ++j) {
            const float* xs = p.x + (size_t)(g0 + j) * slab;
            for (int i = tid; i < slab; i += kThreads) {
                const int g = i / p.Nin, n = i - g * p.Nin;
                z0[(j * N + n) * zs + g] = xs[i];
            }
            if (rezero)                                 // rows n >= Nin must be zero again
                for (int i = tid; i < (N - p.Nin) * p.G; i += kThreads) {
                    const int n = p.Nin + i / p.G, g = i % p.G;
                    z0[(j * N + n) * zs + g] = 0.f;
                }
        }
    }
}

// RT  = 16-row MFMA tiles per workgroup, MTW = output-channel tiles per wave (1: F<=64, 2: F<=128),
// NGT = compile-time number of 16-wide input-feature groups (8 for G = 128; 0 = run-time NG).
template <int RT, int MTW, int NGT>
__global__ __launch_bounds__(kThreads) void lsigf_kernel(const LsigfArgs p) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int a = lane & 15;      // row inside a 16-row tile (MFMA j)
    const int q = lane >> 4;      // MFMA k slot

    const int g0 = blockIdx.x * p.gpw;                 // first graph of this workgroup
    const int ng = min(p.gpw, p.B - g0);               // graphs actually present
    const int N = p.N;
    const int R = ng * N;                              // valid rows
    const int zs = p.zstride;
    constexpr int ROWS = RT * 16;
    const int NG = NGT ? NGT : p.NG;
    constexpr int NGA = NGT ? NGT : 1;

    float* zbuf0 = reinterpret_cast<float*>(gnnpp_smem);
    float* zbuf1 = zbuf0 + ROWS * zs;
    float* Sl = zbuf1 + ROWS * zs;                     // [gpw][N][Ns]

    // Tap weights of the first tap: issued first so their L2 latency hides behind the staging.
    // Packed block (e,k,mt,gg): 64 lanes x 4 floats = the A fragments of four MFMA k-steps.
    const int ntaps = p.E * p.K;
    const size_t tap_stride = (size_t)p.MT * NG * 256;
    v4f Acur[NGA][MTW], Anxt[NGA][MTW];
    auto load_tap = [&](v4f (&A)[NGA][MTW], int tap) {
        if (NGT) {
            const float* wt = p.wpk + tap * tap_stride + lane * 4;
#pragma unroll
            for (int gg = 0; gg < NGA; ++gg)
#pragma unroll
                for (int i = 0; i < MTW; ++i) {
                    const int mt = wave + kWaves * i;
                    A[gg][i] = (mt < p.MT) ? *reinterpret_cast<const v4f*>(
                                                 wt + (size_t)(mt * NGA + gg) * 256)
                                           : vzero();
                }
        }
    };
    load_tap(Acur, 0);

    // ---- zero both z buffers (pad rows / pad columns must be finite zeros) --------------------
    {
        v4f* zz = reinterpret_cast<v4f*>(zbuf0);
        const int n4 = (2 * ROWS * zs) >> 2;           // zs is a multiple of 8
        for (int i = tid; i < n4; i += kThreads) zz[i] = vzero();
    }
    __syncthreads();
    stage_x(p, zbuf0, g0, ng, tid, false);             // z_0 (node-major rows)

    v4f acc[MTW][RT];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[i][rt] = vzero();

    const int NN = N * N;
    int tap = 0;
    for (int e = 0; e < p.E; ++e) {
        // ---- stage the S slabs of edge feature e (only needed when K > 1) ---------------------
        if (p.K > 1) {
            if (e > 0) __syncthreads();                // previous e is done with Sl and the z's
            for (int j = 0; j < ng; ++j) {
                const size_t sidx = ((size_t)(p.s_batched ? (g0 + j) * p.E : 0) + e) * NN;
                float* dst = Sl + j * N * p.Ns;
                if (p.s_is_f64) {
                    const double* src = reinterpret_cast<const double*>(p.S) + sidx;
                    for (int i = tid; i < NN; i += kThreads) {
                        const int m = i / N, n = i - m * N;
                        dst[m * p.Ns + n] = (float)src[i];
                    }
                } else {
                    const float* src = reinterpret_cast<const float*>(p.S) + sidx;
                    for (int i = tid; i < NN; i += kThreads) {
                        const int m = i / N, n = i - m * N;
                        dst[m * p.Ns + n] = src[i];
                    }
                }
            }
            // z_{e,0} = x: the ping-pong overwrote it when K > 2, so edge features e > 0 re-stage
            // it (E > 1 is outside the planner's configs: simple and correct beats fast here).
            if (e > 0 && p.K > 2) stage_x(p, zbuf0, g0, ng, tid, true);
        }
        __syncthreads();                               // z_0 (and Sl) visible

        for (int k = 0; k < p.K; ++k, ++tap) {
            float* zcur = (k & 1) ? zbuf1 : zbuf0;
            if (tap + 1 < ntaps) load_tap(Anxt, tap + 1);       // in flight during the shift
            if (k > 0) {
                gather_rows(p, Sl, (k & 1) ? zbuf0 : zbuf1, zcur, R, wave, lane);
                __syncthreads();
            }
            // ---- contraction of tap (e,k) on MFMA: D[f, row] += W[f, g] z[row, g] --------------
            if (NGT) {
#pragma unroll
                for (int gg = 0; gg < NGA; ++gg) {
                    v4f Bf[RT];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
                        Bf[rt] = *reinterpret_cast<const v4f*>(zcur + (rt * 16 + a) * zs +
                                                               gg * 16 + q * 4);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int i = 0; i < MTW; ++i)
#pragma unroll
                            for (int rt = 0; rt < RT; ++rt)
                                acc[i][rt] = mfma16(Acur[gg][i][s], Bf[rt][s], acc[i][rt]);
                }
#pragma unroll
                for (int gg = 0; gg < NGA; ++gg)
#pragma unroll
                    for (int i = 0; i < MTW; ++i) Acur[gg][i] = Anxt[gg][i];
            } else {
                const float* wtap = p.wpk + tap * tap_stride;
                for (int gg = 0; gg < NG; ++gg) {
                    v4f A[MTW];
#pragma unroll
                    for (int i = 0; i < MTW; ++i) {
                        const int mt = wave + kWaves * i;
                        A[i] = (mt < p.MT) ? *reinterpret_cast<const v4f*>(
                                                 wtap + ((size_t)(mt * NG + gg) * 64 + lane) * 4)
                                           : vzero();
                    }
                    v4f Bf[RT];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
                        Bf[rt] = *reinterpret_cast<const v4f*>(zcur + (rt * 16 + a) * zs +
                                                               gg * 16 + q * 4);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int i = 0; i < MTW; ++i)
#pragma unroll
                            for (int rt = 0; rt < RT; ++rt)
                                acc[i][rt] = mfma16(A[i][s], Bf[rt][s], acc[i][rt]);
                }
            }
        }
    }

    // ---- epilogue: bias (+ReLU) -> LDS [row][f] -> coalesced store / fused action head --------
    __syncthreads();                                   // every wave is done reading z
    float* ybuf = zbuf0;
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int mt = wave + kWaves * i;
        if (mt < p.MT) {
            const int f0 = mt * 16 + q * 4;
            v4f bv = vzero();
            if (p.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[r] = (f0 + r < p.F) ? p.bias[f0 + r] : 0.f;
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                v4f v = acc[i][rt] + bv;
                if (p.relu) v = vrelu(v);
                *reinterpret_cast<v4f*>(ybuf + (rt * 16 + a) * zs + f0) = v;
            }
        }
    }
    __syncthreads();

    if (p.y) {
        if (p.y_node_major) {
            float* yd = p.y + (size_t)g0 * N * p.F;
            if ((p.F & 3) == 0) {
                const int F4 = p.F >> 2;
                for (int i = tid; i < R * F4; i += kThreads) {
                    const int r = i / F4, c = i - r * F4;
                    *reinterpret_cast<v4f*>(yd + (size_t)r * p.F + 4 * c) =
                        *reinterpret_cast<const v4f*>(ybuf + r * zs + 4 * c);
                }
            } else {
                for (int i = tid; i < R * p.F; i += kThreads) {
                    const int r = i / p.F, c = i - r * p.F;
                    yd[(size_t)r * p.F + c] = ybuf[r * zs + c];
                }
            }
        } else {
            const int slab = p.F * p.Nin;
            for (int j = 0; j < ng; ++j) {
                float* yd = p.y + (size_t)(g0 + j) * slab;
                for (int i = tid; i < slab; i += kThreads) {
                    const int f = i / p.Nin, n = i - f * p.Nin;
                    yd[i] = ybuf[(j * N + n) * zs + f];
                }
            }
        }
    }
    if (p.act_w) {
        // Action head on MFMA: D[a5, row] = sum_f act_w[a5, f] * y[row, f]; the A fragment is read
        // straight from act_w[5,F] (rows >= 5 are zero), the B fragment from the staged y tile.
        const int i5 = lane & 15;
        for (int rt = wave; rt < RT; rt += kWaves) {
            v4f d = vzero();
            for (int gg = 0; gg < p.MT; ++gg) {
                const int f0 = gg * 16 + q * 4;
                v4f A = vzero();
                if (i5 < 5) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        if (f0 + s < p.F) A[s] = p.act_w[i5 * p.F + f0 + s];
                }
                const v4f Bv = *reinterpret_cast<const v4f*>(ybuf + (rt * 16 + a) * zs + f0);
                d = mfma16x4(A, Bv, d);
            }
            const int r = rt * 16 + a;                  // this lane's row; it holds a5 = 4*q + reg
            if (r < R && q < 2) {
                const int j = r / N, n = r - j * N;
                float* dst = p.logits + ((size_t)n * p.B + (g0 + j)) * 5;
                if (q == 0) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) dst[t] = d[t] + p.act_b[t];
                } else {
                    dst[4] = d[0] + p.act_b[4];
                }
            }
        }
    }
}

// logits [N,B,5] -> actions [B,N]; first maximum wins (torch.max semantics).
__global__ void decode_actions_kernel(const float* __restrict__ logits, int* __restrict__ actions,
                                      int B, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * N) return;
    const int b = i / N, n = i - b * N;
    const float* l = logits + ((size_t)n * B + b) * 5;
    int best = 0;
    float bv = l[0];
#pragma unroll
    for (int k = 1; k < 5; ++k)
        if (l[k] > bv) { bv = l[k]; best = k; }
    actions[i] = best;
}

// ---- host-side launcher -----------------------------------------------------------------------
int g_filter_gpw = 0;               // 0: heuristic below; > 0: forced graphs per workgroup (tuning)

template <int RT, int MTW, int NGT>
static hipError_t launch_one(const LsigfArgs& a, int grid, size_t smem, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lsigf_kernel<RT, MTW, NGT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        attr_set = true;
    }
    hipLaunchKernelGGL((lsigf_kernel<RT, MTW, NGT>), dim3(grid), dim3(kThreads), smem, st, a);
    return hipGetLastError();
}

template <int RT>
static hipError_t launch_rt(const LsigfArgs& a, int grid, size_t smem, hipStream_t st) {
    if (a.NG == 8 && a.MT > kWaves) return launch_one<RT, 2, 8>(a, grid, smem, st);   // G = F = 128
    if (a.MT <= kWaves) return launch_one<RT, 1, 0>(a, grid, smem, st);
    return launch_one<RT, 2, 0>(a, grid, smem, st);
}

static size_t lsigf_smem(const LsigfArgs& a, int gpw) {
    const int rt = (gpw * a.N + 15) / 16;
    return (size_t)2 * rt * 16 * a.zstride * 4 + (size_t)gpw * a.N * a.Ns * 4;
}

// Chooses graphs-per-workgroup, checks the LDS budget and launches.  Returns a GNNPP_* code.
int lsigf_launch(LsigfArgs a, hipStream_t st) {
    a.NG = (a.G + 15) / 16;
    a.MT = (a.F + 15) / 16;
    if (a.MT > 2 * kWaves) return -2;                 // F > 128: the caller splits F
    const int wide = a.NG > a.MT ? a.NG : a.MT;
    a.zstride = 16 * wide + 8;
    a.Ns = a.N | 1;
    if (a.N > 112) return -2;
    // graphs per workgroup: fill the 16-row MFMA tiles, but keep the 256 CUs busy.  Cost model:
    // rounds over the chip x (fixed staging/latency cost + MFMA work per row tile).
    int best = 1;
    double best_cost = 1e30;
    const int max_gpw = 112 / a.N;
    for (int g = 1; g <= max_gpw && g <= a.B; ++g) {
        if (lsigf_smem(a, g) > (size_t)kLdsBytes) break;
        const int rt = (g * a.N + 15) / 16;
        const int wgs = (a.B + g - 1) / g;
        const int rounds = (wgs + 255) / 256;
        const double cost = rounds * (1.0 + rt);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = g; }
    }
    if (g_filter_gpw > 0 && g_filter_gpw <= max_gpw && g_filter_gpw <= a.B &&
        lsigf_smem(a, g_filter_gpw) <= (size_t)kLdsBytes)
        best = g_filter_gpw;
    a.gpw = best;
    const int rt = (a.gpw * a.N + 15) / 16;
    const size_t smem = lsigf_smem(a, a.gpw);
    if (smem > (size_t)kLdsBytes) return -2;
    const int grid = (a.B + a.gpw - 1) / a.gpw;
    hipError_t err;
    switch (rt) {
        case 1: err = launch_rt<1>(a, grid, smem, st); break;
        case 2: err = launch_rt<2>(a, grid, smem, st); break;
        case 3: err = launch_rt<3>(a, grid, smem, st); break;
        case 4: err = launch_rt<4>(a, grid, smem, st); break;
        case 5: err = launch_rt<5>(a, grid, smem, st); break;
        case 6: err = launch_rt<6>(a, grid, smem, st); break;
        case 7: err = launch_rt<7>(a, grid, smem, st); break;
        default: return -2;
    }
    return err == hipSuccess ? 0 : -3;
}

int filter_pack_launch(const float* h, float* packed, int G, int F, int K, int E, hipStream_t st) {
    const size_t total = filter_packed_floats(G, F, K, E);
    const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(pack_filter_kernel, dim3(grid), dim3(256), 0, st, h, packed, G, F, K, E);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int decode_actions_launch(const float* logits, int* actions, int B, int N, hipStream_t st) {
    const int total = B * N;
    hipLaunchKernelGGL(decode_actions_kernel, dim3((total + 255) / 256), dim3(256), 0, st, logits,
                       actions, B, N);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace gnnpp
