// Graph filter + ReLU + action head of the POLICY step for teams of 17 .. 100 agents (gfx950): the second
// kernel of gnnpp_policy_fwd when a graph no longer fits the fused one-launch kernel.
//
//   logits[n, b, :] = act_b + act_w . relu(bias + sum_k W_k . z_k[b, n, :]),   z_0 = x,  z_k = S^T-gather of z_{k-1}
//
// Same arithmetic as lsigf_kernel<.., H2 = true> with the fused head (graphML.py:2273-2367 BatchLSIGF,
// decentralplanner.py:301-315): exact fp32 shifts in ascending neighbour order, split-f16 contraction with the
// taps accumulated in the order k = 0, 1, .., the cross terms in their own accumulator.  What differs is the
// schedule.  lsigf_kernel is one kernel for every layout / precision / training dump the filter API has; one
// workgroup's time there (19.5 us at 256 graphs of 50 nodes, profiles/r02_phase_stamps.jsonl) is a chain of
// exposed latencies: a load -> store staging loop (one memory round trip per element: 4.5 us), the bias /
// act_w / act_b fetched where they are used (two more round trips: 3.9 us of epilogue), one conversion pass
// and one barrier per tap, a 32-MFMA dependent chain per row tile for the head, and register spills on the way
// (128 VGPRs for 16 waves).  Here, for the one shape the policy has (E = 1, G = F = 128, node-major x, one
// graph per workgroup):
//   * every global load of the kernel is issued in the first instructions -- x (at most four 16-byte loads per
//     thread), S (all of it in flight), ONE float per thread of act_w / bias / act_b / the split scale (parked in
//     LDS until the epilogue), then the first tap's A fragments: one memory round trip in total;
//   * z_0 goes to LDS twice while it is in registers, as fp32 (the first shift reads it) and as hi | lo halves:
//     tap 0 is contracted beside the list building and needs no conversion pass;
//   * a QUARTER wave gathers a node (8 features per lane, 4 neighbours per trip): 64 nodes in flight per pass,
//     one pass for 50 nodes;
//   * the last shift writes z_{K-1} directly as hi | lo halves: the last tap needs no conversion pass and no
//     barrier between it and the tap before;
//   * the head is applied to the accumulators where they are: after bias + ReLU a lane holds y[row, 4 features]
//     -- exactly the B operand of the 16x16x4 MFMA against act_w's columns of its own 16-feature tile -- so every
//     wave multiplies its tile (4 MFMAs per row tile) and the eight tiles' partial logits are summed through LDS
//     in the fixed order mt = 0..7 (deterministic; the rounding differs from the pair-chain head of lsigf_kernel
//     in the last bit, tests/test_gpu_parity.py compares the two).  No y tile in LDS, no 32-MFMA chain.
// MODE (the call's `precision`, include/gnnpp.h): 0 = the split-f16 schedule described above (GNNPP_PREC_SPLIT_F16);
// 2 = bf16x3, the fp32-equivalent default (GNNPP_PREC_FP32): z_k exists as fp32 rows (what the next shift reads) and,
// for the workgroup's own rows, as three bf16 planes in ONE extra buffer PB (row stride 800 B) written by the
// PRODUCER of z_k -- the staging loop for z_0, the shift's epilogue for z_k -- so there is no conversion pass at all;
// six MFMAs per (row tile, 32 channels); needs N * 800 more bytes of LDS, which exists up to about 64 nodes;
// 1 = exact fp32 MFMA (GNNPP_PREC_FP32_MFMA, and GNNPP_PREC_FP32 on graphs whose planes do not fit): the taps read
// the fp32 rows directly, 32 MFMAs of K = 4 per (row tile, 128 channels).
// nsplit workgroups per graph (as in lsigf_kernel; lsigf_plan: as many as keep every workgroup on a CU of its own, at
// most one per 16-row tile) when at most 128 large graphs would leave half of the CUs idle -- 16 graphs of 100 agents,
// the shard one GPU of eight holds of config 5, run as 112 workgroups of one row tile: all of them stage the graph and
// run the shifts k < K-1 on all rows; the last shift, the contraction and the head run on the workgroup's own row tiles.
#include "gnnpp_common.h"

namespace gnnpp {

constexpr int kPfZs = 136;                 // LDS row stride of a z buffer in floats (128 + 8, as lsigf_kernel).  Measured
                                           // (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per C3 launch, filter time): 136:
                                           // 0.60 M / 2.73 M, 14.8 us; 132: 1.31 M / 3.44 M, 15.4 us; 140: 1.21 M / 3.34 M,
                                           // 15.2 us; wider strides do not fit 100 nodes.
constexpr int kPfConsts = 776;             // act_w [5][128] | bias [128] | act_b [5] | 1 / split scale | pad
constexpr int kPfMaxNodes = 100;
constexpr int kPfMinNodes = 17;

// LDS bytes up to the constants (z buffers, S slab, neighbour lists, degrees)
__host__ __device__ inline size_t pf_lds_base(int N, int Ns, int K) {
    const size_t lists = K > 1 ? (size_t)N * Ns + ((N + 15) & ~15) : 0;
    return (2 * (size_t)N * kPfZs * 4 + (K > 1 ? (size_t)N * Ns * 4 : 0) + lists + 15) & ~(size_t)15;
}

// z_{k+1}[r, :] = sum over the neighbours m of node r (ascending) of S[m, r] * z_k[m, :], rows [row_lo, row_hi):
// a quarter wave per row; lane ql holds features [4 ql, 4 ql + 4) and [64 + 4 ql, 64 + 4 ql + 4).
// A shift reads 512 bytes of LDS per edge, and a launch waits for its slowest workgroup: the time follows the
// DENSEST graph of the batch (geometric graphs of 100 agents: mean degree 8, but a crowded corner is a near-clique
// of 30 .. 50 nodes; per-workgroup stamps at C5: median 16.4 us, slowest 19.9 / 25.9 us for a batch whose densest
// graph has a node of degree 30 / 52, correlation of a workgroup's time with its graph's largest degree 0.9).
// Measured and dropped: fetching the next trip's indices / weights one trip ahead (no change: the compiler's
// schedule already overlaps them), and a whole-wave path for nodes of more than 16 neighbours (two features per
// lane, 16 neighbours per trip: C3 14.4 -> 18.6 us, C5 24.0 -> 32.5 us on the same inputs -- the work is edges,
// not a long loop on one node).
// znxt: fp32 result rows (may be null); zsplit: the same rows as f16 hi | lo halves, the layout split_rows
// produces (may be null) -- the last shift writes only those.  (Writing both in a middle shift, into a third
// buffer, to drop the conversion pass of the tap before the last was measured at N = 50: no gain -- the extra
// stores cost the shift what the pass saved.)
constexpr int kPfPRow = 3 * 256 + 32;     // MODE 2: row stride of the plane buffer PB (bytes)

template <int MODE>
__device__ __forceinline__ void pf_gather(const float* __restrict__ Sl, const unsigned char* __restrict__ idx,
                                          const unsigned char* __restrict__ cnt, const float* __restrict__ zprev,
                                          float* __restrict__ znxt, float* __restrict__ zsplit, int Ns, int row_lo,
                                          int row_hi, int wave, int lane, unsigned long long& bad,
                                          char* __restrict__ planes = nullptr, int own_lo = 0, int own_hi = 0) {
    // MODE 2: `planes` = PB; rows [own_lo, own_hi) also leave as bf16x3 planes (PB row r - own_lo); zsplit unused
    typedef _Float16 v4h __attribute__((ext_vector_type(4)));
    const int quarter = lane >> 4, ql = lane & 15;
    for (int rb = row_lo + 4 * wave; rb < row_hi; rb += 64) {           // wave-uniform trip count
        const int r = rb + quarter;
        const bool rv = r < row_hi;
        const int rr = rv ? r : rb;
        const int lo = rr * Ns;
        const float* wl = Sl + lo;                                       // compacted weights of node rr
        const unsigned char* il = idx + lo;
        const int deg = rv ? (int)cnt[rr] : 0;
        const float* zc = zprev + 4 * ql;
        v4f acc0 = vzero(), acc1 = vzero();
        for (int d = 0; __ballot(d < deg) != 0ull; d += 4) {             // until all four rows are done
            // entries past the degree: weight 0 and a stale (but valid) row index.  (Masking those reads per
            // lane was measured: the branches cost more than the LDS cycles they save, 1.72 -> 1.94 us per shift.)
            const unsigned pk = *reinterpret_cast<const unsigned*>(il + d);
            const v4f w = *reinterpret_cast<const v4f*>(wl + d);
            const float* zr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) zr[u] = zc + __umul24((pk >> (8 * u)) & 255u, (unsigned)kPfZs);
            v4f za[4], zb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) za[u] = *reinterpret_cast<const v4f*>(zr[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) zb[u] = *reinterpret_cast<const v4f*>(zr[u] + 64);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc0[c] = fmaf(w[u], za[u][c], acc0[c]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc1[c] = fmaf(w[u], zb[u][c], acc1[c]);
        }
        if (znxt && rv) {
            float* row = znxt + rr * kPfZs;
            *reinterpret_cast<v4f*>(row + 4 * ql) = acc0;
            *reinterpret_cast<v4f*>(row + 64 + 4 * ql) = acc1;
        }
        if (MODE == 2 && planes) {
            v2f p0[3], p1[3];
            b3_split4(acc0, p0);
            b3_split4(acc1, p1);
            if (rv && rr >= own_lo && rr < own_hi) {
                char* row = planes + (rr - own_lo) * kPfPRow;
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    *reinterpret_cast<v2f*>(row + p * 256 + 8 * ql) = p0[p];           // features 4 ql ..
                    *reinterpret_cast<v2f*>(row + p * 256 + 128 + 8 * ql) = p1[p];     // features 64 + 4 ql ..
                }
            }
        }
        if (MODE == 0 && zsplit) {
            float* row = zsplit + rr * kPfZs;
            // |z| >= 65504 does not fit the hi half: range guard as in split_rows
            float mx = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) mx = fmaxf(mx, fmaxf(fabsf(acc0[c]), fabsf(acc1[c])));
            bad |= __ballot(rv && mx >= 65504.f);
            v4h h0, l0, h1, l1;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                h0[c] = (_Float16)acc0[c]; l0[c] = (_Float16)(acc0[c] - (float)h0[c]);
                h1[c] = (_Float16)acc1[c]; l1[c] = (_Float16)(acc1[c] - (float)h1[c]);
            }
            if (rv) {
                *reinterpret_cast<v2f*>(row + 2 * ql) = __builtin_bit_cast(v2f, h0);          // features 4 ql ..
                *reinterpret_cast<v2f*>(row + 32 + 2 * ql) = __builtin_bit_cast(v2f, h1);     // features 64 + 4 ql ..
                *reinterpret_cast<v2f*>(row + 64 + 2 * ql) = __builtin_bit_cast(v2f, l0);
                *reinterpret_cast<v2f*>(row + 96 + 2 * ql) = __builtin_bit_cast(v2f, l1);
            }
        }
    }
}

// RTW = 16-row MFMA tiles per wave (waves 0..7 own the first RTW tiles of the workgroup's range, waves 8..15 the
// next RTW; wave & 7 is the 16-feature output tile).
template <int RTW, int MODE>
__global__ __launch_bounds__(1024) void policy_filter_kernel(const LsigfArgs p) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    constexpr int NT = 1024, NW = 16;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int a = lane & 15;      // row inside a 16-row tile (MFMA j)
    const int q = lane >> 4;      // MFMA k slot
    const int N = p.N, Ns = p.Ns, K = p.K;

    int gblk = blockIdx.x, part = 0;
    if (p.nsplit > 1) {                                // a group of 8 nsplit blocks = 8 graphs; blocks b, b + 8, b + 16, ..
        const int grp = gblk / (8 * p.nsplit), in = gblk - grp * 8 * p.nsplit;    // of the group: one graph (same XCD)
        part = in >> 3;
        gblk = grp * 8 + (in & 7);
    }
    if (gblk >= p.B) return;                           // padding block of a split grid (uniform)
    const int mt = wave & 7;
    const int rt_all = (N + 15) >> 4;
    const int tile_lo = part * rt_all / p.nsplit;      // (nsplit <= rt_all: every part owns at least one row tile)
    const int tile_hi = (part + 1) * rt_all / p.nsplit;
    const int row_lo = tile_lo * 16, row_hi = min(tile_hi * 16, N);
    const int rt0 = tile_lo + (wave >> 3) * RTW;       // first row tile of this wave
    const bool has_mfma = rt0 < tile_hi;

    constexpr bool B3 = MODE == 2;                      // bf16x3 planes in PB
    float* zbuf0 = reinterpret_cast<float*>(gnnpp_smem);
    float* zbuf1 = zbuf0 + N * kPfZs;
    // the dense slab [N][Ns] (row n = column n of the GSO, compacted in place into its list) behind the z buffers
    float* const Sl = zbuf1 + N * kPfZs;
    unsigned char* const idx = reinterpret_cast<unsigned char*>(Sl + (K > 1 ? N * Ns : 0));
    unsigned char* const cnt = idx + (K > 1 ? N * Ns : 0);
    float* cb = reinterpret_cast<float*>(gnnpp_smem + p.pf_const_off);
    float* const part_sums = reinterpret_cast<float*>(gnnpp_smem + p.pf_part_off);      // [8][N][8]
    // MODE 2: bf16x3 planes of this workgroup's own rows, one tap at a time.  Two layouts (policy_filter_dispatch):
    //   separate (pf_plane_off >= 0): a buffer of its own behind the partial logits -- teams up to ~64 agents;
    //   aliased  (pf_plane_off < 0, r06): the planes of tap k live in whichever z buffer is DEAD while tap k is
    //   contracted -- teams of 65 .. 100 agents, whose two fp32 z buffers + S slab + lists fill the LDS (r05 ran
    //   those on the exact fp32 MFMA: 2.7x the matrix-pipe time, 9 of the 24 us of a 128 x 100 launch):
    //     tap 0        -> z buffer 1 (written by the staging loop; shift 1 overwrites it after the barrier)
    //     tap k < K-1  -> z buffer (k-1) & 1 = the SOURCE of shift k, dead behind the shift's barrier: a conversion
    //                     pass over the workgroup's own rows of z_k (0.3 us) fills it
    //     tap K-1      -> z buffer (K-1) & 1 = where the last shift would have put fp32 rows: it writes planes instead
    //   needs own rows x 800 B <= N x 544 B: any split of the graph over >= 2 workgroups.
    const bool pb_alias = p.pf_plane_off < 0;
    auto plane_buf = [&](int k) -> char* {
        if (!pb_alias) return gnnpp_smem + p.pf_plane_off;
        const int b = (k == 0) ? 1 : (k + 1 == K) ? (k & 1) : ((k - 1) & 1);
        return reinterpret_cast<char*>(b ? zbuf1 : zbuf0);
    };
    char* PB = plane_buf(0);

    GNNPP_STAMP(blockIdx.x, 0, tid == 0);
    // ---- every global load of the kernel, issued now -------------------------------------------------------
    // x: [N][128] contiguous, 32 N 16-byte pieces
    const v4f* xs = reinterpret_cast<const v4f*>(p.x + (size_t)gblk * N * 128);
    v4f xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = tid + u * NT;
        if (i < N * 32) xv[u] = xs[i];
    }
    // S: dense [N][N] slab, fp32 (16 bytes at a time when the slabs allow it) or fp64
    const int NN = N * N;
    const size_t sidx = (size_t)gblk * NN;
    v4f s4[3];
    float s1[10];
    double sd[10];
    if (K > 1) {
        if (p.s_is_f64) {
            const double* src = reinterpret_cast<const double*>(p.S) + sidx;
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const int i = tid + u * NT;
                if (i < NN) sd[u] = src[i];
            }
        } else if (p.s_vec4) {
            const v4f* src = reinterpret_cast<const v4f*>(reinterpret_cast<const float*>(p.S) + sidx);
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int i = tid + u * NT;
                if (i < (NN >> 2)) s4[u] = src[i];
            }
        } else {
            const float* src = reinterpret_cast<const float*>(p.S) + sidx;
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const int i = tid + u * NT;
                if (i < NN) s1[u] = src[i];
            }
        }
    }
    // constants of the epilogue: one float per thread
    float cpre = 0.f;
    if (tid < 640) cpre = p.act_w[tid];
    else if (tid < 768) cpre = p.bias ? p.bias[tid - 640] : 0.f;
    else if (tid < 773) cpre = p.act_b[tid - 768];
    else if (tid == 773) cpre = MODE == 0 ? p.wpk_h[filter_packed_h2_floats(128, 128, K, 1) + 1] : 1.f;
    // A fragments of the first tap.  MODE 0: packed block (k, mt, gg) = 64 lanes x 16 bytes = hi | lo of four
    // k-steps; MODE 1: the fp32 fragments of four k-steps; MODE 2: block (k, mt, kb) = three 16-byte planes
    constexpr int NA = B3 ? 12 : 8;
    constexpr size_t tap_stride = (size_t)8 * NA * 256;
    v4f Acur[NA];
    auto load_tap = [&](v4f (&A)[NA], int tap) {
        if (has_mfma) {
            const float* wt = (MODE == 0 ? p.wpk_h : MODE == 1 ? p.wpk : p.wpk_b) + tap * tap_stride +
                              ((size_t)mt * NA * 64 + lane) * 4;
#pragma unroll
            for (int gg = 0; gg < NA; ++gg) A[gg] = *reinterpret_cast<const v4f*>(wt + gg * 256);
        }
    };

    // ---- LDS: zero the index lists, then the staged data ------------------------------------------------
    if (K > 1) {                                       // stale list entries must be valid rows
        unsigned* iz = reinterpret_cast<unsigned*>(idx);
        for (int i = tid; i < (N * Ns) >> 2; i += NT) iz[i] = 0u;
    }
    // z_0 twice: fp32 in the first buffer (the first shift reads it) and as f16 hi | lo halves in the second
    // (the first tap's MFMA operand, layout of split_rows) -- tap 0 then needs no conversion pass of its own
    unsigned long long bad = 0;                        // lanes that handed |z| >= 65504 to the f16 pipe
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        typedef _Float16 v4h __attribute__((ext_vector_type(4)));
        const int i = tid + u * NT;
        const bool ok = i < N * 32;
        const v4f v = ok ? xv[u] : vzero();
        const int r = i >> 5, c4 = i & 31;
        if (MODE == 0) {
            bad |= __ballot(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))) >= 65504.f);
            v4h hh, ll;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                hh[c] = (_Float16)v[c];
                ll[c] = (_Float16)(v[c] - (float)hh[c]);
            }
            if (ok) {
                *reinterpret_cast<v2f*>(zbuf1 + r * kPfZs + 2 * c4) = __builtin_bit_cast(v2f, hh);
                *reinterpret_cast<v2f*>(zbuf1 + r * kPfZs + 64 + 2 * c4) = __builtin_bit_cast(v2f, ll);
            }
        }
        if (B3) {
            v2f pl[3];
            b3_split4(v, pl);
            if (ok && r >= row_lo && r < row_hi) {
#pragma unroll
                for (int pp = 0; pp < 3; ++pp)
                    *reinterpret_cast<v2f*>(PB + (r - row_lo) * kPfPRow + pp * 256 + 8 * c4) = pl[pp];
            }
        }
        if (ok) *reinterpret_cast<v4f*>(zbuf0 + r * kPfZs + 4 * c4) = v;
    }
    if (K > 1) {
        // element e = m * N + n of the slab goes to Sl[n][m]   (m = e / N exactly: (e + 0.5) / N is at least
        // 0.5 / N away from an integer, the float error is < 1e-3 of that for e < 2^14)
        const float inv_n = 1.0f / (float)N;
        if (p.s_is_f64) {
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const int e = tid + u * NT;
                if (e < NN) {
                    const int m = (int)(((float)e + 0.5f) * inv_n), n = e - m * N;
                    Sl[n * Ns + m] = (float)sd[u];
                }
            }
        } else if (p.s_vec4) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int e = 4 * (tid + u * NT);
                if (e < NN) {
                    int m = (int)(((float)e + 0.5f) * inv_n), n = e - m * N;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        Sl[n * Ns + m] = s4[u][c];
                        if (++n == N) { n = 0; ++m; }
                    }
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const int e = tid + u * NT;
                if (e < NN) {
                    const int m = (int)(((float)e + 0.5f) * inv_n), n = e - m * N;
                    Sl[n * Ns + m] = s1[u];
                }
            }
        }
    }
    if (tid < 774) cb[tid] = cpre;
    // The first tap's fragments (8 KB per wave, 128 KB per workgroup) are requested only now, behind the staged
    // data: ahead of it they would queue in front of the later waves' x / S loads.  They arrive during the
    // barrier and the list building.
    load_tap(Acur, 0);
    __syncthreads();                                   // z_0 (both forms), S, constants visible
    GNNPP_STAMP(blockIdx.x, 1, tid == 0);

    v4f acc[RTW], acc2[RTW];                           // hi.hi products | cross terms (MODE 1: acc only)
#pragma unroll
    for (int t = 0; t < RTW; ++t) { acc[t] = vzero(); acc2[t] = vzero(); }
    auto brow = [&](int t) { return min((rt0 + t) * 16 + a, N - 1); };     // (rows >= N: copies, never stored)
    // contraction of one tap: D[f, row] += W_k[f, g] z_k[row, g]; `zsrc`: MODE 0 z_k as hi | lo halves, MODE 1 the
    // fp32 rows, MODE 2 ignored (the planes are in PB)
    auto contract = [&](const float* zsrc, const v4f (&A)[NA]) {
        if (!has_mfma) return;
        if (MODE == 0) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const v8h Ah = __builtin_bit_cast(v8h, A[2 * kb]);
                const v8h Al = __builtin_bit_cast(v8h, A[2 * kb + 1]);
#pragma unroll
                for (int t = 0; t < RTW; ++t) {
                    const float* zrow = zsrc + brow(t) * kPfZs + q * 4;
                    const v8h Bh = __builtin_bit_cast(v8h, *reinterpret_cast<const v4f*>(zrow + kb * 16));
                    const v8h Bl = __builtin_bit_cast(v8h, *reinterpret_cast<const v4f*>(zrow + 64 + kb * 16));
                    acc2[t] = mfma16h(Ah, Bl, acc2[t]);
                    acc[t] = mfma16h(Ah, Bh, acc[t]);
                    acc2[t] = mfma16h(Al, Bh, acc2[t]);
                }
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int gg = 0; gg < 8; ++gg) {
                v4f Bf[RTW];
#pragma unroll
                for (int t = 0; t < RTW; ++t)
                    Bf[t] = *reinterpret_cast<const v4f*>(zsrc + brow(t) * kPfZs + q * 4 + gg * 16);
#pragma unroll
                for (int st = 0; st < 4; ++st)
#pragma unroll
                    for (int t = 0; t < RTW; ++t) acc[t] = mfma16(A[gg % NA][st], Bf[t][st], acc[t]);
            }
        } else {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
                for (int t = 0; t < RTW; ++t) {
                    // PB row of this lane's node, clamped into the workgroup's own rows (a D column depends only
                    // on its own B column; columns outside are never stored)
                    const int pr = min((rt0 + t) * 16 + a, row_hi - 1) - row_lo;
                    const char* zrow = PB + pr * kPfPRow + kb * 64 + q * 16;
                    v8b Bp[3];
#pragma unroll
                    for (int pp = 0; pp < 3; ++pp) Bp[pp] = as_b8(*reinterpret_cast<const v4f*>(zrow + pp * 256));
#pragma unroll
                    for (int term = 0; term < kB3Terms; ++term) {
                        const v8b Ap = as_b8(A[(3 * kb + b3_term_a(term)) % NA]);
                        if (term == kB3Terms - 1) acc[t] = mfma16b(Ap, Bp[0], acc[t]);
                        else acc2[t] = mfma16b(Ap, Bp[b3_term_b(term)], acc2[t]);
                    }
                }
            }
        }
    };

    // Schedule (taps accumulate in the order 0, 1, ..):
    //   lists || tap 0 (from the staged hi | lo z_0: the list building is ballot / latency work, the tap is
    //   LDS-bandwidth work -- different waves are in different phases and fill each other's gaps)
    //   shift k = 1 .. K-1: z_{k-1} (fp32, buffer (k-1) & 1) -> z_k (buffer k & 1; the LAST shift writes hi | lo);
    //   once shift k has read the fp32 z_{k-1}, that is converted in place and contracted (k >= 2);
    //   the last tap follows the one before without a barrier.
    if (K > 1) build_lists(p, Sl, idx, cnt, N, wave, NW, lane);
    GNNPP_STAMP(blockIdx.x, 2, tid == 0);
    contract(MODE == 0 ? zbuf1 : zbuf0, Acur);
    if (K > 1) load_tap(Acur, 1);                      // in flight during the shift(s)
    GNNPP_STAMP(blockIdx.x, 3, tid == 0);
    if (MODE != 0) {
        // fp32 rows ping-pong; tap k is contracted right behind shift k.  MODE 2: the shift's epilogue writes the
        // planes of the workgroup's own rows into PB (which tap k-1 has finished reading: first barrier).
        for (int k = 1; k < K; ++k) {
            const float* zsrc = ((k - 1) & 1) ? zbuf1 : zbuf0;
            float* zdst = (k & 1) ? zbuf1 : zbuf0;
            const bool last = k + 1 == K;
            if (B3 || k == 1) __syncthreads();         // lists visible / PB free (MODE 1, k >= 2: zdst was z_{k-2},
                                                       // whose tap finished before the previous shift's barrier)
            GNNPP_STAMP(blockIdx.x, 4, tid == 0 && k == 1);
            GNNPP_STAMP(blockIdx.x, 7, tid == 0 && k == 2);
            if (B3) PB = plane_buf(k);
            pf_gather<MODE>(Sl, idx, cnt, zsrc, (B3 && last) ? nullptr : zdst, nullptr, Ns, last ? row_lo : 0,
                            last ? row_hi : N, wave, lane, bad, (B3 && pb_alias && !last) ? nullptr : PB, row_lo, row_hi);
            GNNPP_STAMP(blockIdx.x, 8, tid == 0 && k == 2);
            __syncthreads();                           // z_k visible
            if (B3 && pb_alias && !last) {
                // aliased planes: z_{k-1} (zsrc, = PB) is dead now -- the own rows of z_k (fp32, zdst) as planes into it
                for (int i = tid; i < (row_hi - row_lo) * 32; i += NT) {
                    const int r = i >> 5, c4 = i & 31;
                    v2f pl[3];
                    b3_split4(*reinterpret_cast<const v4f*>(zdst + (row_lo + r) * kPfZs + 4 * c4), pl);
#pragma unroll
                    for (int pp = 0; pp < 3; ++pp)
                        *reinterpret_cast<v2f*>(PB + r * kPfPRow + pp * 256 + 8 * c4) = pl[pp];
                }
                __syncthreads();
            }
            GNNPP_STAMP(blockIdx.x, 5, tid == 0 && k == 1);
            GNNPP_STAMP(blockIdx.x, 9, tid == 0 && k == 2);
            contract(zdst, Acur);
            GNNPP_STAMP(blockIdx.x, 6, tid == 0 && k == 1);
            if (k + 1 < K) load_tap(Acur, k + 1);
        }
    } else {
    for (int k = 1; k + 1 < K; ++k) {                  // the shifts before the last one: all rows, fp32 out
        float* zsrc = (k & 1) ? zbuf0 : zbuf1;         // z_{k-1}, fp32
        float* zdst = (k & 1) ? zbuf1 : zbuf0;
        __syncthreads();                               // lists visible / tap k-2 is done with zdst
        GNNPP_STAMP(blockIdx.x, 4, tid == 0 && k == 1);
        pf_gather<0>(Sl, idx, cnt, zsrc, zdst, nullptr, Ns, 0, N, wave, lane, bad);
        GNNPP_STAMP(blockIdx.x, 5, tid == 0 && k == 1);
        if (k >= 2) {
            __syncthreads();                           // every reader of the fp32 z_{k-1} is done
            split_rows(zsrc, row_lo, row_hi, kPfZs, wave, NW, lane, bad);
            __syncthreads();
            contract(zsrc, Acur);                      // tap k-1
            load_tap(Acur, k);
        }
    }
    if (K > 1) {                                       // the last shift: own rows only, written as hi | lo
        const int k = K - 1;
        float* zsrc = (k & 1) ? zbuf0 : zbuf1;
        float* zdst = (k & 1) ? zbuf1 : zbuf0;
        __syncthreads();
        GNNPP_STAMP(blockIdx.x, k == 1 ? 4 : 6, tid == 0);
        pf_gather<0>(Sl, idx, cnt, zsrc, nullptr, zdst, Ns, row_lo, row_hi, wave, lane, bad);
        __syncthreads();                               // z_{K-1} visible; every reader of the fp32 z_{K-2} is done
        GNNPP_STAMP(blockIdx.x, 7, tid == 0);
        if (k >= 2) {
            // (A second register set for the last tap's fragments, requested before this conversion pass, was
            // measured: the conversion pass took 1.2 instead of 0.5 us and the kernel 16.1 instead of 14.9 us.)
            split_rows(zsrc, row_lo, row_hi, kPfZs, wave, NW, lane, bad);
            __syncthreads();
            GNNPP_STAMP(blockIdx.x, 8, tid == 0);
            contract(zsrc, Acur);                      // tap K-2
            load_tap(Acur, k);
            GNNPP_STAMP(blockIdx.x, 9, tid == 0);
            contract(zdst, Acur);                      // tap K-1, no barrier in between
        } else {
            contract(zdst, Acur);                      // K == 2: tap 1 (tap 0 ran beside the list building)
        }
    }
    }
    if (MODE == 0 && p.range_flag && bad) *p.range_flag = 1;

    // ---- epilogue: bias + ReLU in registers, this wave's 16 features of the head, partial logits to LDS ----
    GNNPP_STAMP(blockIdx.x, 12, tid == 0);
    if (has_mfma) {
        const int f0 = mt * 16 + q * 4;
        const float h2_inv = cb[773];
        const v4f bv = *reinterpret_cast<const v4f*>(cb + 640 + f0);
        v4f A5 = vzero();                              // act_w[a5 = a][f0 .. f0 + 4), rows a5 >= 5 zero
        if (a < 5) A5 = *reinterpret_cast<const v4f*>(cb + a * 128 + f0);
#pragma unroll
        for (int t = 0; t < RTW; ++t) {
            const int row = (rt0 + t) * 16 + a;
            v4f v = MODE == 0 ? (acc[t] + acc2[t]) * h2_inv + bv : MODE == 1 ? acc[t] + bv : (acc[t] + acc2[t]) + bv;
            if (p.relu) v = vrelu(v);
            const v4f d = mfma16x4(A5, v, vzero());    // d[r] = logit part a5 = 4 q + r of this lane's row
            if (rt0 + t < tile_hi && row < N) {
                float* ps = part_sums + (mt * N + row) * 8;
                if (q == 0) *reinterpret_cast<v4f*>(ps) = d;
                else if (q == 1) ps[4] = d[0];
            }
        }
    }
    __syncthreads();
    GNNPP_STAMP(blockIdx.x, 13, tid == 0);
    const int nout = (row_hi - row_lo) * 5;
    for (int i = tid; i < nout; i += NT) {
        const int row = row_lo + i / 5, a5 = i % 5;
        float s = part_sums[row * 8 + a5];
#pragma unroll
        for (int m = 1; m < 8; ++m) s += part_sums[(m * N + row) * 8 + a5];
        p.logits[((size_t)row * p.B + gblk) * 5 + a5] = s + cb[768 + a5];
    }
    GNNPP_STAMP(blockIdx.x, 14, tid == 0);
}

std::atomic<int> g_filter_policy_kernel{1};            // GNNPP_TUNE_FILTER_POLICY_KERNEL: 0 = lsigf_kernel everywhere
std::atomic<int> g_filter_plane_alias{1};              // GNNPP_TUNE_FILTER_PLANE_ALIAS: 0 = never alias planes onto z buffers

// Does the planned filter launch have the policy step's shape?  (a is complete: lsigf_plan ran.)
static bool policy_filter_applies(const LsigfArgs& a) {
    return g_filter_policy_kernel.load(std::memory_order_relaxed) && a.act_w && a.act_b && a.logits && !a.y &&
           !a.zs && a.E == 1 && a.G == 128 && a.F == 128 && a.F_all == 128 && a.x_node_major && a.Nin == a.N &&
           !a.bias_per_node && !a.s_transposed && a.s_batched && a.gpw == 1 &&
           g_filter_waves.load(std::memory_order_relaxed) != 8 &&
           a.N >= kPfMinNodes && a.N <= kPfMaxNodes && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0
#ifdef GNNPP_MEASURE
           && a.ablate == 0
#endif
        ;
}

template <int RTW, int MODE>
static hipError_t policy_filter_launch_one(const LsigfArgs& a, int grid, size_t smem, hipStream_t st) {
    static LdsAttrOnce once;
    set_lds_attr_once(once, reinterpret_cast<const void*>(&policy_filter_kernel<RTW, MODE>), kLdsBytes);
    hipLaunchKernelGGL((policy_filter_kernel<RTW, MODE>), dim3(grid), dim3(1024), smem, st, a);
    return hipGetLastError();
}

template <int MODE>
static hipError_t policy_filter_launch_rtw(int rtw, const LsigfArgs& a, int grid, size_t smem, hipStream_t st) {
    switch (rtw) {
        case 1: return policy_filter_launch_one<1, MODE>(a, grid, smem, st);
        case 2: return policy_filter_launch_one<2, MODE>(a, grid, smem, st);
        case 3: return policy_filter_launch_one<3, MODE>(a, grid, smem, st);
        case 4: return policy_filter_launch_one<4, MODE>(a, grid, smem, st);
        default: return hipErrorInvalidValue;
    }
}

// What the policy filter would run for a planned call: mode 0 split-f16 | 1 exact fp32 MFMA | 2 bf16x3 planes in a buffer
// of their own | 3 bf16x3 planes aliased onto the dead z buffer (kernel MODE 2, pf_plane_off < 0); fills the LDS offsets
// of `a`.  Returns false when the shape is not the policy's (the caller runs lsigf_kernel).
struct PfLaunch { int mode, rtw; size_t smem; };
static bool policy_filter_plan(LsigfArgs& a, PfLaunch& L) {
    if (!policy_filter_applies(a)) return false;
    const int tiles = (a.rt_total + a.nsplit - 1) / a.nsplit;           // row tiles of the largest part
    if ((tiles + 1) / 2 > 4) return false;
    const size_t base = pf_lds_base(a.N, a.Ns, a.K) + kPfConsts * 4;
    const size_t parts = (size_t)8 * a.N * 8 * 4;
    // GNNPP_PREC_FP32: bf16x3 planes of the workgroup's own rows -- in a buffer of their own while the LDS has room
    // (teams up to ~64 agents), else ALIASED onto whichever z buffer is dead while a tap is contracted (r06: possible
    // as soon as the graph is split over >= 2 workgroups: own rows x 800 B <= N x 544 B), else the exact fp32 MFMA
    // (same accuracy class, 2.7x the matrix-pipe time; a whole team of 65 .. 100 agents in ONE workgroup).
    // (r04 built planes beside COMPACT neighbour lists for those teams -- "MODE 3" -- and measured it no faster than
    // the exact fp32 MFMA: what its contraction saved its plane production spent; removed in r05.)
    int mode = a.prec == kPrecSplitF16 ? 0 : a.prec == kPrecFp32Mfma ? 1 : 2;
    const size_t planes = (size_t)tiles * 16 * kPfPRow;
    a.pf_const_off = (int)pf_lds_base(a.N, a.Ns, a.K);
    a.pf_part_off = (int)base;
    a.pf_plane_off = (int)(base + parts);
    if (mode == 2 && base + parts + planes > (size_t)kLdsBytes) {
        const bool alias_ok = g_filter_plane_alias.load(std::memory_order_relaxed) &&
                              planes <= (size_t)a.N * kPfZs * 4;
        mode = alias_ok ? 3 : 1;
        if (alias_ok) a.pf_plane_off = -1;
    }
    size_t smem = base + parts + (mode == 2 ? planes : 0);
    if (smem > (size_t)kLdsBytes) {
        // large graphs: the partial logits reuse the S slab (dead after the last shift)
        if (a.K < 2 || (size_t)a.N * a.Ns * 4 < parts || base > (size_t)kLdsBytes) return false;
        smem = base;
        a.pf_part_off = 2 * a.N * kPfZs * 4;
    }
    L.mode = mode; L.rtw = (tiles + 1) / 2; L.smem = smem;
    return true;
}

// Launches the policy filter for a planned call; returns 1 when the shape is not the policy's (the caller runs
// lsigf_kernel), 0 on success, -3 on a launch error.
static int policy_filter_dispatch(LsigfArgs a, const LsigfPlan& plan, hipStream_t st) {
    PfLaunch L;
    if (!policy_filter_plan(a, L)) return 1;
    const hipError_t err = L.mode == 0 ? policy_filter_launch_rtw<0>(L.rtw, a, plan.grid, L.smem, st)
                         : L.mode == 1 ? policy_filter_launch_rtw<1>(L.rtw, a, plan.grid, L.smem, st)
                                       : policy_filter_launch_rtw<2>(L.rtw, a, plan.grid, L.smem, st);
    return err == hipSuccess ? 0 : -3;
}

}  // namespace gnnpp
