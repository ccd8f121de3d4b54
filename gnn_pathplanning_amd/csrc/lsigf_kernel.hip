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
//   2. the dense S slabs are staged into LDS (fp64 -> fp32 on the fly, like `S.float()`), and every
//      column n is compacted ONCE into its neighbour list: a half-wavefront scans the column 32
//      candidates at a time, a ballot + prefix popcount gives each non-zero its slot, and the row
//      indices m (ascending) go to a byte array idx[n][.] with the degree in cnt[n] (exact: structural
//      zeros contribute nothing).  All K-1 shifts reuse the lists;
//   3. shift k: one half-wavefront per node (two nodes in flight per wave).  Four neighbours per trip:
//      one 4-byte read of the index list, the four weights S[m,n] and four 512-byte feature rows from
//      LDS (ds_read_b128 per lane), an fmaf chain in ascending m with 4 features per lane -- exact
//      fp32, no ballots or bit scans left in the loop.  z ping-pongs between two LDS buffers;
//   4. contraction of tap k right after its shift: D[f, row] += W_k[f, g] z_k[row, g] on the MFMA with
//      the accumulators living in registers across all taps; W_k fragments stream from L2 in a
//      pre-packed order (one 16-byte load per lane per four MFMAs);
//   5. epilogue: + bias, optional ReLU, staged through LDS so the store is coalesced in either
//      output layout; optionally the 128 -> 5 action head (decentralplanner.py:304-315) is fused.
//
// Large graphs on an under-filled chip (one graph per workgroup and at most 128 workgroups, e.g. 128
// graphs of 100 agents): TWO workgroups per graph.  Both stage the whole graph and run the shifts
// k < K-1 on all rows, but each runs the last shift, the contraction and the epilogue on its own half
// of the 16-row tiles only; blocks b and b + 8 (same XCD under round-robin dispatch) share a graph so
// the second reader of x / S hits that XCD's L2.
//
// HBM traffic per launch is the algorithmic minimum: x, S and y once (+ the packed taps, L2 hits).
#include "gnnpp_common.h"

namespace gnnpp {

struct LsigfArgs {
    const float* x;
    const void* S;
    const float* wpk;      // packed taps, see pack_filter_kernel
    const float* wpk_h;    // split-f16 fragments + {2^k, 2^-k} (inside the same packed buffer)
    const float* wpk_b;    // bf16x3 fragments (inside the same packed buffer)
    int prec;              // GNNPP_PREC_*: arithmetic of the tap contraction
    const float* bias;     // [F] or nullptr
    float* y;              // may be nullptr when only the action head is wanted
    const float* act_w;    // [5,F] or nullptr: fused action head
    const float* act_b;    // [5]
    float* logits;         // [N,B,5]
    int B, N, Nin, G, F, K, E;
    int NG, MT;            // ceil(G/16), ceil(F/16)   (F <= 128 per launch -> MT <= 8)
    // A filter wider than 128 output features runs as several launches over output-feature chunks:
    // this launch computes features [f0, f0 + F) of F_all (each chunk recomputes the cheap shifts).
    // The packed taps, the bias and y are the FULL tensors; mt0 = f0 / 16 is the chunk's first tile.
    int F_all, f0, mt0, MT_all;
    int zstride;           // LDS row stride in floats = 16*max(NG,MT) + 8
    int gpw;               // graphs per workgroup
    int rt_total;          // 16-row MFMA tiles of a workgroup's graphs = ceil(gpw*N / 16)
    int Ns;                // LDS row stride of an S slab in floats = N rounded up to a multiple of 4
    int Nl;                // bytes per neighbour-index list (= Ns)
    int nsplit;            // workgroups per graph: 1, or 2 .. rt_total sharing its row tiles (gpw == 1 only)
    int s_vec4;            // fp32 S slabs are 16-byte aligned multiples of four floats: v4f staging loads
    int s_is_f64, s_batched, x_node_major, y_node_major, relu;
    int bias_per_node;     // bias is [F_all, N] (one value per feature AND node, graphML.py:2300-2302)
    int s_transposed;      // use S^T: turns the kernel into the input-gradient of the filter
    const float* y_mask;   // optional, y's node-major layout: y is stored as 0 where y_mask <= 0 (the input-gradient
                           // launch of the training step: the ReLU backward of the layer below, gnnpp_lsigf_input_grad)
    float* zs;             // optional [E*K][B*N][G] node-major dump of every tap signal z_{e,k}
    int* range_flag;       // optional device int: set to 1 when the split-f16 contraction saw |z| >= 65504
    int pf_part_off;       // policy_filter_kernel.hip: LDS byte offset of the partial logits
    int pf_plane_off;      // policy_filter_kernel.hip, bf16x3 mode: LDS byte offset of the plane buffer
    int pf_const_off;      // policy_filter_kernel.hip: LDS byte offset of the epilogue constants
    int ablate;            // GNNPP_MEASURE builds only (tools/ab_bench.py): bit 0 skip the shifts, bit 1
                           // skip the MFMA contraction, bit 2 skip staging of S, bit 3 skip the epilogue
};

// Re-order h[F,E,K,G] into MFMA A fragments: block (e,k,mt,gg) holds, for lane l = q*16 + i and
// k-step s, h[f = mt*16 + i][e][k][g = gg*16 + q*4 + s]  (0 outside F x G).
// 2^k with max|h| * 2^k in [512, 1024) (as the encoder: the lo halves of the weights stay normal
// f16 numbers); one block.  packed_h points at the split-f16 region, its last 4 floats hold 2^k, 2^-k.
__global__ void filter_scale_kernel(const float* __restrict__ h, float* __restrict__ scale_out,
                                    size_t n) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    float* red = reinterpret_cast<float*>(gnnpp_smem);
    // 16-byte loads, four of them in flight per thread before the first maximum (max is order-free:
    // exact whatever the association); cudaMalloc-style base alignment makes h 16-byte aligned, checked
    float m = 0.f;
    size_t done = 0;
    if ((reinterpret_cast<size_t>(h) & 15) == 0) {
        const v4f* h4 = reinterpret_cast<const v4f*>(h);
        const size_t n4 = n >> 2;
        for (size_t i0 = threadIdx.x; i0 < n4; i0 += 4 * (size_t)blockDim.x) {
            v4f v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t i = i0 + u * (size_t)blockDim.x;
                v[u] = h4[i < n4 ? i : i0];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                m = fmaxf(m, fmaxf(fmaxf(fabsf(v[u][0]), fabsf(v[u][1])), fmaxf(fabsf(v[u][2]), fabsf(v[u][3]))));
        }
        done = n4 << 2;
    }
    for (size_t i = done + threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, fabsf(h[i]));
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
            (void)frexpf(m, &e);
            k = min(max(10 - e, -60), 60);
        }
        scale_out[0] = ldexpf(1.f, k);
        scale_out[1] = ldexpf(1.f, -k);
    }
}

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
    // split-f16 fragments: block (e, k, mt, kb) = [hi/lo][lane 64][8 halves]; half e8 of lane (q, i)
    // is W[f = 16 mt + i][g = 32 kb + 8 q + e8] * 2^k (natural channel order: the B operand is a
    // row-major z row in LDS)
    const int KB = (G + 31) / 32;
    float* ph = packed + total;
    const float scale = ph[filter_packed_h2_floats(G, F, K, E)];
    _Float16* out = reinterpret_cast<_Float16*>(ph);
    const size_t total_h = (size_t)E * K * MT * KB * 512;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total_h;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int e8 = idx & 7;
        const int l = (idx >> 3) & 63;
        size_t blk = idx >> 9;
        const int kb = blk % KB; blk /= KB;
        const int mt = blk % MT; blk /= MT;
        const int k = blk % K;
        const int e = blk / K;
        const int f = mt * 16 + (l & 15);
        const int g = kb * 32 + (l >> 4) * 8 + e8;
        float v = 0.f;
        if (f < F && g < G) v = h[(((size_t)f * E + e) * K + k) * G + g] * scale;
        const _Float16 hi = (_Float16)v;
        const size_t item = (idx >> 9) * 2;
        out[(item * 64 + l) * 8 + e8] = hi;
        out[((item + 1) * 64 + l) * 8 + e8] = (_Float16)(v - (float)hi);
    }
    // bf16x3 fragments (gnnpp_common.h, "b3"): block (e, k, mt, kb) = [plane 3][lane 64][8 bf16], same channel
    // order, w = h + m + l exactly (no scale)
    unsigned short* ob = reinterpret_cast<unsigned short*>(packed + filter_packed_b3_offset(G, F, K, E));
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total_h;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int e8 = idx & 7;
        const int l = (idx >> 3) & 63;
        size_t blk = idx >> 9;
        const int kb = blk % KB; blk /= KB;
        const int mt = blk % MT; blk /= MT;
        const int k = blk % K;
        const int e = blk / K;
        const int f = mt * 16 + (l & 15);
        const int g = kb * 32 + (l >> 4) * 8 + e8;
        float v = 0.f;
        if (f < F && g < G) v = h[(((size_t)f * E + e) * K + k) * G + g];
        unsigned ph, pm, pl;
        b3_split2(v, 0.f, ph, pm, pl);
        const size_t item = (idx >> 9) * 3;
        ob[(item * 64 + l) * 8 + e8] = (unsigned short)(ph & 0xffffu);
        ob[((item + 1) * 64 + l) * 8 + e8] = (unsigned short)(pm & 0xffffu);
        ob[((item + 2) * 64 + l) * 8 + e8] = (unsigned short)(pl & 0xffffu);
    }
}

// In-place fp32 -> split-f16 conversion of the valid rows of a z buffer (G = 128): a half-wave
// owns a row, reads all of it (16 bytes per lane), then writes the hi halves to the first 256 bytes
// of the row and the lo halves to the second 256 bytes.
__device__ __forceinline__ void split_rows(float* __restrict__ z, int row_lo, int row_hi, int zs,
                                           int wave, int nwaves, int lane, unsigned long long& bad) {
    typedef _Float16 v4h __attribute__((ext_vector_type(4)));
    const int half = lane >> 5, hl = lane & 31;
    for (int rb = row_lo + 2 * wave; rb < row_hi; rb += 2 * nwaves) {
        const int r = rb + half;
        const bool ok = r < row_hi;
        float* row = z + (ok ? r : rb) * zs;
        const v4f v = *reinterpret_cast<const v4f*>(row + 4 * hl);
        __builtin_amdgcn_wave_barrier();                 // all reads of a row precede its writes
        // range guard: |z| >= 65504 does not fit the hi half (rows >= row_hi are copies of valid rows)
        bad |= __ballot(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))) >= 65504.f);
        v4h h, l;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            h[c] = (_Float16)v[c];
            l[c] = (_Float16)(v[c] - (float)h[c]);
        }
        if (ok) {
            *reinterpret_cast<v2f*>(row + 2 * hl) = __builtin_bit_cast(v2f, h);
            *reinterpret_cast<v2f*>(row + 64 + 2 * hl) = __builtin_bit_cast(v2f, l);
        }
    }
}

// The S slabs sit in LDS column-major: row c = j*N + n of Sl holds column n of graph j's GSO, i.e. the
// weights node n gathers with.  Each row is compacted IN PLACE, once: a half-wave reads the whole row
// into registers (N <= 128: four values per lane), then writes the non-zero weights back to the front
// of the row in ascending m (ballot + prefix popcount give every non-zero its slot), their row indices
// m to the byte list idx[c][.], zeros behind them, and the degree to cnt[c].  Exact: structural zeros
// contribute nothing to the reference's dense product either.
__device__ __forceinline__ void build_lists(const LsigfArgs& p, float* __restrict__ Sl,
                                            unsigned char* __restrict__ idx,
                                            unsigned char* __restrict__ cnt, int R, int wave, int nwaves,
                                            int lane) {
    const int N = p.N;
    const int half = lane >> 5, hl = lane & 31;
    for (int rb = 2 * wave; rb < R; rb += 2 * nwaves) {          // wave-uniform trip count
        const int c = rb + half;
        const bool cv = c < R;
        float* wl = Sl + (cv ? c : rb) * p.Ns;
        unsigned char* il = idx + (cv ? c : rb) * p.Nl;
        float sv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = 32 * u + hl;
            sv[u] = (cv && m < N) ? wl[m] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();                 // the whole row is in registers before any write
        int base = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool nz = sv[u] != 0.f;
            const unsigned long long bal = __ballot(nz);
            const unsigned mine = half ? (unsigned)(bal >> 32) : (unsigned)bal;
            if (nz) {
                const int pos = base + __popc(mine & ((1u << hl) - 1u));
                il[pos] = (unsigned char)(32 * u + hl);
                wl[pos] = sv[u];
            }
            base += __popc(mine);
        }
        if (cv) {
            for (int pz = base + hl; pz < p.Ns; pz += 32) wl[pz] = 0.f;      // weights behind the list
            if (hl == 0) cnt[c] = (unsigned char)base;
        }
    }
}

// z_k[r,:] = sum over the neighbours m of node r (ascending) of S[m,n] * z_{k-1}[m,:] for the rows
// r in [row_lo, row_hi); one half-wave per row, 4 features per lane, 4 neighbours per trip:
// one 4-byte read of the index list, one 16-byte read of the weights, four 512-byte row reads.
__device__ __forceinline__ void gather_rows(const LsigfArgs& p, const float* __restrict__ Sl,
                                            const unsigned char* __restrict__ idx,
                                            const unsigned char* __restrict__ cnt,
                                            const float* __restrict__ zprev,
                                            float* __restrict__ zcur, int row_lo, int row_hi, int wave,
                                            int nwaves, int lane) {
    const int N = p.N, zs = p.zstride, GP = p.NG * 16;
    const int half = lane >> 5, hl = lane & 31;
    for (int rb = row_lo + 2 * wave; rb < row_hi; rb += 2 * nwaves) {   // wave-uniform trip count
        const int r = rb + half;
        const bool rv = r < row_hi;
        const int rr = rv ? r : rb;
        const int j = p.gpw == 1 ? 0 : rr / N;                   // (no integer division for big graphs)
        const float* wl = Sl + rr * p.Ns;                        // compacted weights of node rr
        const float* zg = zprev + j * N * zs;
        const unsigned char* il = idx + rr * p.Nl;
        const int deg = rv ? (int)cnt[rr] : 0;
        for (int c0 = 0; c0 < GP; c0 += 128) {
            const int col = c0 + 4 * hl;
            const bool live = rv && col < GP;
            const float* zc = zg + (col < GP ? col : 0);
            v4f acc = vzero();
            for (int d = 0; __ballot(d < deg) != 0ull; d += 4) {  // until both halves are done
                // entries past the degree: weight 0 and a stale (but valid) row index
                const unsigned pk = *reinterpret_cast<const unsigned*>(il + d);
                const v4f w = *reinterpret_cast<const v4f*>(wl + d);
                v4f zv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    zv[u] = *reinterpret_cast<const v4f*>(zc + __umul24((pk >> (8 * u)) & 255u, (unsigned)zs));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[0] = fmaf(w[u], zv[u][0], acc[0]);
                    acc[1] = fmaf(w[u], zv[u][1], acc[1]);
                    acc[2] = fmaf(w[u], zv[u][2], acc[2]);
                    acc[3] = fmaf(w[u], zv[u][3], acc[3]);
                }
            }
            if (live) *reinterpret_cast<v4f*>(zcur + r * zs + col) = acc;
        }
    }
}

// x -> z_0 rows (node-major in LDS) by the threads [t0, t0+nt) of the workgroup.  Index pairs
// (row, column) advance incrementally: no integer division in the loops.
__device__ __forceinline__ void stage_x(const LsigfArgs& p, float* __restrict__ z0, int g0, int ng,
                                        int t0, int nt, bool rezero) {
    const int N = p.N, zs = p.zstride, R = ng * N;
    if (t0 < 0) return;
    if (p.x_node_major) {
        const float* xs = p.x + (size_t)g0 * N * p.G;
        if ((p.G & 3) == 0) {
            const int G4 = p.G >> 2;
            const int dr = nt / G4, dc = nt - dr * G4;
            int r = t0 / G4, c = t0 - r * G4;
            for (; r < R; ) {
                *reinterpret_cast<v4f*>(z0 + r * zs + 4 * c) =
                    *reinterpret_cast<const v4f*>(xs + (size_t)r * p.G + 4 * c);
                r += dr; c += dc;
                if (c >= G4) { c -= G4; ++r; }
            }
        } else {
            const int dr = nt / p.G, dc = nt - dr * p.G;
            int r = t0 / p.G, c = t0 - r * p.G;
            for (; r < R; ) {
                z0[r * zs + c] = xs[(size_t)r * p.G + c];
                r += dr; c += dc;
                if (c >= p.G) { c -= p.G; ++r; }
            }
        }
    } else {
        // x[b][g][n], n < Nin; linear (coalesced) walk over each graph's G x Nin slab
        const int slab = p.G * p.Nin;
        const int dg = nt / p.Nin, dn = nt - dg * p.Nin;
        for (int j = 0; j < ng; ++j) {
            const float* xs = p.x + (size_t)(g0 + j) * slab;
            int g = t0 / p.Nin, n = t0 - g * p.Nin;
            for (int i = t0; i < slab; i += nt) {
                z0[(j * N + n) * zs + g] = xs[i];
                g += dg; n += dn;
                if (n >= p.Nin) { n -= p.Nin; ++g; }
            }
            if (rezero)                                 // rows n >= Nin must be zero again
                for (int i = t0; i < (N - p.Nin) * p.G; i += nt) {
                    const int nn = p.Nin + i / p.G, gg = i % p.G;
                    z0[(j * N + nn) * zs + gg] = 0.f;
                }
        }
    }
}

// dense S slabs of edge feature e -> LDS (fp64 -> fp32 like `S.float()`), threads [t0, t0+nt).
// LDS layout: row n = COLUMN n of the GSO (S[m,n] at Sl[n*Ns + m]); with s_transposed the roles of
// the two indices swap.  fp32 slabs whose size is a multiple of four floats are read 16 bytes at a time.
__device__ __forceinline__ void stage_s(const LsigfArgs& p, float* __restrict__ Sl, int g0, int ng,
                                        int e, int t0, int nt) {
    const int N = p.N, NN = N * N;
    if (t0 < 0) return;
    for (int j = 0; j < ng; ++j) {
        const size_t sidx = ((size_t)(p.s_batched ? (g0 + j) * p.E : 0) + e) * NN;
        float* dst = Sl + j * N * p.Ns;
        if (p.s_is_f64) {
            const double* src = reinterpret_cast<const double*>(p.S) + sidx;
            const int dm = nt / N, dn = nt - dm * N;
            int m = t0 / N, n = t0 - m * N;
            for (int i = t0; i < NN; i += nt) {
                dst[p.s_transposed ? m * p.Ns + n : n * p.Ns + m] = (float)src[i];
                m += dm; n += dn;
                if (n >= N) { n -= N; ++m; }
            }
        } else if (p.s_vec4) {
            const v4f* src = reinterpret_cast<const v4f*>(reinterpret_cast<const float*>(p.S) + sidx);
            const int step = 4 * nt, dm = step / N, dn = step - dm * N;
            int m = (4 * t0) / N, n = 4 * t0 - m * N;
            for (int i = t0; i < (NN >> 2); i += nt) {
                const v4f v = src[i];
                int mm = m, nn = n;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    dst[p.s_transposed ? mm * p.Ns + nn : nn * p.Ns + mm] = v[c];
                    if (++nn == N) { nn = 0; ++mm; }
                }
                m += dm; n += dn;
                if (n >= N) { n -= N; ++m; }
            }
        } else {
            const float* src = reinterpret_cast<const float*>(p.S) + sidx;
            const int dm = nt / N, dn = nt - dm * N;
            int m = t0 / N, n = t0 - m * N;
            for (int i = t0; i < NN; i += nt) {
                dst[p.s_transposed ? m * p.Ns + n : n * p.Ns + m] = src[i];
                m += dm; n += dn;
                if (n >= N) { n -= N; ++m; }
            }
        }
    }
}

// NW   = waves per workgroup (8 or 16).  A wave owns ONE 16-channel output tile, mt = wave % MTP
//        (MTP = 8, or 4 when F <= 64), and the row-tile chunk  wave / MTP  of RTW row tiles;
// RTW  = 16-row MFMA tiles per wave;  NGT = compile-time number of 16-wide input-feature groups
//        (8 for G = 128: the whole tap's A fragments live in registers and the next tap's are
//        prefetched during the shift; 0 = run-time NG, fragments loaded inside the loop).
// H2 (G = 128 only): the contraction runs on the f16 matrix pipe with both operands split in
//        hi + lo halves (3 MFMAs of K = 32 instead of 8 of K = 4, see encoder_kernel_h2.hip); the
//        shifts stay exact fp32.  z_k is converted in place once shift k+1 has read it.
// LDS: z buffers hold exactly R = gpw*N rows (no pad rows): MFMA B-fragment reads of the last,
//        partial row tile clamp their row to R-1 -- a D column only depends on its own B column, and
//        the columns of rows >= R are never stored.
template <int RTW, int NW, int NGT, bool H2>
__global__ __launch_bounds__(NW * 64) void lsigf_kernel(const LsigfArgs p) {
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    constexpr int NT = NW * 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int a = lane & 15;      // row inside a 16-row tile (MFMA j)
    const int q = lane >> 4;      // MFMA k slot

    // workgroup -> (first graph, part): with nsplit > 1 a group of 8 nsplit blocks carries 8 graphs, the blocks
    // b, b + 8, b + 16, .. of the group are the parts of one graph (same XCD under round-robin dispatch)
    int gblk = blockIdx.x, part = 0;
    if (p.nsplit > 1) {
        const int grp = gblk / (8 * p.nsplit), in = gblk - grp * 8 * p.nsplit;
        part = in >> 3;
        gblk = grp * 8 + (in & 7);
    }
    const int g0 = gblk * p.gpw;                       // first graph of this workgroup
    if (g0 >= p.B) return;                             // padding block of a split grid (uniform)
    const int ng = min(p.gpw, p.B - g0);               // graphs actually present
    const int N = p.N;
    const int R = ng * N;                              // valid rows
    const int RA = max(p.gpw * N, 8);                  // rows allocated per z buffer (>= 8: the epilogue
                                                       // parks act_w [5][F] in the second buffer)
    const int zs = p.zstride;
    const int NG = NGT ? NGT : p.NG;
    constexpr int NGA = NGT ? NGT : 1;
    const int mtp = p.MT > 4 ? 8 : 4;
    const int mt = wave & (mtp - 1);
    // this workgroup's row tiles [tile_lo, tile_hi) and rows [row_lo, row_hi)
    const int rt_all = (R + 15) >> 4;
    const int tile_lo = part * rt_all / p.nsplit;      // (nsplit <= rt_all: every part owns at least one row tile;
    const int tile_hi = (part + 1) * rt_all / p.nsplit;   //  nsplit > 1 only with gpw == 1, so rt_all == rt_total)
    const int row_lo = tile_lo * 16, row_hi = min(tile_hi * 16, R);
    const int rt0 = tile_lo + (wave / mtp) * RTW;      // first row tile of this wave
    const bool has_mfma = mt < p.MT && rt0 < tile_hi;

    float* zbuf0 = reinterpret_cast<float*>(gnnpp_smem);
    float* zbuf1 = zbuf0 + RA * zs;
    float* Sl = zbuf1 + RA * zs;                       // [gpw][N][Ns]
    unsigned char* idx = reinterpret_cast<unsigned char*>(Sl + RA * p.Ns);   // [gpw*N][Nl]
    unsigned char* cnt = idx + RA * p.Nl;              // [gpw*N]

    // Tap weights of the first tap: issued first so their L2 latency hides behind the staging.
    // Packed block (e,k,mt,gg): 64 lanes x 4 floats = the A fragments of four MFMA k-steps.
    const int ntaps = p.E * p.K;
    const size_t tap_stride = (size_t)p.MT_all * NG * 256;
    // fp32: NGA fragments of 4 k-steps; H2: 4 blocks x (hi, lo) fragments = the same 8 x 16 bytes
    v4f Acur[NGA], Anxt[NGA];
    auto load_tap = [&](v4f (&A)[NGA], int tap) {
        if (NGT && has_mfma) {
            const float* wt = (H2 ? p.wpk_h : p.wpk) + tap * tap_stride +
                              ((size_t)(p.mt0 + mt) * NGA * 64 + lane) * 4;
#pragma unroll
            for (int gg = 0; gg < NGA; ++gg) A[gg] = *reinterpret_cast<const v4f*>(wt + gg * 256);
        }
    };
    load_tap(Acur, 0);
    GNNPP_STAMP(blockIdx.x, 0, tid == 0);

    // ---- zero what the staging does not overwrite ---------------------------------------------------
    {
        // pad columns / missing nodes must read as zero (pad columns are never read when G % 16 == 0)
        const bool all = (p.G & 15) != 0 || (p.F & 15) != 0 || p.Nin < N;
        if (all) {
            const int n4 = (RA * zs) >> 2;             // zs is a multiple of 8
            v4f* z0 = reinterpret_cast<v4f*>(zbuf0);
            v4f* z1 = reinterpret_cast<v4f*>(zbuf1);
            for (int i = tid; i < n4; i += NT) { z0[i] = vzero(); z1[i] = vzero(); }
        }
        if (p.K > 1) {                                 // index lists: stale entries must be valid rows
            unsigned* iz = reinterpret_cast<unsigned*>(idx);
            for (int i = tid; i < (RA * p.Nl) >> 2; i += NT) iz[i] = 0u;
        }
        if (all) __syncthreads();                      // the x staging below overwrites zeroed cells
    }
    // x and S(e=0) are staged by disjoint thread ranges so their load latencies overlap
    {
        // the last quarter (small graphs) / half (N >= 32) of the threads stage S
        const int ns = (p.K > 1) ? (N >= 32 ? NT / 2 : NT / 4) : 0;
        if (ns && !GNNPP_ABLATE(p, 4)) stage_s(p, Sl, g0, ng, 0, tid >= NT - ns ? tid - (NT - ns) : -1, ns);
        stage_x(p, zbuf0, g0, ng, tid < NT - ns ? tid : -1, NT - ns, false);
    }

    unsigned long long bad = 0;                          // H2: lanes that handed |z| >= 65504 to the f16 pipe
    v4f acc[RTW], acc2[H2 ? RTW : 1];                    // H2: cross terms accumulate separately
#pragma unroll
    for (int t = 0; t < RTW; ++t) acc[t] = vzero();
#pragma unroll
    for (int t = 0; t < (H2 ? RTW : 1); ++t) acc2[t] = vzero();

    // B-fragment row of this lane for the wave's t-th row tile, clamped into the allocated rows
    auto brow = [&](int t) { return min((rt0 + t) * 16 + a, R - 1); };

    int tap = 0;
    for (int e = 0; e < p.E; ++e) {
        if (e > 0 && (p.K > 1 || H2)) {
            __syncthreads();                           // previous e is done with Sl and the z's
            if (p.K > 1) stage_s(p, Sl, g0, ng, e, tid, NT);
            // z_{e,0} = x: the ping-pong overwrote it when K > 2 (H2: converted it in place), so edge
            // features e > 0 re-stage it (E > 1 is outside the planner's configs: simple and
            // correct beats fast here).
            if (p.K > 2 || H2) stage_x(p, zbuf0, g0, ng, tid, NT, true);
        }
        __syncthreads();                               // z_0 (and Sl) visible
        GNNPP_STAMP(blockIdx.x, 1, tid == 0);
        if (p.K > 1 && !GNNPP_ABLATE(p, 1)) {
            build_lists(p, Sl, idx, cnt, R, wave, NW, lane);
            __syncthreads();
        }
        GNNPP_STAMP(blockIdx.x, 2, tid == 0);

        if (H2) {
            // order per tap: shift z_k -> z_{k+1} | convert z_k in place | contract z_k
            for (int k = 0; k < p.K; ++k, ++tap) {
                float* zcur = (k & 1) ? zbuf1 : zbuf0;
                float* znxt = (k & 1) ? zbuf0 : zbuf1;
                if (k + 1 < p.K && !GNNPP_ABLATE(p, 1)) {
                    // only the LAST shift may be restricted to this workgroup's rows
                    const bool last = k + 2 == p.K;
                    gather_rows(p, Sl, idx, cnt, zcur, znxt, last ? row_lo : 0, last ? row_hi : R, wave,
                                NW, lane);
                }
                if (p.zs) {                              // training: keep z_{e,k} (fp32), own rows
                    float* zd = p.zs + ((size_t)tap * p.B + g0) * N * p.G;
                    for (int i = row_lo * p.G + tid; i < row_hi * p.G; i += NT) {
                        const int r = i / p.G, c = i - r * p.G;
                        zd[i] = zcur[r * zs + c];
                    }
                }
                __syncthreads();                         // every reader of the fp32 z_k is done
                GNNPP_STAMP(blockIdx.x, 3 + 3 * k, tid == 0 && k < 3);      // shift k -> k+1 done
                split_rows(zcur, row_lo, row_hi, zs, wave, NW, lane, bad);
                __syncthreads();
                GNNPP_STAMP(blockIdx.x, 4 + 3 * k, tid == 0 && k < 3);      // z_k split
                if (has_mfma && !GNNPP_ABLATE(p, 2)) {
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) {
                        const v8h Ah = __builtin_bit_cast(v8h, Acur[2 * kb]);
                        const v8h Al = __builtin_bit_cast(v8h, Acur[2 * kb + 1]);
#pragma unroll
                        for (int t = 0; t < RTW; ++t) {   // one row tile's B pair at a time: 8 VGPRs
                            const float* zrow = zcur + brow(t) * zs + q * 4;
                            const v8h Bh = __builtin_bit_cast(
                                v8h, *reinterpret_cast<const v4f*>(zrow + kb * 16));
                            const v8h Bl = __builtin_bit_cast(
                                v8h, *reinterpret_cast<const v4f*>(zrow + 64 + kb * 16));
                            acc2[t] = mfma16h(Ah, Bl, acc2[t]);
                            acc[t] = mfma16h(Ah, Bh, acc[t]);
                            acc2[t] = mfma16h(Al, Bh, acc2[t]);
                        }
                    }
                }
                // the next tap's fragments fly during the next shift (no second register set)
                if (tap + 1 < ntaps) load_tap(Acur, tap + 1);
                if (k + 1 < p.K) __syncthreads();         // z_k's buffer is the target of the next shift
                GNNPP_STAMP(blockIdx.x, 5 + 3 * k, tid == 0 && k < 3);      // tap k contracted
            }
        } else
        for (int k = 0; k < p.K; ++k, ++tap) {
            float* zcur = (k & 1) ? zbuf1 : zbuf0;
            if (tap + 1 < ntaps) load_tap(Anxt, tap + 1);       // in flight during the shift
            if (k > 0) {
                if (!GNNPP_ABLATE(p, 1)) {
                    const bool last = k + 1 == p.K;
                    gather_rows(p, Sl, idx, cnt, (k & 1) ? zbuf0 : zbuf1, zcur, last ? row_lo : 0,
                                last ? row_hi : R, wave, NW, lane);
                }
                __syncthreads();
            }
            if (p.zs) {                                  // training: keep z_{e,k} for dW = dy . z^T
                float* zd = p.zs + ((size_t)tap * p.B + g0) * N * p.G;
                for (int i = row_lo * p.G + tid; i < row_hi * p.G; i += NT) {
                    const int r = i / p.G, c = i - r * p.G;
                    zd[i] = zcur[r * zs + c];
                }
            }
            // ---- contraction of tap (e,k) on MFMA: D[f, row] += W[f, g] z[row, g] --------------
            if (has_mfma && !GNNPP_ABLATE(p, 2)) {
                if (NGT) {
#pragma unroll
                    for (int gg = 0; gg < NGA; ++gg) {
                        v4f Bf[RTW];
#pragma unroll
                        for (int t = 0; t < RTW; ++t)
                            Bf[t] = *reinterpret_cast<const v4f*>(zcur + brow(t) * zs + q * 4 + gg * 16);
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int t = 0; t < RTW; ++t)
                                acc[t] = mfma16(Acur[gg][s], Bf[t][s], acc[t]);
                    }
#pragma unroll
                    for (int gg = 0; gg < NGA; ++gg) Acur[gg] = Anxt[gg];
                } else {
                    const float* wt = p.wpk + tap * tap_stride + ((size_t)(p.mt0 + mt) * NG * 64 + lane) * 4;
                    for (int gg = 0; gg < NG; ++gg) {
                        const v4f A = *reinterpret_cast<const v4f*>(wt + gg * 256);
                        v4f Bf[RTW];
#pragma unroll
                        for (int t = 0; t < RTW; ++t)
                            Bf[t] = *reinterpret_cast<const v4f*>(zcur + brow(t) * zs + q * 4 + gg * 16);
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int t = 0; t < RTW; ++t)
                                acc[t] = mfma16(A[s], Bf[t][s], acc[t]);
                    }
                }
            }
        }
    }

    // ---- epilogue: bias (+ReLU) -> LDS [row][f] -> coalesced store / fused action head --------
    if (GNNPP_ABLATE(p, 8)) return;
    if (H2 && p.range_flag && bad) *p.range_flag = 1;
    __syncthreads();                                   // every wave is done reading z
    GNNPP_STAMP(blockIdx.x, 12, tid == 0);
    float* ybuf = zbuf0;
    float* actw = zbuf1;                               // act_w staged here: [5][F]
    if (p.act_w)
        for (int i = tid; i < 5 * p.F; i += NT) actw[i] = p.act_w[i];
    if (has_mfma) {
        const int f0 = mt * 16 + q * 4;
        const float h2_inv = H2 ? p.wpk_h[filter_packed_h2_floats(p.G, p.F_all, p.K, p.E) + 1] : 1.f;
        v4f bv = vzero();
        if (p.bias && !p.bias_per_node) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = (f0 + r < p.F) ? p.bias[p.f0 + f0 + r] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < RTW; ++t) {
            const int row = (rt0 + t) * 16 + a;
            if (rt0 + t < tile_hi && row < R) {          // (rows >= R have no storage)
                if (p.bias_per_node) {                   // b[f, n]: this lane's row is node n
                    const int n = row % N;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        bv[r] = (f0 + r < p.F) ? p.bias[(size_t)(p.f0 + f0 + r) * N + n] : 0.f;
                }
                v4f v = H2 ? (acc[t] + acc2[t]) * h2_inv + bv : acc[t] + bv;
                if (p.relu) v = vrelu(v);
                *reinterpret_cast<v4f*>(ybuf + row * zs + f0) = v;
            }
        }
    }
    __syncthreads();
    GNNPP_STAMP(blockIdx.x, 13, tid == 0);

    const int nrows = row_hi - row_lo;
    if (p.y) {
        if (p.y_node_major) {
            float* yd = p.y + (size_t)g0 * N * p.F_all + p.f0;
            if ((p.F & 3) == 0 && (p.F_all & 3) == 0) {
                const int F4 = p.F >> 2;
                const float* md = p.y_mask ? p.y_mask + (size_t)g0 * N * p.F_all + p.f0 : nullptr;
                for (int i = tid; i < nrows * F4; i += NT) {
                    const int r = row_lo + i / F4, c = i % F4;
                    v4f v = *reinterpret_cast<const v4f*>(ybuf + r * zs + 4 * c);
                    if (md) {
                        const v4f m = *reinterpret_cast<const v4f*>(md + (size_t)r * p.F_all + 4 * c);
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[u] = m[u] > 0.f ? v[u] : 0.f;
                    }
                    *reinterpret_cast<v4f*>(yd + (size_t)r * p.F_all + 4 * c) = v;
                }
            } else {
                const float* md = p.y_mask ? p.y_mask + (size_t)g0 * N * p.F_all + p.f0 : nullptr;
                for (int i = tid; i < nrows * p.F; i += NT) {
                    const int r = row_lo + i / p.F, c = i % p.F;
                    float v = ybuf[r * zs + c];
                    if (md && !(md[(size_t)r * p.F_all + c] > 0.f)) v = 0.f;
                    yd[(size_t)r * p.F_all + c] = v;
                }
            }
        } else if (p.nsplit == 1) {
            const int slab = p.F * p.Nin;
            for (int j = 0; j < ng; ++j) {
                float* yd = p.y + ((size_t)(g0 + j) * p.F_all + p.f0) * p.Nin;
                for (int i = tid; i < slab; i += NT) {
                    const int f = i / p.Nin, n = i - f * p.Nin;
                    yd[i] = ybuf[(j * N + n) * zs + f];
                }
            }
        } else {
            // one graph, this workgroup's nodes [row_lo, row_hi) that exist in the output (n < Nin)
            const int n_lo = row_lo, n_hi = min(row_hi, p.Nin), nn = max(n_hi - n_lo, 0);
            float* yd = p.y + ((size_t)g0 * p.F_all + p.f0) * p.Nin;
            for (int i = tid; i < p.F * nn; i += NT) {
                const int f = i / nn, n = n_lo + i - f * nn;
                yd[(size_t)f * p.Nin + n] = ybuf[n * zs + f];
            }
        }
    }
    if (p.act_w) {
        // Action head on MFMA: D[a5, row] = sum_f act_w[a5, f] * y[row, f]; A fragment from the
        // staged act_w (rows >= 5 are zero), B fragment from the staged y tile.
        const int i5 = lane & 15;
        for (int rt = tile_lo + wave; rt < tile_hi; rt += NW) {
            // Term order of the fused policy kernel (encoder_kernel_h2.hip), whose four waves each chain their
            // two 16-feature tiles from zero and add the partial logits in wave order: identical results.
            v4f d = vzero();
            const int r = rt * 16 + a;                  // this lane's row; it holds a5 = 4*q + reg
            const float* yrow = ybuf + min(r, R - 1) * zs;
            for (int g2 = 0; g2 < p.MT; g2 += 2) {
                v4f dp = vzero();
                for (int gg = g2; gg < min(g2 + 2, p.MT); ++gg) {
                    const int f0 = gg * 16 + q * 4;
                    v4f A = vzero();
                    if (i5 < 5) {
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            if (f0 + s < p.F) A[s] = actw[i5 * p.F + f0 + s];
                    }
                    const v4f Bv = *reinterpret_cast<const v4f*>(yrow + f0);
                    dp = mfma16x4(A, Bv, dp);
                }
                d = g2 == 0 ? dp : d + dp;
            }
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
    GNNPP_STAMP(blockIdx.x, 14, tid == 0);
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
// Tuning state: read on every dispatch, written by gnnpp_set_tuning (relaxed atomics: a concurrent
// dispatch sees the old or the new value, never a torn one; every value computes the same function).
std::atomic<int> g_filter_gpw{0};     // 0: heuristic below; > 0: forced graphs per workgroup
std::atomic<int> g_filter_waves{0};   // 0: heuristic; 8 or 16: forced waves per workgroup
std::atomic<int> g_filter_split{0};   // 0: heuristic; 1: never, 2: always two workgroups per graph (gpw == 1)
#ifdef GNNPP_MEASURE
std::atomic<int> g_filter_ablate{0};  // measurement-only phase ablation mask (see LsigfArgs::ablate)
#endif

template <int RTW, int NW, int NGT, bool H2>
static hipError_t launch_one(const LsigfArgs& a, int grid, size_t smem, hipStream_t st) {
    static LdsAttrOnce once;
    set_lds_attr_once(once, reinterpret_cast<const void*>(&lsigf_kernel<RTW, NW, NGT, H2>), kLdsBytes);
    hipLaunchKernelGGL((lsigf_kernel<RTW, NW, NGT, H2>), dim3(grid), dim3(NW * 64), smem, st, a);
    return hipGetLastError();
}

template <int RTW, int NW>
static hipError_t launch_ng(const LsigfArgs& a, int grid, size_t smem, hipStream_t st) {
    if (a.NG != 8) return launch_one<RTW, NW, 0, false>(a, grid, smem, st);
    // The input-gradient launch (s_transposed: x := dy) keeps the fp32 MFMA: cotangents of 1e-4 .. 1e-7
    // sit in the f16 subnormal range, where the unscaled hi/lo split of the B operand loses them.
    // (GNNPP_PREC_FP32 contracts on the exact fp32 MFMA here: this kernel's two z buffers leave no LDS for bf16x3
    // planes at the sizes it exists for; the policy's own kernels carry the bf16x3 form.)
    return (a.G == 128 && a.prec == kPrecSplitF16 && !a.s_transposed)
               ? launch_one<RTW, NW, 8, true>(a, grid, smem, st)
               : launch_one<RTW, NW, 8, false>(a, grid, smem, st);
}

template <int NW>
static hipError_t launch_rtw(int rtw, const LsigfArgs& a, int grid, size_t smem, hipStream_t st) {
    switch (rtw) {
        case 1: return launch_ng<1, NW>(a, grid, smem, st);
        case 2: return launch_ng<2, NW>(a, grid, smem, st);
        case 3: return launch_ng<3, NW>(a, grid, smem, st);
        case 4: return launch_ng<4, NW>(a, grid, smem, st);
        case 5: return launch_ng<5, NW>(a, grid, smem, st);
        case 6: return launch_ng<6, NW>(a, grid, smem, st);
        case 7: return launch_ng<7, NW>(a, grid, smem, st);
        default: return hipErrorInvalidValue;
    }
}

static size_t lsigf_smem(const LsigfArgs& a, int gpw) {
    const size_t rows = (size_t)(gpw * a.N > 8 ? gpw * a.N : 8);   // = RA of the kernel
    const size_t lists = a.K > 1 ? rows * a.Nl + ((rows + 15) & ~(size_t)15) : 0;
    return 2 * rows * a.zstride * 4 + rows * a.Ns * 4 + lists;
}

// Chooses graphs-per-workgroup and waves-per-workgroup and checks the LDS budget.
// Returns a GNNPP_* code; on success `a` is complete and plan holds the launch geometry.
struct LsigfPlan { int grid, nw, rtw; size_t smem; };
constexpr int kMaxRows = 112;       // rows (graphs x nodes) one workgroup keeps in LDS = 7 MFMA row tiles

int lsigf_plan(LsigfArgs& a, LsigfPlan& plan) {
    a.NG = (a.G + 15) / 16;
    a.MT = (a.F + 15) / 16;
#ifdef GNNPP_MEASURE
    a.ablate = g_filter_ablate.load(std::memory_order_relaxed);
#endif
    if (a.F_all <= 0) { a.F_all = a.F; a.f0 = 0; }    // single launch covering every output feature
    a.mt0 = a.f0 / 16;
    a.MT_all = (a.F_all + 15) / 16;
    a.wpk_h = a.wpk + filter_packed_f32_floats(a.G, a.F_all, a.K, a.E);
    a.wpk_b = a.wpk + filter_packed_b3_offset(a.G, a.F_all, a.K, a.E);
    if (a.MT > 8) return -2;                          // F > 128 per launch: lsigf_launch splits F
    const int wide = a.NG > a.MT ? a.NG : a.MT;
    a.zstride = 16 * wide + 8;
    a.Ns = (a.N + 3) & ~3;                            // 16-byte reads of a row's compacted weights
    a.Nl = a.Ns;
    a.s_vec4 = !a.s_is_f64 && ((a.N * a.N) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.S) & 15) == 0;
    if (a.N > kMaxRows) return -2;
    // graphs per workgroup: fill the 16-row MFMA tiles, but keep the 256 CUs busy.  Cost model:
    // rounds over the chip x (fixed staging/latency cost + work per row tile).
    int best = 1;
    double best_cost = 1e30;
    const int max_gpw = kMaxRows / a.N;
    for (int g = 1; g <= max_gpw && g <= a.B; ++g) {
        if (lsigf_smem(a, g) > (size_t)kLdsBytes) break;
        const int rt = (g * a.N + 15) / 16;
        const int wgs = (a.B + g - 1) / g;
        const int rounds = (wgs + 255) / 256;
        const double cost = rounds * (1.0 + rt);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = g; }
    }
    const int forced_gpw = g_filter_gpw.load(std::memory_order_relaxed);
    if (forced_gpw > 0 && forced_gpw <= max_gpw && forced_gpw <= a.B &&
        lsigf_smem(a, forced_gpw) <= (size_t)kLdsBytes)
        best = forced_gpw;
    a.gpw = best;
    a.rt_total = (a.gpw * a.N + 15) / 16;
    plan.smem = lsigf_smem(a, a.gpw);
    if (plan.smem > (size_t)kLdsBytes) return -2;
    plan.grid = (a.B + a.gpw - 1) / a.gpw;
    // Several workgroups per graph when single-graph workgroups leave at least half of the 256 CUs idle and the
    // graph has row tiles to share (see the header): as many parts as still give every workgroup a CU of its own
    // (the padded grid of 8-graph groups x parts <= 256), at most one per row tile -- 16 graphs of 100 agents (the
    // shard one GPU of eight holds of config 5) run as 112 workgroups of one tile instead of 32 of four / three
    // (r05; r04: two parts at most).  GNNPP_TUNE_FILTER_SPLIT forces 1 / n parts.
    const int forced_split = g_filter_split.load(std::memory_order_relaxed);
    a.nsplit = 1;
    if (a.gpw == 1 && a.rt_total >= 2) {
        const int groups = (plan.grid + 7) / 8;
        if (forced_split >= 2) a.nsplit = forced_split < a.rt_total ? forced_split : a.rt_total;
        else if (forced_split == 0 && plan.grid <= 128 && a.rt_total >= 4) {
            const int room = 256 / (8 * groups);                    // parts that keep one workgroup per CU
            a.nsplit = room < 2 ? 2 : room < a.rt_total ? room : a.rt_total;
        }
    }
    const int tiles_per_wg = (a.rt_total + a.nsplit - 1) / a.nsplit;
    if (a.nsplit > 1) plan.grid = ((plan.grid + 7) / 8) * 8 * a.nsplit;      // groups of 8 graphs x nsplit parts
    // waves per workgroup: 16 when there are enough rows / row tiles to feed them
    const int mtp = a.MT > 4 ? 8 : 4;
    plan.nw = (a.gpw * a.N > 24) ? 16 : 8;
    const int forced_nw = g_filter_waves.load(std::memory_order_relaxed);
    if (forced_nw == 8 || forced_nw == 16) plan.nw = forced_nw;
    const int chunks = plan.nw / mtp;
    plan.rtw = (tiles_per_wg + chunks - 1) / chunks;
    return 0;
}

static int policy_filter_dispatch(LsigfArgs a, const LsigfPlan& plan, hipStream_t st);   // policy_filter_kernel.hip
static int lsigf_small_dispatch(LsigfArgs a, hipStream_t st);                            // lsigf_small_kernel.hip

int lsigf_dispatch(const LsigfArgs& a, const LsigfPlan& plan, hipStream_t st) {
    const int sm = lsigf_small_dispatch(a, st);                       // thousands of small graphs: the throughput kernel
    if (sm <= 0) return sm;
    const int pf = policy_filter_dispatch(a, plan, st);               // the policy step's shape: its own kernel
    if (pf <= 0) return pf;
    const hipError_t err = plan.nw == 16 ? launch_rtw<16>(plan.rtw, a, plan.grid, plan.smem, st)
                                         : launch_rtw<8>(plan.rtw, a, plan.grid, plan.smem, st);
    return err == hipSuccess ? 0 : -3;
}

// Any F: output features in chunks of 128 (the accumulators of one launch); all chunks are planned
// before the first one is enqueued, so a failing call has enqueued nothing.
int lsigf_launch(LsigfArgs a, hipStream_t st) {
    constexpr int kChunk = 128, kMaxChunks = 64;
    const int F_all = a.F;
    const int nchunks = (F_all + kChunk - 1) / kChunk;
    if (nchunks > kMaxChunks) return -2;
    if (nchunks > 1 && a.act_w) return -2;            // the fused action head needs all features at once
    LsigfArgs args[kMaxChunks];
    LsigfPlan plans[kMaxChunks];
    for (int c = 0; c < nchunks; ++c) {
        args[c] = a;
        args[c].F_all = F_all;
        args[c].f0 = c * kChunk;
        args[c].F = F_all - c * kChunk < kChunk ? F_all - c * kChunk : kChunk;
        if (c > 0) args[c].zs = nullptr;              // the tap signals do not depend on the chunk
        const int rc = lsigf_plan(args[c], plans[c]);
        if (rc) return rc;
    }
    for (int c = 0; c < nchunks; ++c) {
        const int rc = lsigf_dispatch(args[c], plans[c], st);
        if (rc) return rc;
    }
    return 0;
}

int filter_pack_launch(const float* h, float* packed, int G, int F, int K, int E, hipStream_t st) {
    const size_t total = filter_packed_floats(G, F, K, E);
    float* scale = packed + filter_packed_f32_floats(G, F, K, E) + filter_packed_h2_floats(G, F, K, E);
    hipLaunchKernelGGL(filter_scale_kernel, dim3(1), dim3(1024), 1024 * sizeof(float), st, h, scale,
                       (size_t)F * E * K * G);
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
