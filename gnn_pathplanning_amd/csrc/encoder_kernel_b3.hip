// Encoder kernel, bf16x3 schedule ("b3") -- the DEFAULT: fp32-equivalent arithmetic on the bf16 matrix pipe.
//
// Same layers, same 16-agent tile, same one-MFMA-tile-per-output-position plan as the other two schedules
// (encoder_pack.hip), same asm-managed weight ring as the split-f16 schedule (encoder_kernel_h2.hip).  What
// differs is the operand form (gnnpp_common.h, "bf16x3"):
//
//      x = xh + xm + xl   exactly (three bf16 planes, fp32's exponent range),
//      w x ~= wh xl + wl xh + wm xm + wh xm + wm xh + wh xh      (fp32 accumulate; dropped terms <= 2^-23 |w x|)
//
// Six v_mfma_f32_16x16x32_bf16 per 32-channel block: 96 matrix-pipe cycles where the exact-fp32 schedule's
// eight v_mfma_f32_16x16x4_f32 need 256 and the split-f16 schedule's three f16 MFMAs 48 -- but unlike the latter
// there is NO input range (|x| < 65504), no weight pre-scaling and no guard flag: every finite fp32 activation
// and weight is represented exactly.  This is what the reference's plain fp32 aten path
// (graphs/models/decentralplanner.py:284-290, utils/graphUtils/graphML.py:2342-2366) is replaced by when the
// caller does not ask for anything else (GNNPP_PREC_FP32).
//
// Activations in LDS: [position][kb][plane 3][lane 64] x 16 bytes -- 6 bytes per element.  The largest
// activation (25 positions x 32 channels x 16 agents) is then 75 KiB, and two workgroups must still share a
// CU (160 KiB), so the kernel lives in ONE 75 KiB region R plus a 2.5 (5) KiB table:
//   * the padded observations (two 32-bit words per pixel: h | m << 16, and l) are staged into R; L0 keeps its
//     pooled outputs in registers (<= 7 windows x 2 channel tiles per wave) until every wave has read its last
//     pixel, then writes them over the observations;
//   * L1 and L2 run "in place" the same way (accumulators live across the barrier that ends the reads);
//   * the 2x2 / 1x1 layers use disjoint 24 / 24 / 12 KiB pieces of R.
// <true, K>: the fused policy kernel (one graph of N <= 16 agents per workgroup: encoder, K-tap graph filter,
// ReLU, action head and optionally the simulator step), as encoder_kernel_h2<true, K>; the filter's shifts stay
// exact fp32 (dense S on the fp32 MFMA), its tap contraction runs on bf16x3 planes written by the producer of
// each z_k (FC epilogue / shift epilogue) -- no conversion pass.
#include <utility>

#include "gnnpp_common.h"

namespace gnnpp {

// per-wave item stream: L1 (54) | L2 (54) | L3 (54) | L4 (108) | FC (24); one item = one 16-byte plane fragment of
// one (kb, tap, mt)
constexpr int kb_L1 = 0, kb_L2 = 54, kb_L3 = 108, kb_L4 = 162, kb_FC = 270, kb_END = 294;
constexpr int kb_FILT = kb_END;           // fused: [tap K][kb 4][mt_local 2][plane 3] = 24 K items

// LDS map (bytes)
constexpr int kB3Frag = 1024;                              // one plane fragment: 64 lanes x 16 bytes
constexpr int kB3Region = 25 * 3 * kB3Frag;                // R: 76 800
constexpr int kB3Obs2 = kObsFloatsLds * 4;                 // second observation word array (l planes)
constexpr int kB3L2out = 0, kB3L3out = 24 * kB3Frag, kB3L4out = 48 * kB3Frag;
constexpr int kB3Table = kB3Region;                        // BatchNorm table (640 floats); FUSED: later the head's
constexpr int kB3TableBytes = EncLayout::kBssFloats * 4 + 16;   // (+ 4 plane-skipping flags)  constants (776 floats) + the GSO [16][17]
constexpr int kB3TableBytesFused = 5120;
constexpr int kB3SsmOff = 776 * 4;
constexpr size_t kB3Smem = kB3Region + kB3TableBytes;              // 79 360: two workgroups per CU
constexpr size_t kB3SmemFused = kB3Region + kB3TableBytesFused;    // 81 920: exactly half of the CU's LDS
static_assert(2 * kObsFloatsLds * 4 <= kB3Region && kB3SsmOff + 16 * 17 * 4 <= kB3TableBytesFused &&
              2 * kB3SmemFused <= (size_t)kLdsBytes, "b3 LDS map");
// fused tail: z ping-pong (fp32 rows, stride kZs) | K plane buffers (row stride 800 B) | partial logits
constexpr int kB3ZBytes = 16 * kZs * 4;                    // 8 704
constexpr int kB3PRow = 3 * 256 + 32;                      // 800: three 256-byte planes + pad (bank spread as kZs)
constexpr int kB3PBytes = 16 * kB3PRow;                    // 12 800
constexpr int kB3POff = 2 * kB3ZBytes;
static_assert(kB3POff + kPolicyTapsMax * kB3PBytes + 4 * 16 * 8 * 4 <= kB3Region, "fused tail fits R");
// the simulator step's LDS lies where the (dead) fp32 z rows were
constexpr size_t policy_sim_occ_bytes_b3() {
    return (size_t)kB3POff - 8 * kMaxAgents * sizeof(int) - kGsoSmemBytes;
}

struct WStreamB {                 // per-wave segment bases (wave-uniform: SGPRs) + this lane's offset
    const float* seg[6];
    int lane_bytes;
};

__device__ __forceinline__ const float* ring_item_ptr(const WStreamB& ws, int idx) {
    if (idx < kb_L2) return ws.seg[0] + (idx - kb_L1) * EncLayout::kHItem;
    if (idx < kb_L3) return ws.seg[1] + (idx - kb_L2) * EncLayout::kHItem;
    if (idx < kb_L4) return ws.seg[2] + (idx - kb_L3) * EncLayout::kHItem;
    if (idx < kb_FC) return ws.seg[3] + (idx - kb_L4) * EncLayout::kHItem;
    if (idx < kb_FILT) return ws.seg[4] + (idx - kb_FC) * EncLayout::kHItem;
    // filter block (tap, mt, kb) of gnnpp_filter_pack's b3 region: ((tap * 8 + mt) * 4 + kb) * 768 floats, plane p
    // at + 256 p; seg[5] already points at this wave's first channel tile
    const int j = idx - kb_FILT, pl = j % 3, ml = (j / 3) & 1, kb = (j / 6) & 3, tap = j / 24;
    return ws.seg[5] + tap * (8 * 4 * 768) + ml * (4 * 768) + kb * 768 + pl * 256;
}

// One (kb, tap) step IT of NMT channel tiles over a compile-time position set: the three plane fragments of every
// position the tap reaches (from LDS, or from the preloaded registers Pin), six MFMAs per (position, tile), small
// terms first; consecutive MFMAs hit different accumulators.
template <int IT, int NKB, int H, int W, int NMT, int NSLOT, class PosFn, bool PRELOAD, int CH = NSLOT>
__device__ __forceinline__ void b3_tap_mfma(const v4f* in, const v4f* Pin, const v8b (&A)[NMT][3],
                                            v4f (&acc)[NSLOT][NMT], int lane) {
    constexpr int kb = IT / 9, tap = IT % 9;
    constexpr int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
    for (int j0 = 0; j0 < NSLOT; j0 += CH) {               // CH slots at a time bounds the B registers
        v8b B[CH][3];
#pragma unroll
        for (int jj = 0; jj < CH; ++jj) {
            const int j = j0 + jj;
            int y = 0, x = 0;
            const bool used = j < NSLOT && PosFn::get(j, y, x);
            const int iy = y + dy, ix = x + dx;
            if (used && iy >= 0 && iy < H && ix >= 0 && ix < W) {
                const int o = ((iy * W + ix) * NKB + kb) * 3;
#pragma unroll
                for (int p = 0; p < 3; ++p) B[jj][p] = as_b8(PRELOAD ? Pin[o + p] : in[(o + p) * 64 + lane]);
            }
        }
#pragma unroll
        for (int term = 0; term < kB3Terms; ++term) {
#pragma unroll
            for (int jj = 0; jj < CH; ++jj) {
                const int j = j0 + jj;
                int y = 0, x = 0;
                const bool used = j < NSLOT && PosFn::get(j, y, x);
                const int iy = y + dy, ix = x + dx;
                if (used && iy >= 0 && iy < H && ix >= 0 && ix < W) {
                    const bool fresh = term == 0 && first_step_of_slot<H, W, PosFn>(IT, j);
#pragma unroll
                    for (int m = 0; m < NMT; ++m)
                        acc[j][m] = mfma16b(A[m][b3_term_a(term)], B[jj][b3_term_b(term)],
                                            fresh ? vzero() : acc[j][m]);
                }
            }
        }
    }
}

// ---- software-pipelined form of the same step, for the layers whose input sits in LDS (L1, L2) ----------------
// hipcc schedules b3_tap_mfma as  [LDS reads of a chunk] [wait] [its MFMAs] [next chunk's reads] [wait] ...: with one
// workgroup on a CU (and for the YOUNGER of two, which only gets the issue slots the older one leaves: measured,
// profiles/r03a_wg_spread.jsonl) every chunk pays the LDS latency with the matrix pipe idle -- a lone workgroup ran
// L1 / L2 at 47 / 63 % of the pipe's rate.  Here the plane fragments of chunk c + 1 are REQUESTED before the MFMAs of
// chunk c are issued (two register sets, alternating), across step boundaries too; sched_barrier(0) pins the order,
// the compiler's s_waitcnt lgkmcnt(n) then waits only for the older set (LDS returns in order).
template <int IT, int CHI, int NKB, int H, int W, int NSLOT, int CH, class PosFn>
__device__ __forceinline__ void b3_load_chunk(const v4f* in, v4f (&B)[CH][3], int lane) {
    constexpr int kb = IT / 9, tap = IT % 9;
    constexpr int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
    for (int jj = 0; jj < CH; ++jj) {
        const int j = CHI * CH + jj;
        int y = 0, x = 0;
        const bool used = j < NSLOT && PosFn::get(j, y, x);
        const int iy = y + dy, ix = x + dx;
        if (used && iy >= 0 && iy < H && ix >= 0 && ix < W) {
            const int o = ((iy * W + ix) * NKB + kb) * 3;
#pragma unroll
            for (int p = 0; p < 3; ++p) B[jj][p] = in[(o + p) * 64 + lane];
        }
    }
}

template <int IT, int CHI, int NKB, int H, int W, int NMT, int NSLOT, int CH, class PosFn>
__device__ __forceinline__ void b3_mfma_chunk(const v4f (&B)[CH][3], const v8b (&A)[NMT][3],
                                              v4f (&acc)[NSLOT][NMT]) {
    constexpr int tap = IT % 9;
    constexpr int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
    for (int term = 0; term < kB3Terms; ++term) {
#pragma unroll
        for (int jj = 0; jj < CH; ++jj) {
            const int j = CHI * CH + jj;
            int y = 0, x = 0;
            const bool used = j < NSLOT && PosFn::get(j, y, x);
            const int iy = y + dy, ix = x + dx;
            if (used && iy >= 0 && iy < H && ix >= 0 && ix < W) {
                const bool fresh = term == 0 && first_step_of_slot<H, W, PosFn>(IT, j);
#pragma unroll
                for (int m = 0; m < NMT; ++m)
                    acc[j][m] = mfma16b(A[m][b3_term_a(term)], as_b8(B[jj][b3_term_b(term)]),
                                        fresh ? vzero() : acc[j][m]);
            }
        }
    }
}

// step IT of NIT: for each of its chunks, request the NEXT chunk's planes (next chunk of this step, or the first
// chunk of step IT + 1), then issue this chunk's MFMAs.  Bb[(linear chunk index) & 1] holds a chunk's planes;
// the layer's prologue loads chunk (0, 0) into Bb[0].
template <int IT, int NIT, int NKB, int H, int W, int NMT, int NSLOT, int CH, class PosFn, int... CHS>
__device__ __forceinline__ void b3_step_pipelined_impl(const v4f* in, v4f (&Bb)[2][CH][3], const v8b (&A)[NMT][3],
                                                       v4f (&acc)[NSLOT][NMT], int lane,
                                                       std::integer_sequence<int, CHS...>) {
    constexpr int NCH = (NSLOT + CH - 1) / CH;
    auto chunk = [&](auto chc) {
        constexpr int ch = decltype(chc)::value;
        constexpr int L = IT * NCH + ch;
        if constexpr (ch + 1 < NCH) b3_load_chunk<IT, ch + 1, NKB, H, W, NSLOT, CH, PosFn>(in, Bb[(L + 1) & 1], lane);
        else if constexpr (IT + 1 < NIT) b3_load_chunk<IT + 1, 0, NKB, H, W, NSLOT, CH, PosFn>(in, Bb[(L + 1) & 1], lane);
        __builtin_amdgcn_sched_barrier(0);
        b3_mfma_chunk<IT, ch, NKB, H, W, NMT, NSLOT, CH, PosFn>(Bb[L & 1], A, acc);
        __builtin_amdgcn_sched_barrier(0);
    };
    (chunk(std::integral_constant<int, CHS>{}), ...);
}
template <int IT, int NIT, int NKB, int H, int W, int NMT, int NSLOT, int CH, class PosFn>
__device__ __forceinline__ void b3_step_pipelined(const v4f* in, v4f (&Bb)[2][CH][3], const v8b (&A)[NMT][3],
                                                  v4f (&acc)[NSLOT][NMT], int lane) {
    b3_step_pipelined_impl<IT, NIT, NKB, H, W, NMT, NSLOT, CH, PosFn>(
        in, Bb, A, acc, lane, std::make_integer_sequence<int, (NSLOT + CH - 1) / CH>{});
}

// The weight stream of a layer (item order [kb][tap][mt][plane]): for every step IT take the 3 NMT fragments off
// the ring, refill the slots, hand them to body(IT, A).  Wave-uniform, branch-free (tools/check_ring_isa.py).
template <int END, int START, int NMT, class Body, int... IT>
__device__ __forceinline__ void b3_stream_steps(const WStreamB& ws, v4f (&ring)[kRingH], Body&& body,
                                                std::integer_sequence<int, IT...>) {
    auto step = [&](auto itc) {
        constexpr int it = decltype(itc)::value;
        __builtin_amdgcn_sched_barrier(kSchedItemMask);
        v8b A[NMT][3];
#pragma unroll
        for (int m = 0; m < NMT; ++m) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int idx = START + (it * NMT + m) * 3 + p;
                A[m][p] = as_b8(ring_take_f4<END>(ring, idx));
                h2_ring_load<END>(ws, ring, idx + kRingH);
            }
        }
        body(itc, A);
    };
    (step(std::integral_constant<int, IT>{}), ...);
}

// The same stream for the column-packed layers (NMT = 2), whose steps are short (12 NA MFMAs): the six fragments of step
// it + 1 are taken (and their slots refilled) BETWEEN the MFMAs of step it -- body(itc, A, item) calls item(kc) for
// kc = 0 .. 5 at the places it chooses --, two fragment sets alternating.  Ring order, and with it every vmcnt, is the one
// of b3_stream_steps; only step 0's six takes stay a block in front of the layer.  (A lone wave issues one MFMA per 16
// cycles and anything else in between for free; as a block the 36 instructions of a step's takes cost it 4 cycles each:
// tools/probe/cp_stream_probe.hip, profiles/r04_cp_stream_probe.jsonl.)
template <int END, int START, class Body, int... IT>
__device__ __forceinline__ void b3_stream_steps_spread(const WStreamB& ws, v4f (&ring)[kRingH], Body&& body,
                                                       std::integer_sequence<int, IT...>) {
    constexpr int NIT = sizeof...(IT);
    v8b A[2][2][3];
    auto item = [&](auto itc, auto kc) {
        constexpr int it = decltype(itc)::value, k = decltype(kc)::value, idx = START + it * 6 + k;
        A[it & 1][k / 3][k % 3] = as_b8(ring_take_f4<END>(ring, idx));
        h2_ring_load<END>(ws, ring, idx + kRingH);
    };
    __builtin_amdgcn_sched_barrier(kSchedItemMask);
    [&]<int... K>(std::integer_sequence<int, K...>) {
        (item(std::integral_constant<int, 0>{}, std::integral_constant<int, K>{}), ...);
    }(std::make_integer_sequence<int, 6>{});
    auto step = [&](auto itc) {
        constexpr int it = decltype(itc)::value;
        body(itc, A[it & 1], [&](auto kc) {
            if constexpr (it + 1 < NIT) item(std::integral_constant<int, it + 1>{}, kc);
        });
    };
    (step(std::integral_constant<int, IT>{}), ...);
}

// ... and with the ring's slots as the MFMAs' A operands (ring_mfma16b: no copy out of the ring, tied accumulators): step it
// waits ONCE for its six items (the six loads of step it + 1 are the only younger ones), body(itc, base, refill) issues the
// MFMAs on items base .. base + 5 and calls refill(kc) behind the last MFMA that reads fragment kc -- the slot then takes
// item base + kc + 12.  A fragment is requested one step (12 NA MFMAs) before its first use, not two: waves with a single
// tile keep the form above.  The ring's invariant at a step boundary (loads issued up to base + 11) is the one of the other
// two forms, so layers of different forms follow each other; the ORDER of a step's six refills only matters to a following
// layer that takes fragments one by one (vmcnt counts loads in issue order): body must refill in ascending order then.
template <int END, int START, class Body, int... IT>
__device__ __forceinline__ void b3_stream_steps_direct(const WStreamB& ws, v4f (&ring)[kRingH], Body&& body,
                                                       std::integer_sequence<int, IT...>) {
    auto step = [&](auto itc) {
        constexpr int it = decltype(itc)::value, base = START + it * 6;
        __builtin_amdgcn_sched_barrier(0);
        ring_wait_for<END>(base + 5, base + 11);
        __builtin_amdgcn_sched_barrier(0);
        body(itc, std::integral_constant<int, base>{}, [&](auto kc) {
            h2_ring_load<END>(ws, ring, base + decltype(kc)::value + kRingH);
        });
    };
    (step(std::integral_constant<int, IT>{}), ...);
}

// a whole 2x2 layer for one position set, its input preloaded into registers
template <int END, int START, int NKB, int H, int W, int NMT, int NSLOT, class PosFn>
__device__ __forceinline__ void b3_conv_preload(const WStreamB& ws, v4f (&ring)[kRingH], const v4f* in,
                                                v4f (&acc)[NSLOT][NMT], int lane) {
    v4f Pin[H * W * NKB * 3];
#pragma unroll
    for (int i = 0; i < H * W * NKB * 3; ++i) Pin[i] = in[i * 64 + lane];
    b3_stream_steps<END, START, NMT>(ws, ring, [&](auto itc, const v8b (&A)[NMT][3]) {
        b3_tap_mfma<decltype(itc)::value, NKB, H, W, NMT, NSLOT, PosFn, true>(in, Pin, A, acc, lane);
    }, std::make_integer_sequence<int, 9 * NKB>{});
}

// ==== COLUMN PACKING for teams of N <= kCpMaxAgents agents (fused policy kernel, <true, K, CP = true>) ==================
// With agents on the MFMA's 16 columns a 10-agent graph uses 10 / 16 of every MFMA.  For the two layers that hold 57 %
// of the kernel's matrix work the columns are (agent, position) PAIRS instead:
//   L1 (32 -> 32 @ 5x5)  column c = pos * N + n, c < 25 N: ceil(25 N / 16) tiles of 16 consecutive columns (N = 10: 16
//        tiles x 9 taps instead of 25 position tiles x 6.76 reachable taps = 144 instead of 169 tile-taps), a wave takes
//        tiles wave, wave + 4, ..: 4 / 4 / 4 / 4 instead of 7 / 6 / 6 / 6 positions;
//   L2 (32 -> 64 @ 5x5, the 4x4 the pool reads)  one tile PER AGENT whose 16 columns are the 16 output positions, the four
//        positions of a pool window in four adjacent lanes (the 2x2 max is two quad_perm DPP steps): N tiles x 9 taps
//        instead of 16 position tiles x 7.56 (N = 10: 90 instead of 121 tile-taps).
// A tap is then a per-lane LDS address, not a compile-time fragment: a column whose tap leaves the image reads a cell
// of zeros (its products are exact zeros: acc + 0 == acc, so every output keeps the value -- and the bits -- of the
// agents-on-columns schedule, whose first reachable tap starts from a zero accumulator too).  L0 keeps agents on its
// columns (it is VALU- / latency-bound, not pipe-bound) and only writes its pooled windows in L1's input layout;
// L3 / L4 / FC / the filter tail are unchanged (2x2 images: tap skipping beats packing there).
//
// LDS layouts (both inside region R; a "row" = 16 cells x [plane 3][q 4] x 16 bytes = 3 KiB, cell slot s at
// row + plane * 1024 + q * 256 + s * 16, so the 16 lanes (q, q + 1 halves) of a ds_read_b128 group hit 16 distinct slots):
//   L1 input (L0's output)  cell u = pos * N + n at row u >> 4, slot u & 15; the rows END at the end of R (L0 writes
//        them while the pixel words at the front of R are still being read), followed by one row of zeros (a lane
//        whose tap leaves the image reads ITS slot of that row);
//   L2 input (L1's output)  agent n's 25 cells: slot (4 (y & 3) + (x & 3) + o(n)) & 15, o(n) = 4 (n & 3) + (n >> 2), in row
//        n (y, x < 4) | 12 + (n >> 2) (y = 4) | 15 + (n & 3) (x = 4) | 19 (y = x = 4); row 20 = zeros.  Any 4x4 window of
//        one agent's 5x5 image lies in 16 distinct slots (conflict-free tap reads), and the rows are shared between
//        agents without collisions (the rotation o(n) interleaves them): 20 rows hold 12 x 25 cells.
constexpr int kCpMaxAgents = 12;
constexpr int kCpRow = 3 * kB3Frag;                        // 3 072
constexpr int kCpL1Tiles = 5;                              // column tiles of L1 per wave: ceil(ceil(25 * 12 / 16) / 4)
constexpr int kCpL2Tiles = 6;                              // agent tiles of L2 per wave: ceil(12 / 2)
constexpr int kCpL2Rows = 21;
static_assert(kCpL2Rows * kCpRow <= kB3Region && ((25 * kCpMaxAgents + 15) / 16 + 3) / 4 <= kCpL1Tiles, "column-packed layouts fit R");

struct CpGeom {                   // wave-uniform (SGPRs)
    int N, ncol, T1, l1in;
    unsigned rcpN;                // (c * rcpN) >> 16 == c / N for c < 5 000
};
__device__ __forceinline__ CpGeom cp_geom(int N) {
    CpGeom g;
    g.N = N;
    g.ncol = 25 * N;
    g.T1 = (25 * N + 15) >> 4;
    g.l1in = kB3Region - (g.T1 + 1) * kCpRow;                      // (+ 1: the row of zeros behind the data rows)
    g.rcpN = (unsigned)(65536.f / (float)N) + 1u;
    return g;
}
// byte offset of cell (y, x) of agent n inside a (plane, q) block of the L2-input layout
__device__ __forceinline__ int cp_l2in_cell(int y, int x, int n) {
    const int o = 4 * (n & 3) + (n >> 2);
    const int slot = (4 * (y & 3) + (x & 3) + o) & 15;
    const int row = y == 4 ? (x == 4 ? 19 : 12 + (n >> 2)) : (x == 4 ? 15 + (n & 3) : n);
    return __mul24(row, kCpRow) + slot * 16;
}
// L0's pooled window `win` (three plane fragments of lane (q, agent a)): fragment win of R, or -- CP -- cell win * N + a of
// the L1-input layout (lanes beyond the graph's agents hold nothing)
template <bool CP>
__device__ __forceinline__ void b3_store_window(char* smem, const CpGeom& g, int win, int lane, const v4f (&pl)[3]) {
    if (!CP) {
        v4f* const R4 = reinterpret_cast<v4f*>(smem);
#pragma unroll
        for (int p = 0; p < 3; ++p) R4[(win * 3 + p) * 64 + lane] = pl[p];
    } else {
        const int a = lane & 15, q = lane >> 4;
        if (a < g.N) {
            const int u = win * g.N + a;
            char* const dst = smem + g.l1in + (u >> 4) * kCpRow + q * 256 + (u & 15) * 16;
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<v4f*>(dst + p * kB3Frag) = pl[p];
        }
    }
}

// L0 for observations that need all three planes (anything but exact bf16 values): window by window, sixteen word
// reads per position (eight h | m words, eight l words, position offset = instruction immediate), three v_perm per
// register pair, 6 MFMAs per channel tile; the pooled outputs of all of a wave's windows stay in registers until
// every wave is done with the pixels.  (Scheduling left to the compiler: this is the rare path -- the simulator's
// observations take b3_l0_stream<true> -- and the explicit stream's two word sets do not fit beside 56 held outputs.)
template <bool CP>
__device__ __forceinline__ void b3_l0_generic(const float* __restrict__ pk, const unsigned* obsw, const float* sstab,
                                              char* smem, const CpGeom& g, int wave, int lane) {
    const int a = lane & 15, q = lane >> 4;
    v8b A0[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p)
            A0[i][p] = as_b8(*reinterpret_cast<const v4f*>(pk + EncLayout::kB0 + ((i * 3 + p) * 64 + lane) * 4));
    int aoff[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
        aoff[e] = a * kAgentStride + (q < 3 ? q * (kPadHW * kPadHW) + (e / 3) * kPadHW + e % 3
                                            : e < 3 ? e * (kPadHW * kPadHW) + 2 * kPadHW + 2 : 0);
    v4f res[7][2];
#pragma unroll
    for (int wi = 0; wi < 7; ++wi) {
        const int win = wave + 4 * wi;
        if (win < 25) {                                  // (wave-uniform)
            __builtin_amdgcn_sched_barrier(kSchedItemMask);
            const int wy = win / 5, wx = win - wy * 5;
            const unsigned* base = obsw + (2 * wy) * kPadHW + 2 * wx;
            v4f acc[4][2];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                unsigned d1[8], d2[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    d1[e] = base[aoff[e] + (pp >> 1) * kPadHW + (pp & 1)];
                    d2[e] = base[kObsFloatsLds + aoff[e] + (pp >> 1) * kPadHW + (pp & 1)];
                }
                unsigned bh[4], bm[4], bl[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    bh[w] = __builtin_amdgcn_perm(d1[2 * w + 1], d1[2 * w], 0x05040100u);
                    bm[w] = __builtin_amdgcn_perm(d1[2 * w + 1], d1[2 * w], 0x07060302u);
                    bl[w] = __builtin_amdgcn_perm(d2[2 * w + 1], d2[2 * w], 0x05040100u);
                }
                const v4u bhv = {bh[0], bh[1], bh[2], bh[3]}, bmv = {bm[0], bm[1], bm[2], bm[3]},
                          blv = {bl[0], bl[1], bl[2], bl[3]};
                const v8b B[3] = {__builtin_bit_cast(v8b, bhv), __builtin_bit_cast(v8b, bmv),
                                  __builtin_bit_cast(v8b, blv)};
#pragma unroll
                for (int term = 0; term < kB3Terms; ++term)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[pp][i] = mfma16b(A0[i][b3_term_a(term)], B[b3_term_b(term)],
                                             term == 0 ? vzero() : acc[pp][i]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                v4f sc, sh;
                load_ss(sstab + EncLayout::kBssL0, 32, i, q, sc, sh);
                // pooled on the raw accumulators; `sc` is |scale| (the sign is in the weights: encoder_pack.hip)
                v4f r = vmax(vmax(acc[0][i], acc[1][i]), vmax(acc[2][i], acc[3][i]));
                r = vfma(r, sc, sh);
#pragma unroll
                for (int c = 0; c < 4; ++c) r[c] = b3_relu_clamp(r[c]);
                res[wi][i] = r;
            }
        }
    }
    __syncthreads();                                     // every wave has read its last pixel
#pragma unroll
    for (int wi = 0; wi < 7; ++wi) {
        const int win = wave + 4 * wi;
        if (win < 25) {
            v4f pl[3];
            b3_split8_clamped(res[wi][0], res[wi][1], pl);
            b3_store_window<CP>(smem, g, win, lane, pl);
        }
    }
}

// L0 of one wave: windows wave, wave + 4, .. (7 for wave 0, 6 otherwise) x their 4 positions, one stream.
//   position P:  [request the words of P + DEPTH]  |  [v_perm the words of P into plane fragments; its MFMAs;
//   BatchNorm + ReLU / running max of P - 1]   -- the second group is one scheduling region: the VALU work slots in
//   between the MFMAs.  ONE_PLANE (plane skipping, see the staging code): only the h | m word array is read, only the
//   h plane is built, three MFMAs (wh, wm, wl against xh) per channel tile instead of six.
template <bool ONE_PLANE, bool CP>
__device__ __forceinline__ void b3_l0_stream(const float* __restrict__ pk, const unsigned* obsw,
                                             const float* sstab, char* smem, const CpGeom& g,
                                             v4f (&res)[ONE_PLANE ? 3 : 5][2], int wave, int lane) {
    constexpr int DEPTH = ONE_PLANE ? 3 : 1;              // positions the word requests run ahead
    constexpr int NHELD = ONE_PLANE ? 3 : 5;              // windows (per wave) whose output must wait for the barrier
    static_assert((4 * NHELD) * 3 * kB3Frag >= (ONE_PLANE ? 1 : 2) * kObsFloatsLds * 4, "held windows cover the pixels");
    // (CP: window win >= 4 NHELD is written at l1in + ((win N) >> 4) rows >= 40 KB for every N <= kCpMaxAgents, behind
    // the 27 KB of h | m pixel words the ONE_PLANE stream reads: cp_direct_windows_clear_pixels below)
    constexpr int NW = ONE_PLANE ? 8 : 16;                // words per position
    const int a = lane & 15, q = lane >> 4;
    v8b A0[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p)
            A0[i][p] = as_b8(*reinterpret_cast<const v4f*>(pk + EncLayout::kB0 + ((i * 3 + p) * 64 + lane) * 4));
    int aoff[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
        aoff[e] = a * kAgentStride + (q < 3 ? q * (kPadHW * kPadHW) + (e / 3) * kPadHW + e % 3
                                            : e < 3 ? e * (kPadHW * kPadHW) + 2 * kPadHW + 2 : 0);
    v4f sc[2], sh[2];
    load_ss(sstab + EncLayout::kBssL0, 32, 0, q, sc[0], sh[0]);
    load_ss(sstab + EncLayout::kBssL0, 32, 1, q, sc[1], sh[1]);
    unsigned d[DEPTH + 1][NW];
    v4f acc[2][2];
    v4f run[2];
    auto request = [&](unsigned (&dd)[NW], int P) {        // the pixel words of position P (window P / 4)
        const int win = wave + 4 * (P >> 2), pp = P & 3;
        const int wy = win / 5, wx = win - wy * 5;
        const unsigned* base = obsw + (2 * wy + (pp >> 1)) * kPadHW + 2 * wx + (pp & 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            dd[e] = base[aoff[e]];
            if (!ONE_PLANE) dd[8 + e] = base[kObsFloatsLds + aoff[e]];
        }
    };
    auto compute = [&](const unsigned (&dd)[NW], v4f (&ac)[2]) {
        unsigned bh[4], bm[4], bl[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            bh[w] = __builtin_amdgcn_perm(dd[2 * w + 1], dd[2 * w], 0x05040100u);
            if (!ONE_PLANE) {
                bm[w] = __builtin_amdgcn_perm(dd[2 * w + 1], dd[2 * w], 0x07060302u);
                bl[w] = __builtin_amdgcn_perm(dd[8 + 2 * w + 1], dd[8 + 2 * w], 0x05040100u);
            }
        }
        const v4u bhv = {bh[0], bh[1], bh[2], bh[3]};
        const v8b Bh = __builtin_bit_cast(v8b, bhv);
        if (ONE_PLANE) {
#pragma unroll
            for (int p = 2; p >= 0; --p)                      // small planes first: wl xh, wm xh, wh xh
#pragma unroll
                for (int i = 0; i < 2; ++i) ac[i] = mfma16b(A0[i][p], Bh, p == 2 ? vzero() : ac[i]);
        } else {
            const v4u bmv = {bm[0], bm[1], bm[2], bm[3]}, blv = {bl[0], bl[1], bl[2], bl[3]};
            const v8b B[3] = {Bh, __builtin_bit_cast(v8b, bmv), __builtin_bit_cast(v8b, blv)};
#pragma unroll
            for (int term = 0; term < kB3Terms; ++term)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    ac[i] = mfma16b(A0[i][b3_term_a(term)], B[b3_term_b(term)], term == 0 ? vzero() : ac[i]);
        }
    };
    // (An inline-asm v_max_f32 here -- to spare the canonicalising v_max x, x, x the compiler puts in front of fmaxf on
    // MFMA results -- was measured WRONG on the GPU: asm consumers of MFMA results get no hazard wait states.)
    auto epilogue = [&](const v4f (&ac)[2], int P) {       // the 2x2 max of the window so far, on the RAW accumulators;
#pragma unroll                                             // BatchNorm (|scale|: the sign is in the weights) + ReLU +
        for (int i = 0; i < 2; ++i) {                      // the split's clamp once per window
            run[i] = (P & 3) == 0 ? ac[i] : vmax(run[i], ac[i]);
            if ((P & 3) == 3) {
                run[i] = vfma(run[i], sc[i], sh[i]);
#pragma unroll
                for (int c = 0; c < 4; ++c) run[i][c] = b3_relu_clamp(run[i][c]);
            }
        }
#if defined(__HIP_DEVICE_COMPILE__)
        // Pin the evaluation HERE: these are pure VALU operations, which the optimiser otherwise sinks to their use
        // at the end of the window -- every position's accumulators would stay live until then (sched_barrier orders
        // instructions, not the places where values are computed).
        asm volatile("" : "+v"(run[0]), "+v"(run[1]));
#endif
    };
    auto window_done = [&](auto wic) {                     // the pooled window wi of this wave: keep, or write now
        constexpr int wi = decltype(wic)::value;
        if constexpr (wi < NHELD) {
            res[wi][0] = run[0];
            res[wi][1] = run[1];
        } else {
            const int win = wave + 4 * wi;
            v4f pl[3];
            b3_split8_clamped(run[0], run[1], pl);
            b3_store_window<CP>(smem, g, win, lane, pl);
        }
    };
    // positions [P0, P1): every index below is a compile-time constant (explicit unrolling: the register arrays must
    // never be indexed dynamically)
    auto prologue = [&](auto p0c, auto p1c, auto kc) {
        constexpr int P = decltype(p0c)::value + decltype(kc)::value;
        if constexpr (P < decltype(p1c)::value) request(d[P % (DEPTH + 1)], P);
    };
    auto body = [&](auto p0c, auto p1c, auto kc) {
        constexpr int P0 = decltype(p0c)::value, P1 = decltype(p1c)::value, P = P0 + decltype(kc)::value;
        if constexpr (P + DEPTH < P1) request(d[(P + DEPTH) % (DEPTH + 1)], P + DEPTH);
        __builtin_amdgcn_sched_barrier(0);
        compute(d[P % (DEPTH + 1)], acc[P & 1]);
        if constexpr (P > P0) {
            epilogue(acc[(P - 1) & 1], P - 1);
            if constexpr (((P - 1) & 3) == 3) window_done(std::integral_constant<int, ((P - 1) >> 2)>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (P == P1 - 1) {
            epilogue(acc[P & 1], P);
            window_done(std::integral_constant<int, (P >> 2)>{});
        }
    };
    auto stream = [&](auto p0c, auto p1c) {
        constexpr int N = decltype(p1c)::value - decltype(p0c)::value;
        auto pro = [&](auto... ks) { (prologue(p0c, p1c, ks), ...); };
        auto run_all = [&](auto... ks) { (body(p0c, p1c, ks), ...); };
        [&]<int... K>(std::integer_sequence<int, K...>) { pro(std::integral_constant<int, K>{}...); }(
            std::make_integer_sequence<int, DEPTH>{});
        [&]<int... K>(std::integer_sequence<int, K...>) { run_all(std::integral_constant<int, K>{}...); }(
            std::make_integer_sequence<int, N>{});
    };
    stream(std::integral_constant<int, 0>{}, std::integral_constant<int, 24>{});
    if (wave == 0) stream(std::integral_constant<int, 24>{}, std::integral_constant<int, 28>{});   // window 24
}

// ---- L1, column-packed: 32 -> 32 @ 5x5, columns c = pos * N + n; this wave's tiles are wave, wave + 4, .. ---------------
// Per tap the source column of lane j is c + (dy 5 + dx) N: tile-independent slot (j + shift) & 15 and a tile-independent
// row offset, so a tile's three plane reads are `base + immediate`; a lane whose tap leaves the 5x5 image (four flag
// bits per tile, computed once) reads the zero cell instead.
// The MFMA stream of a wave is the sequence of units (tap, tile); the plane fragments of unit u + 1 -- the next tile, or
// the first tile of the next tap -- are requested before the MFMAs of unit u are issued (two register sets).  NA, the
// number of tiles of THIS wave, is a template argument: a run-time `if (i < na)` around every unit puts each unit in
// its own basic block, the compiler then waits for lgkmcnt(0) -- the prefetch included -- in front of every MFMA group
// (measured: L2 14.7 us instead of 11.9 for 26 % fewer MFMAs); the layer is instantiated for NA = 0 .. NT and the
// wave jumps to its copy once (every copy carries the same ring traffic: tools/check_ring_isa.py walks each path).
// One unit = the 12 MFMAs of a (tap, tile).  EVERY unit's per-lane LDS address is computed at the START of the layer
// (9 NA registers, pinned with an empty asm so that nothing is rematerialised next to its read): a unit's instruction
// stream is then what the agents-on-columns layers' is -- the three plane reads of unit u + 1 (address register +
// immediate), twelve MFMAs, one s_waitcnt.  History (C2, two workgroups per CU): run-time `if (i < na)` around every
// unit -> lgkmcnt(0) in front of every MFMA group, L2 14.7 us; addresses computed per unit, as a block in front of the
// MFMAs or spread between them -> 10.9 - 11.9 us, 21 - 24 cycles per MFMA whatever the placement (a VALU instruction
// between two MFMAs costs the wave far more than its issue slot -- MI355X_MICROARCH.md, "one EXTRA issue slot" -- and
// the compiler re-used freshly written accumulator registers for the temporaries); this form: profiles/r04_cp_ab.jsonl.
// The twelve MFMAs in a FIXED order -- terms ascending, the two channel tiles alternating, so that consecutive MFMAs
// never share an accumulator; sched_barrier(0) after every statement: nothing moves across.
// which of the next step's six ring items (b3_stream_steps_spread) unit i of NA handles at hook slot s (0 .. 5; -1: none):
// item k belongs to unit k NA / 6, a unit's items are spread evenly over its slots
__host__ __device__ __forceinline__ constexpr int cp_ring_item_at(int NA, int i, int s) {
    int first = -1, cnt = 0;
    for (int k = 0; k < 6; ++k)
        if (k * NA / 6 == i) { if (first < 0) first = k; ++cnt; }
    if (cnt == 0 || s % (6 / cnt) != 0 || s / (6 / cnt) >= cnt) return -1;
    return first + s / (6 / cnt);
}
template <class Load, class Hook>
__device__ __forceinline__ void cp_unit(const v8b (&A)[2][3], const v4f (&B)[3], v4f& c0, v4f& c1, Load&& load, Hook&& hook) {
#define GNNPP_CP_M(T, ACC, MI)                                                                          \
    ACC = mfma16b(A[MI][b3_term_a(T)], as_b8(B[b3_term_b(T)]), ACC);                                   \
    __builtin_amdgcn_sched_barrier(0);
#define GNNPP_CP_H(S)                                                                                   \
    hook(std::integral_constant<int, S>{});                                                             \
    __builtin_amdgcn_sched_barrier(0);
    GNNPP_CP_M(0, c0, 0)
    load(0);
    __builtin_amdgcn_sched_barrier(0);
    GNNPP_CP_M(0, c1, 1)
    load(1);
    __builtin_amdgcn_sched_barrier(0);
    GNNPP_CP_M(1, c0, 0)
    load(2);
    __builtin_amdgcn_sched_barrier(0);
    GNNPP_CP_M(1, c1, 1)
    GNNPP_CP_H(0)
    GNNPP_CP_M(2, c0, 0)
    GNNPP_CP_H(1)
    GNNPP_CP_M(2, c1, 1)
    GNNPP_CP_H(2)
    GNNPP_CP_M(3, c0, 0)
    GNNPP_CP_H(3)
    GNNPP_CP_M(3, c1, 1)
    GNNPP_CP_H(4)
    GNNPP_CP_M(4, c0, 0)
    GNNPP_CP_H(5)
    GNNPP_CP_M(4, c1, 1)
    GNNPP_CP_M(5, c0, 0)
    GNNPP_CP_M(5, c1, 1)
#undef GNNPP_CP_H
#undef GNNPP_CP_M
}
// The same unit on the ring's registers (b3_stream_steps_direct; BASE = the step's first stream item, fragment (m, plane) =
// item BASE + 3 m + plane).  LAST: the step's last unit -- every fragment is refilled behind its last reader (plane 2 after
// term 1, plane 1 after term 4, plane 0 after term 5), or, ORDERED, all six in ascending order behind the unit.
template <int BASE, bool FRESH, bool LAST, bool ORDERED, class Load, class Refill>
__device__ __forceinline__ void cp_unit_direct(v4f (&ring)[kRingH], const v4f (&B)[3], v4f& c0, v4f& c1, Load&& load,
                                               Refill&& refill) {
    // (FRESH: the layer's first tap -- the accumulators start from the first product, nobody clears them)
#define GNNPP_CP_M(T, ACC, MI)                                                                          \
    if constexpr (FRESH && T == 0) ring_mfma16b_first(ring, BASE + 3 * MI + b3_term_a(T), B[b3_term_b(T)], ACC);   \
    else ring_mfma16b(ring, BASE + 3 * MI + b3_term_a(T), B[b3_term_b(T)], ACC);                       \
    __builtin_amdgcn_sched_barrier(0);
#define GNNPP_CP_R(K)                                                                                   \
    if constexpr (LAST && !ORDERED) { refill(std::integral_constant<int, K>{}); __builtin_amdgcn_sched_barrier(0); }
    GNNPP_CP_M(0, c0, 0)
    load(0);
    __builtin_amdgcn_sched_barrier(0);
    GNNPP_CP_M(0, c1, 1)
    load(1);
    __builtin_amdgcn_sched_barrier(0);
    GNNPP_CP_M(1, c0, 0)
    load(2);
    __builtin_amdgcn_sched_barrier(0);
    GNNPP_CP_M(1, c1, 1)
    GNNPP_CP_R(2)
    GNNPP_CP_M(2, c0, 0)
    GNNPP_CP_R(5)
    GNNPP_CP_M(2, c1, 1)
    GNNPP_CP_M(3, c0, 0)
    GNNPP_CP_M(3, c1, 1)
    GNNPP_CP_M(4, c0, 0)
    GNNPP_CP_R(1)
    GNNPP_CP_M(4, c1, 1)
    GNNPP_CP_R(4)
    GNNPP_CP_M(5, c0, 0)
    GNNPP_CP_R(0)
    GNNPP_CP_M(5, c1, 1)
    GNNPP_CP_R(3)
    if constexpr (LAST && ORDERED) {
        [&]<int... K>(std::integer_sequence<int, K...>) { (refill(std::integral_constant<int, K>{}), ...); }(
            std::make_integer_sequence<int, 6>{});
        __builtin_amdgcn_sched_barrier(0);
    }
#undef GNNPP_CP_R
#undef GNNPP_CP_M
}
// an address is COMPLETE where it is computed (left to the optimiser its last additions sink to the read that uses it)
__device__ __forceinline__ void cp_pin(int& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
}

struct CpL1Lane {                 // per-lane constants of cp_layer1
    int j, q, zaddr, tile0;       // tile0: this lane's (q) block of the wave's first tile row
    unsigned fl;                  // per tile: y == 0 | y == 4 | x == 0 | x == 4  (4 bits each)
};
// LDS address of lane j's planes for unit (TAP, I): the source column of lane j is c + (dy 5 + dx) N -- a tile-independent
// slot (j + shift) & 15 and row offset, plus the tile's four rows --, or the lane's OWN slot of the zero row when the tap
// leaves the image: sixteen lanes keep sixteen distinct slots whichever of them are redirected (one shared zero cell put
// every redirected lane on the bank of whichever valid lane held that slot: PMC, r04).  Branch-free on purpose: as a
// ternary the compiler turns the choice into a divergent branch around the address arithmetic, and EXEC games inside the
// ring's region are what tools/check_ring_isa.py refuses to reason about.
template <int TAP, int I>
__device__ __forceinline__ int cp_l1_addr(const CpGeom& g, const CpL1Lane& ln, int wave) {
    constexpr int dy = TAP / 3 - 1, dx = TAP % 3 - 1;
    constexpr unsigned tmask = (dy < 0 ? 1u : 0u) | (dy > 0 ? 2u : 0u) | (dx < 0 ? 4u : 0u) | (dx > 0 ? 8u : 0u);
    // (tap part | tile part: written so that the tap part is common to a tap's units and the tile part to a tile's --
    // as one product (wave + 4 I + (sh >> 4)) * row the compiler issued a 64-bit multiply-add per unit)
    const int sh = ln.j + (dy * 5 + dx) * g.N;                     // (arithmetic shift / mask below: floor semantics)
    const int slot = (sh & 15) * 16;
    const int tap_part = __mul24(sh >> 4, kCpRow) + slot;
    const int cell = tap_part + (ln.tile0 + I * 4 * kCpRow);
    if (tmask == 0u) return cell;
    const int zero_cell = ln.zaddr + slot;
    const int outside = -(int)(((ln.fl >> (4 * I)) & tmask) != 0u);     // all ones / zero
    return cell ^ ((cell ^ zero_cell) & outside);
}
__device__ __forceinline__ void cp_l1_load(const char* smem, int addr, v4f (&B)[3]) {
#pragma unroll
    for (int p = 0; p < 3; ++p) B[p] = *reinterpret_cast<const v4f*>(smem + addr + p * kB3Frag);
}
template <int END, int NA>
__device__ __forceinline__ void cp_layer1_main(const WStreamB& ws, v4f (&ring)[kRingH], const char* smem, const CpGeom& g,
                                               const CpL1Lane& ln, int wave, v4f (&acc)[kCpL1Tiles][2]) {
    constexpr int NU = 9 * NA;                                      // units of this wave
    v4f Bb[2][3];
    if constexpr (NA > 0) {
        int addr[NU];                                               // unit u = (tap, tile): this lane's cell, tile row included
        [&]<int... U>(std::integer_sequence<int, U...>) {
            ((addr[U] = cp_l1_addr<U / NA, U % NA>(g, ln, wave), cp_pin(addr[U])), ...);
        }(std::make_integer_sequence<int, NU>{});
        __builtin_amdgcn_sched_barrier(0);
        GNNPP_STAMP(blockIdx.x, 7, wave == 0 && ln.j == 0 && ln.q == 0);
        cp_l1_load(smem, addr[0], Bb[0]);
        if constexpr (NA >= 2) {
            // (ordered refills in the last two steps: whatever form the next layer's copy for THIS wave has -- a wave with one
            // tile there takes its fragments one by one, and its waits count loads in issue order)
            b3_stream_steps_direct<END, kb_L1>(ws, ring, [&](auto itc, auto basec, auto&& refill) {
                constexpr int tap = decltype(itc)::value, base = decltype(basec)::value;
                constexpr bool ordered = tap >= 7;
                auto unit = [&](auto ic) {
                    constexpr int i = decltype(ic)::value, u = tap * NA + i;
                    cp_unit_direct<base, tap == 0, i == NA - 1, ordered>(ring, Bb[u & 1], acc[i][0], acc[i][1],
                        [&](int p) {
                            if constexpr (u + 1 < NU)
                                Bb[(u + 1) & 1][p] = *reinterpret_cast<const v4f*>(smem + addr[u + 1] + p * kB3Frag);
                        }, refill);
                };
                [&]<int... I>(std::integer_sequence<int, I...>) { (unit(std::integral_constant<int, I>{}), ...); }(
                    std::make_integer_sequence<int, NA>{});
            }, std::make_integer_sequence<int, 9>{});
            ring_mfma_fence();
        } else {
            b3_stream_steps_spread<END, kb_L1>(ws, ring, [&](auto itc, const v8b (&A)[2][3], auto&& ring_item) {
                constexpr int tap = decltype(itc)::value;
                auto unit = [&](auto ic) {
                    constexpr int i = decltype(ic)::value, u = tap * NA + i;
                    cp_unit(A, Bb[u & 1], acc[i][0], acc[i][1],
                            [&](int p) {
                                if constexpr (u + 1 < NU)
                                    Bb[(u + 1) & 1][p] = *reinterpret_cast<const v4f*>(smem + addr[u + 1] + p * kB3Frag);
                            },
                            [&](auto sc) {
                                constexpr int k = cp_ring_item_at(NA, i, decltype(sc)::value);
                                if constexpr (k >= 0) ring_item(std::integral_constant<int, k>{});
                            });
                };
                [&]<int... I>(std::integer_sequence<int, I...>) { (unit(std::integral_constant<int, I>{}), ...); }(
                    std::make_integer_sequence<int, NA>{});
            }, std::make_integer_sequence<int, 9>{});
        }
    } else {
        b3_stream_steps<END, kb_L1, 2>(ws, ring, [&](auto, const v8b (&)[2][3]) {}, std::make_integer_sequence<int, 9>{});
    }
}

template <int END>
__device__ __forceinline__ void cp_layer1(const WStreamB& ws, v4f (&ring)[kRingH], char* smem, const float* sstab,
                                          const CpGeom& g, int wave, int lane, int tid) {
    constexpr int NT = kCpL1Tiles;
    CpL1Lane ln;
    ln.j = lane & 15;
    ln.q = lane >> 4;
    const int j = ln.j, q = ln.q;
    const int na = min(max((g.T1 - wave + 3) >> 2, 0), NT);        // tiles of this wave (wave-uniform)
    ln.fl = 0;
    int l2dst[NT];                                                  // where the epilogue stores column (tile i, j): L2's input cell
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int c = 16 * (wave + 4 * i) + j;
        const int pos = (int)(__umul24((unsigned)c, g.rcpN) >> 16), n = c - __mul24(pos, g.N);   // (24-bit operands)
        const int y = (pos * 13) >> 6, x = pos - 5 * y;            // pos / 5, pos % 5   (pos < 25 where it matters)
        unsigned f = (y == 0 ? 1u : 0u) | (y == 4 ? 2u : 0u) | (x == 0 ? 4u : 0u) | (x == 4 ? 8u : 0u);
        if (c >= g.ncol) f = 15u;                                   // no such column: every shifted tap reads zeros
        ln.fl |= f << (4 * i);
        l2dst[i] = c < g.ncol ? cp_l2in_cell(y, x, n) + q * 256 : -1;
    }
    ln.zaddr = g.l1in + g.T1 * kCpRow + q * 256;                    // the row of zeros (slot 0 of this lane's q)
    ln.tile0 = g.l1in + wave * kCpRow + q * 256;
    v4f acc[NT][2];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i][0] = acc[i][1] = vzero();
    static_assert(NT == 5, "one case per tile count");
    switch (na) {                                                   // (wave-uniform: a scalar branch)
        case 0: cp_layer1_main<END, 0>(ws, ring, smem, g, ln, wave, acc); break;
        case 1: cp_layer1_main<END, 1>(ws, ring, smem, g, ln, wave, acc); break;
        case 2: cp_layer1_main<END, 2>(ws, ring, smem, g, ln, wave, acc); break;
        case 3: cp_layer1_main<END, 3>(ws, ring, smem, g, ln, wave, acc); break;
        case 4: cp_layer1_main<END, 4>(ws, ring, smem, g, ln, wave, acc); break;
        default: cp_layer1_main<END, 5>(ws, ring, smem, g, ln, wave, acc); break;
    }
    GNNPP_STAMP(blockIdx.x, 8, tid == 0);
    __syncthreads();                                               // everyone is done reading L0's output
    GNNPP_STAMP(blockIdx.x, 9, tid == 0);
    if (tid < kCpRow / 16) *reinterpret_cast<v4f*>(smem + 20 * kCpRow + tid * 16) = vzero();   // L2 input's zero row
    v4f sc[2], shf[2];
    load_ss(sstab + EncLayout::kBssL1, 32, 0, q, sc[0], shf[0]);
    load_ss(sstab + EncLayout::kBssL1, 32, 1, q, sc[1], shf[1]);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if (i < na) {
            v4f pl[3];
            b3_split8_clamped(vrelu_clamp(vfma(acc[i][0], sc[0], shf[0])), vrelu_clamp(vfma(acc[i][1], sc[1], shf[1])), pl);
            if (l2dst[i] >= 0) {
                char* const dst = smem + l2dst[i];
#pragma unroll
                for (int p = 0; p < 3; ++p) *reinterpret_cast<v4f*>(dst + p * kB3Frag) = pl[p];
            }
        }
    }
}

// ---- L2, column-packed: 32 -> 64 @ 5x5 (the 4x4 the pool reads) -> 2x2 max -> L3's input fragments ---------------------
// Wave (mp, pair): channel tiles 2 mp, 2 mp + 1 of the agents pair, pair + 2, ..; one MFMA tile = one agent, column
// j = its output position (window w = j >> 2, position p = j & 3 inside the window).  After BatchNorm the 2x2 max runs
// over the four lanes of a quad as a reduce-scatter (two quad_perm steps): lane p ends with the pooled channel pair p
// of its fragment, splits THAT pair and stores its three dwords -- no lane repeats another lane's conversion.
// Units (tap, agent) are software-pipelined and the agent count is a template argument, as in cp_layer1.
struct CpL2Lane {
    int q, y, x;
};
// LDS address of the planes of this lane's source cell for unit (TAP, agent n)   (n: wave-uniform), in two stages:
// A = the slot inside the row, B = the row of the cell's class (interior / bottom edge / right edge / corner / outside)
template <int TAP>
__device__ __forceinline__ int cp_l2_addr_a(const CpL2Lane& ln, int n) {
    constexpr int dy = TAP / 3 - 1, dx = TAP % 3 - 1;
    const int yy = ln.y + dy, xx = ln.x + dx;                       // source cell of this lane: in -1 .. 4
    const int o = 4 * (n & 3) + (n >> 2);
    return ln.q * 256 + ((4 * (yy & 3) + (xx & 3) + o) & 15) * 16;
}
template <int TAP>
__device__ __forceinline__ int cp_l2_addr_b(const CpL2Lane& ln, int n, int slot) {
    constexpr int dy = TAP / 3 - 1, dx = TAP % 3 - 1;
    const int yy = ln.y + dy, xx = ln.x + dx;
    int row = n;
    if (dy > 0) row = yy == 4 ? 12 + (n >> 2) : row;
    if (dx > 0) row = xx == 4 ? ((dy > 0 && yy == 4) ? 19 : 15 + (n & 3)) : row;
    if (dy < 0) row = yy < 0 ? 20 : row;
    if (dx < 0) row = xx < 0 ? 20 : row;
    return __mul24(row, kCpRow) + slot;                             // (row <= 20)
}
template <int TAP>
__device__ __forceinline__ int cp_l2_addr(const CpL2Lane& ln, int n) {
    return cp_l2_addr_b<TAP>(ln, n, cp_l2_addr_a<TAP>(ln, n));
}
__device__ __forceinline__ void cp_l2_load(const char* smem, int addr, v4f (&B)[3]) {
#pragma unroll
    for (int pp = 0; pp < 3; ++pp) B[pp] = *reinterpret_cast<const v4f*>(smem + addr + pp * kB3Frag);
}
template <int END, int NA>
__device__ __forceinline__ void cp_layer2_main(const WStreamB& ws, v4f (&ring)[kRingH], const char* smem,
                                               const CpL2Lane& ln, int pair, v4f (&acc)[kCpL2Tiles][2]) {
    constexpr int NU = 9 * NA;
    v4f Bb[2][3];
    if constexpr (NA > 0) {
        int addr[NU];                                               // unit u = (tap, agent pair + 2 (u % NA))
        [&]<int... U>(std::integer_sequence<int, U...>) {
            ((addr[U] = cp_l2_addr<U / NA>(ln, pair + 2 * (U % NA)), cp_pin(addr[U])), ...);
        }(std::make_integer_sequence<int, NU>{});
        __builtin_amdgcn_sched_barrier(0);
        GNNPP_STAMP(blockIdx.x, 15, pair == 0 && ln.q == 0 && ln.y == 0 && ln.x == 0);
        cp_l2_load(smem, addr[0], Bb[0]);
        if constexpr (NA >= 2) {
            // (ordered refills in the last two steps: L3 takes its fragments one by one, its waits count loads in issue order)
            b3_stream_steps_direct<END, kb_L2>(ws, ring, [&](auto itc, auto basec, auto&& refill) {
                constexpr int tap = decltype(itc)::value, base = decltype(basec)::value;
                constexpr bool ordered = tap >= 7;
                auto unit = [&](auto ic) {
                    constexpr int i = decltype(ic)::value, u = tap * NA + i;
                    cp_unit_direct<base, tap == 0, i == NA - 1, ordered>(ring, Bb[u & 1], acc[i][0], acc[i][1],
                        [&](int p) {
                            if constexpr (u + 1 < NU)
                                Bb[(u + 1) & 1][p] = *reinterpret_cast<const v4f*>(smem + addr[u + 1] + p * kB3Frag);
                        }, refill);
                };
                [&]<int... I>(std::integer_sequence<int, I...>) { (unit(std::integral_constant<int, I>{}), ...); }(
                    std::make_integer_sequence<int, NA>{});
            }, std::make_integer_sequence<int, 9>{});
            ring_mfma_fence();
        } else {
            b3_stream_steps_spread<END, kb_L2>(ws, ring, [&](auto itc, const v8b (&A)[2][3], auto&& ring_item) {
                constexpr int tap = decltype(itc)::value;
                auto unit = [&](auto ic) {
                    constexpr int i = decltype(ic)::value, u = tap * NA + i;
                    cp_unit(A, Bb[u & 1], acc[i][0], acc[i][1],
                            [&](int p) {
                                if constexpr (u + 1 < NU)
                                    Bb[(u + 1) & 1][p] = *reinterpret_cast<const v4f*>(smem + addr[u + 1] + p * kB3Frag);
                            },
                            [&](auto sc) {
                                constexpr int k = cp_ring_item_at(NA, i, decltype(sc)::value);
                                if constexpr (k >= 0) ring_item(std::integral_constant<int, k>{});
                            });
                };
                [&]<int... I>(std::integer_sequence<int, I...>) { (unit(std::integral_constant<int, I>{}), ...); }(
                    std::make_integer_sequence<int, NA>{});
            }, std::make_integer_sequence<int, 9>{});
        }
    } else {
        b3_stream_steps<END, kb_L2, 2>(ws, ring, [&](auto, const v8b (&)[2][3]) {}, std::make_integer_sequence<int, 9>{});
    }
}

template <int END>
__device__ __forceinline__ void cp_layer2(const WStreamB& ws, v4f (&ring)[kRingH], char* smem, const float* sstab,
                                          const CpGeom& g, int wave, int lane) {
    constexpr int NT = kCpL2Tiles;
    const int mp = wave & 1, pair = wave >> 1;
    const int na = min(max((g.N - pair + 1) >> 1, 0), NT);         // agents of this wave (wave-uniform)
    const int j = lane & 15, q = lane >> 4, w = j >> 2, p = j & 3;
    CpL2Lane ln;
    ln.q = q;
    ln.y = 2 * (w >> 1) + (p >> 1);
    ln.x = 2 * (w & 1) + (p & 1);
    v4f acc[NT][2];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i][0] = acc[i][1] = vzero();
    static_assert(NT == 6, "one case per agent count");
    switch (na) {
        case 0: cp_layer2_main<END, 0>(ws, ring, smem, ln, pair, acc); break;
        case 1: cp_layer2_main<END, 1>(ws, ring, smem, ln, pair, acc); break;
        case 2: cp_layer2_main<END, 2>(ws, ring, smem, ln, pair, acc); break;
        case 3: cp_layer2_main<END, 3>(ws, ring, smem, ln, pair, acc); break;
        case 4: cp_layer2_main<END, 4>(ws, ring, smem, ln, pair, acc); break;
        case 5: cp_layer2_main<END, 5>(ws, ring, smem, ln, pair, acc); break;
        default: cp_layer2_main<END, 6>(ws, ring, smem, ln, pair, acc); break;
    }
    v4f sc[2], shf[2];
    load_ss(sstab + EncLayout::kBssL2, 64, 2 * mp, q, sc[0], shf[0]);
    load_ss(sstab + EncLayout::kBssL2, 64, 2 * mp + 1, q, sc[1], shf[1]);
    const bool hi2 = (p & 2) != 0, hi1 = (p & 1) != 0;
    float pooled[NT][2];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        pooled[i][0] = pooled[i][1] = 0.f;
        if (i < na) {
            const v4f v0 = vfma(acc[i][0], sc[0], shf[0]), v1 = vfma(acc[i][1], sc[1], shf[1]);
            // step 1, partner lane ^ 2: lanes with bit 1 clear keep channel tile 0 (pairs 0, 1), the others tile 1
            v4f keep, send;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                keep[c] = hi2 ? v1[c] : v0[c];
                send[c] = hi2 ? v0[c] : v1[c];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) keep[c] = fmaxf(keep[c], quad_xor2(send[c]));
            // step 2, partner lane ^ 1: bit 0 clear keeps registers 0, 1 (the even pair), set keeps 2, 3
            const float k0 = hi1 ? keep[2] : keep[0], k1 = hi1 ? keep[3] : keep[1];
            const float s0 = hi1 ? keep[0] : keep[2], s1 = hi1 ? keep[1] : keep[3];
            pooled[i][0] = fmaxf(fmaxf(k0, quad_xor1(s0)), 0.f);    // + ReLU (commutes with the max)
            pooled[i][1] = fmaxf(fmaxf(k1, quad_xor1(s1)), 0.f);
        }
    }
    __syncthreads();                                               // everyone is done reading L1's output
    // fragment (window w, block mp, plane) of L3's input, lane slot (q, agent n), dword p = channel pair p
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if (i < na) {
            const int n = pair + 2 * i;
            unsigned h, m, l;
            b3_split2(pooled[i][0], pooled[i][1], h, m, l);
            char* const dst = smem + kB3L2out + ((w * 2 + mp) * 3) * kB3Frag + (q * 16 + n) * 16 + p * 4;
            *reinterpret_cast<unsigned*>(dst) = h;
            *reinterpret_cast<unsigned*>(dst + kB3Frag) = m;
            *reinterpret_cast<unsigned*>(dst + 2 * kB3Frag) = l;
        }
    }
    // the lane slots of agents the graph does not have: zeros (finite values down to the filter tail, whose dense shift
    // multiplies their rows by zero GSO weights)
    if (j >= g.N) {
#pragma unroll
        for (int k = 0; k < 6; ++k)
            *reinterpret_cast<v4f*>(smem + kB3L2out + (wave + 4 * k) * kB3Frag + lane * 16) = vzero();
    }
}

template <bool FUSED, int KT, bool CP = false>
__global__ GNNPP_H2_VGPR_BUDGET __launch_bounds__(kThreads, 2) void encoder_kernel_b3(const float* __restrict__ obs,
                                                                 const float* __restrict__ pk,
                                                                 float* __restrict__ feat, int M, int stop,
                                                                 const PolicyTail pt) {
    constexpr int END = FUSED ? kb_FILT + 24 * KT : kb_END;
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    v4f* const R4 = reinterpret_cast<v4f*>(gnnpp_smem);              // region R as 16-byte fragments' lanes
    unsigned* const obsw = reinterpret_cast<unsigned*>(gnnpp_smem);  // observation words (h | m << 16), l at + kObsFloatsLds
    float* const sstab = reinterpret_cast<float*>(gnnpp_smem + kB3Table);
    unsigned* const planeflag = reinterpret_cast<unsigned*>(gnnpp_smem + kB3Table + EncLayout::kBssFloats * 4);   // [4 waves]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: real branches per wave
    const int a = lane & 15;
    const int q = lane >> 4;
    // FUSED: the tile is graph blockIdx.x (pt.N agents).  <false, K, CP>: the encoder in the LATENCY regime -- tiles of
    // pt.N <= kCpMaxAgents agents, chosen by encoder_launch_b3 so that every tile has a CU to itself (a 16-agent tile
    // alone on a CU takes 31 us however few tiles the launch has; with (agent, position) pairs on the columns of L1 /
    // L2 a tile's work shrinks with its agents).
    const int tile_agents = (FUSED || CP) ? pt.N : kTileAgents;
    const int agent0 = blockIdx.x * tile_agents;
    const int n_agents = FUSED ? pt.N : min(tile_agents, M - agent0);   // (>= 1: the grid is ceil(M / tile))
    const CpGeom geom = cp_geom(CP ? n_agents : kTileAgents);        // (CP only; folded away otherwise)
    GNNPP_STAMP(blockIdx.x, 11, tid == 0);

    WStreamB ws;
    ws.seg[0] = pk + EncLayout::kB1;
    ws.seg[1] = pk + EncLayout::kB2 + (wave & 1) * (54 * EncLayout::kHItem);
    ws.seg[2] = pk + EncLayout::kB3 + wave * (54 * EncLayout::kHItem);
    ws.seg[3] = pk + EncLayout::kB4 + wave * (108 * EncLayout::kHItem);
    ws.seg[4] = pk + EncLayout::kBfcw + wave * (24 * EncLayout::kHItem);
    ws.seg[5] = FUSED ? pt.filt_h2 + wave * (2 * 4 * 768) : pk;     // (pt.filt_h2: the b3 region of the filter pack)
    ws.lane_bytes = lane * 16;
    // BatchNorm scale/shift of all five layers: fetched now, parked in the table behind R, so that no
    // compiler-issued global load (whose wait would drain the ring) sits between the layers
    float ssv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        ssv[i] = pk[EncLayout::kBss + min(tid + i * kThreads, EncLayout::kBssFloats - 1)];
    // FUSED: what the epilogue of the filter reads -- act_w [5][128] | bias [128] | act_b [5] -- and the GSO take the
    // same road, but the table has no room for them before L4 is done: they wait in five registers
    float hcv[4] = {0.f, 0.f, 0.f, 0.f};
    float sval = 0.f;
    if (FUSED) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * kThreads;
            if (c < 640) hcv[i] = pt.act_w[c];
            else if (c < 768) hcv[i] = pt.gf_bias ? pt.gf_bias[c - 640] : 0.f;
            else if (c < 773) hcv[i] = pt.act_b[c - 768];
        }
        const int m = tid >> 4, n = tid & 15;                        // one GSO entry per thread
        if (m < pt.N && n < pt.N) {
            const size_t i = ((size_t)blockIdx.x * pt.N + m) * pt.N + n;
            sval = pt.s_is_f64 ? (float)reinterpret_cast<const double*>(pt.S)[i]
                               : reinterpret_cast<const float*>(pt.S)[i];
        }
    }
    v4f ring[kRingH];
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("; weight ring lives in v[208:255]" ::: "v255");   // (see encoder_kernel_h2.hip)
#endif
#pragma unroll
    for (int i = 0; i < kRingH; ++i) h2_ring_load<END>(ws, ring, i);

    // ---- observations: all loads first, zero-fill while they fly, then split + scatter -------------------
    {
        constexpr int NV4 = (kTileAgents * kObsFloats + 3) / 4 + 1;
        constexpr int PER = (NV4 + kThreads - 1) / kThreads;
        const int valid = n_agents * kObsFloats;
        const float* src = obs + (size_t)agent0 * kObsFloats;
        const int shift = (int)((reinterpret_cast<uintptr_t>(src) >> 2) & 3);
        const float* src4 = src - shift;
        const long floats_left = (long)(M - agent0) * kObsFloats + shift;   // from src4 to the end of the tensor
        const bool head_ok = shift == 0 || agent0 > 0;                      // (never read in front of the tensor)
        v4f v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int idx = tid + k * kThreads;
            const int e0 = 4 * idx - shift;
            if (head_ok && 4L * idx + 4 <= floats_left && e0 < valid) {
                v[k] = *reinterpret_cast<const v4f*>(src4 + 4 * idx);
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[k][c] = src[min(max(e0 + c, 0), valid - 1)];
            }
        }
        for (int i = tid; i < 2 * kObsFloatsLds / 4; i += kThreads) R4[i] = vzero();
        if (CP && tid < kCpRow / 16)                       // the row of zeros behind L1's input rows
            *reinterpret_cast<v4f*>(gnnpp_smem + geom.l1in + geom.T1 * kCpRow + tid * 16) = vzero();
        unsigned residual = 0;                             // any non-zero m / l plane among this thread's pixels
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (tid + i * kThreads < EncLayout::kBssFloats) sstab[tid + i * kThreads] = ssv[i];
        __syncthreads();
        // Scatter into the padded images, BRANCH-FREE (this loop is VALU-issue-bound and runs in both workgroups of a
        // CU at the same time: every instruction counts).  Flat element e of [agent][3][11][11] -> word
        //   o = agent * kAgentStride + ch * 144 + (y + 1) * 12 + x + 1 = base + r + y + 13,   r = 11 y + x in 0..120;
        // a thread's four consecutive elements cross at most ONE channel / agent boundary (r wraps at 121).  Divisions
        // by constants as 24-bit multiplies + shifts (exact on the ranges used); elements outside the tile go to a
        // dump word behind the images instead of around a branch.
        const unsigned dump = 2 * kObsFloatsLds + lane;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e0 = (tid + k * kThreads) * 4 - shift;
            const unsigned es = (unsigned)max(e0, 0);
            const unsigned ag = __umul24(es, 46223u) >> 24;                   // es / 363   (es < 5816)
            const unsigned rem = es - __umul24(ag, (unsigned)kObsFloats);
            const unsigned ch = __umul24(rem, 543u) >> 16;                    // rem / 121  (rem < 363)
            const int r0 = (int)(rem - __umul24(ch, 121u)) + min(e0, 0);        // (e0 < 0: the first thread of an unaligned tile)
            const unsigned base = __umul24(ag, (unsigned)kAgentStride) + __umul24(ch, (unsigned)(kPadHW * kPadHW)) + 13u;
            const unsigned wrapped = base + (ch == 2u ? (unsigned)(kAgentStride - 2 * kPadHW * kPadHW) : (unsigned)(kPadHW * kPadHW));
            unsigned w1[4], w2[4];
            const v4u vb = __builtin_bit_cast(v4u, v[k]);
            const bool inexact = ((vb[0] | vb[1] | vb[2] | vb[3]) & 0xffffu) != 0u;   // some value of this lane is not ONE bf16
            const bool any_l = __ballot(inexact) != 0ull;               // (wave-uniform)
            if (!any_l) {
                // every value IS its h plane (the simulator's {0, 1} observations): no conversions, no l words
                w1[0] = vb[0] >> 16; w1[1] = vb[1] >> 16; w1[2] = vb[2] >> 16; w1[3] = vb[3] >> 16;
            } else {
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    unsigned h, m, l;
                    b3_split2(v[k][2 * c2], v[k][2 * c2 + 1], h, m, l);
                    // (a lane's elements outside the tile are copies of valid ones or other agents' pixels: they can
                    // only make the flag conservative)
                    residual |= m | l;
                    w1[2 * c2] = __builtin_amdgcn_perm(m, h, 0x05040100u);        // (h | m << 16) of the even element
                    w1[2 * c2 + 1] = __builtin_amdgcn_perm(m, h, 0x07060302u);    // ... of the odd element
                    w2[2 * c2] = l & 0xffffu;
                    w2[2 * c2 + 1] = l >> 16;
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int e = e0 + c;
                int r = r0 + c;
                const bool wrap = r >= 121;
                r = wrap ? r - 121 : r;
                const unsigned y = __umul24((unsigned)r & 0xffu, 745u) >> 13;         // r / 11   (0 <= r < 121 where it matters)
                const unsigned o = (wrap ? wrapped : base) + (unsigned)r + y;
                const bool ok = e >= 0 && e < valid;
                obsw[ok ? o : dump] = w1[c];
                if (any_l) obsw[ok ? o + kObsFloatsLds : dump] = w2[c];
            }
        }
        // PLANE SKIPPING: when every pixel of the tile is exactly one bf16 plane (the simulator's observations are
        // {0, 1}: AgentState.toInputTensor, dataloader/statetransformer.py:82-130) the m and l planes are identically
        // zero and L0 does not issue their products -- skipping exact zeros, the result is the same to the bit.
        const bool wave_residual = __ballot(residual != 0) != 0ull;
        if (lane == 0) planeflag[wave] = wave_residual ? 1u : 0u;
    }
    __syncthreads();
    if (GNNPP_STOP_AT(stop, 1)) return;
    if (!pt.with_sim) GNNPP_STAMP(blockIdx.x, 0, tid == 0);

    // ---- L0: 3 -> 32 @ 11x11 (the 10x10 the pool reads), direct, K = 27 in ONE 32-slot block -------
    // Lane (q, agent) owns k-slots (q, e) as in the split-f16 schedule.  A wave walks its windows' positions as ONE
    // software-pipelined stream (b3_l0_stream below): the pixel words of position P + DEPTH are requested before the
    // MFMAs of position P are issued, the BatchNorm / ReLU / max epilogue of position P - 1 runs beside them.  The
    // pooled outputs stay in registers until every wave is done with the pixels.
    {
        const v4u pf = *reinterpret_cast<const v4u*>(planeflag);
        const bool one_plane = (pf[0] | pf[1] | pf[2] | pf[3]) == 0;     // (workgroup-uniform)
        if (one_plane) {
            // Window win's output fragments live at R + 3 KiB * win; the h | m pixel words occupy the first 27.1 KiB
            // of R (the l words behind them are not read on this path): a window beyond them is written as soon as
            // it is pooled, only a wave's first three windows wait in registers for the barrier.
            v4f res[3][2];
            b3_l0_stream<true, CP>(pk, obsw, sstab, gnnpp_smem, geom, res, wave, lane);
            __syncthreads();                                 // every wave has read its last pixel
#pragma unroll
            for (int wi = 0; wi < 3; ++wi) {
                const int win = wave + 4 * wi;
                v4f pl[3];
                b3_split8_clamped(res[wi][0], res[wi][1], pl);
                b3_store_window<CP>(gnnpp_smem, geom, win, lane, pl);
            }
        } else {
            b3_l0_generic<CP>(pk, obsw, sstab, gnnpp_smem, geom, wave, lane);
        }
    }
    __syncthreads();
    if (GNNPP_STOP_AT(stop, 2)) return;
    if (!pt.with_sim) GNNPP_STAMP(blockIdx.x, 1, tid == 0);

    // ---- L1: 32 -> 32 @ 5x5, in place; wave = its positions x both channel tiles ---------------------
    if constexpr (CP) {
        cp_layer1<END>(ws, ring, gnnpp_smem, sstab, geom, wave, lane, tid);
    } else {
        v4f acc[7][2];                                    // first touched by a zero-source MFMA (b3_mfma_chunk)
        constexpr int CH1 = 2;                            // slots per chunk: 6 LDS reads in flight beside <= 24 MFMAs
        v4f Bb[2][CH1][3];
        // The per-wave position sets are compile-time types, and the whole layer (ring traffic included) sits inside
        // the wave's case: one straight-line region per wave, accumulators updated in place.  (A branch per STEP,
        // as in the split-f16 kernel, made every accumulator and both plane sets a phi at every step's join; next to
        // the ring's 48 reserved registers the allocator answered with spills.)  Every case performs the same ring
        // sequence, which tools/check_ring_isa.py verifies path by path.
        auto layer1 = [&](auto wc) {
            using P = PosL1H<decltype(wc)::value>;
            b3_load_chunk<0, 0, 1, 5, 5, 7, CH1, P>(R4, Bb[0], lane);
            b3_stream_steps<END, kb_L1, 2>(ws, ring, [&](auto itc, const v8b (&A)[2][3]) {
                b3_step_pipelined<decltype(itc)::value, 9, 1, 5, 5, 2, 7, CH1, P>(R4, Bb, A, acc, lane);
            }, std::make_integer_sequence<int, 9>{});
        };
        switch (wave) {
            case 0: layer1(std::integral_constant<int, 0>{}); break;
            case 1: layer1(std::integral_constant<int, 1>{}); break;
            case 2: layer1(std::integral_constant<int, 2>{}); break;
            default: layer1(std::integral_constant<int, 3>{}); break;
        }
        __syncthreads();                                   // everyone is done reading L0's output
        v4f sc[2], sh[2];
        load_ss(sstab + EncLayout::kBssL1, 32, 0, q, sc[0], sh[0]);
        load_ss(sstab + EncLayout::kBssL1, 32, 1, q, sc[1], sh[1]);
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int p = wave + 4 * j;
            if (p < 25) {
                v4f pl[3];
                b3_split8_clamped(vrelu_clamp(vfma(acc[j][0], sc[0], sh[0])), vrelu_clamp(vfma(acc[j][1], sc[1], sh[1])), pl);
#pragma unroll
                for (int pp = 0; pp < 3; ++pp) R4[(p * 3 + pp) * 64 + lane] = pl[pp];
            }
        }
    }
    __syncthreads();
    if (GNNPP_STOP_AT(stop, 3)) return;
    if (!pt.with_sim) GNNPP_STAMP(blockIdx.x, 2, tid == 0);

    // ---- L2: 32 -> 64 @ 5x5 (the 4x4 the pool reads), pool -> [4][kb 2], in place (front of R) --------
    if constexpr (CP) {
        cp_layer2<END>(ws, ring, gnnpp_smem, sstab, geom, wave, lane);
    } else {
        const int mp = wave & 1, pair = wave >> 1;         // channel tiles 2 mp, 2 mp + 1 = block mp
        v4f acc[8][2];                                    // first touched by a zero-source MFMA (b3_mfma_chunk)
        constexpr int CH2 = 2;
        v4f Bb[2][CH2][3];
        auto layer2 = [&](auto pc) {
            using P = PosL2H<decltype(pc)::value>;
            b3_load_chunk<0, 0, 1, 5, 5, 8, CH2, P>(R4, Bb[0], lane);
            b3_stream_steps<END, kb_L2, 2>(ws, ring, [&](auto itc, const v8b (&A)[2][3]) {
                b3_step_pipelined<decltype(itc)::value, 9, 1, 5, 5, 2, 8, CH2, P>(R4, Bb, A, acc, lane);
            }, std::make_integer_sequence<int, 9>{});
        };
        switch (pair) {
            case 0: layer2(std::integral_constant<int, 0>{}); break;
            default: layer2(std::integral_constant<int, 1>{}); break;
        }
        v4f sc[2], sh[2];
        load_ss(sstab + EncLayout::kBssL2, 64, 2 * mp, q, sc[0], sh[0]);
        load_ss(sstab + EncLayout::kBssL2, 64, 2 * mp + 1, q, sc[1], sh[1]);
        v4f r[2][2];
#pragma unroll
        for (int wi = 0; wi < 2; ++wi)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                r[wi][m] = vrelu(vfma(acc[4 * wi][m], sc[m], sh[m]));
#pragma unroll
                for (int pp = 1; pp < 4; ++pp)
                    r[wi][m] = vmax(r[wi][m], vfma(acc[4 * wi + pp][m], sc[m], sh[m]));
            }
        __syncthreads();                                   // everyone is done reading L1's output
        v4f* const O4 = reinterpret_cast<v4f*>(gnnpp_smem + kB3L2out);
#pragma unroll
        for (int wi = 0; wi < 2; ++wi) {
            const int t = wi == 0 ? pair : 3 - pair;
            v4f pl[3];
            b3_split8(r[wi][0], r[wi][1], pl);
#pragma unroll
            for (int p = 0; p < 3; ++p) O4[((t * 2 + mp) * 3 + p) * 64 + lane] = pl[p];
        }
    }
    __syncthreads();
    if (GNNPP_STOP_AT(stop, 4)) return;
    if (!pt.with_sim) GNNPP_STAMP(blockIdx.x, 3, tid == 0);

    // ---- L3: 64 -> 64 @ 2x2, one channel tile per wave, input held in registers ----------------------
    {
        const int mt = wave;
        v4f acc[4][1];
        b3_conv_preload<END, kb_L3, 2, 2, 2, 1, 4, Pos2x2H>(
            ws, ring, reinterpret_cast<const v4f*>(gnnpp_smem + kB3L2out), acc, lane);
        v4f sc, sh;
        load_ss(sstab + EncLayout::kBssL3, 64, mt, q, sc, sh);
        v2f* const O2 = reinterpret_cast<v2f*>(gnnpp_smem + kB3L3out);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v2f pl[3];
            b3_split4(vrelu(vfma(acc[j][0], sc, sh)), pl);
            // fragment (pos j, block mt >> 1): this tile is its e = 4 (mt & 1) .. +3 half
            const int o = (j * 2 + (mt >> 1)) * 3;
#pragma unroll
            for (int p = 0; p < 3; ++p) O2[((o + p) * 64 + lane) * 2 + (mt & 1)] = pl[p];
        }
    }
    __syncthreads();
    if (GNNPP_STOP_AT(stop, 5)) return;
    if (!pt.with_sim) GNNPP_STAMP(blockIdx.x, 4, tid == 0);

    // ---- L4: 64 -> 128 @ 2x2, pool -> [1][kb 4], tiles 2w, 2w+1 per wave ------------------------------
    {
        v4f acc[4][2];
        b3_conv_preload<END, kb_L4, 2, 2, 2, 2, 4, Pos2x2H>(
            ws, ring, reinterpret_cast<const v4f*>(gnnpp_smem + kB3L3out), acc, lane);
        v4f sc[2], sh[2];
        load_ss(sstab + EncLayout::kBssL4, 128, 2 * wave, q, sc[0], sh[0]);
        load_ss(sstab + EncLayout::kBssL4, 128, 2 * wave + 1, q, sc[1], sh[1]);
        v4f r[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            r[m] = vrelu(vfma(acc[0][m], sc[m], sh[m]));
#pragma unroll
            for (int j = 1; j < 4; ++j) r[m] = vmax(r[m], vfma(acc[j][m], sc[m], sh[m]));
        }
        v4f pl[3];
        b3_split8(r[0], r[1], pl);
        v4f* const O4 = reinterpret_cast<v4f*>(gnnpp_smem + kB3L4out);
#pragma unroll
        for (int p = 0; p < 3; ++p) O4[(wave * 3 + p) * 64 + lane] = pl[p];
    }
    __syncthreads();
    if (GNNPP_STOP_AT(stop, 6)) return;
    if (!pt.with_sim) GNNPP_STAMP(blockIdx.x, 5, tid == 0);
    float* const hconst = sstab;                                     // FUSED: the table now holds the head's constants
    float* const Ssm = reinterpret_cast<float*>(gnnpp_smem + kB3Table + kB3SsmOff);   // GSO [16][17], zero padded
    if (FUSED) {                                                     // (the BatchNorm table is dead)
        Ssm[(tid >> 4) * 17 + (tid & 15)] = sval;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (tid + i * kThreads < 773) hconst[tid + i * kThreads] = hcv[i];
    }

    // ---- FC 128 -> 128 + ReLU -> feat[agent][128]; tiles 2w, 2w+1 per wave ------------------------
    float* const zA = reinterpret_cast<float*>(gnnpp_smem);          // FUSED: z ping-pong (fp32 rows)
    {
        const v4f* const I4 = reinterpret_cast<const v4f*>(gnnpp_smem + kB3L4out);
        v8b B[4][3];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int p = 0; p < 3; ++p) B[kb][p] = as_b8(I4[(kb * 3 + p) * 64 + lane]);
        v4f acc[2][2] = {{vzero(), vzero()}, {vzero(), vzero()}};    // [tile][hh | the five smaller terms]
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            __builtin_amdgcn_sched_barrier(kSchedItemMask);
            v8b A[2][3];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const int idx = kb_FC + (kb * 2 + m) * 3 + p;
                    A[m][p] = as_b8(ring_take_f4<END>(ring, idx));
                    h2_ring_load<END>(ws, ring, idx + kRingH);
                }
#pragma unroll
            for (int term = 0; term < kB3Terms; ++term)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    acc[m][term == kB3Terms - 1 ? 0 : 1] =
                        mfma16b(A[m][b3_term_a(term)], B[kb][b3_term_b(term)], acc[m][term == kB3Terms - 1 ? 0 : 1]);
        }
        if (FUSED) {
            // z_0 = the features of the graph's agents: fp32 rows (the first shift reads them) AND bf16x3 planes (tap
            // 0's MFMA operand) while they are in registers; lane (q, a) holds channels 16 mt + 4 q .. + 3 of agent a
            char* const P0 = gnnpp_smem + kB3POff;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int mt = 2 * wave + m;
                const v4f b = *reinterpret_cast<const v4f*>(pk + EncLayout::kBfc + mt * 16 + q * 4);
                const v4f f = vrelu((acc[m][0] + acc[m][1]) + b);
                if (KT > 1) *reinterpret_cast<v4f*>(zA + a * kZs + mt * 16 + q * 4) = f;
                v2f pl[3];
                b3_split4(f, pl);
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    *reinterpret_cast<v2f*>(P0 + a * kB3PRow + p * 256 + (mt * 16 + q * 4) * 2) = pl[p];
            }
        } else if (a < n_agents) {                                  // (agent0 + a < M; lanes beyond a CP tile: other tiles' agents)
            float* dst = feat + (size_t)(agent0 + a) * 128 + q * 4;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int mt = 2 * wave + m;
                const v4f b = *reinterpret_cast<const v4f*>(pk + EncLayout::kBfc + mt * 16 + q * 4);
                *reinterpret_cast<v4f*>(dst + mt * 16) = vrelu((acc[m][0] + acc[m][1]) + b);
            }
        }
    }
    if (!FUSED) {
        GNNPP_STAMP(blockIdx.x, 6, tid == 0);
        return;
    }

    // ==== graph filter + action head of this graph (K = KT taps, G = F = 128) =====================
    // z_k = z_{k-1} S as a dense product on the fp32 MFMA, all m in ascending order: bit-identical to
    // lsigf_kernel's sparse gather (fmaf(0, z, acc) == acc).  The shift's epilogue writes z_k as fp32 rows (if
    // another shift follows) and as bf16x3 planes.
    __syncthreads();                                     // z_0 complete (both forms); head constants and GSO visible
    GNNPP_STAMP(blockIdx.x, 12, tid == 0);
#pragma unroll
    for (int k = 1; k < KT; ++k) {
        const float* zp = zA + ((k - 1) & 1) * (16 * kZs);
        float* zn = zA + (k & 1) * (16 * kZs);
        char* const Pk = gnnpp_smem + kB3POff + k * kB3PBytes;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ft = 2 * wave + t;
            v4f d = vzero();
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int m = 4 * st + q;
                d = mfma16(zp[m * kZs + 16 * ft + a], Ssm[m * 17 + a], d);   // A[i = feature][k = m], B[k = m][j = node]
            }
            if (k + 1 < KT) *reinterpret_cast<v4f*>(zn + a * kZs + 16 * ft + 4 * q) = d;   // node a, features 16 ft + 4 q ..
            v2f pl[3];
            b3_split4(d, pl);
#pragma unroll
            for (int p = 0; p < 3; ++p)
                *reinterpret_cast<v2f*>(Pk + a * kB3PRow + p * 256 + (16 * ft + 4 * q) * 2) = pl[p];
        }
        __syncthreads();
    }
    GNNPP_STAMP(blockIdx.x, 13, tid == 0);
    // contraction on the bf16 pipe: channel tiles 2w, 2w+1; the hh products in one accumulator, the five smaller
    // terms in another
    v4f fa[2] = {vzero(), vzero()}, fc[2] = {vzero(), vzero()};
#pragma unroll
    for (int tap = 0; tap < KT; ++tap) {
        const char* zr = gnnpp_smem + kB3POff + tap * kB3PBytes + a * kB3PRow + q * 16;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            __builtin_amdgcn_sched_barrier(kSchedItemMask);
            v8b A[2][3];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const int idx = kb_FILT + ((tap * 4 + kb) * 2 + m) * 3 + p;
                    A[m][p] = as_b8(ring_take_f4<END>(ring, idx));
                    h2_ring_load<END>(ws, ring, idx + kRingH);
                }
            v8b B[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) B[p] = as_b8(*reinterpret_cast<const v4f*>(zr + p * 256 + kb * 64));
#pragma unroll
            for (int term = 0; term < kB3Terms; ++term)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    if (term == kB3Terms - 1) fa[m] = mfma16b(A[m][0], B[0], fa[m]);
                    else fc[m] = mfma16b(A[m][b3_term_a(term)], B[b3_term_b(term)], fc[m]);
                }
        }
    }
    GNNPP_STAMP(blockIdx.x, 14, tid == 0);
    // bias + ReLU in registers, then the 128 -> 5 action head on the fp32 MFMA where the accumulators are (as
    // encoder_kernel_h2<true, K>): each wave multiplies its two tiles, the four partial logits meet in LDS.
    float* const yb = reinterpret_cast<float*>(gnnpp_smem + kB3POff + KT * kB3PBytes);   // [4 waves][16 nodes][8]
    {
        v4f d = vzero();
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int f0 = (2 * wave + m) * 16 + 4 * q;
            const v4f bv = *reinterpret_cast<const v4f*>(hconst + 640 + f0);
            const v4f A5 = a < 5 ? *reinterpret_cast<const v4f*>(hconst + a * 128 + f0) : vzero();
            d = mfma16x4(A5, vrelu((fa[m] + fc[m]) + bv), d);
        }
        if (q < 2) *reinterpret_cast<v4f*>(yb + (wave * 16 + a) * 8 + 4 * q) = d;   // outputs 4 q + reg of node a
    }
    __syncthreads();
    // simulator state (with_sim): where the fp32 z rows were -- every reader of those passed the barrier above
    int* const spos = reinterpret_cast<int*>(gnnpp_smem);
    if (wave == 0) {
        v4f d = *reinterpret_cast<const v4f*>(yb + a * 8 + 4 * (q & 1));
#pragma unroll
        for (int w = 1; w < kWaves; ++w) d += *reinterpret_cast<const v4f*>(yb + (w * 16 + a) * 8 + 4 * (q & 1));
        const v4f ab4 = *reinterpret_cast<const v4f*>(hconst + 768 + 4 * (q & 1));   // act_b[0..3] | act_b[4], .
        if (a < pt.N && q < 2) {                          // lane holds node a, outputs 4 q + reg
            float* dst = pt.logits + ((size_t)a * pt.B + blockIdx.x) * 5;
            // with the simulator tail: a copy [N][5] in LDS for this wave's move (red + 2 kMaxAgents, see below)
            float* lds = reinterpret_cast<float*>(spos) + 4 * kMaxAgents + a * 5;
            if (q == 0) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    dst[t] = d[t] + ab4[t];
                    if (pt.with_sim) lds[t] = d[t] + ab4[t];
                }
            } else {
                dst[4] = d[0] + ab4[0];
                if (pt.with_sim) lds[4] = d[0] + ab4[0];
            }
        }
        __builtin_amdgcn_wave_barrier();                  // (the LDS copy precedes this wave's reads in move_body)
    }
    if (!pt.with_sim) return;

    // ==== simulator step of this episode (same code as rollout_step_kernel; see encoder_kernel_h2.hip) ==========
    {
        int* red = spos + 2 * kMaxAgents;
        int* goal_l = red + 4 * kMaxAgents;
        char* gso_smem = reinterpret_cast<char*>(goal_l + 2 * kMaxAgents);
        unsigned char* occ = reinterpret_cast<unsigned char*>(gso_smem + kGsoSmemBytes);
        const int b = blockIdx.x;
        GNNPP_STAMP(b, 10, tid == 0);
        const size_t occ_bytes = ((size_t)pt.sim.H * pt.sim.W + 15) & ~(size_t)15;
        unsigned* cellcnt = 2 * occ_bytes <= policy_sim_occ_bytes_b3() ? reinterpret_cast<unsigned*>(occ + occ_bytes)
                                                                       : nullptr;
        sim_tail(pt.sim, b, spos, red, goal_l, gso_smem, occ, tid, kThreads, cellcnt,
                 reinterpret_cast<const float*>(red + 2 * kMaxAgents));
    }
}

std::atomic<int> g_policy_column_packing{1};   // GNNPP_TUNE_POLICY_CP: 0 = agents on the MFMA columns for every team size
std::atomic<int> g_encoder_cp_tile{0};   // GNNPP_TUNE_ENCODER_CP_TILE: 0 = heuristic, 1 .. 12 = agents per tile, 16 = never

// Few agents (M <= 256 x 8): column-packed tiles of ceil(M / 256) agents, one per CU (latency regime: the per-GPU shards
// of the 8-GPU configs, 1 600 agents); otherwise 16-agent tiles, two per CU (throughput regime).  Measured
// (profiles/r04_cp_tiles.jsonl, a workgroup alone on its CU): M = 640: 24.1 us (tiles of 3) against 30.3 (16-agent
// tiles); M = 1 600: 28.0 (7) against 30.4; tiles of 10 or more agents do not pay (30.6 / 32.5 us at 10 / 12: a lone
// wave runs the packed layers at ~45 % of the pipe's rate, and their weight stream does not shrink with the tile).
static int encoder_cp_tile(int M) {
    const int knob = g_encoder_cp_tile.load(std::memory_order_relaxed);
    if (knob >= 1 && knob <= kCpMaxAgents) return knob;
    if (knob != 0 || g_policy_column_packing.load(std::memory_order_relaxed) == 0) return 0;
    const int t = (M + 255) / 256;
    return t <= 8 ? t : 0;
}

int encoder_launch_b3(const float* obs, const float* packed, float* feat, int M, hipStream_t st) {
    if (const int tile = encoder_cp_tile(M)) {
        static LdsAttrOnce once_cp;
        set_lds_attr_once(once_cp, reinterpret_cast<const void*>(&encoder_kernel_b3<false, 3, true>), (int)kB3Smem);
        PolicyTail pt{};
        pt.N = tile;
        hipLaunchKernelGGL((encoder_kernel_b3<false, 3, true>), dim3((M + tile - 1) / tile), dim3(kThreads), kB3Smem, st,
                           obs, packed, feat, M, GNNPP_ENCODER_STOP_VALUE, pt);
        return hipGetLastError() == hipSuccess ? 0 : -3;
    }
    static LdsAttrOnce once;
    set_lds_attr_once(once, reinterpret_cast<const void*>(&encoder_kernel_b3<false, 3>), (int)kB3Smem);
    const int grid = (M + kTileAgents - 1) / kTileAgents;
    hipLaunchKernelGGL((encoder_kernel_b3<false, 3>), dim3(grid), dim3(kThreads), kB3Smem, st, obs, packed, feat, M,
                       GNNPP_ENCODER_STOP_VALUE, PolicyTail{});   // (zero-initialised: unused by <false>)
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// whole policy step of B graphs with N <= 16 agents and K = 2, 3 or 4 taps: one workgroup per graph
template <int KT, bool CP>
static int policy_launch_fused_b3_k(const float* obs, const float* packed, const PolicyTail& pt, hipStream_t st) {
    static LdsAttrOnce once;
    set_lds_attr_once(once, reinterpret_cast<const void*>(&encoder_kernel_b3<true, KT, CP>), (int)kB3SmemFused);
    hipLaunchKernelGGL((encoder_kernel_b3<true, KT, CP>), dim3(pt.B), dim3(kThreads), kB3SmemFused, st, obs, packed,
                       static_cast<float*>(nullptr), pt.B * pt.N, 0, pt);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// pt.filt_h2 must point at the b3 region of the filter pack.  Teams of <= kCpMaxAgents agents take the column-packed
// front half (same logits to the bit: tests/test_gpu_parity.py::test_column_packed_policy_kernel_is_bit_identical).
int policy_launch_fused_b3(const float* obs, const float* packed, const PolicyTail& pt, int K, hipStream_t st) {
    const bool cp = pt.N <= kCpMaxAgents && g_policy_column_packing.load(std::memory_order_relaxed) != 0;
    switch (K) {
        case 2: return cp ? policy_launch_fused_b3_k<2, true>(obs, packed, pt, st)
                          : policy_launch_fused_b3_k<2, false>(obs, packed, pt, st);
        case 3: return cp ? policy_launch_fused_b3_k<3, true>(obs, packed, pt, st)
                          : policy_launch_fused_b3_k<3, false>(obs, packed, pt, st);
        case 4: return cp ? policy_launch_fused_b3_k<4, true>(obs, packed, pt, st)
                          : policy_launch_fused_b3_k<4, false>(obs, packed, pt, st);
        default: return -2;
    }
}

}  // namespace gnnpp
