// Encoder kernel, split-f16 schedule ("h2").  Same layers, same 16-agent tile and the same LDS
// budget as the exact-fp32 schedule (encoder_kernel_f32.hip); every layer runs on the f16 matrix pipe with each fp32
// operand split in two halves:
//
//      x = xh + xl,  xh = f16(x),  xl = f16(x - xh)          (x - xh is exact in fp32)
//      w x ~= wh xh + wh xl + wl xh                           (fp32 accumulate; wl xl ~ 2^-22 dropped)
//
// Three v_mfma_f32_16x16x32_f16 (K = 32 channels, ~17 cycles each) replace eight
// v_mfma_f32_16x16x4_f32 (K = 4, 32 cycles each): ~5x less matrix-pipe time per MAC, with the
// operands still carrying 22 mantissa bits.  Measured against the fp32 oracle: |dlogit| <= 9e-8 on
// the BASELINE configs (tolerance 1e-4), i.e. the summation-order noise of the fp32 schedules, which
// stay selectable (gnnpp_set_tuning).
// Domain: |activation| < 65504 (f16 range; a larger value becomes inf and is not silent).
// Weights are pre-scaled per layer by a power of two so that their lo halves are normal numbers;
// f16 subnormals (small lo halves of activations) are kept by the cvt and by the MFMA
// (tools/probe/f16_probe.hip), which bounds the representation error of an activation by
// max(2^-22 |x|, 2^-25).
//
// Fragment order for K = 32: k-slot (q, e) of block kb is channel 32 kb + 16 (e >> 2) + 4 q + (e & 3),
// so the two D tiles (mt = 2 kb, 2 kb + 1) a lane holds after a layer ARE its B fragment of block kb
// for the next layer (eight values -> one 16-byte hi and one 16-byte lo store, lane for lane).
// Activations in LDS: [position][kb][hi/lo][lane 64] x 16 bytes -- 4 bytes per element, as fp32.
//
// With the matrix pipe 5x cheaper the kernel is shaped by LDS and L2 bandwidth instead:
//   * L1/L2: a wave computes TWO channel tiles for its positions, so a B fragment read from LDS
//     (2 KiB per position, tap) feeds six MFMAs; positions are split over the waves
//     (L1: 7/6/6/6 positions, L2: two pool windows each, 61/60 valid taps -- balanced).
//   * L3/L4/FC (2x2 and 1x1 images): a wave reads its whole input (64 VGPRs) once and streams
//     weights only; these layers are bound by the weight stream out of L2 (0.5 MB per tile).
//   * weights are packed in each wave's consumption order and stream through a 12-deep ring of
//     16-byte loads kept in registers the compiler does not see (h2_ring_load / h2_ring_take below).
//   * L0 (K = 27) is one 32-slot block on observations staged as (hi, lo) half pairs.
//
// Two instantiations: <false> = the encoder (16-agent tiles, features to HBM); <true> = the fused
// policy kernel for teams of N <= 16 agents: one workgroup per graph, which after the FC runs that
// graph's K = 3 graph filter and the action head on the features in LDS (PolicyTail below).
#include <utility>

#include "gnnpp_common.h"

namespace gnnpp {

constexpr int kRingH = 12;
constexpr int kZs = 136;               // fused policy tail: row stride of the z / y rows in LDS (floats)
// per-wave item stream: L1 (36) | L2 (36) | L3 (36) | L4 (72) | FC (16); one item = one 16-byte
// hi or lo fragment of one (kb, tap, mt)
constexpr int kh_L1 = 0, kh_L2 = 36, kh_L3 = 72, kh_L4 = 108, kh_FC = 180, kh_END = 196;
// fused policy kernel (one graph per workgroup): the graph filter's split-f16 taps follow,
// [tap K][kb 4][mt_local 2][hi/lo] = 16 K items, read in place from gnnpp_filter_pack's buffer
constexpr int kh_FILT = kh_END;
constexpr int kPolicyTapsMin = 2, kPolicyTapsMax = 4;     // K of the instantiated fused kernels

struct WStreamH {                 // per-wave segment bases (wave-uniform: SGPRs) + this lane's offset
    const float* seg[6];
    int lane_bytes;               // lane * 16: the lane's 16 bytes inside every 1 KiB fragment
};

__device__ __forceinline__ const float* h2_item_ptr(const WStreamH& ws, int idx) {
    if (idx < kh_L2) return ws.seg[0] + (idx - kh_L1) * EncLayout::kHItem;
    if (idx < kh_L3) return ws.seg[1] + (idx - kh_L2) * EncLayout::kHItem;
    if (idx < kh_L4) return ws.seg[2] + (idx - kh_L3) * EncLayout::kHItem;
    if (idx < kh_FC) return ws.seg[3] + (idx - kh_L4) * EncLayout::kHItem;
    if (idx < kh_FILT) return ws.seg[4] + (idx - kh_FC) * EncLayout::kHItem;
    // filter block (tap, mt, kb) of gnnpp_filter_pack: ((tap * 8 + mt) * 4 + kb) * 512 floats, lo at +256;
    // seg[5] already points at this wave's first channel tile
    const int j = idx - kh_FILT, hl = j & 1, ml = (j >> 1) & 1, kb = (j >> 2) & 3, tap = j >> 4;
    return ws.seg[5] + tap * (8 * 4 * 512) + ml * (4 * 512) + kb * 512 + hl * 256;
}

// The weight stream is what bounds this kernel (tools/probe/wstream_probe.hip: a CU pulls ~70 GB/s
// out of L2 with 8-byte loads, ~125 GB/s with 16-byte loads), so the ring uses global_load_dwordx4.
// The loads must stay where the source puts them (a ring that runs ahead of its consumers); hipcc
// sinks plain loads to their uses, a volatile load is encoded sc0 sc1 (slower), and a register the
// compiler allocates may be copied or spilled while its load is still in flight.  So the ring
// lives in twelve register quads the compiler never sees: the kernel is compiled with a VGPR
// budget that ends at v207 and v[208:255] belong to the inline asm below (they still count towards
// the kernel's 256-register allocation).  Loads return in order: when item idx is consumed,
// min(kRingH - 1, kh_END - 1 - idx) ring loads are younger than it, which is the vmcnt to wait for
// (loads the compiler issues itself only make that wait conservative).  The fragment is then
// copied out with two v_mov_b64.  tools/check_ring_isa.py verifies on the generated ISA that
// nothing else touches v[208:255] and that loads and takes follow the ring discipline.
#define GNNPP_RING_SLOTS(X)                                                                    \
    X(0, 208, 209, 210, 211) X(1, 212, 213, 214, 215) X(2, 216, 217, 218, 219)                 \
    X(3, 220, 221, 222, 223) X(4, 224, 225, 226, 227) X(5, 228, 229, 230, 231)                 \
    X(6, 232, 233, 234, 235) X(7, 236, 237, 238, 239) X(8, 240, 241, 242, 243)                 \
    X(9, 244, 245, 246, 247) X(10, 248, 249, 250, 251) X(11, 252, 253, 254, 255)

__device__ __forceinline__ const float* ring_item_ptr(const WStreamH& ws, int idx) { return h2_item_ptr(ws, idx); }

// WS: the kernel's stream descriptor (WStreamH here, WStreamB in encoder_kernel_b3.hip): a `lane_bytes` member and an
// overload of ring_item_ptr(ws, idx)
template <int END, class WS>
__device__ __forceinline__ void h2_ring_load(const WS& ws, v4f (&ring)[kRingH], int idx) {
    if (idx < END) {
        const float* p = ring_item_ptr(ws, idx);           // wave-uniform (scalar) fragment base
#if defined(__HIP_DEVICE_COMPILE__)
        (void)ring;
        switch (idx % kRingH) {                          // folds: idx is a constant after unrolling
#define GNNPP_X(s, a, b, c, d)                                                                 \
    case s:                                                                                    \
        asm volatile("global_load_dwordx4 v[" #a ":" #d "], %0, %1 ; RINGLOAD " #s             \
                     :: "v"(ws.lane_bytes), "s"(p));                                            \
        break;
            GNNPP_RING_SLOTS(GNNPP_X)
#undef GNNPP_X
        }
#else
        ring[idx % kRingH] = *reinterpret_cast<const v4f*>(
            reinterpret_cast<const char*>(p) + ws.lane_bytes);
#endif
    }
}

// the fragment of item idx (16 raw bytes), once it has landed
template <int END>
__device__ __forceinline__ v4f ring_take_f4(v4f (&ring)[kRingH], int idx) {
#if defined(__HIP_DEVICE_COMPILE__)
    (void)ring;
    const int younger = END - 1 - idx < kRingH - 1 ? END - 1 - idx : kRingH - 1;
    switch (younger) {
#define GNNPP_W(n) case n: asm volatile("s_waitcnt vmcnt(" #n ") ; RINGWAIT" ::: "memory"); break;
        GNNPP_W(0) GNNPP_W(1) GNNPP_W(2) GNNPP_W(3) GNNPP_W(4) GNNPP_W(5) GNNPP_W(6) GNNPP_W(7)
        GNNPP_W(8) GNNPP_W(9) GNNPP_W(10) GNNPP_W(11) GNNPP_W(12) GNNPP_W(13) GNNPP_W(14)
        default: asm volatile("s_waitcnt vmcnt(15) ; RINGWAIT" ::: "memory"); break;
#undef GNNPP_W
    }
    v2f lo, hi;
    switch (idx % kRingH) {
#define GNNPP_X(s, a, b, c, d)                                                                 \
    case s:                                                                                    \
        asm volatile("v_mov_b64 %0, v[" #a ":" #b "] ; RINGTAKE " #s                           \
                     "\n\tv_mov_b64 %1, v[" #c ":" #d "] ; RINGTAKE " #s                        \
                     : "=v"(lo), "=v"(hi));                                                    \
        break;
        GNNPP_RING_SLOTS(GNNPP_X)
#undef GNNPP_X
    }
    v4f r = {lo[0], lo[1], hi[0], hi[1]};
    return r;
#else
    return ring[idx % kRingH];
#endif
}
// DIRECT use of a ring slot (encoder_kernel_b3.hip, column-packed layers): the bf16 MFMA reads its A operand straight from
// the slot's registers -- no copy, and the accumulator is a tied operand, so the register allocator cannot move it -- and
// the slot is refilled behind the LAST MFMA that reads it.  ring_wait_for<END>(idx): item idx has landed, given that the
// loads up to idx_issued have been issued.  tools/check_ring_isa.py follows RINGUSE like RINGTAKE.
template <int END>
__device__ __forceinline__ void ring_wait_for(int idx, int idx_issued) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int last = idx_issued < END - 1 ? idx_issued : END - 1;
    const int younger = last - idx < 0 ? 0 : last - idx;
    switch (younger) {
#define GNNPP_W(n) case n: asm volatile("s_waitcnt vmcnt(" #n ") ; RINGWAIT" ::: "memory"); break;
        GNNPP_W(0) GNNPP_W(1) GNNPP_W(2) GNNPP_W(3) GNNPP_W(4) GNNPP_W(5) GNNPP_W(6) GNNPP_W(7)
        GNNPP_W(8) GNNPP_W(9) GNNPP_W(10) GNNPP_W(11) GNNPP_W(12) GNNPP_W(13) GNNPP_W(14)
        default: asm volatile("s_waitcnt vmcnt(15) ; RINGWAIT" ::: "memory"); break;
#undef GNNPP_W
    }
#else
    (void)idx; (void)idx_issued;
#endif
}
// acc += A(slot of item idx) x B   (v_mfma_f32_16x16x32_bf16)
__device__ __forceinline__ void ring_mfma16b(v4f (&ring)[kRingH], int idx, const v4f& B, v4f& acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    (void)ring;
    switch (idx % kRingH) {
#define GNNPP_X(s, a, b, c, d)                                                                 \
    case s:                                                                                    \
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, v[" #a ":" #d "], %1, %0 ; RINGUSE " #s     \
                     : "+v"(acc) : "v"(B));                                                    \
        break;
        GNNPP_RING_SLOTS(GNNPP_X)
#undef GNNPP_X
    }
#else
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8b, ring[idx % kRingH]), __builtin_bit_cast(v8b, B),
                                                  acc, 0, 0, 0);
#endif
}
// acc = A(slot of item idx) x B: an accumulator's FIRST product.  (Not `acc = 0` followed by the form above: the hazard
// recogniser does not know that the asm is an MFMA and leaves out the wait states between the v_mov that clears the
// accumulator and the MFMA that reads it -- measured as wrong logits at N = 10.)
__device__ __forceinline__ void ring_mfma16b_first(v4f (&ring)[kRingH], int idx, const v4f& B, v4f& acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    (void)ring;
    switch (idx % kRingH) {
#define GNNPP_X(s, a, b, c, d)                                                                 \
    case s:                                                                                    \
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, v[" #a ":" #d "], %1, 0 ; RINGUSE " #s      \
                     : "=&v"(acc) : "v"(B));                                                   \
        break;
        GNNPP_RING_SLOTS(GNNPP_X)
#undef GNNPP_X
    }
#else
    const v4f z = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8b, ring[idx % kRingH]), __builtin_bit_cast(v8b, B),
                                                  z, 0, 0, 0);
#endif
}
// the accumulators the asm MFMAs above wrote are read by compiler-scheduled code next: the wait states the hazard
// recogniser would insert behind a v_mfma it can see (it cannot see into inline asm)
__device__ __forceinline__ void ring_mfma_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
#endif
}

template <int END>
__device__ __forceinline__ v8h h2_ring_take(v4f (&ring)[kRingH], int idx) {
    return __builtin_bit_cast(v8h, ring_take_f4<END>(ring, idx));
}

__device__ __forceinline__ v8h as_h8(v4f v) { return __builtin_bit_cast(v8h, v); }
__device__ __forceinline__ v4f as_f4(v8h v) { return __builtin_bit_cast(v4f, v); }

// Range guard of the split-f16 schedules: the maximum of every group of values about to be split
// (one v_max3 per two values) is compared against 65504 (the f16 range: the hi half would be inf and
// hi + lo no longer x) and the wave-wide outcome is OR-ed into a scalar mask -- no vector register
// stays live for it; at the end of the kernel a non-zero mask raises the caller's range flag.
// Post-ReLU activations are >= 0, so only the observations and the filter's z rows need |x|.
constexpr float kF16Max = 65504.f;
typedef unsigned long long RangeMask;
__device__ __forceinline__ void range_note(float group_max, RangeMask& bad) {
    bad |= __ballot(group_max >= kF16Max);
}
__device__ __forceinline__ float max4(float m, v4f a) {
    return fmaxf(fmaxf(fmaxf(m, a[0]), fmaxf(a[1], a[2])), a[3]);
}
__device__ __forceinline__ float max4abs(float m, v4f a) {
    return fmaxf(fmaxf(fmaxf(m, fabsf(a[0])), fmaxf(fabsf(a[1]), fabsf(a[2]))), fabsf(a[3]));
}

// eight fp32 values (two D tiles) -> hi and lo f16 fragments
__device__ __forceinline__ void split8(v4f a, v4f b, v4f& hi, v4f& lo, RangeMask& bad) {
    range_note(max4(max4(0.f, a), b), bad);
    v8h h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = (_Float16)a[e];
        l[e] = (_Float16)(a[e] - (float)h[e]);
        h[4 + e] = (_Float16)b[e];
        l[4 + e] = (_Float16)(b[e] - (float)h[4 + e]);
    }
    hi = as_f4(h);
    lo = as_f4(l);
}
// one fp32 value -> the word (lo half << 16 | hi half), stored bit-for-bit in a float slot
__device__ __forceinline__ float split_word(float x) {
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    typedef _Float16 v2h __attribute__((ext_vector_type(2)));
    v2h p = {h, l};
    return __builtin_bit_cast(float, p);
}
// four fp32 values (one D tile) -> the 8-byte half of a hi and of a lo fragment
__device__ __forceinline__ void split4(v4f a, v2f& hi, v2f& lo, RangeMask& bad) {
    range_note(max4(0.f, a), bad);
    typedef _Float16 v4h __attribute__((ext_vector_type(4)));
    v4h h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = (_Float16)a[e];
        l[e] = (_Float16)(a[e] - (float)h[e]);
    }
    hi = __builtin_bit_cast(v2f, h);
    lo = __builtin_bit_cast(v2f, l);
}

// L1: wave w owns positions w, w + 4, ... (7 for wave 0, 6 otherwise)
template <int WAVE>
struct PosL1H {
    static __device__ __forceinline__ bool get(int j, int& y, int& x) {
        const int p = WAVE + 4 * j;
        y = p / 5; x = p % 5;
        return p < 25;
    }
};
// L2: pool windows {0, 3} (PAIR 0) or {1, 2} (PAIR 1); slot = 4 * which + (py * 2 + px)
template <int PAIR>
struct PosL2H {
    static __device__ __forceinline__ bool get(int j, int& y, int& x) {
        const int t = (j >> 2) == 0 ? PAIR : 3 - PAIR;
        y = 2 * (t >> 1) + ((j >> 1) & 1);
        x = 2 * (t & 1) + (j & 1);
        return true;
    }
};
struct Pos2x2H {
    static __device__ __forceinline__ bool get(int j, int& y, int& x) {
        y = j >> 1; x = j & 1;
        return true;
    }
};

// One (kb, tap) step IT of NMT channel tiles over a compile-time position set: B fragments of the
// positions the tap reaches (from LDS, or from the preloaded registers Pin), 3 MFMAs per pair.
// true iff step IT is the first (kb, tap) step, in stream order, whose tap reaches output slot j:
// its first MFMA then starts from a literal zero instead of a zero-initialised accumulator
template <int H, int W, class PosFn>
__device__ __forceinline__ bool first_step_of_slot(int it, int j) {
    int y = 0, x = 0;
    if (!PosFn::get(j, y, x)) return false;
    for (int s = 0; s < it; ++s) {
        const int tap = s % 9, dy = tap / 3 - 1, dx = tap % 3 - 1;
        if (y + dy >= 0 && y + dy < H && x + dx >= 0 && x + dx < W) return false;
    }
    return true;
}

template <int IT, int NKB, int H, int W, int NMT, int NSLOT, class PosFn, bool PRELOAD, int CH = NSLOT>
__device__ __forceinline__ void tap_mfma(const v4f* in, const v4f* Pin, const v8h (&Ah)[NMT],
                                         const v8h (&Al)[NMT], v4f (&acc)[NSLOT][NMT], int lane) {
    constexpr int kb = IT / 9, tap = IT % 9;
    constexpr int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
    for (int j0 = 0; j0 < NSLOT; j0 += CH) {               // CH slots at a time bounds the B registers
        v8h Bh[CH], Bl[CH];
#pragma unroll
        for (int jj = 0; jj < CH; ++jj) {
            const int j = j0 + jj;
            int y = 0, x = 0;
            const bool used = j < NSLOT && PosFn::get(j, y, x);
            const int iy = y + dy, ix = x + dx;
            if (used && iy >= 0 && iy < H && ix >= 0 && ix < W) {
                const int o = ((iy * W + ix) * NKB + kb) * 2;
                Bh[jj] = as_h8(PRELOAD ? Pin[o] : in[o * 64 + lane]);
                Bl[jj] = as_h8(PRELOAD ? Pin[o + 1] : in[(o + 1) * 64 + lane]);
            }
        }
#pragma unroll
        for (int term = 0; term < 3; ++term) {              // small terms first
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
                        acc[j][m] = mfma16h(term == 1 ? Al[m] : Ah[m], term == 0 ? Bl[jj] : Bh[jj],
                                            fresh ? vzero() : acc[j][m]);
                }
            }
        }
    }
}

// The weight stream of a layer: for every step IT (item order [kb][tap][mt][hi/lo]) take the
// 2 NMT fragments off the ring, refill the slots, and hand them to body(IT, Ah, Al).  All ring
// traffic is issued from this wave-uniform, branch-free code; per-wave specialisation (which
// positions a wave owns) lives inside `body`, so every path through the kernel performs the same
// ring sequence (tools/check_ring_isa.py relies on that).
template <int END, int START, int NMT, class Body, int... IT>
__device__ __forceinline__ void stream_steps(const WStreamH& ws, v4f (&ring)[kRingH], Body&& body,
                                             std::integer_sequence<int, IT...>) {
    auto step = [&](auto itc) {
        constexpr int it = decltype(itc)::value;
        __builtin_amdgcn_sched_barrier(kSchedItemMask);
        v8h Ah[NMT], Al[NMT];
#pragma unroll
        for (int m = 0; m < NMT; ++m) {
            const int idx = START + (it * NMT + m) * 2;
            Ah[m] = h2_ring_take<END>(ring, idx);
            h2_ring_load<END>(ws, ring, idx + kRingH);
            Al[m] = h2_ring_take<END>(ring, idx + 1);
            h2_ring_load<END>(ws, ring, idx + 1 + kRingH);
        }
        body(itc, Ah, Al);
    };
    (step(std::integral_constant<int, IT>{}), ...);
}

// a whole layer for one position set (no per-wave specialisation)
template <int END, int START, int NKB, int H, int W, int NMT, int NSLOT, class PosFn, bool PRELOAD>
__device__ __forceinline__ void conv_h2(const WStreamH& ws, v4f (&ring)[kRingH], const v4f* in,
                                        v4f (&acc)[NSLOT][NMT], int lane) {
    v4f Pin[PRELOAD ? H * W * NKB * 2 : 1];
    if (PRELOAD) {
#pragma unroll
        for (int i = 0; i < H * W * NKB * 2; ++i) Pin[i] = in[i * 64 + lane];
    }
    stream_steps<END, START, NMT>(ws, ring, [&](auto itc, const v8h (&Ah)[NMT], const v8h (&Al)[NMT]) {
        tap_mfma<decltype(itc)::value, NKB, H, W, NMT, NSLOT, PosFn, PRELOAD>(in, Pin, Ah, Al, acc, lane);
    }, std::make_integer_sequence<int, 9 * NKB>{});
}

#if defined(__HIP_DEVICE_COMPILE__)
// v[208:255] = the weight ring.  On gfx90a+ the backend doubles "amdgpu-num-vgpr" (the unified
// VGPR+AGPR file), so 104 is what caps the compiler at v207; check_ring_isa.py verifies it.
#define GNNPP_H2_VGPR_BUDGET __attribute__((amdgpu_num_vgpr(104)))
#else
#define GNNPP_H2_VGPR_BUDGET
#endif
// What the fused policy kernel needs after the encoder (FUSED = true): one workgroup = one graph of
// N <= 16 agents, so the graph filter (K = 3 taps, 128 -> 128) and the action head run right here on the
// features, which never leave the chip: no feature round trip through HBM, no second launch, and no
// filter kernel whose cost is one workgroup's latency.
struct PolicyTail {
    const void* S;            // [B,N,N] fp32 or fp64
    const float* filt_h2;     // split-f16 taps of gnnpp_filter_pack (+ {2^k, 2^-k} behind them)
    const float* gf_bias;     // [128] or nullptr
    const float* act_w;       // [5,128]
    const float* act_b;       // [5]
    float* logits;            // [N,B,5]
    int* range_flag;          // optional: set to 1 when a value left the f16 range
    int B, N, s_is_f64;
    int with_sim;             // 1: continue with the simulator step of this episode (gnnpp_rollout_policy_step):
    gnnpp_rollout sim;        //    move on these logits -> gso -> observations of the new positions
};
// LDS left behind the filter's z / y rows for the simulator step: positions, move scratch, GSO scratch
// and the episode's occupancy grid (K taps: z_0 .. z_{K-1} and the y rows precede the simulator's state)
constexpr size_t policy_sim_occ_bytes(int K) {
    return (kBufFloats - (K + 1) * 16 * 136) * sizeof(float) - 8 * kMaxAgents * sizeof(int) - kGsoSmemBytes;
}

template <bool FUSED, int KT>
__global__ GNNPP_H2_VGPR_BUDGET __launch_bounds__(kThreads, 2) void encoder_kernel_h2(const float* __restrict__ obs,
                                                                 const float* __restrict__ pk,
                                                                 float* __restrict__ feat, int M,
                                                                 int stop, int* __restrict__ range_flag,
                                                                 const PolicyTail pt) {
    constexpr int END = FUSED ? kh_FILT + 16 * KT : kh_END;
    // `stop` (GNNPP_MEASURE builds only, gnnpp_set_tuning): return after phase 1 = staging, 2 = L0, 3 = L1,
    // 4 = L2, 5 = L3, 6 = L4; 0 = the whole encoder
    extern __shared__ __attribute__((aligned(16))) char gnnpp_smem[];
    float* const X = reinterpret_cast<float*>(gnnpp_smem);          // activations
    float* const bufObs = X + kBufFloats;                            // padded observations, then Y
    v4f* const X4 = reinterpret_cast<v4f*>(X);
    v4f* const Y4 = reinterpret_cast<v4f*>(bufObs);                  // dead after L0: late layers
                                                                     // ping-pong X <-> Y
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: real branches per wave
    const int a = lane & 15;
    const int q = lane >> 4;
    const int agent0 = blockIdx.x * (FUSED ? pt.N : kTileAgents);   // FUSED: the tile is graph blockIdx.x
    GNNPP_STAMP(blockIdx.x, 11, tid == 0);
    RangeMask bad = 0;                                              // range guard (see range_note above)

    WStreamH ws;
    ws.seg[0] = pk + EncLayout::kH1;
    ws.seg[1] = pk + EncLayout::kH2 + (wave & 1) * (36 * EncLayout::kHItem);
    ws.seg[2] = pk + EncLayout::kH3 + wave * (36 * EncLayout::kHItem);
    ws.seg[3] = pk + EncLayout::kH4 + wave * (72 * EncLayout::kHItem);
    ws.seg[4] = pk + EncLayout::kHfc + wave * (16 * EncLayout::kHItem);
    ws.seg[5] = FUSED ? pt.filt_h2 + wave * (2 * 4 * 512) : pk;     // channel tiles 2w, 2w+1
    ws.lane_bytes = lane * 16;
    // BatchNorm scale/shift of L1..L4: fetched now, parked in LDS after L0 (behind Y's live part), so
    // that no compiler-issued global load (whose wait would drain the ring) sits between the layers
    float ssv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        ssv[i] = pk[EncLayout::kHss + min(tid + i * kThreads, EncLayout::kHssFloats - 1)];
    float* const sstab = bufObs + 16 * 256;                          // Y holds <= 16 fragments
    float* const Ssm = sstab + EncLayout::kHssFloats;                // FUSED: GSO, [16][17], zero padded
    // FUSED: what the epilogue of the filter reads -- act_w [5][128] | bias [128] | act_b [5] | 1 / split scale --
    // takes the same road (774 floats, four per thread): no load is left behind the weight ring
    float* const hconst = Ssm + 16 * 17;
    float hcv[4] = {0.f, 0.f, 0.f, 0.f};
    if (FUSED) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * kThreads;
            if (c < 640) hcv[i] = pt.act_w[c];
            else if (c < 768) hcv[i] = pt.gf_bias ? pt.gf_bias[c - 640] : 0.f;
            else if (c < 773) hcv[i] = pt.act_b[c - 768];
            else if (c == 773) hcv[i] = pt.filt_h2[filter_packed_h2_floats(128, 128, KT, 1) + 1];
        }
    }
    float sval = 0.f;
    if (FUSED) {
        const int m = tid >> 4, n = tid & 15;                        // one GSO entry per thread
        if (m < pt.N && n < pt.N) {
            const size_t i = ((size_t)blockIdx.x * pt.N + m) * pt.N + n;
            sval = pt.s_is_f64 ? (float)reinterpret_cast<const double*>(pt.S)[i]
                               : reinterpret_cast<const float*>(pt.S)[i];
        }
    }
    v4f ring[kRingH];
#if defined(__HIP_DEVICE_COMPILE__)
    // The ONE statement that names a ring register to the compiler: it makes the kernel allocate all
    // 256 VGPRs (v[208:255] are beyond the compiler's own budget, so hipcc warns that this clobber
    // "may not be preserved" -- which is the point: nothing but the ring asm may use them).
    asm volatile("; weight ring lives in v[208:255]" ::: "v255");
#endif
#pragma unroll
    for (int i = 0; i < kRingH; ++i) h2_ring_load<END>(ws, ring, i);

    // ---- observations: all loads first, zero-fill while they fly, then scatter -----------------------
    // The tile's floats start at an arbitrary 4-byte boundary (a graph of N agents is N * 363 floats), so
    // the 16-byte loads start at the aligned address below it: vector `idx` holds the tile's elements
    // 4 idx - shift .. + 3.  Vectors that would reach past the end of the tensor (or before its start)
    // are assembled from scalar loads instead.
    {
        constexpr int NV4 = (kTileAgents * kObsFloats + 3) / 4 + 1;
        constexpr int PER = (NV4 + kThreads - 1) / kThreads;
        const int n_agents = FUSED ? pt.N : min(kTileAgents, M - agent0);
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
        v4f* z = reinterpret_cast<v4f*>(bufObs);
        for (int i = tid; i < kObsFloatsLds / 4; i += kThreads) z[i] = vzero();
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            // (agent, channel, y, x) of the first element by division, the following ones by carry: the
            // index arithmetic was most of this phase's VALU work
            const int e0 = (tid + k * kThreads) * 4 - shift;
            const int es = max(e0, 0);
            int ag = es / kObsFloats;
            const int rem = es - ag * kObsFloats;
            int ch = rem / 121;
            const int r2 = rem - ch * 121;
            int y = r2 / 11, x = r2 - y * 11;
            float vmax = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int e = e0 + c;
                if (e >= 0) {
                    if (e < valid) {
                        vmax = fmaxf(vmax, fabsf(v[k][c]));
                        bufObs[ag * kAgentStride + ch * (kPadHW * kPadHW) + (y + 1) * kPadHW + x + 1] =
                            split_word(v[k][c]);
                    }
                    if (++x == 11) {
                        x = 0;
                        if (++y == 11) {
                            y = 0;
                            if (++ch == 3) { ch = 0; ++ag; }
                        }
                    }
                }
            }
            range_note(vmax, bad);                       // (wave-uniform control flow: a ballot)
        }
    }
    __syncthreads();
    if (GNNPP_STOP_AT(stop, 1)) return;
    if (!pt.with_sim) GNNPP_STAMP(blockIdx.x, 0, tid == 0);

    // ---- L0: 3 -> 32 @ 11x11 (the 10x10 the pool reads), direct, K = 27 in ONE 32-slot block -------
    // A pixel sits in LDS as the word (lo half << 16 | hi half).  Lane (q, agent) owns k-slots
    // (q, e): eight word reads at per-lane offsets (position offset = instruction immediate), two
    // pack operations per register pair, then 3 MFMAs per channel tile -- no im2col buffer.
    {
        const unsigned* const obsw = reinterpret_cast<const unsigned*>(bufObs);
        v8h A0h[2], A0l[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            A0h[i] = as_h8(*reinterpret_cast<const v4f*>(pk + EncLayout::kH0 + ((i * 2 + 0) * 64 + lane) * 4));
            A0l[i] = as_h8(*reinterpret_cast<const v4f*>(pk + EncLayout::kH0 + ((i * 2 + 1) * 64 + lane) * 4));
        }
        v4f sc[2], sh[2];
        load_ss(pk + EncLayout::kHss0, 32, 0, q, sc[0], sh[0]);
        load_ss(pk + EncLayout::kHss0, 32, 1, q, sc[1], sh[1]);
        int aoff[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            aoff[e] = a * kAgentStride + (q < 3 ? q * (kPadHW * kPadHW) + (e / 3) * kPadHW + e % 3
                                                : e < 3 ? e * (kPadHW * kPadHW) + 2 * kPadHW + 2 : 0);
        unsigned dcur[4][8], dnxt[4][8];
        auto load_window = [&](unsigned (&d)[4][8], int win) {
            const int wy = win / 5, wx = win - wy * 5;
            const unsigned* base = obsw + (2 * wy) * kPadHW + 2 * wx;
#pragma unroll
            for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                for (int e = 0; e < 8; ++e) d[pp][e] = base[aoff[e] + (pp >> 1) * kPadHW + (pp & 1)];
        };
        auto window = [&](const unsigned (&d)[4][8], int win) {
            v4f acc[4][2];
            v8h Bh[4], Bl[4];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                v4u h, l;
#pragma unroll
                for (int w = 0; w < 4; ++w) {            // one v_perm_b32 each: low halves / high halves
                    h[w] = __builtin_amdgcn_perm(d[pp][2 * w + 1], d[pp][2 * w], 0x05040100u);
                    l[w] = __builtin_amdgcn_perm(d[pp][2 * w + 1], d[pp][2 * w], 0x07060302u);
                }
                Bh[pp] = __builtin_bit_cast(v8h, h);
                Bl[pp] = __builtin_bit_cast(v8h, l);
                acc[pp][0] = vzero();
                acc[pp][1] = vzero();
            }
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[pp][i] = mfma16h(term == 1 ? A0l[i] : A0h[i], term == 0 ? Bl[pp] : Bh[pp],
                                             acc[pp][i]);
            v4f r[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                r[i] = vrelu(vfma(acc[0][i], sc[i], sh[i]));
#pragma unroll
                for (int pp = 1; pp < 4; ++pp) r[i] = vmax(r[i], vfma(acc[pp][i], sc[i], sh[i]));
            }
            v4f hi, lo;
            split8(r[0], r[1], hi, lo, bad);
            X4[(win * 2 + 0) * 64 + lane] = hi;
            X4[(win * 2 + 1) * 64 + lane] = lo;
        };
        // two windows per trip, operands of the next one in flight (no register copies)
        load_window(dcur, wave);
        for (int win = wave; win < 25; win += 2 * kWaves) {
            if (win + kWaves < 25) load_window(dnxt, win + kWaves);
            window(dcur, win);
            if (win + kWaves < 25) {
                if (win + 2 * kWaves < 25) load_window(dcur, win + 2 * kWaves);
                window(dnxt, win + kWaves);
            }
        }
    }
    __syncthreads();
    if (GNNPP_STOP_AT(stop, 2)) return;
    if (!pt.with_sim) GNNPP_STAMP(blockIdx.x, 1, tid == 0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (tid + i * kThreads < EncLayout::kHssFloats) sstab[tid + i * kThreads] = ssv[i];
    // (first read after L1's mid-layer barrier)
    if (FUSED) {
        Ssm[(tid >> 4) * 17 + (tid & 15)] = sval;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (tid + i * kThreads < 774) hconst[tid + i * kThreads] = hcv[i];
    }

    // ---- L1: 32 -> 32 @ 5x5, in place; wave = its positions x both channel tiles ---------------------
    {
        v4f sc[2], sh[2];
        v4f acc[7][2];                                    // first touched by a zero-source MFMA (tap_mfma)
        stream_steps<END, kh_L1, 2>(ws, ring, [&](auto itc, const v8h (&Ah)[2], const v8h (&Al)[2]) {
            constexpr int IT = decltype(itc)::value;
            switch (wave) {
                case 0: tap_mfma<IT, 1, 5, 5, 2, 7, PosL1H<0>, false>(X4, nullptr, Ah, Al, acc, lane); break;
                case 1: tap_mfma<IT, 1, 5, 5, 2, 7, PosL1H<1>, false>(X4, nullptr, Ah, Al, acc, lane); break;
                case 2: tap_mfma<IT, 1, 5, 5, 2, 7, PosL1H<2>, false>(X4, nullptr, Ah, Al, acc, lane); break;
                default: tap_mfma<IT, 1, 5, 5, 2, 7, PosL1H<3>, false>(X4, nullptr, Ah, Al, acc, lane); break;
            }
        }, std::make_integer_sequence<int, 9>{});
        __syncthreads();                                   // everyone is done reading L0's output
        load_ss(sstab + EncLayout::kHssL1, 32, 0, q, sc[0], sh[0]);
        load_ss(sstab + EncLayout::kHssL1, 32, 1, q, sc[1], sh[1]);
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int p = wave + 4 * j;
            if (p < 25) {
                v4f hi, lo;
                split8(vrelu(vfma(acc[j][0], sc[0], sh[0])), vrelu(vfma(acc[j][1], sc[1], sh[1])),
                       hi, lo, bad);
                X4[(p * 2 + 0) * 64 + lane] = hi;
                X4[(p * 2 + 1) * 64 + lane] = lo;
            }
        }
    }
    __syncthreads();
    if (GNNPP_STOP_AT(stop, 3)) return;
    if (!pt.with_sim) GNNPP_STAMP(blockIdx.x, 2, tid == 0);

    // ---- L2: 32 -> 64 @ 5x5 (the 4x4 the pool reads), pool -> [4][kb 2] : X -> Y -------------------
    {
        const int mp = wave & 1, pair = wave >> 1;         // channel tiles 2 mp, 2 mp + 1 = block mp
        v4f acc[8][2];                                    // first touched by a zero-source MFMA (tap_mfma)
        stream_steps<END, kh_L2, 2>(ws, ring, [&](auto itc, const v8h (&Ah)[2], const v8h (&Al)[2]) {
            constexpr int IT = decltype(itc)::value;
            if (pair == 0) tap_mfma<IT, 1, 5, 5, 2, 8, PosL2H<0>, false, 4>(X4, nullptr, Ah, Al, acc, lane);
            else           tap_mfma<IT, 1, 5, 5, 2, 8, PosL2H<1>, false, 4>(X4, nullptr, Ah, Al, acc, lane);
        }, std::make_integer_sequence<int, 9>{});
        v4f sc[2], sh[2];
        load_ss(sstab + EncLayout::kHssL2, 64, 2 * mp, q, sc[0], sh[0]);
        load_ss(sstab + EncLayout::kHssL2, 64, 2 * mp + 1, q, sc[1], sh[1]);
#pragma unroll
        for (int wi = 0; wi < 2; ++wi) {
            const int t = wi == 0 ? pair : 3 - pair;
            v4f r[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                r[m] = vrelu(vfma(acc[4 * wi][m], sc[m], sh[m]));
#pragma unroll
                for (int pp = 1; pp < 4; ++pp)
                    r[m] = vmax(r[m], vfma(acc[4 * wi + pp][m], sc[m], sh[m]));
            }
            v4f hi, lo;
            split8(r[0], r[1], hi, lo, bad);
            Y4[((t * 2 + mp) * 2 + 0) * 64 + lane] = hi;
            Y4[((t * 2 + mp) * 2 + 1) * 64 + lane] = lo;
        }
    }
    __syncthreads();
    if (GNNPP_STOP_AT(stop, 4)) return;
    if (!pt.with_sim) GNNPP_STAMP(blockIdx.x, 3, tid == 0);

    // ---- L3: 64 -> 64 @ 2x2, one channel tile per wave, input held in registers : Y -> X ------------
    {
        const int mt = wave;
        v4f acc[4][1];                                    // first touched by a zero-source MFMA (tap_mfma)
        conv_h2<END, kh_L3, 2, 2, 2, 1, 4, Pos2x2H, true>(ws, ring, Y4, acc, lane);
        v4f sc, sh;
        load_ss(sstab + EncLayout::kHssL3, 64, mt, q, sc, sh);
        v2f* const X2 = reinterpret_cast<v2f*>(X);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v2f hi, lo;
            split4(vrelu(vfma(acc[j][0], sc, sh)), hi, lo, bad);
            // fragment (pos j, block mt >> 1): this tile is its e = 4 (mt & 1) .. +3 half
            const int o = (j * 2 + (mt >> 1)) * 2;
            X2[((o + 0) * 64 + lane) * 2 + (mt & 1)] = hi;
            X2[((o + 1) * 64 + lane) * 2 + (mt & 1)] = lo;
        }
    }
    __syncthreads();
    if (GNNPP_STOP_AT(stop, 5)) return;
    if (!pt.with_sim) GNNPP_STAMP(blockIdx.x, 4, tid == 0);

    // ---- L4: 64 -> 128 @ 2x2, pool -> [1][kb 4], tiles 2w, 2w+1 per wave : X -> Y -------------------
    {
        v4f acc[4][2];                                    // first touched by a zero-source MFMA (tap_mfma)
        conv_h2<END, kh_L4, 2, 2, 2, 2, 4, Pos2x2H, true>(ws, ring, X4, acc, lane);
        v4f sc[2], sh[2];
        load_ss(sstab + EncLayout::kHssL4, 128, 2 * wave, q, sc[0], sh[0]);
        load_ss(sstab + EncLayout::kHssL4, 128, 2 * wave + 1, q, sc[1], sh[1]);
        v4f r[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            r[m] = vrelu(vfma(acc[0][m], sc[m], sh[m]));
#pragma unroll
            for (int j = 1; j < 4; ++j) r[m] = vmax(r[m], vfma(acc[j][m], sc[m], sh[m]));
        }
        v4f hi, lo;
        split8(r[0], r[1], hi, lo, bad);
        Y4[(wave * 2 + 0) * 64 + lane] = hi;
        Y4[(wave * 2 + 1) * 64 + lane] = lo;
    }
    __syncthreads();
    if (GNNPP_STOP_AT(stop, 6)) return;
    if (!pt.with_sim) GNNPP_STAMP(blockIdx.x, 5, tid == 0);

    // ---- FC 128 -> 128 + ReLU -> feat[agent][128]; tiles 2w, 2w+1 per wave ------------------------
    {
        v8h Bh[4], Bl[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            Bh[kb] = as_h8(Y4[(kb * 2 + 0) * 64 + lane]);
            Bl[kb] = as_h8(Y4[(kb * 2 + 1) * 64 + lane]);
        }
        v4f acc[2][2] = {{vzero(), vzero()}, {vzero(), vzero()}};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            __builtin_amdgcn_sched_barrier(kSchedItemMask);
            v8h Ah[2], Al[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int idx = kh_FC + (kb * 2 + m) * 2;
                Ah[m] = h2_ring_take<END>(ring, idx);
                h2_ring_load<END>(ws, ring, idx + kRingH);
                Al[m] = h2_ring_take<END>(ring, idx + 1);
                h2_ring_load<END>(ws, ring, idx + 1 + kRingH);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                acc[m][1] = mfma16h(Ah[m], Bl[kb], acc[m][1]);
                acc[m][1] = mfma16h(Al[m], Bh[kb], acc[m][1]);
                acc[m][0] = mfma16h(Ah[m], Bh[kb], acc[m][0]);
            }
        }
        if (FUSED) {
            // features of the graph's agents: rows of z_0 in LDS (fp32, row stride 136), lane (q, a)
            // holds channels 16 mt + 4 q .. + 3 of agent a
            const float inv = pk[EncLayout::kHinv + 4];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int mt = 2 * wave + m;
                const v4f b = *reinterpret_cast<const v4f*>(pk + EncLayout::kBfc + mt * 16 + q * 4);
                *reinterpret_cast<v4f*>(X + a * kZs + mt * 16 + q * 4) =
                    vrelu((acc[m][0] + acc[m][1]) * inv + b);
            }
        } else if (agent0 + a < M) {
            const float inv = pk[EncLayout::kHinv + 4];
            float* dst = feat + (size_t)(agent0 + a) * 128 + q * 4;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int mt = 2 * wave + m;
                const v4f b = *reinterpret_cast<const v4f*>(pk + EncLayout::kBfc + mt * 16 + q * 4);
                *reinterpret_cast<v4f*>(dst + mt * 16) = vrelu((acc[m][0] + acc[m][1]) * inv + b);
            }
        }
    }
    if (!FUSED) {
        if (range_flag && bad) *range_flag = 1;
        return;
    }

    // ==== graph filter + action head of this graph (K = KT taps, G = F = 128) =====================
    // z_k = z_{k-1} S as a dense product on the fp32 MFMA, all m in ascending order: bit-identical to
    // lsigf_kernel's sparse gather (fmaf(0, z, acc) == acc).  Rows >= N hold the features of the
    // zero-observation padding lanes; S is zero there, so they never reach a real node.
    float* const z0 = X;
    __syncthreads();                                     // z_0 complete
    GNNPP_STAMP(blockIdx.x, 12, tid == 0);
#pragma unroll
    for (int k = 1; k < KT; ++k) {
        const float* zp = z0 + (k - 1) * (16 * kZs);
        float* zn = z0 + k * (16 * kZs);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ft = 2 * wave + t;
            v4f d = vzero();
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int m = 4 * st + q;
                d = mfma16(zp[m * kZs + 16 * ft + a], Ssm[m * 17 + a], d);   // A[i = feature][k = m], B[k = m][j = node]
            }
            *reinterpret_cast<v4f*>(zn + a * kZs + 16 * ft + 4 * q) = d;    // node a, features 16 ft + 4 q ..
        }
        __syncthreads();
    }
    GNNPP_STAMP(blockIdx.x, 13, tid == 0);
    // z_0..z_{K-1} -> (hi, lo) half rows in place: 16 K rows, a half-wave per row (as lsigf_kernel's split_rows)
    {
        typedef _Float16 v4h __attribute__((ext_vector_type(4)));
        const int half = lane >> 5, hl = lane & 31;
        for (int rb = 2 * wave; rb < 16 * KT; rb += 2 * kWaves) {
            float* row = z0 + (rb + half) * kZs;
            const v4f v = *reinterpret_cast<const v4f*>(row + 4 * hl);
            __builtin_amdgcn_wave_barrier();             // all reads of a row precede its writes
            range_note(max4abs(0.f, v), bad);
            v4h h, l;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                h[c] = (_Float16)v[c];
                l[c] = (_Float16)(v[c] - (float)h[c]);
            }
            *reinterpret_cast<v2f*>(row + 2 * hl) = __builtin_bit_cast(v2f, h);
            *reinterpret_cast<v2f*>(row + 64 + 2 * hl) = __builtin_bit_cast(v2f, l);
        }
    }
    __syncthreads();
    // contraction on the f16 pipe: channel tiles 2w, 2w+1; same term order as lsigf_kernel
    v4f fa[2] = {vzero(), vzero()}, fc[2] = {vzero(), vzero()};
#pragma unroll
    for (int tap = 0; tap < KT; ++tap) {
        const float* zr = z0 + (tap * 16 + a) * kZs + 4 * q;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            __builtin_amdgcn_sched_barrier(kSchedItemMask);
            v8h Ah[2], Al[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int idx = kh_FILT + ((tap * 4 + kb) * 2 + m) * 2;
                Ah[m] = h2_ring_take<END>(ring, idx);
                h2_ring_load<END>(ws, ring, idx + kRingH);
                Al[m] = h2_ring_take<END>(ring, idx + 1);
                h2_ring_load<END>(ws, ring, idx + 1 + kRingH);
            }
            const v8h Bh = as_h8(*reinterpret_cast<const v4f*>(zr + 16 * kb));
            const v8h Bl = as_h8(*reinterpret_cast<const v4f*>(zr + 64 + 16 * kb));
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                fc[m] = mfma16h(Ah[m], Bl, fc[m]);
                fa[m] = mfma16h(Ah[m], Bh, fa[m]);
                fc[m] = mfma16h(Al[m], Bh, fc[m]);
            }
        }
    }
    GNNPP_STAMP(blockIdx.x, 14, tid == 0);
    if (range_flag && bad) *range_flag = 1;               // (every split of this kernel is behind us)
    // bias + ReLU in registers, then the 128 -> 5 action head on the fp32 MFMA where the accumulators are: a lane
    // holds y[node a, 4 features of its channel tile] -- the B operand of the 16x16x4 MFMA against act_w's
    // columns of that tile -- so each wave multiplies its two tiles (one chain of 8 MFMAs from zero) and the four
    // waves' partial logits meet in LDS, summed in wave order (lsigf_kernel's head pairs the tiles the same
    // way: identical logits).  Every constant comes from LDS (parked there before the ring started).
    float* const yb = z0 + KT * (16 * kZs);              // [4 waves][16 nodes][8]: partial logits
    {
        const float finv = hconst[773];
        v4f d = vzero();
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int f0 = (2 * wave + m) * 16 + 4 * q;
            const v4f bv = *reinterpret_cast<const v4f*>(hconst + 640 + f0);
            const v4f A5 = a < 5 ? *reinterpret_cast<const v4f*>(hconst + a * 128 + f0) : vzero();
            d = mfma16x4(A5, vrelu((fa[m] + fc[m]) * finv + bv), d);
        }
        if (q < 2) *reinterpret_cast<v4f*>(yb + (wave * 16 + a) * 8 + 4 * q) = d;   // outputs 4 q + reg of node a
    }
    __syncthreads();
    if (wave == 0) {
        v4f d = *reinterpret_cast<const v4f*>(yb + a * 8 + 4 * (q & 1));
#pragma unroll
        for (int w = 1; w < kWaves; ++w) d += *reinterpret_cast<const v4f*>(yb + (w * 16 + a) * 8 + 4 * (q & 1));
        const v4f ab4 = *reinterpret_cast<const v4f*>(hconst + 768 + 4 * (q & 1));   // act_b[0..3] | act_b[4], .
        if (a < pt.N && q < 2) {                          // lane holds node a, outputs 4 q + reg
            float* dst = pt.logits + ((size_t)a * pt.B + blockIdx.x) * 5;
            // with the simulator tail: a copy [N][5] in LDS for this wave's move (red + 2 kMaxAgents, see below)
            float* lds = reinterpret_cast<float*>(z0 + (KT + 1) * (16 * kZs)) + 4 * kMaxAgents + a * 5;
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

    // ==== simulator step of this episode (same code as rollout_step_kernel) ===========================
    // Wave 0 goes from the head straight into the move (the logits wait for it in LDS); the other waves are
    // already staging the map meanwhile -- no barrier between head and simulator: the simulator's LDS lies behind
    // the z / y rows.  Then everybody builds the GSO and the observations of the new positions -- into the very
    // obs / S rows this workgroup consumed at its start, which nobody else reads.
    {
        int* spos = reinterpret_cast<int*>(z0 + (KT + 1) * (16 * kZs));
        int* red = spos + 2 * kMaxAgents;
        int* goal_l = red + 4 * kMaxAgents;
        char* gso_smem = reinterpret_cast<char*>(goal_l + 2 * kMaxAgents);
        unsigned char* occ = reinterpret_cast<unsigned char*>(gso_smem + kGsoSmemBytes);
        const int b = blockIdx.x;
        GNNPP_STAMP(b, 10, tid == 0);
        // the collision passes' cell-count map (one more byte per grid cell) behind the occupancy grid, if it fits
        const size_t occ_bytes = ((size_t)pt.sim.H * pt.sim.W + 15) & ~(size_t)15;
        unsigned* cellcnt = 2 * occ_bytes <= policy_sim_occ_bytes(KT) ? reinterpret_cast<unsigned*>(occ + occ_bytes)
                                                                      : nullptr;
        // (no LDS stage for the observation rows here: measured no gain at N <= 16 -- 14.5 KB per workgroup leave
        // in two store rounds either way -- for one more barrier)
        sim_tail(pt.sim, b, spos, red, goal_l, gso_smem, occ, tid, kThreads, cellcnt,
                 reinterpret_cast<const float*>(red + 2 * kMaxAgents));
    }
}

#ifdef GNNPP_MEASURE
std::atomic<int> g_encoder_stop{0};  // measurement only (GNNPP_TUNE_ENCODER_STOP)
#define GNNPP_ENCODER_STOP_VALUE g_encoder_stop.load(std::memory_order_relaxed)
#else
#define GNNPP_ENCODER_STOP_VALUE 0
#endif

int encoder_launch_h2(const float* obs, const float* packed, float* feat, int M, int* range_flag,
                      hipStream_t st) {
    static LdsAttrOnce once;
    constexpr size_t smem = (kBufFloats + kObsFloatsLds) * sizeof(float);
    set_lds_attr_once(once, reinterpret_cast<const void*>(&encoder_kernel_h2<false, 3>), (int)smem);
    const int grid = (M + kTileAgents - 1) / kTileAgents;
    hipLaunchKernelGGL((encoder_kernel_h2<false, 3>), dim3(grid), dim3(kThreads), smem, st, obs, packed, feat, M,
                       GNNPP_ENCODER_STOP_VALUE, range_flag, PolicyTail{});   // (zero-initialised: unused by <false>)
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// whole policy step of B graphs with N <= 16 agents and K = 2, 3 or 4 taps: one workgroup per graph
template <int KT>
static int policy_launch_fused_k(const float* obs, const float* packed, const PolicyTail& pt, hipStream_t st) {
    static LdsAttrOnce once;
    constexpr size_t smem = (kBufFloats + kObsFloatsLds) * sizeof(float);
    set_lds_attr_once(once, reinterpret_cast<const void*>(&encoder_kernel_h2<true, KT>), (int)smem);
    hipLaunchKernelGGL((encoder_kernel_h2<true, KT>), dim3(pt.B), dim3(kThreads), smem, st, obs, packed,
                       static_cast<float*>(nullptr), pt.B * pt.N, 0, pt.range_flag, pt);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int policy_launch_fused(const float* obs, const float* packed, const PolicyTail& pt, int K, hipStream_t st) {
    switch (K) {
        case 2: return policy_launch_fused_k<2>(obs, packed, pt, st);
        case 3: return policy_launch_fused_k<3>(obs, packed, pt, st);
        case 4: return policy_launch_fused_k<4>(obs, packed, pt, st);
        default: return -2;
    }
}

}  // namespace gnnpp
