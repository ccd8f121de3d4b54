// Shared device-side definitions for libgnnpp (gfx950 / CDNA4 only).
//
// MFMA convention used by every kernel in this library: v_mfma_f32_16x16x4_f32, exact fp32
// (bitwise an fmaf chain), D[i][j] += sum_k A[i][k] * B[k][j] with
//     A operand of lane l : A[i = l & 15][k = l >> 4]          (one VGPR)
//     B operand of lane l : B[k = l >> 4][j = l & 15]          (one VGPR)
//     D register r of lane l : D[i = (l >> 4) * 4 + r][j = l & 15]   (four VGPRs)
// We always put OUTPUT CHANNELS on i (weights are the A operand, read from L2 in a pre-packed
// fragment order) and AGENTS / graph nodes on j (activations are the B operand, read from LDS).
//
// "Fragment order" of a 16-channel input group: four consecutive MFMA k-steps s = 0..3 consume
// the channels c = 4*q + s (q = l >> 4), so that ONE 16-byte load per lane feeds four MFMAs and
// the D registers of a 16-output-channel tile (channel 4*q + r in register r) are already the
// B fragment of the next layer: a layer's epilogue stores its v4f, the next layer loads it back,
// lane for lane, with ds_write_b128 / ds_read_b128 and no shuffles.
#ifndef GNNPP_COMMON_H_
#define GNNPP_COMMON_H_

#include <hip/hip_runtime.h>

#include <atomic>

namespace gnnpp {

#ifdef GNNPP_MEASURE
__device__ unsigned long long g_stamps[1024 * 32];
#endif

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));   // one 16x16x32 f16 MFMA operand
typedef __bf16 v8b __attribute__((ext_vector_type(8)));     // one 16x16x32 bf16 MFMA operand
typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;            // CDNA wavefront
constexpr int kThreads = 256;        // 4 waves per workgroup, one per SIMD
constexpr int kWaves = kThreads / kWave;
constexpr int kLdsBytes = 160 * 1024;

// Phase ablation / early exit for profiling exist only in -DGNNPP_MEASURE builds (tools/ab_bench.py
// builds its own libgnnpp_measure.so); the product library has no knob that changes results.
#ifdef GNNPP_MEASURE
#define GNNPP_ABLATE(p, bits) ((p).ablate & (bits))
#define GNNPP_STOP_AT(stop, phase) (stop == phase)
// phase time stamps of workgroup `wg`, slot 0..15, read back by gnnpp_measure_read_stamps(): where the
// time goes INSIDE a kernel.  [slot] = 100 MHz wall clock, [16 + slot] = shader clock cycles (their ratio
// is the engine clock the kernel really ran at, which prices the MFMA-pipe bound of a phase)
#define GNNPP_STAMP(wg, slot, leader)                                                            \
    do { if (leader) { gnnpp::g_stamps[((wg) & 1023) * 32 + (slot)] = wall_clock64();             \
                       gnnpp::g_stamps[((wg) & 1023) * 32 + 16 + (slot)] = clock64(); } } while (0)
#else
#define GNNPP_ABLATE(p, bits) 0
#define GNNPP_STOP_AT(stop, phase) false
#define GNNPP_STAMP(wg, slot, leader) do { } while (0)
#endif

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device), thread-safe: one
// LdsAttrOnce per kernel instantiation (a function-local static at the launch site), one bit per
// device ordinal.  A process that drives several GPUs sets the attribute on each of them.
struct LdsAttrOnce {
    std::atomic<unsigned long long> mask[4] = {};
};
inline void set_lds_attr_once(LdsAttrOnce& once, const void* kernel, int bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    std::atomic<unsigned long long>& m = once.mask[(dev >> 6) & 3];
    if (m.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    m.fetch_or(bit, std::memory_order_release);
}

__device__ __forceinline__ v4f mfma16(float a, float b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// four k-steps fed by one packed A fragment and one packed B fragment
__device__ __forceinline__ v4f mfma16x4(v4f a, v4f b, v4f c) {
    c = mfma16(a[0], b[0], c);
    c = mfma16(a[1], b[1], c);
    c = mfma16(a[2], b[2], c);
    c = mfma16(a[3], b[3], c);
    return c;
}

// v_mfma_f32_16x16x32_f16: lane l holds A[i = l & 15][k-slots (q = l >> 4, e = 0..7)] and
// B[k-slots (q, e)][j = l & 15]; D as for the fp32 form.  fp32 accumulate; f16 subnormal operands
// are not flushed (tools/probe/f16_probe.hip).
__device__ __forceinline__ v4f mfma16h(v8h a, v8h b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// v_mfma_f32_16x16x32_bf16: same lane map and rate as the f16 form; bf16 has fp32's exponent range, so three
// bf16 planes represent ANY finite fp32 value exactly (bf16x3, below).
__device__ __forceinline__ v4f mfma16b(v8b a, v8b b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ---- bf16x3: the fp32-equivalent operand form of the matrix pipe ("b3") ------------------------------------
//      x = h + m + l,   h = bf16(x),  m = bf16(x - h),  l = bf16(x - h - m)      (round to nearest even)
// Both residuals are exact in fp32 and l is exactly representable (24 significand bits = 8 + 8 + 8), so the
// three planes carry x EXACTLY, with fp32's exponent range: no domain restriction, no weight pre-scaling, no
// guard.  A product keeps six of the nine plane products,
//      w x ~= wh xl + wl xh + wm xm + wh xm + wm xh + wh xh        (fp32 accumulate, small terms first),
// dropping wm xl + wl xm + wl xl <= 2^-23 |w x| in the worst case (2^-26 typical): below the rounding of one
// fp32 multiply-add.  Six v_mfma_f32_16x16x32_bf16 per 32 channels = 96 matrix-pipe cycles against the 256 of
// eight v_mfma_f32_16x16x4_f32: 2.7x the fp32 pipe's rate at fp32's accuracy.
// |x| > 0x7f7f0000 (3.39e38, the largest bf16) would round h to infinity: the value handed to the first
// conversion is clamped there (v_med3), the residual then carries the rest -- still exact.
constexpr int kB3Planes = 3;
constexpr int kB3Terms = 6;
// (weight plane, activation plane) of term t, small terms first
// (always inlined: left to the inliner's budget, the calls survive in the largest kernels -- the plane index then is a
// run-time value and every fragment array a dynamically indexed one: 350 000 lines of select chains and spills)
__host__ __device__ __forceinline__ constexpr int b3_term_a(int t) { return t == 0 ? 0 : t == 1 ? 2 : t == 2 ? 1 : t == 3 ? 0 : t == 4 ? 1 : 0; }
__host__ __device__ __forceinline__ constexpr int b3_term_b(int t) { return t == 0 ? 2 : t == 1 ? 0 : t == 2 ? 1 : t == 3 ? 1 : t == 4 ? 0 : 0; }

__device__ __forceinline__ unsigned b3_cvt_pk(float lo, float hi) {      // two fp32 -> two bf16 (RNE), lo in bits 0..15
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
#else
    auto one = [](float f) -> unsigned {
        unsigned u;
        __builtin_memcpy(&u, &f, 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;                 // NaN stays NaN
        return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    };
    return (one(lo) & 0xffffu) | (one(hi) << 16);
#endif
}
__device__ __forceinline__ float b3_sub(float a, float b) {              // a - b as ONE v_sub_f32 (the SLP vectoriser
#if defined(__HIP_DEVICE_COMPILE__)                                      // would pair these into v_pk_add_f32, which
    float r;                                                             // is slower beside MFMAs)
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return a - b;
#endif
}
__device__ __forceinline__ float b3_bits(unsigned u) {
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
// (NaN: v_med3_f32 returns min3 of its operands when one of them is NaN, and min ignores a NaN operand -- the h plane
// of a NaN is -big, the residual x - h is NaN again, so the m plane carries the NaN on: a NaN input stays a NaN.)
__device__ __forceinline__ float b3_clamp(float x) {
    const float big = b3_bits(0x7f7f0000u);
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(x, -big, big);
#else
    return x != x ? -big : (x < -big ? -big : x > big ? big : x);     // (the device's med3, NaN included)
#endif
}
// two fp32 values -> their three planes as packed bf16 pairs (x0 in the low halves)
__device__ __forceinline__ void b3_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = b3_cvt_pk(b3_clamp(x0), b3_clamp(x1));
    const float r0 = b3_sub(x0, b3_bits(h << 16));
    const float r1 = b3_sub(x1, b3_bits(h & 0xffff0000u));
    m = b3_cvt_pk(r0, r1);
    const float s0 = b3_sub(r0, b3_bits(m << 16));
    const float s1 = b3_sub(r1, b3_bits(m & 0xffff0000u));
    l = b3_cvt_pk(s0, s1);
}
// ... of values that are already inside [-big, big] (a ReLU fused with the upper clamp, b3_relu_clamp)
__device__ __forceinline__ void b3_split2_clamped(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = b3_cvt_pk(x0, x1);
    const float r0 = b3_sub(x0, b3_bits(h << 16));
    const float r1 = b3_sub(x1, b3_bits(h & 0xffff0000u));
    m = b3_cvt_pk(r0, r1);
    const float s0 = b3_sub(r0, b3_bits(m << 16));
    const float s1 = b3_sub(r1, b3_bits(m & 0xffff0000u));
    l = b3_cvt_pk(s0, s1);
}
// max(x, 0) and the split's upper clamp in ONE v_med3_f32.  NON-FINITE activations are flushed here, unlike
// torch.relu: +inf becomes 3.39e38 (the largest bf16) and NaN becomes 0 (med3 with a NaN operand returns min3, and
// min ignores the NaN: min3(NaN, 0, big) = 0).  A network whose L0 activations overflow fp32 is outside what the
// parity tests pin (the reference produces inf / NaN logits there); the host build of this header (tests) mirrors the device.
__device__ __forceinline__ float b3_relu_clamp(float x) {
    const float big = b3_bits(0x7f7f0000u);
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(x, 0.f, big);
#else
    return x != x ? 0.f : (x < 0.f ? 0.f : x > big ? big : x);
#endif
}
// eight fp32 values (two D tiles: a = k-slots e 0..3, b = e 4..7) -> the three 16-byte plane fragments
__device__ __forceinline__ void b3_split8(v4f a, v4f b, v4f (&pl)[3]) {
    unsigned h[4], m[4], l[4];
    b3_split2(a[0], a[1], h[0], m[0], l[0]);
    b3_split2(a[2], a[3], h[1], m[1], l[1]);
    b3_split2(b[0], b[1], h[2], m[2], l[2]);
    b3_split2(b[2], b[3], h[3], m[3], l[3]);
    const v4u hv = {h[0], h[1], h[2], h[3]}, mv = {m[0], m[1], m[2], m[3]}, lv = {l[0], l[1], l[2], l[3]};
    pl[0] = __builtin_bit_cast(v4f, hv);
    pl[1] = __builtin_bit_cast(v4f, mv);
    pl[2] = __builtin_bit_cast(v4f, lv);
}
// ReLU + the split's upper clamp of four values, one v_med3_f32 each (same result as vrelu followed by the clamp of
// b3_split2 for every input, NaN and the infinities included: both give 0 / 0 / big)
__device__ __forceinline__ v4f vrelu_clamp(v4f v) {
    v4f r;
    r[0] = b3_relu_clamp(v[0]); r[1] = b3_relu_clamp(v[1]);
    r[2] = b3_relu_clamp(v[2]); r[3] = b3_relu_clamp(v[3]);
    return r;
}
__device__ __forceinline__ void b3_split8_clamped(v4f a, v4f b, v4f (&pl)[3]) {
    unsigned h[4], m[4], l[4];
    b3_split2_clamped(a[0], a[1], h[0], m[0], l[0]);
    b3_split2_clamped(a[2], a[3], h[1], m[1], l[1]);
    b3_split2_clamped(b[0], b[1], h[2], m[2], l[2]);
    b3_split2_clamped(b[2], b[3], h[3], m[3], l[3]);
    const v4u hv = {h[0], h[1], h[2], h[3]}, mv = {m[0], m[1], m[2], m[3]}, lv = {l[0], l[1], l[2], l[3]};
    pl[0] = __builtin_bit_cast(v4f, hv);
    pl[1] = __builtin_bit_cast(v4f, mv);
    pl[2] = __builtin_bit_cast(v4f, lv);
}
// four fp32 values (one D tile) -> the 8-byte halves of the three plane fragments
__device__ __forceinline__ void b3_split4(v4f a, v2f (&pl)[3]) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    unsigned h[2], m[2], l[2];
    b3_split2(a[0], a[1], h[0], m[0], l[0]);
    b3_split2(a[2], a[3], h[1], m[1], l[1]);
    const v2u hv = {h[0], h[1]}, mv = {m[0], m[1]}, lv = {l[0], l[1]};
    pl[0] = __builtin_bit_cast(v2f, hv);
    pl[1] = __builtin_bit_cast(v2f, mv);
    pl[2] = __builtin_bit_cast(v2f, lv);
}
__device__ __forceinline__ v8b as_b8(v4f v) { return __builtin_bit_cast(v8b, v); }

__device__ __forceinline__ v4f vzero() { v4f z = {0.f, 0.f, 0.f, 0.f}; return z; }

__device__ __forceinline__ v4f vrelu(v4f v) {
    v4f r;
    r[0] = fmaxf(v[0], 0.f); r[1] = fmaxf(v[1], 0.f);
    r[2] = fmaxf(v[2], 0.f); r[3] = fmaxf(v[3], 0.f);
    return r;
}

__device__ __forceinline__ v4f vmax(v4f a, v4f b) {
    v4f r;
    r[0] = fmaxf(a[0], b[0]); r[1] = fmaxf(a[1], b[1]);
    r[2] = fmaxf(a[2], b[2]); r[3] = fmaxf(a[3], b[3]);
    return r;
}

__device__ __forceinline__ v4f vfma(v4f a, v4f b, v4f c) {
    v4f r;
    r[0] = fmaf(a[0], b[0], c[0]); r[1] = fmaf(a[1], b[1], c[1]);
    r[2] = fmaf(a[2], b[2], c[2]); r[3] = fmaf(a[3], b[3], c[3]);
    return r;
}

// sum of `v` over the 64 lanes of the wave, in every lane: xor butterfly, a fixed association order
// (deterministic), no LDS.  Every lane of the wave must take part.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// value of `v` held by the lane whose index differs in bit 0 / bit 1 (the partner inside a quad of four lanes): one
// v_mov_b32_dpp quad_perm -- no LDS, no SALU.  Every lane of the wave must take part.
__device__ __forceinline__ float quad_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm:[1,0,3,2]
}
__device__ __forceinline__ float quad_xor2(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm:[2,3,0,1]
}

// value of `v` held by lane `src` (src must be wave-uniform)
__device__ __forceinline__ float wave_read_lane(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// ---- packed layouts (in floats) -------------------------------------------------------------
// graph-filter taps: [fp32 fragments: block (e, k, mt, gg) of 64 lanes x 4 floats]
//                    [split-f16 fragments: block (e, k, mt, kb, hi/lo) of 64 lanes x 8 halves]
//                    [2^k, 2^-k, 0, 0]
//                    [bf16x3 fragments: block (e, k, mt, kb, plane) of 64 lanes x 8 bf16]   see lsigf_kernel.hip
__host__ __device__ inline size_t filter_packed_f32_floats(int G, int F, int K, int E) {
    const size_t NG = (G + 15) / 16, MT = (F + 15) / 16;
    return (size_t)E * K * MT * NG * 256;
}
__host__ __device__ inline size_t filter_packed_h2_floats(int G, int F, int K, int E) {
    const size_t KB = (G + 31) / 32, MT = (F + 15) / 16;
    return (size_t)E * K * MT * KB * 512;
}
// bf16x3 fragments: block (e, k, mt, kb) = [plane 3][lane 64][8 bf16] = 768 floats, behind the scale pair
__host__ __device__ inline size_t filter_packed_b3_floats(int G, int F, int K, int E) {
    const size_t KB = (G + 31) / 32, MT = (F + 15) / 16;
    return (size_t)E * K * MT * KB * 768;
}
__host__ __device__ inline size_t filter_packed_b3_offset(int G, int F, int K, int E) {
    return filter_packed_f32_floats(G, F, K, E) + filter_packed_h2_floats(G, F, K, E) + 4;
}
__host__ __device__ inline size_t filter_packed_floats(int G, int F, int K, int E) {
    return filter_packed_b3_offset(G, F, K, E) + filter_packed_b3_floats(G, F, K, E);
}

// encoder: offsets of each layer's block inside the packed buffer
struct EncLayout {
    // conv layers 1..4 and the FC use the generic fragment order [mt][tap][g][lane][4]
    static constexpr int kCin[5] = {3, 32, 32, 64, 64};
    static constexpr int kCout[5] = {32, 32, 64, 64, 128};
    static constexpr int kL0Steps = 7;                                   // K = 27 padded to 28
    static constexpr int kW0 = 0;                                        // [mt 2][s 7][lane 64]
    static constexpr int kSS0 = kW0 + 2 * kL0Steps * 64;                 // scale[32], shift[32]
    static constexpr int kW1 = kSS0 + 64;                                // [2][9][2][256]
    static constexpr int kSS1 = kW1 + 2 * 9 * 2 * 256;
    static constexpr int kW2 = kSS1 + 64;                                // [4][9][2][256]
    static constexpr int kSS2 = kW2 + 4 * 9 * 2 * 256;
    static constexpr int kW3 = kSS2 + 128;                               // [4][9][4][256]
    static constexpr int kSS3 = kW3 + 4 * 9 * 4 * 256;
    static constexpr int kW4 = kSS3 + 128;                               // [8][9][4][256]
    static constexpr int kSS4 = kW4 + 8 * 9 * 4 * 256;
    static constexpr int kWfc = kSS4 + 256;                              // [8][8][256]
    static constexpr int kBfc = kWfc + 8 * 8 * 256;                      // bias[128]
    // Winograd F(2x2,3x3) weights U = G g G^T (16 values per 3x3 kernel) for L0 and L2
    static constexpr int kU0 = kBfc + 128;                               // [mt 2][wpos 16][lane 64]
    static constexpr int kU2 = kU0 + 2 * 16 * 64;                        // [mt 4][g 2][wpos 16][lane 64][4]
    // split-f16 path (encoder_kernel_h2.hip): w * 2^k = hi + lo, two f16 fragments of 16 bytes per
    // lane, stored in each wave's consumption order [group][kb][tap][mt_local][hi/lo][lane 64][8 h]
    static constexpr int kHItem = 256;                                   // floats per 1 KiB fragment
    static constexpr int kH1 = kU2 + 4 * 2 * 16 * 256;                   // [1][1][9][2][2] items
    static constexpr int kH2 = kH1 + 36 * kHItem;                        // [2][1][9][2][2]
    static constexpr int kH3 = kH2 + 72 * kHItem;                        // [4][2][9][1][2]
    static constexpr int kH4 = kH3 + 144 * kHItem;                       // [4][2][9][2][2]
    static constexpr int kHfc = kH4 + 288 * kHItem;                      // [4][4][1][2][2]
    static constexpr int kHinv = kHfc + 64 * kHItem;                     // 2^-k of layers 1..4, FC, 0
    static constexpr int kHscale = kHinv + 8;                            // 2^k  (same order)
    // L0 (K = 27 in one 32-slot block): [mt 2][hi/lo][lane 64][e 8]; k-slot (q, e) is
    // (channel q, tap e) for q < 3, (channel e, tap 8) for q = 3 and e < 3, unused (zero) otherwise
    static constexpr int kH0 = kHscale + 8;
    // BatchNorm scale/shift with the weight scale undone (scale * 2^-k): L0 [sc 32][sh 32], then the
    // table the kernel keeps in LDS: L1 [32][32] | L2 [64][64] | L3 [64][64] | L4 [128][128]
    static constexpr int kHss0 = kH0 + 4 * kHItem;
    static constexpr int kHss = kHss0 + 64;
    static constexpr int kHssL1 = 0, kHssL2 = 64, kHssL3 = 192, kHssL4 = 320, kHssFloats = 576;
    // bf16x3 path (encoder_kernel_b3.hip): w = h + m + l exactly, three bf16 fragments of 16 bytes per lane in
    // each wave's consumption order [group][kb][tap][mt_local][plane 3][lane 64][8 bf16]; no weight scale
    static constexpr int kB1 = kHss + kHssFloats;                        // [1][1][9][2][3] items
    static constexpr int kB2 = kB1 + 54 * kHItem;                        // [2][1][9][2][3]
    static constexpr int kB3 = kB2 + 108 * kHItem;                       // [4][2][9][1][3]
    static constexpr int kB4 = kB3 + 216 * kHItem;                       // [4][2][9][2][3]
    static constexpr int kBfcw = kB4 + 432 * kHItem;                     // [4][4][1][2][3]
    static constexpr int kB0 = kBfcw + 96 * kHItem;                      // L0: [mt 2][plane 3][lane 64][e 8]
    // the BatchNorm table the b3 kernel keeps in LDS: L0 [32][32] | L1 [32][32] | L2 [64][64] | L3 | L4 [128][128]
    static constexpr int kBss = kB0 + 6 * kHItem;
    static constexpr int kBssL0 = 0, kBssL1 = 64, kBssL2 = 128, kBssL3 = 256, kBssL4 = 384, kBssFloats = 640;
    static constexpr int kTotal = kBss + kBssFloats;
};

// ---- in-launch hand-off between workgroups (cdna_hip_programming.md, guideline 16) ---------------------------------
// Per-XCD L2s are not coherent with each other and a CU's L1 is not refreshed by other CUs' stores: data one workgroup
// wrote for another workgroup of the SAME launch is published by  every writing wave drains its stores -> barrier ->
// one lane: agent-scope RELEASE fence (+ the drain restated where the compiler cannot drop it) -> relaxed agent-scope
// ticket;  the workgroup that draws the last ticket: one lane agent-scope ACQUIRE fence -> barrier -> plain loads.
__device__ __forceinline__ void handoff_drain_stores() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
__device__ __forceinline__ void handoff_release() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
__device__ __forceinline__ void handoff_acquire() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}

}  // namespace gnnpp
#endif
