// Weight-fragment register ring of the exact-fp32 encoder schedule (encoder_kernel_f32.hip).
//
// Every wave consumes a fixed sequence of packed A fragments (1 KiB each, L2 resident) from L1 to the
// FC.  In the 2x2 layers a fragment feeds only 4..16 MFMAs, far less than the L2 latency, and the
// compiler keeps 1-2 loads in flight; here they stream through a 12-deep register ring (48 VGPRs)
// that runs ahead of the MFMAs ACROSS layer boundaries.  Traversal is group-major (g, then tap) so
// consecutive fragments of the 2x2 layers hit different accumulators (no dependent MFMA chains).
#include "gnnpp_common.h"

namespace gnnpp {

constexpr int kRing = 12;
// sched_barrier mask: ALU | VALU | SALU | DS | DS read | DS write | transcendental may cross
constexpr int kSchedItemMask = 0x1 | 0x2 | 0x4 | 0x80 | 0x100 | 0x200 | 0x400;
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

}  // namespace gnnpp
