// Measurement probe: rate at which the waves of a CU can stream 1 KiB weight fragments out of L2
// through a register ring (the encoder's weight stream), 8-byte ordered loads vs 16-byte loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

// 16-byte loads pinned in program order: (a) buffer load with the compiler-level volatile bit
// (encoded with sc0 sc1), (b) inline asm global_load_dwordx4 + explicit s_waitcnt
template <int RING, int MODE>
__global__ __launch_bounds__(256) void stream_kernel_pinned(const float* __restrict__ w, float* out,
                                                            int items, int wrap_items) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* base = w + (size_t)wave * wrap_items * 256 + lane * 4;
    __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base - lane * 4), 0, 0x7fffffff, 0x00020000);
    v4f ring[RING];
    auto ld = [&](int idx) {
        const int item = idx % wrap_items;
        if (MODE == 0) {
            ring[idx % RING] = __builtin_bit_cast(
                v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + item * 1024, 0, (int)0x80000000));
        } else {
            const float* p = base + (size_t)item * 256;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ring[idx % RING]) : "v"(p) : "memory");
        }
    };
#pragma unroll
    for (int i = 0; i < RING; ++i) ld(i);
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < items; it += RING) {
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            if (MODE == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ring[u]) : "n"(RING - 1));
            acc += ring[u];
            ld(it + u + RING);
        }
    }
    if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int RING, bool X4>
__global__ __launch_bounds__(256) void stream_kernel(const float* __restrict__ w, float* out, int items,
                                                     int wrap_items) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // every wave walks its own segment (as the encoder: per-wave weight streams), all WGs the same
    const float* base = w + (size_t)wave * wrap_items * 256 + lane * 4;
    v4f ring[RING];
    auto ld = [&](int idx) {
        const float* p = base + (size_t)(idx % wrap_items) * 256;
        if (X4) {
            ring[idx % RING] = *reinterpret_cast<const v4f*>(p);
        } else {
            typedef unsigned long long u64;
            u64* q = reinterpret_cast<u64*>(const_cast<float*>(p));
            const u64 lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            const u64 hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            v4f r;
            r[0] = __int_as_float((int)(lo & 0xffffffffu)); r[1] = __int_as_float((int)(lo >> 32));
            r[2] = __int_as_float((int)(hi & 0xffffffffu)); r[3] = __int_as_float((int)(hi >> 32));
            ring[idx % RING] = r;
        }
    };
#pragma unroll
    for (int i = 0; i < RING; ++i) ld(i);
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < items; it += RING) {
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            acc += ring[u];
            ld(it + u + RING);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <class F>
static float time_ms(F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

int main() {
    const int wrap = 256;                       // 256 KiB per wave segment, 1 MiB per WG: L2 resident
    float *w, *out;
    (void)hipMalloc(&w, (size_t)4 * wrap * 256 * 4);
    (void)hipMemset(w, 0, (size_t)4 * wrap * 256 * 4);
    (void)hipMalloc(&out, 1 << 22);
    const int items = 4096;                     // 4 MiB per wave, 16 MiB per WG
    const double bytes_per_wg = 4.0 * items * 1024.0;
    for (int wgs : {256, 512}) {
        const float a = time_ms([&] { stream_kernel<16, false><<<wgs, 256>>>(w, out, items, wrap); });
        const float b = time_ms([&] { stream_kernel<16, true><<<wgs, 256>>>(w, out, items, wrap); });
        const float c = time_ms([&] { stream_kernel<8, false><<<wgs, 256>>>(w, out, items, wrap); });
        const float d = time_ms([&] { stream_kernel<32, true><<<wgs, 256>>>(w, out, items, wrap); });
        const float e = time_ms([&] { stream_kernel_pinned<16, 0><<<wgs, 256>>>(w, out, items, wrap); });
        const float f = time_ms([&] { stream_kernel_pinned<16, 1><<<wgs, 256>>>(w, out, items, wrap); });
        const double per_cu = bytes_per_wg * (wgs / 256.0);
        printf("{\"probe\": \"weight stream pinned x4\", \"wgs\": %d, \"GBps_per_CU_buffer_volatile_sc0sc1\": %.1f, "
               "\"GBps_per_CU_asm_global_load\": %.1f}\n", wgs, per_cu / (e * 1e6), per_cu / (f * 1e6));
        printf("{\"probe\": \"weight stream\", \"wgs\": %d, \"GBps_per_CU_ring16_x2\": %.1f, "
               "\"GBps_per_CU_ring16_x4\": %.1f, \"GBps_per_CU_ring8_x2\": %.1f, \"GBps_per_CU_ring32_x4\": %.1f}\n",
               wgs, per_cu / (a * 1e6), per_cu / (b * 1e6), per_cu / (c * 1e6), per_cu / (d * 1e6));
    }
    return 0;
}
