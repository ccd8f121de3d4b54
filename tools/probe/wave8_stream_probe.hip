// would ONE 16-agent tile / one graph across EIGHT waves of <= 128 VGPRs (two such workgroups per
// CU = 4 waves per SIMD) run the column-packed L1 / L2 streams of encoder_kernel_b3 faster than today's four
// 256-VGPR waves (2 per SIMD)?  Stand-alone model of the stream with its REAL traffic: a "step" is one (kb, tap) of a
// layer -- NM x 3 weight-plane fragments (1 KiB each per wave) come off a register ring refilled from L2, then NT
// units of [three ds_read_b128 B planes (double-buffered, issued behind the unit's first MFMAs) + NM x 6 MFMAs
// v_mfma_f32_16x16x32_bf16 on NM accumulators].  Both workgroup shapes do the SAME matrix work per CU.
//
//   form               NW  NT  NM   what a wave pair shares                    per-WG traffic vs today
//   today  (L1)         4   4   2   --                                         1x ring, 1x LDS
//   mt-split (L1)       8   4   1   B planes (both waves read them)            1x ring, 2x LDS reads
//   tile-split (L1)     8   2   2   A fragments (both waves stream them)       2x ring, 1x LDS reads
//   today  (L2, N=10)   4   5   2
//   mt-split (L2)       8   5   1                                              1x ring, 2x LDS reads
// MODE bits: 1 = LDS reads, 2 = ring.  Reported: shader cycles per MFMA and SIMD (s_memtime) and the WALL time of
// the launch (hipEvents) -- DVFS stretches cycles under matrix load, so cycles alone can flatter a form.
//   hipcc --offload-arch=gfx950 -O3 -o wave8_stream_probe wave8_stream_probe.hip && ./wave8_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8b __attribute__((ext_vector_type(8)));
__device__ __forceinline__ v8b b8(v4f x) { return __builtin_bit_cast(v8b, x); }
__device__ __forceinline__ v4f mf(v8b a, v8b b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

constexpr int kLdsBytes = 75 * 1024;

template <int NW, int NT, int NM, int SLOTS, int MODE>
__global__ __launch_bounds__(NW * 64, NW / 2) void k(const float* __restrict__ w, float* out, long long* cyc, int steps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NA = NM * 3;                 // fragments a step takes off the ring
    constexpr int PH = SLOTS / NA;             // steps per ring revolution
    static_assert(SLOTS % NA == 0 && PH >= 2, "ring");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 48 * 1024 / 16; i += NW * 64)
        reinterpret_cast<v4f*>(smem)[i] = v4f{1.f, 2.f, 3.f, 4.f} * (float)(i & 7);
    __syncthreads();
    // waves of a pair (NW = 8) stream DIFFERENT halves (mt-split) or the same items (tile-split: same w4 base)
    const int sw = (NW == 8 && NM == 2) ? (wave & 3) : wave;
    const v4f* w4 = reinterpret_cast<const v4f*>(w) + sw * 64 * 64 + lane;        // item i of this wave: w4[i * 64]
    v4f ring[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) ring[i] = w4[i * 64];
    int addr[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        // mt-split pairs read the SAME tiles; 16 rows of 3 KiB hold the layer's input
        const int tile = ((NW == 8 && NM == 1) ? (wave & 3) : (wave % 4)) + 4 * i;
        addr[i] = (tile % 16) * 3072 + (lane >> 4) * 256 + (lane & 15) * 16;
        asm volatile("" : "+v"(addr[i]));
    }
    v4f acc[NT][NM];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[i][m] = v4f{0.f, 0.f, 0.f, 0.f};
    v4f Bb[2][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) Bb[0][p] = Bb[1][p] = *reinterpret_cast<const v4f*>(smem + addr[0] + p * 1024);
    v8b A[NM][3];
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int p = 0; p < 3; ++p) A[m][p] = b8(ring[m * 3 + p]);
    __syncthreads();
    const long long t0 = clock64();
    int item = SLOTS;
    for (int st = 0; st < steps; st += PH) {
#pragma unroll
        for (int ph = 0; ph < PH; ++ph) {
            if (MODE & 2) {
#pragma unroll
                for (int m = 0; m < NM; ++m)
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        A[m][p] = b8(ring[ph * NA + m * 3 + p]);
                        ring[ph * NA + m * 3 + p] = w4[((item++) & 63) * 64];
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int u = ph * NT + i;
                v4f (&B)[3] = Bb[u & 1];
                v4f (&Bn)[3] = Bb[(u + 1) & 1];
#pragma unroll
                for (int t = 0; t < 6; ++t) {
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        acc[i][m] = mf(A[m][t % 3], b8(B[t % 3]), acc[i][m]);
                        __builtin_amdgcn_sched_barrier(0);
                        if ((MODE & 1) && m == 0 && t < 3) {            // next unit's planes behind this unit's first MFMAs
                            Bn[t] = *reinterpret_cast<const v4f*>(smem + addr[(i + 1) % NT] + t * 1024);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
        }
    }
    const long long t1 = clock64();
    v4f s = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int m = 0; m < NM; ++m) s += acc[i][m];
    *reinterpret_cast<v4f*>(out + ((size_t)blockIdx.x * NW * 64 + tid) * 4) = s;
    if (lane == 0) cyc[blockIdx.x * NW + wave] = t1 - t0;
}

template <int NW, int NT, int NM, int SLOTS, int MODE>
void run(const char* form, int grid, float* w, float* out, long long* cyc) {
    // equal matrix work per workgroup: 400 steps x (4 x 4 x 2 x 6 = 192 MFMAs) for the L1 shapes
    const int steps = 480;                                     // a multiple of every PH (2, 4)
    auto fn = k<NW, NT, NM, SLOTS, MODE>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(fn));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<float> ms;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(fn, dim3(grid), dim3(NW * 64), kLdsBytes, 0, w, out, cyc, steps);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float t;
        hipEventElapsedTime(&t, e0, e1);
        if (rep) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    std::vector<long long> h((size_t)grid * NW);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double m = 0;
    for (long long c : h) m += (double)c;
    m /= (double)h.size();
    const double mfma_per_wave = (double)steps * NT * NM * 6;
    const int wg_per_cu = grid >= 512 ? 2 : 1;
    const int wps = wg_per_cu * NW / 4;
    const double us = ms[ms.size() / 2] * 1e3;
    const double total_mfma = mfma_per_wave * NW * grid;
    printf("{\"probe\": \"wave8 stream\", \"form\": \"%s\", \"waves_per_wg\": %d, \"units\": %d, \"mt\": %d, \"ring_slots\": %d, "
           "\"lds_reads\": %d, \"ring\": %d, \"workgroups\": %d, \"waves_per_simd\": %d, \"vgprs\": %d, \"scratch_bytes\": %d, "
           "\"cycles_per_mfma_per_wave\": %.2f, \"cycles_per_mfma_per_simd\": %.2f, \"launch_us\": %.1f, "
           "\"mfma_per_us_per_cu\": %.1f, \"pflops\": %.3f, \"eff_clock_ghz\": %.3f}\n",
           form, NW, NT, NM, SLOTS, MODE & 1, (MODE >> 1) & 1, grid, wps, fa.numRegs, (int)fa.localSizeBytes,
           m / mfma_per_wave, m / mfma_per_wave / wps, us, total_mfma / us / 256.0,
           total_mfma * 16384.0 / us * 1e-9, m / us * 1e-3);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main() {
    float *w, *out;
    long long* cyc;
    hipMalloc(&w, 8 * 64 * 64 * 16);
    hipMemset(w, 0, 8 * 64 * 64 * 16);
    hipMalloc(&out, (size_t)1024 * 512 * 16);
    hipMalloc(&cyc, 8192 * sizeof(long long));
    for (int grid : {256, 512}) {
        // bare MFMAs
        run<4, 4, 2, 12, 0>("today L1", grid, w, out, cyc);
        run<8, 4, 1, 12, 0>("mt-split L1", grid, w, out, cyc);
        // + LDS reads only
        run<4, 4, 2, 12, 1>("today L1", grid, w, out, cyc);
        run<8, 4, 1, 12, 1>("mt-split L1", grid, w, out, cyc);
        run<8, 2, 2, 12, 1>("tile-split L1", grid, w, out, cyc);
        // + ring only
        run<4, 4, 2, 12, 2>("today L1", grid, w, out, cyc);
        run<8, 4, 1, 12, 2>("mt-split L1", grid, w, out, cyc);
        run<8, 4, 1, 6, 2>("mt-split L1", grid, w, out, cyc);
        // the real stream: both
        run<4, 4, 2, 12, 3>("today L1", grid, w, out, cyc);
        run<8, 4, 1, 12, 3>("mt-split L1", grid, w, out, cyc);
        run<8, 4, 1, 6, 3>("mt-split L1", grid, w, out, cyc);
        run<8, 2, 2, 12, 3>("tile-split L1", grid, w, out, cyc);
        run<4, 5, 2, 12, 3>("today L2", grid, w, out, cyc);
        run<8, 5, 1, 12, 3>("mt-split L2", grid, w, out, cyc);
        run<8, 5, 1, 6, 3>("mt-split L2", grid, w, out, cyc);
    }
    return 0;
}
