// What does the instruction stream of the column-packed L1 / L2 cost on ONE wave per SIMD, piece by piece?
// A "step" = 4 units of 12 v_mfma_f32_16x16x32_bf16 (two accumulators per unit, alternating), as encoder_kernel_b3<.., CP>
// issues them.  MODE bits: 1 = every unit's three B planes come from LDS (double-buffered ds_read_b128, per-lane address
// registers), 2 = every step's six A fragments come off a 12-slot register ring refilled from global memory (L2-warm),
// 4 = four accumulators in rotation instead of two (two units interleaved), 8 = the three reads behind the first MFMAs
// instead of in front of the unit.
//   hipcc --offload-arch=gfx950 -O3 -o cp_stream_probe cp_stream_probe.hip && ./cp_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8b __attribute__((ext_vector_type(8)));
__device__ __forceinline__ v8b b8(v4f x) { return __builtin_bit_cast(v8b, x); }
__device__ __forceinline__ v4f mf(v8b a, v8b b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ w, float* out, long long* cyc, int steps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 48 * 1024 / 16; i += 256) reinterpret_cast<v4f*>(smem)[i] = v4f{1.f, 2.f, 3.f, 4.f} * (float)(i & 7);
    __syncthreads();
    const v4f* w4 = reinterpret_cast<const v4f*>(w) + wave * 64 * 64 + lane;      // item i of this wave: w4[i * 64]
    v4f ring[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) ring[i] = w4[i * 64];
    int addr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        addr[i] = (wave + 4 * i) * 3072 + (lane >> 4) * 256 + (lane & 15) * 16;
        asm volatile("" : "+v"(addr[i]));
    }
    v4f acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = v4f{0.f, 0.f, 0.f, 0.f};
    v4f Bb[2][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) Bb[0][p] = Bb[1][p] = *reinterpret_cast<const v4f*>(smem + addr[0] + p * 1024);
    v8b A[2][3];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int p = 0; p < 3; ++p) A[m][p] = b8(ring[m * 3 + p]);
    __syncthreads();
    const long long t0 = clock64();
    int item = 12;
    for (int st = 0; st < steps; st += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (MODE & 2) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        A[m][p] = b8(ring[half * 6 + m * 3 + p]);
                        ring[half * 6 + m * 3 + p] = w4[((item++) & 63) * 64];
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE & 4) {
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    v4f B0[3], B1[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        B0[p] = (MODE & 1) ? *reinterpret_cast<const v4f*>(smem + addr[i] + p * 1024) : Bb[0][p];
                        B1[p] = (MODE & 1) ? *reinterpret_cast<const v4f*>(smem + addr[i + 1] + p * 1024) : Bb[1][p];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 6; ++t) {
                        acc[i][0] = mf(A[0][t % 3], b8(B0[t % 3]), acc[i][0]);
                        __builtin_amdgcn_sched_barrier(0);
                        acc[i][1] = mf(A[1][t % 3], b8(B0[t % 3]), acc[i][1]);
                        __builtin_amdgcn_sched_barrier(0);
                        acc[i + 1][0] = mf(A[0][t % 3], b8(B1[t % 3]), acc[i + 1][0]);
                        __builtin_amdgcn_sched_barrier(0);
                        acc[i + 1][1] = mf(A[1][t % 3], b8(B1[t % 3]), acc[i + 1][1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int u = half * 4 + i;
                    v4f (&B)[3] = Bb[u & 1];
                    v4f (&Bn)[3] = Bb[(u + 1) & 1];
                    if ((MODE & 1) && !(MODE & 8)) {
#pragma unroll
                        for (int p = 0; p < 3; ++p) Bn[p] = *reinterpret_cast<const v4f*>(smem + addr[(i + 1) & 3] + p * 1024);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int t = 0; t < 6; ++t) {
                        acc[i][0] = mf(A[0][t % 3], b8(B[t % 3]), acc[i][0]);
                        __builtin_amdgcn_sched_barrier(0);
                        if ((MODE & 1) && (MODE & 8) && t < 3) {
                            Bn[t] = *reinterpret_cast<const v4f*>(smem + addr[(i + 1) & 3] + t * 1024);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        acc[i][1] = mf(A[1][t % 3], b8(B[t % 3]), acc[i][1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    }
    const long long t1 = clock64();
    v4f s = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1];
    *reinterpret_cast<v4f*>(out + (blockIdx.x * 256 + tid) * 4) = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE>
void run(int grid, float* w, float* out, long long* cyc) {
    const int steps = 400;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 75 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 75 * 1024, 0, w, out, cyc, steps);
        hipDeviceSynchronize();
    }
    static long long h[4096];
    hipMemcpy(h, cyc, grid * 4 * sizeof(long long), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < grid * 4; ++i) m += (double)h[i];
    m /= grid * 4;
    printf("{\"probe\": \"cp stream\", \"lds_reads\": %d, \"ring\": %d, \"accumulators\": %d, \"reads_behind_mfmas\": %d, "
           "\"workgroups\": %d, \"cycles_per_mfma_per_wave\": %.2f}\n",
           MODE & 1, (MODE >> 1) & 1, (MODE & 4) ? 4 : 2, (MODE >> 3) & 1, grid, m / (steps * 48.0));
}

int main() {
    float *w, *out;
    long long* cyc;
    hipMalloc(&w, 4 * 64 * 64 * 16);
    hipMemset(w, 0, 4 * 64 * 64 * 16);
    hipMalloc(&out, 1024 * 256 * 16);
    hipMalloc(&cyc, 4096 * sizeof(long long));
    for (int grid : {1, 256, 512}) {
        run<0>(grid, w, out, cyc);
        run<1>(grid, w, out, cyc);
        run<9>(grid, w, out, cyc);
        run<2>(grid, w, out, cyc);
        run<3>(grid, w, out, cyc);
        run<4>(grid, w, out, cyc);
        run<5>(grid, w, out, cyc);
        run<7>(grid, w, out, cyc);
        run<11>(grid, w, out, cyc);
    }
    return 0;
}
