// Issue behaviour of v_mfma_f32_16x16x32_bf16 from ONE wave per SIMD: cycles per MFMA with 1 / 2 / 4 / 8 independent
// accumulators (dependent-accumulate latency), and with two waves per SIMD.   hipcc --offload-arch=gfx950 -O3 ... && ./a.out
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8b __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256) void k(const float* in, float* out, long long* cyc, int iters) {
    v4f x = *reinterpret_cast<const v4f*>(in + threadIdx.x * 4);
    v8b a = __builtin_bit_cast(v8b, x), b = a;
    v4f acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = x;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = clock64();
    v4f s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    *reinterpret_cast<v4f*>(out + (blockIdx.x * 256 + threadIdx.x) * 4) = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
void run(int blocks_per_cu, float* in, float* out, long long* cyc) {
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[4096];
    hipMemcpy(h, cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < grid; ++i) m += (double)h[i];
    m /= grid;
    printf("{\"accumulators\": %d, \"waves_per_simd\": %d, \"cycles_per_mfma_per_wave\": %.2f, \"cycles_per_mfma_per_simd\": %.2f}\n",
           NACC, blocks_per_cu, m / (iters * 8.0 * NACC), m / (iters * 8.0 * NACC) / blocks_per_cu);
}
int main() {
    float *in, *out; long long* cyc;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 4096 * 256 * 16); hipMalloc(&cyc, 4096 * 8);
    hipMemset(in, 0, 4096 * 4);
    for (int w = 1; w <= 2; ++w) { run<1>(w, in, out, cyc); run<2>(w, in, out, cyc); run<3>(w, in, out, cyc); run<4>(w, in, out, cyc); run<8>(w, in, out, cyc); }
    return 0;
}
