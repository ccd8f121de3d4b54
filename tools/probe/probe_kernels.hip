// Measurement-only kernels (tools/dualpipe_probe.py); not part of the product library.
#include <hip/hip_runtime.h>

// Pure-VALU fp32 FMA burner: 8 independent chains per thread, no memory traffic.
__global__ __launch_bounds__(256) void valu_burn_kernel(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,
          a6 = a0 + 6, a7 = a0 + 7;
    const float m = 0.999f, c = 0.001f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
            a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

extern "C" int probe_valu_burn(float* out, int blocks, int iters, void* stream) {
    hipLaunchKernelGGL(valu_burn_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters,
                       1.0f);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// One wave that waits `us` microseconds: mode 0 = s_sleep only, 1 = s_sleep + relaxed atomic load
// of *flag per iteration, 2 = busy loop reading the clock (no sleep).
__global__ void spin_kernel(const unsigned long long* flag, int us, int mode, unsigned long long* sink) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();                           // 100 MHz
    unsigned long long acc = 0;
    while (wall_clock64() - t0 < (long long)us * 100) {
        if (mode != 2) __builtin_amdgcn_s_sleep(32);
        if (mode == 1) acc += __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    *sink = acc;
}

extern "C" int probe_spin(const void* flag, int us, int mode, void* sink, void* stream) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                       (const unsigned long long*)flag, us, mode, (unsigned long long*)sink);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
