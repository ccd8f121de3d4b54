// Measurement probe (not part of the library): facts about v_mfma_f32_16x16x32_f16 on gfx950 that
// the split-f16 encoder relies on: (1) f16 subnormal operands are not flushed, (2) the A/B k-slot
// pairing is (q, e) <-> (q, e), (3) issue rate with independent / dependent accumulators.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void mfma_once(const _Float16* A, const _Float16* B, float* D) {
    // A: [16][32] row-major (i, k); B: [32][16] (k, j); D: [16][16]
    const int l = threadIdx.x, q = l >> 4, ij = l & 15;
    v8h a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = A[ij * 32 + 8 * q + e];
        b[e] = B[(8 * q + e) * 16 + ij];
    }
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + ij] = c[r];
}

__global__ void cvt_probe(const float* in, float* out, int n) {
    const int i = threadIdx.x;
    if (i < n) {
        const _Float16 h = (_Float16)in[i];
        out[i] = (float)h;
    }
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma_rate(float* out, int iters, float seed) {
    v8h a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(seed + threadIdx.x * 0.001f + e); b[e] = (_Float16)(seed * 0.5f + e); }
    v4f acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void mfma_rate_f32(float* out, int iters, float seed) {
    float a = seed + threadIdx.x * 0.001f, b = seed * 0.5f;
    v4f acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <class F>
static float time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    std::vector<_Float16> A(16 * 32), B(32 * 16);
    std::vector<float> D(256), ref(256);
    _Float16 *dA, *dB; float* dD;
    CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dD, 1 << 20));
    // (2) layout: random small integers, exact in fp32
    srand(1);
    for (auto& v : A) v = (_Float16)(float)(rand() % 7 - 3);
    for (auto& v : B) v = (_Float16)(float)(rand() % 5 - 2);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        float s = 0.f;
        for (int k = 0; k < 32; ++k) s += (float)A[i * 32 + k] * (float)B[k * 16 + j];
        ref[i * 16 + j] = s;
    }
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    mfma_once<<<1, 64>>>(dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += D[i] != ref[i];
    printf("{\"probe\": \"layout k=8q+e, D[i=4q+r][j]\", \"mismatches\": %d}\n", bad);
    // (1) subnormals: A = 1 on the diagonal k = i, B = 2^-20 (f16 subnormal) everywhere
    for (auto& v : A) v = (_Float16)0.f;
    for (int i = 0; i < 16; ++i) A[i * 32 + i] = (_Float16)1.f;
    const float sub = ldexpf(1.f, -20);
    for (auto& v : B) v = (_Float16)sub;
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    mfma_once<<<1, 64>>>(dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    printf("{\"probe\": \"subnormal B operand 2^-20 x 1\", \"expected\": %g, \"got\": %g}\n", sub, D[0]);
    for (auto& v : A) v = (_Float16)0.f;
    for (int i = 0; i < 16; ++i) A[i * 32 + i] = (_Float16)ldexpf(3.f, -24);
    for (auto& v : B) v = (_Float16)1024.f;
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    mfma_once<<<1, 64>>>(dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    printf("{\"probe\": \"subnormal A operand 3*2^-24 x 1024\", \"expected\": %g, \"got\": %g}\n",
           ldexpf(3.f, -14), D[0]);
    // product of two subnormal-range values and a tiny product (fp32 accumulate, no flush expected)
    for (auto& v : A) v = (_Float16)0.f;
    for (int i = 0; i < 16; ++i) A[i * 32 + i] = (_Float16)ldexpf(1.f, -14);
    for (auto& v : B) v = (_Float16)ldexpf(1.f, -14);
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    mfma_once<<<1, 64>>>(dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    printf("{\"probe\": \"tiny product 2^-14 x 2^-14\", \"expected\": %g, \"got\": %g}\n", ldexpf(1.f, -28), D[0]);
    // cvt f32 -> f16 of values in the subnormal range
    {
        float in[4] = {1e-6f, 3e-8f, 6.0e-5f, 70000.f}, out[4];
        float *din, *dout;
        CK(hipMalloc(&din, 16)); CK(hipMalloc(&dout, 16));
        CK(hipMemcpy(din, in, 16, hipMemcpyHostToDevice));
        cvt_probe<<<1, 64>>>(din, dout, 4);
        CK(hipMemcpy(out, dout, 16, hipMemcpyDeviceToHost));
        printf("{\"probe\": \"cvt f32->f16->f32\", \"in\": [%g, %g, %g, %g], \"out\": [%g, %g, %g, %g]}\n",
               in[0], in[1], in[2], in[3], out[0], out[1], out[2], out[3]);
    }
    // (3) issue rate: 256 CUs x 1 WG x 4 waves, 12 MFMAs per iteration
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, iters = 20000;
    const double n_per_simd = 12.0 * iters;
    const float t4 = time_ms([&] { mfma_rate<4><<<cus, 256>>>(dD, iters, 1.f); });
    const float t3 = time_ms([&] { mfma_rate<3><<<cus, 256>>>(dD, iters, 1.f); });
    const float t1 = time_ms([&] { mfma_rate<1><<<cus, 256>>>(dD, iters, 1.f); });
    const float t2w = time_ms([&] { mfma_rate<4><<<2 * cus, 256>>>(dD, iters, 1.f); });
    const float tf = time_ms([&] { mfma_rate_f32<<<cus, 256>>>(dD, iters, 1.f); });
    printf("{\"probe\": \"mfma issue\", \"cus\": %d, \"clock_MHz\": %d, \"f16_ns_per_mfma_4acc\": %.3f, "
           "\"f16_ns_per_mfma_3acc\": %.3f, \"f16_ns_per_mfma_1acc_dependent\": %.3f, "
           "\"f16_ns_per_mfma_2waves_per_simd\": %.3f, \"f32_16x16x4_ns_per_mfma\": %.3f}\n",
           cus, prop.clockRate / 1000, t4 * 1e6 / n_per_simd, t3 * 1e6 / n_per_simd,
           t1 * 1e6 / n_per_simd, t2w * 1e6 / (2 * n_per_simd), tf * 1e6 / n_per_simd);
    return 0;
}
