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
// ---- numerics: does v_mfma_f32_16x16x32_bf16 flush SUBNORMAL bf16 operands / subnormal fp32 results, and does
// v_cvt_pk_bf16_f32 keep subnormal values?  (The bf16x3 operand split relies on all three: the m / l planes of a small
// activation are bf16 subnormals.)  One wave; lane (q, i) holds k-slots (q, e); only k-slot (0, 0) is non-zero.
__global__ void numerics(const unsigned* abits, const unsigned* bbits, const float* cvt_in, float* out, unsigned* cvt_out,
                         int ncase) {
    const int lane = threadIdx.x;
    for (int c = 0; c < ncase; ++c) {
        typedef unsigned short v8u __attribute__((ext_vector_type(8)));
        v8u a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
        if (lane < 16) { a[0] = (unsigned short)abits[c]; b[0] = (unsigned short)bbits[c]; }   // k-slot (q = 0, e = 0)
        v4f z = {0.f, 0.f, 0.f, 0.f};
        const v4f d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8b, a), __builtin_bit_cast(v8b, b), z, 0, 0, 0);
        if (lane == 0) out[c] = d[0];                       // D[0][0] = a * b
    }
    if (lane < 8) {
        unsigned r;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(cvt_in[2 * lane]), "v"(cvt_in[2 * lane + 1]));
        cvt_out[lane] = r;
    }
}
static void run_numerics() {
    // (a bits, b bits, what): bf16 bit patterns; 0x0001 = 2^-133 (smallest subnormal), 0x0040 = 2^-127, 0x0080 = 2^-126
    // (smallest normal), 0x7100 = 2^99, 0x3f80 = 1, 0x1000 = 2^-95, 0x2c80 = 2^-38
    const unsigned A[] = {0x0001, 0x0040, 0x0080, 0x0001, 0x1000, 0x0040};
    const unsigned B[] = {0x7100, 0x7100, 0x7100, 0x3f80, 0x2c80, 0x0040};
    const char* what[] = {"subnormal operand 2^-133 x 2^99 (want 2^-34)", "subnormal operand 2^-127 x 2^99 (want 2^-28)",
                          "smallest normal 2^-126 x 2^99 (want 2^-27)", "subnormal operand x 1 -> subnormal fp32 result (want 2^-133)",
                          "normal x normal -> subnormal fp32 result 2^-95 x 2^-38 (want 2^-133)",
                          "subnormal x subnormal 2^-127 x 2^-127 (want 0: underflow)"};
    const int n = 6;
    unsigned *da, *db, *dc; float *din, *dout;
    hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dc, 64); hipMalloc(&din, 64); hipMalloc(&dout, 64);
    hipMemcpy(da, A, n * 4, hipMemcpyHostToDevice); hipMemcpy(db, B, n * 4, hipMemcpyHostToDevice);
    // conversions: fp32 values whose bf16 is subnormal / the fp32 itself is subnormal
    unsigned cin_bits[16] = {0x00400000 /* 2^-127 */, 0x00010000 /* 2^-133 */, 0x00008000 /* 2^-134: ties to even -> 0 */,
                             0x0000c000 /* 1.5 x 2^-134 -> 2^-133 */, 0x00000001 /* 2^-149 */, 0x007fffff, 0x3f800000, 0x80010000};
    hipMemcpy(din, cin_bits, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(numerics, dim3(1), dim3(64), 0, 0, da, db, din, dout, dc, n);
    hipDeviceSynchronize();
    float ho[16]; unsigned hc[16];
    hipMemcpy(ho, dout, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hc, dc, 32, hipMemcpyDeviceToHost);
    for (int c = 0; c < n; ++c) {
        unsigned u; __builtin_memcpy(&u, &ho[c], 4);
        printf("{\"probe\": \"bf16 mfma numerics\", \"case\": \"%s\", \"a_bits\": \"0x%04x\", \"b_bits\": \"0x%04x\", "
               "\"result\": %.9g, \"result_bits\": \"0x%08x\"}\n", what[c], A[c], B[c], ho[c], u);
    }
    for (int i = 0; i < 4; ++i)
        printf("{\"probe\": \"v_cvt_pk_bf16_f32\", \"in_bits\": [\"0x%08x\", \"0x%08x\"], \"out_bits\": \"0x%08x\"}\n",
               cin_bits[2 * i], cin_bits[2 * i + 1], hc[i]);
}
int main() {
    run_numerics();
    float *in, *out; long long* cyc;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 4096 * 256 * 16); hipMalloc(&cyc, 4096 * 8);
    hipMemset(in, 0, 4096 * 4);
    for (int w = 1; w <= 2; ++w) { run<1>(w, in, out, cyc); run<2>(w, in, out, cyc); run<3>(w, in, out, cyc); run<4>(w, in, out, cyc); run<8>(w, in, out, cyc); }
    for (int w = 3; w <= 4; ++w) { run<2>(w, in, out, cyc); run<4>(w, in, out, cyc); }   // three and four waves per SIMD
    return 0;
}
