// Measurement probe: can a stream be gated on a device-side counter without occupying CUs
// (hipStreamWaitValue64 on signal memory), and how long after the counter reaches the value does
// the gated kernel start?  Used to judge a resource-free gate for the filter/encoder-tail overlap.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"probe\": \"waitvalue\", \"error\": \"%s at line %d\"}\n", hipGetErrorString(e_), __LINE__); return 0; } } while (0)

__global__ void producer(unsigned long long* sig, long long* t_done, int spin_us_lo, int spin_us_hi) {
    const long long t0 = wall_clock64();                       // 100 MHz
    const int us = (blockIdx.x & 1) ? spin_us_hi : spin_us_lo;
    while (wall_clock64() - t0 < (long long)us * 100) {}
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        t_done[blockIdx.x] = wall_clock64();
        atomicAdd_system(sig, 1ull);
    }
}

__global__ void consumer(long long* t_start) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *t_start = wall_clock64();
}

int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    if (!can) { printf("{\"probe\": \"waitvalue\", \"supported\": false}\n"); return 0; }
    unsigned long long* sig;
    CK(hipExtMallocWithFlags(reinterpret_cast<void**>(&sig), 8, hipMallocSignalMemory));
    CK(hipMemset(sig, 0, 8));
    long long *t_done, *t_start;
    const int nblk = 256;
    CK(hipMalloc(&t_done, nblk * 8));
    CK(hipMalloc(&t_start, 8));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    unsigned long long expected = 0;
    for (int rep = 0; rep < 5; ++rep) {
        for (int mode = 0; mode < 2; ++mode) {                // 0: wait for all blocks, 1: for the early half
            const unsigned long long thr = expected + (mode ? nblk / 2 : nblk);
            hipLaunchKernelGGL(producer, dim3(nblk), dim3(64), 0, s1, sig, t_done, 20, 40);
            CK(hipStreamWaitValue64(s2, sig, thr, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull));
            hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, s2, t_start);
            CK(hipStreamSynchronize(s1));
            CK(hipStreamSynchronize(s2));
            expected += nblk;
            long long h_done[nblk], h_start;
            CK(hipMemcpy(h_done, t_done, sizeof(h_done), hipMemcpyDeviceToHost));
            CK(hipMemcpy(&h_start, t_start, 8, hipMemcpyDeviceToHost));
            // time at which the counter reached the threshold: the (thr - base)-th smallest t_done
            long long sorted[nblk];
            for (int i = 0; i < nblk; ++i) sorted[i] = h_done[i];
            for (int i = 0; i < nblk; ++i) for (int j = i + 1; j < nblk; ++j) if (sorted[j] < sorted[i]) { long long t = sorted[i]; sorted[i] = sorted[j]; sorted[j] = t; }
            const long long t_reach = sorted[(mode ? nblk / 2 : nblk) - 1];
            printf("{\"probe\": \"waitvalue\", \"supported\": true, \"mode\": \"%s\", \"gate_to_start_us\": %.2f, \"last_block_to_start_us\": %.2f}\n",
                   mode ? "early_half" : "all", (h_start - t_reach) / 100.0, (h_start - sorted[nblk - 1]) / 100.0);
        }
    }
    return 0;
}
