// TEST INFRASTRUCTURE ONLY -- a stand-in for <hip/hip_runtime.h> that lets the UNMODIFIED kernel
// sources under gnn_pathplanning_amd/csrc/ be compiled for the x86 host and executed lane by lane
// (one fiber per work-item, 64-lane wavefronts, emulated MFMA / ballot / readlane, real LDS
// semantics).  It exists so that MFMA fragment layouts, LDS indexing, tap skipping and epilogues
// can be checked against the oracle in the CPU-only build container, where no GPU is visible.
// It is ~10^4 x slower than the GPU, is never shipped, and nothing under gnn_pathplanning_amd/
// can load it: it is NOT a fallback path.
//
// v_mfma_f32_16x16x32_f16: lane l holds the k-slots (q = l >> 4, e = 0..7) of row i / column j = l & 15;
// products are exact, the sum is formed in double and rounded once (the hardware's internal order
// is not modelled; tests compare at a tolerance).
// Lane mappings of v_mfma_f32_16x16x4_f32 as documented for gfx950:
//   A: lane l holds A[i = l & 15][k = l >> 4];  B: lane l holds B[k = l >> 4][j = l & 15];
//   D: register r of lane l is D[i = (l >> 4) * 4 + r][j = l & 15].
#ifndef GNNPP_EMU_HIP_RUNTIME_H_
#define GNNPP_EMU_HIP_RUNTIME_H_

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __shared__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct int2 { int x, y; };
typedef void* hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorInvalidValue = 1;
constexpr int hipFuncAttributeMaxDynamicSharedMemorySize = 8;
inline hipError_t hipGetLastError() { return hipSuccess; }
// host API used by the overlapped policy step: in the emulator launches are synchronous and in order
typedef void* hipEvent_t;
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0 };
constexpr unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2;
constexpr int hipMemcpyDeviceToHost = 2;
inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* s) { *s = hipStreamCaptureStatusNone; return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
typedef int hipDevice_t;
inline hipError_t hipStreamGetDevice(hipStream_t, hipDevice_t* d) { *d = 0; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }

namespace gnnpp_emu {
typedef float f4 __attribute__((ext_vector_type(4)));
struct Item {                      // per-fiber identity
    dim3 tid, bid, bdim, gdim;
};
extern Item* cur;                  // the running fiber
void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body);
void sync_block();
unsigned long long ballot(int pred);
int readlane(int v, int src_lane);
float shfl_xor(float v, int mask);   // value of lane (lane ^ mask); every lane of the wave takes part
int mov_dpp_quad(int v, int ctrl);   // v_mov_b32_dpp quad_perm:[ctrl & 3, ctrl >> 2 & 3, ctrl >> 4 & 3, ctrl >> 6 & 3]
void wave_sync();                 // lockstep point: every lane of the wave reaches it before any leaves
f4 mfma16x16x4(float a, float b, f4 c, int, int, int);
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
f4 mfma16x16x32_f16(h8 a, h8 b, f4 c, int, int, int);
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
f4 mfma16x16x32_bf16(b8 a, b8 b, f4 c, int, int, int);
}  // namespace gnnpp_emu

#define threadIdx (gnnpp_emu::cur->tid)
#define blockIdx (gnnpp_emu::cur->bid)
#define blockDim (gnnpp_emu::cur->bdim)
#define gridDim (gnnpp_emu::cur->gdim)

#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) \
    gnnpp_emu::launch((grid), (block), (size_t)(smem), [=]() { kernel(__VA_ARGS__); })

#define __syncthreads() gnnpp_emu::sync_block()
#define __ballot(p) gnnpp_emu::ballot((p) ? 1 : 0)
#define __ffsll(x) __builtin_ffsll(x)
#define __ffs(x) __builtin_ffs(x)
#define __popcll(x) __builtin_popcountll(x)
#define __popc(x) __builtin_popcount(x)
#define __umul24(a, b) ((unsigned)(a) * (unsigned)(b))
#define __mul24(a, b) ((int)(a) * (int)(b))
#define __builtin_amdgcn_readlane(v, l) gnnpp_emu::readlane((v), (l))
#define __shfl_xor(v, m) gnnpp_emu::shfl_xor((v), (m))
// (only the quad_perm controls 0x00..0xff with full row / bank masks are used by the kernels)
#define __builtin_amdgcn_mov_dpp(v, ctrl, row_mask, bank_mask, bound_ctrl) gnnpp_emu::mov_dpp_quad((v), (ctrl))
#define __builtin_amdgcn_mfma_f32_16x16x4f32 gnnpp_emu::mfma16x16x4
#define __builtin_amdgcn_mfma_f32_16x16x32_f16 gnnpp_emu::mfma16x16x32_f16
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 gnnpp_emu::mfma16x16x32_bf16
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(mask, size, sync_id) ((void)0)
#define __builtin_amdgcn_wave_barrier() gnnpp_emu::wave_sync()
#define __builtin_amdgcn_readfirstlane(x) (x)          // only used on wave-uniform values
#define __HIP_MEMORY_SCOPE_WAVEFRONT 1
#define __hip_atomic_load(ptr, order, scope) (*(ptr))
#define __hip_atomic_store(ptr, val, order, scope) (*(ptr) = (val))
// (returns the OLD value, like the builtin; fibers of a workgroup run one at a time, so a plain update is atomic here)
template <class T, class V> inline T gnnpp_emu_fetch_add(T* p, V v) { const T old = *p; *p = (T)(old + v); return old; }
#define __hip_atomic_fetch_add(ptr, val, order, scope) gnnpp_emu_fetch_add((ptr), (val))
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_s_waitcnt(n) ((void)0)
inline long long wall_clock64() { return 0; }
inline void __threadfence() {}
inline void __threadfence_block() {}

// v_perm_b32: byte k of the result is byte sel[k] of the 8-byte value {hi (bytes 4..7), lo (bytes 0..3)}
inline unsigned __builtin_amdgcn_perm(unsigned hi, unsigned lo, unsigned sel) {
    const unsigned long long v = ((unsigned long long)hi << 32) | lo;
    unsigned r = 0;
    for (int k = 0; k < 4; ++k) r |= (unsigned)((v >> (8 * ((sel >> (8 * k)) & 7))) & 0xff) << (8 * k);
    return r;
}
inline float __int_as_float(int v) { float f; std::memcpy(&f, &v, 4); return f; }
inline int __float_as_int(float f) { int v; std::memcpy(&v, &f, 4); return v; }
using std::min;
using std::max;

#endif
