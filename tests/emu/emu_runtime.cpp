// TEST INFRASTRUCTURE ONLY -- fiber scheduler + wavefront collectives for tests/emu/hip/hip_runtime.h.
#include <hip/hip_runtime.h>

#include <sys/mman.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace gnnpp {
alignas(16) char gnnpp_smem[160 * 1024];   // the kernels' `extern __shared__ char gnnpp_smem[]`
}

extern "C" void gnnpp_emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl gnnpp_emu_switch
.type gnnpp_emu_switch,@function
gnnpp_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
)");

namespace gnnpp_emu {

Item* cur = nullptr;

namespace {
constexpr size_t kStack = 512 * 1024;
struct Fiber {
    Item item;
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = false;
    int wave = 0, lane = 0;
};
struct Wave {
    int arrived = 0;
    unsigned gen = 0;
    int n = 0;                     // lanes in this wave
    float a[64], b[64];
    float a8[64][8], b8[64][8];
    int iv[64];
};
std::vector<Fiber> fibers;
std::vector<Wave> waves;
void* sched_sp = nullptr;
int cur_idx = -1;
int block_arrived = 0;
unsigned block_gen = 0;
const std::function<void()>* body_fn = nullptr;

unsigned long progress = 0;            // barriers completed + fibers finished: a scheduler pass without any is a deadlock
int wait_kind[4096];                   // what fiber t is parked on: 1 wave-level collective, 2 workgroup barrier
const char* wait_what[4096];           // ... and which collective
void yield() { gnnpp_emu_switch(&fibers[cur_idx].sp, sched_sp); }

void trampoline() {
    (*body_fn)();
    fibers[cur_idx].done = true;
    yield();
    std::abort();
}

void wave_barrier() {
    Wave& w = waves[fibers[cur_idx].wave];
    const unsigned g = w.gen;
    if (++w.arrived == w.n) {
        w.arrived = 0;
        ++w.gen;
        ++progress;
    } else {
        wait_kind[cur_idx] = 1;
        while (w.gen == g) yield();
        wait_kind[cur_idx] = 0;
    }
}
}  // namespace

void sync_block() {
    const unsigned g = block_gen;
    if (++block_arrived == (int)fibers.size()) {
        block_arrived = 0;
        ++block_gen;
        ++progress;
    } else {
        wait_kind[cur_idx] = 2;
        while (block_gen == g) yield();
        wait_kind[cur_idx] = 0;
    }
}

void wave_sync() { wait_what[cur_idx] = "wave_barrier"; wave_barrier(); }

unsigned long long ballot(int pred) {
    wait_what[cur_idx] = "ballot";
    Fiber& f = fibers[cur_idx];
    Wave& w = waves[f.wave];
    w.iv[f.lane] = pred;
    wave_barrier();
    unsigned long long m = 0;
    for (int l = 0; l < w.n; ++l)
        if (w.iv[l]) m |= 1ull << l;
    wave_barrier();
    return m;
}

int readlane(int v, int src) {
    wait_what[cur_idx] = "readlane";
    Fiber& f = fibers[cur_idx];
    Wave& w = waves[f.wave];
    w.iv[f.lane] = v;
    wave_barrier();
    const int r = w.iv[src];
    wave_barrier();
    return r;
}

float shfl_xor(float v, int mask) {
    wait_what[cur_idx] = "shfl_xor";
    Fiber& f = fibers[cur_idx];
    Wave& w = waves[f.wave];
    w.a[f.lane] = v;
    wave_barrier();
    const float r = w.a[(f.lane ^ mask) & 63];
    wave_barrier();
    return r;
}

int mov_dpp_quad(int v, int ctrl) {
    wait_what[cur_idx] = "mov_dpp";
    if (ctrl < 0 || ctrl > 0xff) { std::fprintf(stderr, "emu: only quad_perm DPP controls are modelled\n"); std::abort(); }
    Fiber& f = fibers[cur_idx];
    Wave& w = waves[f.wave];
    w.iv[f.lane] = v;
    wave_barrier();
    const int src = (f.lane & ~3) | ((ctrl >> (2 * (f.lane & 3))) & 3);
    const int r = w.iv[src];
    wave_barrier();
    return r;
}

f4 mfma16x16x4(float a, float b, f4 c, int, int, int) {
    wait_what[cur_idx] = "mfma f32";
    Fiber& f = fibers[cur_idx];
    Wave& w = waves[f.wave];
    if (w.n != 64) { std::fprintf(stderr, "emu: MFMA needs a full wave\n"); std::abort(); }
    w.a[f.lane] = a;
    w.b[f.lane] = b;
    wave_barrier();
    const int j = f.lane & 15;
    for (int r = 0; r < 4; ++r) {
        const int i = (f.lane >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = std::fmaf(w.a[i + 16 * k], w.b[j + 16 * k], acc);
        c[r] = acc;
    }
    wave_barrier();
    return c;
}

f4 mfma16x16x32_f16(h8 a, h8 b, f4 c, int, int, int) {
    wait_what[cur_idx] = "mfma f16";
    Fiber& f = fibers[cur_idx];
    Wave& w = waves[f.wave];
    if (w.n != 64) { std::fprintf(stderr, "emu: MFMA needs a full wave\n"); std::abort(); }
    for (int e = 0; e < 8; ++e) {
        w.a8[f.lane][e] = (float)a[e];
        w.b8[f.lane][e] = (float)b[e];
    }
    wave_barrier();
    const int j = f.lane & 15;
    for (int r = 0; r < 4; ++r) {
        const int i = (f.lane >> 4) * 4 + r;
        double acc = c[r];
        for (int qq = 0; qq < 4; ++qq)
            for (int e = 0; e < 8; ++e)
                acc += (double)w.a8[i + 16 * qq][e] * (double)w.b8[j + 16 * qq][e];
        c[r] = (float)acc;
    }
    wave_barrier();
    return c;
}

// v_mfma_f32_16x16x32_bf16: same lane map as the f16 form; operands are bf16 bit patterns (the upper half of an
// fp32), products exact, the sum formed in double and rounded once
f4 mfma16x16x32_bf16(b8 a, b8 b, f4 c, int, int, int) {
    wait_what[cur_idx] = "mfma bf16";
    Fiber& f = fibers[cur_idx];
    Wave& w = waves[f.wave];
    if (w.n != 64) { std::fprintf(stderr, "emu: MFMA needs a full wave\n"); std::abort(); }
    for (int e = 0; e < 8; ++e) {
        w.a8[f.lane][e] = (float)a[e];
        w.b8[f.lane][e] = (float)b[e];
    }
    wave_barrier();
    const int j = f.lane & 15;
    for (int r = 0; r < 4; ++r) {
        const int i = (f.lane >> 4) * 4 + r;
        double acc = c[r];
        for (int qq = 0; qq < 4; ++qq)
            for (int e = 0; e < 8; ++e)
                acc += (double)w.a8[i + 16 * qq][e] * (double)w.b8[j + 16 * qq][e];
        c[r] = (float)acc;
    }
    wave_barrier();
    return c;
}

// LDS bounds: everything behind the launch's dynamic shared-memory size is filled with a canary before
// every workgroup and checked afterwards -- on the GPU an out-of-range LDS write is silently dropped
// (r02: an epilogue buffer that outgrew a tiny graph's allocation passed here and failed there).
static void lds_canary(size_t smem_bytes, bool check) {
    constexpr size_t kLds = 160 * 1024;
    char* base = gnnpp::gnnpp_smem;
    for (size_t i = smem_bytes < kLds ? smem_bytes : kLds; i < kLds; ++i) {
        if (!check) base[i] = (char)0xA5;
        else if (base[i] != (char)0xA5) {
            std::fprintf(stderr, "gnnpp emu: LDS write at byte %zu beyond the %zu bytes this launch allocated\n",
                         i, smem_bytes);
            std::abort();
        }
    }
}

void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body) {
    const int nt = (int)block.x;
    static char* arena = nullptr;
    static size_t arena_sz = 0;
    if (arena_sz < (size_t)nt * kStack) {
        arena_sz = (size_t)nt * kStack;
        arena = (char*)mmap(nullptr, arena_sz, PROT_READ | PROT_WRITE,
                            MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (arena == MAP_FAILED) { std::perror("mmap"); std::abort(); }
    }
    body_fn = &body;
    for (unsigned bb = 0; bb < grid.x * grid.y * grid.z; ++bb) {
        const unsigned b = bb % grid.x, by = (bb / grid.x) % grid.y, bz = bb / (grid.x * grid.y);
        lds_canary(smem_bytes, false);
        fibers.assign(nt, Fiber());
        waves.assign((nt + 63) / 64, Wave());
        block_arrived = 0;
        block_gen = 0;
        for (int t = 0; t < nt; ++t) {
            Fiber& f = fibers[t];
            f.item.tid = dim3(t, 0, 0);
            f.item.bid = dim3(b, by, bz);
            f.item.bdim = block;
            f.item.gdim = grid;
            f.wave = t / 64;
            f.lane = t % 64;
            waves[f.wave].n++;
            f.stack = arena + (size_t)t * kStack;
            uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
            void** sp = (void**)top;
            *--sp = nullptr;                       // fake return address of trampoline
            *--sp = (void*)&trampoline;            // `ret` target of the first switch
            for (int r = 0; r < 6; ++r) *--sp = nullptr;
            f.sp = sp;
        }
        int live = nt;
        while (live > 0) {
            const unsigned long before = progress;
            for (int t = 0; t < nt; ++t) {
                if (fibers[t].done) continue;
                cur_idx = t;
                cur = &fibers[t].item;
                gnnpp_emu_switch(&sched_sp, fibers[t].sp);
                if (fibers[t].done) { --live; ++progress; }
            }
            if (live > 0 && progress == before) {
                // every live work-item is parked and nothing completed: divergent barriers / collectives (on the GPU:
                // a hang, or lanes silently missing from a ballot)
                int nw = 0, nb = 0, first_w = -1, first_b = -1;
                for (int t = 0; t < nt; ++t) {
                    if (fibers[t].done) continue;
                    if (wait_kind[t] == 1) { if (first_w < 0) first_w = t; ++nw; }
                    if (wait_kind[t] == 2) { if (first_b < 0) first_b = t; ++nb; }
                }
                std::fprintf(stderr, "gnnpp emu: DEADLOCK in workgroup %u: %d work-items finished, %d parked on a wave "
                             "collective (first: thread %d, in %s), %d on __syncthreads (first: thread %d)\n", bb,
                             nt - live, nw, first_w, first_w >= 0 && wait_what[first_w] ? wait_what[first_w] : "?", nb,
                             first_b);
                for (int t = 0; t < nt; ++t)
                    if (!fibers[t].done && wait_kind[t] == 1 && (t % 64 == 0 || wait_kind[t - 1] != 1))
                        std::fprintf(stderr, "   thread %d.. : %s\n", t, wait_what[t] ? wait_what[t] : "?");
                std::abort();
            }
        }
        lds_canary(smem_bytes, true);
    }
    cur = nullptr;
    cur_idx = -1;
}

}  // namespace gnnpp_emu
