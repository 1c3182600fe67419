// TEST INFRASTRUCTURE ONLY - fiber scheduler behind tests/emu/mst_rt.h (see the header there).
// One OS thread runs one HIP workgroup at a time: every HIP thread is a fiber with its own stack;
// __syncthreads() and the wave collectives (MFMA, shuffles) are rendezvous points where the fiber
// yields to the scheduler.  Workgroups are spread over OS threads (std::thread) for speed.
#include "mst_rt.h"

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

thread_local emu_uint3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

extern "C" void emu_ctx_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl emu_ctx_switch
.type emu_ctx_switch,@function
emu_ctx_switch:
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
.size emu_ctx_switch, .-emu_ctx_switch
)");

double emu_now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

namespace emu {

enum State { RUNNABLE, WAIT_BLOCK, WAIT_WAVE, DONE };
constexpr size_t STACK_BYTES = 256 * 1024;
constexpr int MAX_THREADS = 1024;

struct Fiber {
    void *sp;
    State st;
    unsigned char *stack;
};

struct WaveX {
    alignas(64) unsigned char slot[2][64 * 64];
    int parity;     // slot the NEXT publish writes
    int arrived;
    int live;       // fibers of this wave not DONE
};

struct Sched {
    Fiber fib[MAX_THREADS];
    WaveX wave[MAX_THREADS / 64];
    void *main_sp;
    int nthreads, cur;
    int block_arrived, block_live;
    const std::function<void()> *body;
    std::vector<unsigned char *> stacks;
};
static thread_local Sched *S = nullptr;

static void yield_to_main() { emu_ctx_switch(&S->fib[S->cur].sp, S->main_sp); }

static void fiber_entry() {
    (*S->body)();
    Fiber &f = S->fib[S->cur];
    f.st = DONE;
    S->block_live--;
    WaveX &w = S->wave[S->cur / 64];
    w.live--;
    // a finished thread no longer takes part in barriers: release waiters if it was the last one missing
    if (S->block_live > 0 && S->block_arrived == S->block_live) {
        for (int i = 0; i < S->nthreads; ++i)
            if (S->fib[i].st == WAIT_BLOCK) S->fib[i].st = RUNNABLE;
        S->block_arrived = 0;
    }
    if (w.live > 0 && w.arrived == w.live) {
        const int w0 = (S->cur / 64) * 64;
        for (int i = w0; i < w0 + 64 && i < S->nthreads; ++i)
            if (S->fib[i].st == WAIT_WAVE) S->fib[i].st = RUNNABLE;
        w.arrived = 0;
        w.parity ^= 1;
    }
    yield_to_main();
    abort();  // never resumed
}

void block_barrier() {
    Sched *s = S;
    s->block_arrived++;
    if (s->block_arrived == s->block_live) {
        for (int i = 0; i < s->nthreads; ++i)
            if (s->fib[i].st == WAIT_BLOCK) s->fib[i].st = RUNNABLE;
        s->block_arrived = 0;
        return;
    }
    s->fib[s->cur].st = WAIT_BLOCK;
    yield_to_main();
}

int lane_id() { return S->cur & 63; }

unsigned char *wave_publish(const void *src, size_t bytes) {
    Sched *s = S;
    WaveX &w = s->wave[s->cur / 64];
    const int par = w.parity;
    memcpy(w.slot[par] + 64 * (s->cur & 63), src, bytes);
    w.arrived++;
    if (w.arrived == w.live) {
        const int w0 = (s->cur / 64) * 64;
        for (int i = w0; i < w0 + 64 && i < s->nthreads; ++i)
            if (s->fib[i].st == WAIT_WAVE) s->fib[i].st = RUNNABLE;
        w.arrived = 0;
        w.parity ^= 1;
    } else {
        s->fib[s->cur].st = WAIT_WAVE;
        yield_to_main();
    }
    return w.slot[par];
}

static void run_block(Sched *s, unsigned bx, unsigned by, unsigned bz, dim3 grid, dim3 block) {
    const int n = (int)(block.x * block.y * block.z);
    s->nthreads = n;
    s->block_arrived = 0;
    s->block_live = n;
    for (int w = 0; w < (n + 63) / 64; ++w) {
        s->wave[w].parity = 0;
        s->wave[w].arrived = 0;
        s->wave[w].live = (n - w * 64) < 64 ? (n - w * 64) : 64;
    }
    while ((int)s->stacks.size() < n) s->stacks.push_back((unsigned char *)aligned_alloc(64, STACK_BYTES));
    for (int i = 0; i < n; ++i) {
        Fiber &f = s->fib[i];
        f.st = RUNNABLE;
        f.stack = s->stacks[i];
        uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
        void **sp = (void **)top;
        *--sp = nullptr;                 // fake return address of fiber_entry (keeps ABI alignment)
        *--sp = (void *)&fiber_entry;    // 'ret' target of the first switch
        for (int k = 0; k < 6; ++k) *--sp = nullptr;   // rbp rbx r12 r13 r14 r15
        f.sp = sp;
    }
    int remaining = n;
    while (remaining > 0) {
        bool progressed = false;
        for (int i = 0; i < n; ++i) {
            Fiber &f = s->fib[i];
            if (f.st != RUNNABLE) continue;
            s->cur = i;
            threadIdx.x = i % block.x;
            threadIdx.y = (i / block.x) % block.y;
            threadIdx.z = i / (block.x * block.y);
            blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz;
            blockDim = block; gridDim = grid;
            emu_ctx_switch(&s->main_sp, f.sp);
            progressed = true;
            if (f.st == DONE) remaining--;
        }
        if (!progressed) {
            fprintf(stderr, "emu: deadlock in block (%u,%u,%u): a barrier/collective not reached by all threads\n", bx, by, bz);
            abort();
        }
    }
}

void launch(dim3 grid, dim3 block, const std::function<void()> &body) {
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    if (block.x * block.y * block.z > (unsigned)MAX_THREADS) { fprintf(stderr, "emu: block too large\n"); abort(); }
    unsigned nt = std::thread::hardware_concurrency();
    if (const char *e = getenv("MST_EMU_THREADS")) nt = (unsigned)atoi(e);
    if (nt < 1) nt = 1;
    if (nt > nblocks) nt = (unsigned)nblocks;
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        Sched *s = new Sched();
        S = s;
        s->body = &body;
        for (;;) {
            const size_t b = next.fetch_add(1);
            if (b >= nblocks) break;
            run_block(s, (unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((size_t)grid.x * grid.y)), grid, block);
        }
        for (auto p : s->stacks) free(p);
        S = nullptr;
        delete s;
    };
    if (nt == 1) { worker(); return; }
    std::vector<std::thread> th;
    for (unsigned i = 0; i < nt; ++i) th.emplace_back(worker);
    for (auto &t : th) t.join();
}

}  // namespace emu
