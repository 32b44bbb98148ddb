// Runtime of the host lane-level emulator (see tests/emul/hip/hip_runtime.h).  TEST ONLY.
#include <hip/hip_runtime.h>

#include <sys/mman.h>

#include <atomic>
#include <thread>
#include <vector>

extern "C" void emul_switch(void** from_sp, void* to_sp);
asm(R"(
.text
.globl emul_switch
.type emul_switch,@function
emul_switch:
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
.size emul_switch,.-emul_switch
)");

namespace emul {
thread_local Fiber* cur = nullptr;
thread_local Block* blk = nullptr;

namespace {
constexpr size_t kStack = 256 * 1024;
constexpr int kMaxThreads = 1024;

struct Worker {
    void* sched_sp = nullptr;
    char* stacks = nullptr;
    Fiber fibers[kMaxThreads];
    Wave waves[kMaxThreads / kWave];
    const std::function<void()>* body = nullptr;
};
thread_local Worker* W = nullptr;
struct WorkerReaper {
    ~WorkerReaper() {
        if (W) { munmap(W->stacks, kStack * kMaxThreads); delete W; W = nullptr; }
    }
};
thread_local WorkerReaper reaper;

void fiber_entry() {
    (*W->body)();
    if (cur->dma_n) { fprintf(stderr, "emul: kernel ended with %d LDS-DMAs never waited for\n", cur->dma_n); abort(); }
    cur->done = true;
    // give the barrier bookkeeping a chance: a finished thread never arrives again
    emul_switch(&cur->sp, W->sched_sp);
    abort();
}

Worker* worker() {
    if (!W) {
        (void)&reaper;
        W = new Worker();
        W->stacks = (char*)mmap(nullptr, kStack * kMaxThreads, PROT_READ | PROT_WRITE,
                                MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (W->stacks == MAP_FAILED) { perror("mmap"); abort(); }
    }
    return W;
}

void run_block(Block& b, const std::function<void()>& body) {
    Worker* w = worker();
    w->body = &body;
    blk = &b;
    int n = b.nthreads;
    int nw = (n + kWave - 1) / kWave;
    b.waves = w->waves;
    for (int i = 0; i < nw; ++i) {
        w->waves[i].arrive = 0; w->waves[i].gen = 0;
        w->waves[i].nlanes = (i == nw - 1) ? n - i * kWave : kWave;
    }
    for (int t = 0; t < n; ++t) {
        Fiber& f = w->fibers[t];
        f.tid = dim3(t % b.bdim.x, (t / b.bdim.x) % b.bdim.y, t / (b.bdim.x * b.bdim.y));
        f.lane = t % kWave; f.wave = t / kWave; f.done = false;
        f.dma_head = 0; f.dma_n = 0;
        uintptr_t top = (uintptr_t)(w->stacks + kStack * (t + 1));
        top &= ~(uintptr_t)15;
        void** sp = (void**)(top - 64);
        sp[0] = sp[1] = sp[2] = sp[3] = sp[4] = sp[5] = nullptr;   // r15 r14 r13 r12 rbx rbp
        sp[6] = (void*)&fiber_entry;                               // return address
        sp[7] = nullptr;
        f.sp = sp;
    }
    int live = n;
    while (live > 0) {
        int progressed = 0;
        for (int t = 0; t < n; ++t) {
            Fiber& f = w->fibers[t];
            if (f.done) continue;
            cur = &f;
            emul_switch(&w->sched_sp, f.sp);
            if (f.done) { --live; }
            ++progressed;
        }
        if (!progressed) break;
    }
    cur = nullptr; blk = nullptr;
}
}  // namespace

void yield() { emul_switch(&cur->sp, W->sched_sp); }

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    int nthreads = block.x * block.y * block.z;
    if (nthreads > kMaxThreads) { fprintf(stderr, "emul: block too large\n"); abort(); }
    static int nworkers = [] {
        const char* e = getenv("EMUL_THREADS");
        int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
        return n < 1 ? 1 : n;
    }();
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= nblocks) break;
            Block b;
            b.bid = dim3(i % grid.x, (i / grid.x) % grid.y, i / ((size_t)grid.x * grid.y));
            b.bdim = block; b.gdim = grid; b.nthreads = nthreads;
            run_block(b, body);
        }
    };
    int nt = (int)std::min<size_t>(nworkers, nblocks);
    if (nt <= 1) { work(); return; }
    std::vector<std::thread> th;
    for (int i = 0; i < nt; ++i) th.emplace_back(work);
    for (auto& t : th) t.join();
}
}  // namespace emul
