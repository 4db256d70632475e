// wavesim runtime: cooperative fibers (ucontext), one OS thread per concurrently simulated
// workgroup.  See wavesim.h for what this is and is not.
#include "wavesim.h"

#include <ucontext.h>

#include <atomic>
#include <thread>
#include <vector>

namespace wavesim {

struct Fiber {
    ucontext_t ctx;
    uint3_ tid;
    int wave, lane;
    long wgen;   // wave collectives completed
    long bgen;   // block barriers completed
    bool done;
    char* stack;
};

struct WaveState {
    long arrived[2];
    int nwords[2];
    uint32_t slot[2][64 * 20];
};

struct BlockCtx {
    std::vector<Fiber> fibers;
    std::vector<WaveState> waves;
    long bar_arrived = 0;
    long progress = 0;
    dim3 bdim_, gdim_;
    uint3_ bid_;
    char* lds = nullptr;
    ucontext_t sched;
    Fiber* cur = nullptr;
    const std::function<void()>* body = nullptr;
    int nthreads = 0;
};

static thread_local BlockCtx* g = nullptr;
static const size_t kStack = 256 * 1024;

Fiber* cur() { return g->cur; }
uint3_ tid_of(const Fiber* f) { return f->tid; }
uint3_ bid() { return g->bid_; }
dim3 bdim() { return g->bdim_; }
dim3 gdim() { return g->gdim_; }
void* dyn_lds() { return g->lds; }
int lane_id() { return g->cur->lane; }

static void yield_() {
    Fiber* f = g->cur;
    swapcontext(&f->ctx, &g->sched);
}

void sync_block() {
    Fiber* f = g->cur;
    g->bar_arrived++;
    const long need = (f->bgen + 1) * (long)g->nthreads;
    while (g->bar_arrived < need) yield_();
    f->bgen++;
    g->progress++;
}

const uint32_t* wave_exchange(const uint32_t* words, int n) {
    Fiber* f = g->cur;
    WaveState& w = g->waves[f->wave];
    const int p = (int)(f->wgen & 1);
    if (n > 20) { fprintf(stderr, "wavesim: exchange too wide\n"); abort(); }
    const long base = (f->wgen / 2) * 64;
    if (w.arrived[p] == base) w.nwords[p] = n;
    else if (w.nwords[p] != n) {
        fprintf(stderr, "wavesim: lanes of wave %d disagree on collective #%ld (divergent collective)\n", f->wave, f->wgen);
        abort();
    }
    memcpy(&w.slot[p][f->lane * n], words, sizeof(uint32_t) * n);
    w.arrived[p]++;
    const long need = base + 64;
    while (w.arrived[p] < need) yield_();
    f->wgen++;
    g->progress++;
    return w.slot[p];
}

static void fiber_entry() {
    (*g->body)();
    g->cur->done = true;
    g->progress++;
    swapcontext(&g->cur->ctx, &g->sched);
}

static void run_block(BlockCtx& B) {
    g = &B;
    B.bar_arrived = 0;
    for (auto& w : B.waves) { w.arrived[0] = w.arrived[1] = 0; }
    for (int i = 0; i < B.nthreads; ++i) {
        Fiber& f = B.fibers[i];
        f.tid = {(unsigned)(i % B.bdim_.x), (unsigned)((i / B.bdim_.x) % B.bdim_.y), (unsigned)(i / (B.bdim_.x * B.bdim_.y))};
        f.wave = i / 64;
        f.lane = i % 64;
        f.wgen = f.bgen = 0;
        f.done = false;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, fiber_entry, 0);
    }
    int alive = B.nthreads;
    while (alive > 0) {
        const long before = B.progress;
        for (int i = 0; i < B.nthreads; ++i) {
            Fiber& f = B.fibers[i];
            if (f.done) continue;
            B.cur = &f;
            swapcontext(&B.sched, &f.ctx);
            if (f.done) --alive;
        }
        if (alive > 0 && B.progress == before) {
            fprintf(stderr, "wavesim: deadlock in block (%u,%u,%u): %d fibers blocked (barrier arrivals %ld)\n", B.bid_.x,
                    B.bid_.y, B.bid_.z, alive, B.bar_arrived);
            abort();
        }
    }
}

void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body) {
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads % 64 != 0) { fprintf(stderr, "wavesim: block size must be a multiple of 64\n"); abort(); }
    const long nblocks = (long)grid.x * grid.y * grid.z;
    int nworkers = (int)std::thread::hardware_concurrency();
    if (const char* e = getenv("WAVESIM_THREADS")) nworkers = atoi(e);
    if (nworkers < 1) nworkers = 1;
    if (nworkers > nblocks) nworkers = (int)nblocks;
    std::atomic<long> next{0};
    auto worker = [&]() {
        BlockCtx B;
        B.nthreads = nthreads;
        B.bdim_ = block;
        B.gdim_ = grid;
        B.body = &body;
        B.fibers.resize(nthreads);
        B.waves.resize(nthreads / 64);
        for (auto& f : B.fibers) f.stack = (char*)malloc(kStack);
        B.lds = (char*)aligned_alloc(64, ((lds_bytes + 63) / 64) * 64 + 64);
        for (;;) {
            const long b = next.fetch_add(1);
            if (b >= nblocks) break;
            B.bid_ = {(unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y))};
            memset(B.lds, 0xFF, lds_bytes);  // uninitialised LDS reads surface as NaN
            run_block(B);
        }
        for (auto& f : B.fibers) free(f.stack);
        free(B.lds);
        g = nullptr;
    };
    if (nworkers == 1) worker();
    else {
        std::vector<std::thread> ts;
        for (int i = 0; i < nworkers; ++i) ts.emplace_back(worker);
        for (auto& t : ts) t.join();
    }
}

}  // namespace wavesim
