// tests/emu/hip_emu.cpp -- TEST INFRASTRUCTURE ONLY (see hip_emu.hpp).
// Fiber scheduler that runs the threads of one emulated HIP block at a time.
#include "hip_emu.hpp"
#include <mutex>

namespace emu {

Block g_blk;
Fiber * g_cur = nullptr;
void * g_sched_sp = nullptr;
std::vector<Fiber> g_pool;
bool g_lockstep_release = getenv("BZ3_EMU_SCHED") != nullptr;

asm(R"(
.text
.globl emu_swap
.type emu_swap,@function
emu_swap:
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
.size emu_swap,.-emu_swap
)");

static void fiber_main() {
    g_blk.body();
    g_cur->done = true;
    fiber_exit_hook();
    emu_swap(&g_cur->sp, g_sched_sp);
    abort();  // never resumed
}

void fiber_exit_hook() {
    Block & b = g_blk;
    int w = cur_wave();
    b.alive--;
    b.wave_alive[w]--;
    b.live_mask[w] &= ~(1ull << cur_lane());
    // an exiting thread must not strand threads already waiting at a rendezvous
    if (b.alive > 0 && b.bar_count == (unsigned)b.alive) {
        b.bar_count = 0;
        b.bar_gen++;
    }
    if (b.wave_alive[w] > 0 && b.wbar_count[w] == (unsigned)b.wave_alive[w]) {
        b.wbar_count[w] = 0;
        b.wbar_gen[w]++;
    }
}

static void prepare(Fiber & f) {
    if (!f.stack) {
        f.stack = (char *)mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (f.stack == (char *)MAP_FAILED) { perror("mmap"); abort(); }
    }
    uintptr_t top = ((uintptr_t)(f.stack + kStack)) & ~(uintptr_t)15;
    void ** sp = (void **)(top - 16);  // slot holding the entry address, 16-byte aligned
    *sp = (void *)&fiber_main;
    for (int i = 0; i < 6; i++) *--sp = nullptr;  // rbp rbx r12 r13 r14 r15
    f.sp = (void *)sp;
    f.done = false;
}

static std::mutex g_launch_mu;  // one emulated GPU: launches from several host threads (one per "device") take turns

void run_grid(dim3 grid, dim3 block, size_t shmem, const std::function<void()> & body) {
    std::lock_guard<std::mutex> launch_lock(g_launch_mu);
    if (g_cur != nullptr) { fprintf(stderr, "emu: nested launch\n"); abort(); }
    int nthreads = (int)(block.x * block.y * block.z);
    if (block.y != 1 || block.z != 1 || nthreads > 32 * kWave) { fprintf(stderr, "emu: unsupported block shape\n"); abort(); }
    if ((int)g_pool.size() < nthreads) g_pool.resize(nthreads);
    char * dyn = shmem ? (char *)aligned_alloc(64, (shmem + 63) / 64 * 64) : nullptr;
    Block & b = g_blk;
    b.body = body;
    b.bdim = block;
    b.gdim = grid;
    b.dyn = dyn;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                b.bid = dim3(bx, by, bz);
                b.nthreads = b.alive = nthreads;
                b.bar_count = 0;
                b.nwaves = (nthreads + kWave - 1) / kWave;
                for (int w = 0; w < b.nwaves; w++) {
                    int cnt = nthreads - w * kWave;
                    if (cnt > kWave) cnt = kWave;
                    b.wave_alive[w] = cnt;
                    b.wbar_count[w] = 0;
                    b.live_mask[w] = cnt == 64 ? ~0ull : ((1ull << cnt) - 1);
                }
                for (int t = 0; t < nthreads; t++) {
                    prepare(g_pool[t]);
                    g_pool[t].tid = dim3((unsigned)t, 0, 0);
                }
                // BZ3_EMU_SCHED=<seed>: stress mode for the inter-wave hand-off protocols.  In every sweep each wave is
                // either given its turn or skipped (coin flip per wave), so the waves of a block advance at wildly
                // different and changing speeds, as they do on a CU shared with other workgroups.
                static const char * sched = getenv("BZ3_EMU_SCHED");
                static uint64_t rng = sched ? (uint64_t)atoll(sched) * 0x9E3779B97F4A7C15ull + 0x2545F4914F6CDD1Dull : 0;
                while (b.alive > 0) {
                    uint32_t skip = 0;
                    if (sched) {
                        rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
                        if (sched[0] == '-') {  // negative seed: long stalls -- now and then a wave sleeps for up to 63 sweeps
                            static int stall[32];
                            for (int w = 0; w < b.nwaves; w++) {
                                if (stall[w] > 0) stall[w]--;
                                else if (((rng >> (3 * w)) & 7u) == 0) stall[w] = (int)((rng >> 40) & 63u);
                                if (stall[w] > 0) skip |= 1u << w;
                            }
                        } else {
                            skip = (uint32_t)(rng >> 20) & ((1u << b.nwaves) - 1u);
                            if ((rng >> 60) < 4) skip = 0;                              // now and then everybody runs
                        }
                        if (skip == ((1u << b.nwaves) - 1u)) skip &= ~(1u << ((rng >> 8) % (unsigned)b.nwaves));
                    }
#ifdef BZ3_EMU_WATCH
                    static unsigned long long sweeps = 0;
                    if ((++sweeps % 2000000ull) == 0) {
                        fprintf(stderr, "[emu] sweep %llu block %u alive %d skip %x waves alive:", sweeps, bx, b.alive, skip);
                        for (int w = 0; w < b.nwaves; w++) fprintf(stderr, " %d", b.wave_alive[w]);
                        fprintf(stderr, "\n");
                    }
#endif
                    for (int t = 0; t < nthreads; t++) {
                        Fiber & f = g_pool[t];
                        if (f.done || ((skip >> (t / kWave)) & 1u)) continue;
                        g_cur = &f;
                        emu_swap(&g_sched_sp, f.sp);
                    }
                }
                g_cur = nullptr;
            }
    free(dyn);
    b.body = nullptr;
}

}  // namespace emu
