// api.hip -- the C ABI (include/libbz3.h + include/bz3_hip.h) on top of the HIP stage pipelines.
//
// Mirrors the reference's orchestration exactly: bz3_encode_block (src/libbz3.c:585-654) and
// bz3_decode_block (:656-809) are ping-pong pipelines over the caller's buffer and the state's swap
// buffer; every size/flag decision and every error code is reproduced.  What changes is WHERE things
// live: both buffers are in HBM, the stages are the kernels of crc32c/mrle/lzp/bwt/unbwt/cm.hip, the
// suffix-sort workspace is one arena per GPU shared by all states, and a batch call
// (bz3_encode_blocks / bz3_decode_blocks, :845-870, without pthreads) runs in phases on ONE stream per GPU:
// the whole-GPU stages block after block, the serial stages (LZP driver, LZP decoder, CM coder) as ONE
// launch with one workgroup per block, so up to 256 blocks share the GPU's 256 CUs.
//
// There is no CPU implementation of any stage in this library: without a usable HIP device
// bz3_new() returns NULL and the stage hooks abort loudly.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <mutex>
#include <new>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/bz3_hip.h"
#include "prims.hpp"
#include "sort.hpp"
#include "stages.hpp"

using namespace bz3;

namespace {

constexpr s32 KiB65 = 65 * 1024;
constexpr s32 MiB511 = 511 * 1024 * 1024;

inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

inline u32 rd_le32(const u8 * p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }
inline void wr_le32(u8 * p, u32 v) {
    p[0] = (u8)v; p[1] = (u8)(v >> 8); p[2] = (u8)(v >> 16); p[3] = (u8)(v >> 24);
}

// ---- per-device context -----------------------------------------------------------------------
struct DeviceCtx {
    int device = 0;
    std::mutex mu;          // serialises users of the shared workspace
    CrcTables * d_crc = nullptr;
    char * ws = nullptr;
    size_t ws_cap = 0;
    int cus = 256;          // compute units: one full-model CM workgroup fits per CU
    // side streams of the device: the serial LZP drivers of a window of blocks run there while the calling thread drives the
    // whole-GPU stages of the neighbouring windows on the group's stream (encode_group and decode_group: one side stream per slot
    // of their rings, so that the serial kernels of consecutive windows overlap)
    static constexpr int AUX = 8;           // side streams / ring slots available; the rings use four unless a test or an experiment asks for more
    static constexpr int RING_SLOTS = 4;
    hipStream_t aux[AUX] = {};
    hipStream_t duo = nullptr;  // round 6: phase A of the encoder's front end when it runs on a thread of its own (encode_group)
    hipEvent_t ev_prep = nullptr, ev_d0[AUX] = {}, ev_d1[AUX] = {};
    bool aux_ready = false;
    // CU partition (round 5).  The serial kernels on the side streams hold a CU each for the better part of a second (an LZP decoder: 1024 lanes, ~0.5 s per
    // 256 MiB block), and a whole-GPU kernel that finds half of such a CU's wave slots taken runs its workgroups there at a fraction of the pace -- the
    // inverse BWT's walk kernels, whose launch lasts as long as their slowest workgroup, measured 2.5x their stand-alone time beside the decoders of three
    // windows (profiles/r05_rings_256x64MiB.txt: the tail's host thread spent 82 % of its time in the inverse-BWT loop, 0.6 % waiting for LZP decoders).
    // So the side streams are created on `reserve` CUs of their own and the tail's whole-GPU kernels run on a stream masked to the OTHER CUs
    // (hipExtStreamCreateWithCUMask); the CM launches keep the group's unmasked stream (they need every CU).  BZ3_HIP_CU_RESERVE=<CUs> (read once;
    // default 64 since round 6 -- the most cu_masks hands out --, round 5: 48; 0 = no partition).  Falls back to plain streams when the runtime refuses.
    hipStream_t rest = nullptr;   // whole-GPU kernels beside the side streams' serial kernels: every CU but the reserved ones (null: no partition)
    hipStream_t aux_m[AUX] = {};  // the decoder's side streams, on the reserved CUs (the encoder's rings keep the plain ones: call 4 measured its
                                         // front end 6 % SLOWER with its LZP drivers confined to 32 CUs, profiles/r05_cu_partition_256x64MiB.txt)
    int reserved_cus = 0;
    static int reserved_cus_wanted() { return cu_reserve_setting(); }
    static int cu_reserve_setting() {
        static const int v = [] {
            const char * e = getenv("BZ3_HIP_CU_RESERVE");
            return e ? atoi(e) : 64;  // the most cu_masks hands out (8 per block of 32 CUs); round 5: 48 = three windows of 16 LZP decoders, a CU each
        }();
        return v;
    }
    // Mask bit i of the reserved set: 32 a + 8 j + ((a + 2 j + t / 4) mod 8), j = t mod 4, for a < 8 and t < reserve / 8 (at most 8) -- the same number
    // of CUs from every block of 32 bits AND from every residue class mod 8, so that each XCD gives up equally many whether the driver deals the mask
    // bits to the XCDs in blocks or round robin.
    static void cu_masks(int cus, int reserve, std::vector<uint32_t> & side, std::vector<uint32_t> & main) {
        const int words = (cus + 31) / 32;
        side.assign((size_t)words, 0u);
        main.assign((size_t)words, 0u);
        for (int i = 0; i < cus; i++) main[(size_t)(i >> 5)] |= 1u << (i & 31);
        const int per = reserve / 8 > 0 ? reserve / 8 : 1;
        for (int a = 0; a < 8 && a * 32 < cus; a++)
            for (int t = 0; t < per && t < 8; t++) {
                const int j = t & 3;
                const int i = 32 * a + 8 * j + ((a + 2 * j + (t >> 2)) & 7);
                if (i < cus) {
                    side[(size_t)(i >> 5)] |= 1u << (i & 31);
                    main[(size_t)(i >> 5)] &= ~(1u << (i & 31));
                }
            }
    }
    void ensure_aux() {  // caller holds mu
        if (aux_ready) return;
        // built into locals and committed only when everything exists: a failure half way must not leave a context whose first
        // stream is there and whose events are not (every later call would record on null events)
        hipStream_t st[AUX] = {}, sm[AUX] = {}, rs = nullptr, du = nullptr;
        hipEvent_t e0[AUX] = {}, e1[AUX] = {}, ep = nullptr;
        int reserved = 0;
        try {
            HIP_CHECK(hipEventCreate(&ep));
#ifndef BZ3_EMU
            const int want = cu_reserve_setting();
            int real_cus = 0;  // (the device's own count: `cus` may be a test's pretence, BZ3_HIP_CUS)
            if (hipDeviceGetAttribute(&real_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) real_cus = 0;
            if (want >= 8 && real_cus >= 64 && real_cus % 32 == 0) {
                std::vector<uint32_t> side, mainm;
                cu_masks(real_cus, want, side, mainm);
                bool ok = hipExtStreamCreateWithCUMask(&rs, (uint32_t)mainm.size(), mainm.data()) == hipSuccess;
                // (RING_SLOTS of them: every masked stream is a hardware queue of its own, and call 8 -- eight of them, a tail ring of 8 x 8 -- ran every
                // phase of the tail slower than call 6's four; a ring with more slots than that runs unpartitioned on the plain streams)
                for (int k = 0; ok && k < RING_SLOTS; k++) ok = hipExtStreamCreateWithCUMask(&sm[k], (uint32_t)side.size(), side.data()) == hipSuccess;
                if (!ok) {  // the runtime refuses: plain streams only
                    (void)hipGetLastError();
                    if (rs) (void)hipStreamDestroy(rs);
                    rs = nullptr;
                    for (int k = 0; k < AUX; k++) {
                        if (sm[k]) (void)hipStreamDestroy(sm[k]);
                        sm[k] = nullptr;
                    }
                } else {
                    for (uint32_t wd : side) reserved += __builtin_popcount(wd);
                }
            }
#endif
            for (int k = 0; k < AUX; k++) {
                HIP_CHECK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
                HIP_CHECK(hipEventCreate(&e0[k]));
                HIP_CHECK(hipEventCreate(&e1[k]));
            }
            HIP_CHECK(hipStreamCreateWithFlags(&du, hipStreamNonBlocking));
        } catch (...) {
            if (du) (void)hipStreamDestroy(du);
            if (ep) (void)hipEventDestroy(ep);
            if (rs) (void)hipStreamDestroy(rs);
            for (int k = 0; k < AUX; k++)
                if (sm[k]) (void)hipStreamDestroy(sm[k]);
            for (int k = 0; k < AUX; k++) {
                if (st[k]) (void)hipStreamDestroy(st[k]);
                if (e0[k]) (void)hipEventDestroy(e0[k]);
                if (e1[k]) (void)hipEventDestroy(e1[k]);
            }
            throw;
        }
        ev_prep = ep;
        duo = du;
        rest = rs;
        for (int k = 0; k < AUX; k++) aux_m[k] = sm[k];
        reserved_cus = reserved;
        for (int k = 0; k < AUX; k++) {
            aux[k] = st[k];
            ev_d0[k] = e0[k];
            ev_d1[k] = e1[k];
        }
        aux_ready = true;
    }

    // Swap buffers lent to "lean" states for the duration of a stage sequence (see bz3_hip_set_lean_states).
    std::mutex temp_mu;
    std::vector<std::pair<u8 *, size_t>> temps_free;
    std::vector<std::pair<u8 *, size_t>> temps_out;
    u8 * temp_get(size_t cap) {
        std::lock_guard<std::mutex> lk(temp_mu);
        for (size_t k = 0; k < temps_free.size(); k++)
            if (temps_free[k].second >= cap) {
                auto t = temps_free[k];
                temps_free.erase(temps_free.begin() + (long)k);
                temps_out.push_back(t);
                return t.first;
            }
        u8 * p = nullptr;
        if (hipMalloc((void **)&p, cap) != hipSuccess) {
            // (the ring's budget is an estimate: before the block fails, give back what the pool holds idle -- buffers of other sizes -- and ask again)
            (void)hipGetLastError();
            // A buffer goes back to the pool by STREAM ORDER (encode_front_b): kernels already queued on the group's stream may still read it.
            // Every borrower launches on that one stream, which is what protects a re-borrowed buffer; a FREE is not a stream operation, so wait.
            (void)hipDeviceSynchronize();
            for (auto & t : temps_free) (void)hipFree(t.first);
            temps_free.clear();
            HIP_CHECK(hipMalloc((void **)&p, cap));
        }
        temps_out.push_back({p, cap});
        return p;
    }
    void temp_put(u8 * p) {
        if (!p) return;
        std::lock_guard<std::mutex> lk(temp_mu);
        for (size_t k = 0; k < temps_out.size(); k++)
            if (temps_out[k].first == p) {
                temps_free.push_back(temps_out[k]);
                temps_out.erase(temps_out.begin() + (long)k);
                return;
            }
    }
    void temp_trim() {  // hand the idle swap buffers back to the driver
        std::lock_guard<std::mutex> lk(temp_mu);
        if (temps_free.empty()) return;
        (void)hipDeviceSynchronize();  // (see temp_get: a pooled buffer may still be read by kernels in flight; hipFree's own synchronisation is not documented)
        for (auto & t : temps_free) (void)hipFree(t.first);
        temps_free.clear();
    }
    size_t temp_idle_bytes() {
        std::lock_guard<std::mutex> lk(temp_mu);
        size_t b = 0;
        for (auto & t : temps_free) b += t.second;
        return b;
    }

    // What arena_for allocates beyond the request (alignment losses of the takes, a somewhat larger next call): a sixteenth, but never more than
    // 512 MiB -- round 5's unbounded sixteenth was 5 GB of an 80 GB ring that nobody used and the host program could not have.
    static size_t arena_slack(size_t bytes) {
        const size_t s16 = bytes >> 4, cap = (size_t)512 << 20;
        return (s16 < cap ? s16 : cap) + ((size_t)1 << 20);
    }
    Arena arena_for(size_t bytes) {  // caller holds mu
        if (bytes > ws_cap) {
            if (ws) HIP_CHECK(hipFree(ws));
            ws = nullptr;
            ws_cap = 0;
            size_t want = bytes + arena_slack(bytes);
            HIP_CHECK(hipMalloc((void **)&ws, want));
            ws_cap = want;
        }
        Arena a;
        a.base = ws;
        a.cap = ws_cap;
        a.used = 0;
        return a;
    }
};

std::mutex g_mu;
std::vector<DeviceCtx *> g_ctx;
int g_device_count = -1;
std::atomic<int> g_bound_device{-2};  // -2 = not initialised from the environment yet, -1 = round robin
std::atomic<unsigned> g_rr{0};
std::atomic<int> g_lean{-1};  // -1 = not read from the environment yet; see bz3_hip_set_lean_states
std::atomic<int> g_front_end_ring{0};  // window | slots << 16 of the last encode_group (bz3_hip_debug_front_end_ring)
std::atomic<int> g_arena_swaps{0};  // swap buffers served from the arena (bz3_hip_debug_arena_swap_buffers)
std::atomic<unsigned> g_cm_given_up{0};  // blocks the row-cache CM kernels handed back to the full-model kernels (statistics)
std::atomic<unsigned> g_cm_launches{0};  // CM kernel launches (statistics, bz3_hip_debug_cm_launches)
std::atomic<unsigned> g_cm_routed_full{0};  // blocks sent straight to the full-model kernels by their histogram / payload size (statistics)

#ifndef BZ3_EMU
// The rings run a group's whole-GPU kernels on one stream and the serial one-workgroup-per-block kernels (LZP drivers / decoders) of up to four windows on
// side streams.  The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a queue do not
// overlap: with five streams on four queues the tail of 256 x 64 MiB blocks measured 3.23 s, with more queues 3.08 s (profiles/r05_tail_hw_queues.txt).
// The library asks for 8 ONCE, when it is loaded (before any thread of the host program can be reading the environment through it), never overrides the
// user's setting, and BZ3_HIP_SET_HW_QUEUES=0 turns even that off.  It only has an effect if the runtime is not up yet: a host program that initialises HIP
// first (bench.py through torch) sets the variable itself -- INTEGRATION.md lists it as a requirement on the host.
__attribute__((constructor)) static void bz3_hip_on_load() {
    const char * e = getenv("BZ3_HIP_SET_HW_QUEUES");
    if (!e || atoi(e) != 0) (void)setenv("GPU_MAX_HW_QUEUES", "8", 0);
}
#endif

int device_count() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_device_count < 0) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
        g_device_count = n;
        g_ctx.assign((size_t)(n > 0 ? n : 0), nullptr);
    }
    return g_device_count;
}

DeviceCtx * get_ctx(int dev) {
    const int n = device_count();
    if (dev < 0 || dev >= n) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_ctx[dev]) {
        HIP_CHECK(hipSetDevice(dev));
        DeviceCtx * c = new DeviceCtx;
        c->device = dev;
        CrcTables t;
        memset(&t, 0, sizeof t);
        crc_build_tables(t);
        HIP_CHECK(hipMalloc((void **)&c->d_crc, sizeof(CrcTables)));
        HIP_CHECK(hipMemcpy(c->d_crc, &t, sizeof t, hipMemcpyHostToDevice));
#ifndef BZ3_EMU
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) c->cus = cus;
#endif
        if (const char * e = getenv("BZ3_HIP_CUS"))  // tests / experiments: pretend the GPU has this many CUs (batch-size policies)
            if (atoi(e) > 0) c->cus = atoi(e);
        g_ctx[dev] = c;
    }
    return g_ctx[dev];
}

int pick_device() {
    const int n = device_count();
    if (n <= 0) return -1;
    int b = g_bound_device.load();
    if (b == -2) {
        const char * e = getenv("BZ3_HIP_DEVICE");
        b = (e && *e) ? atoi(e) : -1;
        if (b >= n) b = -1;
        g_bound_device.store(b);
    }
    if (b >= 0) return b;
    return (int)(g_rr.fetch_add(1) % (unsigned)n);
}

// The two halves of the encoder's front end have scratch needs of their own (round 6: they may run on two host threads and two streams, see encode_group):
// phase A = CRC, mRLE, LZP preparation (hash sort, binned links); phase B = LZP emission and the suffix sort.
size_t front_a_scratch_bytes(u64 n) { return (size_t)n * 30 + radix_temp_bytes(n, 9) + scan_temp_words(n / 8 + 4096) * 4 + (4u << 20); }
size_t front_b_scratch_bytes(u64 n) { return bwt_workspace_bytes(n) + (size_t)(n / 1024 + 4096) * 4; }
size_t workspace_bytes_for(u64 n) {
    size_t a = bwt_workspace_bytes(n), b = unbwt_workspace_bytes(n);
    size_t c = (size_t)n * 30 + radix_temp_bytes(n, 9) + scan_temp_words(n / 8 + 4096) * 4 + (4u << 20);  // LZP: links, mlen, bitmaps, the hash sort's buffers + the binned link records (lzp.hip)
    size_t m = a > b ? a : b;
    return m > c ? m : c;
}

// ---- CM kernel variant -------------------------------------------------------------------------------
// The full-model CM kernels need a whole CU's LDS per block; the row-cache kernels (cm.hip) need half or a third of it, so two or
// three blocks share a CU, but they give up blocks whose order-1 working set does not fit (binary data), which are then coded again
// by the full-model kernel.  Policy (BZ3_HIP_CM_MODE=auto|full|rows|rows3, bz3_hip_set_cm_mode): `auto` picks by batch size.
std::atomic<int> g_cm_mode{-2};  // -2 = not read from the environment yet, -1 = auto, else CM_VARIANT_*

int cm_mode() {
    int m = g_cm_mode.load();
    if (m == -2) {
        const char * e = getenv("BZ3_HIP_CM_MODE");
        m = -1;
        if (e && !strcmp(e, "full")) m = CM_VARIANT_FULL;
        else if (e && !strcmp(e, "rows")) m = CM_VARIANT_ROWS;
        else if (e && !strcmp(e, "rows3")) m = CM_VARIANT_ROWS3;
#ifdef BZ3_EMU
        else if (e && !strcmp(e, "rows-test")) m = CM_VARIANT_ROWS_TEST;
#endif
        g_cm_mode.store(m);
    }
    return m;
}

bool lean_states() {
    int v = g_lean.load();
    if (v < 0) {
        const char * e = getenv("BZ3_HIP_LEAN");
        v = (e && *e && strcmp(e, "0")) ? 1 : 0;
        g_lean.store(v);
    }
    return v != 0;
}

int cm_variant_for(const DeviceCtx * ctx, size_t njobs, bool encode) {
    (void)encode;
    const int m = cm_mode();
    const size_t c = (size_t)ctx->cus;
    if (m >= 0) return m;
    // auto, by batch size.  Up to one block per CU the whole model sits in LDS; beyond that the CM launch would need a second round of
    // workgroups, and the row-cache kernels put two / three blocks on a CU instead: ns per byte and block of a 2 MiB text block on
    // MI355X, decoder 551 / ~650 / 737 with one / two / three blocks per CU (profiles/r03_cm_decoder_steps.txt), the encoder's
    // model waves interleave even better (x1.5 / x2.0 throughput, profiles/r01_cm_rows_probe.txt).
    if (njobs > 2 * c) return CM_VARIANT_ROWS3;
    if (njobs > c) return CM_VARIANT_ROWS;
    return CM_VARIANT_FULL;
}

size_t cm_scratch_bytes(size_t njobs) { return njobs * (CM_SPILL_BYTES + 256) + 4096 + CM_CLAIM_WORDS * 4 + 256; }

// Runs the CM kernel over `jobs` (host copies; d_jobs has room for all of them) and returns the kernel time in ms.
// Row-cache variants: blocks the kernel gave up are coded again by the full-model kernel in a second launch.
// to_full (optional, one flag per job): blocks known not to fit the row cache -- many live byte values (encode: from the BWT's histogram), or a
// payload that hardly shrank (decode) -- skip the row-cache kernels and go straight into the whole-model launch (round 5; g_cm_given_up counts only
// blocks the row-cache kernels really handed back).
template <class Job, class Launch>
float run_cm_jobs(const DeviceCtx * ctx, Arena & arena, std::vector<Job> & jobs, Job * d_jobs, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, Launch && go,
                  const std::vector<char> * to_full = nullptr) {
    if (jobs.empty()) return 0.f;
    std::vector<Job> direct;  // straight to the whole-model kernel
    // Routing only pays when the WHOLE batch would take a row-cache variant: a batch that fits one block per CU is whole-model work anyway, and splitting it
    // would run two serial launches, each as long as its slowest block, where one does.
    if (to_full && cm_mode() < 0 && cm_variant_has_rows(cm_variant_for(ctx, jobs.size(), std::is_same<Job, CmEncodeJob>::value))) {
        std::vector<Job> keep;
        for (size_t i = 0; i < jobs.size(); i++) ((*to_full)[i] ? direct : keep).push_back(jobs[i]);
        if (!direct.empty()) {
            g_cm_routed_full.fetch_add((unsigned)direct.size());
            jobs.swap(keep);
        }
    }
    if (jobs.empty()) {  // everything is whole-model work: one launch
        float ms0 = 0.f;
        HIP_CHECK(hipMemcpyAsync(d_jobs, direct.data(), sizeof(Job) * direct.size(), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipEventRecord(ev0, s));
        g_cm_launches.fetch_add(1);
        go(d_jobs, (u32)direct.size(), s, (int)CM_VARIANT_FULL);
        HIP_CHECK(hipEventRecord(ev1, s));
        HIP_CHECK(hipStreamSynchronize(s));
        (void)hipEventElapsedTime(&ms0, ev0, ev1);
        jobs.swap(direct);
        return ms0;
    }
    const int variant = cm_variant_for(ctx, jobs.size(), std::is_same<Job, CmEncodeJob>::value);
    const size_t mk = arena.mark();
    if constexpr (std::is_same<Job, CmEncodeJob>::value) {  // a word per CU for the coder waves' SIMD claims (cm.hip)
        static const bool no_claim = getenv("BZ3_CM_NO_CLAIM") != nullptr;  // (experiments, read once: the block index decides, as up to round 4)
        if (!no_claim) {
            u32 * claim = arena.take<u32>(CM_CLAIM_WORDS);
            HIP_CHECK(hipMemsetAsync(claim, 0, CM_CLAIM_WORDS * sizeof(u32), s));
            for (auto & j : jobs) j.claim = dev_addr(claim);
            for (auto & j : direct) j.claim = dev_addr(claim);
        }
    }
    u32 * d_status = nullptr;
    if (cm_variant_has_rows(variant)) {
        u8 * spill = arena.take<u8>(jobs.size() * CM_SPILL_BYTES);
        d_status = arena.take<u32>(jobs.size());
        HIP_CHECK(hipMemsetAsync(d_status, 0, jobs.size() * 4, s));
        for (size_t i = 0; i < jobs.size(); i++) {
            jobs[i].spill = dev_addr(spill + i * CM_SPILL_BYTES);
            jobs[i].status = dev_addr(d_status + i);
            jobs[i].miss_base = cm_variant_is_test(variant) ? 64u : CM_MISS_BASE;
            jobs[i].miss_shift = cm_variant_is_test(variant) ? 3u : CM_MISS_SHIFT;  // give up beyond 3 % misses (test variants: 12.5 %)
        }
    }
    float ms = 0.f, ms2 = 0.f;
    HIP_CHECK(hipMemcpyAsync(d_jobs, jobs.data(), sizeof(Job) * jobs.size(), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipEventRecord(ev0, s));
    g_cm_launches.fetch_add(1);
    go(d_jobs, (u32)jobs.size(), s, variant);
    HIP_CHECK(hipEventRecord(ev1, s));
    HIP_CHECK(hipStreamSynchronize(s));
    (void)hipEventElapsedTime(&ms, ev0, ev1);
    if (cm_variant_has_rows(variant)) {
        std::vector<u32> status(jobs.size());
        HIP_CHECK(hipMemcpyAsync(status.data(), d_status, jobs.size() * 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        std::vector<Job> again;
        for (size_t i = 0; i < jobs.size(); i++)
            if (status[i]) again.push_back(jobs[i]);
        g_cm_given_up.fetch_add((unsigned)again.size());
        again.insert(again.end(), direct.begin(), direct.end());
        direct.clear();
        if (!again.empty()) {
            HIP_CHECK(hipMemcpyAsync(d_jobs, again.data(), sizeof(Job) * again.size(), hipMemcpyHostToDevice, s));
            HIP_CHECK(hipEventRecord(ev0, s));
            g_cm_launches.fetch_add(1);
            go(d_jobs, (u32)again.size(), s, (int)CM_VARIANT_FULL);  // whole model in LDS: nothing to give up
            HIP_CHECK(hipEventRecord(ev1, s));
            HIP_CHECK(hipStreamSynchronize(s));
            (void)hipEventElapsedTime(&ms2, ev0, ev1);
        }
    }
    if (!direct.empty()) {  // (the variant had no row cache: nothing was handed back, the routed blocks still wait)
        HIP_CHECK(hipMemcpyAsync(d_jobs, direct.data(), sizeof(Job) * direct.size(), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipEventRecord(ev0, s));
        g_cm_launches.fetch_add(1);
        go(d_jobs, (u32)direct.size(), s, (int)CM_VARIANT_FULL);
        HIP_CHECK(hipEventRecord(ev1, s));
        HIP_CHECK(hipStreamSynchronize(s));
        float ms3 = 0.f;
        (void)hipEventElapsedTime(&ms3, ev0, ev1);
        ms2 += ms3;
    }
    arena.release(mk);
    return ms + ms2;
}

// ---- small kernels of the orchestration layer ---------------------------------------------------
__global__ void k_write_header(u8 * __restrict__ b, const u32 * __restrict__ crc, const u32 * __restrict__ idx_word, u32 model, u32 lzp_size, u32 rle_size) {
    if (threadIdx.x != 0) return;
    const u32 c = *crc, idx = *idx_word;  // (the primary index stays on the device: bwt_forward's d_idx)
    b[0] = (u8)c; b[1] = (u8)(c >> 8); b[2] = (u8)(c >> 16); b[3] = (u8)(c >> 24);
    b[4] = (u8)idx; b[5] = (u8)(idx >> 8); b[6] = (u8)(idx >> 16); b[7] = (u8)(idx >> 24);
    b[8] = (u8)model;
    u32 o = 9;
    if (model & 2u) { b[o] = (u8)lzp_size; b[o + 1] = (u8)(lzp_size >> 8); b[o + 2] = (u8)(lzp_size >> 16); b[o + 3] = (u8)(lzp_size >> 24); o += 4; }
    if (model & 4u) { b[o] = (u8)rle_size; b[o + 1] = (u8)(rle_size >> 8); b[o + 2] = (u8)(rle_size >> 16); b[o + 3] = (u8)(rle_size >> 24); }
}

// Blocks shorter than 64 bytes are stored: [crc][0xFFFFFFFF][bytes] (src/libbz3.c:596-601).
__global__ void __launch_bounds__(64) k_store_small(u8 * __restrict__ b, u32 n, const u32 * __restrict__ crc) {
    const u32 t = threadIdx.x;
    const u8 v = t < n ? b[t] : (u8)0;
    __syncthreads();
    if (t < n) b[8 + t] = v;
    if (t == 0) {
        const u32 c = *crc;
        b[0] = (u8)c; b[1] = (u8)(c >> 8); b[2] = (u8)(c >> 16); b[3] = (u8)(c >> 24);
        b[4] = b[5] = b[6] = b[7] = 0xFF;
    }
}
__global__ void __launch_bounds__(128) k_unstore_small(u8 * __restrict__ b, u32 n) {  // memmove(buffer, buffer + 8, n), :684
    const u32 t = threadIdx.x;
    const u8 v = t < n ? b[8 + t] : (u8)0;
    __syncthreads();
    if (t < n) b[t] = v;
}

}  // namespace

// ---- the state ------------------------------------------------------------------------------------
struct bz3_state {
    s32 block_size = 0;
    s8 last_error = BZ3_OK;
    int device = 0;
    DeviceCtx * ctx = nullptr;
    hipStream_t stream = nullptr;  // owned
    hipStream_t xs = nullptr;      // execution stream of the current call: the lead state's stream of its device group
    u8 * d_swap = nullptr;  // the reference's swap_buffer, in HBM (lean states: borrowed from the device's pool while a call needs it)
    bool lean = false;      // see bz3_hip_set_lean_states
    bool timed = true;      // this block's stages are timed with host clocks (a stream synchronisation per stage): the first block of a group only
    u8 * d_io = nullptr;    // staging for the host-buffer API (lazy)
    size_t cap = 0;         // bz3_bound(block_size) rounded up
    u32 * d_words = nullptr;  // [0..1] crc scratch/result, [2] cm coded size, [4] rle total, [5] lzp result
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float t[BZ3_HIP_T_COUNT] = {0};
    BwtStats bwt;

    // block in flight inside one API call
    enum Pending { NONE, ENC_STORED, ENC_CODED, DEC_STORED, DEC_CODED, FAILED } pending = NONE;
    u8 * user = nullptr;  // caller's device buffer
    u8 * b1 = nullptr, * b2 = nullptr;
    s32 size = 0, overhead = 0, result = -1;
    u32 n_cm = 0;         // bytes entering / leaving the CM stage
    const u8 * cm_in = nullptr;
    u32 cm_in_size = 0;
    u8 * side = nullptr;  // lean encode: this block's slice of the in-place coder's side buffer
    bool hold_swap = false;  // lean encode, two-thread front end: encode_front_b leaves the swap buffer with the state; the window's thread returns it behind an event
    bool skip = false;    // host-buffer API: staging this block failed (on_failure has set the error); the group leaves it alone
    // decode
    size_t buffer_size = 0;
    u32 crc = 0;
    s32 bwt_idx = 0, model = 0, lzp_size = -1, rle_size = -1, orig_size = 0, size_before_bwt = 0, size_src = 0;
};

namespace {

struct DeviceGuard {
    explicit DeviceGuard(int d) { HIP_CHECK(hipSetDevice(d)); }
};

// A group call that is left by an exception must not leave kernels in flight on either stream of the device: the caller is about to
// hand borrowed swap buffers back to the pool and to reuse the arena (the serial LZP kernels run on the side streams).
struct DrainOnUnwind {
    hipStream_t main;
    hipStream_t * side;  // n_side side streams
    int n_side;
    int live = std::uncaught_exceptions();
    ~DrainOnUnwind() {
        if (std::uncaught_exceptions() > live) {
            if (main) (void)hipStreamSynchronize(main);
            for (int k = 0; k < n_side; k++)
                if (side[k]) (void)hipStreamSynchronize(side[k]);
        }
    }
};

void state_release(bz3_state * st) {
    if (!st) return;
    (void)hipSetDevice(st->device);
    if (st->stream) (void)hipStreamSynchronize(st->stream);
    if (st->ev0) (void)hipEventDestroy(st->ev0);
    if (st->ev1) (void)hipEventDestroy(st->ev1);
    if (st->d_swap && st->lean && st->ctx) st->ctx->temp_put(st->d_swap);
    else if (st->d_swap) (void)hipFree(st->d_swap);
    if (st->d_io) (void)hipFree(st->d_io);
    if (st->d_words) (void)hipFree(st->d_words);
    if (st->stream) (void)hipStreamDestroy(st->stream);
    delete st;
}

void ensure_io(bz3_state * st) {
    if (!st->d_io) HIP_CHECK(hipMalloc((void **)&st->d_io, st->cap));
}

u32 read_word(hipStream_t s, const u32 * d) {
    u32 v = 0;
    HIP_CHECK(hipMemcpyAsync(&v, d, 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    return v;
}

inline bool sizes_fit(size_t buffer_size, s32 lzp_size, s32 rle_size, s32 orig_size) {  // bz3_check_buffer_size, :114-122
    const size_t a = lzp_size < 0 ? 0 : (size_t)lzp_size, b = rle_size < 0 ? 0 : (size_t)rle_size, c = orig_size < 0 ? 0 : (size_t)orig_size;
    return a <= buffer_size && b <= buffer_size && c <= buffer_size;
}

// ---- headroom: the device memory the library leaves to the host program -------------------------------------------------------------------------
// The library keeps memory between calls (the workspace arena, idle pooled swap buffers; with keep-workspace a GPU-filling batch's whole ring).  The caller
// owns its memory, so there is a rule about how much the library may sit on: when a batch call returns, at least `headroom` bytes of the
// device are free (hipMemGetInfo), or the library holds nothing cached at all.  Two halves: (1) the rings are SIZED for it -- the encoder's ring of LZP
// contexts takes the free memory minus headroom minus what the arena adds on top (ring_contexts_for); (2) it is ENFORCED when a group returns
// (enforce_headroom): idle pooled swap buffers go back to the driver first, the workspace second.  bz3_hip_set_workspace_headroom / environment
// BZ3_HIP_WS_HEADROOM_MB; default 4 GiB.  Round 5 had a fixed 6 GiB margin that the arena's own slack consumed: bench.py's next 48 MiB allocation failed.
std::atomic<long long> g_ws_headroom{-1};  // bytes; -1 = the environment decides
inline size_t ws_headroom() {
    static const long long env = [] {
        const char * e = getenv("BZ3_HIP_WS_HEADROOM_MB");
        return e && *e ? (long long)strtoull(e, nullptr, 10) << 20 : (long long)4 << 30;
    }();
    const long long v = g_ws_headroom.load();
    return (size_t)(v < 0 ? env : v);
}
// LZP contexts the encoder's ring may hold.  free_b: free device memory now; have: the arena the context already owns (reused); need: per-block scratch of the
// stages; fixed: the batch's job arrays, side buffers and CM scratch in the arena; ctx_bytes: one LZP context; cap: one borrowed swap buffer (lean states: a block
// in the ring holds exactly one context and one swap buffer; others own their swap buffers, and 7/10 of what is free is the estimate of rounds 1-4).
inline size_t ring_contexts_for(size_t free_b, size_t have, size_t need, size_t fixed, size_t ctx_bytes, size_t cap, bool lean, size_t headroom) {
    const size_t total = free_b + have;
    const size_t base = need + fixed;
    if (total <= base || !ctx_bytes) return 0;
    size_t by_share = (total - base) / 10 * 7 / ctx_bytes;
    // what is left once the headroom, the arena's slack and the takes' alignment are set aside
    const size_t aside = headroom + ((size_t)513 << 20) + ((size_t)64 << 20);
    const size_t room = total > base + aside ? total - base - aside : 0;
    if (!lean) {
        const size_t by_room = room / ctx_bytes;
        return by_share < by_room ? by_share : by_room;
    }
    return room / (ctx_bytes + cap);
}
void enforce_headroom(DeviceCtx * ctx, hipStream_t s);

// Lean states own no swap buffer: they borrow one from the device's pool while a stage sequence needs it.
// BZ3_HIP_KEEP_WS=1 (experiment, round 5's first measurement): a lean batch's workspace survives the call -- the decode call that follows reuses the
// encode call's arena and carves the swap buffers of its tail windows from it -- instead of one hipFree and two multi-GB hipMallocs per round trip
// (30-45 ms per GiB: profiles/r04_first_touch.txt).  Read once.
#ifndef BZ3_FRONT_DUO_DEFAULT
#define BZ3_FRONT_DUO_DEFAULT 0
#endif
std::atomic<int> g_keep_ws{-1};  // bz3_hip_set_keep_workspace: -1 = the environment decides
inline bool keep_workspace() {
    static const bool env_on = [] {
        const char * e = getenv("BZ3_HIP_KEEP_WS");
        return e && atoi(e) != 0;
    }();
    const int v = g_keep_ws.load();
    return v < 0 ? env_on : v != 0;
}
std::atomic<int> g_front_duo{-1};  // bz3_hip_set_front_end_duo: -1 = the environment decides (BZ3_HIP_FRONT_DUO, read once)
inline bool front_end_duo() {
    static const int env_on = [] {
        const char * e = getenv("BZ3_HIP_FRONT_DUO");
        return e ? (atoi(e) != 0 ? 1 : 0) : BZ3_FRONT_DUO_DEFAULT;
    }();
    const int v = g_front_duo.load();
    return v < 0 ? env_on != 0 : v != 0;
}
inline void lean_borrow(bz3_state * st) {
    if (st->lean && !st->d_swap) st->d_swap = st->ctx->temp_get(st->cap);
}
inline void lean_return(bz3_state * st) {
    if (st->lean && st->d_swap) {
        st->ctx->temp_put(st->d_swap);
        st->d_swap = nullptr;
    }
}
constexpr u32 CM_SIDE_BYTES = 64 * 1024;  // per block: where in-place CM output goes should it ever catch up with its input

// ======================================================================================================
// encode.  A call (one block or a batch) runs in three phases per GPU:
//   1. per block, whole-GPU kernels, one block after the other: CRC, mRLE, LZP, BWT        (encode_front)
//   2. ONE launch of the CM kernel, one workgroup (= one CU) per block                      (cm_encode_batch)
//   3. per block: header, copy back into the caller's buffer if the ping-pong ended there   (encode_finish)
// ======================================================================================================
// Phase 1a: CRC, mRLE, LZP preparation (hash links + static events).  c receives the LZP context.
void encode_front_a(bz3_state * st, u8 * buf, s32 data_size, Arena & arena, Arena & ctx_arena, LzpEncodeCtx & c) {
    st->pending = bz3_state::FAILED;
    st->result = -1;
    c.active = false;
    if (st->skip) return;  // last_error as on_failure left it
    if (data_size > st->block_size || data_size < 0) {  // :588-591 (a negative size would walk off the buffer)
        st->last_error = BZ3_ERR_DATA_TOO_BIG;
        return;
    }
    hipStream_t s = st->xs;
    for (float & x : st->t) x = 0.f;
    double t0 = now_ms();
    crc32c_device(buf, (u64)data_size, 1u, st->ctx->d_crc, st->d_words, s);  // :593
    st->user = buf;
    st->size = data_size;
    if (data_size < 64) {  // :596-601 (last_error is left untouched on this path)
        launch(k_store_small, dim3(1), dim3(64), 0, s, buf, (u32)data_size, (const u32 *)(st->d_words + 1));
        st->pending = bz3_state::ENC_STORED;
        return;
    }
    if (st->timed) {
        HIP_CHECK(hipStreamSynchronize(s));
        st->t[BZ3_HIP_T_CRC] = (float)(now_ms() - t0);
    }

    u32 n = (u32)data_size;
    lean_borrow(st);
    u8 *b1 = buf, *b2 = st->d_swap;
    st->model = 0;
    st->rle_size = 0;
    t0 = now_ms();
    {  // :609-614
        MrleEncScratch sc;
        const size_t mk = arena.mark();
        mrle_encode_size(b1, n, sc, arena, s);
        st->rle_size = (s32)(32u + read_word(s, sc.total));
        if (st->rle_size < (s32)n) {
            mrle_encode_write(b1, n, sc, b2, s);
            if (st->timed) HIP_CHECK(hipStreamSynchronize(s));
            u8 * tmp = b1; b1 = b2; b2 = tmp;
            n = (u32)st->rle_size;
            st->model |= 4;
        }
        arena.release(mk);
    }
    st->t[BZ3_HIP_T_RLE] = (float)(now_ms() - t0);
    st->b1 = b1;
    st->b2 = b2;
    st->n_cm = n;
    t0 = now_ms();
    lzp_encode_prepare(b1, n, c, ctx_arena, arena, s);  // :616 (first third)
    st->t[BZ3_HIP_T_LZP] = (float)(now_ms() - t0);
    st->pending = bz3_state::ENC_CODED;
}

// Phase 1b (after the LZP drivers of the window have run): LZP emission, BWT, header.
void encode_front_b(bz3_state * st, Arena & arena, const LzpEncodeCtx & c, float driver_ms) {
    if (st->pending != bz3_state::ENC_CODED) return;
    st->pending = bz3_state::FAILED;
    hipStream_t s = st->xs;
    u8 *b1 = st->b1, *b2 = st->b2;
    u32 n = st->n_cm;
    double t0 = now_ms();
    const s32 lzp_size = lzp_encode_finish(c, b2, arena, s);  // :616-621
    if (lzp_size > 0 && lzp_size < (s32)n) {
        u8 * tmp = b1; b1 = b2; b2 = tmp;
        n = (u32)lzp_size;
        st->model |= 2;
    }
    st->lzp_size = lzp_size;
    st->t[BZ3_HIP_T_LZP] += driver_ms + (float)(now_ms() - t0);

    t0 = now_ms();
    u32 * const d_idx = st->d_words + 6;  // the primary index never visits the host
    u32 * const d_out8 = st->d_words + 8;  // bytes outside the block's 40 most frequent values: decides the CM kernel variant (run_cm_jobs)
    if (!st->lean) {
        (void)bwt_forward(b1, n, b2, arena, s, &st->bwt, d_idx, d_out8);  // :623-627
    } else {
        // Lean state: the coder will work IN PLACE in the caller's buffer (capacity bz3_bound(size), libbz3.h:172-174):
        // the BWT output goes to the END of that buffer, the coded bytes grow from its start (cm.hip CmSink).
        u8 * tail = st->user + bz3_bound((size_t)st->size) - n;
        if (b1 != st->user) {
            (void)bwt_forward(b1, n, tail, arena, s, &st->bwt, d_idx, d_out8);
        } else {
            (void)bwt_forward(b1, n, b2, arena, s, &st->bwt, d_idx, d_out8);
            HIP_CHECK(hipMemcpyAsync(tail, b2, n, hipMemcpyDeviceToDevice, s));
        }
        b1 = st->user;  // receives header + coded bytes
        b2 = tail;      // CM input
    }
    if (st->timed) {
        HIP_CHECK(hipStreamSynchronize(s));
        st->t[BZ3_HIP_T_BWT] = (float)(now_ms() - t0);
    }
    s32 overhead = 2;  // :630-632
    if (st->model & 2) overhead++;
    if (st->model & 4) overhead++;
    launch(k_write_header, dim3(1), dim3(64), 0, s, b1, (const u32 *)(st->d_words + 1), (const u32 *)d_idx, (u32)st->model, (u32)lzp_size, (u32)st->rle_size);  // :641-647
    // The swap buffer goes back to the pool.  Whoever borrows it next is a later block of this group -- the group holds the device's
    // mutex for the whole call and every kernel that touches the buffer, this block's and the next borrower's, is launched on the group's
    // ONE stream (the side streams only run LZP drivers, behind events recorded on it) --, so stream order is what protects it; rounds 1-3
    // waited for the stream here, once per block.
    if (st->lean && !st->hold_swap) lean_return(st);
    st->b1 = b1;
    st->b2 = b2;
    st->n_cm = n;
    st->overhead = overhead;
    st->pending = bz3_state::ENC_CODED;
}

void encode_finish(bz3_state * st, float cm_ms) {
    const bz3_state::Pending p = st->pending;
    st->pending = bz3_state::NONE;
    if (p == bz3_state::FAILED || p == bz3_state::NONE) return;
    HIP_CHECK(hipStreamSynchronize(st->xs));
    if (p == bz3_state::ENC_STORED) {
        st->result = st->size + 8;
        return;
    }
    st->t[BZ3_HIP_T_CM] = cm_ms;
    const u32 coded = read_word(st->xs, st->d_words + 2);
    if (coded == 0xFFFFFFFFu) {  // in-place coding ran out of side buffer (the output outgrew bz3_bound's slack by > 64 KiB mid-block)
        st->last_error = BZ3_ERR_BWT;
        return;
    }
    const s32 total = (s32)coded + st->overhead * 4 + 1;
    if (st->lean) {
        const u32 sw = read_word(st->xs, st->d_words + 3);
        if (sw != 0xFFFFFFFFu && coded > sw)  // part of the coded bytes waited in the side buffer until the input was dead
            HIP_CHECK(hipMemcpyAsync(st->user + st->overhead * 4 + 1 + sw, st->side, coded - sw, hipMemcpyDeviceToDevice, st->xs));
        HIP_CHECK(hipStreamSynchronize(st->xs));
    }
    st->last_error = BZ3_OK;  // :649
    if (st->b1 != st->user) {  // :651
        double t0 = now_ms();
        HIP_CHECK(hipMemcpyAsync(st->user, st->b1, (size_t)total, hipMemcpyDeviceToDevice, st->xs));
        HIP_CHECK(hipStreamSynchronize(st->xs));
        st->t[BZ3_HIP_T_COPY] += (float)(now_ms() - t0);
    }
    st->result = total;
}

std::atomic<unsigned> g_headroom_trims{0}, g_headroom_releases{0};  // statistics (bz3_hip_debug_headroom_events)
// Second half of the headroom rule (see ws_headroom): called with the context's mutex held when a group's call ends.
void enforce_headroom(DeviceCtx * ctx, hipStream_t s) {
    const size_t h = ws_headroom();
    size_t free_b = 0, total_b = 0;
    if (!h || hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b >= h) return;
    if (ctx->temp_idle_bytes() > 0) {
        ctx->temp_trim();
        g_headroom_trims.fetch_add(1);
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b >= h) return;
    }
    if (ctx->ws) {
        (void)hipStreamSynchronize(s);
        (void)hipDeviceSynchronize();  // side streams included
        (void)hipFree(ctx->ws);
        ctx->ws = nullptr;
        ctx->ws_cap = 0;
        g_headroom_releases.fetch_add(1);
    }
}

// Shape of the encoder's front-end pipeline: `contexts` LZP contexts fit into the memory budget, the batch has n blocks.
// ns context slots (2..DeviceCtx::RING_SLOTS; up to DeviceCtx::AUX when forced) of `window` blocks each.  A window's drivers hide behind the whole-GPU work of ns-1 other
// windows, so the drivers' share of the pace is T_driver / ((ns-1) * window) per block: more slots of fewer blocks get more out
// of the same memory (4 x 4 hides as much as 2 x 12).  Small batches keep the two-slot form (nothing to overlap with anyway).
// BZ3_HIP_LZP_PIPE="window,slots" overrides (tests / experiments).
void pipeline_shape(s32 contexts, s32 n, s32 & window, s32 & ns) {
    if (contexts < 2) contexts = 2;
    ns = 2;
    for (s32 cand = DeviceCtx::RING_SLOTS; cand > 2; cand--)
        if (contexts / cand >= 3 && n >= cand * 3) {
            ns = cand;
            break;
        }
    window = contexts / ns;
    if (window > 8) window = 8;
    if (const char * e = getenv("BZ3_HIP_LZP_PIPE")) {
        int w = 0, q = 0;
        if (sscanf(e, "%d,%d", &w, &q) == 2 && w >= 1 && q >= 2 && q <= DeviceCtx::AUX) {
            window = w;
            ns = q;
        }
    }
    if (window > n) window = n;
    if (window < 1) window = 1;
}

// Runs `n` blocks whose states live on ONE device.  bufs are device pointers.
void encode_group(bz3_state ** sts, u8 ** bufs, const s32 * sizes, s32 n) {
    if (n <= 0) return;
    bz3_state * lead = sts[0];
    DeviceGuard g(lead->device);
    std::lock_guard<std::mutex> lk(lead->ctx->mu);
    // One execution stream per call: the whole-GPU phases of the blocks run one after the other anyway, and stream
    // order is what protects the arena's scratch regions, which consecutive blocks reuse while earlier kernels are
    // still in flight.
    for (s32 i = 0; i < n; i++) {
        sts[i]->xs = lead->stream;
        sts[i]->timed = i == 0;  // stage timings (bz3_hip_last_timings) are sampled on the group's first block: timing a stage means waiting for the stream
    }
    size_t need = 0, need_a = 0, need_b = 0;
    for (s32 i = 0; i < n; i++) {
        const u64 nb = (u64)(sizes[i] > 0 ? sizes[i] : 0) + 64;
        const size_t w = workspace_bytes_for(nb), wa = front_a_scratch_bytes(nb), wb = front_b_scratch_bytes(nb);
        if (w > need) need = w;
        if (wa > need_a) need_a = wa;
        if (wb > need_b) need_b = wb;
    }
    // Two-thread front end (round 6, bz3_hip_set_front_end_duo / BZ3_HIP_FRONT_DUO; see below): phase A and phase B have a scratch region each
    const bool duo_wanted = front_end_duo() && n >= 16;
    if (duo_wanted) need = need_a + need_b + 65536;
    // LZP drivers are serial single-workgroup kernels (0.75 s for a 256 MiB text block alone, ~1.2 s beside the whole-GPU
    // kernels of other blocks, however many run side by side).  The blocks go through the front end in WINDOWS, software-pipelined
    // over a ring of `ns` context slots with one side stream each: while the drivers of windows k-ns+2 .. k run on their side
    // streams, this thread prepares window k+1 and finishes window k-ns+2 (LZP emission, BWT, header) on the group's stream, so a
    // window's drivers hide behind ns-1 windows' worth of whole-GPU work.  Round 2 ran two slots of 6 blocks at 768 x 256 MiB:
    // 128 windows x 1.23 s of drivers = 158 s against 146 s of whole-GPU work -- the driver chain, not the kernels, set the pace
    // of the front end (profiles/r02_kernel_stats_bench_768x256MiB.txt: k_lzp_driver 137 calls, 1.23 s each).
    u64 n_max = 64;
    for (s32 i = 0; i < n; i++)
        if (sizes[i] > 0 && (u64)sizes[i] > n_max) n_max = (u64)sizes[i];
    const size_t ctx_bytes = lzp_encode_ctx_bytes(n_max + 64) + 65536;
    // contexts = what fits into 7/10 of the memory that is free right now beside the sorter's workspace (the rest: the blocks' borrowed
    // swap buffers, allocator slack).  A workspace that grew far beyond the sorter's needs for this is handed back when the call
    // ends (below): the decode call that follows needs the room for staged payloads and the swap buffers of its tail windows.
    const size_t headroom = ws_headroom();
    const size_t fixed = (size_t)n * (sizeof(CmEncodeJob) + CM_SIDE_BYTES + 256) + cm_scratch_bytes((size_t)n) + 65536 + (size_t)DeviceCtx::AUX * 8 * (sizeof(LzpDriverJob) + 256);
    size_t contexts = ((size_t)32 << 30) / ctx_bytes;
    {
        // Swap buffers the previous call left idle in the pool (a decode call's tail ring holds up to 4 x 16 of them, 17 GB at 256 MiB)
        // are memory this call's ring of LZP contexts cannot use: round 4's full-size run had a ring of 4 x 4 contexts and an encode
        // front end 14 s longer in its SECOND step than in its first for this.  What the front end borrows again is a few dozen buffers.
        // Round 5: only when that memory is MISSING -- an ordinary lean batch whose ring gets its full shape beside the pool keeps the
        // buffers it is about to borrow again (a free + malloc cycle per buffer and call otherwise, 30-45 ms per GiB).
        size_t free_b = 0, total_b = 0;
        const s32 full = DeviceCtx::RING_SLOTS * 8;  // 4 slots x 8 blocks: what pipeline_shape grants at most
        if (lead->lean && lead->ctx->temp_idle_bytes() > 0 && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            if (ring_contexts_for(free_b, lead->ctx->ws_cap, need, fixed, ctx_bytes, lead->cap, true, headroom) < (size_t)(n < full ? n : full)) lead->ctx->temp_trim();
        }
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            contexts = ring_contexts_for(free_b, lead->ctx->ws_cap, need, fixed, ctx_bytes, lead->cap, lead->lean, headroom);
    }
    s32 window = 1, ns = 2;
    pipeline_shape((s32)(contexts < 1024 ? contexts : 1024), n, window, ns);
    lead->ctx->ensure_aux();
    hipStream_t s = lead->stream;
    DrainOnUnwind drain{s, lead->ctx->aux, DeviceCtx::AUX};
    // BZ3_HIP_FRONT_REST=1 (experiment, read once): the front end's whole-GPU kernels also keep off the CUs reserved for the side streams' LZP drivers
    static const bool front_rest = [] { const char * e = getenv("BZ3_HIP_FRONT_REST"); return e && atoi(e) != 0; }();
    if (front_rest && lead->ctx->rest && n > 1) {
        s = lead->ctx->rest;
        for (s32 i = 0; i < n; i++) sts[i]->xs = s;
    }
    DrainOnUnwind drain_rest{s == lead->stream ? nullptr : s, nullptr, 0};
    Arena arena;
    for (;;) {  // hipMemGetInfo's figure is not a promise (fragmentation, another process on the device): shrink the ring before giving up
        try {
            arena = lead->ctx->arena_for(need + (size_t)ns * (size_t)window * (ctx_bytes + sizeof(LzpDriverJob) + 256) +
                                         (size_t)n * (sizeof(CmEncodeJob) + CM_SIDE_BYTES + 256) + cm_scratch_bytes((size_t)n) + 65536);
            break;
        } catch (const HipError &) {  // out of memory, or whatever else the runtime answers to a size it does not like
            if (ns == 2 && window == 1) throw;
            (void)hipGetLastError();
            if (ns > 2) ns = 2;
            else window = (window + 1) / 2;
        }
    }
    g_front_end_ring.store(window | (ns << 16));
    CmEncodeJob * d_jobs = arena.take<CmEncodeJob>((size_t)n);
    u8 * sides = arena.take<u8>((size_t)n * CM_SIDE_BYTES);  // lean states only
    struct Window {
        s32 w0 = 0, w1 = 0;
        Arena slot;                        // the LZP contexts of the window's blocks
        LzpDriverJob * d_lz = nullptr;
        std::vector<LzpEncodeCtx> ctxs;
        std::vector<LzpDriverJob> lz;      // host copy: stays alive until the window is finished
    } win[DeviceCtx::AUX];
    for (int k = 0; k < ns; k++) {
        win[k].slot.base = arena.take<char>((size_t)window * ctx_bytes);
        win[k].slot.cap = (size_t)window * ctx_bytes;
        win[k].d_lz = arena.take<LzpDriverJob>((size_t)window);
    }
    std::vector<CmEncodeJob> jobs;
    std::vector<s32> job_owner;
    const s32 nwin = (n + window - 1) / window;
    const s32 lag = ns - 1;  // window k is finished in iteration k + lag
    static const bool trace_rings = getenv("BZ3_HIP_TRACE_RINGS") != nullptr;  // (diagnosis, read once: see decode_group)
    double tr_prep = 0, tr_wait = 0, tr_fin = 0, tr_t0 = now_ms();
    // Phase A of window k on stream sp with scratch `scr`: CRC, mRLE, LZP preparation of its blocks, then its drivers on the slot's side stream behind `ev`.
    auto prepare = [&](s32 k, hipStream_t sp, Arena & scr, hipEvent_t ev) {
        const int q = (int)(k % ns);
        Window & w = win[q];
        w.w0 = k * window;
        w.w1 = (w.w0 + window < n) ? w.w0 + window : n;
        w.slot.used = 0;  // its previous tenant (window k-ns) has been finished
        w.ctxs.assign((size_t)(w.w1 - w.w0), LzpEncodeCtx());
        w.lz.clear();
        for (s32 i = w.w0; i < w.w1; i++) {
            sts[i]->xs = sp;
            sts[i]->hold_swap = false;  // (a call that was left by an exception may have left it set)
            encode_front_a(sts[i], bufs[i], sizes[i], scr, w.slot, w.ctxs[(size_t)(i - w.w0)]);
            if (sts[i]->pending == bz3_state::ENC_CODED && w.ctxs[(size_t)(i - w.w0)].active) w.lz.push_back(lzp_driver_job(w.ctxs[(size_t)(i - w.w0)]));
        }
        HIP_CHECK(hipEventRecord(ev, sp));  // the prepares above are in flight on sp
        if (!w.lz.empty()) {
            hipStream_t sq = lead->ctx->aux[q];
            HIP_CHECK(hipStreamWaitEvent(sq, ev, 0));
            HIP_CHECK(hipEventRecord(lead->ctx->ev_d0[q], sq));
            lzp_driver_batch(w.lz.data(), w.d_lz, (u32)w.lz.size(), sq);
            HIP_CHECK(hipEventRecord(lead->ctx->ev_d1[q], sq));
        }
    };
    // Phase B of window j on stream sf with scratch `scr`: waits for its drivers, then LZP emission, BWT, header per block; the CM jobs in block order.
    // held != nullptr: lean states keep their swap buffers (the caller hands them back behind an event: two-thread form).
    auto finish = [&](s32 j, hipStream_t sf, Arena & scr, std::vector<bz3_state *> * held) {
        const int q = (int)(j % ns);
        Window & w = win[q];
        float driver_ms = 0.f;
        if (!w.lz.empty()) {
            const double t0 = now_ms();
            HIP_CHECK(hipEventSynchronize(lead->ctx->ev_d1[q]));
            (void)hipEventElapsedTime(&driver_ms, lead->ctx->ev_d0[q], lead->ctx->ev_d1[q]);
            tr_wait += now_ms() - t0;
        }
        for (s32 i = w.w0; i < w.w1; i++) {
            sts[i]->xs = sf;
            sts[i]->hold_swap = held != nullptr;
            encode_front_b(sts[i], scr, w.ctxs[(size_t)(i - w.w0)], driver_ms);
            sts[i]->hold_swap = false;
            if (sts[i]->pending == bz3_state::ENC_CODED) {
                CmEncodeJob j2{dev_addr(sts[i]->b2), dev_addr(sts[i]->b1 + sts[i]->overhead * 4 + 1), dev_addr(sts[i]->d_words + 2), sts[i]->n_cm, 0u};  // :634-638
                if (sts[i]->lean) {  // in place: input at the end of the caller's buffer, output behind the header
                    sts[i]->side = sides + (size_t)i * CM_SIDE_BYTES;
                    j2.gap = (u32)(sts[i]->b2 - (sts[i]->b1 + sts[i]->overhead * 4 + 1));
                    j2.side = dev_addr(sts[i]->side);
                    j2.side_cap = CM_SIDE_BYTES;
                }
                jobs.push_back(j2);
                job_owner.push_back(i);
            }
            if (held) {
                if (sts[i]->lean && sts[i]->d_swap) held->push_back(sts[i]);
            } else {
                lean_return(sts[i]);  // blocks that left the pipeline early (stored, failed) still hold their swap buffer
            }
        }
    };
    const bool duo = duo_wanted && nwin >= 3 && lead->ctx->aux_ready && s == lead->stream;  // (aux_ready: ctx->duo exists -- a null handle under the emulator)
    if (!duo) {
        for (s32 k = 0; k < nwin + lag; k++) {
            const double tr_a = now_ms();
            if (k < nwin) prepare(k, s, arena, lead->ctx->ev_prep);
            const double tr_b = now_ms();
            tr_prep += tr_b - tr_a;
            if (k >= lag) finish(k - lag, s, arena, nullptr);
            tr_fin += now_ms() - tr_b;
        }
    } else {
        // ---- two-thread front end (round 6).  Phase A (14 ms of whole-GPU kernels per 256 MiB block) and phase B (56 ms) both stop for read-backs the host
        // needs (mRLE size; LZP size, one per sorter pass), and while one waits nothing of THIS stream runs.  A second host thread takes phase A on a stream of its
        // own (ctx->duo) with its own scratch region and runs up to ns - 1 windows ahead of this thread's phase B: the kernels of one fill the other's bubbles.
        //   slot reuse      A prepares window k once B has FINISHED window k - ns (host: `finished`) and behind B's kernels of that window (stream: evB[k - ns]);
        //   hand-over       B starts window j once A has prepared it (host: `prepared`), behind A's kernels (stream: evA[j]); the drivers' events as before;
        //   swap buffers    lean states' buffers go back to the pool when B has recorded evB[j] (`held`), and A's stream waits for the latest such event before the
        //                   kernels of blocks that may have borrowed them: with two streams stream order alone no longer protects a re-borrowed buffer.
        // Output bytes cannot depend on any of this: every block still runs A then B, each on one stream, B behind A's event.
        g_front_end_ring.fetch_or(1 << 29);
        hipStream_t sA = lead->ctx->duo;
        DrainOnUnwind drain_duo{sA, nullptr, 0};
        Arena scr_a;
        scr_a.base = arena.take<char>(need_a);
        scr_a.cap = need_a;
        struct Events {
            std::vector<hipEvent_t> e;
            ~Events() {
                for (hipEvent_t x : e)
                    if (x) (void)hipEventDestroy(x);
            }
        } evs;
        evs.e.assign((size_t)2 * (size_t)nwin, nullptr);
        for (auto & x : evs.e) HIP_CHECK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
        hipEvent_t * evA = evs.e.data(), * evB = evs.e.data() + nwin;
        struct Shared {
            std::mutex m;
            std::condition_variable cv;
            s32 prepared = 0, finished = 0;
            bool abort = false;
            std::exception_ptr err;
            std::mutex pool;              // borrowing a window's buffers / returning a window's buffers + publishing the event that covers them
            hipEvent_t last_return = nullptr;
            double prep_ms = 0;
        } sh;
        const int device = lead->device;
        std::thread ta([&] {
            try {
                DeviceGuard gd(device);
                for (s32 k = 0; k < nwin; k++) {
                    {
                        std::unique_lock<std::mutex> ul(sh.m);
                        sh.cv.wait(ul, [&] { return sh.abort || sh.finished >= k - ns + 1; });
                        if (sh.abort) return;
                    }
                    const double t0 = now_ms();
                    if (k >= ns) HIP_CHECK(hipStreamWaitEvent(sA, evB[k - ns], 0));
                    {
                        hipEvent_t ev = nullptr;
                        const s32 w0 = k * window, w1 = (w0 + window < n) ? w0 + window : n;
                        {
                            std::lock_guard<std::mutex> pl(sh.pool);
                            for (s32 i = w0; i < w1; i++) lean_borrow(sts[i]);
                            ev = sh.last_return;
                        }
                        if (ev) HIP_CHECK(hipStreamWaitEvent(sA, ev, 0));
                    }
                    prepare(k, sA, scr_a, evA[k]);
                    {
                        std::lock_guard<std::mutex> ul(sh.m);
                        sh.prepared = k + 1;
                        sh.prep_ms += now_ms() - t0;
                    }
                    sh.cv.notify_all();
                }
            } catch (...) {
                {
                    std::lock_guard<std::mutex> ul(sh.m);
                    sh.err = std::current_exception();
                    sh.abort = true;
                }
                sh.cv.notify_all();
            }
        });
        struct Joiner {  // whatever happens on this thread, phase A's thread is joined before its std::thread object dies
            std::thread & t;
            Shared & sh;
            int live = std::uncaught_exceptions();
            ~Joiner() {
                if (std::uncaught_exceptions() > live) {  // this thread is unwinding: phase A stops at its next window
                    std::lock_guard<std::mutex> ul(sh.m);
                    sh.abort = true;
                }
                sh.cv.notify_all();
                if (t.joinable()) t.join();
            }
        } joiner{ta, sh};
        std::vector<bz3_state *> held;
        for (s32 j = 0; j < nwin; j++) {
            const double tr_b = now_ms();
            {
                std::unique_lock<std::mutex> ul(sh.m);
                sh.cv.wait(ul, [&] { return sh.abort || sh.prepared >= j + 1; });
                if (sh.abort) break;
            }
            HIP_CHECK(hipStreamWaitEvent(s, evA[j], 0));
            held.clear();
            finish(j, s, arena, &held);
            HIP_CHECK(hipEventRecord(evB[j], s));
            {
                std::lock_guard<std::mutex> pl(sh.pool);
                for (bz3_state * st : held) lean_return(st);
                sh.last_return = evB[j];
            }
            {
                std::lock_guard<std::mutex> ul(sh.m);
                sh.finished = j + 1;
            }
            sh.cv.notify_all();
            tr_fin += now_ms() - tr_b;
        }
        ta.join();
        if (sh.err) std::rethrow_exception(sh.err);
        HIP_CHECK(hipStreamSynchronize(sA));
        tr_prep = sh.prep_ms;
        for (s32 i = 0; i < n; i++) sts[i]->xs = s;
    }
    if (trace_rings)
        fprintf(stderr, "[bz3 rings] encode front end: %d blocks, %d windows of %d x %d slots: %.1f ms = CRC / mRLE / LZP prepare %.1f + waiting for a window's LZP drivers %.1f + LZP emit / BWT / header %.1f\n",
                (int)n, (int)nwin, (int)window, (int)ns, now_ms() - tr_t0, tr_prep, tr_wait, tr_fin - tr_wait);
    if (s != lead->stream) {  // the CM launch needs every CU: back to the group's unmasked stream, behind the front end
        HIP_CHECK(hipStreamSynchronize(s));
        s = lead->stream;
        for (s32 i = 0; i < n; i++) sts[i]->xs = s;
    }
    // which blocks do not fit a row cache: the bytes outside the block's 40 most frequent values (bwt.hip k_bwt_outside) -- one small read-back per
    // block, all behind the one synchronisation the CM launch needs anyway
    std::vector<char> to_full(jobs.size(), 0);
    if (jobs.size() > (size_t)lead->ctx->cus) {  // (only batches that would take a row-cache variant)
        std::vector<u32> outside(jobs.size(), 0u);
        for (size_t k = 0; k < jobs.size(); k++) HIP_CHECK(hipMemcpyAsync(&outside[k], sts[job_owner[k]]->d_words + 8, 4, hipMemcpyDeviceToHost, lead->stream));
        HIP_CHECK(hipStreamSynchronize(lead->stream));
        // Only the hopeless: more than a QUARTER of the block outside its 40 most frequent values (random data: 84 %).  The histogram is global, a row
        // cache lives on what the BWT output uses LOCALLY: the bench's calibrated text has 3 % of its bytes outside the top 40 and misses 0.3 % of its
        // rows -- round 5's first form of this test (outside > 512 + n / 64) sent all 768 text blocks to the whole-model kernel, one per CU, 183 s instead
        // of 65 (profiles/r05_full_size_call5_misrouted.txt).  What lies between is left to the kernels' own give-up test.
        for (size_t k = 0; k < jobs.size(); k++) to_full[k] = outside[k] > jobs[k].n / 4u ? 1 : 0;
    }
    const float cm_ms = run_cm_jobs(lead->ctx, arena, jobs, d_jobs, lead->stream, lead->ev0, lead->ev1,
                                    [](const CmEncodeJob * j, u32 nj, hipStream_t st, int variant) { cm_encode_batch(j, nj, st, variant); }, &to_full);
    for (s32 i = 0; i < n; i++) encode_finish(sts[i], cm_ms);
    // Hand-back only where it matters: a workspace that a GPU-filling batch grew to tens of GB (the decode call that follows needs the
    // room).  A stream of ordinary batches (the CLI, bz3_hip_encode_stream) keeps its workspace: a multi-GB hipMalloc plus a
    // device-synchronising hipFree per call would cost more than the memory is worth.
    size_t slack = (size_t)16 << 30;
    if (const char * e = getenv("BZ3_HIP_WS_KEEP_MB")) slack = (size_t)strtoull(e, nullptr, 10) << 20;  // tests only
    if (lead->ctx->ws_cap > 2 * need + slack && !(keep_workspace() && lead->lean)) {  // mostly LZP contexts of a large batch: hand the memory back (see above)
        HIP_CHECK(hipStreamSynchronize(s));  // the side streams are idle: every driver launch has been waited for
        (void)hipFree(lead->ctx->ws);
        lead->ctx->ws = nullptr;
        lead->ctx->ws_cap = 0;
        g_front_end_ring.fetch_or(1 << 30);
    }
    enforce_headroom(lead->ctx, lead->stream);
}

// ======================================================================================================
// decode.  Phases per GPU: validate headers -> ONE CM launch (one CU per block) -> per block inverse BWT
// (whole GPU) -> ONE LZP-decode launch (one workgroup per block) -> per block mRLE decode + CRC check.
// ======================================================================================================
// hdr: host copy of the first min(17, buffer_size) bytes of the block.
void decode_front(bz3_state * st, u8 * buf, size_t buffer_size, s32 compressed_size, s32 orig_size, const u8 * hdr) {
    st->pending = bz3_state::FAILED;
    st->result = -1;
    if (st->skip) return;  // last_error as on_failure left it
    if (buffer_size < 9 || buffer_size < (size_t)compressed_size) {  // :658-661 (s32 -> size_t as in the reference)
        st->last_error = BZ3_ERR_DATA_SIZE_TOO_SMALL;
        return;
    }
    const u32 crc = rd_le32(hdr);
    const s32 bwt_idx = (s32)rd_le32(hdr + 4);
    const size_t bound = bz3_bound((size_t)st->block_size);
    if (compressed_size < 0 || (size_t)compressed_size > bound) {  // :667-670
        st->last_error = BZ3_ERR_MALFORMED_HEADER;
        return;
    }
    hipStream_t s = st->xs;
    for (float & x : st->t) x = 0.f;
    st->user = buf;
    st->buffer_size = buffer_size;
    st->crc = crc;
    if (bwt_idx == -1) {  // stored block, :672-692
        if (compressed_size - 8 > 64 || compressed_size < 8) {
            st->last_error = BZ3_ERR_MALFORMED_HEADER;
            return;
        }
        if ((size_t)(compressed_size - 8) > buffer_size) {
            st->last_error = BZ3_ERR_DATA_SIZE_TOO_SMALL;
            return;
        }
        st->size = compressed_size - 8;
        launch(k_unstore_small, dim3(1), dim3(128), 0, s, buf, (u32)st->size);
        crc32c_device(buf, (u64)st->size, 1u, st->ctx->d_crc, st->d_words, s);
        st->pending = bz3_state::DEC_STORED;
        return;
    }
    const s32 model = (s8)hdr[8];
    const size_t need = 9 + (size_t)((model & 2) * 4) + (size_t)((model & 4) * 4);  // :697 (9 / 17 / 25 / 33)
    if (buffer_size < need) {
        st->last_error = BZ3_ERR_DATA_SIZE_TOO_SMALL;
        return;
    }
    s32 lzp_size = -1, rle_size = -1, p = 0;
    if (model & 2) lzp_size = (s32)rd_le32(hdr + 9 + 4 * p++);
    if (model & 4) rle_size = (s32)rd_le32(hdr + 9 + 4 * p++);
    p += 2;
    compressed_size -= p * 4 + 1;
    if (((model & 2) && (lzp_size < 0 || (size_t)lzp_size > bound)) || ((model & 4) && (rle_size < 0 || (size_t)rle_size > bound))) {  // :710-714
        st->last_error = BZ3_ERR_MALFORMED_HEADER;
        return;
    }
    if (orig_size < 0 || (size_t)orig_size > bound) {  // :716-719
        st->last_error = BZ3_ERR_MALFORMED_HEADER;
        return;
    }
    const s32 size_before_bwt = (model & 2) ? lzp_size : (model & 4) ? rle_size : orig_size;  // :724-729
    if (!sizes_fit(buffer_size, lzp_size, rle_size, orig_size)) {  // :734-737
        st->last_error = BZ3_ERR_DATA_SIZE_TOO_SMALL;
        return;
    }
    st->bwt_idx = bwt_idx;
    st->model = model;
    st->lzp_size = lzp_size;
    st->rle_size = rle_size;
    st->orig_size = orig_size;
    st->size_before_bwt = size_before_bwt;
    st->cm_in = buf + p * 4 + 1;  // :742-747
    st->cm_in_size = (u32)(compressed_size < 0 ? 0 : compressed_size);
    st->pending = bz3_state::DEC_CODED;
}

// After the CM kernel: index checks and inverse BWT.  Leaves the data in st->b1, the free buffer in st->b2.
// Returns false when the block failed.
bool decode_unbwt(bz3_state * st, Arena & arena, float cm_ms) {
    hipStream_t s = st->xs;
    st->t[BZ3_HIP_T_CM] = cm_ms;
    const s32 n = st->size_before_bwt;
    if (st->bwt_idx > n) {  // :750-753
        st->last_error = BZ3_ERR_MALFORMED_HEADER;
        return false;
    }
    u8 *b1 = st->d_swap, *b2 = st->user;  // after the swap of :748
    if (st->lean) {  // lean state: the coder wrote into the caller's buffer; the borrowed swap buffer receives the text
        b1 = st->user;
        b2 = st->d_swap;
    }
    const double t0 = now_ms();
    // libsais_unbwt's own argument checks (include/libsais.h:5210-5232)
    if (n <= 1) {
        if (st->bwt_idx != n) { st->last_error = BZ3_ERR_BWT; return false; }
        if (n == 1) HIP_CHECK(hipMemcpyAsync(b2, b1, 1, hipMemcpyDeviceToDevice, s));
    } else {
        if (st->bwt_idx <= 0) { st->last_error = BZ3_ERR_BWT; return false; }
        bwt_inverse(b1, (u32)n, (u32)st->bwt_idx, b2, arena, s);  // :758
    }
    st->b1 = b2;
    st->b2 = b1;
    st->size_src = n;
    st->t[BZ3_HIP_T_BWT] = (float)(now_ms() - t0);
    return true;
}

// After the LZP kernel (if any): mRLE decode, size checks, copy back, CRC.
void decode_finish(bz3_state * st, Arena & arena) {
    hipStream_t s = st->xs;
    const size_t bound = bz3_bound((size_t)st->block_size);
    (void)bound;
    u8 *b1 = st->b1, *b2 = st->b2;
    s32 size_src = st->size_src;
    if (st->model & 2) {  // :767-781 (the kernel already ran; its result is in d_words[5])
        size_src = (s32)read_word(s, st->d_words + 5);
        if (size_src == -1) { st->last_error = BZ3_ERR_CRC; return; }
        if ((size_t)size_src > st->buffer_size) { st->last_error = BZ3_ERR_DATA_SIZE_TOO_SMALL; return; }
        u8 * tmp = b1; b1 = b2; b2 = tmp;
    }
    if (st->model & 4) {  // :783-792
        const double t0 = now_ms();
        bool bad = size_src < 32;  // mrled: `if (maxin < 32) return 1`
        if (!bad) {
            mrle_decode(b1, (u32)size_src, b2, (u32)st->orig_size, st->d_words + 4, arena, s);
            bad = read_word(s, st->d_words + 4) != (u32)st->orig_size;
        }
        st->t[BZ3_HIP_T_RLE] = (float)(now_ms() - t0);
        if (bad) { st->last_error = BZ3_ERR_CRC; return; }
        size_src = st->orig_size;
        u8 * tmp = b1; b1 = b2; b2 = tmp;
    }
    st->last_error = BZ3_OK;  // :794
    if (size_src > st->block_size || size_src < 0) {  // :796-799
        st->last_error = BZ3_ERR_MALFORMED_HEADER;
        return;
    }
    if (b1 != st->user) {  // :801
        const double t0 = now_ms();
        HIP_CHECK(hipMemcpyAsync(st->user, b1, (size_t)size_src, hipMemcpyDeviceToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));
        st->t[BZ3_HIP_T_COPY] += (float)(now_ms() - t0);
    }
    const double t0 = now_ms();
    crc32c_device(st->user, (u64)size_src, 1u, st->ctx->d_crc, st->d_words, s);  // :803
    const u32 got = read_word(s, st->d_words + 1);
    st->t[BZ3_HIP_T_CRC] = (float)(now_ms() - t0);
    if (got != st->crc) {
        st->last_error = BZ3_ERR_CRC;
        return;
    }
    st->result = size_src;
}

// hdrs: n x 17 bytes (host copies of the block headers).
void decode_group(bz3_state ** sts, u8 ** bufs, const size_t * buffer_sizes, const s32 * sizes, const s32 * orig_sizes, const u8 * hdrs, s32 n) {
    if (n <= 0) return;
    bz3_state * lead = sts[0];
    DeviceGuard g(lead->device);
    std::lock_guard<std::mutex> lk(lead->ctx->mu);
    hipStream_t s = lead->stream;
    for (s32 i = 0; i < n; i++) sts[i]->xs = s;  // see encode_group
    size_t need = 0;
    bool any_lean = false;
    for (s32 i = 0; i < n; i++) {
        const size_t w = workspace_bytes_for(bz3_bound((size_t)sts[i]->block_size) + 64);
        if (w > need) need = w;
        any_lean = any_lean || sts[i]->lean;
    }
    // ---- phase 1: headers ----------------------------------------------------------------------------------
    std::vector<s32> coded;
    for (s32 i = 0; i < n; i++) {
        decode_front(sts[i], bufs[i], buffer_sizes[i], sizes[i], orig_sizes[i], hdrs + 17 * (size_t)i);
        if (sts[i]->pending == bz3_state::DEC_CODED) coded.push_back(i);
    }
    // A lean state's CM output goes straight into the caller's buffer, which also holds the coded payload: that
    // payload (a fraction of the block) is staged in the workspace first.  Rounds: as many blocks per CM launch as
    // the staging budget holds (normally all of them).
    auto stage_bytes = [&](s32 i) { return sts[i]->lean ? (((size_t)sts[i]->cm_in_size + 64 + 255) & ~(size_t)255) : (size_t)0; };
    size_t stage_budget = (size_t)48 << 30;
    {
        size_t free_b = 0, total_b = 0;
        if (any_lean && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t have = lead->ctx->ws_cap;
            const size_t aside = need + ws_headroom() + ((size_t)577 << 20);  // (+ the arena's slack: see ring_contexts_for)
            const size_t room = free_b + have > aside ? free_b + have - aside : 0;
            stage_budget = room - room / 4;  // leave a quarter for the swap buffers of the tail windows
        }
    }
    std::vector<size_t> round_end;  // indices into `coded`
    size_t max_round = 0;
    for (size_t k = 0; k < coded.size();) {
        size_t bytes = 0, e = k;
        while (e < coded.size() && (e == k || bytes + stage_bytes(coded[e]) <= stage_budget)) bytes += stage_bytes(coded[e++]);
        round_end.push_back(e);
        if (bytes > max_round) max_round = bytes;
        k = e;
    }
    // tail windows, software-pipelined like the encoder's front end: the serial LZP decoders of a window (one workgroup per
    // block, ~1 s for a 256 MiB text block beside other blocks' kernels) run on a side stream while this thread drives the inverse
    // BWTs of the next windows and the mRLE / CRC stages of the previous ones on the group's stream.  Lean states hold a borrowed
    // swap buffer while their window is in flight: 64 buffers at most either way -- two slots of 32 blocks for small batches, four
    // slots of 16 for large ones (a window's decoders then hide behind three other windows' whole-GPU work: 1 s / 48 blocks
    // instead of 1 s / 32, which starts to matter once the inverse BWT of a block takes less than ~30 ms).  Round 3 tried eight slots
    // of 8 on a 128-block batch (profiles/r03_gaps_128x256MiB.txt): no gain there, where the pool's first allocations set the pace.
    s32 tail_slots = n >= 128 ? 4 : 2;
    s32 tail_window = tail_slots == 4 ? 16 : 32;
    // Round 6: 64 reserved CUs and wider windows, where the swap buffers can be had.  What the ring can hide is the whole-GPU work of the blocks whose decoders are in flight,
    // and the tail's whole-GPU kernels do not miss the CUs (they are bound by HBM line fetches).  At 768 x 256 MiB, one step each ("inverse BWTs + LZP launches" + "waiting for a
    // window's decoders" + "mRLE / CRC"; BZ3_HIP_CU_RESERVE above 64 still reserves 64: cu_masks hands out at most 8 CUs per block of 32):
    //   16 x 4 on 48 CUs: 12.7 + 3.9 + 3.0 = 19.6 s    20 x 4 on 64: 13.0 + 2.3 + 3.0 = 18.3 s    25 x 4: 12.4 + 1.5 + 3.0 = 16.9 s    30 x 4: 12.7 + 0.9 + 3.0 = 16.6 s    32 x 4: 13.2 + 0.7 + 3.0 = 16.9 s
    // (profiles/r06_call4_full_*.progress.txt, r06_call6_stdout_tail.txt, r06_tail_ring_full_size.txt): with windows of 30 up to 90 decoders share the 64 CUs, and a CU with two
    // of them still beats a window that waits.  The buffers come out of the kept arena (below) or, without keep-workspace, out of the pool -- only when the device has the room
    // beside the headroom (lean states; classic states own their swap buffers).
    if (tail_slots == 4) {
        size_t cap_max = 0;
        for (s32 i = 0; i < n; i++)
            if (sts[i]->lean && sts[i]->cap > cap_max) cap_max = sts[i]->cap;
        const size_t cap_al = (cap_max + 255) & ~(size_t)255;
        size_t free_b = 0, total_b = 0;
        const bool have_info = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
        for (s32 w : {30, 25, 20}) {
            if (n < 4 * w || lead->ctx->reserved_cus_wanted() < 64) continue;
            const size_t bufs = (size_t)4 * (size_t)w;
            const bool from_arena = keep_workspace() && any_lean && lead->ctx->ws_cap >= need + need / 16 + bufs * cap_al + ((size_t)64 << 20);
            if (!any_lean || from_arena || (have_info && free_b >= bufs * cap_max + ws_headroom() + ((size_t)2 << 30) + max_round)) {
                tail_window = w;
                break;
            }
        }
    }
    lead->ctx->ensure_aux();
    // With the CU partition the whole-GPU kernels run at their stand-alone pace and the LZP decoders become what the ring has to hide: ~1.0 s per launch
    // at 256 MiB beside the streaming kernels (0.55 s alone: their 1 MiB tables do not stay in L2), whatever the window.  Measured at full size on 48
    // reserved CUs (profiles/r05_call{6,7,8}_*): 16 x 4 waits 3.7 s of a 21.9 s tail for them, 12 x 4 7.1 s of 22.8, 8 x 8 (eight masked streams) 7.0 s of
    // 29.5 with every other phase slower too.  Four slots of 16 stay.
    if (const char * e = getenv("BZ3_HIP_TAIL_PIPE")) {  // "window,slots": tests / experiments
        int w = 0, q = 0;
        if (sscanf(e, "%d,%d", &w, &q) == 2 && w >= 1 && q >= 2 && q <= DeviceCtx::AUX) {
            tail_window = w;
            tail_slots = q;
        }
    }
    if (tail_window > n) tail_window = n;
    size_t lzp_in_window = 0;
    for (s32 w0 = 0; w0 < n; w0 += tail_window) {
        size_t c = 0;
        for (s32 i = w0; i < n && i < w0 + tail_window; i++)
            if (sts[i]->pending == bz3_state::DEC_CODED && (sts[i]->model & 2)) c++;
        if (c > lzp_in_window) lzp_in_window = c;
    }
    Arena arena = lead->ctx->arena_for(need + (size_t)tail_slots * lzp_in_window * (LZP_LUT_WORDS * 4 + sizeof(LzpDecodeJob) + 256) + (size_t)n * 256 + cm_scratch_bytes(coded.size()) + max_round + 65536);
    // ---- phase 2: the CM launches (one workgroup per block) ------------------------------------------------------
    float cm_ms = 0.f;
    for (size_t r = 0, k0 = 0; r < round_end.size(); k0 = round_end[r++]) {
        const size_t mk = arena.mark();
        std::vector<CmDecodeJob> cm_jobs;
        std::vector<char> to_full;  // a payload that hardly shrank has (nearly) every byte value live: no row cache holds that
        for (size_t k = k0; k < round_end[r]; k++) {
            bz3_state * st = sts[coded[k]];
            to_full.push_back((u64)st->cm_in_size * 10u >= (u64)(u32)st->size_before_bwt * 9u ? 1 : 0);
            const u8 * in = st->cm_in;
            if (st->lean) {
                u8 * stage = arena.take<u8>(stage_bytes(coded[k]));
                if (st->cm_in_size) HIP_CHECK(hipMemcpyAsync(stage, st->cm_in, st->cm_in_size, hipMemcpyDeviceToDevice, s));
                in = stage;
            }
            cm_jobs.push_back(CmDecodeJob{dev_addr(in), dev_addr(st->lean ? st->user : st->d_swap), st->cm_in_size, (u32)st->size_before_bwt, 0u, 0u});
        }
        CmDecodeJob * d_jobs = arena.take<CmDecodeJob>(cm_jobs.size());
        cm_ms += run_cm_jobs(lead->ctx, arena, cm_jobs, d_jobs, s, lead->ev0, lead->ev1,
                             [](const CmDecodeJob * j, u32 nj, hipStream_t st, int variant) { cm_decode_batch(j, nj, st, variant); }, &to_full);
        arena.release(mk);
    }
    // ---- phases 3-5 per tail window: inverse BWT per block, ONE LZP-decode launch (one workgroup per block), mRLE + CRC ----
    struct TailWindow {
        s32 w0 = 0, w1 = 0;
        std::vector<LzpDecodeJob> lz_jobs;  // host copies: alive until the window is finished
        std::vector<s32> lz_owner;
        std::vector<char> alive;
        LzpDecodeJob * d_lz = nullptr;
        u32 * luts = nullptr;
    } tw[DeviceCtx::AUX];
    for (int k = 0; k < tail_slots; k++) {
        tw[k].d_lz = lzp_in_window ? arena.take<LzpDecodeJob>(lzp_in_window) : nullptr;
        tw[k].luts = lzp_in_window ? arena.take<u32>(lzp_in_window * LZP_LUT_WORDS) : nullptr;
    }
    // keep-workspace mode: the swap buffers the lean states of the windows in flight borrow come out of the arena -- the staging area of the CM
    // rounds is free again by now -- as long as the per-block scratch of the stages (`need`) still fits behind them; the pool serves the rest
    std::vector<u8 *> arena_swaps;
    size_t swap_cap = 0;
    if (keep_workspace() && any_lean) {
        for (s32 i = 0; i < n; i++)
            if (sts[i]->lean && sts[i]->cap > swap_cap) swap_cap = sts[i]->cap;
        const size_t want = (size_t)tail_slots * (size_t)tail_window;
        const size_t step = (swap_cap + 255) & ~(size_t)255;
        const size_t keep_free = need + need / 32 + 65536;  // the per-block scratch of the stages and half of the slack arena_for adds
        while (swap_cap && arena_swaps.size() < want && arena.cap - arena.used >= keep_free + step) arena_swaps.push_back(arena.take<u8>(swap_cap));
    }
    std::vector<char> from_arena((size_t)n, 0);
    auto borrow = [&](s32 i) {
        bz3_state * st = sts[i];
        if (st->lean && !st->d_swap && !arena_swaps.empty() && st->cap <= swap_cap) {
            st->d_swap = arena_swaps.back();
            arena_swaps.pop_back();
            from_arena[(size_t)i] = 1;
            g_arena_swaps.fetch_add(1);
        } else {
            lean_borrow(st);
        }
    };
    auto give_back = [&](s32 i) {
        bz3_state * st = sts[i];
        if (from_arena[(size_t)i]) {
            if (st->d_swap) arena_swaps.push_back(st->d_swap);  // (a failure path may have dropped it already: lean_return of a buffer the pool does not know is a no-op)
            st->d_swap = nullptr;
            from_arena[(size_t)i] = 0;
        } else {
            lean_return(st);
        }
    };
    lead->ctx->ensure_aux();
    DrainOnUnwind drain{s, lead->ctx->aux, DeviceCtx::AUX};
    // The tail's whole-GPU kernels keep off the CUs the side streams' LZP decoders sit on (DeviceCtx::rest), when the device is partitioned
    hipStream_t s_cm = s;
    hipStream_t * side_streams = lead->ctx->aux;
    // Only in the regime it was measured in (profiles/r05_call{6,7,8}_*: 256-768 blocks): a small batch has at most 32 short LZP decoders, and its inverse
    // BWT, mRLE and CRC kernels would give up 19 % of the device and gain a synchronisation for them.  BZ3_HIP_CU_PARTITION_MIN_BLOCKS (read once; tests: 2).
    static const s32 part_min = [] { const char * e = getenv("BZ3_HIP_CU_PARTITION_MIN_BLOCKS"); return e && atoi(e) > 0 ? (s32)atoi(e) : (s32)128; }();
    if (lead->ctx->rest && n >= part_min && tail_slots <= DeviceCtx::RING_SLOTS) {
        side_streams = lead->ctx->aux_m;
        HIP_CHECK(hipStreamSynchronize(s));  // headers, stored blocks' CRCs and the CM launches ran on the group's stream
        s = lead->ctx->rest;
        for (s32 i = 0; i < n; i++) sts[i]->xs = s;
    }
    DrainOnUnwind drain_rest{s == s_cm ? nullptr : s, side_streams == lead->ctx->aux ? nullptr : side_streams, side_streams == lead->ctx->aux ? 0 : DeviceCtx::AUX};
    const s32 nwin = (n + tail_window - 1) / tail_window;
    const s32 lag = tail_slots - 1;  // window k is finished in iteration k + lag
    // BZ3_HIP_TRACE_RINGS=1 (diagnosis, read once): where this thread's wall time goes in the ring -- a line on stderr when the call ends
    static const bool trace_rings = getenv("BZ3_HIP_TRACE_RINGS") != nullptr;
    double tr_unbwt = 0, tr_wait = 0, tr_finish = 0, tr_t0 = now_ms();
    for (s32 k = 0; k < nwin + lag; k++) {
        const double tr_a = now_ms();
        if (k < nwin) {  // window k: inverse BWTs on the group's stream, then its LZP decoders on the slot's side stream
            const int q = (int)(k % tail_slots);
            hipStream_t s2 = side_streams[q];
            TailWindow & w = tw[q];
            w.w0 = k * tail_window;
            w.w1 = (w.w0 + tail_window < n) ? w.w0 + tail_window : n;
            w.lz_jobs.clear();
            w.lz_owner.clear();
            w.alive.assign((size_t)(w.w1 - w.w0), 0);
            for (s32 i = w.w0; i < w.w1; i++) {
                bz3_state * st = sts[i];
                if (st->pending == bz3_state::DEC_STORED) {  // :686-691
                    HIP_CHECK(hipStreamSynchronize(st->xs));
                    if (read_word(st->xs, st->d_words + 1) != st->crc) st->last_error = BZ3_ERR_CRC;
                    else st->result = st->size;  // last_error untouched (:691)
                    continue;
                }
                if (st->pending != bz3_state::DEC_CODED) continue;
                borrow(i);
                if (!decode_unbwt(st, arena, cm_ms)) continue;
                w.alive[(size_t)(i - w.w0)] = 1;
                if (st->model & 2) {
                    if (st->lzp_size < 4) {  // lzp_decompress: `if (n < 4) return -1` (:252) -> BZ3_ERR_CRC (:769-771)
                        st->last_error = BZ3_ERR_CRC;
                        w.alive[(size_t)(i - w.w0)] = 0;
                        continue;
                    }
                    // The reference decodes into its swap buffer (bz3_bound(block_size) bytes) and compares with buffer_size
                    // afterwards (:767-781).  A lean state decodes into the caller's buffer, so the cap is the smaller of the two;
                    // decode_finish tells the two failures apart.
                    const size_t bound = bz3_bound((size_t)st->block_size);
                    const size_t room = st->lean && st->buffer_size < bound ? st->buffer_size : bound;
                    w.lz_jobs.push_back(LzpDecodeJob{dev_addr(st->b1), dev_addr(st->b2), dev_addr(w.luts + w.lz_jobs.size() * LZP_LUT_WORDS), dev_addr(st->d_words + 5),
                                                     (u32)st->lzp_size, (u32)room});
                    w.lz_owner.push_back(i);
                }
            }
            if (!w.lz_jobs.empty()) {
                HIP_CHECK(hipEventRecord(lead->ctx->ev_prep, s));  // the inverse BWTs above are in flight on the group's stream
                HIP_CHECK(hipStreamWaitEvent(s2, lead->ctx->ev_prep, 0));
                HIP_CHECK(hipEventRecord(lead->ctx->ev_d0[q], s2));
                lzp_decode_batch(w.lz_jobs.data(), w.d_lz, (u32)w.lz_jobs.size(), s2);
                HIP_CHECK(hipEventRecord(lead->ctx->ev_d1[q], s2));
            }
        }
        const double tr_b = now_ms();
        tr_unbwt += tr_b - tr_a;
        if (k >= lag) {  // finish window k-lag: its LZP decoders have had the inverse BWTs of `lag` other windows to hide behind
            const int q = (int)((k - lag) % tail_slots);
            TailWindow & w = tw[q];
            if (!w.lz_jobs.empty()) {
                HIP_CHECK(hipEventSynchronize(lead->ctx->ev_d1[q]));
                tr_wait += now_ms() - tr_b;
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, lead->ctx->ev_d0[q], lead->ctx->ev_d1[q]);
                for (s32 i : w.lz_owner) sts[i]->t[BZ3_HIP_T_LZP] = ms;
                // lean state whose buffer is smaller than the reference's swap buffer: the decoder stops at the cap (:211) and
                // returns it, where the reference would have gone on to bz3_bound(block_size) and then either reported a larger
                // size (-> BZ3_ERR_DATA_SIZE_TOO_SMALL, :776) or run into malformed input (-> BZ3_ERR_CRC): when the cap was
                // reached, decode once more into a borrowed buffer of the reference's size and keep that verdict
                for (size_t j = 0; j < w.lz_jobs.size(); j++) {
                    bz3_state * st = sts[w.lz_owner[j]];
                    const size_t bound = bz3_bound((size_t)st->block_size);
                    if (!st->lean || st->buffer_size >= bound || read_word(s, st->d_words + 5) != w.lz_jobs[j].max_out) continue;
                    u8 * big = st->ctx->temp_get(st->cap);
                    LzpDecodeJob again = w.lz_jobs[j];
                    again.out = dev_addr(big);
                    again.max_out = (u32)bound;
                    again.lut = dev_addr(w.luts);
                    lzp_decode_batch(&again, w.d_lz, 1u, s);
                    HIP_CHECK(hipStreamSynchronize(s));
                    st->ctx->temp_put(big);
                }
            }
            for (s32 i = w.w0; i < w.w1; i++) {
                if (w.alive[(size_t)(i - w.w0)]) decode_finish(sts[i], arena);
                give_back(i);
            }
        }
        tr_finish += now_ms() - tr_b;
    }
    if (trace_rings)
        fprintf(stderr, "[bz3 rings] decode tail: %d blocks, %d windows of %d x %d slots: %.1f ms = inverse BWTs + LZP launches %.1f + waiting for a window's LZP decoders %.1f + mRLE / CRC / hand-back %.1f\n",
                (int)n, (int)nwin, (int)tail_window, (int)tail_slots, now_ms() - tr_t0, tr_unbwt, tr_wait, tr_finish - tr_wait);
    for (s32 i = 0; i < n; i++) sts[i]->pending = bz3_state::NONE;
    if (s != s_cm) (void)hipStreamSynchronize(s);  // (the tail ran on the masked stream; decode_finish has waited for every block already)
    enforce_headroom(lead->ctx, s_cm);
}

void on_failure(bz3_state * st) {
    if (st) {
        lean_return(st);
        st->last_error = BZ3_ERR_BWT;
        st->pending = bz3_state::NONE;
        st->result = -1;
    }
}

// Splits a batch by device (states may live on different GPUs) and runs the groups CONCURRENTLY, one host thread per
// GPU (the reference forks a thread per block, src/libbz3.c:845-856): every group has its own device context, stream,
// arena and mutex, and hipSetDevice is per host thread, so the GPUs of a node work on their blocks at the same time.
std::atomic<int> g_groups_running{0}, g_groups_peak{0};  // statistics (bz3_hip_debug_peak_concurrent_groups)

template <typename F>
void for_each_device_group(bz3_state ** states, s32 n, F && f) {
    std::vector<std::vector<s32>> groups;
    std::vector<char> done((size_t)n, 0);
    for (s32 i = 0; i < n; i++) {
        if (done[(size_t)i]) continue;
        groups.emplace_back();
        for (s32 j = i; j < n; j++)
            if (!done[(size_t)j] && states[j]->device == states[i]->device) { groups.back().push_back(j); done[(size_t)j] = 1; }
    }
    auto run = [&](const std::vector<s32> & idx) {
        const int now = g_groups_running.fetch_add(1) + 1;
        int peak = g_groups_peak.load();
        while (now > peak && !g_groups_peak.compare_exchange_weak(peak, now)) {}
        try {
            f(idx);
        } catch (const HipError & e) {
            fprintf(stderr, "bzip3_amd: HIP failure '%s' at %s:%d\n", e.what, e.file, e.line);
            for (s32 j : idx) on_failure(states[j]);
        } catch (...) {  // bad_alloc, system_error, length_error ...: nothing may leave a worker thread (std::terminate)
            for (s32 j : idx) on_failure(states[j]);
        }
        g_groups_running.fetch_sub(1);
    };
    std::vector<std::thread> workers;
    struct Joiner {  // whatever happens on the calling thread, the workers are joined before their std::thread objects die
        std::vector<std::thread> & w;
        ~Joiner() {
            for (std::thread & t : w)
                if (t.joinable()) t.join();
        }
    } joiner{workers};
    size_t spawned = 1;  // groups [1, spawned) run on a worker thread each, group 0 and the rest on this thread
    for (size_t g = 1; g < groups.size(); g++) {
        try {
            workers.emplace_back(run, std::cref(groups[g]));
        } catch (const std::system_error &) {  // no thread to be had
            break;
        }
        spawned = g + 1;
    }
    if (!groups.empty()) run(groups[0]);
    for (size_t g = spawned; g < groups.size(); g++) run(groups[g]);
}

// Host-buffer API: the blocks of a group are staged through the states' d_io buffers inside the group's own thread, all
// copies of a direction enqueued back to back on the group's stream with ONE synchronisation, so the GPUs of a batch copy
// and code at the same time.  A block whose staging fails is marked (skip) and left alone by the group; the others go on.
void stage_in(bz3_state * st, const void * host, size_t bytes, hipStream_t s) {
    try {
        ensure_io(st);
        if (bytes) HIP_CHECK(hipMemcpyAsync(st->d_io, host, bytes, hipMemcpyHostToDevice, s));
    } catch (...) {  // HIP failure or bad_alloc from ensure_io: this block fails, the others of the group go on
        on_failure(st);
        st->skip = true;
    }
}

void run_encode(bz3_state ** states, void ** buffers, s32 * sizes, s32 n, bool host) {
    for_each_device_group(states, n, [&](const std::vector<s32> & idx) {
        std::vector<bz3_state *> sts;
        std::vector<u8 *> bufs;
        std::vector<s32> szs;
        DeviceGuard g(states[idx[0]]->device);
        hipStream_t s = states[idx[0]]->stream;
        const double t0 = now_ms();
        for (s32 j : idx) {
            bz3_state * st = states[j];
            st->skip = false;
            if (host) stage_in(st, buffers[j], (sizes[j] >= 0 && sizes[j] <= st->block_size) ? (size_t)sizes[j] : 0, s);
            sts.push_back(st);
            bufs.push_back(host ? st->d_io : (u8 *)buffers[j]);
            szs.push_back(sizes[j]);
        }
        if (host) HIP_CHECK(hipStreamSynchronize(s));
        const float in_ms = (float)(now_ms() - t0);
        encode_group(sts.data(), bufs.data(), szs.data(), (s32)idx.size());
        if (!host) return;
        const double t1 = now_ms();
        for (s32 j : idx) {
            bz3_state * st = states[j];
            if (st->result > 0 && !st->skip) HIP_CHECK(hipMemcpyAsync(buffers[j], st->d_io, (size_t)st->result, hipMemcpyDeviceToHost, s));
        }
        HIP_CHECK(hipStreamSynchronize(s));
        const float out_ms = (float)(now_ms() - t1);
        for (s32 j : idx) states[j]->t[BZ3_HIP_T_COPY] += (in_ms + out_ms) / (float)idx.size();
    });
    for (s32 i = 0; i < n; i++) {
        sizes[i] = states[i]->result;
        states[i]->skip = false;
    }
}

void run_decode(bz3_state ** states, void ** buffers, const size_t * buffer_sizes, const s32 * sizes, const s32 * orig_sizes, const u8 * hdrs, s32 n, bool host) {
    for_each_device_group(states, n, [&](const std::vector<s32> & idx) {
        std::vector<bz3_state *> sts;
        std::vector<u8 *> bufs;
        std::vector<size_t> bsz;
        std::vector<s32> csz, osz;
        std::vector<u8> hd;
        DeviceGuard g(states[idx[0]]->device);
        hipStream_t s = states[idx[0]]->stream;
        const double t0 = now_ms();
        for (s32 j : idx) {
            bz3_state * st = states[j];
            st->skip = false;
            if (host) {
                // the two checks that protect the H2D copy are the reference's first two (:658, :667); decode_front repeats them
                const bool copy_ok = buffer_sizes[j] >= 9 && buffer_sizes[j] >= (size_t)sizes[j] && sizes[j] >= 0 &&
                                     (size_t)sizes[j] <= bz3_bound((size_t)st->block_size);
                stage_in(st, buffers[j], copy_ok ? (size_t)sizes[j] : 0, s);
            }
            sts.push_back(st); bufs.push_back(host ? st->d_io : (u8 *)buffers[j]); bsz.push_back(buffer_sizes[j]); csz.push_back(sizes[j]); osz.push_back(orig_sizes[j]);
            hd.insert(hd.end(), hdrs + 17 * (size_t)j, hdrs + 17 * (size_t)j + 17);
        }
        if (host) HIP_CHECK(hipStreamSynchronize(s));
        const float in_ms = (float)(now_ms() - t0);
        decode_group(sts.data(), bufs.data(), bsz.data(), csz.data(), osz.data(), hd.data(), (s32)idx.size());
        if (!host) return;
        const double t1 = now_ms();
        for (s32 j : idx) {
            bz3_state * st = states[j];
            if (st->result > 0 && !st->skip) HIP_CHECK(hipMemcpyAsync(buffers[j], st->d_io, (size_t)st->result, hipMemcpyDeviceToHost, s));
        }
        HIP_CHECK(hipStreamSynchronize(s));
        const float out_ms = (float)(now_ms() - t1);
        for (s32 j : idx) states[j]->t[BZ3_HIP_T_COPY] += (in_ms + out_ms) / (float)idx.size();
    });
    for (s32 i = 0; i < n; i++) states[i]->skip = false;
}

// Host copies of the first 17 bytes of n device-resident blocks.
void fetch_headers(bz3_state ** states, void ** buffers, const size_t * buffer_sizes, s32 n, std::vector<u8> & hdrs) {
    hdrs.assign(17 * (size_t)n, 0);
    for (s32 i = 0; i < n; i++) {
        const size_t take = buffer_sizes[i] < 17 ? buffer_sizes[i] : 17;
        if (take < 9) continue;
        try {
            HIP_CHECK(hipSetDevice(states[i]->device));
            HIP_CHECK(hipMemcpyAsync(hdrs.data() + 17 * (size_t)i, buffers[i], take, hipMemcpyDeviceToHost, states[i]->stream));
            HIP_CHECK(hipStreamSynchronize(states[i]->stream));
        } catch (const HipError &) {
        }
    }
}

}  // namespace

// =====================================================================================================
// libbz3.h
// =====================================================================================================
extern "C" {

BZIP3_API const char * bz3_version(void) { return "1.5.2-mi355x"; }

BZIP3_API int8_t bz3_last_error(struct bz3_state * state) { return state->last_error; }

BZIP3_API size_t bz3_bound(size_t input_size) { return input_size + input_size / 50 + 32; }

BZIP3_API const char * bz3_strerror(struct bz3_state * state) {  // messages: src/libbz3.c:512-533
    switch (state->last_error) {
        case BZ3_OK: return "No error";
        case BZ3_ERR_OUT_OF_BOUNDS: return "Data index out of bounds";
        case BZ3_ERR_BWT: return "Burrows-Wheeler transform failed";
        case BZ3_ERR_CRC: return "CRC32 check failed";
        case BZ3_ERR_MALFORMED_HEADER: return "Malformed header";
        case BZ3_ERR_TRUNCATED_DATA: return "Truncated data";
        case BZ3_ERR_DATA_TOO_BIG: return "Too much data";
        case BZ3_ERR_DATA_SIZE_TOO_SMALL:
            return "Size of buffer `buffer_size` passed to the block decoder (bz3_decode_block) is too small. See function docs for details.";
        default: return "Unknown error";
    }
}

BZIP3_API struct bz3_state * bz3_new(int32_t block_size) {
    if (block_size < KiB65 || block_size > MiB511) return nullptr;  // :536
    bz3_state * st = nullptr;
    try {
        const int dev = pick_device();
        if (dev < 0) {
            fprintf(stderr, "bzip3_amd: no HIP device available -- this library has no CPU code path\n");
            return nullptr;
        }
        DeviceCtx * ctx = get_ctx(dev);
        if (!ctx) return nullptr;
        st = new bz3_state;
        st->block_size = block_size;
        st->device = dev;
        st->ctx = ctx;
        HIP_CHECK(hipSetDevice(dev));
        HIP_CHECK(hipStreamCreateWithFlags(&st->stream, hipStreamNonBlocking));
        st->xs = st->stream;
        HIP_CHECK(hipEventCreate(&st->ev0));
        HIP_CHECK(hipEventCreate(&st->ev1));
        st->cap = (bz3_bound((size_t)block_size) + 4096 + 255) & ~(size_t)255;
        st->lean = lean_states();
        if (!st->lean) HIP_CHECK(hipMalloc((void **)&st->d_swap, st->cap));
        HIP_CHECK(hipMalloc((void **)&st->d_words, 64 * sizeof(u32)));
        st->last_error = BZ3_OK;
        return st;
    } catch (const HipError & e) {
        fprintf(stderr, "bzip3_amd: bz3_new failed: %s (%s:%d)\n", e.what, e.file, e.line);
        state_release(st);
        return nullptr;
    } catch (const std::bad_alloc &) {
        state_release(st);
        return nullptr;
    }
}

BZIP3_API void bz3_free(struct bz3_state * state) { state_release(state); }

BZIP3_API size_t bz3_min_memory_needed(int32_t block_size) {
    if (block_size < KiB65 || block_size > MiB511) return 0;
    const size_t cap = (bz3_bound((size_t)block_size) + 4096 + 255) & ~(size_t)255;
    return sizeof(bz3_state) + cap + 64 * sizeof(u32);
}

// ---- device-resident entry points (bz3_hip.h) -----------------------------------------------------
BZIP3_API void bz3_hip_encode_blocks_device(struct bz3_state * states[], void * buffers[], int32_t sizes[], int32_t n) {
    run_encode(states, buffers, sizes, n, false);
}

BZIP3_API void bz3_hip_decode_blocks_device(struct bz3_state * states[], void * buffers[], size_t buffer_sizes[], int32_t sizes[],
                                            int32_t orig_sizes[], int32_t n) {
    std::vector<u8> hdrs;
    fetch_headers(states, buffers, buffer_sizes, n, hdrs);
    run_decode(states, buffers, buffer_sizes, sizes, orig_sizes, hdrs.data(), n, false);
}

BZIP3_API int32_t bz3_hip_encode_block_device(struct bz3_state * st, void * buffer, int32_t size) {
    s32 sz = size;
    run_encode(&st, &buffer, &sz, 1, false);
    return sz;
}

BZIP3_API int32_t bz3_hip_decode_block_device(struct bz3_state * st, void * buffer, size_t buffer_size, int32_t compressed_size,
                                              int32_t orig_size) {
    bz3_hip_decode_blocks_device(&st, &buffer, &buffer_size, &compressed_size, &orig_size, 1);
    return st->result;
}

// ---- host-buffer entry points (libbz3.h): staged through the states' d_io buffers by the device groups (run_encode / run_decode) ------
BZIP3_API void bz3_encode_blocks(struct bz3_state * states[], uint8_t * buffers[], int32_t sizes[], int32_t n) {
    run_encode(states, (void **)buffers, sizes, n, true);
}

BZIP3_API void bz3_decode_blocks(struct bz3_state * states[], uint8_t * buffers[], size_t buffer_sizes[], int32_t sizes[], int32_t orig_sizes[],
                                 int32_t n) {
    std::vector<u8> hdrs(17 * (size_t)n, 0);
    for (s32 i = 0; i < n; i++) memcpy(hdrs.data() + 17 * (size_t)i, buffers[i], buffer_sizes[i] < 17 ? buffer_sizes[i] : 17);
    run_decode(states, (void **)buffers, buffer_sizes, sizes, orig_sizes, hdrs.data(), n, true);
}

// ---- single-block calls from several host threads ----------------------------------------------------------------------------
// The reference's batch API IS "N threads, one bz3_encode_block each" (src/libbz3.c:831-856), and a binding that does its own
// threading calls the single-block functions the same way.  Here a block's CM stage is a serial ~300 ns-per-byte recurrence that only
// pays off with many blocks in ONE launch, and a group call holds its device for its whole duration: N concurrent single-block calls
// would run as N launches one after the other.  So concurrent callers are COLLECTED: the first one in becomes the leader, waits a
// short window (bz3_hip_set_collect_window_us, default 200 us -- nothing beside a block's milliseconds) for the others, and takes
// everything that has arrived through bz3_encode_blocks / bz3_decode_blocks as one batch; callers that arrive while a batch runs form
// the next one.  Same bytes, return values and error codes: a batch call treats its blocks independently.
namespace {
struct SingleReq {
    bz3_state * st;
    u8 * buffer;
    size_t buffer_size;
    s32 size, orig_size;
    bool done = false;
};
struct Collector {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<SingleReq *> pending[2];  // [0] encode, [1] decode
    bool leader[2] = {false, false};
    size_t last_size[2] = {0, 0};  // callers in the previous batch: the leader stops waiting as soon as that many have arrived
    int lonely[2] = {0, 0};        // batches of ONE caller in a row with nobody queued behind them
};
Collector g_collect;
std::atomic<int> g_collect_window_us{200};
std::atomic<unsigned> g_collect_batches{0}, g_collect_largest{0};

void run_collected(int kind, std::vector<SingleReq *> & batch) {
    const s32 n = (s32)batch.size();
    g_collect_batches.fetch_add(1u);
    unsigned big = g_collect_largest.load();
    while ((unsigned)n > big && !g_collect_largest.compare_exchange_weak(big, (unsigned)n)) {}
    std::vector<bz3_state *> sts((size_t)n);
    std::vector<u8 *> bufs((size_t)n);
    std::vector<s32> sizes((size_t)n), orig((size_t)n);
    std::vector<size_t> bsz((size_t)n);
    for (s32 i = 0; i < n; i++) {
        sts[(size_t)i] = batch[(size_t)i]->st;
        bufs[(size_t)i] = batch[(size_t)i]->buffer;
        sizes[(size_t)i] = batch[(size_t)i]->size;
        orig[(size_t)i] = batch[(size_t)i]->orig_size;
        bsz[(size_t)i] = batch[(size_t)i]->buffer_size;
    }
    if (kind == 0) {
        bz3_encode_blocks(sts.data(), bufs.data(), sizes.data(), n);
        for (s32 i = 0; i < n; i++) batch[(size_t)i]->size = sizes[(size_t)i];
    } else {
        bz3_decode_blocks(sts.data(), bufs.data(), bsz.data(), sizes.data(), orig.data(), n);
    }
}

// The window is there for the FIRST batch of a group of threads; after it the callers that arrive while a batch runs form the next one
// by themselves.  So (round 4): a caller that has been alone twice in a row -- a single-threaded program -- no longer
// waits at all (it paid the window on every call, in both directions), and a leader stops waiting as soon as as many callers as
// the previous batch had have arrived instead of sleeping the whole window.  What remains by design: the callers of one batch share
// its latency -- a 64 KiB block that is collected with a 511 MiB block returns when that block does (bz3_hip_set_collect_window_us(0)
// gives every call a batch of its own).
void collect(int kind, SingleReq & r) {
    Collector & c = g_collect;
    std::unique_lock<std::mutex> lk(c.mu);
    c.pending[kind].push_back(&r);
    c.cv.notify_all();  // a leader inside its window counts the arrivals
    while (!r.done) {
        if (c.leader[kind]) {
            c.cv.wait(lk);
            continue;
        }
        c.leader[kind] = true;
        const int win = g_collect_window_us.load();
        if (win > 0 && c.lonely[kind] < 2) {
            const size_t want = c.last_size[kind] > 1 ? c.last_size[kind] : ~(size_t)0;
            c.cv.wait_for(lk, std::chrono::microseconds(win), [&] { return c.pending[kind].size() >= want; });  // the others arrive meanwhile (the lock is released)
        }
        std::vector<SingleReq *> batch;
        batch.swap(c.pending[kind]);
        lk.unlock();
        try {
            run_collected(kind, batch);
        } catch (...) {  // (the batch calls do not throw; belt and braces: nobody may wait for ever)
            for (SingleReq * q : batch) {
                on_failure(q->st);
                if (kind == 0) q->size = -1;  // bz3_encode_block returns this: the request's input size must not pass for a coded size
            }
        }
        lk.lock();
        for (SingleReq * q : batch) q->done = true;
        c.last_size[kind] = batch.size();
        c.lonely[kind] = (batch.size() == 1 && c.pending[kind].empty()) ? c.lonely[kind] + 1 : 0;
        c.leader[kind] = false;
        c.cv.notify_all();
    }
}
}  // namespace

BZIP3_API unsigned bz3_hip_cm_blocks_routed_full(void) { return g_cm_routed_full.load(); }
BZIP3_API void bz3_hip_set_workspace_headroom(long long bytes) { g_ws_headroom.store(bytes < 0 ? -1 : bytes); }
BZIP3_API size_t bz3_hip_workspace_headroom(void) { return ws_headroom(); }
BZIP3_API unsigned bz3_hip_debug_headroom_events(int reset, unsigned * releases) {
    const unsigned t = reset ? g_headroom_trims.exchange(0) : g_headroom_trims.load();
    const unsigned r = reset ? g_headroom_releases.exchange(0) : g_headroom_releases.load();
    if (releases) *releases = r;
    return t;
}
BZIP3_API size_t bz3_hip_debug_ring_contexts(size_t free_b, size_t have, size_t need, size_t fixed, size_t ctx_bytes, size_t cap, int lean, size_t headroom) {
    return ring_contexts_for(free_b, have, need, fixed, ctx_bytes, cap, lean != 0, headroom);
}
BZIP3_API size_t bz3_hip_debug_workspace_bytes(size_t block_bytes, int which) {  // 0: per-block scratch of the stages, 1: one LZP context of the encoder's ring
    return which == 0 ? workspace_bytes_for((u64)block_bytes + 64) : lzp_encode_ctx_bytes((u64)block_bytes + 64) + 65536;
}
BZIP3_API unsigned bz3_hip_debug_cm_launches(int reset) { return reset ? g_cm_launches.exchange(0) : g_cm_launches.load(); }
BZIP3_API size_t bz3_hip_debug_arena_slack(size_t bytes) { return DeviceCtx::arena_slack(bytes); }
BZIP3_API size_t bz3_hip_debug_cached_bytes(int device) {  // workspace + idle pooled swap buffers the library holds on `device` right now
    DeviceCtx * c = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (device >= 0 && (size_t)device < g_ctx.size()) c = g_ctx[(size_t)device];
    }
    if (!c) return 0;
    std::lock_guard<std::mutex> lk(c->mu);
    return c->ws_cap + c->temp_idle_bytes();
}
BZIP3_API int bz3_hip_set_front_end_duo(int on) {
    g_front_duo.store(on < 0 ? -1 : (on ? 1 : 0));
    return 0;
}
BZIP3_API int bz3_hip_set_keep_workspace(int on) {
    g_keep_ws.store(on < 0 ? -1 : (on ? 1 : 0));
    return 0;
}
BZIP3_API void bz3_hip_set_collect_window_us(int us) { g_collect_window_us.store(us < 0 ? 200 : us); }
BZIP3_API unsigned bz3_hip_debug_collected_batches(int reset, unsigned * largest) {  // batches run for single-block callers; *largest = blocks in the largest
    if (largest) *largest = g_collect_largest.load();
    const unsigned b = g_collect_batches.load();
    if (reset) {
        g_collect_batches.store(0);
        g_collect_largest.store(0);
    }
    return b;
}

BZIP3_API int32_t bz3_encode_block(struct bz3_state * st, uint8_t * buffer, int32_t size) {
    SingleReq r{st, buffer, 0, size, 0};
    try {
        collect(0, r);
    } catch (...) {  // (queueing the request could not allocate: nothing may cross the C boundary)
        on_failure(st);
        return -1;
    }
    return r.size;
}

BZIP3_API int32_t bz3_decode_block(struct bz3_state * st, uint8_t * buffer, size_t buffer_size, int32_t compressed_size, int32_t orig_size) {
    SingleReq r{st, buffer, buffer_size, compressed_size, orig_size};
    try {
        collect(1, r);
    } catch (...) {
        on_failure(st);
        return -1;
    }
    return st->result;
}

// ---- frame API (src/libbz3.c:876-997): same header, chunk layout and error codes.  The reference walks the
// blocks of a frame one after the other on one state; here a WINDOW of blocks goes through bz3_encode_blocks /
// bz3_decode_blocks at once (one state per block, blocks spread over the visible GPUs, one CM launch per GPU), because
// the serial CM stage only pays off with many blocks in flight.  Blocks are still committed in order, so the first
// failing block decides the return value exactly as in the sequential loop.
namespace {

struct FrameWindow {
    std::vector<bz3_state *> states;
    std::vector<u8 *> bufs;
    size_t cap = 0;
    ~FrameWindow() {
        for (bz3_state * s : states) bz3_free(s);
        for (u8 * b : bufs) free(b);
    }
    // Up to `want` states + host buffers of bz3_bound(block_size) bytes; at least one or false.
    bool init(u32 block_size, size_t want) {
        cap = bz3_bound(block_size);
        const int ndev = device_count();
        size_t limit = (size_t)(ndev > 0 ? ndev : 1) * 256;             // one CU per block during the CM stage
        const size_t by_mem = ((size_t)16 << 30) / (cap ? cap : 1);      // host staging budget of the window
        if (limit > by_mem) limit = by_mem;
        if (limit < 1) limit = 1;
        if (want > limit) want = limit;
        if (want < 1) want = 1;
        for (size_t i = 0; i < want; i++) {
            bz3_state * st = bz3_new((int32_t)block_size);
            u8 * b = st ? (u8 *)malloc(cap) : nullptr;
            if (!st || !b) {
                if (st) bz3_free(st);
                free(b);
                break;
            }
            states.push_back(st);
            bufs.push_back(b);
        }
        return !states.empty();
    }
};

}  // namespace

BZIP3_API int bz3_compress(uint32_t block_size, const uint8_t * in, uint8_t * out, size_t in_size, size_t * out_size) {
    if (block_size > in_size) block_size = (uint32_t)bz3_bound(in_size);  // :877
    block_size = block_size <= (uint32_t)KiB65 ? (uint32_t)KiB65 : block_size;
    u32 n_blocks = (u32)(in_size / block_size);
    if (in_size % block_size) n_blocks++;
    FrameWindow w;
    if (!w.init(block_size, n_blocks)) return BZ3_ERR_INIT;  // :879-886 (allocation precedes the size check)
    const size_t buf_max = *out_size;
    *out_size = 0;
    if (buf_max < 13 || buf_max < bz3_bound(in_size)) return BZ3_ERR_DATA_TOO_BIG;
    memcpy(out, "BZ3v1", 5);
    wr_le32(out + 5, block_size);
    wr_le32(out + 9, n_blocks);
    *out_size += 13;
    size_t in_off = 0;
    const u32 W = (u32)w.states.size();
    std::vector<s32> sizes(W), orig(W);
    for (u32 i0 = 0; i0 < n_blocks; i0 += W) {
        const u32 cnt = n_blocks - i0 < W ? n_blocks - i0 : W;
        for (u32 k = 0; k < cnt; k++) {
            s32 size = (s32)block_size;
            if (i0 + k == n_blocks - 1) size = (s32)(in_size % block_size);  // (sic) :914 -- 0 when in_size is a multiple
            memcpy(w.bufs[k], in + in_off, (size_t)size);
            sizes[k] = orig[k] = size;
            in_off += (size_t)size;
        }
        bz3_encode_blocks(w.states.data(), w.bufs.data(), sizes.data(), (int32_t)cnt);
        for (u32 k = 0; k < cnt; k++) {
            if (bz3_last_error(w.states[k]) != BZ3_OK) return w.states[k]->last_error;  // :917-922
            const s32 osz = sizes[k];
            memcpy(out + *out_size + 8, w.bufs[k], (size_t)osz);
            wr_le32(out + *out_size, (u32)osz);
            wr_le32(out + *out_size + 4, (u32)orig[k]);
            *out_size += (size_t)osz + 8;
        }
    }
    return BZ3_OK;
}

BZIP3_API int bz3_decompress(const uint8_t * in, uint8_t * out, size_t in_size, size_t * out_size) {
    if (in_size < 13) return BZ3_ERR_MALFORMED_HEADER;
    if (memcmp(in, "BZ3v1", 5) != 0) return BZ3_ERR_MALFORMED_HEADER;
    const u32 block_size = rd_le32(in + 5);
    const u32 n_blocks = rd_le32(in + 9);
    in_size -= 13;
    in += 13;
    FrameWindow w;
    {
        // The header fields are untrusted: size the window from the chunks that are actually PRESENT -- walk the chunk
        // headers with the checks of the loop below and count the chunks that lie inside the input -- never from
        // n_blocks alone (a few hundred bytes claiming thousands of 511 MiB blocks must not allocate a window of states).
        size_t present = 0, off = 0;
        while (present < n_blocks && in_size - off >= 8) {
            const s32 size = (s32)rd_le32(in + off);
            if (size < 0 || (u32)size > block_size || in_size - off < (size_t)size + 8) break;
            off += (size_t)size + 8;
            present++;
        }
        if (!w.init(block_size, present ? present : 1)) return BZ3_ERR_INIT;  // :953-960
    }
    const size_t buf_max = *out_size;
    *out_size = 0;
    const u32 W = (u32)w.states.size();
    std::vector<s32> sizes(W), orig(W);
    std::vector<size_t> caps(W, w.cap);
    u32 i = 0;
    while (i < n_blocks) {
        // Collect chunks until the window is full or a chunk header is bad.  The reference would have decoded the
        // chunks before the bad one first, so they run (and may fail) before the header error is reported.
        u32 cnt = 0;
        int header_error = BZ3_OK;
        size_t planned = *out_size;
        while (cnt < W && i + cnt < n_blocks) {
            if (in_size < 8) { header_error = BZ3_ERR_MALFORMED_HEADER; break; }                 // :963
            const s32 size = (s32)rd_le32(in);
            if (size < 0 || (u32)size > block_size) { header_error = BZ3_ERR_MALFORMED_HEADER; break; }  // :969
            if (in_size < (size_t)size + 8) { header_error = BZ3_ERR_TRUNCATED_DATA; break; }    // :974
            const s32 orig_size = (s32)rd_le32(in + 4);
            if (orig_size < 0) { header_error = BZ3_ERR_MALFORMED_HEADER; break; }               // :980
            if (buf_max < planned + (size_t)orig_size) { header_error = BZ3_ERR_DATA_TOO_BIG; break; }  // :985
            memcpy(w.bufs[cnt], in + 8, (size_t)size);
            sizes[cnt] = size;
            orig[cnt] = orig_size;
            planned += (size_t)orig_size;
            in += size + 8;
            in_size -= (size_t)size + 8;
            cnt++;
        }
        if (cnt) bz3_decode_blocks(w.states.data(), w.bufs.data(), caps.data(), sizes.data(), orig.data(), (int32_t)cnt);
        for (u32 k = 0; k < cnt; k++) {
            if (bz3_last_error(w.states[k]) != BZ3_OK) return w.states[k]->last_error;  // :989-993
            memcpy(out + *out_size, w.bufs[k], (size_t)orig[k]);
            *out_size += (size_t)orig[k];
        }
        if (header_error != BZ3_OK) return header_error;
        i += cnt;
    }
    return BZ3_OK;
}

BZIP3_API int bz3_orig_size_sufficient_for_decode(const uint8_t * block, size_t block_size, int32_t orig_size) {  // :1025-1055
    if (block_size < 9) return -1;
    const s32 bwt_idx = (s32)rd_le32(block + 4);
    if (bwt_idx == -1) return 1;
    const s32 model = (s8)block[8];
    const size_t need = 9 + (size_t)((model & 2) * 4) + (size_t)((model & 4) * 4);
    if (block_size < need) return -1;
    s32 lzp_size = -1, rle_size = -1;
    size_t off = 9;
    if (model & 2) { lzp_size = (s32)rd_le32(block + off); off += 4; }
    if (model & 4) rle_size = (s32)rd_le32(block + off);
    return sizes_fit((size_t)orig_size, lzp_size, rle_size, orig_size) ? 1 : 0;
}

// ---- bz3_hip.h: device control, timings ----------------------------------------------------------------
BZIP3_API int bz3_hip_device_count(void) { return device_count(); }

BZIP3_API int bz3_hip_bind_device(int device) {
    if (device < -1 || device >= device_count()) return -1;
    g_bound_device.store(device);
    return 0;
}

BZIP3_API int bz3_hip_state_device(struct bz3_state * st) { return st->device; }

BZIP3_API int bz3_hip_set_cm_mode(int mode) {
    bool ok = mode >= -1 && mode <= CM_VARIANT_ROWS3;
#ifdef BZ3_EMU
    ok = ok || mode == CM_VARIANT_ROWS_TEST;
#endif
    if (!ok) return -1;
    g_cm_mode.store(mode);
    return 0;
}

BZIP3_API unsigned bz3_hip_cm_blocks_given_up(void) { return g_cm_given_up.load(); }

BZIP3_API void bz3_hip_debug_bwt_big_rounds(int k) { bwt_set_big_rounds(k); }

BZIP3_API int bz3_hip_cm_variant_for(int device, int blocks, int encode) {
    DeviceCtx * c = get_ctx(device);
    return (c && blocks > 0) ? cm_variant_for(c, (size_t)blocks, encode != 0) : -1;
}

BZIP3_API int bz3_hip_debug_front_end_ring(void) { return g_front_end_ring.load(); }
BZIP3_API int bz3_hip_debug_arena_swap_buffers(int reset) { return reset ? g_arena_swaps.exchange(0) : g_arena_swaps.load(); }

BZIP3_API int bz3_hip_debug_peak_concurrent_groups(int reset) {
    const int v = g_groups_peak.load();
    if (reset) g_groups_peak.store(0);
    return v;
}

BZIP3_API int bz3_hip_set_lean_states(int on) {
    g_lean.store(on ? 1 : 0);
    return 0;
}

BZIP3_API void bz3_hip_release_cached_memory(void) {
    const int n = device_count();
    for (int d = 0; d < n; d++) {
        DeviceCtx * c = nullptr;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            c = g_ctx[(size_t)d];
        }
        if (!c) continue;
        (void)hipSetDevice(d);
        std::lock_guard<std::mutex> lk(c->mu);
        c->temp_trim();
        if (c->ws) (void)hipFree(c->ws);
        c->ws = nullptr;
        c->ws_cap = 0;
    }
}

BZIP3_API void bz3_hip_last_timings(struct bz3_state * st, float ms[BZ3_HIP_T_COUNT]) {
    for (int i = 0; i < BZ3_HIP_T_COUNT; i++) ms[i] = st->t[i];
}

BZIP3_API void bz3_hip_last_bwt_stats(struct bz3_state * st, int32_t * rounds, int32_t * radix_passes, uint64_t * sorted_elements) {
    if (rounds) *rounds = st->bwt.rounds;
    if (radix_passes) *radix_passes = st->bwt.radix_passes;
    if (sorted_elements) *sorted_elements = st->bwt.sorted_elements;
}

}  // extern "C"

// =====================================================================================================
// stage hooks on host buffers (tests / profiling)
// =====================================================================================================
namespace {

struct StageEnv {
    DeviceCtx * ctx = nullptr;
    hipStream_t s = nullptr;
    std::vector<void *> allocs;
    std::unique_lock<std::mutex> lock;
    StageEnv() {
        int dev = pick_device();
        if (dev < 0) {
            fprintf(stderr, "bzip3_amd: no HIP device available -- this library has no CPU code path\n");
            abort();
        }
        ctx = get_ctx(dev);
        HIP_CHECK(hipSetDevice(dev));
        HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        lock = std::unique_lock<std::mutex>(ctx->mu);
    }
    ~StageEnv() {
        (void)hipStreamSynchronize(s);
        for (void * p : allocs) (void)hipFree(p);
        (void)hipStreamDestroy(s);
    }
    u8 * dev(size_t bytes, const void * init = nullptr, size_t init_bytes = 0) {
        void * p = nullptr;
        HIP_CHECK(hipMalloc(&p, bytes + 4096));
        allocs.push_back(p);
        if (init && init_bytes) HIP_CHECK(hipMemcpy(p, init, init_bytes, hipMemcpyHostToDevice));
        return (u8 *)p;
    }
    void down(void * host, const void * d, size_t bytes) {
        HIP_CHECK(hipStreamSynchronize(s));
        if (bytes) HIP_CHECK(hipMemcpy(host, d, bytes, hipMemcpyDeviceToHost));
    }
    u32 word(const u32 * d) {
        u32 v = 0;
        down(&v, d, 4);
        return v;
    }
};

template <typename F>
auto stage_guard(F && f) -> decltype(f()) {
    try {
        return f();
    } catch (const HipError & e) {
        fprintf(stderr, "bzip3_amd: HIP failure '%s' at %s:%d\n", e.what, e.file, e.line);
        abort();
    }
}

}  // namespace

extern "C" {

BZIP3_API uint32_t bz3_hip_stage_crc32c(const uint8_t * data, size_t n, uint32_t init) {
    return stage_guard([&]() -> u32 {
        StageEnv e;
        u8 * d = e.dev(n + 16, data, n);
        u32 * w = (u32 *)e.dev(64);
        crc32c_device(d, n, init, e.ctx->d_crc, w, e.s);
        return e.word(w + 1);
    });
}

BZIP3_API int32_t bz3_hip_stage_mrle_encode(const uint8_t * in, int32_t n, uint8_t * out) {
    return stage_guard([&]() -> s32 {
        StageEnv e;
        u8 * d = e.dev((size_t)n + 16, in, (size_t)n);
        u8 * o = e.dev((size_t)n + 64);
        Arena a = e.ctx->arena_for(workspace_bytes_for((u64)n + 64));
        MrleEncScratch sc;
        mrle_encode_size(d, (u32)n, sc, a, e.s);
        const s32 size = (s32)(32u + e.word(sc.total));
        mrle_encode_write(d, (u32)n, sc, o, e.s);
        e.down(out, o, (size_t)size);
        return size;
    });
}

BZIP3_API int bz3_hip_stage_mrle_decode(const uint8_t * in, uint8_t * out, int32_t outlen, int32_t maxin) {
    return stage_guard([&]() -> int {
        if (maxin < 32) return 1;
        StageEnv e;
        u8 * d = e.dev((size_t)maxin + 16, in, (size_t)maxin);
        u8 * o = e.dev((size_t)outlen + 64);
        u32 * w = (u32 *)e.dev(64);
        Arena a = e.ctx->arena_for(workspace_bytes_for((u64)maxin + 64));
        mrle_decode(d, (u32)maxin, o, (u32)outlen, w, a, e.s);
        const u32 got = e.word(w);
        e.down(out, o, (size_t)(got < (u32)outlen ? got : (u32)outlen));
        return got != (u32)outlen;
    });
}

BZIP3_API int32_t bz3_hip_stage_lzp_encode(const uint8_t * in, int32_t n, uint8_t * out) {
    return stage_guard([&]() -> s32 {
        StageEnv e;
        u8 * d = e.dev((size_t)n + 64, in, (size_t)n);
        u8 * o = e.dev((size_t)n + 64);
        Arena a = e.ctx->arena_for(workspace_bytes_for((u64)n + 64));
        const s32 r = lzp_encode(d, (u32)n, o, a, e.s);
        if (r > 0) e.down(out, o, (size_t)r);
        return r;
    });
}

BZIP3_API int32_t bz3_hip_stage_lzp_decode(const uint8_t * in, int32_t n, uint8_t * out, int32_t max) {
    return stage_guard([&]() -> s32 {
        StageEnv e;
        u8 * d = e.dev((size_t)n + 64, in, (size_t)n);
        u8 * o = e.dev((size_t)max + 64);
        Arena a = e.ctx->arena_for(workspace_bytes_for((u64)n + 64));
        if (n < 4) return -1;  // :252
        u32 * d_result = a.take<u32>(4);
        LzpDecodeJob job{dev_addr(d), dev_addr(o), dev_addr(a.take<u32>(LZP_LUT_WORDS)), dev_addr(d_result), (u32)n, (u32)max};
        LzpDecodeJob * d_job = a.take<LzpDecodeJob>(1);
        lzp_decode_batch(&job, d_job, 1, e.s);
        const s32 r = (s32)e.word(d_result);
        if (r > 0) e.down(out, o, (size_t)r);
        return r;
    });
}

// Wall time of the last bz3_hip_stage_bwt / bz3_hip_stage_unbwt call's transform alone (both are synchronous: from the first launch to
// the last result on the host; the hook's own allocations and PCIe copies are outside).  Profiling only.
static std::atomic<float> g_stage_ms{0.f};
BZIP3_API float bz3_hip_stage_last_ms(void) { return g_stage_ms.load(); }

BZIP3_API int32_t bz3_hip_stage_bwt(const uint8_t * in, uint8_t * out, int32_t n) {
    return stage_guard([&]() -> s32 {
        StageEnv e;
        u8 * d = e.dev((size_t)n + 64, in, (size_t)n);
        u8 * o = e.dev((size_t)n + 64);
        Arena a = e.ctx->arena_for(workspace_bytes_for((u64)n + 64));
        HIP_CHECK(hipStreamSynchronize(e.s));
        const auto t0 = std::chrono::steady_clock::now();
        const s32 idx = bwt_forward(d, (u32)n, o, a, e.s, nullptr);
        g_stage_ms.store(std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count());
        e.down(out, o, (size_t)n);
        return idx;
    });
}

BZIP3_API int32_t bz3_hip_stage_unbwt(const uint8_t * in, uint8_t * out, int32_t n, int32_t idx) {
    return stage_guard([&]() -> s32 {
        if (n < 0) return -1;
        if (n <= 1) {
            if (idx != n) return -1;
            if (n == 1) out[0] = in[0];
            return 0;
        }
        if (idx <= 0 || idx > n) return -1;
        StageEnv e;
        u8 * d = e.dev((size_t)n + 64, in, (size_t)n);
        u8 * o = e.dev((size_t)n + 64);
        Arena a = e.ctx->arena_for(workspace_bytes_for((u64)n + 64));
        HIP_CHECK(hipStreamSynchronize(e.s));
        const auto t0 = std::chrono::steady_clock::now();
        bwt_inverse(d, (u32)n, (u32)idx, o, a, e.s);
        HIP_CHECK(hipStreamSynchronize(e.s));
        g_stage_ms.store(std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count());
        e.down(out, o, (size_t)n);
        return 0;
    });
}

// Tests only: the CU masks of the partition (DeviceCtx::cu_masks) for a device of `cus` CUs and `reserve` reserved ones: words = (cus + 31) / 32 each.
BZIP3_API int32_t bz3_hip_debug_cu_masks(int cus, int reserve, uint32_t * side, uint32_t * rest) {
    if (cus < 32 || cus > 1024 || reserve < 8 || !side || !rest) return -1;
    std::vector<uint32_t> a, b;
    DeviceCtx::cu_masks(cus, reserve, a, b);
    for (size_t k = 0; k < a.size(); k++) {
        side[k] = a[k];
        rest[k] = b[k];
    }
    return (int32_t)a.size();
}

// Tests only: sort.hip's device-wide exclusive scan on a host buffer (in place); returns the grand total through *total.
BZIP3_API int32_t bz3_hip_debug_scan_u32(uint32_t * data, uint32_t n, uint32_t * total) {
    return stage_guard([&]() -> s32 {
        if (n == 0) return -1;
        StageEnv e;
        u32 * d = (u32 *)e.dev((size_t)n * 4 + 64, data, (size_t)n * 4);
        u32 * t = (u32 *)e.dev(64);
        Arena a = e.ctx->arena_for(scan_temp_words(n) * 4 + (1u << 20));
        exclusive_scan_u32(d, n, t, a, e.s);
        e.down(data, d, (size_t)n * 4);
        if (total) *total = e.word(t);
        return 0;
    });
}

// Tests only: the stable LSD radix sort of sort.hip on host buffers -- (keys[i], i) sorted over key bits [0, key_bits) with digits of
// digit_bits (8 or 9) bits; passes of up to RS_RAW_TILES tiles take the scatter that reads the raw count table (round 5), larger ones
// the scanned table.  Returns the number of passes, -1 on bad arguments.
BZIP3_API int32_t bz3_hip_debug_sort_u32(const uint32_t * keys, uint32_t n, int key_bits, int digit_bits, uint32_t * sorted_keys, uint32_t * sorted_index) {
    return stage_guard([&]() -> s32 {
        if ((digit_bits != 8 && digit_bits != 9) || key_bits < 1 || key_bits > 32 || n == 0) return -1;
        StageEnv e;
        u32 * k[2] = {(u32 *)e.dev((size_t)n * 4 + 64, keys, (size_t)n * 4), (u32 *)e.dev((size_t)n * 4 + 64)};
        u32 * v[2] = {(u32 *)e.dev((size_t)n * 4 + 64), (u32 *)e.dev((size_t)n * 4 + 64)};
        Arena a = e.ctx->arena_for(radix_temp_bytes(n, digit_bits) + (1u << 20));
        int cur = 0, passes = 0;
        for (int shift = 0; shift < key_bits; shift += digit_bits, passes++) {
            const u32 * vin = passes ? v[cur] : nullptr;  // the first pass generates the indices
            if (digit_bits == 9) radix_pass_bits<u32, 9>(k[cur], k[cur ^ 1], vin, v[cur ^ 1], n, shift, 0xFFFFFFFFu, 0u, a, e.s);
            else radix_pass<u32>(k[cur], k[cur ^ 1], vin, v[cur ^ 1], n, shift, 0xFFFFFFFFu, 0u, a, e.s);
            cur ^= 1;
        }
        e.down(sorted_keys, k[cur], (size_t)n * 4);
        e.down(sorted_index, v[cur], (size_t)n * 4);
        return passes;
    });
}

// One CM job through the variant the current mode selects (auto = full model for a single block); a block the
// row-cache kernel gives up is coded again by the full-model kernel, as in run_cm_jobs.
extern "C++" template <class Job, class Launch>
void stage_cm_job(StageEnv & e, Job job, Launch && go) {
    const int variant = cm_variant_for(e.ctx, 1, std::is_same<Job, CmEncodeJob>::value);
    u32 * status = nullptr;
    if (cm_variant_has_rows(variant)) {
        job.spill = dev_addr(e.dev(CM_SPILL_BYTES));
        status = (u32 *)e.dev(64);
        HIP_CHECK(hipMemsetAsync(status, 0, 64, e.s));  // on the launching stream: a non-blocking stream does not order with the null stream
        job.status = dev_addr(status);
        job.miss_base = cm_variant_is_test(variant) ? 64u : CM_MISS_BASE;
        job.miss_shift = cm_variant_is_test(variant) ? 3u : CM_MISS_SHIFT;
    }
    Job * d_job = (Job *)e.dev(sizeof job, &job, sizeof job);
    go(d_job, 1u, e.s, variant);
    if (cm_variant_has_rows(variant) && e.word(status) != 0u) {
        g_cm_given_up.fetch_add(1u);
        go(d_job, 1u, e.s, (int)CM_VARIANT_FULL);
    }
}

BZIP3_API int32_t bz3_hip_stage_cm_encode(const uint8_t * in, int32_t n, uint8_t * out) {
    return stage_guard([&]() -> s32 {
        StageEnv e;
        u8 * d = e.dev((size_t)n + 64, in, (size_t)n);
        u8 * o = e.dev(bz3_bound((size_t)n) + 64);
        u32 * w = (u32 *)e.dev(64);
        const char * dbg = getenv("BZ3_CM_DEBUG");  // profiling only: 1 = coder alone, 2 = model alone (output invalid)
        CmEncodeJob job{dev_addr(d), dev_addr(o), dev_addr(w), (u32)n, dbg ? (u32)atoi(dbg) : 0u};
        // tests: BZ3_CM_TEST_GAP=<g> codes IN PLACE, the input g bytes above the output in one buffer (cm.hip CmSink), with a
        // side buffer of BZ3_CM_TEST_SIDE bytes (default 64 KiB); returns -1 when the side buffer overflowed
        const char * tg = getenv("BZ3_CM_TEST_GAP");
        u8 * side = nullptr;
        if (tg) {
            const size_t g = (size_t)atol(tg);
            const char * ts = getenv("BZ3_CM_TEST_SIDE");
            const size_t side_cap = ts ? (size_t)atol(ts) : CM_SIDE_BYTES;
            u8 * both = e.dev(g + (size_t)n + 64);
            HIP_CHECK(hipMemcpy(both + g, d, (size_t)n, hipMemcpyDeviceToDevice));
            side = e.dev(side_cap + 64);
            o = both;
            job.in = dev_addr(both + g);
            job.out = dev_addr(both);
            job.gap = (u32)g;
            job.side = dev_addr(side);
            job.side_cap = (u32)side_cap;
        }
        stage_cm_job(e, job, [](const CmEncodeJob * j, u32 nj, hipStream_t st, int variant) { cm_encode_batch(j, nj, st, variant); });
        const u32 coded = e.word(w), sw = e.word(w + 1);
        if (coded == 0xFFFFFFFFu) return -1;
        const u32 head = (tg && sw < coded) ? sw : coded;
        e.down(out, o, (size_t)head);
        if (head < coded) e.down(out + head, side, (size_t)(coded - head));
        return (s32)coded;
    });
}

BZIP3_API void bz3_hip_stage_cm_decode(const uint8_t * in, int32_t in_size, uint8_t * out, int32_t n) {
    stage_guard([&]() -> int {
        StageEnv e;
        u8 * d = e.dev((size_t)in_size + 64, in, (size_t)in_size);
        u8 * o = e.dev((size_t)n + 64);
        const char * dbg = getenv("BZ3_CM_DEBUG");  // profiling only (output invalid)
        CmDecodeJob job{dev_addr(d), dev_addr(o), (u32)in_size, (u32)n, dbg ? (u32)atoi(dbg) : 0u, 0u};
        stage_cm_job(e, job, [](const CmDecodeJob * j, u32 nj, hipStream_t st, int variant) { cm_decode_batch(j, nj, st, variant); });
        e.down(out, o, (size_t)n);
        return 0;
    });
}

// Profiling: `copies` identical CM decode jobs in ONE launch (same coded input, one output buffer each), through the kernel
// variant the current mode selects (no hand-back of given-up blocks: this measures the variant itself).  Returns the launch
// time in ms (HIP events).  out receives the n decoded bytes of copy 0.  With BZ3_CM_DEBUG=3 the guess-ahead decoder leaves
// cycle counters instead of the first output bytes (walker: wait, walk, slow-path bytes, wrong guesses at u64[0..3]; model
// wave 1: speculate, wait, redo, wrong guesses at u64[8..11]); `counters`, if not NULL, receives u64[16] per copy.
BZIP3_API float bz3_hip_stage_cm_decode_many(const uint8_t * in, int32_t in_size, uint8_t * out, int32_t n, int32_t copies, uint64_t * counters) {
    return stage_guard([&]() -> float {
        StageEnv e;
        if (copies < 1 || n < 256) return -1.f;
        u8 * d = e.dev((size_t)in_size + 64, in, (size_t)in_size);
        const size_t stride = ((size_t)n + 64 + 255) & ~(size_t)255;
        u8 * o = e.dev(stride * (size_t)copies);
        const char * dbg = getenv("BZ3_CM_DEBUG");
        const u32 debug = dbg ? (u32)atoi(dbg) : 0u;
        const int variant = cm_variant_for(e.ctx, (size_t)copies, false);
        u8 * spill = cm_variant_has_rows(variant) ? e.dev(CM_SPILL_BYTES * (size_t)copies) : nullptr;
        u32 * status = (u32 *)e.dev(4 * (size_t)copies + 64);
        HIP_CHECK(hipMemsetAsync(status, 0, 4 * (size_t)copies, e.s));
        std::vector<CmDecodeJob> jobs;
        for (int32_t k = 0; k < copies; k++) {
            CmDecodeJob j{dev_addr(d), dev_addr(o + stride * (size_t)k), (u32)in_size, (u32)n, debug, 0u};
            if (spill) {
                j.spill = dev_addr(spill + CM_SPILL_BYTES * (size_t)k);
                j.status = dev_addr(status + k);
                j.miss_base = CM_MISS_BASE;
                j.miss_shift = CM_MISS_SHIFT;
            }
            jobs.push_back(j);
        }
        CmDecodeJob * d_jobs = (CmDecodeJob *)e.dev(sizeof(CmDecodeJob) * jobs.size(), jobs.data(), sizeof(CmDecodeJob) * jobs.size());
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, e.s));
        cm_decode_batch(d_jobs, (u32)copies, e.s, variant, (debug & 15u) == 3u);
        HIP_CHECK(hipEventRecord(e1, e.s));
        HIP_CHECK(hipStreamSynchronize(e.s));
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        e.down(out, o, (size_t)n);
        if (counters)
            for (int32_t k = 0; k < copies; k++) e.down(counters + 16 * (size_t)k, o + stride * (size_t)k, 128);
        if (getenv("BZ3_CM_MANY_CHECK")) {  // every copy must have decoded the same bytes (-2 otherwise)
            std::vector<u8> other((size_t)n);
            for (int32_t k = 1; k < copies; k++) {
                e.down(other.data(), o + stride * (size_t)k, (size_t)n);
                if (memcmp(other.data(), out, (size_t)n) != 0) return -2.f;
            }
        }
        return ms;
    });
}

// Profiling: `copies` identical CM encode jobs in ONE launch through the encoder of the current CM kernel variant; returns the launch
// time in ms (HIP events) and the coded size of copy 0 in *coded (its bytes in `out`, capacity bz3_bound(n)).  BZ3_CM_DEBUG=1 / 2 runs
// the coder wave / the model waves alone (output invalid): which side of the LDS ring limits the kernel at a given co-residency.
// BZ3_CM_MANY_CHECK=1: every copy's coded bytes are compared with copy 0's (-2 when they differ).
BZIP3_API float bz3_hip_stage_cm_encode_many(const uint8_t * in, int32_t n, uint8_t * out, int32_t * coded, int32_t copies) {
    return stage_guard([&]() -> float {
        StageEnv e;
        if (copies < 1 || n < 1) return -1.f;
        u8 * d = e.dev((size_t)n + 64, in, (size_t)n);
        const size_t stride = (bz3_bound((size_t)n) + 64 + 255) & ~(size_t)255;
        u8 * o = e.dev(stride * (size_t)copies);
        u32 * w = (u32 *)e.dev(16 * (size_t)copies + 64);
        const char * dbg = getenv("BZ3_CM_DEBUG");
        const u32 debug = dbg ? (u32)atoi(dbg) : 0u;
        const int variant = cm_variant_for(e.ctx, (size_t)copies, true);
        u8 * spill = cm_variant_has_rows(variant) ? e.dev(CM_SPILL_BYTES * (size_t)copies) : nullptr;
        u32 * status = (u32 *)e.dev(4 * (size_t)copies + 64);
        HIP_CHECK(hipMemsetAsync(status, 0, 4 * (size_t)copies, e.s));
        u32 * claim = getenv("BZ3_CM_NO_CLAIM") ? nullptr : (u32 *)e.dev(CM_CLAIM_WORDS * 4);  // (BZ3_CM_NO_CLAIM: the block index decides which wave codes, as up to round 4)
        if (claim) HIP_CHECK(hipMemsetAsync(claim, 0, CM_CLAIM_WORDS * 4, e.s));
        std::vector<CmEncodeJob> jobs;
        for (int32_t k = 0; k < copies; k++) {
            CmEncodeJob j{dev_addr(d), dev_addr(o + stride * (size_t)k), dev_addr(w + 4 * (size_t)k), (u32)n, debug};
            j.claim = claim ? dev_addr(claim) : 0;
            if (spill) {
                j.spill = dev_addr(spill + CM_SPILL_BYTES * (size_t)k);
                j.status = dev_addr(status + k);
                j.miss_base = CM_MISS_BASE;
                j.miss_shift = CM_MISS_SHIFT;
            }
            jobs.push_back(j);
        }
        CmEncodeJob * d_jobs = (CmEncodeJob *)e.dev(sizeof(CmEncodeJob) * jobs.size(), jobs.data(), sizeof(CmEncodeJob) * jobs.size());
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, e.s));
        cm_encode_batch(d_jobs, (u32)copies, e.s, variant);
        HIP_CHECK(hipEventRecord(e1, e.s));
        HIP_CHECK(hipStreamSynchronize(e.s));
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        const u32 got = e.word(w);
        if (coded) *coded = (int32_t)got;
        if (got != 0xFFFFFFFFu && got <= bz3_bound((size_t)n) && !(debug & 15u)) {
            e.down(out, o, (size_t)got);
            if (getenv("BZ3_CM_MANY_CHECK")) {  // every copy must have coded the same bytes (-2 otherwise)
                std::vector<u8> other((size_t)got);
                for (int32_t k = 1; k < copies; k++) {
                    if (e.word(w + 4 * (size_t)k) != got) return -2.f;
                    e.down(other.data(), o + stride * (size_t)k, (size_t)got);
                    if (memcmp(other.data(), out, (size_t)got) != 0) return -2.f;
                }
            }
        }
        return ms;
    });
}

}  // extern "C"
