// cm.hip -- bzip3's context-mixing model + 32-bit binary arithmetic coder, one workgroup per block.
// Replaces begin / encode_bytes / decode_bytes (reference src/libbz3.c:333-494).
//
// The model -- C0 u16[256], C1 u16[256][256], C2 u16[512][17] = 148,992 B -- lives in the LDS of the compute unit that
// codes the block, so the 5 table reads and 4 counter updates per coded bit never leave the CU; HBM traffic is the
// algorithmic minimum (read n bytes, write the coded bytes, or the reverse).  Blocks are independent: a batch is ONE
// launch with one workgroup per block.  Kernel variants (stages.hpp CM_VARIANT_*):
//   full model   k_cm_encode / k_cm_decode_sync                       the whole C1 table in LDS: one block per CU
//   row cache    k_cm_encode_rows{,3} / k_cm_decode_sync{2,3}         only the C1 rows a block uses are resident (96 or 44 / 56
//                                                                     rows, the others spill to HBM): two / three blocks per CU
// (Rounds 1-2 also had a polling, a lock-step and a single-wave decoder; they were slower at every co-residency and were removed
// in round 3 -- profiles/HISTORY.md has their measurements.)
//
// Exact facts used:
//  * encode: all 8 tree nodes of a byte are known up front (the encoder knows the byte) and touch disjoint counters; the
//    model factorises by tree node, so every node gets a lane of its own and whole chunks of bytes are modelled ahead of
//    the coder; the (low, range) recurrence of the coder is inherently serial: one lane, fed through an LDS ring.
//  * decode: the bit is unknown until decoded, but the 255 nodes of the NEXT byte depend only on state that is fixed
//    once the previous byte is known, so 256 lanes (one tree node each) pre-evaluate every node's 18-bit probability
//    (the guess-ahead decoder even before that byte is known); the 8 serial decisions are speculated across lanes.
// All arithmetic is integer and matches the reference bit for bit, including the signed interpolation
// `x1 + (((x2 - x1) * (p & 4095)) >> 12)` with an arithmetic shift of a possibly negative product.
#include <atomic>

#include "prims.hpp"
#include "stages.hpp"

namespace bz3 {

constexpr int CM_C2_STRIDE = 17;

// R = 0: the whole order-1 table (256 rows) in LDS, one block per CU.  R > 0: only R rows of the order-1 table are
// resident ("row cache", see below), so that two workgroups share a CU's LDS.
template <int R>
struct CmLdsT {
    static constexpr int ROWS = R ? R : 256;
    u16 c1[ROWS * 256];
    u16 c0[256];
    u16 c2[512 * CM_C2_STRIDE];
};
using CmLds = CmLdsT<0>;

__device__ __forceinline__ u32 cm_readlane(u32 v, int lane) {
#ifdef BZ3_EMU
    return __shfl(v, lane);
#else
    return (u32)__builtin_amdgcn_readlane((int)v, lane);
#endif
}

// A zero the compiler cannot see through: what is derived from it stays in vector registers (a wave-uniform counter that would
// otherwise live in an SGPR and cost a v_mov at every use as a store offset).
__device__ __forceinline__ u32 cm_opaque_zero() {
#ifdef BZ3_EMU
    return 0;
#else
    u32 z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
#endif
}

// Tells the compiler a value is wave-uniform so that it lives in scalar registers and branches on it
// are scalar branches (the serial coder recurrences run entirely on the scalar unit).
__device__ __forceinline__ u32 cm_uniform(u32 v) {
#ifdef BZ3_EMU
    return v;
#else
    return (u32)__builtin_amdgcn_readfirstlane((int)v);
#endif
}

// a * b + c for a, b below 2^24, as the one full-rate instruction it is (left to the compiler, `__umul24(a, b) + c` becomes a
// quarter-rate v_mul_lo_u32 or v_mad_u64_u32 whenever it cannot prove the operand ranges itself)
__device__ __forceinline__ u32 cm_mad24(u32 a, u32 b, u32 c) {
#ifdef BZ3_EMU
    return a * b + c;
#else
    u32 r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#endif
}

// bits [lo, lo + width) of v: one v_bfe_u32 (the compiler does not know the value ranges that make `(v >> lo)` enough)
__device__ __forceinline__ u32 cm_bfe(u32 v, u32 lo, u32 width) {
#ifdef BZ3_EMU
    return (v >> lo) & ((1u << width) - 1u);
#else
    return __builtin_amdgcn_ubfe(v, lo, width);
#endif
}

// Nothing is scheduled across this point (keeps a prefetch where it is written instead of at the top of its basic block).
__device__ __forceinline__ void cm_sched_fence() {
#ifndef BZ3_EMU
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// Issue priority of the calling wave among the waves of its SIMD (s_setprio 3 = highest user level).
__device__ __forceinline__ void cm_raise_priority() {
#ifndef BZ3_EMU
    __builtin_amdgcn_s_setprio(3);
#endif
}

template <class M>
__device__ __forceinline__ void cm_model_init(M & m) {  // begin(): :350-358
    for (int i = threadIdx.x; i < M::ROWS * 256; i += blockDim.x) m.c1[i] = 32768;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) m.c0[i] = 32768;
    for (int i = threadIdx.x; i < 512 * CM_C2_STRIDE; i += blockDim.x) {
        const int k = i % CM_C2_STRIDE;
        m.c2[i] = (u16)((k << 12) - (k == 16));
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// encode: two waves.
//
// The model factorises exactly by tree node: C0[node], C1[*][node] and the two C2 rows of a node are touched only by the coded bits
// that pass through that node, and the encoder knows every byte up front.  So the "model" wave walks the block in chunks of 32 bytes:
//   chain loop : lanes 0..7 take one tree LEVEL each; for every byte of the chunk a lane reads the three counters of its level's node
//                on the byte's path, forms p, reads the two C2 cells, updates all four counters, and leaves (p, x1, x2, bit) in an
//                LDS event array.  Only counter traffic is on this serial path, and the LDS keeps it in order;
//   event loop : one lane per event finishes the interpolation (ssep, 18-bit probability) and writes the
//                16-byte coder event into the LDS ring.
// Wave 0 ("coder") drains the ring and runs the serial range recurrence.
//
// Coder formulation (exact).  With high == low + range, the reference's update
//     mid = low + ((range * P) >> 18);   bit ? high = mid : low = mid + 1                  (:388, :402)
// is   bit = 1:  range' = (range * P) >> 18                       low' = low
//      bit = 0:  range' = (range * (2^18 - P) - 1) >> 18          low' = low + (range - range')
// (range - ((range*P)>>18) - 1 == ((range*(2^18-P)) - 1) >> 18 for every range, P < 2^18), so the model waves
// ship (-s, -s, M, s) with M = P or 2^18 - P and the coder needs one v_mad_u64_u32, one 64-bit shift, one
// v_sub and one v_mad per bit -- no bit test, no selects.  The coder state is deliberately kept in VECTOR
// registers (seeded through an opaque v_mov): the events arrive in VGPRs from LDS broadcast reads, and moving
// them to the scalar unit would cost a v_readfirstlane per operand.  A single wave issues one instruction every ~5.4
// cycles (8.5 when dependent; profiles/r01_ubench_single_wave.txt), so instruction count is the currency.
// ------------------------------------------------------------------------------------------------
#ifdef BZ3_EMU
constexpr u32 CM_ENC_SWAP_SHIFT = 0;
#else
constexpr u32 CM_ENC_SWAP_SHIFT = 9;
#endif
constexpr u32 CM_RING = 64;   // bytes of look-ahead between the model waves and the coder wave (6 KiB of LDS: 8 events of 12 bytes per byte)
constexpr u32 CM_CHUNK = 32;  // bytes per model-wave chunk

#ifdef BZ3_EMU
#define LDS_PEEK(var) (*reinterpret_cast<const volatile u32 *>(&(var)))
#define LDS_POKE(var, v) (*reinterpret_cast<volatile u32 *>(&(var)) = (v))
#else
// relaxed workgroup-scope atomics on __shared__ words: plain ds_read_b32 / ds_write_b32 that the compiler
// neither caches in registers nor turns into flat (generic address space) accesses
#define LDS_PEEK(var) __hip_atomic_load(&(var), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define LDS_POKE(var, v) __hip_atomic_store(&(var), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#endif

// ------------------------------------------------------------------------------------------------
// Row cache (R > 0 kernels).  C1[c][node] is indexed by the previous byte c, and a block only ever touches the rows
// of byte values it contains: BWT output of text uses ~64 values for 99.9 % of its bytes.  The R > 0 kernels keep R
// rows in LDS and the others in a per-block 128 KiB spill area in global memory (u16[256][256]), so that TWO
// workgroups fit into one CU's LDS and the latency-bound coder recurrences of two blocks interleave on one CU.
// The cache is exact (it only decides where a row lives).  Every model lane owns the same tree node(s) in every
// row, so a wave moves its own 64 (128) cells of a row with one coalesced global store + load and never needs the
// other waves: each model wave keeps a PRIVATE copy of the cache directory and, because all of them see the same
// byte sequence, they make the same decisions (slots are handed out in order of first use, then recycled FIFO,
// skipping the rows the bytes in flight still need).
// A block whose working set does not fit (binary data: 256 live rows) would thrash; the kernels count misses and
// give the block up once misses > miss_base + (position >> miss_shift) -- status word = 1 -- and the host codes it
// again with the R = 0 kernel.
// ------------------------------------------------------------------------------------------------
constexpr u32 CM_ROW_SPILLED = 0xFEu;   // row_of[]: the row lives in the spill area
constexpr u32 CM_ROW_VIRGIN = 0xFFu;    // row_of[]: never touched, every counter still 32768
constexpr u32 CM_ABORT_MARK = 0xFFFFFFFEu;  // published instead of a position by a model wave that gave the block up

template <int R>
struct CmRowCache {
    static constexpr int SLOTS = R ? R : 1;
    u8 row_of[256];     // byte value -> slot | CM_ROW_SPILLED | CM_ROW_VIRGIN
    u8 sym_of[SLOTS];   // slot -> byte value
    u32 stamp[SLOTS];   // slot is pinned while stamp == the wave's current tick
};

struct CmRowState {  // wave-uniform registers of a model wave
    u32 nused = 1;   // slots handed out so far (slot 0 = byte value 0: the context before the block starts, :367)
    u32 hand = 0;    // FIFO pointer
    u32 misses = 0;
    u32 tick = 0;
};

template <int R>
__device__ __forceinline__ void cm_rows_init(CmRowCache<R> & rc) {  // by one wave, for its own directory
    const int lane = lane_id();
    for (int i = lane; i < 256; i += WAVE) rc.row_of[i] = (u8)(i == 0 ? 0u : CM_ROW_VIRGIN);
    for (int i = lane; i < CmRowCache<R>::SLOTS; i += WAVE) {
        rc.sym_of[i] = 0;
        rc.stamp[i] = 0;
    }
    wave_sync();
}

// Makes the row of byte value `sym` (state `st` = CM_ROW_SPILLED / CM_ROW_VIRGIN) resident and returns its slot.
// NODES = tree nodes per lane (node0, node0 + 64).  Wave-uniform control flow; every lane moves its own cells.
template <int R, int NODES, class M>
__device__ __forceinline__ u32 cm_rows_fetch(M & m, CmRowCache<R> & rc, CmRowState & rs, u16 * __restrict__ spill, u32 sym, u32 st, u32 node0) {
    const int lane = lane_id();
    u32 slot, victim = 0;
    const bool recycle = rs.nused >= (u32)R;
    if (!recycle) {
        slot = rs.nused++;
    } else {
        for (;;) {
            slot = rs.hand;
            rs.hand = rs.hand + 1u == (u32)R ? 0u : rs.hand + 1u;
            if (cm_uniform(rc.stamp[slot]) != rs.tick) break;
        }
        victim = cm_uniform((u32)rc.sym_of[slot]);
    }
    wave_sync();  // every lane has read the directory before lane 0 changes it
#pragma unroll
    for (int k = 0; k < NODES; k++) {
        const u32 cell = slot * 256u + node0 + 64u * (u32)k;
        if (recycle) spill[victim * 256u + node0 + 64u * (u32)k] = m.c1[cell];
        m.c1[cell] = st == CM_ROW_SPILLED ? spill[sym * 256u + node0 + 64u * (u32)k] : (u16)32768;
    }
    if (lane == 0) {
        if (recycle) rc.row_of[victim] = (u8)CM_ROW_SPILLED;
        rc.row_of[sym] = (u8)slot;
        rc.sym_of[slot] = (u8)sym;
        rc.stamp[slot] = rs.tick;
    }
    rs.misses++;
    wave_sync();
    return slot;
}

// The LDS ring between the model waves and the coder: per byte slot the 8 coder events as a structure of arrays, so that the coder
// fetches a byte with six 16-byte LDS reads (four for the addends, two for the multipliers) instead of sixteen 8- / 4-byte ones.
struct CmRing {
    uint2 k[CM_RING * 8];  // (-s, -s): the 64-bit addend of the coder's multiply-add, 0 or 2^64 - 1 (see the header comment)
    u32 m[CM_RING * 8];    // the multiplier M = P or 2^18 - P
};
struct CmByteEvents {      // the events of one byte in registers
    uint2 k[8];
    u32 m[8];
};
struct alignas(8) CmEvent {  // what the chain loop leaves for the event loop (8-byte aligned: it is stored and loaded as ONE 64-bit LDS access)
    u32 px1;      // p | x1 << 16
    u32 x2b;      // x2 | bit << 16
};

// Counter updates without a branch on the bit (:347-348).  For a 16-bit counter x and shift s,
//   bit = 1:  x + ((x ^ 65535) >> s)  ==  x - (x >> s) + (65535 >> s)      bit = 0:  x - (x >> s)
// so every update is "x - (x >> s) + K" with K = bit ? (65535 >> s) : 0; the two C2 cells (s = 6) are done as one
// packed 2 x u16 operation on the 32-bit word that holds both.
__device__ __forceinline__ u32 cm_upd(u32 x, int s, u32 k) { return x - (x >> s) + k; }
__device__ __forceinline__ u32 cm_upd_pair6(u32 w, u32 k2) {
    const u32 sh = (w >> 6) & 0x03FF03FFu;  // per-half x >> 6
    return w - sh + k2;                     // no borrow/carry crosses the halves: each half stays within [0, 65535]
}

// The two neighbouring C2 cells x1 | x2 << 16 at a u16 address.  A C2 row is 17 cells (34 bytes), so half of these pairs straddle a
// dword boundary, and an LDS dword access off its alignment is REPLAYED by the hardware at ~64 cycles per wave instruction: round 4's
// counters of the decoder at three blocks per CU show the LDS busy 56 % of all cycles, three quarters of that in SQ_LDS_UNALIGNED_STALL
// (profiles/r04_pmc_decoder_lds.txt) -- every evaluation of every model wave paid it.  HALVES: two 16-bit accesses instead (each aligned).
template <bool HALVES>
__device__ __forceinline__ u32 cm_pair_load(const u16 * p) {
    if (HALVES) return (u32)p[0] | ((u32)p[1] << 16);
    return reinterpret_cast<const PackedU32 *>(p)->v;
}
template <bool HALVES>
__device__ __forceinline__ void cm_pair_store(u16 * p, u32 w) {
    if (HALVES) {
        p[0] = (u16)w;
        p[1] = (u16)(w >> 16);
    } else {
        reinterpret_cast<PackedU32 *>(p)->v = w;
    }
}
#ifndef BZ3_EMU  // the same through 32-bit LDS pointers (the decoder's model waves); the emulator's LDS is ordinary memory
template <bool HALVES>
__device__ __forceinline__ u32 cm_pair_load(const __attribute__((address_space(3))) u16 * p) {
    if (HALVES) return (u32)p[0] | ((u32)p[1] << 16);
    return reinterpret_cast<const __attribute__((address_space(3))) PackedU32 *>(p)->v;
}
template <bool HALVES>
__device__ __forceinline__ void cm_pair_store(__attribute__((address_space(3))) u16 * p, u32 w) {
    if (HALVES) {
        p[0] = (u16)w;
        p[1] = (u16)(w >> 16);
    } else {
        reinterpret_cast<__attribute__((address_space(3))) PackedU32 *>(p)->v = w;
    }
}
#endif
// Round-6 decoder experiments (compile-time, tools/build_variant.py; profiles/r06_cm_decoder_experiments.txt has what they measured):
//   CM_EXP_EARLY_ROW   the model waves read C1[m][node] of the byte value m that the current run interrupted (the likeliest byte after a wrong guess: a third
//                      of them) BEFORE barrier 1, so that a wrong guess that turns out to be m does not wait for that LDS round trip in the repair
//   CM_EXP_PREFETCH    the walker asks for the speculative table of byte i+1 in the middle of the walk of byte i, behind ready flags of the four model waves
#ifndef CM_EXP_EARLY_ROW
#define CM_EXP_EARLY_ROW 0
#endif
#ifndef CM_EXP_PREFETCH
#define CM_EXP_PREFETCH 0
#endif
constexpr bool CM_ENC_PAIR_HALVES = false;  // (measured in round 4: the encoder gains nothing from the halves -- its eight chain lanes pay two instructions more per bit instead)
constexpr bool CM_DEC_PAIR_HALVES = true;   // the decoder's model waves: 767.7 -> 618.3 ns per byte and block at three blocks per CU together with the subtree form (profiles/r04_cm_decoder_experiments.txt)

// One byte of the chain, for the lane of one tree level: wave-uniform (c, c1 << 8, c2 << 8, f); the level's node on the byte's path.
template <class M>
__device__ __forceinline__ void cm_chain_step(M & m, CmEvent * __restrict__ ev_row, u32 hibit, u32 shr, u32 bitpos, u32 c, u32 c1s, u32 c2s, u32 f) {
    const u32 node = hibit | (c >> shr);
    const u32 bit = (c >> bitpos) & 1u;
    const u32 mk = 0u - bit;                        // all ones when the bit is 1
    const u32 a1 = c1s + node, a2 = c2s + node;     // C1 indices (c1 * 256 + node)
    const u32 p0 = m.c0[node];
    const u32 p1 = m.c1[a1];
    const u32 p2 = m.c1[a2];
    const u32 p = cm_mad24(p0 + p1, 7u, 2u * p2) >> 4;  // :380 (p0 + p1 < 2^17)
    const u32 ci = (2u * node + f) * CM_C2_STRIDE + (p >> 12);
    const u32 w = cm_pair_load<CM_ENC_PAIR_HALVES>(&m.c2[ci]);  // x1 | x2 << 16 (cells j, j+1)
    m.c0[node] = (u16)cm_upd(p0, 2, mk & 16383u);   // :396-399 / :411-414
    m.c1[a1] = (u16)cm_upd(p1, 4, mk & 4095u);
    cm_pair_store<CM_ENC_PAIR_HALVES>(&m.c2[ci], cm_upd_pair6(w, mk & 0x03FF03FFu));
    CmEvent e;
    e.px1 = p | (bit << 16);
    e.x2b = w;
    *ev_row = e;
}

// One chunk of the model wave: the chain (lanes 0..7, one tree level each, serial over the bytes), then the events (all lanes).
template <bool FULL, class M>
__device__ __forceinline__ void cm_model_chunk(M & m, CmEvent * __restrict__ ev, CmRing & ring, const u32 packed, const u32 fmask, const u32 cnt, const u32 base,
                                               const bool chain_lane, const u32 hibit, const u32 shr, const u32 bitpos, const u32 lvl) {
    const int lane = lane_id();
    CmEvent * __restrict__ ev_lvl = ev + lvl * CM_CHUNK;
    // ---- chain loop: serial in time, parallel over the levels ------------------------------------------
    u32 wv[CM_CHUNK];  // byte r | row of byte r-1 << 8 | row of byte r-2 << 16   (wave-uniform; row = byte value when R = 0)
#pragma unroll
    for (int r = 0; r < (int)CM_CHUNK; r++) wv[r] = cm_readlane(packed, r);
    if (chain_lane) {
#pragma unroll
        for (int r = 0; r < (int)CM_CHUNK; r++) {
            if (FULL || (u32)r < cnt) cm_chain_step(m, ev_lvl + r, hibit, shr, bitpos, wv[r] & 0xFFu, wv[r] & 0xFF00u, (wv[r] >> 8) & 0xFF00u, (fmask >> r) & 1u);
        }
    }
    wave_sync();
    // ---- event loop: one lane per event, 18-bit probability -> coder event ----------------------------
    const u32 nev = 8u * CM_CHUNK;
    for (u32 e0 = 0; e0 < nev; e0 += WAVE) {
        const u32 e = e0 + (u32)lane;
        const u32 r = e % CM_CHUNK, k = e / CM_CHUNK;
        if (r < cnt) {
            const CmEvent q = ev[e];
            const int p = (int)(q.px1 & 0xFFFFu), x1 = (int)(q.x2b & 0xFFFFu), x2 = (int)(q.x2b >> 16);
            const u32 bit = q.px1 >> 16;
            const int ssep = x1 + (((x2 - x1) * (p & 4095)) >> 12);  // :385
            const u32 p18 = (u32)(ssep * 3 + p);                     // :388
            const u32 neg = bit ? 0u : 0xFFFFFFFFu;
            const u32 at = ((base + r) & (CM_RING - 1)) * 8 + k;
            ring.k[at] = make_uint2(neg, neg);
            ring.m[at] = bit ? p18 : (1u << 18) - p18;
        }
    }
}

// Coder steps K0 .. K0+CNT-1 of one byte without renormalisation and without any test.  The result must not be used when the final
// interval lies within one 2^24 bucket (=> a renormalisation was due after one of the bits, :390), or when the interval shrank to a
// single value on the way (a renormalisation was due there, and a 0 bit coded from that state wraps the range, after which the
// intervals are no longer nested): rmin follows the smallest range on the way for the caller's test.
template <int K0, int CNT>
__device__ __forceinline__ void cm_code_bits_raw(const CmByteEvents & ev, u32 & r, u32 & l, u32 & rmin) {
#pragma unroll
    for (int kk = K0; kk < K0 + CNT; kk++) {
        const uint2 ek = ev.k[kk];  // (-s, -s) and M: see the header comment
        const u64 prod = (u64)r * ev.m[kk] + (((u64)ek.y << 32) | ek.x);
        const u32 r2 = (u32)(prod >> 18);
        l += (r - r2) & ek.x;
        r = r2;
        rmin = r < rmin ? r : rmin;
    }
}
// Where the coded bytes go.  Normally `out`, a buffer of its own.  In-place coding (gap != CM_NO_GAP): `out` lies
// `gap` bytes BELOW the input inside the same buffer, so a byte may only be stored below the input bytes that every
// model wave has already loaded: while byte i is being coded these are the chunks up to and including the one that
// holds i.  Compressed output trails its input by construction; should it ever catch up (the prefix coded so far
// expands by more than the buffer's slack, the n/50 + 32 bytes bz3_bound adds) the sink switches, for the rest of
// the block, to the side buffer, and the host appends that part once the input is dead.
struct CmSink {
    u8 * __restrict__ out;
    u8 * __restrict__ side;
    u32 gap, side_cap, n;
    u32 op = 0;                // bytes coded so far
    u32 sw = 0xFFFFFFFFu;      // first byte that went to the side buffer
    u32 failed = 0;            // side buffer exhausted
    __device__ __forceinline__ void put(u32 byte, u32 i) {
        if (sw == 0xFFFFFFFFu && gap != CM_NO_GAP) {
            const u32 loaded = (i | (CM_CHUNK - 1u)) + 1u;  // input bytes below this index are in registers
            if ((u64)op >= (u64)gap + (loaded < n ? loaded : n)) sw = op;
        }
        if (sw == 0xFFFFFFFFu) out[op] = (u8)byte;
        else if (op - sw < side_cap) side[op - sw] = (u8)byte;
        else failed = 1;
        op++;
    }
};

// The same steps with the reference's test after every bit (:390-394).
template <int K0, int CNT>
__device__ __forceinline__ void cm_code_bits_checked(const CmByteEvents & ev, u32 & range, u32 & low, CmSink & sink, u32 i) {
#pragma unroll
    for (int kk = K0; kk < K0 + CNT; kk++) {
        const uint2 ek = ev.k[kk];
        const u64 prod = (u64)range * ev.m[kk] + (((u64)ek.y << 32) | ek.x);
        const u32 r2 = (u32)(prod >> 18);
        low += (range - r2) & ek.x;
        range = r2;
        if (__builtin_expect(__ballot(range < (1u << 24)) != 0ull, 0)) {  // necessary for (low ^ high) < 2^24; the exact test follows
            while (__ballot((low ^ (low + range)) < (1u << 24)) != 0ull) {  // :390-394
                sink.put(low >> 24, i);
                low <<= 8;
                range = (range << 8) | 0xFFu;
            }
        }
    }
}

// Brings the rows of a chunk's bytes into the cache (R > 0).  mine = this lane's byte (lanes < cnt), hrow1 / hrow2 =
// slots of the two bytes before the chunk.  Returns the slot of this lane's byte.  The rows the chunk itself and the
// two history bytes refer to are pinned (stamped with the chunk's tick) before any slot is recycled.
template <int R, int NODES, class M>
__device__ __forceinline__ u32 cm_rows_chunk(M & m, CmRowCache<R> & rc, CmRowState & rs, u16 * __restrict__ spill, u32 mine, u32 cnt, u32 hrow1, u32 hrow2,
                                             u32 node0) {
    const int lane = lane_id();
    const bool live = (u32)lane < cnt;
    u32 rowv = live ? (u32)rc.row_of[mine] : 0u;
    u64 miss = __ballot(live && rowv >= CM_ROW_SPILLED);
    if (__builtin_expect(miss != 0ull, 0)) {
        rs.tick++;
        if (live && rowv < CM_ROW_SPILLED) rc.stamp[rowv] = rs.tick;
        if (lane == 0) {
            rc.stamp[hrow1] = rs.tick;
            rc.stamp[hrow2] = rs.tick;
        }
        wave_sync();
        while (miss != 0ull) {
            const int l = __ffsll((unsigned long long)miss) - 1;
            const u32 sym = cm_readlane(mine, l);
            const u32 st = cm_uniform((u32)rc.row_of[sym]);
            const u32 slot = cm_rows_fetch<R, NODES>(m, rc, rs, spill, sym, st, node0);
            const bool same = live && mine == sym;
            rowv = same ? slot : rowv;
            miss &= ~__ballot(same);
        }
    }
    return rowv;
}

// One block.  R = 0: whole model in LDS; R > 0: row cache (see above).
template <int R>
__device__ __forceinline__ void cm_encode_block(const CmEncodeJob * __restrict__ jobs) {
    static_assert(R == 0 || R >= (int)CM_CHUNK + 4, "a chunk pins up to CM_CHUNK + 2 rows: the cache must hold more than that");
    // one workgroup per block: blockIdx.x selects the job
    const u8 * __restrict__ in = global_ptr<const u8>(jobs[blockIdx.x].in);
    const u32 n = jobs[blockIdx.x].n;
    u8 * __restrict__ out = global_ptr<u8>(jobs[blockIdx.x].out);
    u32 * __restrict__ out_size = global_ptr<u32>(jobs[blockIdx.x].out_size);
    const u32 debug = jobs[blockIdx.x].debug & 15u;
    __shared__ CmLdsT<R> m;
    __shared__ __attribute__((aligned(16))) CmRing ring;
    __shared__ CmEvent ev[8 * CM_CHUNK];
    __shared__ u32 s_prod, s_cons;
    __shared__ CmRowCache<R> rc;  // R > 0: the directory of the row cache
    if (threadIdx.x == 0) s_prod = 0;
    if (threadIdx.x == 1) s_cons = 0;
    if (debug == 1)  // profiling only: a ring full of p = 1/2 events, so the lone coder emits exactly one byte per input byte
        for (u32 t = threadIdx.x; t < CM_RING * 8; t += blockDim.x) {
            ring.k[t] = make_uint2(0u, 0u);
            ring.m[t] = 1u << 17;
        }
    const int lane = lane_id();
    // Which of the two waves codes.  The hardware deals the waves of the workgroups of a CU round the four SIMDs in turn, so with two
    // waves per workgroup the first and the third workgroup of a CU land on the same pair of SIMDs, and two coder waves -- the critical
    // path of their blocks -- on one SIMD cost both a quarter of their pace.  Rounds 3-4 let the block index decide (blocks k, k + 256,
    // k + 512 share a CU when the dispatch is undisturbed: the third swaps its roles).  It mostly is at 256 MiB blocks (65.0 s per launch,
    // but 71.3 and 74.6 s in two of round 5's runs) and mostly is NOT for smaller ones: the encode launch of 768 x 32 MiB blocks took 10.76 s
    // in seven of eight launches and 8.25 s in the eighth (profiles/r05_cm_encoder_placement.txt).  Round 5: the workgroups of a CU CLAIM
    // the SIMD of their coder in a word per CU (HW_ID / XCC_ID name the CU and the SIMD a wave runs on): wave 0's SIMD if it is free, else
    // wave 1's.  (No claim word -- the emulator, old callers --: the block index decides as before; the emulator swaps every other block.)
    __shared__ u32 s_simd[2], s_coder;
    u32 claim_key = 0;
#ifndef BZ3_EMU
    {
        const u32 hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        claim_key = ((xcc & 15u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
        if (lane == 0) s_simd[wave_id() & 1] = (hw >> 4) & 3u;
    }
#endif
    cm_model_init(m);  // (ends with a barrier: both SIMD numbers are there)
    if (threadIdx.x == 0) {
        u32 coder = (blockIdx.x >> CM_ENC_SWAP_SHIFT) & 1u;
#ifndef BZ3_EMU
        if (jobs[blockIdx.x].claim) {
            u32 * __restrict__ claim = global_ptr<u32>(jobs[blockIdx.x].claim) + claim_key;
            const u32 b0 = 1u << s_simd[0], b1 = 1u << s_simd[1];
            if (!(atomicOr(claim, b0) & b0)) coder = 0;
            else {
                (void)atomicOr(claim, b1);
                coder = 1;
            }
        }
#endif
        s_coder = coder;
    }
    __syncthreads();
    const u32 role = cm_uniform((u32)wave_id()) ^ cm_uniform(s_coder);
    if (role != 0) {
        if (debug == 1) return;
        // ---- model wave ------------------------------------------------------------------------------
        // Lanes 0..7 walk the counter chain, one tree LEVEL each: the node of a level is picked by the byte (hibit | byte >> shr), its
        // three counters are read from LDS, updated and written back in program order -- the LDS serves a wave's requests in order,
        // so a counter that is hit again by the next byte needs no forwarding.  (Rounds 1-2 gave every NODE a lane and C0 a
        // register, which took three waves of which 8 lanes were busy; at three blocks per CU those ~125 wave instructions per
        // byte, not the coder's ~60, were what the SIMDs ran out of: round 3.)
        const u32 lvl = (u32)lane & 7u;
        const u32 hibit = 1u << lvl, shr = 8u - lvl, bitpos = 7u - lvl;
        const bool chain_lane = lane < 8;
        u32 cons_seen = 0;
        u32 hist = 0;  // the 4 bytes before the chunk, oldest in the top byte (zeros before the block starts)
        CmRowState rs;
        u16 * __restrict__ spill = global_ptr<u16>(jobs[blockIdx.x].spill);
        const u32 miss_base = jobs[blockIdx.x].miss_base, miss_shift = jobs[blockIdx.x].miss_shift;
        // In-place coding: giving a block up is only possible while the coded bytes cannot have reached the input
        // yet (the full-model kernel will read that input again): output <= bz3_bound(position) < gap.
        const u32 gap = jobs[blockIdx.x].gap;
        const u32 abort_limit = gap == CM_NO_GAP ? 0xFFFFFFFFu : (gap > 4096u ? (u32)(((u64)(gap - 2048u) * 32u) / 33u) : 0u);
        u32 hrow1 = 0, hrow2 = 0;  // slots of bytes -1 and -2 (byte value 0 before the block starts: slot 0)
        if (R) cm_rows_init<R>(rc);
        // The chunk's bytes are loaded ONE CHUNK AHEAD (round 5).  The wave used to load them where it needs them: an HBM round trip (1-2 us) in
        // front of every 32 bytes, on a wave whose own work per chunk is ~55 % of what the coder takes for it -- so the launch time moved with the
        // memory latency of the day (65.0 s on five boxes, 71.3 and 74.6 s on two: calls 7 / 8 of round 5, the decoder's launch identical on all).
        // (In-place coding: the sink only assumes that the bytes below the current chunk's end are in registers; loading further ahead is safe.)
        // (Unconditional loads at clamped addresses, masked where they are USED: behind a branch the compiler completes the load -- s_waitcnt -- inside it.)
        const u8 * __restrict__ in_pf = n ? in : reinterpret_cast<const u8 *>(jobs);  // (an empty block: the clamped address must still be memory of ours)
        u32 raw_next = in_pf[(u32)lane < n ? (u32)lane : (n ? n - 1u : 0u)];
        for (u32 base = 0; base < n; base += CM_CHUNK) {
            const u32 cnt = (n - base < CM_CHUNK) ? n - base : CM_CHUNK;
            while (debug != 2 && base + cnt - cons_seen > CM_RING) {  // ring full: wait for the coder
                cons_seen = LDS_PEEK(s_cons);
                if (base + cnt - cons_seen > CM_RING) BZ3_SPIN_PAUSE();
            }
            // lane r holds byte r of the chunk together with its 4 predecessors
            const u32 mine = ((u32)lane < cnt) ? raw_next : 0u;
            {
                const u64 nx = (u64)base + CM_CHUNK + (u32)lane;  // (u64: base + 32 + lane passes 2^32 only beyond the format's block limit, but be exact)
                raw_next = in_pf[nx < (u64)n ? nx : (u64)n - 1u];
            }
            u32 rowv = 0;
            if (R) {
                rowv = cm_rows_chunk<R, 4>(m, rc, rs, spill, mine, cnt, hrow1, hrow2, (u32)lane);  // (the wave moves whole rows: four cells per lane)
                if (__builtin_expect(rs.misses > miss_base + (base >> miss_shift) && base < abort_limit, 0)) {
                    // the working set does not fit: give the block up
                    if (lane == 0) *global_ptr<u32>(jobs[blockIdx.x].status) = 1u;
                    LDS_POKE(s_prod, CM_ABORT_MARK);
                    return;
                }
            }
            u32 prev4 = 0;  // bytes r-1, r-2, r-3, r-4 in bits 0-7, 8-15, 16-23, 24-31
#pragma unroll
            for (int d = 1; d <= 4; d++) {
                const u32 up = __shfl_up(mine, (unsigned)d);
                // lanes < d reach back into the previous chunk: hist holds byte -1 in bits 0-7 ... byte -4 in bits 24-31
                const u32 h = (hist >> (8u * (((u32)d - 1u - (u32)lane) & 3u))) & 0xFFu;
                prev4 |= (((u32)lane >= (u32)d) ? up : h) << (8 * (d - 1));
            }
            u32 packed = mine | ((prev4 & 0xFFFFu) << 8);
            if (R) {  // the chain indexes C1 by slot, not by byte value
                u32 r1 = __shfl_up(rowv, 1u), r2 = __shfl_up(rowv, 2u);
                r1 = lane >= 1 ? r1 : hrow1;
                r2 = lane >= 2 ? r2 : (lane == 1 ? hrow1 : hrow2);
                packed = mine | (r1 << 8) | (r2 << 16);
            }
            // run flag of byte i (:367-372): set iff i >= 2 and the four preceding bytes are equal
            const u32 i = base + (u32)lane;
            const bool fr = (u32)lane < cnt && i >= 2 && (prev4 & 0xFFu) == ((prev4 >> 8) & 0xFFu) && (prev4 & 0xFFFFu) == (prev4 >> 16);
            const u32 fmask = (u32)__ballot(fr);
            if (cnt == CM_CHUNK) cm_model_chunk<true>(m, ev, ring, packed, fmask, cnt, base, chain_lane, hibit, shr, bitpos, lvl);
            else cm_model_chunk<false>(m, ev, ring, packed, fmask, cnt, base, chain_lane, hibit, shr, bitpos, lvl);
            // history for the next chunk (only needed after a full chunk): byte -d of the next chunk = byte cnt-d of this one
            if (cnt == CM_CHUNK) {
                hist = cm_readlane(mine, (int)CM_CHUNK - 1) | (cm_readlane(mine, (int)CM_CHUNK - 2) << 8) | (cm_readlane(mine, (int)CM_CHUNK - 3) << 16) |
                       (cm_readlane(mine, (int)CM_CHUNK - 4) << 24);
                if (R) {
                    hrow1 = cm_readlane(rowv, (int)CM_CHUNK - 1);
                    hrow2 = cm_readlane(rowv, (int)CM_CHUNK - 2);
                }
            }
            lds_release();
            if (lane == 0) LDS_POKE(s_prod, base + cnt);
        }
        return;
    }
    // ---- coder wave: ONE active lane (an LDS read then returns 16 bytes, not 64 x 16) ------------------------
    if (debug == 2 || lane != 0) return;
    cm_raise_priority();  // the coder is the critical path of its block (measured at three per CU: -9 .. -14 % launch time, profiles/r02_cm_priority.txt)
    const u32 vzero = cm_opaque_zero();  // keeps the recurrence on the vector ALU
    u32 range = 0xFFFFFFFFu ^ vzero, low = vzero, prod_seen = debug == 1 ? 0xFFFFFFFFu : 0u;
    CmSink sink{out, global_ptr<u8>(jobs[blockIdx.x].side), jobs[blockIdx.x].gap, jobs[blockIdx.x].side_cap, n};
    // (Round 5 tried the slow path as a search for the FIRST due renormalisation in the values the fast pass has computed anyway -- ~66 instructions per
    // firing on paper against ~78 -- and measured it 6.3 % SLOWER on the launch, 8,641 against 8,130 ms at 768 x 32 MiB: tools/patches/cm_encoder_first_due_search.patch,
    // profiles/r05_cm_encoder_placement.txt.)
    // Fast path: all 8 bits of the byte without a single test.  While nothing is renormalised the intervals are
    // nested, so "a renormalisation was due after some bit" is equivalent to "the final interval lies within one
    // 2^24 bucket" (:390): one test per byte (plus the guard against a range that reached zero on the way).  If it fires
    // (about one byte in four) the byte is coded again in two halves of 4 bits, each first without tests and only then,
    // if its own test fires, bit by bit.
    // The lone live lane issues an instruction every 5-8 cycles whatever it is, so the instruction count of a byte IS the
    // encoder's time (round 3 took ~15 of ~75 out: no test after the first half, a running ring offset in a vector register
    // instead of seven instructions of address arithmetic per fetch, bytes coded in published groups so that the checks
    // "is the next byte there" / "tell the model waves" run once per pair, the result committed before the single-armed test).
    // Branches are on wave-uniform conditions (ballot of the single live lane -> s_cbranch_vccnz): a divergent
    // `if` would save/restore EXEC, and every EXEC write stalls the following VALU op.  __builtin_expect keeps
    // the slow paths out of line: the common case must FALL THROUGH (a taken branch costs ~40 cycles on a lone wave).
    //
    // Two register sets: the events of byte i+1 are fetched from the ring in the MIDDLE of byte i's recurrence whenever they
    // are there already (the model waves run a chunk of 32 bytes ahead), so their LDS latency hides behind bits 4-7 and
    // nothing is waited for when byte i+1 starts.  (Issued at the top of a byte, the compiler's wait for the current byte's
    // registers also waits for the loads just issued; the scheduling barriers keep the fetch where it is written.)
    u32 koff = vzero;  // byte offset of the NEXT byte's events in ring.k (64 bytes per byte; ring.m: half of it), a vector register
    auto fetch = [&](CmByteEvents & e) __attribute__((always_inline)) {
        const uint4 * __restrict__ pk = reinterpret_cast<const uint4 *>(reinterpret_cast<const u8 *>(ring.k) + koff);
        const uint4 * __restrict__ pm = reinterpret_cast<const uint4 *>(reinterpret_cast<const u8 *>(ring.m) + (koff >> 1));
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint4 q = pk[j];
            e.k[2 * j] = make_uint2(q.x, q.y);
            e.k[2 * j + 1] = make_uint2(q.z, q.w);
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const uint4 q = pm[j];
            e.m[4 * j] = q.x; e.m[4 * j + 1] = q.y; e.m[4 * j + 2] = q.z; e.m[4 * j + 3] = q.w;
        }
    };
    auto bucket_or_zero = [](u32 l, u32 r, u32 rmin) __attribute__((always_inline)) -> bool { return (l ^ (l + r)) < (1u << 24) || rmin == 0u; };
    auto code_byte = [&](const CmByteEvents & ev, CmByteEvents & next, const u32 i) __attribute__((always_inline)) {
        u32 r = range, l = low, rmin = 0xFFFFFFFFu;
        cm_code_bits_raw<0, 4>(ev, r, l, rmin);
        cm_sched_fence();
        koff = (koff + 64u) & (CM_RING * 64u - 1u);
        fetch(next);           // unconditional (a branch here would let the compiler move the fetch to the top of the byte): if byte i+1 is not
        cm_sched_fence();      // there yet the slot still holds an older byte and the caller fetches again after waiting
        const u32 r4 = r, l4 = l, rmin4 = rmin;
        cm_code_bits_raw<4, 4>(ev, r, l, rmin);
        const bool bad = bucket_or_zero(l, r, rmin);
        const u32 range_old = range, low_old = low;
        range = r;  // committed before the test: an `if` with one arm (cf. the decoder's walker)
        low = l;
        if (__builtin_expect(__ballot(bad) != 0ull, 0)) {
            range = range_old;
            low = low_old;
            if (__ballot(bucket_or_zero(l4, r4, rmin4)) == 0ull) {  // (the half's own bucket test is implied by the final one; its range-reached-zero guard is not)
                range = r4;
                low = l4;
            } else {
                cm_code_bits_checked<0, 4>(ev, range, low, sink, i);
            }
            r = range, l = low, rmin = 0xFFFFFFFFu;
            cm_code_bits_raw<4, 4>(ev, r, l, rmin);
            if (__ballot(bucket_or_zero(l, r, rmin)) == 0ull) {
                range = r;
                low = l;
            } else {
                cm_code_bits_checked<4, 4>(ev, range, low, sink, i);
            }
        }
    };
    // Waits until the model waves have published byte i (false: they gave the block up).
    auto wait_for = [&](const u32 i) __attribute__((always_inline)) -> bool {
        while (prod_seen <= i) {
            prod_seen = LDS_PEEK(s_prod);
            if (prod_seen <= i) BZ3_SPIN_PAUSE();
            if (R && prod_seen == CM_ABORT_MARK) return false;  // the model wave gave the block up; nothing of it is coded past its last chunk
        }
        lds_acquire();
        return true;
    };
    // the SIMD claimed for this coder is free again when the block is done (a launch with more workgroups than fit at once hands it to a later one)
    auto release_claim = [&]() __attribute__((always_inline)) {
#ifndef BZ3_EMU
        if (jobs[blockIdx.x].claim) (void)atomicAnd(global_ptr<u32>(jobs[blockIdx.x].claim) + claim_key, ~(1u << s_simd[s_coder & 1u]));
#endif
    };
    CmByteEvents eva, evb;
    for (u32 i = 0; i < n;) {
        // the bytes below lim are published (whole chunks of 32, the block's last one apart): coded in pairs, the two register sets
        // alternating.  koff points at byte i; the fetch inside code_byte moves it on.
        if (!wait_for(i)) {
            release_claim();
            return;
        }
        fetch(eva);  // (again, if the fetch behind the previous pair came too early)
        const u32 lim = prod_seen < n ? prod_seen : n;
        while (i + 2u <= lim) {
            code_byte(eva, evb, i);
            code_byte(evb, eva, i + 1u);
            i += 2u;
            if ((i & 15u) == 0u) LDS_POKE(s_cons, i);
        }
        if (i + 1u == lim) {  // one published byte left: the last byte of a block of odd length (lim == n)
            code_byte(eva, evb, i);
            i++;
        }
    }
    for (int j = 0; j < 4; j++) {  // flush (:425-432)
        sink.put(low >> 24, n - 1u);
        low <<= 8;
    }
    out_size[0] = sink.failed ? 0xFFFFFFFFu : sink.op;
    out_size[1] = sink.sw;
    release_claim();
}

constexpr int CM_ROWS_ENC = 96;   // 48 KiB of C1 rows: 76,008 B of LDS per workgroup, two workgroups per CU
constexpr int CM_ROWS_DEC = 96;   // 48 KiB of C1 rows: 72.1 KB of LDS per workgroup (112 rows = 80.6 KB: measured, two of those do NOT share a CU)
constexpr int CM_ROWS3_ENC = 44;  // 22 KiB of C1 rows: 49,124 B of LDS per workgroup, three workgroups per CU (a chunk pins up to 34 rows)
constexpr int CM_ROWS3_DEC = 56;  // 28 KiB of C1 rows: 50.8 KB of LDS per workgroup
#ifdef BZ3_EMU
constexpr int CM_ROWS_TEST = 40;  // emulator tests: small enough that short inputs recycle slots all the time
#endif

__global__ void __launch_bounds__(128) k_cm_encode(const CmEncodeJob * __restrict__ jobs) { cm_encode_block<0>(jobs); }
__global__ void __launch_bounds__(128) k_cm_encode_rows(const CmEncodeJob * __restrict__ jobs) { cm_encode_block<CM_ROWS_ENC>(jobs); }
__global__ void __launch_bounds__(128) k_cm_encode_rows3(const CmEncodeJob * __restrict__ jobs) { cm_encode_block<CM_ROWS3_ENC>(jobs); }
#ifdef BZ3_EMU
__global__ void __launch_bounds__(128) k_cm_encode_rows_test(const CmEncodeJob * __restrict__ jobs) { cm_encode_block<CM_ROWS_TEST>(jobs); }
#endif

// ------------------------------------------------------------------------------------------------
// decode: helpers of the guess-ahead decoder (cm_decode_block_sync below).
// ------------------------------------------------------------------------------------------------
template <u32 V>
struct CmConst {
    static constexpr u32 value = V;
};

// (a ^ b) + c in one instruction
__device__ __forceinline__ u32 cm_xad(u32 a, u32 b, u32 c) {
#ifdef BZ3_EMU
    return (a ^ b) + c;
#else
    u32 r;
    asm("v_xad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#endif
}
// acc = 2 * acc + bit, the bit given as a wave-wide vote mask (v_addc_co_u32 with the mask as carry-in)
__device__ __forceinline__ u32 cm_shift_in(u32 acc, u64 vote, bool bit) {
#ifdef BZ3_EMU
    (void)vote;
    return acc + acc + (bit ? 1u : 0u);
#else
    (void)bit;
    u64 carry_out;
    asm("v_addc_co_u32 %0, %1, %0, %0, %2" : "+v"(acc), "=s"(carry_out) : "s"(vote));
    return acc;
#endif
}

__device__ __forceinline__ u64 cm_clock() {
#ifdef BZ3_EMU
    return 0;
#else
    return (u64)__builtin_readcyclecounter();
#endif
}


// The coded bytes are read 64 at a time (one byte per lane) and handed out by v_readlane.  Bytes past the end read as -1 (:345).
// The NEXT 64 bytes are already on their way when a window runs out (round 5: the load used to be issued where its bytes were needed, an HBM round
// trip on the walker's path every 64 coded bytes).
#define CM_NEXT_BYTE(dst)                                                              \
    do {                                                                               \
        if (ip - ibase >= 64u) {                                                       \
            ibase += 64u;                                                              \
            window = (ibase + lane < in_size) ? window_raw : 0xFFFFFFFFu;              \
            {                                                                          \
                const u64 nx_ = (u64)ibase + 64u + (u32)lane;                          \
                window_raw = in_pf[nx_ < (u64)in_size ? nx_ : (in_size ? (u64)in_size - 1u : 0u)]; \
            }                                                                          \
        }                                                                              \
        dst = cm_readlane(window, (int)(ip - ibase));                                  \
        ip++;                                                                          \
    } while (0)
// ---- checked walk (slow path) -------------------------------------------------------------------------------
// Renormalisation of the surviving path (:470-474).  All valid lanes carry the same (low, range): take them from one
// of those lanes, shift on the scalar unit, and hand the result to every lane (the others are dead anyway).
#define CM_RENORM()                                                                                   \
    do {                                                                                              \
        const u64 need_ = valid & __ballot(range < (1u << 24));                                       \
        if (__builtin_expect(need_ != 0, 0)) {                                                        \
            const int w_ = __ffsll((unsigned long long)need_) - 1;                                    \
            u32 l_ = cm_readlane(low, w_), r_ = cm_readlane(range, w_);                               \
            while ((l_ ^ (l_ + r_)) < (1u << 24)) {                                                   \
                l_ <<= 8;                                                                             \
                r_ = (r_ << 8) | 0xFFu;                                                               \
                u32 b_;                                                                               \
                CM_NEXT_BYTE(b_);                                                                     \
                code = (code << 8) + b_;                                                              \
            }                                                                                         \
            low = l_;                                                                                 \
            range = r_;                                                                               \
        }                                                                                             \
    } while (0)
// One speculated level: the lane assumes its bit (NB = all-ones when that bit is 0); CK = lanes assuming a 1.
#define CM_SPEC_LEVEL(P, NB, CK)                                                                      \
    do {                                                                                              \
        const u32 t_ = (u32)(((u64)range * (P)) >> 32);            /* (range * p18) >> 18, :464 */    \
        const u32 mid_ = low + t_;                                                                    \
        /* NB: compare absolute values: a truncated stream feeds -1 bytes (:345) and can push `code` */ \
        /* below `low`, where the reference still decodes a 1.                                       */ \
        const u64 vote_ = __ballot(code <= mid_);                                                     \
        valid &= ~(vote_ ^ (CK));                                                                     \
        const u32 x_ = t_ ^ (NB);                                  /* bit 1: t; bit 0: ~t */          \
        range = x_ + (range & (NB));                               /* t  |  range - t - 1 */          \
        low -= x_ & (NB);                                          /* low |  mid + 1      */          \
        CM_RENORM();                                                                                  \
    } while (0)
// One decoded level (the lane takes the bit it decodes).
#define CM_REAL_LEVEL(P, BIT)                                                                         \
    do {                                                                                              \
        const u32 t_ = (u32)(((u64)range * (P)) >> 32);                                               \
        const u32 mid_ = low + t_;                                                                    \
        BIT = code <= mid_;                                                                           \
        range = BIT ? t_ : range - t_ - 1u;                                                           \
        low = BIT ? low : mid_ + 1u;                                                                  \
        CM_RENORM();                                                                                  \
    } while (0)

// ------------------------------------------------------------------------------------------------
// decode, barrier-synchronised guess-ahead ("sync"): five waves like cm_decode_block -- wave 0 walks, waves 1..4 hold one
// tree node per lane and evaluate the table of byte i+1 on the guess "byte i repeats byte i-1" WHILE the walker decodes
// byte i -- but the hand-offs are workgroup barriers instead of LDS mailboxes that both sides poll:
//     walker : decode byte i from table i, store it in s_done            | models : (speculative) table i+1
//                                         ---------------- barrier 1 ----------------
//     right guess: walker fetches table i+1 and goes on                  | models go on with table i+2
//     wrong guess: walker waits at barrier 2                             | models undo, apply the real update, evaluate
//                                         ---------------- barrier 2 ----------------  table i+1 again
// A wave that waits in s_barrier issues nothing, so workgroups that share a CU do not pay for each other's waiting (the
// polling decoder loses a factor 2 with two neighbours on the CU, profiles/r02_cm_coresidency.txt), there is no mailbox
// to miss (cf. the deadlock the polling protocol had), and every wave takes the same, data-determined number of barriers
// per byte: after barrier 1 all of them know byte i and the guess.  Everything that crosses a barrier alternates between two
// buffers, because the waves only meet AT the barriers: table i+2 is written into the buffer of table i, which the walker has
// read into registers before barrier 1 of byte i; byte i+1 is stored in the other word than byte i, which a model wave may read
// arbitrarily late after barrier 1 (found by the emulator's stalled-wave scheduling).
// ------------------------------------------------------------------------------------------------
// The model tables live in DYNAMIC LDS (size passed at launch).  With a large static LDS array the compiler pads the kernel's
// register allocation up to what its LDS-derived occupancy estimate allows -- 97 VGPRs for 50 KB, i.e. four waves per SIMD --
// and three five-wave workgroups then no longer fit a CU although their LDS does (measured: the 320-thread decoders ran two per
// CU, the third waited for a second round; tools/occupancy_probe.hip shows that the hardware co-schedules the shape happily).
// Address space of the LDS: a pointer of this kind is 32 bits wide and every access through it a ds_ instruction (a generic pointer
// would be 64 bits and its accesses flat_ instructions).  The emulator's LDS is ordinary memory.
#ifdef BZ3_EMU
#define CM_LDS
#else
#define CM_LDS __attribute__((address_space(3)))
#endif
struct CmEvalP {              // CmEval with the cells remembered by address
    CM_LDS u16 * a1;          // C1[c1][node]
    u32 p1;                   // its value
    CM_LDS u16 * ci;          // the first of the two C2 cells
    u32 w;                    // both cells, x1 | x2 << 16
};

template <int R, bool PROF>  // PROF: cycle counters instead of the first output bytes (profiling only, BZ3_CM_DEBUG=3)
__device__ __forceinline__ void cm_decode_block_sync(const CmDecodeJob * __restrict__ jobs, CmLdsT<R> & m) {
    const u8 * __restrict__ in = global_ptr<const u8>(jobs[blockIdx.x].in);
    const u32 in_size = jobs[blockIdx.x].in_size;
    u8 * __restrict__ out = global_ptr<u8>(jobs[blockIdx.x].out);
    const u32 n = jobs[blockIdx.x].n;
    __shared__ __attribute__((aligned(16))) u32 ptab[2][256];  // (18-bit probability of node) << 14 (aligned: the walker reads the two level-7 entries of a lane as one 64-bit access)
    __shared__ u32 s_done[2];     // [i & 1] = byte i, written by the walker before barrier 1 of byte i.  Two words: after a right guess the
                                  // walker decodes byte i+1 and stores it while a model wave that was held up may not have read byte i yet
    __shared__ u32 s_abort;       // R > 0: the model waves gave the block up
    __shared__ CmRowCache<R> rcs[R ? 4 : 1];  // R > 0: one private directory per model wave
    if (threadIdx.x < 2) s_done[threadIdx.x] = 0;
    if (threadIdx.x == 5) s_abort = 0;
    cm_model_init(m);
    if (n == 0) return;
    const int lane = lane_id();
    const u32 role = cm_uniform((u32)wave_id());
    u32 hw_id = 0, xcc_id = 0;  // PROF: where the hardware put this wave (HW_ID: wave, SIMD, CU, SE; XCC_ID), u32[22 + 2 * role] of the output
#ifndef BZ3_EMU
    if (PROF) {
        hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
    if (role != 0) {
        // ---- model waves ------------------------------------------------------------------------------------
        // These four waves are three quarters of the workgroup's instructions, and at three blocks per CU the waves of a SIMD take turns:
        // a model wave's instruction costs ~8 cycles, so the ~50 instructions of a byte with a right guess are what the walker (80) has to
        // hide.  Hence: LDS cells are remembered as address-space-3 POINTERS (the address that read a cell also writes it back: no index
        // arithmetic), the two C2 rows of the node (run flag 0 / 1) are pointers that only change when the flag does, both order-1
        // counters of the speculative table are the same cell (7 c0 + 9 cell), and two bytes are unrolled per loop trip.
        // Node of a lane (round 4): model wave w owns the 63 nodes BELOW the level-2 node 4 + w, i.e. levels 2..7 of every byte whose two
        // top bits are w, and one spare lane of each wave takes the root, the two level-1 nodes and nothing.  A byte's path then lies in
        // ONE wave plus the spare lanes, and the waves without a node on it fall through the update and undo blocks -- with one node per
        // lane in breadth-first order (rounds 2-3) every byte's path crossed three of the four waves.  With the root and node 2 (top bit
        // 0) in wave 1 -- its leaf for 0x7E / 0x7F moves to wave 3's spare lane to make room -- the bytes 0x40 .. 0x7D, the letters of
        // text, touch wave 1 only.  Measured at three blocks per CU: 731.3 -> 678.5 ns per byte and block (profiles/r04_cm_decoder_experiments.txt).
        u32 node = threadIdx.x - 64u;
        {
            const u32 w = node >> 6, l = node & 63u;
            if (l == 0u) {
                node = w == 0u ? 0u : (w == 1u ? 1u : (w == 2u ? 3u : 191u));  // (node 0: the lane owns none, nodelow = 1 > byte >> 8)
            } else {
                const u32 d = (u32)(31 - __clz((int)l));
                node = ((4u + w) << d) | (l - (1u << d));
            }
            if (w == 1u && l == 63u) node = 2u;  // (its own leaf 191 is wave 3's spare)
        }
        const u32 lvl = node ? (u32)(31 - __clz((int)node)) : 0u;
        const u32 hibit = 1u << lvl, shr = 8u - lvl, bitpos = 7u - lvl;
        const u32 nodelow = node ^ hibit;  // the bits of a byte that lead to this node, i.e. byte >> shr (the lane without a node: nodelow = 1 > byte >> 8)
        u32 c0 = 32768u;  // the node's C0 counter lives in a register
        CM_LDS u16 * const c1col = (CM_LDS u16 *)&m.c1[node];                                // C1[slot 0][node]; slot s is 512 bytes further
        CM_LDS u16 * const c2row0 = (CM_LDS u16 *)&m.c2[(2u * node) * CM_C2_STRIDE];         // run flag 0
        CM_LDS u16 * const c2row1 = (CM_LDS u16 *)&m.c2[(2u * node + 1u) * CM_C2_STRIDE];    // run flag 1
        CM_LDS u32 * const ptab0 = (CM_LDS u32 *)&ptab[0][node];
        CM_LDS u32 * const ptab1 = (CM_LDS u32 *)&ptab[1][node];
        // Probability of the node (:377-388) into *pt, given 16 p = (c0 + p1) * 7 + 2 * p2 and the node's C2 row for the run flag.
        auto evaluate = [&](CM_LDS u32 * pt, CM_LDS u16 * a1, u32 p1, u32 p16, CM_LDS u16 * c2row) __attribute__((always_inline)) -> CmEvalP {
            CmEvalP e;
            e.a1 = a1;
            e.p1 = p1;
#ifdef BZ3_EMU
            e.ci = c2row + cm_bfe(p16, 16, 4);  // cell p >> 12 of the row
#else
            {  // the same in two instructions (the compiler's own rendering of the line above takes three)
                u32 j, a;
                asm("v_bfe_u32 %0, %1, 16, 4" : "=v"(j) : "v"(p16));
                asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(a) : "v"(j), "v"((u32)(__UINTPTR_TYPE__)c2row));
                e.ci = (CM_LDS u16 *)(__UINTPTR_TYPE__)a;
            }
#endif
            e.w = cm_pair_load<CM_DEC_PAIR_HALVES>(e.ci);       // x1 | x2 << 16 (cells j, j + 1)
            const int p = (int)(p16 >> 4);
            const int x1 = (int)(e.w & 0xFFFFu), x2 = (int)(e.w >> 16);
            const int ssep = x1 + (((x2 - x1) * (p & 4095)) >> 12);
            *pt = cm_mad24((u32)ssep, 3u, (u32)p) << 14;       // (ssep < 2^16)
            return e;
        };
        // byte 0: nothing to guess (c1 = c2 = 0, run = 1, :367-372)
        const u32 first = *c1col;
        CmEvalP prev = evaluate(ptab0, c1col, first, cm_mad24(c0 + first, 7u, 2u * first), c2row0);
        u32 k1 = 0;        // newest confirmed byte (byte i-2 inside the loop; the initial c1 = 0 before the block starts)
        u32 run_prev = 1;  // run counter the evaluation of byte i-1 was made with
        CM_LDS u16 * c2row = c2row0;  // the C2 row for the run flag of the NEXT speculative evaluation (flag = run_prev + 1 > 2)
        CmRowCache<R> & rc = rcs[R ? role - 1 : 0];
        CmRowState rs;
        u16 * __restrict__ spill = global_ptr<u16>(jobs[blockIdx.x].spill);
        const u32 miss_base = jobs[blockIdx.x].miss_base, miss_shift = jobs[blockIdx.x].miss_shift;
        if (R) cm_rows_init<R>(rc);
        // the directory byte value -> slot also lives in a register (lane k: entries 4k .. 4k+3): the lookup on a wrong guess is a
        // v_readlane instead of an LDS round trip on the path the walker waits for
        u32 rowreg = R ? reinterpret_cast<const u32 *>(rc.row_of)[lane] : 0u;
        u32 mru = 0x100u;  // CM_EXP_EARLY_ROW: the byte value the current run interrupted (never k1); 0x100 = none yet
        u64 mprof_spec = 0, mprof_wait = 0, mprof_redo = 0;  // PROF
        __syncthreads();  // barrier 0: table 0 is there
        // One byte: `prev` = what the evaluation of byte i-1 left, `cur` receives that of byte i.  Two bytes per loop trip with the two
        // records swapping roles, so that neither they nor the table buffer / the s_done word of a byte cost a move or an address
        // computation (BUF = i & 1 is a compile-time constant).  Returns false when the block was given up.
        auto step = [&](const u32 i, auto buf_tag, const CmEvalP & prev, CmEvalP & cur) __attribute__((always_inline)) -> bool {
            constexpr u32 BUF = decltype(buf_tag)::value;
            CM_LDS u32 * const pt = BUF ? ptab1 : ptab0;
            bool give_up = false;
            u64 m0 = 0, m1 = 0, m2 = 0;
            if (PROF) m0 = cm_clock();
            // -- speculate: byte i-1 == k1.  Update of byte i-1 (:396-399, :411-414; branch-free, see cm_upd) ...
            const u32 g = k1;
            const bool on_g = (g >> shr) == nodelow;
            const u32 c0_old = c0;
            u32 cell = prev.p1;  // C1[k1][node]: byte i-1 was evaluated with c1 = k1, so this is the cell its update moves
            if (on_g) {
                const u32 mk = 0u - ((g >> bitpos) & 1u);
                c0 = cm_upd(c0, 2, mk & 16383u);
                cell = cm_upd(prev.p1, 4, mk & 4095u);
                *prev.a1 = (u16)cell;
                cm_pair_store<CM_DEC_PAIR_HALVES>(prev.ci, cm_upd_pair6(prev.w, mk & 0x03FF03FFu));
            }
            // ... and the table of byte i with c1 = g, c2 = k1: both order-1 counters are `cell`, the run counter goes up
            run_prev++;
            if (__builtin_expect(run_prev == 3u, 0)) c2row = c2row1;
            cur = evaluate(pt, prev.a1, cell, cm_mad24(cell, 9u, cm_mad24(c0, 7u, 0u)), c2row);
            u32 early = 0;       // CM_EXP_EARLY_ROW: C1[mru][node], asked for while the walker still walks
            bool early_ok = false;
            if (CM_EXP_EARLY_ROW && mru < 0x100u) {
                u32 row_m = mru;
                if (R) row_m = (cm_readlane(rowreg, (int)(mru >> 2)) >> (8u * (mru & 3u))) & 0xFFu;
                early_ok = !R || row_m < CM_ROW_SPILLED;  // (wave-uniform; a row that is not resident is fetched the ordinary way should it be needed)
                if (early_ok) early = *(c1col + row_m * 256u);  // nobody writes that row before the repair reads it: the updates of byte i-1 go to the row of byte i-2 = k1 != mru
            }
            if (PROF) m1 = cm_clock();
            __syncthreads();  // barrier 1: the walker has decoded byte i-1
            const u32 c = cm_uniform(LDS_PEEK(s_done[BUF ^ 1u]));
            if (PROF) {
                m2 = cm_clock();
                mprof_spec += m1 - m0;
                mprof_wait += m2 - m1;
            }
            k1 = c;
            if (c != g) {
                // wrong guess: put the old counters back (the old values are still in `prev`), apply the real update
                // and evaluate again.  The new c1 row differs from the row being repaired, so its read goes first.
                // (Issue priority for these waves during the repair -- the walker waits for it -- was measured in round 3: 804 -> 800 ns per
                // byte at three per CU, nothing.)
                u32 row = c;
                if (R) {
                    row = (cm_readlane(rowreg, (int)(c >> 2)) >> (8u * (c & 3u))) & 0xFFu;
                    if (__builtin_expect(row >= CM_ROW_SPILLED, 0)) {
                        // the only row still needed is the one of byte i-2 (prev.a1 points into it): pin it
                        rs.tick++;
                        if (lane == 0) rc.stamp[cm_uniform((u32)(prev.a1 - c1col) >> 8)] = rs.tick;
                        wave_sync();
                        row = cm_rows_fetch<R, 1>(m, rc, rs, spill, c, row, node);
                        rowreg = reinterpret_cast<const u32 *>(rc.row_of)[lane];
                        if (__builtin_expect(rs.misses > miss_base + (i >> miss_shift), 0)) {
                            // the working set does not fit (every model wave gets here at the same byte): the block is given up.  The repair
                            // below still runs -- no second way out of this branch, the compiler pays for one with moves on every path --
                            // and the walker reads s_abort behind barrier 2
                            give_up = true;
                            if (threadIdx.x == 64) *global_ptr<u32>(jobs[blockIdx.x].status) = 1u;
                            LDS_POKE(s_abort, 1u);
                        }
                    }
                }
                CM_LDS u16 * const a1 = c1col + row * 256u;
                u32 p1;
                if (CM_EXP_EARLY_ROW && early_ok && c == mru) p1 = early;  // (wave-uniform branch)
                else p1 = *a1;
                if (CM_EXP_EARLY_ROW) mru = g;  // the run of g's ends here: g is what a later wrong guess most likely returns to
                u32 cell2 = prev.p1;
                if (on_g) {
                    c0 = c0_old;
                    cm_pair_store<CM_DEC_PAIR_HALVES>(prev.ci, prev.w);
                }
                if ((c >> shr) == nodelow) {
                    const u32 mk = 0u - ((c >> bitpos) & 1u);
                    c0 = cm_upd(c0, 2, mk & 16383u);
                    cell2 = cm_upd(prev.p1, 4, mk & 4095u);
                    cm_pair_store<CM_DEC_PAIR_HALVES>(prev.ci, cm_upd_pair6(prev.w, mk & 0x03FF03FFu));
                }
                *prev.a1 = (u16)cell2;  // (unconditionally: storing the value that is there already costs less than finding out)
                c2row = c2row0;  // c != k1: the run counter restarts
                cur = evaluate(pt, a1, p1, cm_mad24(c0 + p1, 7u, 2u * cell2), c2row0);
                if (R) wave_sync();  // (test emulation: a row fetch de-synchronises the fibers of a wave; no instruction on the GPU)
                if (PROF) mprof_redo += cm_clock() - m2;  // up to the arrival at barrier 2
                __syncthreads();  // barrier 2: the corrected table of byte i is there
                run_prev = 0;
            }
            return !give_up;
        };
        CmEvalP other = prev;
        for (u32 i = 1; i < n;) {
            if (!step(i, CmConst<1>{}, prev, other)) return;
            if (++i >= n) break;
            if (!step(i, CmConst<0>{}, other, prev)) return;
            ++i;
        }
        if (PROF && n >= 256 && threadIdx.x == 64) {  // profiling only: model wave 1's phases (cycles) at u64[8..10] of the output
            u64 * o = reinterpret_cast<u64 *>(out) + 8;
            o[0] = mprof_spec;
            o[1] = mprof_wait;
            o[2] = mprof_redo;
        }
        if (PROF && n >= 256 && lane == 0) {
            u32 * o = reinterpret_cast<u32 *>(out) + 22 + 2 * role;
            o[0] = hw_id;
            o[1] = xcc_id;
        }
        return;
    }
    // ---- walker ---------------------------------------------------------------------------------------------
    // The walker is the critical path of its block: highest issue priority among the waves of its SIMD.  Measured with three blocks
    // per CU (the other blocks' model waves share the SIMD): 918 -> 805 ns per byte and block in round 2 (profiles/r02_cm_priority.txt),
    // 762 -> 737 in round 3 (profiles/r03_cm_decoder_steps.txt).
    //
    // A single wave issues one instruction every 5-8 cycles whatever it is, so the walker's instruction count per byte IS its time
    // (round 3: the loop the compiler made of the first version spent 165 instructions on a byte, 40 of them scalar moves between the
    // arms of its control flow).  What keeps the count down:
    //  * validity without votes: a lane that took a wrong turn ends with d > range and stays there (assumed 1 but d > t: range' = t < d;
    //    assumed 0 but d <= t: d' = d - t - 1 wraps above range' = range - t - 1; once d > range both ways keep it, and so do the two
    //    real levels, which then decode 0s), the lane on the true path keeps d <= range -- so the six speculated levels are mul_hi + four
    //    ALU operations each, and a comparison every other level names the surviving lane.  An invalid lane whose range reached zero
    //    can wrap back to d <= range (range - t - 1 with range = t = 0): more than one survivor sends the byte to the checked walk;
    //  * two bytes per loop trip, so that the table buffer, the s_done word and the LDS offsets of a byte are compile-time constants;
    //  * every lane stores the decoded byte to the same address (one instruction; its offset is a vector counter);
    //  * the cycle counters are a separate kernel instantiation (PROF).
    cm_raise_priority();
    const u32 ul = (u32)lane;
    const bool as0 = (ul >> 5) & 1u, as1 = (ul >> 4) & 1u, as2 = (ul >> 3) & 1u, as3 = (ul >> 2) & 1u, as4 = (ul >> 1) & 1u, as5 = ul & 1u;
    const u32 nb0 = (u32)as0 - 1u, nb1 = (u32)as1 - 1u, nb2 = (u32)as2 - 1u, nb3 = (u32)as3 - 1u, nb4 = (u32)as4 - 1u, nb5 = (u32)as5 - 1u;
    const u32 ix0 = 1u, ix1 = 2u | (ul >> 5), ix2 = 4u | (ul >> 4), ix3 = 8u | (ul >> 3), ix4 = 16u | (ul >> 2), ix5 = 32u | (ul >> 1);
    const u32 ix6 = 64u | ul, ix7 = 128u | (ul << 1);
    u32 low_u = 0, range_u = 0xFFFFFFFFu, code = 0, c1 = 0;
    u32 ip = 0, ibase = 0;
    u32 window = (ibase + lane < in_size) ? in[ibase + lane] : 0xFFFFFFFFu;
    // (loaded unconditionally at a clamped address and masked when it becomes the window: see the encoder's model wave)
    const u8 * __restrict__ in_pf = in_size ? in : reinterpret_cast<const u8 *>(jobs);  // (no coded bytes at all: the clamped address must still be memory of ours)
    u32 window_raw = in_pf[64u + (u32)lane < in_size ? 64u + (u32)lane : (in_size ? in_size - 1u : 0u)];
    for (int j = 0; j < 4; j++) {  // :438-441; bytes past the end read as -1 (:345)
        u32 b;
        CM_NEXT_BYTE(b);
        code = (code << 8) + b;
    }
    u64 prof_wait = 0, prof_walk = 0, prof_slow = 0, prof_miss = 0, prof_wait_miss = 0;  // PROF
    u64 t0 = 0, t1 = 0;
    u32 P0, P1, P2, P3, P4, P5, P6, P7a, P7b;  // this lane's slice of the table of the byte being decoded
    u32 vi = cm_opaque_zero();                 // the byte index as a vector register: the offset of the byte's store
#define CM_SYNC_FETCH(BUF)                                                                            \
    do {                                                                                              \
        const u32 * __restrict__ pt_ = ptab[BUF];                                                     \
        P0 = pt_[ix0]; P1 = pt_[ix1]; P2 = pt_[ix2]; P3 = pt_[ix3]; P4 = pt_[ix4]; P5 = pt_[ix5];     \
        P6 = pt_[ix6]; P7a = pt_[ix7]; P7b = pt_[ix7 + 1u];                                           \
    } while (0)
// one speculated level of the fast walk: (range, d) along the lane's assumed bit, no comparison (see above)
#define CM_WALK_SPEC(P, NB)                                                                           \
    do {                                                                                              \
        const u32 keep_ = range & (NB);                                                               \
        const u32 t_ = (u32)(((u64)range * (P)) >> 32);            /* (range * p18) >> 18, :464 */    \
        range = cm_xad(t_, (NB), keep_);                           /* t  |  range - t - 1 */          \
        d = cm_xad(t_ | ~(NB), 0xFFFFFFFFu, d);                    /* d  |  d - t - 1     */          \
    } while (0)
// one decoded level: the lane takes the bit it decodes and shifts it into cb
#define CM_WALK_REAL(P, BIT)                                                                          \
    do {                                                                                              \
        const u32 t_ = (u32)(((u64)range * (P)) >> 32);                                               \
        BIT = d <= t_;                                                                                \
        cb = cm_shift_in(cb, __ballot(BIT), BIT);                                                     \
        range = BIT ? t_ : range + ~t_;                            /* t  |  range - t - 1 */          \
        d = BIT ? d : d + ~t_;                                                                        \
    } while (0)
    // Decodes byte i from the table in P0..P7b; returns it.
    auto walk = [&]() __attribute__((always_inline)) -> u32 {
        u32 range = range_u, low;  // per-lane copies of the wave-uniform coder state
        u32 d = code - low_u;
        const bool inside = d <= range_u;  // low <= code <= high: always, unless a truncated stream fed -1 bytes (:345)
        CM_WALK_SPEC(P0, nb0);  // :453-489
        CM_WALK_SPEC(P1, nb1);
        const u64 ok1 = __ballot(d <= range);
        CM_WALK_SPEC(P2, nb2);
        CM_WALK_SPEC(P3, nb3);
        const u64 ok3 = __ballot(d <= range);
        CM_WALK_SPEC(P4, nb4);
        CM_WALK_SPEC(P5, nb5);
        // the lane that decoded the six bits it had assumed.  (Tested every other level: a lane on an improbable wrong path often
        // reaches range 0 within two levels and wraps at the next 0 it assumes -- with one test at the end 14 % of the bytes of
        // text had a second "survivor" -- but between two tests it has no time for both.)
        const u64 ok = ok1 & ok3 & __ballot(d <= range);
        u32 cb = 0;
        bool bit6, bit7;
        CM_WALK_REAL(P6, bit6);
        const u32 P7f = bit6 ? P7b : P7a;
        CM_WALK_REAL(P7f, bit7);
        const int w = __ffsll((unsigned long long)ok) - 1;
        const u32 low_f = code - cm_readlane(d, w), range_f = cm_readlane(range, w);
        // The fast result is committed unconditionally and the checked walk is a plain `if` without an `else`: with two arms the
        // compiler routes every loop-carried value (code, the input window, ...) through copies on BOTH arms.
        const u32 low_old = low_u, range_old = range_u;
        u32 c = ((u32)w << 2) | cm_readlane(cb, w);
        low_u = low_f;
        range_u = range_f;
        const bool good = inside & ((ok & (ok - 1ull)) == 0ull) & (ok != 0ull) & ((low_f ^ (low_f + range_f)) >= (1u << 24));
        if (__builtin_expect(!good, 0)) {  // a renormalisation was due on the true path: decode this byte again, checking every level
            if (PROF) prof_slow++;
            low = low_old;
            range = range_old;
            u64 valid = ~0ull;
            CM_SPEC_LEVEL(P0, nb0, 0xFFFFFFFF00000000ull);
            CM_SPEC_LEVEL(P1, nb1, 0xFFFF0000FFFF0000ull);
            CM_SPEC_LEVEL(P2, nb2, 0xFF00FF00FF00FF00ull);
            CM_SPEC_LEVEL(P3, nb3, 0xF0F0F0F0F0F0F0F0ull);
            CM_SPEC_LEVEL(P4, nb4, 0xCCCCCCCCCCCCCCCCull);
            CM_SPEC_LEVEL(P5, nb5, 0xAAAAAAAAAAAAAAAAull);
            CM_REAL_LEVEL(P6, bit6);
            const u32 P7 = bit6 ? P7b : P7a;
            CM_REAL_LEVEL(P7, bit7);
            const int w2 = __ffsll((unsigned long long)valid) - 1;  // exactly one lane survives
            low_u = cm_readlane(low, w2);
            range_u = cm_readlane(range, w2);
            c = cm_readlane((ul << 2) | ((u32)bit6 << 1) | (u32)bit7, w2);
        }
        return c;
    };
    // What follows the walk of byte i (BUF = i & 1, a compile-time constant): hand the byte over, meet the model waves, fetch the next
    // table.  Returns false when the loop ends (last byte, or the block was given up).
    auto after = [&](const u32 i, const u32 c, auto buf_tag) __attribute__((always_inline)) -> bool {
        constexpr u32 BUF = decltype(buf_tag)::value;
        LDS_POKE(s_done[BUF], c);  // every lane stores the same word: no EXEC juggling on the critical path
        if (PROF) t1 = cm_clock();
        out[vi] = (u8)c;
        vi++;
        const bool hit = c == c1;  // the models' guess for byte i was byte i-1 (0 before the block starts)
        c1 = c;
        if (i + 1u == n) return false;
        __syncthreads();  // barrier 1: the speculative table of byte i+1 is complete, the models read byte i
        if (!hit) {
            if (PROF) prof_miss++;
            __syncthreads();  // barrier 2: the corrected table
            if (R && LDS_PEEK(s_abort) != 0u) return false;  // given up (R > 0): the block is decoded again by the full-model kernel
        }
        CM_SYNC_FETCH(BUF ^ 1u);
        if (PROF) {
            const u64 t2 = cm_clock();
            prof_walk += t1 - t0;
            prof_wait += t2 - t1;
            if (!hit) prof_wait_miss += t2 - t1;
        }
        return true;
    };
    __syncthreads();  // barrier 0
    CM_SYNC_FETCH(0);
    u32 i = 0;
    for (;;) {
        if (PROF) t0 = cm_clock();
        const u32 ca = walk();
        if (!after(i, ca, CmConst<0>{})) break;
        i++;
        if (PROF) t0 = cm_clock();
        const u32 cb2 = walk();
        if (!after(i, cb2, CmConst<1>{})) break;
        i++;
    }
    if (PROF && n >= 256 && lane == 0) {  // profiling only: output bytes 0..31 become counters
        u64 * o = reinterpret_cast<u64 *>(out);
        o[0] = prof_wait;
        o[1] = prof_walk;
        o[2] = prof_slow;
        o[3] = prof_miss;
        o[4] = prof_wait_miss;  // the part of prof_wait spent behind wrong guesses
        reinterpret_cast<u32 *>(out)[22] = hw_id;
        reinterpret_cast<u32 *>(out)[23] = xcc_id;
    }
#undef CM_SYNC_FETCH
#undef CM_WALK_SPEC
#undef CM_WALK_REAL
}

#undef CM_NEXT_BYTE
#undef CM_RENORM
#undef CM_SPEC_LEVEL
#undef CM_REAL_LEVEL

template <int R, bool PROF = false>
__device__ __forceinline__ void cm_decode_sync_entry(const CmDecodeJob * __restrict__ jobs) {
    BZ3_DYN_SMEM(dyn_lds);
    cm_decode_block_sync<R, PROF>(jobs, *reinterpret_cast<CmLdsT<R> *>(dyn_lds));
}
__global__ void __launch_bounds__(320) k_cm_decode_sync(const CmDecodeJob * __restrict__ jobs) { cm_decode_sync_entry<0>(jobs); }
__global__ void __launch_bounds__(320) k_cm_decode_sync2(const CmDecodeJob * __restrict__ jobs) { cm_decode_sync_entry<CM_ROWS_DEC>(jobs); }
__global__ void __launch_bounds__(320) k_cm_decode_sync3(const CmDecodeJob * __restrict__ jobs) { cm_decode_sync_entry<CM_ROWS3_DEC>(jobs); }
// the same kernels with the walker's cycle counters (tools/cm_coresidency.py --cycles)
__global__ void __launch_bounds__(320) k_cm_decode_sync_prof(const CmDecodeJob * __restrict__ jobs) { cm_decode_sync_entry<0, true>(jobs); }
__global__ void __launch_bounds__(320) k_cm_decode_sync2_prof(const CmDecodeJob * __restrict__ jobs) { cm_decode_sync_entry<CM_ROWS_DEC, true>(jobs); }
__global__ void __launch_bounds__(320) k_cm_decode_sync3_prof(const CmDecodeJob * __restrict__ jobs) { cm_decode_sync_entry<CM_ROWS3_DEC, true>(jobs); }
#ifdef BZ3_EMU
__global__ void __launch_bounds__(320) k_cm_decode_sync_test(const CmDecodeJob * __restrict__ jobs) { cm_decode_sync_entry<CM_ROWS_TEST>(jobs); }
#endif
void cm_encode_batch(const CmEncodeJob * d_jobs, u32 njobs, hipStream_t s, int variant) {
    if (!njobs) return;
#ifdef BZ3_EMU
    if (variant == CM_VARIANT_ROWS_TEST) return launch(k_cm_encode_rows_test, dim3(njobs), dim3(128), 0, s, d_jobs);
#endif
    if (variant == CM_VARIANT_ROWS3) launch(k_cm_encode_rows3, dim3(njobs), dim3(128), 0, s, d_jobs);
    else if (variant == CM_VARIANT_ROWS) launch(k_cm_encode_rows, dim3(njobs), dim3(128), 0, s, d_jobs);
    else launch(k_cm_encode, dim3(njobs), dim3(128), 0, s, d_jobs);
}

void cm_decode_batch(const CmDecodeJob * d_jobs, u32 njobs, hipStream_t s, int variant, bool prof) {
    if (!njobs) return;
#ifdef BZ3_EMU
    if (variant == CM_VARIANT_ROWS_TEST) return launch(k_cm_decode_sync_test, dim3(njobs), dim3(320), sizeof(CmLdsT<CM_ROWS_TEST>), s, d_jobs);
#else
    {  // dynamic LDS beyond 64 KB has to be asked for, once per kernel AND per device (a batch may span several GPUs of one process)
        static std::atomic<u64> prepared[4] = {{0}, {0}, {0}, {0}};  // one bit per device ordinal (up to 256)
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        const u64 bit = 1ull << (dev & 63);
        std::atomic<u64> & word = prepared[(dev >> 6) & 3];
        if (!(word.load() & bit)) {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cm_decode_sync), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(CmLdsT<0>)));
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cm_decode_sync2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(CmLdsT<CM_ROWS_DEC>)));
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cm_decode_sync_prof), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(CmLdsT<0>)));
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cm_decode_sync2_prof), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(CmLdsT<CM_ROWS_DEC>)));
            word.fetch_or(bit);
        }
    }
#endif
    if (variant == CM_VARIANT_ROWS3) launch(prof ? k_cm_decode_sync3_prof : k_cm_decode_sync3, dim3(njobs), dim3(320), sizeof(CmLdsT<CM_ROWS3_DEC>), s, d_jobs);
    else if (variant == CM_VARIANT_ROWS) launch(prof ? k_cm_decode_sync2_prof : k_cm_decode_sync2, dim3(njobs), dim3(320), sizeof(CmLdsT<CM_ROWS_DEC>), s, d_jobs);
    else launch(prof ? k_cm_decode_sync_prof : k_cm_decode_sync, dim3(njobs), dim3(320), sizeof(CmLdsT<0>), s, d_jobs);
}

}  // namespace bz3
