// lzp.hip -- bzip3's LZP (Lempel-Ziv prediction) byte filter, encode and decode, on gfx950.
// Replaces lzp_compress -> lzp_encode_block (reference src/libbz3.c:243-249, :124-198) and
// lzp_decompress -> lzp_decode_block (:251-257, :200-241): sequential hash automata.
//
// Exact reformulation.  The reference's table slot for position p holds the
// last VISITED position with the same 18-bit hash of its 4 preceding bytes; positions inside a taken
// match are the only ones not visited.  So:
//   1. prev[p] = previous position with the same hash, ALL positions assumed visited.  Grid-parallel:
//      one stable radix sort of (hash, position) with sort.hip, neighbours in sorted order.
//   2. Grid-parallel "static" pass: every position runs the reference's 8-byte pre-test against prev[p]
//      and the positions that pass are flagged in a bitmap (0.02-0.5 % of a text block).
//   3. A single workgroup (the "driver") visits only flagged positions, 131072 positions per bitmap
//      window: lanes resolve the true candidate of each flagged position (prev chain, skipping positions
//      already swallowed by matches), re-run the pre-test and record a 48-bit byte-equality mask.  One
//      lane replays the reference's `heur` chain over these events in order using only the masks; the
//      first event that reaches 40 matching bytes is a real match: the whole workgroup measures its
//      length, marks the swallowed positions and flags next[s] of every swallowed s (the only positions
//      whose candidate just changed), then the scan resumes behind the match.  Driver work is
//      proportional to the number of events and matches, not to the block size.
//   4. Emission is grid-parallel again: per-position output counts (literal / escaped 0xF2 literal /
//      match token), device-wide scan, scatter.
// Decode mirrors it: literals between two 0xF2 bytes are copied and inserted into the table in bulk
// (scatter-max), every 0xF2 is a sequencing point resolved against the table.
#include "prims.hpp"
#include "sort.hpp"
#include "stages.hpp"

namespace bz3 {

constexpr int LZ_HASH_BITS = 18;   // LZP_DICTIONARY, :84
constexpr int LZ_MIN = 40;         // LZP_MIN_MATCH, :85
constexpr u8 LZ_ESC = 0xF2;        // MATCH, :87
constexpr int LZ_DRV = 1024;       // driver workgroup size
constexpr int LZ_BLOCK = 256;
constexpr int LZ_ITEMS = 16;
constexpr int LZ_ETILE = LZ_BLOCK * LZ_ITEMS;

__device__ __forceinline__ u32 lz_hash(u32 ctx) { return ((ctx >> 15) ^ ctx ^ (ctx >> 3)) & ((1u << LZ_HASH_BITS) - 1u); }
__device__ __forceinline__ bool lz_eq4(const u8 * __restrict__ a, const u8 * __restrict__ b) { return load_u32_any(a) == load_u32_any(b); }
__device__ __forceinline__ bool lz_skipped(const u32 * __restrict__ skip, u32 p) { return (skip[p >> 5] >> (p & 31)) & 1u; }

// ---- 1. predecessor / successor links -------------------------------------------------------------
__global__ void __launch_bounds__(LZ_BLOCK) k_lzp_hash(const u8 * __restrict__ in, u32 n, u32 * __restrict__ keys) {
    const u32 k = blockIdx.x * LZ_BLOCK + threadIdx.x;  // position p = k + 4
    if (k + 4 < n) keys[k] = lz_hash(load_be32(in + k));
}

// The two links of a position live side by side: link[2p] = prev[p], link[2p + 1] = next[p].  Position p of the sorted list lands
// at a random place, so the scatter is bound by the number of memory transactions, not by bytes: one 8-byte store per position
// instead of two 4-byte stores into two arrays (round 2 measured 21 ms per 256 MiB block for the two-array form, 25 G stores/s).
__device__ __forceinline__ size_t lz_lk(u32 p) { return 2 * (size_t)p; }

__global__ void __launch_bounds__(LZ_BLOCK) k_lzp_links(const u32 * __restrict__ keys, const u32 * __restrict__ vals, u32 m, u32 * __restrict__ link) {
    const u32 k = blockIdx.x * LZ_BLOCK + threadIdx.x;
    if (k >= m) return;
    // all six loads are requested before the first compare (clamped indices; `cond ? vals[k - 1] : 0` would be a branch around a load
    // and one exposed round trip per link, see sort.hip)
    const u32 km = k > 0 ? k - 1 : 0u, kp = k + 1 < m ? k + 1 : k;
    const u32 key = keys[k], key_m = keys[km], key_p = keys[kp];
    const u32 p = vals[k] + 4, val_m = vals[km], val_p = vals[kp];
    uint2 l;
    l.x = (k > 0 && key_m == key) ? val_m + 4 : 0u;
    l.y = (k + 1 < m && key_p == key) ? val_p + 4 : 0u;
    *reinterpret_cast<uint2 *>(link + lz_lk(p)) = l;  // the array is 8-byte aligned (Arena::take)
}

// ---- 1b. the same links through POSITION BINS (round 5) ------------------------------------------------------------------------
// k_lzp_links above costs 11.5 ms per 256 MiB block: 268 M scattered 8-byte stores all over a 2 GB array, every one a partial line that
// HBM reads, merges and writes back.  The suffix sorter's ISA scatter (bwt.hip k_isa_scatter) met the same wall and went round it: first
// bucket the records by the top bits of their destination with one pass of the radix scatter (coalesced), then scatter bucket by bucket
// -- a bucket's destinations lie in a window of a few MB, and with a contiguous range of tiles per XCD that XCD's L2 absorbs the random
// stores and writes whole lines back.  Here: k_lzp_bin_scatter walks the hash-sorted (key, position) list, forms (p, prev, next) of every
// element from its two neighbours and delivers the 12-byte record to the bin of p's top 9 bits; k_lzp_links_local then stores
// link[2p] = (prev, next) bin by bin.  Bytes per position: 4 (bin histogram) + 8 + 12 (binning) + 12 + 8 (local scatter) = 44,
// all but the last 8 streamed.
struct LzLinkRec {
    u32 p, prev, next;
};
constexpr int LZB_BITS = 9;
constexpr int LZB_RADIX = 1 << LZB_BITS;
constexpr int LZB_WAVES = RS_BLOCK / WAVE;
constexpr int LZB_ROUNDS = RS_TILE / RS_BLOCK;
constexpr int LZB_SPAN = RS_TILE / LZB_WAVES;
constexpr int LZB_DPT = LZB_RADIX / RS_BLOCK;

__device__ __forceinline__ u32 lzb_tile_of_block(u32 b, u32 tiles) {  // a contiguous range of tiles per XCD (sort.hip rs_tile_of_block)
    const u32 per = (tiles + 7u) / 8u;
    return (b & 7u) * per + (b >> 3);
}

// keys / vals: the hash-sorted list (vals[k] = position - 4); offs: the scanned digit-major table of k_rs_hist<u32, 9>(vals, shift).
__global__ void __launch_bounds__(RS_BLOCK) k_lzp_bin_scatter(const u32 * __restrict__ keys, const u32 * __restrict__ vals, u32 m, int shift,
                                                             const u32 * __restrict__ offs, u32 tiles, LzLinkRec * __restrict__ out) {
    __shared__ u32 cnt[LZB_WAVES][LZB_RADIX];
    __shared__ u32 dstart[LZB_RADIX];
    __shared__ u32 gdelta[LZB_RADIX];
    __shared__ u32 scan_lds[RS_BLOCK / WAVE + 1];
    // the tile's keys and values with one element of halo on either side, then (the same LDS) its records in bin order
    __shared__ __attribute__((aligned(16))) u32 stage[3 * RS_TILE + 8];
    u32 * const in_k = stage;                    // [RS_TILE + 2]
    u32 * const in_v = stage + RS_TILE + 4;      // [RS_TILE + 2]
    const u32 tile = lzb_tile_of_block(blockIdx.x, tiles);
    if (tile >= tiles) return;
    const int w = wave_id(), l = lane_id();
#pragma unroll
    for (int k = 0; k < LZB_WAVES; k++)
#pragma unroll
        for (int d = threadIdx.x; d < LZB_RADIX; d += RS_BLOCK) cnt[k][d] = 0;
    const u64 tile_base = (u64)tile * RS_TILE;
    const u64 wbase = tile_base + (u64)w * LZB_SPAN + l;
    const u64 last = (u64)m - 1;
    u32 key[LZB_ROUNDS], val[LZB_ROUNDS];
#pragma unroll
    for (int r = 0; r < LZB_ROUNDS; r++) {  // all loads in flight before the first use (cf. sort.hip)
        const u64 i = wbase + (u64)r * WAVE;
        key[r] = keys[i < m ? i : last];
        val[r] = vals[i < m ? i : last];
    }
    if (threadIdx.x < 2u) {  // halo: the element before the tile / behind it (a key no hash has where there is none)
        const bool left = threadIdx.x == 0u;
        const u64 i = left ? tile_base - 1 : tile_base + RS_TILE;
        const bool there = left ? tile_base > 0 : i < m;
        in_k[left ? 0 : RS_TILE + 1] = there ? keys[i] : 0xFFFFFFFFu;
        in_v[left ? 0 : RS_TILE + 1] = there ? vals[i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < LZB_ROUNDS; r++) {
        const u32 j = (u32)w * LZB_SPAN + (u32)r * WAVE + (u32)l;
        const bool valid = wbase + (u64)r * WAVE < m;
        in_k[1 + j] = valid ? key[r] : 0xFFFFFFFFu;
        in_v[1 + j] = val[r];
    }
    __syncthreads();
    u32 prev[LZB_ROUNDS], next[LZB_ROUNDS], local[LZB_ROUNDS];
#pragma unroll
    for (int r = 0; r < LZB_ROUNDS; r++) {
        const u32 j = (u32)w * LZB_SPAN + (u32)r * WAVE + (u32)l;
        prev[r] = in_k[j] == key[r] ? in_v[j] + 4u : 0u;          // (as k_lzp_links)
        next[r] = in_k[j + 2] == key[r] ? in_v[j + 2] + 4u : 0u;
    }
    const u64 lt = lanemask_lt();
#pragma unroll
    for (int r = 0; r < LZB_ROUNDS; r++) {
        const bool valid = wbase + (u64)r * WAVE < m;
        const u32 d = (val[r] >> shift) & (LZB_RADIX - 1u);
        u64 peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < LZB_BITS; b++) {
            const bool bit = (d >> b) & 1u;
            const u64 bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const u32 below = (u32)__popcll(peers & lt);
        const u32 pre = valid ? cnt[w][d] : 0u;
        wave_sync();
        if (valid && below == 0) cnt[w][d] = pre + (u32)__popcll(peers);
        wave_sync();
        local[r] = pre + below;
    }
    __syncthreads();  // (also: every neighbour read of the staged tile is done -- the records overwrite it below)
    {
        u32 run[LZB_DPT];
        u32 mine = 0;
#pragma unroll
        for (int j = 0; j < LZB_DPT; j++) {
            const u32 d = LZB_DPT * threadIdx.x + j;
            u32 acc = 0;
#pragma unroll
            for (int k = 0; k < LZB_WAVES; k++) {
                const u32 c = cnt[k][d];
                cnt[k][d] = acc;
                acc += c;
            }
            run[j] = acc;
            mine += acc;
        }
        u32 total;
        u32 start = block_excl_add<RS_BLOCK>(mine, scan_lds, total);
#pragma unroll
        for (int j = 0; j < LZB_DPT; j++) {
            const u32 d = LZB_DPT * threadIdx.x + j;
            dstart[d] = start;
            gdelta[d] = offs[(u64)d * tiles + tile] - start;
            start += run[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < LZB_ROUNDS; r++) {
        const u32 d = (val[r] >> shift) & (LZB_RADIX - 1u);
        local[r] += dstart[d] + cnt[w][d];
    }
#pragma unroll
    for (int r = 0; r < LZB_ROUNDS; r++) {
        if (wbase + (u64)r * WAVE < m) {
            stage[3u * local[r]] = val[r] + 4u;
            stage[3u * local[r] + 1u] = prev[r];
            stage[3u * local[r] + 2u] = next[r];
        }
    }
    __syncthreads();
    const u64 left = (u64)m - tile_base;
    const u32 count = left < (u64)RS_TILE ? (u32)left : (u32)RS_TILE;
#pragma unroll
    for (int q = 0; q < LZB_ROUNDS; q++) {
        const u32 j = (u32)q * RS_BLOCK + threadIdx.x;
        if (j < count) {
            LzLinkRec rec{stage[3u * j], stage[3u * j + 1u], stage[3u * j + 2u]};
            out[gdelta[((rec.p - 4u) >> shift) & (LZB_RADIX - 1u)] + j] = rec;
        }
    }
}

// link[2p] = (prev, next) from the binned records: a contiguous range of tiles per XCD, so that the random 8-byte stores of the bin an
// XCD is working on meet in that XCD's L2 (cf. bwt.hip k_isa_scatter).
constexpr int LZL_ITEMS = 8;
__global__ void __launch_bounds__(LZ_BLOCK) k_lzp_links_local(const LzLinkRec * __restrict__ rec, u32 m, u32 tiles, u32 * __restrict__ link) {
    const u32 tile = lzb_tile_of_block(blockIdx.x, tiles);
    if (tile >= tiles) return;
    const u64 base = (u64)tile * (LZ_BLOCK * LZL_ITEMS) + threadIdx.x;
    LzLinkRec x[LZL_ITEMS];
#pragma unroll
    for (int k = 0; k < LZL_ITEMS; k++) {
        const u64 i = base + (u64)k * LZ_BLOCK;
        x[k] = rec[i < m ? i : (u64)m - 1];
    }
#pragma unroll
    for (int k = 0; k < LZL_ITEMS; k++)
        if (base + (u64)k * LZ_BLOCK < m) *reinterpret_cast<uint2 *>(link + lz_lk(x[k].p)) = make_uint2(x[k].prev, x[k].next);
}

// Nearest visited ancestor of p in its hash chain (0 = none).  Swallowed positions are never un-swallowed, so
// every link of the walked path may be short-cut to the answer (path compression): without it a phrase that
// recurs k times costs O(k) per later occurrence, because all but its first copy sit inside earlier matches.
// Concurrent walkers only ever write the same value into the same link, so the race is benign.
__device__ __forceinline__ u32 lz_candidate(u32 * __restrict__ prev, const u32 * __restrict__ skip, u32 p) {
    u32 v = prev[lz_lk(p)];
    if (v == 0 || !lz_skipped(skip, v)) return v;
    u32 steps = 0;
    while (v != 0 && lz_skipped(skip, v)) { v = prev[lz_lk(v)]; steps++; }
    if (steps > 1) {
        u32 x = prev[lz_lk(p)];
        prev[lz_lk(p)] = v;
        while (x != v && x != 0) {
            const u32 nx = prev[lz_lk(x)];
            prev[lz_lk(x)] = v;
            x = nx;
        }
    }
    return v;
}

__device__ __forceinline__ bool lz_pretest(const u8 * __restrict__ in, u32 p, u32 v) {  // :143-144
    // all four loads are issued before the first compare (no short-circuit: one memory round trip, not four)
    const u32 a0 = load_u32_any(in + p), b0 = load_u32_any(in + v);
    const u32 a1 = load_u32_any(in + p + LZ_MIN - 4), b1 = load_u32_any(in + v + LZ_MIN - 4);
    return ((a0 ^ b0) | (a1 ^ b1)) == 0;
}

// bit b of the result = (in[p+b] == in[v+b]) for b < 48
__device__ __forceinline__ u64 lz_eqmask48(const u8 * __restrict__ in, u32 p, u32 v) {
    u32 x[12];
#pragma unroll
    for (int w = 0; w < 12; w++) x[w] = load_u32_any(in + p + 4 * w) ^ load_u32_any(in + v + 4 * w);
    u64 mask = 0;
#pragma unroll
    for (int w = 0; w < 12; w++) {
        const u32 t = x[w];
        const u32 nib = ((t & 0xFFu) == 0 ? 1u : 0u) | ((t & 0xFF00u) == 0 ? 2u : 0u) | ((t & 0xFF0000u) == 0 ? 4u : 0u) | ((t & 0xFF000000u) == 0 ? 8u : 0u);
        mask |= (u64)nib << (4 * w);
    }
    return mask;
}

// ---- 2. static events: one bit per position whose hash predecessor passes the pre-test ---------------
__global__ void __launch_bounds__(LZ_BLOCK) k_lzp_static(const u8 * __restrict__ in, u32 n, const u32 * __restrict__ prev, u32 * __restrict__ cand_bits,
                                                        u32 nwords) {
    // A wave takes 256 consecutive positions per step, four per lane: the four links first, then the sixteen pre-test loads of the
    // four positions, all in flight together (branch-free: a position without a predecessor compares with position 0 and is
    // masked) -- per position two dependent round trips, which one position per lane and step exposed in full.
    const u32 main_end = n - (LZ_MIN + 32);
    const u64 wave_global = (u64)blockIdx.x * (LZ_BLOCK / WAVE) + wave_id();
    const u64 stride = (u64)gridDim.x * (LZ_BLOCK / WAVE);
    const u64 last = (u64)n - 1;
    for (u64 base = wave_global * (4 * WAVE); base < (u64)nwords * 32; base += stride * (4 * WAVE)) {
        u32 v[4];
        bool live[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u64 p = base + (u64)q * WAVE + lane_id();
            live[q] = p >= 4 && p < main_end;
            v[q] = prev[lz_lk((u32)(p < last ? p : last))];
        }
        u32 a0[4], b0[4], a1[4], b1[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u64 p = base + (u64)q * WAVE + lane_id();
            const u32 pp = live[q] ? (u32)p : 0u, vv = live[q] ? v[q] : 0u;  // both + LZ_MIN stay inside the block
            a0[q] = load_u32_any(in + pp);
            b0[q] = load_u32_any(in + vv);
            a1[q] = load_u32_any(in + pp + LZ_MIN - 4);
            b1[q] = load_u32_any(in + vv + LZ_MIN - 4);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const bool ev = live[q] && v[q] > 0 && ((a0[q] ^ b0[q]) | (a1[q] ^ b1[q])) == 0;  // :143-144, as lz_pretest
            const u64 bits = __ballot(ev);
            if (lane_id() == 0) {
                const u32 w = (u32)((base + (u64)q * WAVE) >> 5);
                if (w < nwords) cand_bits[w] = (u32)bits;
                if (w + 1 < nwords) cand_bits[w + 1] = (u32)(bits >> 32);
            }
        }
    }
}

// ---- 3. driver ----------------------------------------------------------------------------------
constexpr int LZ_EV_CAP = 2048;              // events resolved per driver iteration
constexpr int LZ_WIN_WORDS = 4 * LZ_DRV;     // bitmap words scanned per iteration (131072 positions)



__global__ void __launch_bounds__(LZ_DRV) k_lzp_driver(const LzpDriverJob * __restrict__ jobs) {
    // one workgroup per block: a batch of blocks runs its (serial) drivers side by side
    const u8 * __restrict__ in = global_ptr<const u8>(jobs[blockIdx.x].in);
    const u32 n = jobs[blockIdx.x].n;
    u32 * __restrict__ prev = global_ptr<u32>(jobs[blockIdx.x].prev);
    const u32 * __restrict__ next = global_ptr<const u32>(jobs[blockIdx.x].next);
    u32 * __restrict__ cand_bits = global_ptr<u32>(jobs[blockIdx.x].cand_bits);
    const u32 nwords = jobs[blockIdx.x].nwords;
    u32 * __restrict__ skip = global_ptr<u32>(jobs[blockIdx.x].skip);
    u32 * __restrict__ mstart = global_ptr<u32>(jobs[blockIdx.x].mstart);
    u32 * __restrict__ mpos = global_ptr<u32>(jobs[blockIdx.x].mpos);
    u32 * __restrict__ mlen = global_ptr<u32>(jobs[blockIdx.x].mlen);
    LzDriverOut * __restrict__ result = global_ptr<LzDriverOut>(jobs[blockIdx.x].result);
    __shared__ u32 cq[LZ_EV_CAP];
    __shared__ u32 ev_ref[LZ_EV_CAP];
    __shared__ u64 ev_mask[LZ_EV_CAP];
    __shared__ u32 red[LZ_DRV / WAVE + 1];
    __shared__ u32 s_cur, s_heur, s_accept, s_len, s_matches;
    const u32 tid = threadIdx.x;
    const u32 main_end = n - (LZ_MIN + 32);  // the main loop handles positions < n - 72 (:137)
    u32 iterations = 0, evaluated = 0;
    if (tid == 0) { s_cur = 4; s_heur = 0; s_matches = 0; }
    __syncthreads();
    for (;;) {
        const u32 cur = s_cur;
        if (cur >= main_end) break;
        iterations++;
        // ---- scan one bitmap window for flagged positions >= cur ------------------------------------
        const u32 w0 = (cur >> 5) & ~3u;  // 16-byte aligned window start: one dwordx4 load per lane
        u32 word[4];
        u32 cnt = 0;
        {
            const u32 wb = w0 + 4u * tid;
            uint4 q4 = make_uint4(0u, 0u, 0u, 0u);
            if (wb + 3 < nwords) q4 = *reinterpret_cast<const uint4 *>(cand_bits + wb);
            else {
                if (wb < nwords) q4.x = cand_bits[wb];
                if (wb + 1 < nwords) q4.y = cand_bits[wb + 1];
                if (wb + 2 < nwords) q4.z = cand_bits[wb + 2];
            }
            word[0] = q4.x; word[1] = q4.y; word[2] = q4.z; word[3] = q4.w;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u32 w = wb + (u32)k;
                if (w < (cur >> 5)) word[k] = 0;
                else if (w == (cur >> 5)) word[k] &= 0xFFFFFFFFu << (cur & 31u);
                cnt += (u32)__popc(word[k]);
            }
        }
        u32 total;
        const u32 pre = block_excl_add<LZ_DRV>(cnt, red, total);
        const bool included = pre + cnt <= (u32)LZ_EV_CAP;
        // first word NOT covered this iteration, and the number of events that are covered
        const u32 cut_word = block_min<LZ_DRV>(included ? 0xFFFFFFFFu : w0 + 4u * tid, red);
        const u32 cut_events = block_min<LZ_DRV>(included ? 0xFFFFFFFFu : pre, red);
        const u32 nev = cut_events == 0xFFFFFFFFu ? total : cut_events;
        u32 window_end = cut_word == 0xFFFFFFFFu ? (w0 + (u32)LZ_WIN_WORDS) * 32u : cut_word * 32u;
        if (window_end > main_end || window_end < cur) window_end = main_end;
        if (included && cnt) {
            u32 o = pre;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                u32 b = word[k];
                const u32 pos0 = (w0 + 4u * tid + (u32)k) * 32u;
                while (b) {
                    const int bit = __ffs((int)b) - 1;
                    b &= b - 1;
                    cq[o++] = pos0 + (u32)bit;
                }
            }
        }
        __syncthreads();
        if (nev == 0) {
            if (tid == 0) s_cur = window_end;
            __syncthreads();
            continue;
        }
        // ---- resolve the flagged positions against the current skip state, a sub-batch at a time (the first
        //      event of a repeated region is usually the one that gets accepted) and replay the heur chain ----
        if (tid == 0) s_accept = 0xFFFFFFFFu;
        for (u32 b0 = 0, sub = WAVE; b0 < nev; b0 += sub, sub = LZ_DRV) {
            const u32 b1 = (nev - b0 < sub) ? nev : b0 + sub;
            evaluated += b1 - b0;
            {
                const u32 e = b0 + tid;
                if (e < b1) {
                    const u32 q = cq[e];
                    u32 v = 0;
                    if (q < main_end) {
                        v = lz_candidate(prev, skip, q);
                        if (v > 0 && !lz_pretest(in, q, v)) v = 0;
                    }
                    ev_mask[e] = v > 0 ? lz_eqmask48(in, q, v) : 0ull;
                    ev_ref[e] = v;
                }
            }
            __syncthreads();
            if (tid == 0) {  // one lane; masks only, no memory traffic
                u32 heur = s_heur;
                u32 accept = 0xFFFFFFFFu;
                for (u32 e = b0; e < b1; e++) {
                    if (ev_ref[e] == 0) continue;
                    const u32 p = cq[e];
                    const u64 mask = ev_mask[e];
                    if (heur > p && ((mask >> (heur - p)) & 0xFull) != 0xFull) continue;  // :145
                    u32 len = 4;
                    while (len < (u32)LZ_MIN && p + len < main_end && ((mask >> len) & 0xFull) == 0xFull) len += 4;  // :148-150
                    if (len < (u32)LZ_MIN) {
                        if (heur < p + len) heur = p + len;  // :152-155
                        continue;
                    }
                    accept = e;
                    break;
                }
                s_heur = heur;
                s_accept = accept;
            }
            __syncthreads();
            if (s_accept != 0xFFFFFFFFu) break;
        }
        const u32 accept = s_accept;
        if (accept == 0xFFFFFFFFu) {
            if (tid == 0) s_cur = window_end;
            __syncthreads();
            continue;
        }
        // ---- a real match: measure it with the whole workgroup ----------------------------------------
        const u32 p = cq[accept], v = ev_ref[accept];
        u32 len = LZ_MIN;
        for (;;) {
            const u32 off = len + 4u * tid;
            const bool ok = (p + off < main_end) && lz_eq4(in + p + off, in + v + off);
            const u32 first_bad = block_min<LZ_DRV>(ok ? 0xFFFFFFFFu : off, red);
            if (first_bad != 0xFFFFFFFFu) { len = first_bad; break; }
            len += 4u * LZ_DRV;
        }
        if (tid == 0) {
            len += in[p + len] == in[v + len];  // :157-159
            len += in[p + len] == in[v + len];
            len += in[p + len] == in[v + len];
            const u32 k = s_matches;
            mpos[k] = p;
            mlen[k] = len;
            mstart[p >> 5] |= 1u << (p & 31u);
            s_len = len;
            s_matches = k + 1;
            s_cur = p + len;
        }
        __syncthreads();
        len = s_len;
        {
            const u32 a = p + 1, b = p + len;  // swallowed positions [a, b)
            for (u32 w = (a >> 5) + tid; w <= ((b - 1) >> 5); w += LZ_DRV) {
                const u32 lo = w << 5;
                u32 m = 0xFFFFFFFFu;
                if (lo < a) m &= 0xFFFFFFFFu << (a - lo);
                if (lo + 32 > b) m &= 0xFFFFFFFFu >> (lo + 32 - b);
                skip[w] |= m;  // only this workgroup writes the bitmap; distinct lanes own distinct words
            }
            // the candidate of next[s] just changed for every swallowed s: flag it for re-evaluation
            for (u32 sidx = a + tid; sidx < b; sidx += LZ_DRV) {
                const u32 nq = next[lz_lk(sidx)];
                if (nq >= b && nq < main_end) atomicOr(&cand_bits[nq >> 5], 1u << (nq & 31u));
            }
        }
        __threadfence_block();
        __syncthreads();
        l1_invalidate();  // the flags above were set by atomics at L2; make the next window's plain loads see them
    }
    if (tid == 0) {
        result->end_pos = s_cur;
        result->n_matches = s_matches;
        result->iterations = iterations;
        result->evaluated = evaluated;
    }
}

// ---- 4. emission -------------------------------------------------------------------------------
__device__ __forceinline__ u32 lz_match_len(const u32 * __restrict__ mpos, const u32 * __restrict__ mlen, u32 nm, u32 p) {
    u32 lo = 0, hi = nm;  // mpos is ascending; p is known to be a match start
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (mpos[mid] <= p) lo = mid; else hi = mid;
    }
    return mlen[lo];
}

__device__ __forceinline__ u32 lz_emit_count(const u8 * __restrict__ in, u32 * __restrict__ prev, const u32 * __restrict__ skip,
                                             const u32 * __restrict__ mstart, const u32 * __restrict__ mpos, const u32 * __restrict__ mlen, u32 nm, u32 p,
                                             u32 & ml) {
    ml = 0;
    if (p < 4) return 1u;
    if (lz_skipped(skip, p)) return 0u;
    if (lz_skipped(mstart, p)) {
        ml = lz_match_len(mpos, mlen, nm, p);
        return 2u + (ml - LZ_MIN) / 254u;  // MATCH, 254 x q, remainder (:164-173)
    }
    if (in[p] != LZ_ESC) return 1u;
    return lz_candidate(prev, skip, p) > 0 ? 2u : 1u;  // escape only when the slot was live (:176-181, :194)
}

template <int MODE>
__global__ void __launch_bounds__(LZ_BLOCK) k_lzp_emit(const u8 * __restrict__ in, u32 n, u32 * __restrict__ prev, const u32 * __restrict__ skip,
                                                      const u32 * __restrict__ mstart, const u32 * __restrict__ mpos, const u32 * __restrict__ mlen,
                                                      const LzDriverOut * __restrict__ drv, u32 * __restrict__ tile_sum, u8 * __restrict__ out) {
    __shared__ u32 lds[LZ_BLOCK / WAVE + 1];
    const u32 nm = drv->n_matches;
    const u64 base = (u64)blockIdx.x * LZ_ETILE + (u64)threadIdx.x * LZ_ITEMS;
    u32 cnt[LZ_ITEMS], ml[LZ_ITEMS];
    u32 sum = 0;
#pragma unroll 4
    for (int k = 0; k < LZ_ITEMS; k++) {
        ml[k] = 0;
        cnt[k] = (base + k < n) ? lz_emit_count(in, prev, skip, mstart, mpos, mlen, nm, (u32)(base + k), ml[k]) : 0u;
        sum += cnt[k];
    }
    u32 tot;
    const u32 pre = block_excl_add<LZ_BLOCK>(sum, lds, tot);
    if (MODE == 0) {
        if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
        return;
    }
    u64 o = (u64)tile_sum[blockIdx.x] + pre;
#pragma unroll 1
    for (int k = 0; k < LZ_ITEMS; k++) {
        if (cnt[k] == 0) continue;
        const u32 p = (u32)(base + k);
        if (ml[k]) {
            u32 rest = ml[k] - LZ_MIN;
            out[o++] = LZ_ESC;
            while (rest >= 254u) { rest -= 254u; out[o++] = 254; }
            out[o++] = (u8)rest;
        } else {
            out[o++] = in[p];
            if (cnt[k] == 2) out[o++] = 255;
        }
    }
}

size_t lzp_encode_ctx_bytes(u64 n) { return n * 8 + (n / 32 + 8) * 12 + (n / LZ_MIN + 8) * 8 + 4096; }

// Phase A (asynchronous): hash links + static event bitmap.  The context (what the driver and the emission need) is
// carved from `ctx` and stays there until the caller recycles it, so several blocks can be prepared before their drivers
// run as one batch; the sort buffers are transient and come from `tmp` (the two may be the same arena).
void lzp_encode_prepare(const u8 * d_in, u32 n, LzpEncodeCtx & c, Arena & tmp, hipStream_t s) { lzp_encode_prepare(d_in, n, c, tmp, tmp, s); }

void lzp_encode_prepare(const u8 * d_in, u32 n, LzpEncodeCtx & c, Arena & ctx, Arena & tmp, hipStream_t s) {
    c.active = false;
    c.in = d_in;
    c.n = n;
    if (n < LZ_MIN + 32) return;  // :244
    c.active = true;
    const u32 m = n - 4;
    c.nwords = (n + 31) / 32 + 2;
    c.prev = ctx.take<u32>(2 * (size_t)n);  // interleaved links (k_lzp_links): prev[p] = c.prev[2p], next[p] = c.next[2p]
    c.next = c.prev + 1;
    c.skip = ctx.take<u32>(c.nwords);
    c.mstart = ctx.take<u32>(c.nwords);
    c.cand_bits = ctx.take<u32>(c.nwords);
    c.mpos = ctx.take<u32>(n / LZ_MIN + 2);
    c.mlen = ctx.take<u32>(n / LZ_MIN + 2);
    c.d_res = reinterpret_cast<LzDriverOut *>(ctx.take<u32>(8));
    {
        const size_t mk2 = tmp.mark();
        u32 * k0 = tmp.take<u32>(m);
        u32 * v0 = tmp.take<u32>(m);
        u32 * k1 = tmp.take<u32>(3 * (size_t)m + 64);  // (k1, v1) and a third buffer in one piece: the binned link records (12 bytes each) overlay them
        u32 * v1 = k1 + (((size_t)m + 63) & ~(size_t)63);
        launch(k_lzp_hash, dim3((m + LZ_BLOCK - 1) / LZ_BLOCK), dim3(LZ_BLOCK), 0, s, d_in, n, k0);
        HIP_CHECK(hipMemsetAsync(c.prev, 0, 32, s));  // positions 0..3 have no context: no links
        static const int form = [] { const char * e = getenv("BZ3_LZP_LINKS"); return e ? atoi(e) : 2; }();  // (experiments, read once) 0 = round 4's build, 1 = two 9-bit passes, 2 = + binned links
        if (form == 0) {
            radix_pass<u32>(k0, k1, (const u32 *)nullptr, v1, m, 0, 0xFFFFFFFFu, 0u, tmp, s);
            radix_pass<u32>(k1, k0, (const u32 *)v1, v0, m, 8, 0xFFFFFFFFu, 0u, tmp, s);
            radix_pass<u32>(k0, k1, (const u32 *)v0, v1, m, 16, 0xFFFFFFFFu, 0u, tmp, s);
            launch(k_lzp_links, dim3((m + LZ_BLOCK - 1) / LZ_BLOCK), dim3(LZ_BLOCK), 0, s, (const u32 *)k1, (const u32 *)v1, m, c.prev);
        } else {
            // the 18-bit hashes in TWO stable passes of 9-bit digits (rounds 1-4: three of 8 bits, the third for two bits)
            static_assert(LZ_HASH_BITS == 2 * LZB_BITS, "two digits cover the hash");
            radix_pass_bits<u32, LZB_BITS>(k0, k1, (const u32 *)nullptr, v1, m, 0, 0xFFFFFFFFu, 0u, tmp, s);
            radix_pass_bits<u32, LZB_BITS>(k1, k0, (const u32 *)v1, v0, m, LZB_BITS, 0xFFFFFFFFu, 0u, tmp, s);
            if (form == 1 || m <= 2u * RS_TILE) {  // (a tile or two: nothing to bin)
                launch(k_lzp_links, dim3((m + LZ_BLOCK - 1) / LZ_BLOCK), dim3(LZ_BLOCK), 0, s, (const u32 *)k0, (const u32 *)v0, m, c.prev);
            } else {
                // records binned by the top 9 bits of the position, over (k1, v1) and the third buffer, then the local scatter
                LzLinkRec * rec = reinterpret_cast<LzLinkRec *>(k1);
                static_assert(sizeof(LzLinkRec) == 12, "three words");
                int nbits = 0;
                for (u32 x = m - 1; x; x >>= 1) nbits++;
                const int shift = nbits > LZB_BITS ? nbits - LZB_BITS : 0;
                const u32 tiles = (m + RS_TILE - 1) / RS_TILE;
                const size_t mk3 = tmp.mark();
                u32 * hist = tmp.take<u32>((size_t)tiles * LZB_RADIX);
                const dim3 sgrid(8u * ((tiles + 7u) / 8u));
                radix_hist_bits9(v0, m, shift, hist, tiles, s);
                exclusive_scan_u32(hist, (u64)tiles * LZB_RADIX, nullptr, tmp, s);
                launch(k_lzp_bin_scatter, sgrid, dim3(RS_BLOCK), 0, s, (const u32 *)k0, (const u32 *)v0, m, shift, (const u32 *)hist, tiles, rec);
                const u32 ltiles = (m + LZ_BLOCK * LZL_ITEMS - 1) / (LZ_BLOCK * LZL_ITEMS);
                launch(k_lzp_links_local, dim3(8u * ((ltiles + 7u) / 8u)), dim3(LZ_BLOCK), 0, s, (const LzLinkRec *)rec, m, ltiles, c.prev);
                tmp.release(mk3);
            }
        }
        tmp.release(mk2);  // the sort buffers are dead once the links are written (stream order protects them)
    }
    HIP_CHECK(hipMemsetAsync(c.skip, 0, (size_t)c.nwords * 4, s));
    HIP_CHECK(hipMemsetAsync(c.mstart, 0, (size_t)c.nwords * 4, s));
    u32 grid = (c.nwords * 32u / (4 * WAVE) + (LZ_BLOCK / WAVE)) / (LZ_BLOCK / WAVE);  // a wave per 256 positions (k_lzp_static)
    if (grid > 8192) grid = 8192;
    launch(k_lzp_static, dim3(grid), dim3(LZ_BLOCK), 0, s, d_in, n, (const u32 *)c.prev, c.cand_bits, c.nwords);
}

LzpDriverJob lzp_driver_job(const LzpEncodeCtx & c) {
    return LzpDriverJob{dev_addr(c.in), dev_addr(c.prev), dev_addr(c.next), dev_addr(c.cand_bits), dev_addr(c.skip), dev_addr(c.mstart), dev_addr(c.mpos),
                        dev_addr(c.mlen), dev_addr(c.d_res), c.n, c.nwords};
}

// Phase B (asynchronous): the serial drivers of a batch of blocks, one workgroup each.
void lzp_driver_batch(const LzpDriverJob * h_jobs, LzpDriverJob * d_jobs, u32 njobs, hipStream_t s) {
    if (!njobs) return;
    HIP_CHECK(hipMemcpyAsync(d_jobs, h_jobs, sizeof(LzpDriverJob) * njobs, hipMemcpyHostToDevice, s));
    launch(k_lzp_driver, dim3(njobs), dim3(LZ_DRV), 0, s, (const LzpDriverJob *)d_jobs);
}

// Phase C (synchronous): sizes the output, gives up like the reference when it would not shrink, emits.
s32 lzp_encode_finish(const LzpEncodeCtx & c, u8 * d_out, Arena & tmp, hipStream_t s) {
    if (!c.active) return -1;
    const size_t mk = tmp.mark();
    const u32 n = c.n;
    const u32 tiles = (n + LZ_ETILE - 1) / LZ_ETILE;
    u32 * tile_sum = tmp.take<u32>(tiles + 1);
    u32 * d_total = tmp.take<u32>(1);
    launch(k_lzp_emit<0>, dim3(tiles), dim3(LZ_BLOCK), 0, s, c.in, n, c.prev, (const u32 *)c.skip, (const u32 *)c.mstart, (const u32 *)c.mpos, (const u32 *)c.mlen,
           (const LzDriverOut *)c.d_res, tile_sum, (u8 *)nullptr);
    exclusive_scan_u32(tile_sum, tiles, d_total, tmp, s);
    u32 total = 0;
    HIP_CHECK(hipMemcpyAsync(&total, d_total, 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    static const bool trace = getenv("BZ3_HIP_TRACE") != nullptr;  // (read once)
    if (trace) {
        LzDriverOut h;
        HIP_CHECK(hipMemcpy(&h, c.d_res, sizeof h, hipMemcpyDeviceToHost));
        fprintf(stderr, "[bz3 lzp] n=%u matches=%u driver_iterations=%u evaluated=%u out=%u\n", n, h.n_matches, h.iterations, h.evaluated, total);
    }
    s32 result = -1;
    if (total < n - 8) {  // the reference gives up once the output reaches n - 8 bytes (:128, :197)
        launch(k_lzp_emit<1>, dim3(tiles), dim3(LZ_BLOCK), 0, s, c.in, n, c.prev, (const u32 *)c.skip, (const u32 *)c.mstart, (const u32 *)c.mpos,
               (const u32 *)c.mlen, (const LzDriverOut *)c.d_res, tile_sum, d_out);
        // (no wait here since round 6: whatever reads d_out or reuses the scratch is launched on this stream behind the emission; rounds 1-5 synchronised once more per block)
        result = (s32)total;
    }
    tmp.release(mk);
    return result;
}

// One block, start to finish (stage hook).
s32 lzp_encode(const u8 * d_in, u32 n, u8 * d_out, Arena & tmp, hipStream_t s) {
    const size_t mk = tmp.mark();
    LzpEncodeCtx c;
    lzp_encode_prepare(d_in, n, c, tmp, s);
    s32 r = -1;
    if (c.active) {
        LzpDriverJob job = lzp_driver_job(c);
        LzpDriverJob * d_job = tmp.take<LzpDriverJob>(1);
        lzp_driver_batch(&job, d_job, 1, s);
        r = lzp_encode_finish(c, d_out, tmp, s);
    }
    tmp.release(mk);
    return r;
}

// ---- decode -------------------------------------------------------------------------------------
constexpr int LZD_CHUNK = 16384;  // input bytes staged in LDS per iteration (16 per lane)

__device__ __forceinline__ u64 lz_clock() {
#ifdef BZ3_EMU
    return 0;
#else
    return (u64)__builtin_readcyclecounter();
#endif
}
// PROF (BZ3_LZP_PROF=1, profiling only): lane 0 sums the cycles of a trip's four phases and prints them when the block is done.
template <bool PROF>
__device__ __forceinline__ void lzp_decode_block(const LzpDecodeJob * __restrict__ jobs) {
    const u8 * __restrict__ in = global_ptr<const u8>(jobs[blockIdx.x].in);
    const u32 n = jobs[blockIdx.x].n;
    u8 * __restrict__ out = global_ptr<u8>(jobs[blockIdx.x].out);
    const u32 max_out = jobs[blockIdx.x].max_out;
    u32 * __restrict__ lut = global_ptr<u32>(jobs[blockIdx.x].lut);
    s32 * __restrict__ result = global_ptr<s32>(jobs[blockIdx.x].result);
    __shared__ __attribute__((aligned(16))) u8 stage[LZD_CHUNK + 32];  // [12..15] = the 4 output bytes before the chunk, [16..] = input bytes
    __shared__ u32 red[LZ_DRV / WAVE + 1];
    __shared__ u32 s_ip, s_op, s_copy_src, s_copy_cnt, s_fail;
    const u32 tid = threadIdx.x;
    // the table starts empty (:254): zeroed here, by the workgroup that owns it -- a hipMemsetAsync per block on the launching stream was sixteen
    // more launches per window, each of which had to find free wave slots on the CUs reserved for these kernels (round 5, api.hip DeviceCtx)
    {
        uint4 * __restrict__ lz = reinterpret_cast<uint4 *>(lut);  // (256-byte aligned: Arena::take)
        for (u32 k = tid; k < (u32)(LZP_LUT_WORDS / 4); k += LZ_DRV) lz[k] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (tid < 4) out[tid] = in[tid];
    if (tid == 0) { s_ip = 4; s_op = 4; s_fail = 0; }
    __threadfence();  // the zeroes are in L2 before the first atomic of any lane gets there
    __syncthreads();
    u64 pc_stage = 0, pc_lits = 0, pc_lane0 = 0, pc_copy = 0, pt0 = 0, pt1 = 0, pt2 = 0, pt3 = 0;
    u32 pc_trips = 0, pc_matches = 0;
    for (;;) {
        if (PROF) pt0 = lz_clock();
        const u32 ip = s_ip, op = s_op;
        if (s_fail || ip >= n || op >= max_out) break;
        const u32 chunk = (n - ip > (u32)LZD_CHUNK) ? (u32)LZD_CHUNK : n - ip;
        // stage the chunk (16 consecutive bytes per lane: ONE 16-byte load at any alignment; byte by byte only in the last, ragged
        // lane of the block) and the 4 bytes of output context
        const u32 j0 = tid * 16u;
        u32 w[4] = {0u, 0u, 0u, 0u};  // bytes j0 .. j0+15, little endian
        if (j0 + 16u <= chunk) {
            const PackedU128 q = *reinterpret_cast<const PackedU128 *>(in + ip + j0);
            w[0] = q.v[0]; w[1] = q.v[1]; w[2] = q.v[2]; w[3] = q.v[3];
        } else {
            for (u32 k = 0; k < 16u; k++)
                if (j0 + k < chunk) w[k >> 2] |= (u32)in[ip + j0 + k] << (8u * (k & 3u));
        }
        if (tid < 4) stage[12 + tid] = out[op - 4 + tid];
        *reinterpret_cast<uint4 *>(&stage[16 + j0]) = make_uint4(w[0], w[1], w[2], w[3]);
        // first 0xF2 of the lane's 16 bytes: a zero byte of (word ^ 0xF2F2F2F2), lowest first
        u32 first = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            const u32 x = w[k] ^ (0x01010101u * (u32)LZ_ESC);
            const u32 z = (x - 0x01010101u) & ~x & 0x80808080u;  // bit 7 of a byte is set if that byte of x is zero (exact for the lowest such byte)
            if (z) first = j0 + 4u * (u32)k + ((u32)__ffs((int)z) - 1u) / 8u;
        }
        if (first != 0xFFFFFFFFu && first >= chunk) first = 0xFFFFFFFFu;  // (the zero padding of a ragged lane holds no 0xF2, but be exact)
        first = block_min<LZ_DRV>(first, red);  // (also orders the LDS staging)
        if (PROF) pt1 = lz_clock();
        const bool hit = first != 0xFFFFFFFFu;
        u32 lit = hit ? first : chunk;
        bool room = true;
        if (lit > max_out - op) { lit = max_out - op; room = false; }
        // bulk literals: copy out and insert every (visited) output position into the table.  The context of position j is the 4 bytes
        // before it: the lane's own bytes and the 4 bytes before them (one LDS word), as a sliding window in registers.
        if (j0 < lit) {
            u32 before = *reinterpret_cast<const u32 *>(&stage[12 + j0]);  // bytes j0-4 .. j0-1, little endian
            if (j0 + 16u <= lit) {
                PackedU128 q;
                q.v[0] = w[0]; q.v[1] = w[1]; q.v[2] = w[2]; q.v[3] = w[3];
                *reinterpret_cast<PackedU128 *>(out + op + j0) = q;
            } else {
                for (u32 k = 0; j0 + k < lit; k++) out[op + j0 + k] = (u8)(w[k >> 2] >> (8u * (k & 3u)));
            }
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const u32 j = j0 + (u32)k;
                if (j < lit) atomicMax(&lut[lz_hash(__builtin_bswap32(before))], op + j);  // (ctx = the 4 bytes, oldest in the top byte)
                before = (before >> 8) | (((w[k >> 2] >> (8 * (k & 3))) & 0xFFu) << 24);
            }
        }
        __threadfence_block();
        __syncthreads();
        if (PROF) pt2 = lz_clock();
        if (tid == 0) {
            u32 i2 = ip + lit, o2 = op + lit;
            s_copy_cnt = 0;
            if (hit && room && o2 < max_out) {
                // hash context of o2: the 4 bytes before it are stage[lit .. lit+3]
                const u32 ctx = ((u32)stage[12 + lit] << 24) | ((u32)stage[13 + lit] << 16) | ((u32)stage[14 + lit] << 8) | (u32)stage[15 + lit];
                const u32 cand = atomicExch(&lut[lz_hash(ctx)], o2);  // read the slot (at L2, where the atomics landed) and visit o2
                // the bytes behind the 0xF2 are normally part of the staged chunk: LDS instead of dependent ~1.5 us loads from HBM on the
                // block's serial path (41 k sequencing points in a 256 MiB text block)
                auto rd = [&](u32 x) -> u8 { return x - ip < chunk ? stage[16u + (x - ip)] : in[x]; };
                if (cand > 0) {
                    i2++;
                    if (i2 == n) s_fail = 1;  // :215
                    else if (rd(i2) != 255) {
                        u32 len = LZ_MIN;
                        for (;;) {  // :218-222
                            if (i2 == n) { s_fail = 1; break; }
                            const u8 b = rd(i2++);
                            len += b;
                            if (b != 254) break;
                        }
                        if (!s_fail) {
                            u64 stop = (u64)o2 + len;
                            if (stop > max_out) stop = max_out;
                            s_copy_src = cand;
                            s_copy_cnt = (u32)(stop - o2);
                        }
                    } else {
                        i2++;
                        out[o2++] = LZ_ESC;
                    }
                } else {
                    out[o2++] = rd(i2++);
                }
            }
            s_ip = i2;
            s_op = o2;
        }
        __threadfence_block();
        __syncthreads();
        if (PROF) pt3 = lz_clock();
        const u32 cnt = s_copy_cnt;
        if (cnt) {
            const u32 dst = s_op, src = s_copy_src;
            const u32 period = dst - src;  // a self-overlapping copy repeats with this period (:228)
            for (u32 j = tid; j < cnt; j += LZ_DRV) out[dst + j] = out[src + (j % period)];
            __threadfence_block();
            __syncthreads();
            if (tid == 0) s_op = dst + cnt;
            __syncthreads();
        }
        if (PROF) {
            const u64 pt4 = lz_clock();
            pc_stage += pt1 - pt0; pc_lits += pt2 - pt1; pc_lane0 += pt3 - pt2; pc_copy += pt4 - pt3;
            pc_trips++;
            pc_matches += cnt ? 1u : 0u;
        }
    }
    if (tid == 0) *result = s_fail ? -1 : (s32)s_op;
    if (PROF && tid == 0)
        printf("[bz3 lzp-decode prof] n=%u trips=%u matches=%u cycles: stage+search %llu  literals+inserts+fence %llu  lane0+fence %llu  copy %llu\n", n, pc_trips, pc_matches,
               (unsigned long long)pc_stage, (unsigned long long)pc_lits, (unsigned long long)pc_lane0, (unsigned long long)pc_copy);
}
__global__ void __launch_bounds__(LZ_DRV) k_lzp_decode(const LzpDecodeJob * __restrict__ jobs) { lzp_decode_block<false>(jobs); }
__global__ void __launch_bounds__(LZ_DRV) k_lzp_decode_prof(const LzpDecodeJob * __restrict__ jobs) { lzp_decode_block<true>(jobs); }

void lzp_decode_batch(const LzpDecodeJob * h_jobs, LzpDecodeJob * d_jobs, u32 njobs, hipStream_t s) {
    if (!njobs) return;
    HIP_CHECK(hipMemcpyAsync(d_jobs, h_jobs, sizeof(LzpDecodeJob) * njobs, hipMemcpyHostToDevice, s));
    static const bool prof = [] { const char * e = getenv("BZ3_LZP_PROF"); return e && atoi(e) != 0; }();  // (profiling only, read once)
    if (prof) launch(k_lzp_decode_prof, dim3(njobs), dim3(LZ_DRV), 0, s, (const LzpDecodeJob *)d_jobs);
    else launch(k_lzp_decode, dim3(njobs), dim3(LZ_DRV), 0, s, (const LzpDecodeJob *)d_jobs);
}

}  // namespace bz3
