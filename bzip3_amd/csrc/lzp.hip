// lzp.hip -- bzip3's LZP (Lempel-Ziv prediction) byte filter, encode and decode, on gfx950.
// Replaces lzp_compress -> lzp_encode_block (reference src/libbz3.c:243-249, :124-198) and
// lzp_decompress -> lzp_decode_block (:251-257, :200-241): sequential hash automata.
//
// Exact reformulation.  The reference's table slot for position p holds the
// last VISITED position with the same 18-bit hash of its 4 preceding bytes; positions inside a taken
// match are the only ones not visited.  So:
//   1. prev[p] = previous position with the same hash, ALL positions assumed visited.  Grid-parallel:
//      one stable radix sort of (hash, position) with sort.hip, neighbours in sorted order.
//   2. A single workgroup (the "driver") walks the block in 4096-position tiles.  Every lane resolves
//      its candidate (prev chain, skipping positions already swallowed by matches), runs the
//      reference's 8-byte pre-test and, for the few positions that pass ("events"), records a
//      48-bit byte-equality mask.  One lane then replays the `heur` chain over the events in order
//      using only those masks; the first event that reaches 40 matching bytes is a real match: the
//      whole workgroup measures its length, marks the swallowed positions, and the tile restarts
//      behind it.  Text has few events and very few matches; repetitive data has few, long matches.
//   3. Emission is grid-parallel again: per-position output counts (literal / escaped 0xF2 literal /
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
constexpr int LZ_TILE = 4096;      // positions per driver iteration
constexpr int LZ_BLOCK = 256;
constexpr int LZ_ITEMS = 16;
constexpr int LZ_ETILE = LZ_BLOCK * LZ_ITEMS;

__device__ __forceinline__ u32 lz_hash(u32 ctx) { return ((ctx >> 15) ^ ctx ^ (ctx >> 3)) & ((1u << LZ_HASH_BITS) - 1u); }
__device__ __forceinline__ bool lz_eq4(const u8 * __restrict__ a, const u8 * __restrict__ b) {
    return a[0] == b[0] && a[1] == b[1] && a[2] == b[2] && a[3] == b[3];
}
__device__ __forceinline__ bool lz_skipped(const u32 * __restrict__ skip, u32 p) { return (skip[p >> 5] >> (p & 31)) & 1u; }

// ---- 1. predecessor array ---------------------------------------------------------------------
__global__ void __launch_bounds__(LZ_BLOCK) k_lzp_hash(const u8 * __restrict__ in, u32 n, u32 * __restrict__ keys) {
    const u32 k = blockIdx.x * LZ_BLOCK + threadIdx.x;  // position p = k + 4
    if (k + 4 < n + 0u && k + 4 >= 4) keys[k] = lz_hash(load_be32(in + k));
}

__global__ void __launch_bounds__(LZ_BLOCK) k_lzp_prev(const u32 * __restrict__ keys, const u32 * __restrict__ vals, u32 m, u32 * __restrict__ prev) {
    const u32 k = blockIdx.x * LZ_BLOCK + threadIdx.x;
    if (k >= m) return;
    prev[vals[k] + 4] = (k > 0 && keys[k] == keys[k - 1]) ? vals[k - 1] + 4 : 0u;
}

// Nearest visited ancestor of p in its hash chain (0 = none).
__device__ __forceinline__ u32 lz_candidate(const u32 * __restrict__ prev, const u32 * __restrict__ skip, u32 p) {
    u32 v = prev[p];
    while (v != 0 && lz_skipped(skip, v)) v = prev[v];
    return v;
}

// ---- 2. driver ----------------------------------------------------------------------------------
struct LzDriverOut {
    u32 end_pos;    // first position handled by the tail loop (>= n - 72)
    u32 n_matches;
};

__global__ void __launch_bounds__(LZ_DRV) k_lzp_driver(const u8 * __restrict__ in, u32 n, const u32 * __restrict__ prev, u32 * __restrict__ skip,
                                                      u32 * __restrict__ mlen, LzDriverOut * __restrict__ result) {
    __shared__ u32 ev_pos[LZ_TILE];
    __shared__ u32 ev_ref[LZ_TILE];
    __shared__ u64 ev_mask[LZ_TILE];
    __shared__ u32 red[LZ_DRV / WAVE + 1];
    __shared__ u32 s_cur, s_heur, s_nev, s_accept, s_len, s_matches;
    const u32 tid = threadIdx.x;
    const u32 main_end = n - (LZ_MIN + 32);  // the main loop handles positions < n - 72 (:137)
    if (tid == 0) { s_cur = 4; s_heur = 0; s_matches = 0; }
    __syncthreads();
    for (;;) {
        const u32 cur = s_cur;
        if (cur >= main_end) break;
        const u32 tile_end = (main_end - cur > (u32)LZ_TILE) ? cur + LZ_TILE : main_end;
        // ---- candidates + 8-byte pre-test, events compacted in position order -------------------
        u32 nev = 0;
#pragma unroll 1
        for (int k = 0; k < LZ_TILE / LZ_DRV; k++) {
            const u32 p = cur + (u32)k * LZ_DRV + tid;
            u32 v = 0;
            bool ev = false;
            if (p < tile_end) {
                v = lz_candidate(prev, skip, p);
                ev = v > 0 && lz_eq4(in + p + LZ_MIN - 4, in + v + LZ_MIN - 4) && lz_eq4(in + p, in + v);
            }
            u32 tot;
            const u32 e = block_excl_add<LZ_DRV>(ev ? 1u : 0u, red, tot);
            if (ev) {
                u64 mask = 0;
                for (int b = 0; b < 48; b++) mask |= (u64)(in[p + b] == in[v + b]) << b;
                ev_pos[nev + e] = p;
                ev_ref[nev + e] = v;
                ev_mask[nev + e] = mask;
            }
            nev += tot;
        }
        __syncthreads();
        // ---- replay the heur chain over the events (one lane; masks only, no memory traffic) ----
        if (tid == 0) {
            u32 heur = s_heur;
            u32 accept = 0xFFFFFFFFu;
            for (u32 e = 0; e < nev; e++) {
                const u32 p = ev_pos[e];
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
        const u32 accept = s_accept;
        if (accept == 0xFFFFFFFFu) {
            if (tid == 0) s_cur = tile_end;
            __syncthreads();
            continue;
        }
        // ---- a real match: measure it with the whole workgroup ----------------------------------
        const u32 p = ev_pos[accept], v = ev_ref[accept];
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
            mlen[p] = len;
            s_len = len;
            s_matches++;
            s_cur = p + len;
        }
        __syncthreads();
        len = s_len;
        // swallowed positions p+1 .. p+len-1
        {
            const u32 a = p + 1, b = p + len;  // [a, b)
            for (u32 w = (a >> 5) + tid; w <= ((b - 1) >> 5) && a < b; w += LZ_DRV) {
                const u32 lo = w << 5;
                u32 m = 0xFFFFFFFFu;
                if (lo < a) m &= 0xFFFFFFFFu << (a - lo);
                if (lo + 32 > b) m &= 0xFFFFFFFFu >> (lo + 32 - b);
                skip[w] |= m;  // only this workgroup writes the bitmap; distinct lanes own distinct words
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    if (tid == 0) {
        result->end_pos = s_cur;
        result->n_matches = s_matches;
    }
}

// ---- 3. emission -------------------------------------------------------------------------------
__device__ __forceinline__ u32 lz_emit_count(const u8 * __restrict__ in, const u32 * __restrict__ prev, const u32 * __restrict__ skip,
                                             const u32 * __restrict__ mlen, u32 p, u32 & ml) {
    ml = 0;
    if (p < 4) return 1u;
    if (lz_skipped(skip, p)) return 0u;
    ml = mlen[p];
    if (ml) return 2u + (ml - LZ_MIN) / 254u;  // MATCH, 254 x q, remainder (:164-173)
    if (in[p] != LZ_ESC) return 1u;
    return lz_candidate(prev, skip, p) > 0 ? 2u : 1u;  // escape only when the slot was live (:176-181, :194)
}

template <int MODE>
__global__ void __launch_bounds__(LZ_BLOCK) k_lzp_emit(const u8 * __restrict__ in, u32 n, const u32 * __restrict__ prev, const u32 * __restrict__ skip,
                                                      const u32 * __restrict__ mlen, u32 * __restrict__ tile_sum, u8 * __restrict__ out) {
    __shared__ u32 lds[LZ_BLOCK / WAVE + 1];
    const u64 base = (u64)blockIdx.x * LZ_ETILE + (u64)threadIdx.x * LZ_ITEMS;
    u32 cnt[LZ_ITEMS], ml[LZ_ITEMS];
    u32 sum = 0;
#pragma unroll 4
    for (int k = 0; k < LZ_ITEMS; k++) {
        cnt[k] = (base + k < n) ? lz_emit_count(in, prev, skip, mlen, (u32)(base + k), ml[k]) : 0u;
        if (base + k >= n) ml[k] = 0;
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

s32 lzp_encode(const u8 * d_in, u32 n, u8 * d_out, Arena & tmp, hipStream_t s) {
    if (n < LZ_MIN + 32) return -1;  // :244
    const size_t mk = tmp.mark();
    const u32 m = n - 4;
    u32 * prev = tmp.take<u32>(n);
    u32 * mlen = tmp.take<u32>(n);
    u32 * skip = tmp.take<u32>((n >> 5) + 2);
    LzDriverOut * d_res = reinterpret_cast<LzDriverOut *>(tmp.take<u32>(4));
    u32 * d_total = tmp.take<u32>(1);
    {
        const size_t mk2 = tmp.mark();
        u32 * k0 = tmp.take<u32>(m);
        u32 * k1 = tmp.take<u32>(m);
        u32 * v0 = tmp.take<u32>(m);
        u32 * v1 = tmp.take<u32>(m);
        launch(k_lzp_hash, dim3((m + LZ_BLOCK - 1) / LZ_BLOCK), dim3(LZ_BLOCK), 0, s, d_in, n, k0);
        radix_pass<u32>(k0, k1, (const u32 *)nullptr, v1, m, 0, 0xFFFFFFFFu, 0u, tmp, s);
        radix_pass<u32>(k1, k0, (const u32 *)v1, v0, m, 8, 0xFFFFFFFFu, 0u, tmp, s);
        radix_pass<u32>(k0, k1, (const u32 *)v0, v1, m, 16, 0xFFFFFFFFu, 0u, tmp, s);
        HIP_CHECK(hipMemsetAsync(prev, 0, 16, s));
        launch(k_lzp_prev, dim3((m + LZ_BLOCK - 1) / LZ_BLOCK), dim3(LZ_BLOCK), 0, s, (const u32 *)k1, (const u32 *)v1, m, prev);
        tmp.release(mk2);
    }
    HIP_CHECK(hipMemsetAsync(mlen, 0, (size_t)n * 4, s));
    HIP_CHECK(hipMemsetAsync(skip, 0, ((size_t)(n >> 5) + 2) * 4, s));
    launch(k_lzp_driver, dim3(1), dim3(LZ_DRV), 0, s, d_in, n, (const u32 *)prev, skip, mlen, d_res);
    const u32 tiles = (n + LZ_ETILE - 1) / LZ_ETILE;
    u32 * tile_sum = tmp.take<u32>(tiles + 1);
    launch(k_lzp_emit<0>, dim3(tiles), dim3(LZ_BLOCK), 0, s, d_in, n, (const u32 *)prev, (const u32 *)skip, (const u32 *)mlen, tile_sum, (u8 *)nullptr);
    exclusive_scan_u32(tile_sum, tiles, d_total, tmp, s);
    u32 total = 0;
    HIP_CHECK(hipMemcpyAsync(&total, d_total, 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    s32 result = -1;
    if (total < n - 8) {  // the reference gives up once the output reaches n - 8 bytes (:128, :197)
        launch(k_lzp_emit<1>, dim3(tiles), dim3(LZ_BLOCK), 0, s, d_in, n, (const u32 *)prev, (const u32 *)skip, (const u32 *)mlen, tile_sum, d_out);
        result = (s32)total;
    }
    tmp.release(mk);
    return result;
}

// ---- decode -------------------------------------------------------------------------------------
constexpr int LZD_CHUNK = 16384;

struct LzDecodeOut {
    s32 size;  // decoded size or -1
};

__global__ void __launch_bounds__(LZ_DRV) k_lzp_decode(const u8 * __restrict__ in, u32 n, u8 * __restrict__ out, u32 max_out, u32 * __restrict__ lut,
                                                      LzDecodeOut * __restrict__ result) {
    __shared__ u32 red[LZ_DRV / WAVE + 1];
    __shared__ u32 s_ip, s_op, s_copy_src, s_copy_cnt, s_fail;
    const u32 tid = threadIdx.x;
    if (tid < 4) out[tid] = in[tid];
    if (tid == 0) { s_ip = 4; s_op = 4; s_fail = 0; }
    __threadfence_block();
    __syncthreads();
    for (;;) {
        const u32 ip = s_ip, op = s_op;
        if (s_fail || ip >= n || op >= max_out) break;
        // first 0xF2 in the next chunk of input
        u32 first = 0xFFFFFFFFu;
        const u32 chunk_end = (n - ip > (u32)LZD_CHUNK) ? ip + LZD_CHUNK : n;
        for (u32 i = ip + tid; i < chunk_end; i += LZ_DRV)
            if (in[i] == LZ_ESC) { first = i; break; }
        first = block_min<LZ_DRV>(first, red);
        const bool hit = first != 0xFFFFFFFFu;
        u32 lit = (hit ? first : chunk_end) - ip;
        bool room = true;
        if (lit > max_out - op) { lit = max_out - op; room = false; }
        // bulk literals: copy, then insert every (visited) output position into the table
        for (u32 j = tid; j < lit; j += LZ_DRV) out[op + j] = in[ip + j];
        __threadfence_block();
        __syncthreads();
        for (u32 j = tid; j < lit; j += LZ_DRV) atomicMax(&lut[lz_hash(load_be32(out + op + j - 4))], op + j);
        __threadfence_block();
        __syncthreads();
        if (tid == 0) {
            u32 i2 = ip + lit, o2 = op + lit;
            s_copy_cnt = 0;
            if (hit && room && o2 < max_out) {
                const u32 h = lz_hash(load_be32(out + o2 - 4));
                const u32 cand = lut[h];
                lut[h] = o2;
                if (cand > 0) {
                    i2++;
                    if (i2 == n) s_fail = 1;  // :215
                    else if (in[i2] != 255) {
                        u32 len = LZ_MIN;
                        for (;;) {  // :218-222
                            if (i2 == n) { s_fail = 1; break; }
                            const u8 b = in[i2++];
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
                    out[o2++] = in[i2++];
                }
            }
            s_ip = i2;
            s_op = o2;
        }
        __threadfence_block();
        __syncthreads();
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
    }
    if (tid == 0) result->size = s_fail ? -1 : (s32)s_op;
}

s32 lzp_decode(const u8 * d_in, u32 n, u8 * d_out, u32 max_out, Arena & tmp, hipStream_t s) {
    if (n < 4) return -1;  // :252
    const size_t mk = tmp.mark();
    u32 * lut = tmp.take<u32>((size_t)1 << LZ_HASH_BITS);
    LzDecodeOut * d_res = reinterpret_cast<LzDecodeOut *>(tmp.take<u32>(2));
    HIP_CHECK(hipMemsetAsync(lut, 0, sizeof(u32) << LZ_HASH_BITS, s));
    launch(k_lzp_decode, dim3(1), dim3(LZ_DRV), 0, s, d_in, n, d_out, max_out, lut, d_res);
    LzDecodeOut h;
    HIP_CHECK(hipMemcpyAsync(&h, d_res, sizeof h, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    tmp.release(mk);
    return h.size;
}

}  // namespace bz3
