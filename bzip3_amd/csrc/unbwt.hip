// unbwt.hip -- inverse Burrows-Wheeler transform of one block on gfx950.
// Replaces libsais_unbwt (reference include/libsais.h:5260-5262; init :4593-4617, decode :4618-4636),
// which builds a bigram psi table and then follows ONE dependent pointer chain of n/2 steps.
//
// Here:
//   1. psi is built by ONE stable radix pass (sort.hip) over the BWT bytes: the row of the k-th byte
//      (rows are 1-based around a virtual sentinel row inserted at `idx`; row 0 is the empty suffix)
//      is scattered to slot 1 + (stable rank by symbol); psi[0] = idx closes the cycle.
//      T[i] = F[psi^i(idx)], where F[row] is recovered from the 257-entry cumulative symbol table.
//   2. The single chain is cut at pseudo-random splitter rows (a multiplicative hash of the row
//      number, plus rows idx and 0).  One lane per splitter walks to the next splitter ONCE: it records
//      the segment length and parks the segment's text bytes in its slab (latency-bound, ~n/256-way
//      parallel random 4-byte reads, each of which fetches a 128-byte line).
//   3. The ~n/256-element splitter list is ranked by pointer jumping (log2 rounds, tiny).
//   4. The parked bytes are copied to their now-known positions (16 lanes per segment); the few segments
//      longer than their slab are walked on from where the first walk left them.
//   5. Input that is NOT a genuine BWT (a corrupted block) must still decode to what the reference produces, because
//      the CRC / LZP / size checks that follow decide the error code.  psi(0) = idx puts rows 0 and idx on one
//      cycle, so the chain from idx always ends at row 0, after D <= n bytes (D = n for a genuine BWT); the text is
//      laid out from position 0 and k_ub_tail reproduces what the reference's bigram chase emits once it is stuck
//      on its zero-filled table entries (oracle/bz3_oracle.c orc_unbwt spells the rules out; pinned against the
//      reference on random inputs).
// HBM layout: psi u32[n+1], splitter-id u32[n+1], a few arrays of n/256 words, slabs of 4 bytes per row.
// Algorithmic traffic: 11 B per byte; the walks are random 4-byte reads.
#include "prims.hpp"
#include "sort.hpp"
#include "stages.hpp"

namespace bz3 {

constexpr int UB_BLOCK = 256;

__device__ __forceinline__ bool ub_is_splitter(u32 row, u32 idx, int log_stride) {
    if (row == 0 || row == idx || log_stride == 0) return true;
    return ((row * 0x9E3779B1u) >> (32 - log_stride)) == 0u;
}

// 64 bytes per thread, 16 at a time, every load of a group in flight before the first is counted (a grid-stride loop of single-byte
// loads is one exposed HBM round trip per byte: 1.2 ms for 256 MiB in round 2's profile; same pattern as bwt.hip k_bwt_sym_hist).
__global__ void __launch_bounds__(UB_BLOCK) k_ub_hist(const u8 * __restrict__ in, u32 n, u32 * __restrict__ hist) {
    __shared__ u32 bins[256];
    bins[threadIdx.x] = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * (UB_BLOCK * 64);
    const u64 last = (u64)n - 1;  // n >= 2 here
    for (u32 g = 0; g < 4; g++) {
        u8 c[16];
#pragma unroll
        for (u32 k = 0; k < 16; k++) {
            const u64 i = base + (u64)(g * 16 + k) * UB_BLOCK + threadIdx.x;
            c[k] = in[i < n ? i : last];
        }
#pragma unroll
        for (u32 k = 0; k < 16; k++) {
            const u64 i = base + (u64)(g * 16 + k) * UB_BLOCK + threadIdx.x;
            if (i < n) atomicAdd(&bins[c[k]], 1u);
        }
    }
    __syncthreads();
    if (bins[threadIdx.x]) atomicAdd(&hist[threadIdx.x], bins[threadIdx.x]);
}

// cum[c] = 1 + number of bytes smaller than c; cum[256] = n + 1.  Also closes the cycle psi[0] = idx.
__global__ void __launch_bounds__(256) k_ub_cum(const u32 * __restrict__ hist, u32 * __restrict__ cum, u32 * __restrict__ psi, u32 idx) {
    __shared__ u32 lds[256 / WAVE + 1];
    u32 tot;
    u32 pre = block_excl_add<256>(hist[threadIdx.x], lds, tot);
    cum[threadIdx.x] = pre + 1;
    if (threadIdx.x == 0) {
        cum[256] = tot + 1;
        psi[0] = idx;
    }
}

__global__ void __launch_bounds__(UB_BLOCK) k_ub_flags(u32 rows, u32 idx, int log_stride, u32 * __restrict__ flags) {
    const u32 r = blockIdx.x * UB_BLOCK + threadIdx.x;
    if (r < rows) flags[r] = ub_is_splitter(r, idx, log_stride) ? 1u : 0u;
}

__global__ void __launch_bounds__(UB_BLOCK) k_ub_collect(u32 rows, u32 idx, int log_stride, const u32 * __restrict__ sid, u32 * __restrict__ split_row) {
    const u32 r = blockIdx.x * UB_BLOCK + threadIdx.x;
    if (r < rows && ub_is_splitter(r, idx, log_stride)) split_row[sid[r]] = r;
}

// F[r], the first symbol of row r >= 1: largest symbol s with c[s] <= r (c = the 257 cumulative counts, in LDS).
__device__ __forceinline__ u32 ub_symbol_lds(const u32 * c, u32 r) {
    u32 lo = 0, hi = 256;
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const u32 mid = (lo + hi) >> 1;
        if (c[mid] <= r) lo = mid; else hi = mid;
    }
    return lo;
}

// THE walk (round 3: it used to be two -- one for the lengths, one for the bytes -- and every step of either fetches a whole 128-byte
// line for one 4-byte word, 33 GB per walk of a 256 MiB block by the FETCH_SIZE counter: profiles/r03_pmc_fetch_stages_256MiB.txt).
// Every splitter walks to the next one ONCE: segment length, successor, and the text bytes of the segment, which cannot go to their
// place yet (the position is known after the list ranking) and are parked in the splitter's slab of `cap` bytes (cap = 4 x the mean
// segment length; the few longer segments leave the row they reached in `resume` and k_ub_walk_long finishes them).  Bytes are
// collected four at a time: one store per four steps.
__global__ void __launch_bounds__(UB_BLOCK) k_ub_walk(const u32 * __restrict__ psi, const u32 * __restrict__ sid, const u32 * __restrict__ split_row, u32 nsplit,
                                                     u32 idx, int log_stride, u32 rows, const u32 * __restrict__ cum, u32 cap, u32 * __restrict__ succ,
                                                     u32 * __restrict__ dist, u8 * __restrict__ slab, u32 * __restrict__ resume) {
    __shared__ u32 c[257];
    for (int k = threadIdx.x; k < 257; k += UB_BLOCK) c[k] = cum[k];
    __syncthreads();
    const u32 j = blockIdx.x * UB_BLOCK + threadIdx.x;
    if (j >= nsplit) return;
    u32 r = split_row[j];
    if (r == 0) {  // terminal of the list: row 0 is the empty suffix, nothing is emitted for it
        succ[j] = j;
        dist[j] = 0;
        return;
    }
    u32 * __restrict__ mine = reinterpret_cast<u32 *>(slab + (size_t)j * cap);  // (cap is a multiple of 4)
    u32 len = 0, word = 0;
    do {
        const u32 nr = psi[r];  // issue the dependent load first; the symbol search overlaps it
        if (len < cap) {
            word |= ub_symbol_lds(c, r) << (8u * (len & 3u));
            if ((len & 3u) == 3u) {
                mine[len >> 2] = word;
                word = 0;
            }
        } else if (len == cap) {
            resume[j] = r;
        }
        r = nr;
        len++;
    } while (!ub_is_splitter(r, idx, log_stride) && len <= rows);
    if (len < cap && (len & 3u)) mine[len >> 2] = word;  // the last, partial word
    succ[j] = sid[r];
    dist[j] = len;
}

// One pointer-jumping round: dist = distance to the terminal, succ = 2^k-th successor.
__global__ void __launch_bounds__(UB_BLOCK) k_ub_jump(const u32 * __restrict__ succ_in, const u32 * __restrict__ dist_in, u32 nsplit, u32 * __restrict__ succ_out,
                                                     u32 * __restrict__ dist_out) {
    const u32 j = blockIdx.x * UB_BLOCK + threadIdx.x;
    if (j >= nsplit) return;
    const u32 sj = succ_in[j];
    u64 d = (u64)dist_in[j] + dist_in[sj];
    dist_out[j] = d > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)d;
    succ_out[j] = succ_in[sj];
}

// After the list ranking: the parked bytes of every segment go to their place.  16 lanes per segment, 16 bytes per lane and trip
// (the slab is aligned, the destination is not: 128-bit stores at any byte address).  seg_len was saved before the jumping rounds.
__global__ void __launch_bounds__(UB_BLOCK) k_ub_place(const u8 * __restrict__ slab, const u32 * __restrict__ seg_len, const u32 * __restrict__ dist,
                                                      const u32 * __restrict__ succ, const u32 * __restrict__ sid, u32 idx, u32 cap, u32 nsplit, u32 n,
                                                      u8 * __restrict__ out) {
    const u32 t = blockIdx.x * UB_BLOCK + threadIdx.x;
    const u32 j = t >> 4, part = t & 15u;
    if (j >= nsplit) return;
    // On the chain idx -> ... -> row 0?  After the jumping rounds every chain element points at the terminal; splitters
    // of other cycles (corrupt input only) never do.  D = bytes on the chain (n for a genuine BWT).
    if (succ[j] != sid[0]) return;
    const u32 D = dist[sid[idx]];
    const u32 d = dist[j];
    if (d > D) return;
    const u32 len = seg_len[j];
    const u32 m = len < cap ? len : cap;
    const u32 limit = n & ~1u;  // the reference's loop writes 2 * (n / 2) bytes, then U[n-1] separately (:5155)
    const u64 pos0 = (u64)D - d;
    const u8 * __restrict__ src = slab + (size_t)j * cap;
    for (u32 o = part * 16u; o < m; o += 256u) {
        if (o + 16u <= m && pos0 + o + 16u <= limit) {
            const uint4 q = *reinterpret_cast<const uint4 *>(src + o);
            PackedU128 w;
            w.v[0] = q.x; w.v[1] = q.y; w.v[2] = q.z; w.v[3] = q.w;
            *reinterpret_cast<PackedU128 *>(out + pos0 + o) = w;
        } else {
            for (u32 k = o; k < m && k < o + 16u; k++)
                if (pos0 + k < limit) out[pos0 + k] = src[k];
        }
    }
}

// The segments that did not fit their slab (longer than 4 x the mean: ~2 % of them, ~9 % of the rows) are walked on from the row
// the first walk left in `resume`, their bytes going straight to their place.
__global__ void __launch_bounds__(UB_BLOCK) k_ub_walk_long(const u32 * __restrict__ psi, const u32 * __restrict__ resume, const u32 * __restrict__ seg_len,
                                                          const u32 * __restrict__ dist, const u32 * __restrict__ succ, const u32 * __restrict__ sid, u32 idx,
                                                          const u32 * __restrict__ cum, u32 cap, u32 nsplit, u32 n, u8 * __restrict__ out) {
    __shared__ u32 c[257];
    for (int k = threadIdx.x; k < 257; k += UB_BLOCK) c[k] = cum[k];
    __syncthreads();
    const u32 j = blockIdx.x * UB_BLOCK + threadIdx.x;
    if (j >= nsplit) return;
    const u32 len = seg_len[j];
    if (len <= cap) return;
    if (succ[j] != sid[0]) return;
    const u32 D = dist[sid[idx]];
    const u32 d = dist[j];
    if (d > D) return;
    const u32 limit = n & ~1u;
    u32 r = resume[j];
    u64 pos = (u64)D - d + cap;
    for (u32 t = cap; t < len; t++) {
        const u32 nr = psi[r];
        const u32 sym = ub_symbol_lds(c, r);
        if (pos < limit) out[pos] = (u8)sym;
        pos++;
        r = nr;
    }
}

__device__ __forceinline__ u32 ub_first_symbol(const u32 * __restrict__ cum, u32 r) {  // F[r], r >= 1
    u32 lo = 0, hi = 256;
    for (int b = 0; b < 8; b++) {
        const u32 mid = (lo + hi) >> 1;
        if (cum[mid] <= r) lo = mid; else hi = mid;
    }
    return lo;
}

// What the reference leaves behind the chain (nothing for a genuine BWT, where D = n), and U[n-1] = first BWT byte.
// Rules and their derivation from include/libsais.h:4534-4636: oracle/bz3_oracle.c, orc_unbwt.
__global__ void __launch_bounds__(UB_BLOCK) k_ub_tail(const u8 * __restrict__ in, const u32 * __restrict__ psi, const u32 * __restrict__ cum,
                                                     const u32 * __restrict__ dist, const u32 * __restrict__ sid, u32 idx, u32 n, u8 * __restrict__ out) {
    const u32 D = dist[sid[idx]];
    const u32 limit = n & ~1u;
    const u32 lastc = in[0];
    const u32 E = cum[lastc];  // row of the suffix that is the last character alone = LF(0)
    const u32 gid = blockIdx.x * UB_BLOCK + threadIdx.x;
    u32 start = D;
    if (D < limit && (D & 1u)) {  // E was hit on an even step of the bigram chase
        int shift = 0;
        while ((n >> shift) > (1u << 17)) shift++;
        const bool same_group = E >= 2 && ((E - 1) >> shift) == (E >> shift);
        if (gid == 0) {
            if (same_group || E + 1 > n) {
                out[D] = 0;
            } else {
                out[D - 1] = (u8)ub_first_symbol(cum, E + 1);
                out[D] = (u8)ub_first_symbol(cum, psi[E + 1]);
            }
        }
        start = D + 1;
    }
    if (start < limit) {  // stuck at table entry 0: the bigram of the first non-empty bucket, over and over
        const u32 q0 = (E == 1) ? 2u : 1u;
        const u32 hi = ub_first_symbol(cum, q0), lo = ub_first_symbol(cum, psi[q0]);
        for (u64 j = (u64)start + gid; j < limit; j += (u64)gridDim.x * UB_BLOCK)
            if (j != n - 1) out[j] = (u8)((j & 1u) ? lo : hi);  // U[n-1] belongs to thread 0 below
    }
    if (gid == 0) out[n - 1] = (u8)lastc;
}

// psi, splitter ids, the sorter's and the scans' scratch, seven splitter arrays, the slabs (4 bytes per row and a margin).  Splitters:
// every row below 2^17 rows, between 2^17 and 2^18 of them up to 2^25 rows, one row in 256 beyond (hashed: a few per cent either way).
size_t unbwt_workspace_bytes(u64 n) {
    const u64 splitters = n + 1 < 300000 ? n + 1 : 300000 + (n >> 7);
    return (n + 64) * 8 + radix_temp_bytes(n) + scan_temp_words(n + 1) * 4 + (splitters + 4096) * 32 + (n + (n >> 3)) * 4 + (1u << 20);
}

void bwt_inverse(const u8 * d_in, u32 n, u32 idx, u8 * d_out, Arena & tmp, hipStream_t s) {
    if (n == 0) return;
    if (n == 1) {
        HIP_CHECK(hipMemcpyAsync(d_out, d_in, 1, hipMemcpyDeviceToDevice, s));
        return;
    }
    const size_t mk = tmp.mark();
    const u32 rows = n + 1;
    u32 * psi = tmp.take<u32>(rows);
    u32 * sid = tmp.take<u32>(rows);
    u32 * hist = tmp.take<u32>(256);
    u32 * cum = tmp.take<u32>(257);
    u32 * d_total = tmp.take<u32>(1);

    // 1. psi by one stable radix pass: value of byte k is its row k + (k >= idx), destination base 1
    HIP_CHECK(hipMemsetAsync(hist, 0, 256 * 4, s));
    launch(k_ub_hist, dim3((u32)(((u64)n + UB_BLOCK * 64 - 1) / (UB_BLOCK * 64))), dim3(UB_BLOCK), 0, s, d_in, n, hist);
    launch(k_ub_cum, dim3(1), dim3(256), 0, s, (const u32 *)hist, cum, psi, idx);
    radix_pass<u8>(d_in, (u8 *)nullptr, (const u32 *)nullptr, psi, n, 0, idx, 1u, tmp, s);

    // 2. splitters
    // Segment lengths are geometric (splitters are a hash of the row number), every lane walks its whole segment, and all lanes of
    // a launch are resident at once, so a walk lasts as long as its LONGEST segment: mean x ln(number of segments) dependent HBM
    // round trips.  One splitter per 256 rows (round 2 had one per 1024: 21.0 against 14.6 ms for the then two walks of a 256 MiB
    // block) keeps the longest segment at ~3.5 k steps; the list ranking is n / 256 elements (tens of microseconds per jump round).
    constexpr int max_log_stride = 8;
    int log_stride = 0;
    while (log_stride < max_log_stride && ((u64)rows >> (log_stride + 1)) >= 65536) log_stride++;
    const dim3 grows((rows + UB_BLOCK - 1) / UB_BLOCK);
    launch(k_ub_flags, grows, dim3(UB_BLOCK), 0, s, rows, idx, log_stride, sid);
    exclusive_scan_u32(sid, rows, d_total, tmp, s);
    u32 nsplit = 0;
    HIP_CHECK(hipMemcpyAsync(&nsplit, d_total, 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    u32 * split_row = tmp.take<u32>(nsplit);
    u32 * seg_len = tmp.take<u32>(nsplit);
    u32 * resume = tmp.take<u32>(nsplit);
    u32 * succ[2] = {tmp.take<u32>(nsplit), tmp.take<u32>(nsplit)};
    u32 * dist[2] = {tmp.take<u32>(nsplit), tmp.take<u32>(nsplit)};
    const u32 cap = 4u << log_stride;  // bytes per slab: 4 x the mean segment length (splitters are one row in 2^log_stride)
    u8 * slab = tmp.take<u8>((size_t)nsplit * cap + 64);
    const dim3 gs((nsplit + UB_BLOCK - 1) / UB_BLOCK);
    launch(k_ub_collect, grows, dim3(UB_BLOCK), 0, s, rows, idx, log_stride, (const u32 *)sid, split_row);
    launch(k_ub_walk, gs, dim3(UB_BLOCK), 0, s, (const u32 *)psi, (const u32 *)sid, (const u32 *)split_row, nsplit, idx, log_stride, rows, (const u32 *)cum, cap, succ[0],
           dist[0], slab, resume);
    HIP_CHECK(hipMemcpyAsync(seg_len, dist[0], (size_t)nsplit * 4, hipMemcpyDeviceToDevice, s));

    // 3. rank the splitter list
    int cur = 0;
    for (u64 span = 1; span < nsplit; span <<= 1) {
        launch(k_ub_jump, gs, dim3(UB_BLOCK), 0, s, (const u32 *)succ[cur], (const u32 *)dist[cur], nsplit, succ[cur ^ 1], dist[cur ^ 1]);
        cur ^= 1;
    }
    // 4. the parked bytes to their places, the long segments' rest straight there
    launch(k_ub_place, dim3((u32)(((u64)nsplit * 16 + UB_BLOCK - 1) / UB_BLOCK)), dim3(UB_BLOCK), 0, s, (const u8 *)slab, (const u32 *)seg_len, (const u32 *)dist[cur],
           (const u32 *)succ[cur], (const u32 *)sid, idx, cap, nsplit, n, d_out);
    launch(k_ub_walk_long, gs, dim3(UB_BLOCK), 0, s, (const u32 *)psi, (const u32 *)resume, (const u32 *)seg_len, (const u32 *)dist[cur], (const u32 *)succ[cur],
           (const u32 *)sid, idx, (const u32 *)cum, cap, nsplit, n, d_out);
    launch(k_ub_tail, dim3(256), dim3(UB_BLOCK), 0, s, d_in, (const u32 *)psi, (const u32 *)cum, (const u32 *)dist[cur], (const u32 *)sid, idx, n, d_out);
    HIP_CHECK(hipStreamSynchronize(s));
    tmp.release(mk);
}

}  // namespace bz3
