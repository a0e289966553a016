// bwt.hip -- forward Burrows-Wheeler transform of one block on gfx950.
// Replaces libsais_bwt (reference include/libsais.h:4095-4121, SA-IS: :3941-3983, :3740-3939), which
// is a sequential induced-sorting algorithm whose inner scans carry a dependency through 256 bucket
// cursors.  This is NOT a port of it: the suffix array is built by prefix doubling on top of the
// stable LSD radix sorter of sort.hip, which is the HBM-streaming formulation the MI355X wants.
//
//   round 0 : key(i) = the first 8 symbols of suffix i as ranks among the byte values present in the block (0 = past the end),
//             most significant first; sort all n with one 8-bit LSD pass per bit of a rank (7 passes for text, 8 at most)
//   round h : every suffix still sharing its h-prefix with another one ("active") gets the 64-bit key
//             (group << 32) | rank(i + h); only the active elements are sorted, written back to their
//             group's slots, re-grouped, and the now-unique ones are dropped.  h doubles: 8, 16, 32...
//   rank(j) : position of the head of j's group in the current order, + h;   past-the-end suffixes
//             get n-1-i (< h), so that a suffix that is a proper prefix of another sorts first and two
//             such suffixes order by length -- exactly the order libsais produces.
//   output  : U[0] = T[n-1]; U[i < i0 ? i+1 : i] = T[SA[i]-1] for i != i0 = rank of suffix 0; idx = i0+1.
//
// HBM layout per block of n bytes (carved from the per-device workspace):
//   SA u32[n], ISA u32[n], 2 x key u64[m], 2 x suffix u32[m], 2 x slot u32[m], 2 x flag/scan u32[m]
// Algorithmic traffic: 11 B per input byte; implementation traffic is
// radix passes x 32 B per sorted element and is reported through BwtStats.
#include "prims.hpp"
#include "sort.hpp"
#include "stages.hpp"

namespace bz3 {

constexpr int BW_BLOCK = 256;

__device__ __forceinline__ u64 bswap64(u64 v) {
    v = ((v & 0x00FF00FF00FF00FFull) << 8) | ((v >> 8) & 0x00FF00FF00FF00FFull);
    v = ((v & 0x0000FFFF0000FFFFull) << 16) | ((v >> 16) & 0x0000FFFF0000FFFFull);
    return (v << 32) | (v >> 32);
}

__device__ __forceinline__ u64 load_be64_padded(const u8 * __restrict__ t, u64 i, u64 n) {
    if (i + 8 <= n && ((reinterpret_cast<uintptr_t>(t) + i) & 7) == 0) return bswap64(*reinterpret_cast<const u64 *>(t + i));
    if (i >= n) return 0;
    const u64 last = n - 1;
    u8 c[8];
#pragma unroll
    for (int k = 0; k < 8; k++) c[k] = t[i + k < n ? i + k : last];  // in flight together (see sort.hip), masked below
    u64 v = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) v = (v << 8) | (i + k < n ? (u64)c[k] : 0ull);
    return v;
}

// 8 keys per thread from two aligned 8-byte loads.
__global__ void __launch_bounds__(BW_BLOCK) k_bwt_prefix_keys(const u8 * __restrict__ t, u32 n, u64 * __restrict__ keys) {
    const u64 base = ((u64)blockIdx.x * BW_BLOCK + threadIdx.x) * 8;
    if (base >= n) return;
    const u64 a = load_be64_padded(t, base, n);
    const u64 b = load_be64_padded(t, base + 8, n);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (base + k < n) keys[base + k] = k == 0 ? a : ((a << (8 * k)) | (b >> (64 - 8 * k)));
    }
}

// ---- alphabet compaction (round 2) ------------------------------------------------------------------------------------------
// The initial sort orders all n suffixes by their first 8 symbols with one LSD radix pass per 8 key bits.  Text uses far fewer than
// 256 byte values, so the bytes are replaced by their rank among the byte values PRESENT in the block (order-preserving, 1-based;
// 0 = "past the end", which also makes a suffix that ends inside its 8-symbol prefix sort first right away): with s = present
// values the 8 symbols need 8 * bits(s) key bits = bits(s) passes instead of 8 (7 for text, 5 for a 16-symbol source).
__global__ void __launch_bounds__(BW_BLOCK) k_bwt_sym_hist(const u8 * __restrict__ t, u32 n, u32 * __restrict__ hist) {
    __shared__ u32 bins[256];
    bins[threadIdx.x] = 0;
    __syncthreads();
    // 64 bytes per thread, 16 at a time, every load of a group in flight before the first is counted (sort.hip explains why the
    // obvious `if (i < n) ... t[i]` loop is one exposed HBM round trip per byte); index clamped, lane masked.
    const u64 base = (u64)blockIdx.x * (BW_BLOCK * 64);
    const u64 last = (u64)n - 1;
    for (u32 g = 0; g < 4; g++) {
        u8 c[16];
#pragma unroll
        for (u32 k = 0; k < 16; k++) {
            const u64 i = base + (u64)(g * 16 + k) * BW_BLOCK + threadIdx.x;
            c[k] = t[i < n ? i : last];
        }
#pragma unroll
        for (u32 k = 0; k < 16; k++) {
            const u64 i = base + (u64)(g * 16 + k) * BW_BLOCK + threadIdx.x;
            if (i < n) atomicAdd(&bins[c[k]], 1u);
        }
    }
    __syncthreads();
    if (bins[threadIdx.x]) atomicAdd(&hist[threadIdx.x], bins[threadIdx.x]);
}

// sym[c] (in: count of byte value c) -> 1-based rank of c among the present values (0 if absent); sym[256] = number of present values.
__global__ void __launch_bounds__(256) k_bwt_sym_map(u32 * __restrict__ sym) {
    __shared__ u32 lds[256 / WAVE + 1];
    const u32 present = sym[threadIdx.x] ? 1u : 0u;
    u32 total;
    const u32 before = block_excl_add<256>(present, lds, total);
    sym[threadIdx.x] = present ? before + 1u : 0u;
    if (threadIdx.x == 0) sym[256] = total;
}

// 8 keys per thread: key(i) = the ranks of bytes i .. i+7, `bits` bits each, most significant first; 0 past the end.
__global__ void __launch_bounds__(BW_BLOCK) k_bwt_prefix_keys_mapped(const u8 * __restrict__ t, u32 n, const u32 * __restrict__ sym, u32 bits, u64 * __restrict__ keys) {
    __shared__ u32 map[256];
    map[threadIdx.x] = sym[threadIdx.x];
    __syncthreads();
    const u64 base = ((u64)blockIdx.x * BW_BLOCK + threadIdx.x) * 8;
    if (base >= n) return;
    const u64 last = (u64)n - 1;
    u8 c[16];
    if (base + 16 <= n && ((reinterpret_cast<uintptr_t>(t) + base) & 7) == 0) {  // two aligned 8-byte loads (little endian)
        const u64 a = *reinterpret_cast<const u64 *>(t + base);
        const u64 b = *reinterpret_cast<const u64 *>(t + base + 8);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            c[k] = (u8)(a >> (8 * k));
            c[k + 8] = (u8)(b >> (8 * k));
        }
    } else {
#pragma unroll
        for (int k = 0; k < 15; k++) c[k] = t[base + k < n ? base + k : last];  // 15 loads in flight, then the table look-ups
    }
    u32 m[15];
#pragma unroll
    for (int k = 0; k < 15; k++) m[k] = (base + k < n) ? map[c[k]] : 0u;
    const u64 mask = (bits * 8u >= 64u) ? ~0ull : ((1ull << (bits * 8u)) - 1ull);
    u64 key = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) key = (key << bits) | m[k];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (base + k < n) keys[base + k] = key;
        if (k < 7) key = ((key << bits) & mask) | m[k + 8];
    }
}

// flags[k] = 1 if sorted element k starts a new group (its key differs from its predecessor's).
__global__ void __launch_bounds__(BW_BLOCK) k_bwt_head_flags(const u64 * __restrict__ keys, u32 m, u32 * __restrict__ flags) {
    const u32 k = blockIdx.x * BW_BLOCK + threadIdx.x;
    if (k < m) flags[k] = (k == 0 || keys[k] != keys[k - 1]) ? 1u : 0u;
}

// headslot[dense group id] = slot of the group head.  `excl` = exclusive scan of the head flags.
__global__ void __launch_bounds__(BW_BLOCK) k_bwt_head_slots(const u64 * __restrict__ keys, const u32 * __restrict__ excl, const u32 * __restrict__ slots, u32 m,
                                                            u32 * __restrict__ headslot) {
    const u32 k = blockIdx.x * BW_BLOCK + threadIdx.x;
    if (k >= m) return;
    if (k == 0 || keys[k] != keys[k - 1]) headslot[excl[k]] = slots ? slots[k] : k;
}

// Publishes the rank (ISA) of every sorted element, the SA entry of the ones that just became unique, and flags the ones that
// stay active (group size > 1).  Both scatters are random 4-byte stores, the costliest thing here, so none is made in vain:
// SA is only read by k_bwt_emit, and a suffix that stays active gets its slot again in a later round, when it is unique; in a
// doubling round (DOUBLING) the upper key word IS the suffix's current rank (k_bwt_doubling_keys), and the first sub-group of
// every group keeps it.
template <bool DOUBLING>
__global__ void __launch_bounds__(BW_BLOCK) k_bwt_assign(const u64 * __restrict__ keys, const u32 * __restrict__ vals, const u32 * __restrict__ excl,
                                                        const u32 * __restrict__ slots, const u32 * __restrict__ headslot, u32 m, u32 * __restrict__ sa,
                                                        u32 * __restrict__ isa, u32 * __restrict__ keep) {
    const u32 k = blockIdx.x * BW_BLOCK + threadIdx.x;
    if (k >= m) return;
    const u64 key = keys[k];
    const bool head = (k == 0 || key != keys[k - 1]);
    const bool next_head = (k + 1 == m) || keys[k + 1] != key;
    const u32 gid = excl[k] + (head ? 1u : 0u) - 1u;
    const u32 v = vals[k];
    const u32 rank = headslot[gid];
    const bool unique = head && next_head;
    if (unique) sa[slots ? slots[k] : k] = v;
    if (!DOUBLING || rank != (u32)(key >> 32)) isa[v] = rank;
    keep[k] = unique ? 0u : 1u;
}

// Stream compaction of the still-active elements.  `excl` = exclusive scan of keep flags, which are
// re-derived from the keys (the scan overwrote them).
__global__ void __launch_bounds__(BW_BLOCK) k_bwt_compact(const u64 * __restrict__ keys, const u32 * __restrict__ vals, const u32 * __restrict__ slots,
                                                         const u32 * __restrict__ excl, u32 m, u32 * __restrict__ vals_out, u32 * __restrict__ slots_out) {
    const u32 k = blockIdx.x * BW_BLOCK + threadIdx.x;
    if (k >= m) return;
    const u64 key = keys[k];
    const bool head = (k == 0 || key != keys[k - 1]);
    const bool next_head = (k + 1 == m) || keys[k + 1] != key;
    if (head && next_head) return;
    const u32 j = excl[k];
    vals_out[j] = vals[k];
    slots_out[j] = slots ? slots[k] : k;
}

// key(k) = (group of suffix s) << 32 | rank of suffix s + h   (s = vals[k])
__global__ void __launch_bounds__(BW_BLOCK) k_bwt_doubling_keys(const u32 * __restrict__ vals, const u32 * __restrict__ isa, u32 m, u32 n, u32 h,
                                                               u64 * __restrict__ keys) {
    const u32 k = blockIdx.x * BW_BLOCK + threadIdx.x;
    if (k >= m) return;
    const u32 s = vals[k];
    const u64 j = (u64)s + h;
    const u32 lo = (j < n) ? isa[j] + h : (n - 1u - s);
    keys[k] = ((u64)isa[s] << 32) | lo;
}

// ---- fused grouping (opt-in: BZ3_BWT_FUSED=1; NOT the default until it has been timed on the GPU) ---------------------------
// The regrouping of a freshly sorted list above is seven launches that stream the list again and again: head flags (write), scan,
// head slots, assign (flags recomputed, keep flags written), scan, compact, a copy of the compacted suffixes -- about 110 bytes of
// sequential traffic per element and round beside the one random store that is the actual work.  Here it is two passes over tiles
// of 2048 elements (8 consecutive elements per thread) and a one-workgroup spine between them:
//   k_bg_reduce : per tile, the position of its last group head and the number of elements that stay active
//   k_bg_spine  : exclusive running maximum / sum over the tiles (at most n / 2048 of them)
//   k_bg_apply  : flags again from the keys; in-tile running maximum of the head positions (+ the tile's carry) gives every
//                 element its group head, hence its rank; in-tile sum of the keep flags (+ the tile's offset) gives the slot in
//                 the compacted list; writes SA / ISA as k_bwt_assign does and the compacted (suffix, slot, rank) triples --
//                 the rank rides along so that k_bwt_doubling_keys_grp reads it in order instead of gathering isa[s]
// ~45 bytes of sequential traffic per element and round; same SA / ISA / active list by construction.
constexpr int BG_ITEMS = 8;
constexpr int BG_TILE = BW_BLOCK * BG_ITEMS;
struct alignas(16) BgU64x2 { u64 x, y; };
struct alignas(16) BgU32x4 { u32 x, y, z, w; };

// kv[j + 1] = keys[base + j] for j = -1 .. BG_ITEMS (indices clamped into [0, m - 1]: a clamped copy only ever meets a flag test
// that is overridden by its own bounds check).  Whole tiles: four 16-byte loads + the two neighbours, all in flight together.
__device__ __forceinline__ void bg_load_keys(const u64 * __restrict__ keys, u32 m, u64 base, u64 (&kv)[BG_ITEMS + 2]) {
    const u64 last = (u64)m - 1;
    kv[0] = keys[base > 0 ? (base - 1 < last ? base - 1 : last) : 0];
    kv[BG_ITEMS + 1] = keys[base + BG_ITEMS < last ? base + BG_ITEMS : last];
    if (base + BG_ITEMS <= m) {
        const BgU64x2 * __restrict__ q = reinterpret_cast<const BgU64x2 *>(keys + base);  // base is a multiple of 8 elements, the array 256-byte aligned
#pragma unroll
        for (int j = 0; j < BG_ITEMS / 2; j++) {
            const BgU64x2 t = q[j];
            kv[1 + 2 * j] = t.x;
            kv[2 + 2 * j] = t.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < BG_ITEMS; j++) kv[1 + j] = keys[base + j < last ? base + j : last];
    }
}
__device__ __forceinline__ void bg_load_u32(const u32 * __restrict__ a, u32 m, u64 base, u32 (&v)[BG_ITEMS]) {
    if (base + BG_ITEMS <= m) {
        const BgU32x4 * __restrict__ q = reinterpret_cast<const BgU32x4 *>(a + base);
        const BgU32x4 t0 = q[0], t1 = q[1];
        v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w;
        v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
    } else {
        const u64 last = (u64)m - 1;
#pragma unroll
        for (int j = 0; j < BG_ITEMS; j++) v[j] = a[base + j < last ? base + j : last];
    }
}
// flags of element k = base + j (k < m): head = first of its group, uniq = a group of one
#define BG_FLAGS(j, k, head, uniq)                                                   \
    const bool head = (k) == 0 || kv[(j) + 1] != kv[(j)];                            \
    const bool uniq = head && ((k) + 1 == (u64)m || kv[(j) + 2] != kv[(j) + 1])

__global__ void __launch_bounds__(BW_BLOCK) k_bg_reduce(const u64 * __restrict__ keys, u32 m, u32 * __restrict__ tile_head, u32 * __restrict__ tile_keep) {
    __shared__ u32 lds[BW_BLOCK / WAVE + 1];
    const u64 base = (u64)blockIdx.x * BG_TILE + (u64)threadIdx.x * BG_ITEMS;
    u64 kv[BG_ITEMS + 2];
    bg_load_keys(keys, m, base < m ? base : (u64)m - 1, kv);
    u32 hp = 0, keep = 0;  // hp = 1 + position of the last head seen (0 = none)
#pragma unroll
    for (int j = 0; j < BG_ITEMS; j++) {
        const u64 k = base + j;
        if (k < m) {
            BG_FLAGS(j, k, head, uniq);
            if (head) hp = (u32)k + 1u;
            keep += uniq ? 0u : 1u;
        }
    }
    hp = block_max<BW_BLOCK>(hp, lds);
    keep = block_sum<BW_BLOCK>(keep, lds);
    if (threadIdx.x == 0) {
        tile_head[blockIdx.x] = hp;
        tile_keep[blockIdx.x] = keep;
    }
}

template <int BLOCK>
__device__ __forceinline__ u32 bg_block_excl_max(u32 v, u32 * lds) {  // lds: BLOCK / 64 words; ends with a barrier
    const u32 incl = wave_incl_max(v);
    if (lane_id() == WAVE - 1) lds[wave_id()] = incl;
    u32 up = __shfl_up(incl, 1u);
    if (lane_id() == 0) up = 0u;
    __syncthreads();
    u32 carry = 0;
    for (int w = 0; w < wave_id(); w++) carry = lds[w] > carry ? lds[w] : carry;
    __syncthreads();
    return up > carry ? up : carry;
}

// One workgroup: tile_head[t] <- maximum over the tiles before t, tile_keep[t] <- sum over the tiles before t, *total <- sum of all.
constexpr int BG_SPINE = 1024;
__global__ void __launch_bounds__(BG_SPINE) k_bg_spine(u32 * __restrict__ tile_head, u32 * __restrict__ tile_keep, u32 tiles, u32 * __restrict__ total) {
    __shared__ u32 lds[BG_SPINE / WAVE + 1];
    // a thread owns `per` consecutive tiles and walks them eight at a time, the loads of a batch in flight together (up to 256
    // tiles per thread at 511 MiB: one exposed round trip per tile would cost more than the passes this kernel sits between)
    const u32 per = (((tiles + BG_SPINE - 1) / BG_SPINE) + 7u) & ~7u;
    const u32 t0 = threadIdx.x * per, t1 = t0 + per < tiles ? t0 + per : tiles;
    const u32 last = tiles - 1u;
    u32 hp = 0, sum = 0;
    for (u32 t = t0; t < t1; t += 8) {
        u32 h[8], c[8];
#pragma unroll
        for (u32 k = 0; k < 8; k++) {
            const u32 i = t + k < last ? t + k : last;
            h[k] = tile_head[i];
            c[k] = tile_keep[i];
        }
#pragma unroll
        for (u32 k = 0; k < 8; k++)
            if (t + k < t1) {
                hp = h[k] > hp ? h[k] : hp;
                sum += c[k];
            }
    }
    u32 run_hp = bg_block_excl_max<BG_SPINE>(hp, lds);
    u32 all;
    u32 run_sum = block_excl_add<BG_SPINE>(sum, lds, all);
    for (u32 t = t0; t < t1; t += 8) {
        u32 h[8], c[8];
#pragma unroll
        for (u32 k = 0; k < 8; k++) {
            const u32 i = t + k < last ? t + k : last;
            h[k] = tile_head[i];
            c[k] = tile_keep[i];
        }
#pragma unroll
        for (u32 k = 0; k < 8; k++)
            if (t + k < t1) {  // entries at and beyond t1 belong to the next thread: read (clamped), never written
                tile_head[t + k] = run_hp;
                tile_keep[t + k] = run_sum;
                run_hp = h[k] > run_hp ? h[k] : run_hp;
                run_sum += c[k];
            }
    }
    if (threadIdx.x == 0) *total = all;
}

template <bool DOUBLING>
__global__ void __launch_bounds__(BW_BLOCK) k_bg_apply(const u64 * __restrict__ keys, const u32 * __restrict__ vals, const u32 * __restrict__ slots, u32 m,
                                                      const u32 * __restrict__ tile_head, const u32 * __restrict__ tile_keep, u32 * __restrict__ sa,
                                                      u32 * __restrict__ isa, u32 * __restrict__ vals_out, u32 * __restrict__ slots_out, u32 * __restrict__ grp_out) {
    __shared__ u32 lds[BW_BLOCK / WAVE + 1];
    __shared__ u32 st_v[BG_TILE], st_s[BG_TILE], st_g[BG_TILE];  // the tile's part of the compacted list, staged so that it leaves coalesced
    const u64 base = (u64)blockIdx.x * BG_TILE + (u64)threadIdx.x * BG_ITEMS;
    const u64 lbase = base < m ? base : (u64)m - 1;  // threads past the end load something valid and use none of it
    u64 kv[BG_ITEMS + 2];
    u32 v[BG_ITEMS], sl[BG_ITEMS];
    bg_load_keys(keys, m, lbase, kv);
    bg_load_u32(vals, m, lbase, v);
    if (slots) {
        bg_load_u32(slots, m, lbase, sl);
    } else {
#pragma unroll
        for (int j = 0; j < BG_ITEMS; j++) sl[j] = (u32)(base + j);
    }
    const u32 carry_hp = tile_head[blockIdx.x], tile_off = tile_keep[blockIdx.x];
    // this thread's own last head and keep count
    u32 hp = 0, keep = 0;
#pragma unroll
    for (int j = 0; j < BG_ITEMS; j++) {
        const u64 k = base + j;
        if (k < m) {
            BG_FLAGS(j, k, head, uniq);
            if (head) hp = (u32)k + 1u;
            keep += uniq ? 0u : 1u;
        }
    }
    u32 run_hp = bg_block_excl_max<BW_BLOCK>(hp, lds);
    run_hp = carry_hp > run_hp ? carry_hp : run_hp;
    u32 all;  // elements of this tile that stay active
    u32 out = block_excl_add<BW_BLOCK>(keep, lds, all);  // this thread's first slot in the tile's part of the compacted list
    // rank of an element = slot of its group's head.  The head of the group that reaches into this thread's elements from the left
    // costs one gather; from the first head on, the slots are in registers (sl[j] = k itself in the first round).
    u32 run_rank = 0;  // run_hp == 0 only where element 0 of the list starts the thread, and that one is a head
    if (run_hp > 0 && base < m) run_rank = slots ? slots[run_hp - 1u] : run_hp - 1u;
    u32 rank[BG_ITEMS];
#pragma unroll
    for (int j = 0; j < BG_ITEMS; j++) {
        const u64 k = base + j;
        rank[j] = 0;
        if (k < m) {
            BG_FLAGS(j, k, head, uniq);
            (void)uniq;
            if (head) run_rank = sl[j];
            rank[j] = run_rank;
        }
    }
#pragma unroll
    for (int j = 0; j < BG_ITEMS; j++) {
        const u64 k = base + j;
        if (k < m) {
            BG_FLAGS(j, k, head, uniq);
            (void)head;
            if (!DOUBLING || rank[j] != (u32)(kv[j + 1] >> 32)) isa[v[j]] = rank[j];
            if (uniq) {
                sa[sl[j]] = v[j];
            } else {
                st_v[out] = v[j];
                st_s[out] = sl[j];
                st_g[out] = rank[j];
                out++;
            }
        }
    }
    __syncthreads();
    // (a thread's kept elements are consecutive slots: written directly, every store instruction of a wave would touch 64 scattered
    // words; from LDS consecutive lanes write consecutive words)
    for (u32 idx = threadIdx.x; idx < all; idx += BW_BLOCK) {
        vals_out[tile_off + idx] = st_v[idx];
        slots_out[tile_off + idx] = st_s[idx];
        grp_out[tile_off + idx] = st_g[idx];
    }
}
#undef BG_FLAGS

// key(k) = (current rank of suffix s, carried along by k_bg_apply) << 32 | rank of suffix s + h
__global__ void __launch_bounds__(BW_BLOCK) k_bwt_doubling_keys_grp(const u32 * __restrict__ vals, const u32 * __restrict__ grp, const u32 * __restrict__ isa, u32 m,
                                                                   u32 n, u32 h, u64 * __restrict__ keys) {
    const u32 k = blockIdx.x * BW_BLOCK + threadIdx.x;
    if (k >= m) return;
    const u32 s = vals[k];
    const u64 j = (u64)s + h;
    const u32 lo = (j < n) ? isa[j] + h : (n - 1u - s);
    keys[k] = ((u64)grp[k] << 32) | lo;
}

__global__ void __launch_bounds__(BW_BLOCK) k_bwt_emit(const u8 * __restrict__ t, const u32 * __restrict__ sa, const u32 * __restrict__ isa, u32 n,
                                                      u8 * __restrict__ out, u32 * __restrict__ idx_out) {
    const u32 i = blockIdx.x * BW_BLOCK + threadIdx.x;
    if (i >= n) return;
    const u32 i0 = isa[0];
    if (i == 0) {
        out[0] = t[n - 1];
        *idx_out = i0 + 1;
    }
    if (i != i0) out[i < i0 ? i + 1 : i] = t[sa[i] - 1];
}

static int bits_for(u64 x) {
    int b = 0;
    while (x) { b++; x >>= 1; }
    return b ? b : 1;
}

size_t bwt_workspace_bytes(u64 n) {
    // SA + ISA + 2 keys + 2 vals + 2 slots + 2 scan arrays + radix temp + slack
    return n * (4 + 4 + 16 + 8 + 8 + 8) + radix_temp_bytes(n) + scan_temp_words(n) * 4 + (1u << 20) + 4096;
}

s32 bwt_forward(const u8 * d_in, u32 n, u8 * d_out, Arena & tmp, hipStream_t s, BwtStats * stats) {
    if (n == 0) return 0;
    if (n == 1) {
        HIP_CHECK(hipMemcpyAsync(d_out, d_in, 1, hipMemcpyDeviceToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));
        return 1;
    }
    const size_t mk = tmp.mark();
    u32 * sa = tmp.take<u32>(n);
    u32 * isa = tmp.take<u32>(n);
    u64 * key[2] = {tmp.take<u64>(n), tmp.take<u64>(n)};
    u32 * val[2] = {tmp.take<u32>(n), tmp.take<u32>(n)};
    u32 * slot[2] = {tmp.take<u32>(n), tmp.take<u32>(n)};
    u32 * scanA = tmp.take<u32>(n);
    u32 * headslot = tmp.take<u32>(n);
    u32 * d_words = tmp.take<u32>(4);  // [0] = scan total, [1] = primary index
    u32 * sym = tmp.take<u32>(264);    // byte value -> rank among the present values; [256] = how many are present
    BwtStats st;

    auto grid = [](u64 m) { return dim3((u32)((m + BW_BLOCK - 1) / BW_BLOCK)); };

    // ---- round 0: all suffixes by their 8-symbol prefix ------------------------------------------
    // bytes -> ranks among the byte values present (see k_bwt_sym_map): one LSD pass per bit of a rank instead of 8 passes
    int key_bits = 64;
    {
        HIP_CHECK(hipMemsetAsync(sym, 0, 264 * sizeof(u32), s));
        launch(k_bwt_sym_hist, grid(((u64)n + 63) / 64), dim3(BW_BLOCK), 0, s, d_in, n, sym);
        launch(k_bwt_sym_map, dim3(1), dim3(256), 0, s, sym);
        u32 present = 256;
        HIP_CHECK(hipMemcpyAsync(&present, sym + 256, 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        const int bits = bits_for(present);  // ranks 0 .. present
        static const bool plain = getenv("BZ3_BWT_PLAIN") != nullptr;  // experiments: always the plain-byte keys (8 passes)
        if (bits < 8 && !plain) {
            key_bits = 8 * bits;
            launch(k_bwt_prefix_keys_mapped, grid(((u64)n + 7) / 8), dim3(BW_BLOCK), 0, s, d_in, n, (const u32 *)sym, (u32)bits, key[0]);
        } else {  // (nearly) every byte value occurs: the bytes themselves, zero padded
            launch(k_bwt_prefix_keys, grid(((u64)n + 7) / 8), dim3(BW_BLOCK), 0, s, d_in, n, key[0]);
        }
    }
    int cur = 0;
    {
        // first pass generates the suffix numbers on the fly (iota), later passes carry them
        radix_pass<u64>(key[0], key[1], (const u32 *)nullptr, val[1], n, 0, 0xFFFFFFFFu, 0u, tmp, s);
        cur = 1;
        for (int shift = 8; shift < key_bits; shift += 8) {
            radix_pass<u64>(key[cur], key[cur ^ 1], (const u32 *)val[cur], val[cur ^ 1], n, shift, 0xFFFFFFFFu, 0u, tmp, s);
            cur ^= 1;
        }
        st.radix_passes += key_bits / 8;
        st.sorted_elements += n;
    }
    u32 m = n;              // elements in the current (sorted) active list
    const u32 * slots = nullptr;  // nullptr = identity (round 0 covers every slot)
    int scur = 0;           // slot[scur] holds the slots of the active list (when slots != nullptr)
    u32 h = 8;
    // experiments: BZ3_BWT_FUSED=1 = the two-pass regrouping (k_bg_*); read per call so that tests can switch it in-process
    const bool fused = getenv("BZ3_BWT_FUSED") != nullptr;
    if (fused) {
        u32 * vcur = val[cur], * vfree = val[cur ^ 1];  // suffixes of the sorted list / a free buffer of the same size
        const u32 max_tiles = (u32)(((u64)n + BG_TILE - 1) / BG_TILE);
        u32 * tile_head = scanA;  // the flag / scan buffer of the seven-launch form is free here
        u32 * tile_keep = scanA + max_tiles;  // 2 * ceil(n / 2048) <= n for every n >= 2
        u32 * grp = headslot;     // ranks of the compacted suffixes
        for (;;) {
            st.rounds++;
            const u32 tiles = (u32)(((u64)m + BG_TILE - 1) / BG_TILE);
            launch(k_bg_reduce, dim3(tiles), dim3(BW_BLOCK), 0, s, (const u64 *)key[cur], m, tile_head, tile_keep);
            launch(k_bg_spine, dim3(1), dim3(BG_SPINE), 0, s, tile_head, tile_keep, tiles, d_words);
            if (slots)
                launch(k_bg_apply<true>, dim3(tiles), dim3(BW_BLOCK), 0, s, (const u64 *)key[cur], (const u32 *)vcur, slots, m, (const u32 *)tile_head,
                       (const u32 *)tile_keep, sa, isa, vfree, slot[scur ^ 1], grp);
            else
                launch(k_bg_apply<false>, dim3(tiles), dim3(BW_BLOCK), 0, s, (const u64 *)key[cur], (const u32 *)vcur, slots, m, (const u32 *)tile_head,
                       (const u32 *)tile_keep, sa, isa, vfree, slot[scur ^ 1], grp);
            u32 m_next = 0;
            HIP_CHECK(hipMemcpyAsync(&m_next, d_words, 4, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
            if (m_next == 0) break;
            scur ^= 1;
            slots = slot[scur];
            m = m_next;
            // the active suffixes are in vfree now, their ranks in grp; the old list (key[cur], vcur) is dead
            launch(k_bwt_doubling_keys_grp, grid(m), dim3(BW_BLOCK), 0, s, (const u32 *)vfree, (const u32 *)grp, (const u32 *)isa, m, n, h, key[0]);
            u32 * vv[2] = {vfree, vcur};
            const int lo_bits = bits_for((u64)n + h);
            const int hi_bits = bits_for(n);
            int c = radix_sort_pairs<u64>(key[0], key[1], vv[0], vv[1], m, 0, lo_bits, tmp, s);
            c ^= radix_sort_pairs<u64>(key[c], key[c ^ 1], vv[c], vv[c ^ 1], m, 32, 32 + hi_bits, tmp, s);
            cur = c;
            vcur = vv[c];
            vfree = vv[c ^ 1];
            st.radix_passes += (lo_bits + 7) / 8 + (hi_bits + 7) / 8;
            st.sorted_elements += m;
            if (h >= 0x40000000u) throw HipError{hipErrorUnknown, "suffix sort did not converge", __FILE__, __LINE__};
            h *= 2;
        }
    } else
    for (;;) {
        st.rounds++;
        // ---- regroup the freshly sorted list, publish SA/ISA, drop the singletons ---------------
        launch(k_bwt_head_flags, grid(m), dim3(BW_BLOCK), 0, s, (const u64 *)key[cur], m, scanA);
        exclusive_scan_u32(scanA, m, nullptr, tmp, s);
        launch(k_bwt_head_slots, grid(m), dim3(BW_BLOCK), 0, s, (const u64 *)key[cur], (const u32 *)scanA, slots, m, headslot);
        u32 * keep = val[cur ^ 1];  // free buffer at this point
        if (slots)  // a doubling round: the keys carry the current ranks
            launch(k_bwt_assign<true>, grid(m), dim3(BW_BLOCK), 0, s, (const u64 *)key[cur], (const u32 *)val[cur], (const u32 *)scanA, slots, (const u32 *)headslot,
                   m, sa, isa, keep);
        else
            launch(k_bwt_assign<false>, grid(m), dim3(BW_BLOCK), 0, s, (const u64 *)key[cur], (const u32 *)val[cur], (const u32 *)scanA, slots, (const u32 *)headslot,
                   m, sa, isa, keep);
        exclusive_scan_u32(keep, m, d_words, tmp, s);
        u32 m_next = 0;
        HIP_CHECK(hipMemcpyAsync(&m_next, d_words, 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        if (m_next == 0) break;
        // compact into (scanA as vals, slot[scur^1]) -- scanA is free again after k_bwt_assign
        launch(k_bwt_compact, grid(m), dim3(BW_BLOCK), 0, s, (const u64 *)key[cur], (const u32 *)val[cur], slots, (const u32 *)keep, m, scanA,
               slot[scur ^ 1]);
        scur ^= 1;
        slots = slot[scur];
        m = m_next;
        // active suffix list now lives in scanA; move it into val[0] and build the doubling keys in key[0]
        HIP_CHECK(hipMemcpyAsync(val[0], scanA, (size_t)m * 4, hipMemcpyDeviceToDevice, s));
        launch(k_bwt_doubling_keys, grid(m), dim3(BW_BLOCK), 0, s, (const u32 *)val[0], (const u32 *)isa, m, n, h, key[0]);
        const int lo_bits = bits_for((u64)n + h);
        const int hi_bits = bits_for(n);
        cur = radix_sort_pairs<u64>(key[0], key[1], val[0], val[1], m, 0, lo_bits, tmp, s);
        int passes = (lo_bits + 7) / 8;
        {
            // continue on the high word from whichever buffer holds the data
            const int c2 = radix_sort_pairs<u64>(key[cur], key[cur ^ 1], val[cur], val[cur ^ 1], m, 32, 32 + hi_bits, tmp, s);
            cur ^= c2;
            passes += (hi_bits + 7) / 8;
        }
        st.radix_passes += passes;
        st.sorted_elements += m;
        if (h >= 0x40000000u) throw HipError{hipErrorUnknown, "suffix sort did not converge", __FILE__, __LINE__};
        h *= 2;
    }
    launch(k_bwt_emit, grid(n), dim3(BW_BLOCK), 0, s, d_in, (const u32 *)sa, (const u32 *)isa, n, d_out, d_words + 1);
    u32 idx = 0;
    HIP_CHECK(hipMemcpyAsync(&idx, d_words + 1, 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    tmp.release(mk);
    if (stats) *stats = st;
    return (s32)idx;
}

}  // namespace bz3
