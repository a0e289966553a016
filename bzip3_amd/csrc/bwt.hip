// bwt.hip -- forward Burrows-Wheeler transform of one block on gfx950.
// Replaces libsais_bwt (reference include/libsais.h:4095-4121, SA-IS: :3941-3983, :3740-3939), which is a sequential
// induced-sorting algorithm whose inner scans carry a dependency through 256 bucket cursors.  This is NOT a port of it.
//
// Design of rounds 3-4 ("sort once, then resolve groups where they lie"):
//
//   codes   : the bytes of the block are given an order-preserving prefix-free code of at most 8 bits per symbol, the optimal
//             height-limited alphabetic tree for the block's byte histogram (frequent bytes get short codes), built on the DEVICE
//             (k_vlc_init / k_vlc_level / k_vlc_codes; rounds 1-3: on the host).  Comparing the concatenated code bits of two suffixes
//             is the same as comparing their bytes, and a 56-bit window holds 7 symbols at least and ~12 of English text.
//   round 0 : key(i) = first 56 code bits of suffix i (16 symbols at most, zero padded past the end) << 8 | T[i-1]; ONE stable LSD
//             radix sort of all n (key, i) pairs over the upper 56 bits (7 passes of sort.hip).  The byte that precedes the suffix
//             rides in the low byte: it is the BWT symbol of the suffix, so the output needs no gather through the suffix array.
//   groups  : suffixes with equal 56-bit windows form a group of neighbouring slots.  V[slot] = suffix | head flag (bit 31);
//             k_bwt_heads marks them and snapshots the flags, k_bwt_spine_a / _b give every anchor tile of 512 slots (WR_A) the last
//             head before it.
//   route   : k_bwt_route -- a wave per anchor tile sends the groups that start there to one of three lists: up to 64 members -> the
//             tail list (one entry per suffix), 65 .. 256 members (WR_G) -> a descriptor for the wide kernel, more -> the big list.
//   wide    : k_bwt_wide -- a wave per descriptor, 2 or 4 suffixes per lane in registers: a step fetches the next 40 code bits of
//             every still-ambiguous suffix straight from the text (no inverse suffix array, no rank table: groups are independent of
//             each other), a bitonic network over the wave's registers sorts all sub-groups at once; what is down to <= 64 members
//             moves on to the tail list.
//   tail    : k_bwt_tail -- a wave per 64 list entries takes the whole groups that start there, a lane per suffix, ranks groups of up
//             to 12 by counting and larger ones with a 64-lane bitonic network, up to TL_CAP steps.  Text needs 2-3 steps.
//             The wide and the tail kernel read the lengths of their lists from the router's counters on the device (round 4).
//   big     : groups of > 256 suffixes (a few % of text) get one more 56-bit window each through the global radix sorter
//             (k_big_*), after which they are small and are routed again.
//   deep    : what is still ambiguous after that (long repeats: runs, periodic data) falls back to classic prefix doubling on
//             ranks (ISA built once, k_bg_* / k_isa_scatter / k_bwt_doubling_keys_grp) -- the only path that needs random 4-byte scatters.
//   output  : U[0] = T[n-1]; U[i < i0 ? i+1 : i] = payload byte of slot i for i != i0 = slot of suffix 0; idx = i0+1.
//
// Order of equal windows past the end of the block: a suffix that is a proper prefix of another sorts first (zero padding is the
// smallest continuation, and a suffix with no symbol left at its depth sorts before every live one, shorter first) -- exactly the
// order libsais produces.
//
// HBM per block of n bytes (carved from the per-device workspace): 2 x key u64[n], 2 x suffix u32[n], payload u8[n]; the deep path
// additionally ISA u32[n], 2 x slot u32[n], rank u32[n], tile words.  Algorithmic traffic: 11 B per input byte.
#include <atomic>
#include <vector>

#include "prims.hpp"
#include "sort.hpp"
#include "stages.hpp"

namespace bz3 {

constexpr int BW_BLOCK = 256;
constexpr u32 V_HEAD = 0x80000000u;  // slot starts a group
constexpr u32 V_MASK = 0x3FFFFFFFu;  // suffix number (n < 2^30: bz3_bound(511 MiB) = 546.5 M)

// Bytes of the block that are NOT among its `keep` most frequent byte values (one workgroup of 256: thread c ranks the count of byte value c).
// The BWT output has the block's histogram, and the CM stage's row-cache kernels hold ~40 order-1 rows per block: api.hip routes a block
// whose rarer values make up more than half of the row misses those kernels tolerate straight to the whole-model kernels (round 5; before, the row-cache kernels had to give it up first).
__global__ void __launch_bounds__(256) k_bwt_outside(const u32 * __restrict__ hist, u32 keep, u32 * __restrict__ out) {
    __shared__ u32 h[256];
    __shared__ u32 red[256 / WAVE + 1];
    const u32 c = threadIdx.x;
    h[c] = hist[c];
    __syncthreads();
    const u32 mine = h[c];
    u32 rank = 0;  // byte values that come before c in (count descending, value ascending) order
    for (u32 k = 0; k < 256u; k++) rank += (h[k] > mine || (h[k] == mine && k < c)) ? 1u : 0u;
    const u32 tot = block_sum<256>(rank >= keep ? mine : 0u, red);
    if (c == 0) *out = tot;
}

// ---- the order-preserving code ----------------------------------------------------------------------------------------------
// vlc[c] = code << 4 | len (1 <= len <= 8) for byte values present in the block, 0 otherwise.
constexpr int VLC_MAXLEN = 8;
constexpr int VLC_WINDOW = 16;  // symbols a window looks at

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

// Optimal alphabetic (order-preserving) prefix-free code of height <= 8 for the byte values present, by dynamic programming over
// intervals: cost[l][i][j] = lightest tree of height <= l over the present values i .. j-1 = min over the split k of
// cost[l-1][i][k] + cost[l-1][k][j], plus the weight of the interval.  On the DEVICE since round 4 (rounds 1-3 copied the histogram to
// the host, ran the recurrence there in ~1 ms and copied the table back: a stream synchronisation and a millisecond of idle GPU per
// block, and a front end that slowed down with the host): a level is one launch with a thread per interval -- every cell of a level
// only reads the level below --, the tables (two cost planes of 257 x 257 u64, nine root planes of u16) live in the block's arena.
// Which optimal tree comes out does not matter to the BWT (any order-preserving prefix code sorts the suffixes the same way): ties
// go to the smallest split.
constexpr int VLC_S1 = 257;                         // intervals [i, j) over up to 256 present values: 0 <= i < j <= 256
constexpr u64 VLC_INF = ~0ull >> 2;
constexpr size_t VLC_PLANE = (size_t)VLC_S1 * VLC_S1;
struct VlcWork {
    u32 s;            // byte values present
    u32 sym[256];     // ... in increasing order
    u64 pre[257];     // prefix sums of their counts
};
constexpr size_t VLC_WORK_BYTES = ((sizeof(VlcWork) + 255) & ~(size_t)255) + 2 * VLC_PLANE * sizeof(u64) + (VLC_MAXLEN + 1) * VLC_PLANE * sizeof(u16) + 256;

// the present values, their prefix sums, level 0 (single values cost nothing, nothing else fits height 0), the trivial tables
__global__ void __launch_bounds__(256) k_vlc_init(const u32 * __restrict__ hist, VlcWork * __restrict__ w, u64 * __restrict__ cost0, u32 * __restrict__ table) {
    __shared__ u32 lds[256 / WAVE + 1];
    __shared__ u64 s_cnt[256];
    const u32 c = threadIdx.x;
    const u32 cnt = hist[c];
    u32 total;
    const u32 at = block_excl_add<256>(cnt ? 1u : 0u, lds, total);
    s_cnt[c] = 0;
    __syncthreads();
    if (cnt) {
        w->sym[at] = c;
        s_cnt[at] = cnt;
    }
    table[c] = (total == 1u && cnt) ? ((0u << 4) | 1u) : 0u;  // one value: code 0, one bit; the others are filled in by k_vlc_codes
    __syncthreads();
    if (c == 0) {
        w->s = total;
        u64 run = 0;
        for (u32 k = 0; k <= 256u; k++) {
            w->pre[k] = run;
            if (k < 256u) run += s_cnt[k];
        }
    }
    for (u32 i = 0; i < (u32)VLC_S1; i++) {  // row i of the level-0 plane
        const u32 j0 = c, j1 = c + 256u;
        cost0[(size_t)i * VLC_S1 + j0] = j0 == i + 1u ? 0ull : VLC_INF;
        if (j1 < (u32)VLC_S1) cost0[(size_t)i * VLC_S1 + j1] = j1 == i + 1u ? 0ull : VLC_INF;
    }
}
// level l from level l-1: workgroup i, thread j-1
__global__ void __launch_bounds__(256) k_vlc_level(const VlcWork * __restrict__ w, u32 l, const u64 * __restrict__ below, u64 * __restrict__ cost, u16 * __restrict__ root) {
    const u32 s = w->s;
    const u32 i = blockIdx.x, j = threadIdx.x + 1u;
    if (i >= s) return;
    u64 best = VLC_INF;
    u32 bk = 0;
    if (j > i && j <= s) {
        const u32 len = j - i, half = 1u << (l - 1u);
        if (len == 1u) {
            best = 0;
        } else if (len <= 2u * half) {
            const u32 klo = j > i + half ? j - half : i + 1u;      // both parts must fit height l-1: at most 2^(l-1) values each
            const u32 khi = i + half < j - 1u ? i + half : j - 1u;
            for (u32 k = klo; k <= khi; k++) {
                const u64 a = below[(size_t)i * VLC_S1 + k], b = below[(size_t)k * VLC_S1 + j];
                if (a < VLC_INF && b < VLC_INF && a + b < best) {
                    best = a + b;
                    bk = k;
                }
            }
            if (best < VLC_INF) best += w->pre[j] - w->pre[i];
        }
    }
    if (j > i && j < (u32)VLC_S1) {
        cost[(size_t)i * VLC_S1 + j] = best;
        root[(size_t)i * VLC_S1 + j] = (u16)bk;
    }
}
// every present value walks down from the root of the height-8 tree: table[c] = code << 4 | length
__global__ void __launch_bounds__(256) k_vlc_codes(const VlcWork * __restrict__ w, const u16 * __restrict__ roots, u32 * __restrict__ table) {
    const u32 s = w->s, x = threadIdx.x;
    if (s < 2u || x >= s) return;
    u32 i = 0, j = s, l = VLC_MAXLEN, code = 0, len = 0;
    while (j - i > 1u) {
        const u32 k = roots[(size_t)l * VLC_PLANE + (size_t)i * VLC_S1 + j];
        if (x < k) {
            j = k;
            code <<= 1;
        } else {
            i = k;
            code = (code << 1) | 1u;
        }
        l--;
        len++;
    }
    table[w->sym[x]] = (code << 4) | len;
}
static void vlc_build_device(const u32 * d_hist, u32 * d_table, Arena & tmp, hipStream_t s) {
    char * base = tmp.take<char>(VLC_WORK_BYTES);
    VlcWork * w = reinterpret_cast<VlcWork *>(base);
    u64 * cost[2] = {reinterpret_cast<u64 *>(base + ((sizeof(VlcWork) + 255) & ~(size_t)255)), nullptr};
    cost[1] = cost[0] + VLC_PLANE;
    u16 * roots = reinterpret_cast<u16 *>(cost[1] + VLC_PLANE);
    launch(k_vlc_init, dim3(1), dim3(256), 0, s, d_hist, w, cost[0], d_table);
    for (u32 l = 1; l <= (u32)VLC_MAXLEN; l++)
        launch(k_vlc_level, dim3(256), dim3(256), 0, s, (const VlcWork *)w, l, (const u64 *)cost[(l - 1) & 1], cost[l & 1], roots + (size_t)l * VLC_PLANE);
    launch(k_vlc_codes, dim3(1), dim3(256), 0, s, (const VlcWork *)w, (const u16 *)roots, d_table);
}

// The first B code bits of the symbols in the window (wa, wb: 16 bytes, little endian; avail <= 16 of them exist), zero padded;
// cnt = symbols that lie entirely inside those bits.
template <int B, int SYMS = VLC_WINDOW>
__device__ __forceinline__ void vlc_pack(const u32 * __restrict__ tab, u64 wa, u64 wb, u32 avail, u64 & key, u32 & cnt) {
    u64 acc = 0;
    u32 bits = 0;
    cnt = 0;
#pragma unroll
    for (u32 k = 0; k < (u32)SYMS; k++) {
        const u32 e = tab[(u8)(k < 8 ? wa >> (8 * k) : wb >> (8 * (k - 8)))];
        const u32 len = e & 15u;
        const bool take = k < avail && bits < (u32)B;
        acc = take ? ((acc << len) | (u64)(e >> 4)) : acc;
        bits = take ? bits + len : bits;
        cnt = (take && bits <= (u32)B) ? k + 1u : cnt;
    }
    key = bits >= (u32)B ? (acc >> (bits - (u32)B)) : (acc << ((u32)B - bits));
}

struct __attribute__((packed)) PackedU64 { u64 v; };

// The 16 bytes at t[pos ..] as two little-endian words (any alignment); avail = how many of them exist (bytes past the end read as
// zero).  Branch-free for n >= 16, so that the loads of all the elements a thread handles are in flight together: the address is
// clamped to the last 16 bytes of the block and the words are shifted down by the difference.
__device__ __forceinline__ void load_window(const u8 * __restrict__ t, u64 pos, u64 n, u64 & a, u64 & b, u32 & avail) {
    avail = pos >= n ? 0u : (n - pos < (u64)VLC_WINDOW ? (u32)(n - pos) : (u32)VLC_WINDOW);
    if (n >= (u64)VLC_WINDOW) {
        const u64 p2 = pos + VLC_WINDOW <= n ? pos : n - VLC_WINDOW;
        u64 lo = reinterpret_cast<const PackedU64 *>(t + p2)->v;
        u64 hi = reinterpret_cast<const PackedU64 *>(t + p2 + 8)->v;
        const u64 delta = pos - p2;          // bytes to drop from the front
        u32 sh = delta >= 16 ? 128u : (u32)delta * 8u;
        if (sh >= 64u) {
            lo = hi;
            hi = 0;
            sh -= 64u;
        }
        if (sh >= 64u) {
            lo = 0;
            sh = 0;
        }
        if (sh) {
            lo = (lo >> sh) | (hi << (64u - sh));
            hi >>= sh;
        }
        a = lo;
        b = hi;
    } else {  // tiny block: byte by byte
        a = b = 0;
        for (u32 k = 0; k < avail; k++) {
            const u64 c = t[pos + k];
            if (k < 8) a |= c << (8 * k); else b |= c << (8 * (k - 8));
        }
    }
}
__device__ __forceinline__ u8 window_byte(u64 a, u64 b, u32 k) { return (u8)(k < 8 ? a >> (8 * k) : b >> (8 * (k - 8))); }

// keys[i] = (first 56 code bits of suffix i) << 8 | T[i-1]; 8 positions per thread from 24 bytes.
__global__ void __launch_bounds__(BW_BLOCK) k_bwt_vlc_keys(const u8 * __restrict__ t, u32 n, const u32 * __restrict__ vlc, u64 * __restrict__ keys) {
    __shared__ u32 tab[256];
    tab[threadIdx.x] = vlc[threadIdx.x];
    __syncthreads();
    const u64 base = ((u64)blockIdx.x * BW_BLOCK + threadIdx.x) * 8;
    if (base >= n) return;
    const u64 last = (u64)n - 1;
    u8 b[24];  // b[k] = t[base - 1 + k]
#pragma unroll
    for (int k = 0; k < 24; k++) {
        const u64 i = base + k;  // index + 1
        b[k] = t[i == 0 ? 0 : (i - 1 < n ? i - 1 : last)];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const u64 p = base + k;
        if (p < n) {
            u64 wa = 0, wb = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                wa |= (u64)b[k + 1 + q] << (8 * q);
                wb |= (u64)b[k + 9 + q] << (8 * q);
            }
            const u64 left = n - p;
            u64 key;
            u32 cnt;
            vlc_pack<56>(tab, wa, wb, left < (u64)VLC_WINDOW ? (u32)left : (u32)VLC_WINDOW, key, cnt);
            keys[p] = (key << 8) | (p == 0 ? 0ull : (u64)b[k]);
        }
    }
}

// ---- groups of the sorted list ----------------------------------------------------------------------------------------------
// Resolve-kernel geometry: a workgroup owns the groups whose head lies in its RS_S anchor slots; such a group of <= TR_G suffixes
// ends inside the TR_WIN-slot window.
constexpr int WR_A = 512;     // anchor slots per wave
constexpr int WR_G = 256;     // largest group a wave resolves (4 suffixes per lane; with 8 the wide kernel needs 200 registers: two waves per SIMD)
constexpr int WR_WAVES = 4;   // waves per workgroup (independent of each other after the code table is loaded)
constexpr int WR_CAP = 160;   // resolve steps before a group is left to the deep path
constexpr int TL_CAP = 160;   // tail steps before a group is left to the deep path (>= 800 symbols)
constexpr u32 WR_FAR = 0xFFFFu;
constexpr int TR_S = WR_A;    // granularity of tile_last / carry / dirty
constexpr int TR_G = WR_G;

// V[p] = suffix | head flag, PB[p] = payload byte, tile_last[t] = 1 + slot of the last head in anchor tile t (0: none), pos0, and
// the SNAPSHOT of the head flags as a bitmap: the resolve kernel takes the group boundaries from the snapshot, because the flags in V
// change under it (a neighbouring workgroup splits its own groups while this one is still reading its window).
__global__ void __launch_bounds__(BW_BLOCK) k_bwt_heads(const u64 * __restrict__ keys, u32 * __restrict__ v, u8 * __restrict__ pb, u32 n,
                                                       u32 * __restrict__ tile_last, u32 * __restrict__ hbits, u32 * __restrict__ counters) {
    __shared__ u32 lds[BW_BLOCK / WAVE + 1];
    const u64 tbase = (u64)blockIdx.x * TR_S;
    u32 hp = 0;
    for (u32 q = threadIdx.x; q < (u32)TR_S; q += BW_BLOCK) {  // TR_S is a multiple of the workgroup size: uniform trip count
        const u64 p = tbase + q;
        bool head = true;  // past the end: heads
        if (p < n) {
            const u64 k = keys[p];
            head = p == 0 || (k >> 8) != (keys[p - 1] >> 8);
            const u32 s = v[p];
            v[p] = s | (head ? V_HEAD : 0u);
            pb[p] = (u8)k;
            if (head) hp = (u32)p + 1u;
            if (s == 0) counters[2] = (u32)p;
        }
        const u64 bal = __ballot(head);
        if (lane_id() == 0 && p - (p & 63u) < (((u64)n + 63u) & ~63ull)) {
            hbits[p >> 5] = (u32)bal;
            hbits[(p >> 5) + 1] = (u32)(bal >> 32);
        }
    }
    hp = block_max<BW_BLOCK>(hp, lds);
    if (threadIdx.x == 0) tile_last[blockIdx.x] = hp;
}

// The same reduction and snapshot from the flags in V (after a big round changed them).
__global__ void __launch_bounds__(BW_BLOCK) k_bwt_reduce_heads(const u32 * __restrict__ v, u32 n, u32 * __restrict__ tile_last, u32 * __restrict__ hbits) {
    __shared__ u32 lds[BW_BLOCK / WAVE + 1];
    const u64 tbase = (u64)blockIdx.x * TR_S;
    u32 hp = 0;
    for (u32 q = threadIdx.x; q < (u32)TR_S; q += BW_BLOCK) {
        const u64 p = tbase + q;
        bool head = true;
        if (p < n) {
            head = (v[p] & V_HEAD) != 0u;
            if (head) hp = (u32)p + 1u;
        }
        const u64 bal = __ballot(head);
        if (lane_id() == 0 && p - (p & 63u) < (((u64)n + 63u) & ~63ull)) {
            hbits[p >> 5] = (u32)bal;
            hbits[(p >> 5) + 1] = (u32)(bal >> 32);
        }
    }
    hp = block_max<BW_BLOCK>(hp, lds);
    if (threadIdx.x == 0) tile_last[blockIdx.x] = hp;
}

// carry of tile t = 1 + slot of the last head before the tile = max(carry_local[t], group_carry[t / SP_GROUP]) (0: none):
//   k_bwt_spine_a : one workgroup per SP_GROUP tiles, exclusive running maximum inside the group + the group's maximum
//   k_bwt_spine_b : one workgroup, exclusive running maximum over the groups (at most ~1000 of them)
// (one workgroup over all 520 k tiles of a 256 MiB block took as long as a radix pass)
constexpr int SP_BLOCK = 256;
constexpr int SP_PER = 4;
constexpr int SP_GROUP = SP_BLOCK * SP_PER;
template <int BLOCK>
__device__ __forceinline__ u32 sp_block_excl_max(u32 v, u32 * lds, u32 & all) {  // lds: BLOCK / 64 + 1 words; ends with a barrier
    const u32 incl = wave_incl_max(v);
    if (lane_id() == WAVE - 1) lds[wave_id()] = incl;
    u32 up = __shfl_up(incl, 1u);
    if (lane_id() == 0) up = 0u;
    __syncthreads();
    u32 carry = 0, tot = 0;
    for (int w = 0; w < BLOCK / WAVE; w++) {
        if (w < wave_id()) carry = lds[w] > carry ? lds[w] : carry;
        tot = lds[w] > tot ? lds[w] : tot;
    }
    __syncthreads();
    all = tot;
    return up > carry ? up : carry;
}
__global__ void __launch_bounds__(SP_BLOCK) k_bwt_spine_a(const u32 * __restrict__ tile_last, u32 tiles, u32 * __restrict__ carry_local, u32 * __restrict__ group_max) {
    __shared__ u32 lds[SP_BLOCK / WAVE + 1];
    const u32 t0 = blockIdx.x * SP_GROUP + threadIdx.x * SP_PER;
    u32 h[SP_PER];
#pragma unroll
    for (int k = 0; k < SP_PER; k++) h[k] = t0 + k < tiles ? tile_last[t0 + k] : 0u;
    u32 mine = 0;
#pragma unroll
    for (int k = 0; k < SP_PER; k++) mine = h[k] > mine ? h[k] : mine;
    u32 all;
    u32 run = sp_block_excl_max<SP_BLOCK>(mine, lds, all);
#pragma unroll
    for (int k = 0; k < SP_PER; k++) {
        if (t0 + k < tiles) carry_local[t0 + k] = run;
        run = h[k] > run ? h[k] : run;
    }
    if (threadIdx.x == 0) group_max[blockIdx.x] = all;
}
__global__ void __launch_bounds__(1024) k_bwt_spine_b(const u32 * __restrict__ group_max, u32 groups, u32 * __restrict__ group_carry) {
    __shared__ u32 lds[1024 / WAVE + 1];
    u32 run_in = 0;  // maximum over the chunks of 1024 groups before this one (one chunk for every block size the API allows)
    for (u32 base = 0; base < groups; base += 1024u) {
        const u32 g = base + threadIdx.x;
        const u32 x = g < groups ? group_max[g] : 0u;
        u32 all;
        const u32 ex = sp_block_excl_max<1024>(x, lds, all);
        if (g < groups) group_carry[g] = ex > run_in ? ex : run_in;
        run_in = all > run_in ? all : run_in;
    }
}

// ---- the resolve kernel ------------------------------------------------------------------------------------------------------
// One WAVE per 512 anchor slots, no workgroup barriers.  The wave owns the groups whose head lies in its anchor slots; it takes them
// in batches of whole groups that are neighbours in the slot order, at most 512 slots per batch (8 per lane, blocked: position
// j = lane * E + r), and resolves a batch completely before it goes on:
//   step  : every suffix that still shares its group fetches the next 40 code bits at its depth straight from the text (16 bytes, all
//           loads of a lane in flight together) -- no inverse suffix array, no rank table: groups are independent of each other;
//           sort word = [group start : 9][live : 1][40 code bits, or the length of a suffix that has ended][position : 9];
//           a bitonic network over the wave's registers (strides below E inside a lane, the others by cross-lane exchange) sorts all
//           groups of the batch at once; the payloads (suffix, depth, BWT symbol) follow through a per-wave LDS buffer; positions
//           whose word differs from their left neighbour's become group heads.
//   shrink: once the suffixes still ambiguous fit half the lanes' capacity, the final ones are written out and the rest move up
//           (E halves: a 64-element network costs a tenth of a 512-element one), so long tails of tiny deep groups stay cheap.
// Round 3 measured the first, workgroup-wide version of this kernel (2048-slot windows, block scans and barriers between the phases
// of a step) at 40 ms per 256 MiB block, nearly all of it waiting at barriers with 16 waves per CU; a wave-level step is the same
// work without a single barrier, and the small footprint (~5 KB of LDS per wave) lets many waves cover each other's gathers.
__device__ __forceinline__ u32 bw_readlane(u32 v, int lane) {
#ifdef BZ3_EMU
    return __shfl(v, lane);
#else
    return (u32)__builtin_amdgcn_readlane((int)v, lane);
#endif
}
// Sort word layout
constexpr int WW_IDX = 9, WW_KEY = 40;
__device__ __forceinline__ u64 ww_make(u32 gs, bool live, u64 key, u32 idx) {
    return ((u64)gs << (WW_IDX + WW_KEY + 1)) | ((live ? 1ull : 0ull) << (WW_IDX + WW_KEY)) | (key << WW_IDX) | (u64)idx;
}
// payload: [suffix : 30][depth : 16][BWT symbol : 8]
__device__ __forceinline__ u64 pl_make(u32 v, u32 d, u32 p) { return ((u64)v << 24) | ((u64)(d & 0xFFFFu) << 8) | (u64)(p & 0xFFu); }
__device__ __forceinline__ u32 pl_v(u64 x) { return (u32)(x >> 24); }
__device__ __forceinline__ u32 pl_d(u64 x) { return (u32)(x >> 8) & 0xFFFFu; }
__device__ __forceinline__ u32 pl_p(u64 x) { return (u32)x & 0xFFu; }

// Bitonic sort of the 64 * E words held by the wave (position j = lane * E + r), ascending.
template <int E>
__device__ __forceinline__ void wave_bitonic(u64 (&w)[E]) {
    const u32 lane = (u32)lane_id();
#pragma unroll
    for (u32 k = 2; k <= 64u * E; k <<= 1) {
#pragma unroll
        for (u32 j = k >> 1; j >= 1; j >>= 1) {
            if (j >= (u32)E) {  // partner in another lane
                const u32 lj = j / E;
                const bool upper = (lane & lj) != 0u;
                const bool asc = ((lane * E) & k) == 0u;  // k >= 2 E here: the bit lies in the lane number
#pragma unroll
                for (int r = 0; r < E; r++) {
                    const u64 y = __shfl_xor(w[r], (int)lj);
                    const u64 lo = w[r] < y ? w[r] : y, hi = w[r] < y ? y : w[r];
                    w[r] = (upper == asc) ? hi : lo;
                }
            } else {  // partner in the same lane
#pragma unroll
                for (int r = 0; r < E; r++) {
                    if (!(r & j)) {
                        const bool asc = ((lane * E + r) & k) == 0u;
                        const u64 x = w[r], y = w[r | j];
                        const u64 lo = x < y ? x : y, hi = x < y ? y : x;
                        w[r] = asc ? lo : hi;
                        w[r | j] = asc ? hi : lo;
                    }
                }
            }
        }
    }
}

// per-wave LDS
struct WrLds {
    u64 pl[WR_G];       // payloads by position (permutation / compaction buffer)
    u16 aux[WR_G];      // slot of a position (offset from the window start)
    u8 hd[WR_G];        // compaction: head flags of the positions
};

// smallest set bit position >= p in the wave's window bitmap (33 words), WR_FAR if none
__device__ __forceinline__ u32 wr_next_head(const u32 * hb, u32 p) {
    const u32 lane = (u32)lane_id();
    u32 word = lane < 33u ? hb[lane] : 0u;
    const u32 wp = p >> 5;
    if (lane < wp) word = 0;
    else if (lane == wp) word &= 0xFFFFFFFFu << (p & 31u);
    const u64 any = __ballot(word != 0u);
    if (!any) return WR_FAR;
    const int l = __ffsll((unsigned long long)any) - 1;
    const u32 wv = bw_readlane(word, l);
    return (u32)l * 32u + (u32)(__ffs(wv) - 1);
}
// largest set bit position <= t
__device__ __forceinline__ u32 wr_prev_head(const u32 * hb, u32 t) {
    const u32 lane = (u32)lane_id();
    u32 word = lane < 33u ? hb[lane] : 0u;
    const u32 wp = t >> 5;
    if (lane > wp) word = 0;
    else if (lane == wp) word &= 0xFFFFFFFFu >> (31u - (t & 31u));
    const u64 any = __ballot(word != 0u);
    if (!any) return WR_FAR;
    const int l = 63 - __clzll((unsigned long long)any);
    const u32 wv = bw_readlane(word, l);
    return (u32)l * 32u + (31u - (u32)__clz(wv));
}

struct WrCtx {
    const u8 * t;
    u32 n;
    u32 * v;
    u8 * pb;
    const u32 * tab;
    u32 * counters;
    u32 chain;
    u64 slot0;  // global slot of window position 0
    u32 * tail_v;
    u32 * tail_slot;
    u16 * tail_d;
    u8 * tail_pb;
    u32 tail_cap;
};

// One batch: window positions [c, c + L) (whole groups, L <= 512), E = suffixes per lane.  Returns when every group is resolved (or the
// step cap is reached: the groups are written back as they are and counted as given up).
template <int E>
__device__ __forceinline__ u32 wr_run(const WrCtx & cx, WrLds & lds, u64 (&pl)[E], u32 hm, u32 L, u32 step);

// the `total` ambiguous suffixes a shrink left in the wave's LDS buffer, E2 per lane
template <int E2>
__device__ __forceinline__ u32 wr_reload(const WrCtx & cx, WrLds & lds, u32 total, u32 step) {
    const u32 lane = (u32)lane_id();
    u64 q[E2];
    u32 h2 = 0;
#pragma unroll
    for (int r = 0; r < E2; r++) {
        const u32 j = lane * E2 + r;
        const bool in = j < total;
        q[r] = in ? lds.pl[j] : 0ull;
        h2 |= (in ? (u32)lds.hd[j] : 1u) << r;
    }
    wave_sync();
    return wr_run<E2>(cx, lds, q, h2, total, step);
}

// Returns the number of suffixes it left in lds.pl / aux / hd for the tail list (0: the batch is done).
template <int E>
__device__ __forceinline__ u32 wr_run(const WrCtx & cx, WrLds & lds, u64 (&pl)[E], u32 hm, u32 L, u32 step) {
    // pl[r] / bit r of hm: payload and head flag of position j = lane * E + r; lds.aux[j] = the position's slot (offset from cx.slot0);
    // positions >= L are heads without content
    const u32 lane = (u32)lane_id();
    for (;; step++) {
        // ---- which positions still share a group: not (head and followed by a head)
        const u32 nxt0 = __shfl_down(hm & 1u, 1u);
        const u32 hnext = (hm >> 1) | ((lane == 63u ? 1u : nxt0) << (E - 1));
        u32 amb = ~(hm & hnext) & ((1u << E) - 1u);
#pragma unroll
        for (int r = 0; r < E; r++)
            if (lane * E + r >= L) amb &= ~(1u << r);
        const u32 ambs = (u32)__popc(amb);
        const u32 total = wave_sum(ambs);
        if (total == 0u || step >= (u32)WR_CAP) {
            // ---- out: every position of the batch (a suffix alone in its group is final)
#pragma unroll
            for (int r = 0; r < E; r++)
                if (lane * E + r < L) {
                    const u64 p = cx.slot0 + lds.aux[lane * E + r];
                    const u32 sv = pl_v(pl[r]);
                    cx.v[p] = sv | (((hm >> r) & 1u) ? V_HEAD : 0u);
                    cx.pb[p] = (u8)pl_p(pl[r]);
                    if (sv == 0u) cx.counters[2] = (u32)p;
                }
            if (total && lane == 0) atomicAdd(&cx.counters[1], total);
            return 0u;
        }
        // ---- group starts (running maximum of the head positions) and the largest group
        u32 lasth = 0;  // 1 + position of this lane's last head
#pragma unroll
        for (int r = 0; r < E; r++)
            if ((hm >> r) & 1u) lasth = lane * E + r + 1u;
        u32 run0 = wave_incl_max(lasth);
        run0 = __shfl_up(run0, 1u);
        if (lane == 0) run0 = 0;  // (position 0 is a head)
        u32 span = 0;  // largest distance of a position from its group's head
        {
            u32 rn = run0;
#pragma unroll
            for (int r = 0; r < E; r++) {
                const u32 j = lane * E + r;
                if ((hm >> r) & 1u) rn = j + 1u;
                if (j < L && j + 1u - rn > span) span = j + 1u - rn;
            }
        }
        const bool small_groups = wave_max(span) < 64u && step > 0u;  // every group has at most 64 members: the tail kernel's
        u32 run = run0;
        if (small_groups || total <= 32u * E) {
            // ---- shrink: the final ones out, the others move up into half (or less) of the lanes' capacity
            const u32 before = wave_incl_add(ambs) - ambs;
            u32 so[E];
#pragma unroll
            for (int r = 0; r < E; r++) so[r] = lane * E + r < L ? (u32)lds.aux[lane * E + r] : 0u;
            wave_sync();  // every lane has read the slots of its positions before any is overwritten
            u32 d = before;
#pragma unroll
            for (int r = 0; r < E; r++) {
                if (lane * E + r < L) {
                    if ((amb >> r) & 1u) {
                        lds.pl[d] = pl[r];
                        lds.aux[d] = (u16)so[r];
                        lds.hd[d] = (u8)((hm >> r) & 1u);
                        d++;
                    } else {
                        const u64 p = cx.slot0 + so[r];
                        const u32 sv = pl_v(pl[r]);
                        cx.v[p] = sv | V_HEAD;
                        cx.pb[p] = (u8)pl_p(pl[r]);
                        if (sv == 0u) cx.counters[2] = (u32)p;
                    }
                }
            }
            wave_sync();
            // continue with the smallest capacity that holds them
            if (small_groups || total <= 64u) return total;  // the rest is the tail kernel's: the caller appends lds.pl / aux / hd [0, total) to its list
            if constexpr (E == 8) {
                if (total > 128u) return wr_reload<4>(cx, lds, total, step);
                return wr_reload<2>(cx, lds, total, step);
            }
            if constexpr (E == 4) return wr_reload<2>(cx, lds, total, step);
        }
        // ---- next code bits of the ambiguous suffixes, four suffixes of a lane at a time (their windows in flight together: with all eight
        // the kernel needs every register a wave can have, and one wave per SIMD hides no latency at all)
        constexpr int CH = E < 4 ? E : 4;
        u64 w[E];
#pragma unroll
        for (int r0 = 0; r0 < E; r0 += CH) {
            u64 wa[CH], wb[CH];
            u32 avl[CH], dep[CH];
#pragma unroll
            for (int k = 0; k < CH; k++) dep[k] = pl_d(pl[r0 + k]);
            if (step == 0) {
                // depth = the symbols inside the 56-bit windows the group was formed on: one from the sort of all suffixes, one more per
                // big round its members went through.  Every member walks the same symbols, so all arrive at the same depth (a member
                // whose text ends on the way arrives at the end and sorts first).
                for (u32 hop = 0; hop < cx.chain; hop++) {
#pragma unroll
                    for (int k = 0; k < CH; k++) load_window(cx.t, (u64)pl_v(pl[r0 + k]) + dep[k], cx.n, wa[k], wb[k], avl[k]);
#pragma unroll
                    for (int k = 0; k < CH; k++) {
                        u64 k56;
                        u32 cnt;
                        vlc_pack<56>(cx.tab, wa[k], wb[k], avl[k], k56, cnt);
                        dep[k] += ((amb >> (r0 + k)) & 1u) ? cnt : 0u;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < CH; k++) load_window(cx.t, (u64)pl_v(pl[r0 + k]) + dep[k], cx.n, wa[k], wb[k], avl[k]);
#pragma unroll
            for (int k = 0; k < CH; k++) {
                const int r = r0 + k;
                const u32 j = lane * E + r;
                if ((hm >> r) & 1u) run = j + 1u;
                const u32 gs = run - 1u;
                u64 key;
                u32 cnt;
                vlc_pack<WW_KEY, 12>(cx.tab, wa[k], wb[k], avl[k], key, cnt);
                const u32 sv = pl_v(pl[r]);
                const bool live = (u64)sv + dep[k] < cx.n;
                if (!live) {
                    key = (u64)(cx.n - sv);  // ended: shorter first
                    cnt = 0;
                }
                const bool am = (amb >> r) & 1u;
                w[r] = am ? ww_make(gs, live, key, j) : ww_make(j < L ? j : (u32)(64 * E - 1), false, 0ull, j);  // a final suffix stays where it is
                if (am) pl[r] = pl_make(sv, dep[k] + cnt, pl_p(pl[r]));
            }
        }
        // ---- sort; the payloads follow
#pragma unroll
        for (int r = 0; r < E; r++) lds.pl[lane * E + r] = pl[r];
        wave_bitonic<E>(w);
        wave_sync();
#pragma unroll
        for (int r = 0; r < E; r++) pl[r] = lds.pl[(u32)w[r] & ((1u << WW_IDX) - 1u)];
        wave_sync();
        // ---- new heads: where the word (without the position) differs from the left neighbour's
        const u64 left = __shfl_up(w[E - 1], 1u);
#pragma unroll
        for (int r = 0; r < E; r++) {
            const u64 prev = r ? w[r - 1] : left;
            if ((w[r] >> WW_IDX) != (prev >> WW_IDX)) hm |= 1u << r;  // (lane 0, r = 0 compares with its own last word: it is a head already)
        }
    }
}

template <int E>
__device__ __forceinline__ u32 wr_batch(const WrCtx & cx, WrLds & lds, u32 L) {  // ONE group: slots cx.slot0 .. + L
    const u32 lane = (u32)lane_id();
    u64 pl[E];
    u32 hm = 0;
    u32 xv[E];
    u8 xp[E];
#pragma unroll
    for (int r = 0; r < E; r++) {
        const u32 j = lane * E + r;
        const u64 p = cx.slot0 + (j < L ? j : L - 1u);
        xv[r] = cx.v[p];
        xp[r] = cx.pb[p];
    }
#pragma unroll
    for (int r = 0; r < E; r++) {
        const u32 j = lane * E + r;
        const bool in = j < L;
        hm |= ((in && j > 0u) ? 0u : 1u) << r;  // the group's head; positions past it are heads without content
        pl[r] = in ? pl_make(xv[r] & V_MASK, 0u, xp[r]) : 0ull;
        lds.aux[j] = (u16)(in ? j : 0u);
    }
    wave_sync();
    return wr_run<E>(cx, lds, pl, hm, L, 0u);
}

// The router: one wave per 512 anchor slots looks at the head bits and sends every group headed there where it belongs --
//   up to 64 members   : its ambiguous suffixes to the tail list (k_bwt_tail)
//   65 .. 512 members  : a (head slot, size) descriptor to the mid list (k_bwt_wide)
//   more               : its slots to the big list (k_big_*)
// It keeps nothing but a 33-word bitmap per wave, so dozens of waves share a CU (the first version did the wide work in the same
// kernel and ran two waves per SIMD: 13 ms instead of 3).
constexpr int RT_WAVES = 16;  // waves per router workgroup: one append to each list per WORKGROUP (an append per wave made the
                              // lists' counters the bottleneck: a single word takes ~90 atomics per microsecond)
struct RtLds {
    u32 hb[36];     // head bits of the window [a, a + 1024]
    u16 lg_c[10];   // groups of more than 64 headed in the anchor slots (at most 7) and the run that enters from the left: start ...
    u16 lg_e[10];   // ... end (exclusive, clipped to the anchor slots for the big ones) ...
    u32 lg_hp[10];  // ... and for the big ones the group's head slot; 0xFFFFFFFF marks a mid-size group (a descriptor)
    u32 cnt[4];     // what this wave appends: [0] tail entries, [1] mid descriptors, [2] big slots
    u32 base[4];    // where
};
__global__ void __launch_bounds__(RT_WAVES * WAVE) k_bwt_route(u32 n, const u32 * __restrict__ v, const u8 * __restrict__ pb, const u32 * __restrict__ hbits,
                                                              const u32 * __restrict__ carry_local, const u32 * __restrict__ group_carry,
                                                              const u8 * __restrict__ dirty, u32 * __restrict__ big_slot, u32 * __restrict__ big_hp, u32 big_cap,
                                                              u32 * __restrict__ mid_slot, u16 * __restrict__ mid_size, u32 mid_cap, u32 * __restrict__ tail_v,
                                                              u32 * __restrict__ tail_slot, u16 * __restrict__ tail_d, u8 * __restrict__ tail_pb, u32 tail_cap,
                                                              u32 * __restrict__ counters) {
    __shared__ RtLds wl[RT_WAVES];
    __shared__ u32 ok[3];  // the workgroup's appends fit their lists
    const u32 lane = (u32)lane_id();
    const u32 tile = blockIdx.x * RT_WAVES + (u32)wave_id();
    const u64 a = (u64)tile * WR_A;
    RtLds & lds = wl[wave_id()];
    const bool active = a < n && !(dirty && !(dirty[tile] | dirty[tile + 1]));
    const u32 wend = !active ? 0u : ((u64)n - a < 1025ull ? (u32)((u64)n - a) : 1025u);  // window positions that exist (the first slot past the end is a head)
    if (lane < 4u) lds.cnt[lane] = 0;
    // head bits of [a, a + 1024] from the snapshot; slots past the end are heads
    if (active && lane < 33u) {
        const u64 first = a + 32ull * lane;
        const u64 limit = ((u64)n + 63u) & ~63ull;  // the snapshot is written in whole 64-slot words
        lds.hb[lane] = first < limit ? hbits[first >> 5] : 0xFFFFFFFFu;
    }
    wave_sync();
    // 64 head bits starting at window position `start` (may be negative: nothing is known before the window)
    auto bits64 = [&](int start) -> u64 {
        const int s0 = start < 0 ? 0 : start;
        const u32 wi = (u32)s0 >> 5, sh = (u32)s0 & 31u;
        const u64 lo = ((u64)lds.hb[wi + 1 < 33u ? wi + 1 : 32u] << 32) | lds.hb[wi < 33u ? wi : 32u];
        const u64 hi = lds.hb[wi + 2 < 33u ? wi + 2 : 32u];
        u64 x = sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
        if (wi + 1 >= 33u) x &= 0xFFFFFFFFull >> sh;  // (beyond the window: unknown, read as no head)
        if (start < 0) x = start > -64 ? x << (u32)(-start) : 0ull;
        return x;
    };
    // ---- groups of up to 64 suffixes go to the tail kernel (one suffix per lane, ~50 registers, dozens of waves per CU: measured four
    // times the throughput per suffix of the wide path).  Every slot decides for itself from the head bits around it: its group's head h
    // (within 63 slots before it) and end e (within 64 after it); it goes if h lies in this wave's anchor slots and 2 <= e - h <= 64.
    u32 flags = 0, tcnt = 0;  // bit k: position lane + 64 k goes
    u64 row_ballot[9];
    u32 row_base[9];
    u32 nlg = 0;  // large groups / runs noted in lds.lg_*
    if (active) {
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const u32 p = lane + 64u * k;
            const u64 w1 = bits64((int)p - 63), w2 = bits64((int)p + 1);
            bool go = false;
            if (w1 && w2 && p < wend) {
                const u32 h = p - (u32)__clzll((unsigned long long)w1);
                const u32 e = p + (u32)__ffsll((unsigned long long)w2);
                go = h < (u32)WR_A && e - h >= 2u && e - h <= 64u;
            }
            row_ballot[k] = __ballot(go);
            flags |= (go ? 1u : 0u) << k;
        }
#pragma unroll
        for (int k = 0; k < 9; k++) {
            row_base[k] = tcnt;
            tcnt += (u32)__popcll((unsigned long long)row_ballot[k]);
        }
        // ---- the slots before the first head continue a group headed before this window (tile > 0: slot 0 is always a head)
        u32 nmid = 0, nbig = 0;
        const u32 c0 = wr_next_head(lds.hb, 0u);
        if (c0 > 0u) {
            const u32 cl = carry_local[tile], cg = group_carry[tile / (u32)SP_GROUP];
            const u32 hp = (cl > cg ? cl : cg) - 1u;  // (tile > 0: slot 0 is a head, so there is one)
            const bool big = c0 == WR_FAR || (u32)a + c0 - hp > (u32)WR_G;
            u32 stop = c0 < (u32)WR_A ? c0 : (u32)WR_A;
            stop = stop < wend ? stop : wend;
            if (big && stop > 0u) {
                if (lane == 0) {
                    lds.lg_c[nlg] = 0;
                    lds.lg_e[nlg] = (u16)stop;
                    lds.lg_hp[nlg] = hp;
                }
                nlg++;
                nbig += stop;
            }
        }
        // ---- groups of more than 64 headed here: a descriptor for the wide kernel (up to 512 members), or the big list
        for (u32 k = 0; k < 8u; k++) {
            const u32 p = lane + 64u * k;
            const bool head = (lds.hb[p >> 5] >> (p & 31u)) & 1u;
            u64 large = __ballot(head && p < wend && bits64((int)p + 1) == 0ull);  // no other head within the next 64 slots
            while (large) {
                const u32 l = (u32)__ffsll((unsigned long long)large) - 1u;
                large &= large - 1ull;
                const u32 c = l + 64u * k;
                const u32 nh = wr_next_head(lds.hb, c + 1u);  // end of the group
                const bool big = nh == WR_FAR || nh - c > (u32)WR_G;  // too large for a wave
                u32 stop = big ? (nh < (u32)WR_A ? nh : (u32)WR_A) : nh;
                stop = stop < wend ? stop : wend;
                if (lane == 0) {
                    lds.lg_c[nlg] = (u16)c;
                    lds.lg_e[nlg] = (u16)stop;
                    lds.lg_hp[nlg] = big ? (u32)a + c : 0xFFFFFFFFu;
                }
                nlg++;
                if (big) nbig += stop - c;
                else nmid++;
            }
        }
        if (lane == 0) {
            lds.cnt[0] = tcnt;
            lds.cnt[1] = nmid;
            lds.cnt[2] = nbig;
        }
    }
    __syncthreads();
    // ---- one reservation per list for the whole workgroup
    if (threadIdx.x < 3u) {
        const u32 q = threadIdx.x;
        u32 tot = 0;
        for (int w = 0; w < RT_WAVES; w++) {
            wl[w].base[q] = tot;
            tot += wl[w].cnt[q];
        }
        u32 fits = 1;
        if (tot) {
            u32 * ctr = q == 0 ? &counters[4] : q == 1 ? &counters[8] : &counters[0];
            const u32 cap = q == 0 ? tail_cap : q == 1 ? mid_cap : big_cap;
            const u32 b0 = atomicAdd(ctr, tot);
            fits = b0 + tot <= cap ? 1u : 0u;
            if (!fits) {
                counters[3] = 1u;
                if (q == 0) atomicMin(&counters[5], b0);  // the tail list is valid up to the first append that did not fit
                if (q != 2) atomicAdd(&counters[1], 1u);  // (suffixes that stay where they are: the deep path takes them)
            }
            for (int w = 0; w < RT_WAVES; w++) wl[w].base[q] += b0;
        }
        ok[q] = fits;
    }
    __syncthreads();
    if (!active) return;
    if (tcnt && ok[0]) {
        const u32 base = lds.base[0];
#pragma unroll
        for (int k = 0; k < 9; k++)
            if ((flags >> k) & 1u) {
                const u32 p = lane + 64u * k;
                const u32 d = base + row_base[k] + (u32)__popcll((unsigned long long)(row_ballot[k] & (((u64)1 << lane) - 1ull)));
                const u64 slot = a + p;
                tail_v[d] = (v[slot] & V_MASK) | (((lds.hb[p >> 5] >> (p & 31u)) & 1u) ? V_HEAD : 0u);
                tail_slot[d] = (u32)slot;
                tail_d[d] = 0;  // the tail kernel walks the windows the group was formed on
                tail_pb[d] = pb[slot];
            }
    }
    u32 dm = lds.base[1], db = lds.base[2];
    for (u32 i = 0; i < nlg; i++) {
        const u32 c = lds.lg_c[i], e = lds.lg_e[i], hp = lds.lg_hp[i];
        if (hp == 0xFFFFFFFFu) {
            if (ok[1] && lane == 0) {
                mid_slot[dm] = (u32)a + c;
                mid_size[dm] = (u16)(e - c);
            }
            dm++;
        } else {
            if (ok[2])
                for (u32 q = c + lane; q < e; q += WAVE) {
                    big_slot[db + (q - c)] = (u32)a + q;
                    big_hp[db + (q - c)] = hp;
                }
            db += e - c;
        }
    }
}

// The wide kernel: one wave per group of 65 .. 512 suffixes (a descriptor of the router), 2 / 4 / 8 suffixes per lane in registers.
// A step or two later its sub-groups have at most 64 members and go to the tail list.
__global__ void __launch_bounds__(WR_WAVES * WAVE) k_bwt_wide(const u8 * __restrict__ t, u32 n, u32 * __restrict__ v, u8 * __restrict__ pb, const u32 * __restrict__ vlc,
                                                             const u32 * __restrict__ mid_slot, const u16 * __restrict__ mid_size, u32 mid_cap, u32 * __restrict__ tail_v,
                                                             u32 * __restrict__ tail_slot, u16 * __restrict__ tail_d, u8 * __restrict__ tail_pb, u32 tail_cap,
                                                             u32 * __restrict__ counters, u32 chain) {
    __shared__ u32 tab[256];
    __shared__ WrLds wl[WR_WAVES];
    __shared__ u32 pend[WR_WAVES], pbase[WR_WAVES], pfit;
    // The number of descriptors is where the router left it, on the device (round 4: the host used to read it back to size this
    // launch -- a stream synchronisation per pass); the grid is fixed and walks the list.
    const u32 nmid = counters[8] < mid_cap ? counters[8] : mid_cap;
    if (blockIdx.x * WR_WAVES >= nmid) return;
    tab[threadIdx.x] = vlc[threadIdx.x];
    __syncthreads();
    const u32 lane = (u32)lane_id();
    WrLds & lds = wl[wave_id()];
    for (u32 g0 = blockIdx.x * WR_WAVES; g0 < nmid; g0 += gridDim.x * WR_WAVES) {  // (uniform trip count per workgroup: barriers inside)
        const u32 g = g0 + (u32)wave_id();
        u32 pending = 0;
        u64 slot0 = 0;
        if (g < nmid) {
            const u32 L = mid_size[g];
            slot0 = mid_slot[g];
            WrCtx cx{t, n, v, pb, tab, counters, chain, slot0, tail_v, tail_slot, tail_d, tail_pb, tail_cap};
            if (L <= 128u) pending = wr_batch<2>(cx, lds, L);
            else pending = wr_batch<4>(cx, lds, L);
        }
        // ---- one append to the tail list per workgroup (an append per wave made the list's counter the bottleneck)
        if (lane == 0) pend[wave_id()] = pending;
        __syncthreads();
        if (threadIdx.x == 0) {
            u32 tot = 0;
            for (int w = 0; w < WR_WAVES; w++) {
                pbase[w] = tot;
                tot += pend[w];
            }
            u32 fits = 1;
            if (tot) {
                const u32 b0 = atomicAdd(&counters[4], tot);
                fits = b0 + tot <= tail_cap ? 1u : 0u;
                if (!fits) {
                    counters[3] = 1u;
                    atomicMin(&counters[5], b0);  // the list is valid up to the first append that did not fit
                    atomicAdd(&counters[1], tot);
                }
                for (int w = 0; w < WR_WAVES; w++) pbase[w] += b0;
            }
            pfit = fits;
        }
        __syncthreads();
        const u32 base = pbase[wave_id()];
        for (u32 q = lane; q < pending; q += WAVE) {
            const u64 x = lds.pl[q];
            if (pfit) {
                tail_v[base + q] = pl_v(x) | (lds.hd[q] ? V_HEAD : 0u);
                tail_slot[base + q] = (u32)(slot0 + lds.aux[q]);
                tail_d[base + q] = (u16)pl_d(x);
                tail_pb[base + q] = (u8)pl_p(x);
            } else {  // back where they are: the deep path takes them
                const u64 p = slot0 + lds.aux[q];
                v[p] = pl_v(x) | (lds.hd[q] ? V_HEAD : 0u);
                pb[p] = (u8)pl_p(x);
            }
        }
        __syncthreads();  // pend / pbase / the waves' LDS are reused by the next descriptors
    }
}

// ---- the tail: tiny groups that need many more windows ---------------------------------------------------------------------------
// What the resolve kernel hands over: batches that are down to <= 64 ambiguous suffixes (and batches that never had more).  They sit
// in one global list, every batch a run of whole groups that starts with a head.  One wave per 64 list entries takes the groups
// whose head lies in its entries (a group has at most 64 members: they end within the next 64 entries), a lane per suffix, as many
// whole groups at a time as fit the 64 lanes.  A step costs one gather from the text and a handful of cross-lane operations; nothing
// in it waits for another wave, and with ~50 registers and 2 KB of LDS dozens of these waves share a CU, so the gather latency of
// one is covered by the others -- which the resolve kernel, with eight suffixes per lane in registers, cannot offer.
__global__ void __launch_bounds__(WAVE) k_bwt_tail(const u8 * __restrict__ t, u32 n, u32 * __restrict__ v, u8 * __restrict__ pb, const u32 * __restrict__ vlc,
                                                  const u32 * __restrict__ tail_v, const u32 * __restrict__ tail_slot, const u16 * __restrict__ tail_d,
                                                  const u8 * __restrict__ tail_pb, u32 * __restrict__ counters, u32 chain) {
    __shared__ u32 tab[256];
    __shared__ u64 s_word[WAVE];
    __shared__ u32 s_v[WAVE];
    __shared__ u16 s_d[WAVE];
    __shared__ u8 s_pb[WAVE];
    const u32 lane = (u32)lane_id();
    // The length of the list is where the router and the wide kernel left it, on the device ([5]: where the first append that did
    // not fit would have started); the grid is fixed and every wave walks the list in strides (round 4: no read-back to size the launch).
    const u32 total_entries = counters[4] < counters[5] ? counters[4] : counters[5];
    if (blockIdx.x * WAVE >= total_entries) return;
#pragma unroll
    for (int k = 0; k < 4; k++) tab[lane + 64u * k] = vlc[lane + 64u * k];
    __syncthreads();
  for (u32 off = blockIdx.x * WAVE; off < total_entries; off += gridDim.x * WAVE) {  // this wave's 64 entries; their groups end before off + 128
    u32 cursor, cnt;
    {
        const u32 i0 = off + lane, i1 = off + WAVE + lane;
        const u64 h0 = __ballot(i0 >= total_entries || (tail_v[i0 < total_entries ? i0 : 0u] >> 31) != 0u);
        const u64 h1 = __ballot(i1 >= total_entries || (tail_v[i1 < total_entries ? i1 : 0u] >> 31) != 0u);
        if (!h0) continue;  // these 64 entries continue a group headed in the previous wave's entries
        cursor = (u32)__ffsll((unsigned long long)h0) - 1u;
        cnt = h1 ? (u32)WAVE + (u32)__ffsll((unsigned long long)h1) - 1u : 2u * WAVE;
        if (off + cnt > total_entries) cnt = total_entries - off;
    }
    while (cursor < cnt) {
        const u32 i = cursor + lane;
        const bool have = i < cnt;
        const u32 x = have ? tail_v[off + i] : V_HEAD;
        // the batch: whole groups only.  If the entry behind the 64 loaded ones continues a group, that group waits for the next batch.
        const u64 hm = __ballot((x >> 31) != 0u);
        u32 e;
        if (cursor + WAVE >= cnt) e = cnt - cursor;
        else if (tail_v[off + cursor + WAVE] >> 31) e = WAVE;
        else e = 63u - (u32)__clzll((unsigned long long)hm);  // >= 1: no group here is larger than 64
        const bool act = lane < e;
        u32 sv = x & V_MASK;
        u32 sd = have ? (u32)tail_d[off + i] : 0u;
        const u32 slot = have ? tail_slot[off + i] : 0u;
        u32 sp = have ? (u32)tail_pb[off + i] : 0u;
        bool headf = !act || (x >> 31);
        bool resolved = false;
        const bool fresh = act && sd == 0u;  // came straight from the slot array: depth = the windows its group was formed on
        if (__ballot(fresh)) {
            for (u32 hop = 0; hop < chain; hop++) {
                u64 wa, wb, k56;
                u32 avl, c56;
                load_window(t, (u64)sv + sd, n, wa, wb, avl);
                vlc_pack<56>(tab, wa, wb, avl, k56, c56);
                sd += fresh ? c56 : 0u;
            }
        }
        for (u32 step = 0; step < (u32)TL_CAP; step++) {
            const u64 H = __ballot(headf);  // bit 0 is set: the batch starts with a head
            const u32 gs = 63u - (u32)__clzll((unsigned long long)(H & ((2ull << lane) - 1ull)));
            const u64 above = lane == 63u ? 0ull : (H >> (lane + 1u));
            const u32 ge = above ? lane + (u32)__ffsll((unsigned long long)above) : (u32)WAVE;
            const u32 sz = act ? (ge < e ? ge : e) - gs : 1u;
            const u32 maxsz = wave_max(sz);
            if (maxsz <= 1u) {
                resolved = true;
                break;
            }
            u64 wa, wb, key;
            u32 avl, c;
            load_window(t, (u64)sv + sd, n, wa, wb, avl);
            vlc_pack<40>(tab, wa, wb, avl, key, c);
            const bool live = (u64)sv + sd < n;
            if (!live) {
                key = (u64)(n - sv);
                c = 0;
            }
            const u64 w = ((live ? 1ull : 0ull) << 40) | key;
            u64 w2, wl;
            if (maxsz <= 12u) {
                // small groups: rank by counting (a cross-lane read per member of the largest group)
                u32 below = 0;
                for (u32 k = 0; k < maxsz; k++) {
                    const u32 q = gs + k;
                    const u64 wq = __shfl(w, (int)(q & 63u));
                    if (k < sz) below += (wq < w || (wq == w && q < lane)) ? 1u : 0u;
                }
                const u32 np = act ? gs + below : lane;
                s_word[np] = w;
                s_v[np] = sv;
                s_d[np] = (u16)(sd + c);
                s_pb[np] = (u8)sp;
                __syncthreads();
                w2 = s_word[lane];
                wl = s_word[lane ? lane - 1u : 0u];
                sv = s_v[lane];
                sd = s_d[lane];
                sp = s_pb[lane];
                __syncthreads();
            } else {
                // larger groups: one 64-lane bitonic network over [group start : 6][word : 41][lane : 6] sorts all groups at once
                // (21 stages whatever the group sizes; counting costs a cross-lane read per member: 64 of them for a group of 64)
                u64 x[1] = {((u64)(act ? gs : lane) << 47) | (w << 6) | (u64)lane};
                s_v[lane] = sv;
                s_d[lane] = (u16)(sd + c);
                s_pb[lane] = (u8)sp;
                wave_bitonic<1>(x);
                __syncthreads();
                const u32 src = (u32)x[0] & 63u;
                sv = s_v[src];
                sd = s_d[src];
                sp = s_pb[src];
                __syncthreads();
                w2 = (x[0] >> 6) & ((1ull << 41) - 1ull);
                const u64 xl = __shfl_up(x[0], 1u);
                wl = (xl >> 6) & ((1ull << 41) - 1ull);
            }
            headf = headf || w2 != wl;  // (a lane alone in its group only ever compares its own word: it stays a head)
        }
        if (act) {
            v[slot] = sv | (headf ? V_HEAD : 0u);
            pb[slot] = (u8)sp;
            if (sv == 0) counters[2] = slot;
        }
        if (!resolved) {
            const u64 H = __ballot(headf);
            const u64 nxt = (H >> 1) | (1ull << 63);
            const u64 amb = ~(H & nxt) & (e == 64u ? ~0ull : ((1ull << e) - 1ull));  // lanes not alone in their group
            if (lane == 0 && amb) atomicAdd(&counters[1], (u32)__popcll((unsigned long long)amb));
        }
        cursor += e;
    }
  }
}

// ---- the big path: one more window for the members of groups too large for the resolve kernel --------------------------------
__global__ void __launch_bounds__(BW_BLOCK) k_big_keys(const u8 * __restrict__ t, u32 n, const u32 * __restrict__ v, const u32 * __restrict__ vlc,
                                                      const u32 * __restrict__ big_slot, u32 nb, u32 chain, u64 * __restrict__ keys) {
    __shared__ u32 tab[256];
    tab[threadIdx.x] = vlc[threadIdx.x];
    __syncthreads();
    const u32 i = blockIdx.x * BW_BLOCK + threadIdx.x;
    if (i >= nb) return;
    const u32 s = v[big_slot[i]] & V_MASK;
    u64 pos = s;
    u64 wa, wb, key;
    u32 avl, cnt;
    for (u32 hop = 0; hop < chain; hop++) {  // past the windows the group was formed on (cf. k_bwt_tail's `fresh` entries)
        load_window(t, pos, n, wa, wb, avl);
        vlc_pack<56>(tab, wa, wb, avl, key, cnt);
        pos += cnt;
    }
    load_window(t, pos, n, wa, wb, avl);
    vlc_pack<56>(tab, wa, wb, avl, key, cnt);
    const bool live = pos < n;
    keys[i] = live ? ((1ull << 56) | key) : (u64)(n - s);
}

// after the two sorts: order[j] = index into the big list of the element that comes j-th; collect what moves
__global__ void __launch_bounds__(BW_BLOCK) k_big_gather(const u32 * __restrict__ order, const u32 * __restrict__ big_slot, const u32 * __restrict__ big_hp,
                                                        const u64 * __restrict__ keys, const u32 * __restrict__ v, const u8 * __restrict__ pb, u32 nb,
                                                        u32 * __restrict__ gv, u8 * __restrict__ gpb, u64 * __restrict__ gkey, u32 * __restrict__ ghp) {
    const u32 j = blockIdx.x * BW_BLOCK + threadIdx.x;
    if (j >= nb) return;
    const u32 src = order[j];
    const u32 slot = big_slot[src];
    gv[j] = v[slot] & V_MASK;
    gpb[j] = pb[slot];
    gkey[j] = keys[src];
    ghp[j] = big_hp[src];
}
__global__ void __launch_bounds__(BW_BLOCK) k_big_order_keys(const u32 * __restrict__ order, const u32 * __restrict__ big_hp, u32 nb, u32 * __restrict__ out) {
    const u32 j = blockIdx.x * BW_BLOCK + threadIdx.x;
    if (j < nb) out[j] = big_hp[order[j]];
}
// the j-th element of the sorted list goes to the j-th smallest slot of the list
__global__ void __launch_bounds__(BW_BLOCK) k_big_apply(const u32 * __restrict__ sorted_slots, const u32 * __restrict__ gv, const u8 * __restrict__ gpb,
                                                       const u64 * __restrict__ gkey, const u32 * __restrict__ ghp, u32 nb, u32 * __restrict__ v,
                                                       u8 * __restrict__ pb, u8 * __restrict__ dirty, u32 * __restrict__ counters) {
    const u32 j = blockIdx.x * BW_BLOCK + threadIdx.x;
    if (j >= nb) return;
    const bool head = j == 0 || ghp[j] != ghp[j - 1] || gkey[j] != gkey[j - 1];
    const u32 slot = sorted_slots[j];
    const u32 s = gv[j];
    v[slot] = s | (head ? V_HEAD : 0u);
    pb[slot] = gpb[j];
    dirty[slot / (u32)TR_S] = 1;
    if (s == 0) counters[2] = slot;
}

// U from the payload bytes: U[0] = T[n-1], slot i0 (suffix 0) is skipped.
__global__ void __launch_bounds__(BW_BLOCK) k_bwt_finish(const u8 * __restrict__ t, const u8 * __restrict__ pb, u32 n, const u32 * __restrict__ slot_of_suffix0,
                                                        u8 * __restrict__ out, u32 * __restrict__ idx_out) {
    const u32 i = blockIdx.x * BW_BLOCK + threadIdx.x;
    if (i >= n) return;
    const u32 i0 = *slot_of_suffix0;
    if (i == 0) {
        out[0] = t[n - 1];
        *idx_out = i0 + 1;
    }
    if (i != i0) out[i < i0 ? i + 1 : i] = pb[i];
}

// ---- the deep path: prefix doubling on ranks -----------------------------------------------------------------------------------
// Regrouping of a sorted list in two passes over tiles of 2048 elements (8 consecutive elements per thread) and a one-workgroup spine:
//   reduce : per tile, the position of its last group head and the number of elements that stay active
//   spine  : exclusive running maximum / sum over the tiles
//   apply  : in-tile running maximum of the head positions (+ the tile's carry) gives every element its group head, hence its rank;
//            in-tile sum of the keep flags (+ the tile's offset) gives the slot in the compacted list; writes SA / ISA and the
//            compacted (suffix, slot, rank) triples -- the rank rides along so that the next keys need one gather instead of two.
// The list is described either by its sorted 64-bit keys (a doubling round) or by the head flags in V (entry from the resolve passes).
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
// Head / unique flags of the BG_ITEMS elements at base (bit j), from the keys or (VF) from the head flags in `vals`; elements past m: 0.
template <bool VF>
__device__ __forceinline__ void bg_flags(const u64 * __restrict__ keys, const u32 * __restrict__ vals, u32 m, u64 base, u32 & heads, u32 & uniqs) {
    heads = uniqs = 0;
    if (VF) {
        u32 v[BG_ITEMS];
        const u64 lb = base < m ? base : (u64)m - 1;
        bg_load_u32(vals, m, lb, v);
        const u32 nxt = vals[base + BG_ITEMS < (u64)m ? base + BG_ITEMS : (u64)m - 1];
#pragma unroll
        for (int j = 0; j < BG_ITEMS; j++) {
            const u64 k = base + j;
            if (k < m) {
                const bool head = v[j] >> 31;
                const bool nh = k + 1 == (u64)m || ((j + 1 < BG_ITEMS ? v[j + 1 < BG_ITEMS ? j + 1 : j] : nxt) >> 31);
                heads |= (head ? 1u : 0u) << j;
                uniqs |= ((head && nh) ? 1u : 0u) << j;
            }
        }
    } else {
        u64 kv[BG_ITEMS + 2];
        bg_load_keys(keys, m, base < m ? base : (u64)m - 1, kv);
#pragma unroll
        for (int j = 0; j < BG_ITEMS; j++) {
            const u64 k = base + j;
            if (k < m) {
                const bool head = k == 0 || kv[j + 1] != kv[j];
                const bool uniq = head && (k + 1 == (u64)m || kv[j + 2] != kv[j + 1]);
                heads |= (head ? 1u : 0u) << j;
                uniqs |= (uniq ? 1u : 0u) << j;
            }
        }
    }
}

template <bool VF>
__global__ void __launch_bounds__(BW_BLOCK) k_bg_reduce(const u64 * __restrict__ keys, const u32 * __restrict__ vals, u32 m, u32 * __restrict__ tile_head,
                                                       u32 * __restrict__ tile_keep) {
    __shared__ u32 lds[BW_BLOCK / WAVE + 1];
    const u64 base = (u64)blockIdx.x * BG_TILE + (u64)threadIdx.x * BG_ITEMS;
    u32 heads, uniqs;
    bg_flags<VF>(keys, vals, m, base, heads, uniqs);
    u32 hp = 0, keep = 0;  // hp = 1 + position of the last head seen (0 = none)
#pragma unroll
    for (int j = 0; j < BG_ITEMS; j++) {
        const u64 k = base + j;
        if (k < m) {
            if ((heads >> j) & 1u) hp = (u32)k + 1u;
            keep += ((uniqs >> j) & 1u) ? 0u : 1u;
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
    u32 hp = 0, sum = 0;
    for (u32 t = t0; t < t1; t += 8) {
        u32 h[8], c[8];
#pragma unroll
        for (u32 k = 0; k < 8; k++) {
            const u32 i = t + k < t1 ? t + k : t1 - 1u;  // never another thread's entries
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
            const u32 i = t + k < t1 ? t + k : t1 - 1u;
            h[k] = tile_head[i];
            c[k] = tile_keep[i];
        }
#pragma unroll
        for (u32 k = 0; k < 8; k++)
            if (t + k < t1) {
                tile_head[t + k] = run_hp;
                tile_keep[t + k] = run_sum;
                run_hp = h[k] > run_hp ? h[k] : run_hp;
                run_sum += c[k];
            }
    }
    if (threadIdx.x == 0) *total = all;
}

// DOUBLING: the upper key word is the suffix's current rank (k_bwt_doubling_keys_grp), and the first sub-group of a group keeps it: no store.
template <bool VF, bool DOUBLING>
__global__ void __launch_bounds__(BW_BLOCK) k_bg_apply(const u64 * __restrict__ keys, const u32 * __restrict__ vals, const u32 * __restrict__ slots, u32 m,
                                                      const u32 * __restrict__ tile_head, const u32 * __restrict__ tile_keep, u32 * __restrict__ sa,
                                                      u32 * __restrict__ isa, u32 * __restrict__ vals_out, u32 * __restrict__ slots_out, u32 * __restrict__ grp_out,
                                                      const u8 * __restrict__ t, u8 * __restrict__ pb, u32 * __restrict__ rank_out) {
    __shared__ u32 lds[BW_BLOCK / WAVE + 1];
    __shared__ u32 st_v[BG_TILE], st_s[BG_TILE], st_g[BG_TILE];  // the tile's part of the compacted list, staged so that it leaves coalesced
    const u64 base = (u64)blockIdx.x * BG_TILE + (u64)threadIdx.x * BG_ITEMS;
    const u64 lbase = base < m ? base : (u64)m - 1;  // threads past the end load something valid and use none of it
    u32 heads, uniqs;
    bg_flags<VF>(keys, vals, m, base, heads, uniqs);
    u32 v[BG_ITEMS], sl[BG_ITEMS], hi[BG_ITEMS];
    bg_load_u32(vals, m, lbase, v);
#pragma unroll
    for (int j = 0; j < BG_ITEMS; j++) {
        v[j] &= V_MASK;
        hi[j] = 0;
    }
    if (DOUBLING) {
        u64 kv[BG_ITEMS + 2];
        bg_load_keys(keys, m, lbase, kv);
#pragma unroll
        for (int j = 0; j < BG_ITEMS; j++) hi[j] = (u32)(kv[j + 1] >> 32);
    }
    if (slots) {
        bg_load_u32(slots, m, lbase, sl);
    } else {
#pragma unroll
        for (int j = 0; j < BG_ITEMS; j++) sl[j] = (u32)(base + j);
    }
    const u32 carry_hp = tile_head[blockIdx.x], tile_off = tile_keep[blockIdx.x];
    u32 hp = 0, keep = 0;
#pragma unroll
    for (int j = 0; j < BG_ITEMS; j++) {
        const u64 k = base + j;
        if (k < m) {
            if ((heads >> j) & 1u) hp = (u32)k + 1u;
            keep += ((uniqs >> j) & 1u) ? 0u : 1u;
        }
    }
    u32 run_hp = bg_block_excl_max<BW_BLOCK>(hp, lds);
    run_hp = carry_hp > run_hp ? carry_hp : run_hp;
    u32 all;  // elements of this tile that stay active
    u32 out = block_excl_add<BW_BLOCK>(keep, lds, all);  // this thread's first slot in the tile's part of the compacted list
    // rank of an element = slot of its group's head.  The head of the group that reaches into this thread's elements from the left
    // costs one gather; from the first head on, the slots are in registers (sl[j] = k itself when the list is the whole array).
    u32 run_rank = 0;  // run_hp == 0 only where element 0 of the list starts the thread, and that one is a head
    if (run_hp > 0 && base < m) run_rank = slots ? slots[run_hp - 1u] : run_hp - 1u;
    u32 rank[BG_ITEMS];
#pragma unroll
    for (int j = 0; j < BG_ITEMS; j++) {
        const u64 k = base + j;
        rank[j] = 0;
        if (k < m) {
            if ((heads >> j) & 1u) run_rank = sl[j];
            rank[j] = run_rank;
        }
    }
#pragma unroll
    for (int j = 0; j < BG_ITEMS; j++) {
        const u64 k = base + j;
        if (k < m) {
            if (VF) rank_out[k] = rank[j];  // in slot order; the inverse suffix array is filled from these by isa_from_ranks (bucketed, not n random stores)
            else if (!DOUBLING || rank[j] != hi[j]) isa[v[j]] = rank[j];
            if ((uniqs >> j) & 1u) {
                if (!VF) {  // (VF: the list IS the array, the suffix is in its slot already -- and neighbours still read its flag)
                    sa[sl[j]] = v[j];
                    pb[sl[j]] = v[j] ? t[v[j] - 1u] : (u8)0;  // the slot's BWT symbol: the output is assembled from these bytes
                }
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

// key(k) = (current rank of suffix s, carried along by k_bg_apply) << 32 | rank of suffix s + h;   past-the-end suffixes get
// n-1-s (< h), so that a suffix that is a proper prefix of another sorts first and two such suffixes order by length.
__global__ void __launch_bounds__(BW_BLOCK) k_bwt_doubling_keys_grp(const u32 * __restrict__ vals, const u32 * __restrict__ grp, const u32 * __restrict__ isa, u32 m,
                                                                   u32 n, u32 h, u64 * __restrict__ keys) {
    const u32 k = blockIdx.x * BW_BLOCK + threadIdx.x;
    if (k >= m) return;
    const u32 s = vals[k];
    const u64 j = (u64)s + h;
    const u32 lo = (j < n) ? isa[j] + h : (n - 1u - s);
    keys[k] = ((u64)grp[k] << 32) | lo;
}

// isa[suffix] = rank for pairs that a radix pass has bucketed by the top 8 bits of the suffix number: a bucket's destinations lie in
// a window of n / 256 words (4 MB at 256 MiB), and with a contiguous range of tiles per XCD (cf. sort.hip) the XCD's L2 absorbs the
// random stores of the bucket it is working on -- measured against n stores all over the 1 GB array: see DESIGN.md.
constexpr int IS_ITEMS = 8;
__global__ void __launch_bounds__(BW_BLOCK) k_isa_scatter(const u32 * __restrict__ suf, const u32 * __restrict__ rank, u32 n, u32 tiles, u32 * __restrict__ isa) {
    const u32 per = (tiles + 7u) / 8u;
    const u32 tile = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (tile >= tiles) return;
    const u64 base = (u64)tile * (BW_BLOCK * IS_ITEMS) + threadIdx.x;
    u32 x[IS_ITEMS], r[IS_ITEMS];
#pragma unroll
    for (int k = 0; k < IS_ITEMS; k++) {
        const u64 i = base + (u64)k * BW_BLOCK;
        x[k] = suf[i < n ? i : (u64)n - 1];
        r[k] = rank[i < n ? i : (u64)n - 1];
    }
#pragma unroll
    for (int k = 0; k < IS_ITEMS; k++)
        if (base + (u64)k * BW_BLOCK < n) isa[x[k] & V_MASK] = r[k];
}

static int bits_for(u64 x) {
    int b = 0;
    while (x) { b++; x >>= 1; }
    return b ? b : 1;
}

size_t bwt_workspace_bytes(u64 n) {
    // 2 keys (16) + 2 suffix arrays (8) + payload (1) + tail list (11) + ISA (4) + 2 slot lists (8) + ranks (4) + tile words (4) + a third suffix list (4)
    return n * (16 + 8 + 1 + 11 + 4 + 8 + 4 + 4 + 4) + n / 8 + radix_temp_bytes(n) + scan_temp_words(n) * 4 + (1u << 20) + 16384 + VLC_WORK_BYTES;
}

// full LSD sort of (keys, iota) over key bits [bit_lo, bit_hi): the first pass generates the values.  Returns the buffer index of the result.
template <typename K>
static int sort_iota(const K * kin, K * k0, K * k1, u32 * v0, u32 * v1, u64 n, int bit_lo, int bit_hi, Arena & tmp, hipStream_t s, BwtStats & st) {
    radix_pass<K>(kin, k1, (const u32 *)nullptr, v1, n, bit_lo, 0xFFFFFFFFu, 0u, tmp, s);
    int cur = 1;
    K * kk[2] = {k0, k1};
    u32 * vv[2] = {v0, v1};
    st.radix_passes++;
    for (int shift = bit_lo + 8; shift < bit_hi; shift += 8) {
        radix_pass<K>(kk[cur], kk[cur ^ 1], (const u32 *)vv[cur], vv[cur ^ 1], n, shift, 0xFFFFFFFFu, 0u, tmp, s);
        cur ^= 1;
        st.radix_passes++;
    }
    return cur;
}

// Test / diagnosis switches, read from the environment ONCE (rounds 1-3 called getenv per block and per pass):
//   BZ3_BWT_TRACE=1        one line per pass on stderr
//   BZ3_BWT_BIG_ROUNDS=<k> tests only: how many more windows the big groups get before the deep path takes them (0: none; default 1);
//                          within a process: bz3_hip_debug_bwt_big_rounds(k), k < 0 = default
struct BwtSwitches {
    bool trace;
    int big_rounds_default;  // BZ3_BWT_BIG_ROUNDS as read at load (1 without it)
    std::atomic<int> big_rounds;
};
static BwtSwitches & bwt_switches() {
    static BwtSwitches sw{getenv("BZ3_BWT_TRACE") != nullptr, getenv("BZ3_BWT_BIG_ROUNDS") ? atoi(getenv("BZ3_BWT_BIG_ROUNDS")) : 1,
                          {getenv("BZ3_BWT_BIG_ROUNDS") ? atoi(getenv("BZ3_BWT_BIG_ROUNDS")) : 1}};
    return sw;
}
void bwt_set_big_rounds(int k) { bwt_switches().big_rounds.store(k < 0 ? bwt_switches().big_rounds_default : k); }  // tests: bz3_hip_debug_bwt_big_rounds (k < 0: the default as read at load)
// Grids of the wide / tail kernels: fixed by the block's size (their work lists' lengths stay on the device).  Tail: n / 256 waves, a
// quarter of what the longest possible list (n entries, 64 per wave) would take: at most four strides per wave.  Wide: n / 4096
// workgroups of WR_WAVES waves; the list holds at most n / 65 descriptors, so the worst case is ~16 strides per wave -- the lists of
// real blocks are far shorter (text: a few 10^5 descriptors at 256 MiB, one or two strides).  Call 3 of round 4 measured a tail grid of
// 24,576 waves, dozens of strides per wave, 13 % slower than the exactly sized launch of rounds 1-3.  BZ3_BWT_GRIDS="wide,tail" (read
// once) overrides: experiments.
struct BwtGrids {
    u32 wide, tail;
};
static BwtGrids bwt_grids(u32 n) {
    static const BwtGrids forced = [] {
        BwtGrids g{0, 0};
        if (const char * e = getenv("BZ3_BWT_GRIDS")) (void)sscanf(e, "%u,%u", &g.wide, &g.tail);
        return g;
    }();
    if (forced.wide && forced.tail) return forced;
    const u32 tail = n / (64u * 4u), wide = n / (64u * 4u * 16u);  // (lists: <= n entries, 64 per wave; <= n / 64 descriptors, 4 per workgroup)
    return BwtGrids{wide < 8u ? 8u : wide, tail < 32u ? 32u : tail};
}

__global__ void k_bwt_set_word(u32 * p, u32 v) { *p = v; }

// d_idx == nullptr: synchronous, returns the primary index.  d_idx != nullptr: the primary index is left THERE (a device word that outlives
// the arena) and the call returns 0 without waiting for the stream -- the block's header is written by a kernel that reads it (round 4).
s32 bwt_forward(const u8 * d_in, u32 n, u8 * d_out, Arena & tmp, hipStream_t s, BwtStats * stats, u32 * d_idx, u32 * d_outside) {
    if (d_outside && n < 2) HIP_CHECK(hipMemsetAsync(d_outside, 0, 4, s));
    if (n == 0) {
        if (d_idx) launch(k_bwt_set_word, dim3(1), dim3(1), 0, s, d_idx, 0u);
        return 0;
    }
    if (n == 1) {
        HIP_CHECK(hipMemcpyAsync(d_out, d_in, 1, hipMemcpyDeviceToDevice, s));
        if (d_idx) {
            launch(k_bwt_set_word, dim3(1), dim3(1), 0, s, d_idx, 1u);
            return 0;
        }
        HIP_CHECK(hipStreamSynchronize(s));
        return 1;
    }
    if (n > V_MASK) throw HipError{hipErrorUnknown, "block too large for the suffix sorter", __FILE__, __LINE__};
    const size_t mk = tmp.mark();
    u64 * key[2] = {tmp.take<u64>(n), tmp.take<u64>(n)};
    u32 * val[2] = {tmp.take<u32>(n), tmp.take<u32>(n)};
    u8 * pb = tmp.take<u8>(n);
    const u32 tiles = (u32)(((u64)n + TR_S - 1) / TR_S);
    u32 * tile_last = tmp.take<u32>(tiles + 1);
    u32 * carry = tmp.take<u32>(tiles + 1);  // carry_local
    const u32 sp_groups = (tiles + SP_GROUP - 1) / SP_GROUP;
    u32 * group_max = tmp.take<u32>(sp_groups + 1);
    u32 * group_carry = tmp.take<u32>(sp_groups + 1);
    u8 * dirty = tmp.take<u8>(tiles + 2);
    u32 * hbits = tmp.take<u32>(((size_t)n + 63) / 64 * 2 + (size_t)TR_S / 32 + 8);  // snapshot of the head flags, whole 64-slot words of every anchor tile
    u32 * d_words = tmp.take<u32>(16);   // counters [0] big elements, [1] left ambiguous by the resolve / tail kernels, [2] slot of suffix 0, [3] a list overflowed, [4] tail entries, [5] start of the first tail append that did not fit; [6] scan total, [7] primary index, [8] mid descriptors
    u32 * d_vlc = tmp.take<u32>(256);
    u32 * d_hist = tmp.take<u32>(256);
    BwtStats st;

    auto grid = [](u64 m) { return dim3((u32)((m + BW_BLOCK - 1) / BW_BLOCK)); };

    // ---- codes
    HIP_CHECK(hipMemsetAsync(d_hist, 0, 256 * sizeof(u32), s));
    HIP_CHECK(hipMemsetAsync(d_words, 0, 16 * sizeof(u32), s));
    HIP_CHECK(hipMemsetAsync(d_words + 5, 0xFF, sizeof(u32), s));
    launch(k_bwt_sym_hist, grid(((u64)n + 63) / 64), dim3(BW_BLOCK), 0, s, d_in, n, d_hist);
    if (d_outside) launch(k_bwt_outside, dim3(1), dim3(256), 0, s, (const u32 *)d_hist, (u32)BWT_ROUTE_KEEP, d_outside);
    {
        const size_t vm = tmp.mark();
        vlc_build_device(d_hist, d_vlc, tmp, s);  // (stream order protects the scratch: what reuses it is launched after the last level)
        tmp.release(vm);
    }

    // ---- round 0: all suffixes by their first 56 code bits
    launch(k_bwt_vlc_keys, grid(((u64)n + 7) / 8), dim3(BW_BLOCK), 0, s, d_in, n, (const u32 *)d_vlc, key[0]);
    const int cur = sort_iota<u64>(key[0], key[0], key[1], val[0], val[1], n, 8, 64, tmp, s, st);
    st.sorted_elements += n;
    st.rounds++;
    u32 * V = val[cur];
    launch(k_bwt_heads, dim3(tiles), dim3(BW_BLOCK), 0, s, (const u64 *)key[cur], V, pb, n, tile_last, hbits, d_words);
    launch(k_bwt_spine_a, dim3(sp_groups), dim3(SP_BLOCK), 0, s, (const u32 *)tile_last, tiles, carry, group_max);
    launch(k_bwt_spine_b, dim3(1), dim3(1024), 0, s, (const u32 *)group_max, sp_groups, group_carry);

    // the key buffers are free from here on: the big path's lists live there
    const u32 big_cap = n / 4;
    char * scratch = reinterpret_cast<char *>(key[0]);  // 16 n bytes (key[0] and key[1] are adjacent 256-byte-rounded regions: use key[0]'s 8 n and key[1]'s 8 n separately)
    (void)scratch;
    u32 * big_slot = reinterpret_cast<u32 *>(key[0]);                 // n/4 words
    u32 * big_hp = big_slot + big_cap;                                 // n/4 words
    u32 * bord[2] = {big_hp + big_cap, big_hp + 2 * (size_t)big_cap};  // order / sorted slots, 2 x n/4 words   (key[0]: 4 x n/4 words = 4 n bytes of 8 n)
    u32 * bgk[2] = {bord[1] + big_cap, bord[1] + 2 * (size_t)big_cap}; // group keys for the second sort        (6 n bytes)
    u32 * gv = bgk[1] + big_cap;                                       // 7 n bytes
    u32 * ghp = gv + big_cap;                                          // 8 n bytes: end of key[0]
    u64 * bkey0 = key[1];                                              // n/4 keys = 2 n bytes
    u64 * bkey[2] = {bkey0 + big_cap, bkey0 + 2 * (size_t)big_cap};    // 4 n, 6 n
    u64 * gkey = bkey0 + 3 * (size_t)big_cap;                          // 8 n: end of key[1]
    u8 * gpb = reinterpret_cast<u8 *>(val[cur ^ 1]);                   // the other suffix buffer is free as well
    // the tail kernel's input: every suffix that shares a group of up to 64 may be in it
    const u32 tail_cap = n;
    u32 * tail_v = tmp.take<u32>(tail_cap);
    u32 * tail_slot = tmp.take<u32>(tail_cap);
    u16 * tail_d = tmp.take<u16>(tail_cap);
    u8 * tail_pb = tmp.take<u8>(tail_cap);
    const u32 mid_cap = n / 64 + 16;  // groups of more than 64
    u32 * mid_slot = tmp.take<u32>(mid_cap);
    u16 * mid_size = tmp.take<u16>(mid_cap);
    u32 g = 7;            // symbols every group is known to share at least (7 per 56-bit window)
    bool deep = false;    // fall back to rank doubling
    u32 h_words[16];
    const BwtGrids grids = bwt_grids(n);
    for (int pass = 0;; pass++) {
        launch(k_bwt_route, dim3((tiles + RT_WAVES - 1) / RT_WAVES), dim3(RT_WAVES * WAVE), 0, s, n, (const u32 *)V, (const u8 *)pb, (const u32 *)hbits,
               (const u32 *)carry, (const u32 *)group_carry, (const u8 *)(pass ? dirty : nullptr), big_slot, big_hp, big_cap, mid_slot, mid_size, mid_cap, tail_v, tail_slot,
               tail_d, tail_pb, tail_cap, d_words);
        // the wide and the tail kernel take their work lists' lengths from the counters on the device: fixed grids, ONE read-back per pass
        launch(k_bwt_wide, dim3(grids.wide), dim3(WR_WAVES * WAVE), 0, s, d_in, n, V, pb, (const u32 *)d_vlc, (const u32 *)mid_slot, (const u16 *)mid_size, mid_cap, tail_v,
               tail_slot, tail_d, tail_pb, tail_cap, d_words, (u32)pass + 1u);
        launch(k_bwt_tail, dim3(grids.tail), dim3(WAVE), 0, s, d_in, n, V, pb, (const u32 *)d_vlc, (const u32 *)tail_v, (const u32 *)tail_slot, (const u16 *)tail_d,
               (const u8 *)tail_pb, d_words, (u32)pass + 1u);
        HIP_CHECK(hipMemcpyAsync(h_words, d_words, sizeof h_words, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        const u32 nb = h_words[0];
        if (bwt_switches().trace) fprintf(stderr, "[bwt] n %u pass %d depth %u: %u suffixes in groups > %d, %u groups through the wide kernel, %u suffixes through the tail kernel, %u given up, overflow %u\n", n, pass, g, nb, TR_G, h_words[8], h_words[4], h_words[1], h_words[3]);
        if (nb == 0) {  // no group left that is too large: done, unless the resolve kernel gave some up (counted over all passes)
            deep = h_words[1] != 0;
            break;
        }
        const int max_big = bwt_switches().big_rounds.load();  // one more window takes text from ~12 % of its suffixes in big groups to ~1 %; what is left is deep and goes to rank doubling
        if (h_words[3] || pass >= max_big || nb > n / 4) {  // groups too many / too deep for windows of code bits
            deep = true;
            break;
        }
        // ---- one more window for the members of the big groups
        st.rounds++;
        st.sorted_elements += nb;
        launch(k_big_keys, grid(nb), dim3(BW_BLOCK), 0, s, d_in, n, (const u32 *)V, (const u32 *)d_vlc, (const u32 *)big_slot, nb, (u32)pass + 1u, bkey0);
        int c = sort_iota<u64>(bkey0, bkey[0], bkey[1], bord[0], bord[1], nb, 0, 57, tmp, s, st);
        u32 * order = bord[c];
        u32 * ofree = bord[c ^ 1];
        launch(k_big_order_keys, grid(nb), dim3(BW_BLOCK), 0, s, (const u32 *)order, (const u32 *)big_hp, nb, bgk[0]);
        {
            const int hb = bits_for(n);
            u32 * oo[2] = {order, ofree};
            const int c2 = radix_sort_pairs<u32>(bgk[0], bgk[1], oo[0], oo[1], nb, 0, hb, tmp, s);
            st.radix_passes += (hb + 7) / 8;
            order = oo[c2];
            ofree = oo[c2 ^ 1];
        }
        launch(k_big_gather, grid(nb), dim3(BW_BLOCK), 0, s, (const u32 *)order, (const u32 *)big_slot, (const u32 *)big_hp, (const u64 *)bkey0, (const u32 *)V,
               (const u8 *)pb, nb, gv, gpb, gkey, ghp);
        // the slots of the list in increasing order (the resolve kernel emitted them tile by tile in no particular order)
        {
            const int hb = bits_for(n);
            // keys = slots, values unused (the order buffers are free now)
            HIP_CHECK(hipMemcpyAsync(bgk[0], big_slot, (size_t)nb * 4, hipMemcpyDeviceToDevice, s));
            u32 * oo[2] = {order, ofree};
            const int c3 = radix_sort_pairs<u32>(bgk[0], bgk[1], oo[0], oo[1], nb, 0, hb, tmp, s);
            st.radix_passes += (hb + 7) / 8;
            HIP_CHECK(hipMemsetAsync(dirty, 0, tiles + 2, s));
            launch(k_big_apply, grid(nb), dim3(BW_BLOCK), 0, s, (const u32 *)bgk[c3], (const u32 *)gv, (const u8 *)gpb, (const u64 *)gkey, (const u32 *)ghp, nb, V, pb,
                   dirty, d_words);
        }
        g += 7;
        HIP_CHECK(hipMemsetAsync(d_words, 0, sizeof(u32), s));      // the big list and the tail list are rebuilt by the next pass
        HIP_CHECK(hipMemsetAsync(d_words + 4, 0, sizeof(u32), s));
        HIP_CHECK(hipMemsetAsync(d_words + 5, 0xFF, sizeof(u32), s));
        HIP_CHECK(hipMemsetAsync(d_words + 8, 0, sizeof(u32), s));
        launch(k_bwt_reduce_heads, dim3(tiles), dim3(BW_BLOCK), 0, s, (const u32 *)V, n, tile_last, hbits);
        launch(k_bwt_spine_a, dim3(sp_groups), dim3(SP_BLOCK), 0, s, (const u32 *)tile_last, tiles, carry, group_max);
    launch(k_bwt_spine_b, dim3(1), dim3(1024), 0, s, (const u32 *)group_max, sp_groups, group_carry);
    }

    u32 idx = 0;
    if (!deep) {
        launch(k_bwt_finish, grid(n), dim3(BW_BLOCK), 0, s, d_in, (const u8 *)pb, n, (const u32 *)(d_words + 2), d_out, d_words + 7);
    } else {
        // ---- deep path: ISA from the flags, then prefix doubling on (rank, rank of the suffix h further on)
        u32 * sa = V;  // in place: a slot holds its final suffix (without flag) once the suffix is alone in its group
        u32 * isa = tmp.take<u32>(n);
        u32 * slot[2] = {tmp.take<u32>(n), tmp.take<u32>(n)};
        u32 * grp = tmp.take<u32>(n);
        u32 * vv[2] = {val[cur ^ 1], tmp.take<u32>(n)};
        const u32 max_tiles = (u32)(((u64)n + BG_TILE - 1) / BG_TILE);
        u32 * tile_head = tmp.take<u32>(2 * (size_t)max_tiles + 16);
        u32 * tile_keep = tile_head + max_tiles + 8;
        // every group still ambiguous shares at least h symbols: 7 per window of the big rounds; a group the resolve kernel gave up
        // shares the first window's 7 and 5 more per step it took (WR_CAP steps): never the shallower of the two.
        u32 m = n, h = g < 7u + 5u * (u32)TL_CAP ? g : 7u + 5u * (u32)TL_CAP;
        if (bwt_switches().trace) fprintf(stderr, "[bwt] n %u deep path from depth %u\n", n, h);
        {
            // ranks in slot order, then the inverse suffix array from (suffix, rank) pairs bucketed by the top bits of the suffix
            u32 * rk = reinterpret_cast<u32 *>(key[0]);
            u32 * kb = rk + n;                            // key[0] holds 2 n words
            u32 * rb = reinterpret_cast<u32 *>(key[1]);
            const u32 tl = (u32)(((u64)m + BG_TILE - 1) / BG_TILE);
            launch(k_bg_reduce<true>, dim3(tl), dim3(BW_BLOCK), 0, s, (const u64 *)nullptr, (const u32 *)V, m, tile_head, tile_keep);
            launch(k_bg_spine, dim3(1), dim3(BG_SPINE), 0, s, tile_head, tile_keep, tl, d_words + 6);
            launch(k_bg_apply<true, false>, dim3(tl), dim3(BW_BLOCK), 0, s, (const u64 *)nullptr, (const u32 *)V, (const u32 *)nullptr, m, (const u32 *)tile_head,
                   (const u32 *)tile_keep, sa, isa, vv[0], slot[0], grp, d_in, pb, rk);
            const int nbits = bits_for((u64)n - 1);
            const int shift = nbits > 8 ? nbits - 8 : 0;
            radix_pass<u32>((const u32 *)V, kb, (const u32 *)rk, rb, n, shift, 0xFFFFFFFFu, 0u, tmp, s);
            st.radix_passes++;
            const u32 itiles = (u32)(((u64)n + BW_BLOCK * IS_ITEMS - 1) / (BW_BLOCK * IS_ITEMS));
            launch(k_isa_scatter, dim3(8u * ((itiles + 7u) / 8u)), dim3(BW_BLOCK), 0, s, (const u32 *)kb, (const u32 *)rb, n, itiles, isa);
        }
        int scur = 0;
        u32 * vact = vv[0], * vfree = vv[1];
        for (;;) {
            u32 m_next = 0;
            HIP_CHECK(hipMemcpyAsync(&m_next, d_words + 6, 4, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
            if (m_next == 0) break;
            m = m_next;
            st.rounds++;
            if (bwt_switches().trace) fprintf(stderr, "[bwt] n %u doubling round at depth %u: %u active suffixes\n", n, h, m);
            // the active suffixes are in vact, their slots in slot[scur], their ranks in grp
            launch(k_bwt_doubling_keys_grp, grid(m), dim3(BW_BLOCK), 0, s, (const u32 *)vact, (const u32 *)grp, (const u32 *)isa, m, n, h, key[0]);
            u32 * pv[2] = {vact, vfree};
            const int lo_bits = bits_for((u64)n + h);
            const int hi_bits = bits_for(n);
            int c = radix_sort_pairs<u64>(key[0], key[1], pv[0], pv[1], m, 0, lo_bits, tmp, s);
            c ^= radix_sort_pairs<u64>(key[c], key[c ^ 1], pv[c], pv[c ^ 1], m, 32, 32 + hi_bits, tmp, s);
            u32 * vsorted = pv[c];
            vfree = pv[c ^ 1];
            st.radix_passes += (lo_bits + 7) / 8 + (hi_bits + 7) / 8;
            st.sorted_elements += m;
            const u32 tl = (u32)(((u64)m + BG_TILE - 1) / BG_TILE);
            launch(k_bg_reduce<false>, dim3(tl), dim3(BW_BLOCK), 0, s, (const u64 *)key[c], (const u32 *)nullptr, m, tile_head, tile_keep);
            launch(k_bg_spine, dim3(1), dim3(BG_SPINE), 0, s, tile_head, tile_keep, tl, d_words + 6);
            launch(k_bg_apply<false, true>, dim3(tl), dim3(BW_BLOCK), 0, s, (const u64 *)key[c], (const u32 *)vsorted, (const u32 *)slot[scur], m,
                   (const u32 *)tile_head, (const u32 *)tile_keep, sa, isa, vfree, slot[scur ^ 1], grp, d_in, pb, (u32 *)nullptr);
            scur ^= 1;
            vact = vfree;
            vfree = vsorted;
            if (h >= 0x40000000u) throw HipError{hipErrorUnknown, "suffix sort did not converge", __FILE__, __LINE__};
            h *= 2;
        }
        // isa[0] = the slot of suffix 0
        launch(k_bwt_finish, grid(n), dim3(BW_BLOCK), 0, s, d_in, (const u8 *)pb, n, (const u32 *)isa, d_out, d_words + 7);
    }
    if (d_idx) {
        HIP_CHECK(hipMemcpyAsync(d_idx, d_words + 7, 4, hipMemcpyDeviceToDevice, s));
    } else {
        HIP_CHECK(hipMemcpyAsync(&idx, d_words + 7, 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
    tmp.release(mk);
    if (stats) *stats = st;
    return (s32)idx;
}

}  // namespace bz3
