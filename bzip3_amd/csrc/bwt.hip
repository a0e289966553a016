// bwt.hip -- forward Burrows-Wheeler transform of one block on gfx950.
// Replaces libsais_bwt (reference include/libsais.h:4095-4121, SA-IS: :3941-3983, :3740-3939), which is a sequential
// induced-sorting algorithm whose inner scans carry a dependency through 256 bucket cursors.  This is NOT a port of it.
//
// Round 3 design ("sort once, then resolve groups where they lie"):
//
//   codes   : the bytes of the block are given an order-preserving prefix-free code of at most 8 bits per symbol, built on the
//             host from the block's byte histogram (an optimal height-limited alphabetic tree: frequent bytes get short codes).
//             Comparing the concatenated code bits of two suffixes is the same as comparing their bytes, and a 56-bit window
//             holds 7 symbols at least and ~12 of English text.
//   round 0 : key(i) = first 56 code bits of suffix i (16 symbols at most, zero padded past the end) << 8 | T[i-1]; ONE stable LSD
//             radix sort of all n (key, i) pairs over the upper 56 bits (7 passes of sort.hip).  The byte that precedes the suffix
//             rides in the low byte: it is the BWT symbol of the suffix, so the output needs no gather through the suffix array.
//   groups  : suffixes with equal 56-bit windows form a group of neighbouring slots.  V[slot] = suffix | head flag (bit 31).
//   resolve : k_bwt_resolve -- one workgroup per 1536 slots sorts every group of <= 512 suffixes that starts there COMPLETELY, in
//             LDS: each step fetches the next 40 code bits of every still-ambiguous suffix straight from the text (no inverse
//             suffix array, no rank table: groups are independent of each other), ranks small groups by counting and larger ones
//             with a bitonic network, splits the groups and drops the suffixes that became unique.  Text needs 2-3 steps.
//   big     : groups of > 512 suffixes (a few % of text) get one more 56-bit window each through the global radix sorter
//             (k_big_*), after which they are small and go through k_bwt_resolve again.
//   deep    : what is still ambiguous after that (long repeats: runs, periodic data) falls back to classic prefix doubling on
//             ranks (ISA built once, k_fb_* / k_bg_* / doubling_rounds) -- the only path that needs random 4-byte scatters.
//   output  : U[0] = T[n-1]; U[i < i0 ? i+1 : i] = payload byte of slot i for i != i0 = slot of suffix 0; idx = i0+1.
//
// Order of equal windows past the end of the block: a suffix that is a proper prefix of another sorts first (zero padding is the
// smallest continuation, and a suffix with no symbol left at its depth sorts before every live one, shorter first) -- exactly the
// order libsais produces.
//
// HBM per block of n bytes (carved from the per-device workspace): 2 x key u64[n], 2 x suffix u32[n], payload u8[n]; the deep path
// additionally ISA u32[n], 2 x slot u32[n], rank u32[n], tile words.  Algorithmic traffic: 11 B per input byte.
#include <vector>

#include "prims.hpp"
#include "sort.hpp"
#include "stages.hpp"

namespace bz3 {

constexpr int BW_BLOCK = 256;
constexpr u32 V_HEAD = 0x80000000u;  // slot starts a group
constexpr u32 V_MASK = 0x3FFFFFFFu;  // suffix number (n < 2^30: bz3_bound(511 MiB) = 546.5 M)

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
// intervals: cost[l][i][j] = lightest tree of height <= l over the present values i .. j-1 (Knuth's root bounds; a cell whose
// bounded search finds nothing feasible is searched in full, so the result is always a valid code).  Host, ~1 ms.
static void vlc_build(const u32 * cnt, u32 * table) {
    int sym[256], s = 0;
    for (int c = 0; c < 256; c++) {
        table[c] = 0;
        if (cnt[c]) sym[s++] = c;
    }
    if (s == 0) return;
    if (s == 1) {
        table[sym[0]] = (0u << 4) | 1u;
        return;
    }
    constexpr u64 INF = ~0ull >> 2;
    const int S1 = s + 1;
    std::vector<u64> pre((size_t)S1, 0);
    for (int i = 0; i < s; i++) pre[(size_t)i + 1] = pre[(size_t)i] + cnt[sym[i]];
    static thread_local std::vector<u64> cost;
    static thread_local std::vector<u16> root;
    cost.assign((size_t)(VLC_MAXLEN + 1) * S1 * S1, INF);
    root.assign((size_t)(VLC_MAXLEN + 1) * S1 * S1, 0);
    auto at = [&](int l, int i, int j) -> size_t { return ((size_t)l * S1 + (size_t)i) * S1 + (size_t)j; };
    for (int l = 0; l <= VLC_MAXLEN; l++)
        for (int i = 0; i < s; i++) cost[at(l, i, i + 1)] = 0;
    for (int l = 1; l <= VLC_MAXLEN; l++) {
        for (int len = 2; len <= s && len <= (1 << l); len++) {
            for (int i = 0; i + len <= s; i++) {
                const int j = i + len;
                int lo = i + 1, hi = j - 1;
                if (len > 2) {
                    const int a = root[at(l, i, j - 1)], b = root[at(l, i + 1, j)];
                    if (a >= i + 1 && b >= a && b <= j - 1) { lo = a; hi = b; }
                }
                u64 best = INF;
                int bk = 0;
                for (int pass = 0; pass < 2 && best >= INF; pass++) {
                    if (pass == 1) { lo = i + 1; hi = j - 1; }
                    for (int k = lo; k <= hi; k++) {
                        const u64 a = cost[at(l - 1, i, k)], b = cost[at(l - 1, k, j)];
                        if (a >= INF || b >= INF) continue;
                        if (a + b < best) { best = a + b; bk = k; }
                    }
                }
                if (best < INF) {
                    cost[at(l, i, j)] = best + (pre[(size_t)j] - pre[(size_t)i]);
                    root[at(l, i, j)] = (u16)bk;
                }
            }
        }
    }
    // walk the tree (explicit stack: interval, level, code so far)
    struct Node { int i, j, l; u32 code, len; };
    std::vector<Node> st;
    st.push_back({0, s, VLC_MAXLEN, 0u, 0u});
    while (!st.empty()) {
        const Node nd = st.back();
        st.pop_back();
        if (nd.j - nd.i == 1) {
            table[sym[nd.i]] = (nd.code << 4) | nd.len;
            continue;
        }
        const int k = root[at(nd.l, nd.i, nd.j)];
        st.push_back({nd.i, k, nd.l - 1, nd.code << 1, nd.len + 1});
        st.push_back({k, nd.j, nd.l - 1, (nd.code << 1) | 1u, nd.len + 1});
    }
}

// The first B code bits of the symbols in the window (wa, wb: 16 bytes, little endian; avail <= 16 of them exist), zero padded;
// cnt = symbols that lie entirely inside those bits.
template <int B>
__device__ __forceinline__ void vlc_pack(const u32 * __restrict__ tab, u64 wa, u64 wb, u32 avail, u64 & key, u32 & cnt) {
    u64 acc = 0;
    u32 bits = 0;
    cnt = 0;
#pragma unroll
    for (u32 k = 0; k < (u32)VLC_WINDOW; k++) {
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
constexpr int TR_NT = 256;
constexpr int TR_PER = 8;
constexpr int TR_WIN = TR_NT * TR_PER;  // 2048
constexpr int TR_G = 512;
constexpr int TR_S = TR_WIN - TR_G;     // 1536
constexpr int TR_SMALL = 64;            // groups up to this size are ranked by counting, larger ones by the bitonic network
constexpr int TR_STEPS = 3;             // workgroup-wide steps; what is still ambiguous then (a few %, in tiny groups) goes to the tail kernel
constexpr int TL_MAX = 64;              // largest group the tail kernel takes (one wave sorts it)
constexpr int TL_CAP = 160;             // tail steps before a group is left to the deep path (>= 800 symbols)
constexpr u32 TR_FAR = 0xFFFFu;

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

// One workgroup: carry[t] = slot of the last head before tile t (tile 0: 0, never used: slot 0 is a head).
constexpr int SP_BLOCK = 1024;
__global__ void __launch_bounds__(SP_BLOCK) k_bwt_spine_max(const u32 * __restrict__ tile_last, u32 tiles, u32 * __restrict__ carry) {
    __shared__ u32 lds[SP_BLOCK / WAVE + 1];
    const u32 per = (tiles + SP_BLOCK - 1) / SP_BLOCK;
    const u32 t0 = threadIdx.x * per, t1 = t0 + per < tiles ? t0 + per : tiles;
    u32 hp = 0;
    for (u32 t = t0; t < t1; t++) {
        const u32 h = tile_last[t];
        hp = h > hp ? h : hp;
    }
    // exclusive running maximum over the threads
    const u32 incl = wave_incl_max(hp);
    if (lane_id() == WAVE - 1) lds[wave_id()] = incl;
    u32 up = __shfl_up(incl, 1u);
    if (lane_id() == 0) up = 0u;
    __syncthreads();
    u32 run = 0;
    for (int w = 0; w < wave_id(); w++) run = lds[w] > run ? lds[w] : run;
    run = up > run ? up : run;
    for (u32 t = t0; t < t1; t++) {
        carry[t] = run ? run - 1u : 0u;
        const u32 h = tile_last[t];
        run = h > run ? h : run;
    }
}

// ---- the resolve kernel ------------------------------------------------------------------------------------------------------
// Sort word of an ambiguous suffix: [group start : 11][live : 1][next 40 code bits, or the suffix length when it has ended][index : 12].
__device__ __forceinline__ u32 tr_pad(u32 i) { return i + (i >> 3); }  // LDS index of element i of the bitonic buffer (bank spread)

// Bitonic sort of buf[0 .. mp) (mp a power of two <= TR_WIN, padded indices), ascending; every thread of the workgroup calls it.
// Strides are taken three at a time: a thread loads the 8 elements whose indices differ in those three bits, runs the three
// compare-exchange stages in registers and stores them back -- one LDS round trip per three stages.
__device__ __forceinline__ void tr_bitonic(u64 * __restrict__ buf, u32 mp) {
    for (u32 k = 2; k <= mp; k <<= 1) {
        u32 j = k >> 1;  // largest stride of this merge
        while (j >= 1) {
            // strides j, j/2, .. down to jl: as many as three, and so that what remains below is a multiple of three stages
            u32 lj = 0;
            while ((1u << lj) < j) lj++;  // log2(j)
            const u32 take = (lj % 3u) + 1u;  // lj+1 stages remain: take ((lj+1) mod 3, or 3) first
            const u32 c = take > lj + 1u ? lj + 1u : take;
            const u32 lq = lj + 1u - c;  // lowest owned bit
            const u32 sets = mp >> c;
            for (u32 sidx = threadIdx.x; sidx < sets; sidx += TR_NT) {
                const u32 base = ((sidx >> lq) << (lq + c)) | (sidx & ((1u << lq) - 1u));
                const bool asc = (base & k) == 0u;
                u64 x[8];
#pragma unroll
                for (u32 r = 0; r < 8; r++)
                    if (r < (1u << c)) x[r] = buf[tr_pad(base + (r << lq))];
#pragma unroll
                for (u32 b = 3; b-- > 0;) {
                    if (b < c) {
#pragma unroll
                        for (u32 r = 0; r < 8; r++) {
                            if (r < (1u << c) && !(r & (1u << b))) {
                                const u32 r2 = r | (1u << b);
                                const u64 lo = x[r] < x[r2] ? x[r] : x[r2];
                                const u64 hi = x[r] < x[r2] ? x[r2] : x[r];
                                x[r] = asc ? lo : hi;
                                x[r2] = asc ? hi : lo;
                            }
                        }
                    }
                }
#pragma unroll
                for (u32 r = 0; r < 8; r++)
                    if (r < (1u << c)) buf[tr_pad(base + (r << lq))] = x[r];
            }
            __syncthreads();
            j = lq ? (1u << (lq - 1u)) : 0u;
            if (lq == 0) break;
        }
    }
}

__global__ void __launch_bounds__(TR_NT) k_bwt_resolve(const u8 * __restrict__ t, u32 n, u32 * __restrict__ v, u8 * __restrict__ pb,
                                                      const u32 * __restrict__ hbits, const u32 * __restrict__ carry, const u8 * __restrict__ dirty, const u32 * __restrict__ vlc,
                                                      u32 * __restrict__ big_slot, u32 * __restrict__ big_hp, u32 big_cap, u32 * __restrict__ chunk_off,
                                                      u32 * __restrict__ chunk_cnt, u32 * __restrict__ tail_v, u32 * __restrict__ tail_slot,
                                                      u16 * __restrict__ tail_d, u8 * __restrict__ tail_pb, u32 tail_cap, u32 * __restrict__ counters, u32 chain) {
    __shared__ u64 word[TR_WIN];                      // sort words by active index; the window copy of V / PB lives here during the prologue
    __shared__ u64 sb[TR_WIN + TR_WIN / 8 + 8];       // bitonic buffer of the larger groups
    __shared__ u32 av[TR_WIN];                        // suffix of active element e
    __shared__ u16 ad[TR_WIN + 8];                    // its depth (symbols known equal inside its group); head positions during the prologue
    __shared__ u16 aslot[TR_WIN];                     // its slot inside the window (slots never move, suffixes do)
    __shared__ u16 gstart[TR_WIN / 2 + 8];            // first active index of group number o (1-based), [groups + 1] = m
    __shared__ u16 mact[TR_WIN];                      // active index of compact "larger group" element q
    __shared__ u8 apb[TR_WIN];                        // payload byte
    __shared__ u8 ahead[TR_WIN + 8];                  // 1: active element e starts a group
    __shared__ u32 tab[256];
    __shared__ u32 red[TR_NT / WAVE + 1];
    __shared__ u32 bcast[4];

    const u32 tile = blockIdx.x;
    const u64 a = (u64)tile * TR_S;
    if (a >= n) return;
    if (dirty && !(dirty[tile] | dirty[tile + 1])) return;
    const u32 tid = threadIdx.x;
    u32 * vw = reinterpret_cast<u32 *>(word);                 // TR_WIN + 1 words
    u8 * pbw = reinterpret_cast<u8 *>(vw + TR_WIN + 8);       // TR_WIN bytes (16 KB region: 8 KB + 32 + 2 KB)
    u16 * hpos = ad;                                          // head positions by ordinal (prologue only)
    tab[tid] = vlc[tid];
    {
        u32 x[TR_PER], hb[TR_PER];
        u8 y[TR_PER];
#pragma unroll
        for (int r = 0; r < TR_PER; r++) {  // coalesced: consecutive threads, consecutive slots
            const u32 w = (u32)r * TR_NT + tid;
            const u64 p = a + w;
            const u64 pc = p < n ? p : (u64)n - 1;
            x[r] = v[pc];
            y[r] = pb[pc];
            hb[r] = hbits[pc >> 5];
        }
#pragma unroll
        for (int r = 0; r < TR_PER; r++) {
            const u32 w = (u32)r * TR_NT + tid;
            const u64 p = a + w;
            // group boundaries from the snapshot; past the end: a head, so that the last group ends there
            vw[w] = p < n ? ((x[r] & V_MASK) | (((hb[r] >> (p & 31u)) & 1u) << 31)) : V_HEAD;
            pbw[w] = y[r];
        }
        if (tid == 0) {
            const u64 p = a + TR_WIN;
            vw[TR_WIN] = p < n ? (((hbits[p >> 5] >> (p & 31u)) & 1u) << 31) : V_HEAD;
        }
    }
    __syncthreads();
    // head ordinals: thread owns window positions tid*8 .. tid*8+7
    const u32 w0 = tid * TR_PER;
    u32 hflag = 0;  // bit r: position w0 + r is a head
    u32 nheads = 0;
#pragma unroll
    for (int r = 0; r < TR_PER; r++) {
        const u32 h = vw[w0 + r] >> 31;
        hflag |= h << r;
        nheads += h;
    }
    u32 total_heads;
    u32 ord = block_excl_add<TR_NT>(nheads, red, total_heads);  // heads before w0
    {
        u32 o = ord;
#pragma unroll
        for (int r = 0; r < TR_PER; r++)
            if ((hflag >> r) & 1u) hpos[++o] = (u16)(w0 + r);
    }
    if (tid == 0) hpos[total_heads + 1] = (vw[TR_WIN] >> 31) ? (u16)TR_WIN : (u16)TR_FAR;
    __syncthreads();
    // classification
    u32 mine = 0, bigf = 0;  // bit r
    u32 bighp[TR_PER];
    const u32 carry_hp = carry[tile];
    {
        u32 o = ord;
#pragma unroll
        for (int r = 0; r < TR_PER; r++) {
            const u32 w = w0 + r;
            o += (hflag >> r) & 1u;
            const u32 hl = o ? hpos[o] : 0u;
            const u32 nh = hpos[o + 1];  // o + 1 <= total_heads + 1
            const bool known = nh != TR_FAR;
            bighp[r] = 0;
            if (o && hl < (u32)TR_S && known && nh - hl >= 2u && nh - hl <= (u32)TR_G) mine |= 1u << r;
            if (w < (u32)TR_S && a + w < n) {
                const u32 hg = o ? (u32)a + hl : carry_hp;
                const bool big = !known || ((u32)a + nh - hg) > (u32)TR_G;
                if (big) {
                    bigf |= 1u << r;
                    bighp[r] = hg;
                }
            }
        }
    }
    // the groups too large for this kernel: (slot, head slot) to the global list
    {
        u32 tot;
        const u32 nb = (u32)__popc(bigf);
        u32 off = block_excl_add<TR_NT>(nb, red, tot);
        if (tot) {
            if (tid == 0) bcast[0] = atomicAdd(&counters[0], tot);
            __syncthreads();
            const u32 gb = bcast[0];
            if (gb + tot > big_cap) {
                if (tid == 0) counters[3] = 1u;
            } else {
#pragma unroll
                for (int r = 0; r < TR_PER; r++)
                    if ((bigf >> r) & 1u) {
                        big_slot[gb + off] = (u32)a + w0 + r;
                        big_hp[gb + off] = bighp[r];
                        off++;
                    }
            }
        }
    }
    // the active list
    u32 m;
    {
        const u32 nm = (u32)__popc(mine);
        u32 e = block_excl_add<TR_NT>(nm, red, m);
        u32 sv[TR_PER];
        u8 sp[TR_PER];
#pragma unroll
        for (int r = 0; r < TR_PER; r++) {
            sv[r] = vw[w0 + r];
            sp[r] = pbw[w0 + r];
        }
        __syncthreads();  // the window copy (inside `word`) and hpos (inside `ad`) are dead from here on
#pragma unroll
        for (int r = 0; r < TR_PER; r++)
            if ((mine >> r) & 1u) {
                av[e] = sv[r] & V_MASK;
                apb[e] = sp[r];
                aslot[e] = (u16)(w0 + r);
                ahead[e] = (u8)(sv[r] >> 31);
                ad[e] = 0;
                e++;
            }
    }
    __syncthreads();

    for (u32 iter = 0; m > 0; iter++) {
        const u32 e0 = tid * TR_PER;  // this thread owns active indices e0 .. e0+7
        // ---- group structure of the active list
        u32 hf = 0, nh_ = 0;
#pragma unroll
        for (int r = 0; r < TR_PER; r++) {
            const u32 h = e0 + r < m ? (u32)ahead[e0 + r] : 0u;
            hf |= h << r;
            nh_ += h;
        }
        u32 groups;
        const u32 obase = block_excl_add<TR_NT>(nh_, red, groups);
        {
            u32 o = obase;
#pragma unroll
            for (int r = 0; r < TR_PER; r++)
                if ((hf >> r) & 1u) gstart[++o] = (u16)(e0 + r);
        }
        if (tid == 0) gstart[groups + 1] = (u16)m;
        if (iter == (u32)TR_STEPS) {
            // ---- hand-over: what is still ambiguous sits in tiny groups scattered over all tiles and may need dozens of further
            // windows; a workgroup-wide step for a handful of suffixes costs as much as one for 2048.  Groups of <= TL_MAX go to the
            // tail kernel (one wave per tile's leftovers, no workgroup barriers, many waves per CU); larger ones are written back as
            // they are and left to the deep path.
            __syncthreads();
            u32 tf = 0;
            {
                u32 o = obase;
#pragma unroll
                for (int r = 0; r < TR_PER; r++) {
                    o += (hf >> r) & 1u;
                    if (e0 + r < m && (u32)gstart[o + 1] - (u32)gstart[o] <= (u32)TL_MAX) tf |= 1u << r;
                }
            }
            u32 nt;
            u32 off = block_excl_add<TR_NT>((u32)__popc(tf), red, nt);
            if (tid == 0) bcast[1] = nt ? atomicAdd(&counters[4], nt) : 0u;
            __syncthreads();
            const u32 tb = bcast[1];
            const bool fits = tb + nt <= tail_cap;
            if (tid == 0) {
                if (!fits) counters[3] = 1u;
                chunk_off[tile] = tb;
                chunk_cnt[tile] = fits ? nt : 0u;
                if (m - (fits ? nt : 0u)) atomicAdd(&counters[1], m - (fits ? nt : 0u));
            }
#pragma unroll
            for (int r = 0; r < TR_PER; r++) {
                const u32 e = e0 + r;
                if (e < m) {
                    const u32 sv_ = av[e] | ((u32)ahead[e] << 31);
                    const u64 p = a + aslot[e];
                    if (((tf >> r) & 1u) && fits) {
                        tail_v[tb + off] = sv_;
                        tail_slot[tb + off] = (u32)p;
                        tail_d[tb + off] = ad[e];
                        tail_pb[tb + off] = apb[e];
                        off++;
                    } else {
                        v[p] = sv_;
                        pb[p] = apb[e];
                        if ((sv_ & V_MASK) == 0) counters[2] = (u32)p;
                    }
                }
            }
            break;
        }
        // ---- next code bits of every active suffix
        u64 wd[TR_PER];
        {
            u32 sv[TR_PER], sd[TR_PER];
#pragma unroll
            for (int r = 0; r < TR_PER; r++) {
                const u32 e = e0 + r < m ? e0 + r : (m - 1);
                sv[r] = av[e];
                sd[r] = ad[e];
            }
            u64 wa[TR_PER], wb[TR_PER];
            u32 avl[TR_PER];
            if (iter == 0) {
                // depth = the symbols inside the 56-bit windows the group was formed on: one from the sort of all suffixes, one more
                // per big round its members went through.  Every member walks the same symbols, so all arrive at the same depth
                // (a member whose text ends on the way arrives at the end and sorts first).
                for (u32 hop = 0; hop < chain; hop++) {
#pragma unroll
                    for (int r = 0; r < TR_PER; r++) load_window(t, (u64)sv[r] + sd[r], n, wa[r], wb[r], avl[r]);
#pragma unroll
                    for (int r = 0; r < TR_PER; r++) {
                        u64 k56;
                        u32 cnt;
                        vlc_pack<56>(tab, wa[r], wb[r], avl[r], k56, cnt);
                        sd[r] += cnt;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < TR_PER; r++) load_window(t, (u64)sv[r] + sd[r], n, wa[r], wb[r], avl[r]);
#pragma unroll
            for (int r = 0; r < TR_PER; r++) {
                u64 key;
                u32 cnt;
                vlc_pack<40>(tab, wa[r], wb[r], avl[r], key, cnt);
                const bool live = (u64)sv[r] + sd[r] < n;
                if (!live) {
                    key = (u64)(n - sv[r]);  // ended: shorter first
                    cnt = 0;
                }
                wd[r] = ((live ? 1ull : 0ull) << 52) | (key << 12) | (u64)(e0 + r);
                if (e0 + r < m) {
                    word[e0 + r] = wd[r];
                    ad[e0 + r] = (u16)(sd[r] + cnt);
                }
            }
        }
        __syncthreads();
        // ---- new position of every active element inside its group
        u32 newpos[TR_PER];
        u32 medf = 0;
        {
            u32 o = obase;
#pragma unroll
            for (int r = 0; r < TR_PER; r++) {
                o += (hf >> r) & 1u;
                newpos[r] = e0 + r;
                if (e0 + r < m) {
                    const u32 gs = gstart[o], ge = gstart[o + 1];
                    if (ge - gs <= (u32)TR_SMALL) {
                        // rank by counting; eight words of the group in flight per round trip (one at a time, the loop is a chain
                        // of LDS latencies: measured 3x the whole rest of the step)
                        u32 below = 0;
                        const u64 mineW = wd[r];
                        for (u32 q = gs; q < ge; q += 8u) {
                            u64 x[8];
#pragma unroll
                            for (u32 k = 0; k < 8u; k++) x[k] = word[q + k < ge ? q + k : ge - 1u];
#pragma unroll
                            for (u32 k = 0; k < 8u; k++) below += (q + k < ge && x[k] < mineW) ? 1u : 0u;
                        }
                        newpos[r] = gs + below;
                    } else {
                        medf |= 1u << r;
                    }
                }
            }
        }
        u32 M;
        {
            const u32 nm = (u32)__popc(medf);
            u32 q = block_excl_add<TR_NT>(nm, red, M);
            if (M) {
                u32 o = obase;
#pragma unroll
                for (int r = 0; r < TR_PER; r++) {
                    o += (hf >> r) & 1u;
                    if ((medf >> r) & 1u) {
                        const u64 gs = gstart[o];
                        sb[tr_pad(q)] = (gs << 53) | (wd[r] & ~0xFFFull) | (u64)q;
                        mact[q] = (u16)(e0 + r);
                        q++;
                    }
                }
            }
        }
        u32 mp = 0;
        if (M) {
            mp = 2;
            while (mp < M) mp <<= 1;
            for (u32 q = M + tid; q < mp; q += TR_NT) sb[tr_pad(q)] = ~0ull;
            __syncthreads();
            tr_bitonic(sb, mp);  // ends with a barrier
        }
        // ---- move: read everything that moves into registers, then write it to its new place
        u32 mv_v[TR_PER], mv_dst[TR_PER];
        u16 mv_d[TR_PER];
        u8 mv_p[TR_PER];
        u64 mv_w[TR_PER];
        u32 sm_v[TR_PER];
        u16 sm_d[TR_PER];
        u8 sm_p[TR_PER];
#pragma unroll
        for (int r = 0; r < TR_PER; r++) {
            const u32 e = e0 + r;
            if (e < m && !((medf >> r) & 1u)) {
                sm_v[r] = av[e];
                sm_d[r] = ad[e];
                sm_p[r] = apb[e];
            }
            const u32 j = e0 + r;  // sorted position among the larger groups' elements
            mv_dst[r] = 0xFFFFFFFFu;
            if (j < M) {
                const u64 x = sb[tr_pad(j)];
                const u32 src = mact[(u32)(x & 0xFFFu)];
                mv_dst[r] = mact[j];
                mv_v[r] = av[src];
                mv_d[r] = ad[src];
                mv_p[r] = apb[src];
                mv_w[r] = word[src];
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < TR_PER; r++) {
            const u32 e = e0 + r;
            if (e < m && !((medf >> r) & 1u)) {
                const u32 d = newpos[r];
                av[d] = sm_v[r];
                ad[d] = sm_d[r];
                apb[d] = sm_p[r];
                word[d] = wd[r];
            }
            if (mv_dst[r] != 0xFFFFFFFFu) {
                const u32 d = mv_dst[r];
                av[d] = mv_v[r];
                ad[d] = mv_d[r];
                apb[d] = mv_p[r];
                word[d] = mv_w[r];
            }
        }
        __syncthreads();
        // ---- new heads: where the bits just compared differ from the left neighbour's
        u32 nhf = 0;
#pragma unroll
        for (int r = 0; r < TR_PER; r++) {
            const u32 e = e0 + r;
            if (e < m) {
                const bool h = ((hf >> r) & 1u) || (word[e] >> 12) != (word[e - 1] >> 12);  // e > 0 here: element 0 is a head
                nhf |= (h ? 1u : 0u) << r;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < TR_PER; r++)
            if (e0 + r < m) ahead[e0 + r] = (u8)((nhf >> r) & 1u);
        if (tid == 0) ahead[m] = 1;
        __syncthreads();
        // ---- suffixes that are alone now are final: out they go; the others move up
        u32 keepf = 0;
        u32 kv[TR_PER];
        u16 kd[TR_PER], ks[TR_PER];
        u8 kp[TR_PER], kh[TR_PER];
#pragma unroll
        for (int r = 0; r < TR_PER; r++) {
            const u32 e = e0 + r;
            if (e < m) {
                const bool h = (nhf >> r) & 1u;
                const bool uniq = h && ahead[e + 1];
                kv[r] = av[e];
                kd[r] = ad[e];
                ks[r] = aslot[e];
                kp[r] = apb[e];
                kh[r] = (u8)h;
                if (uniq) {
                    const u64 p = a + ks[r];
                    v[p] = kv[r] | V_HEAD;
                    pb[p] = kp[r];
                    if (kv[r] == 0) counters[2] = (u32)p;
                } else {
                    keepf |= 1u << r;
                }
            }
        }
        u32 mnext;
        {
            const u32 nk = (u32)__popc(keepf);
            u32 d = block_excl_add<TR_NT>(nk, red, mnext);  // ends with a barrier: every read above is done
#pragma unroll
            for (int r = 0; r < TR_PER; r++)
                if ((keepf >> r) & 1u) {
                    av[d] = kv[r];
                    ad[d] = kd[r];
                    aslot[d] = ks[r];
                    apb[d] = kp[r];
                    ahead[d] = kh[r];
                    d++;
                }
        }
        m = mnext;
        __syncthreads();
    }
}

// ---- the tail: tiny groups that need many more windows ---------------------------------------------------------------------------
// One wave per tile's leftovers (chunk_off / chunk_cnt), a lane per suffix, as many whole groups at a time as fit the 64 lanes.  A step
// costs one gather from the text and a handful of cross-lane operations; nothing in it waits for another wave, and the kernel's small
// footprint puts dozens of waves on a CU, so the gather latency of one is covered by the others.
__global__ void __launch_bounds__(WAVE) k_bwt_tail(const u8 * __restrict__ t, u32 n, u32 * __restrict__ v, u8 * __restrict__ pb, const u32 * __restrict__ vlc,
                                                  const u32 * __restrict__ chunk_off, const u32 * __restrict__ chunk_cnt, const u32 * __restrict__ tail_v,
                                                  const u32 * __restrict__ tail_slot, const u16 * __restrict__ tail_d, const u8 * __restrict__ tail_pb,
                                                  u32 * __restrict__ counters) {
    __shared__ u32 tab[256];
    __shared__ u64 s_word[WAVE];
    __shared__ u32 s_v[WAVE];
    __shared__ u16 s_d[WAVE];
    __shared__ u8 s_pb[WAVE];
    const u32 cnt = chunk_cnt[blockIdx.x];
    if (cnt == 0) return;
    const u32 off = chunk_off[blockIdx.x];
    const u32 lane = (u32)lane_id();
#pragma unroll
    for (int k = 0; k < 4; k++) tab[lane + 64u * k] = vlc[lane + 64u * k];
    __syncthreads();
    u32 cursor = 0;
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
            const u64 w2 = s_word[lane];
            const u64 wl = s_word[lane ? lane - 1u : 0u];
            sv = s_v[lane];
            sd = s_d[lane];
            sp = s_pb[lane];
            __syncthreads();
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
    for (u32 hop = 0; hop < chain; hop++) {  // past the windows the group was formed on (see k_bwt_resolve)
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
__global__ void __launch_bounds__(BW_BLOCK) k_bwt_finish(const u8 * __restrict__ t, const u8 * __restrict__ pb, u32 n, const u32 * __restrict__ counters,
                                                        u8 * __restrict__ out, u32 * __restrict__ idx_out) {
    const u32 i = blockIdx.x * BW_BLOCK + threadIdx.x;
    if (i >= n) return;
    const u32 i0 = counters[2];
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
                                                      u32 * __restrict__ isa, u32 * __restrict__ vals_out, u32 * __restrict__ slots_out, u32 * __restrict__ grp_out) {
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
            if (!DOUBLING || rank[j] != hi[j]) isa[v[j]] = rank[j];
            if ((uniqs >> j) & 1u) {
                if (!VF) sa[sl[j]] = v[j];  // VF: the list IS the array, the suffix is in its slot already (and neighbours still read its flag)
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

__global__ void __launch_bounds__(BW_BLOCK) k_bwt_emit(const u8 * __restrict__ t, const u32 * __restrict__ sa, const u32 * __restrict__ isa, u32 n,
                                                      u8 * __restrict__ out, u32 * __restrict__ idx_out) {
    const u32 i = blockIdx.x * BW_BLOCK + threadIdx.x;
    if (i >= n) return;
    const u32 i0 = isa[0];
    if (i == 0) {
        out[0] = t[n - 1];
        *idx_out = i0 + 1;
    }
    if (i != i0) out[i < i0 ? i + 1 : i] = t[(sa[i] & V_MASK) - 1];  // (slots that were final before the deep path still carry their flag)
}

static int bits_for(u64 x) {
    int b = 0;
    while (x) { b++; x >>= 1; }
    return b ? b : 1;
}

size_t bwt_workspace_bytes(u64 n) {
    // 2 keys (16) + 2 suffix arrays (8) + payload (1) + ISA (4) + 2 slot lists (8) + ranks (4) + tile words (4) + a third suffix list (4)
    return n * (16 + 8 + 1 + 4 + 8 + 4 + 4 + 4) + radix_temp_bytes(n) + scan_temp_words(n) * 4 + (1u << 20) + 16384;
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

s32 bwt_forward(const u8 * d_in, u32 n, u8 * d_out, Arena & tmp, hipStream_t s, BwtStats * stats) {
    if (n == 0) return 0;
    if (n == 1) {
        HIP_CHECK(hipMemcpyAsync(d_out, d_in, 1, hipMemcpyDeviceToDevice, s));
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
    u32 * carry = tmp.take<u32>(tiles + 1);
    u8 * dirty = tmp.take<u8>(tiles + 2);
    u32 * hbits = tmp.take<u32>(((size_t)n + 63) / 64 * 2 + (size_t)TR_S / 32 + 8);  // snapshot of the head flags, whole 64-slot words of every anchor tile
    u32 * d_words = tmp.take<u32>(8);   // counters [0] big elements, [1] left ambiguous by the resolve / tail kernels, [2] slot of suffix 0, [3] a list overflowed, [4] tail elements; [6] scan total, [7] primary index
    u32 * chunk_off = tmp.take<u32>(tiles + 1);
    u32 * chunk_cnt = tmp.take<u32>(tiles + 1);
    u32 * d_vlc = tmp.take<u32>(256);
    u32 * d_hist = tmp.take<u32>(256);
    BwtStats st;

    auto grid = [](u64 m) { return dim3((u32)((m + BW_BLOCK - 1) / BW_BLOCK)); };

    // ---- codes
    u32 h_hist[256], h_vlc[256];
    HIP_CHECK(hipMemsetAsync(d_hist, 0, 256 * sizeof(u32), s));
    HIP_CHECK(hipMemsetAsync(d_words, 0, 8 * sizeof(u32), s));
    launch(k_bwt_sym_hist, grid(((u64)n + 63) / 64), dim3(BW_BLOCK), 0, s, d_in, n, d_hist);
    HIP_CHECK(hipMemcpyAsync(h_hist, d_hist, sizeof h_hist, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    vlc_build(h_hist, h_vlc);
    HIP_CHECK(hipMemcpyAsync(d_vlc, h_vlc, sizeof h_vlc, hipMemcpyHostToDevice, s));

    // ---- round 0: all suffixes by their first 56 code bits
    launch(k_bwt_vlc_keys, grid(((u64)n + 7) / 8), dim3(BW_BLOCK), 0, s, d_in, n, (const u32 *)d_vlc, key[0]);
    const int cur = sort_iota<u64>(key[0], key[0], key[1], val[0], val[1], n, 8, 64, tmp, s, st);
    st.sorted_elements += n;
    st.rounds++;
    u32 * V = val[cur];
    launch(k_bwt_heads, dim3(tiles), dim3(BW_BLOCK), 0, s, (const u64 *)key[cur], V, pb, n, tile_last, hbits, d_words);
    launch(k_bwt_spine_max, dim3(1), dim3(SP_BLOCK), 0, s, (const u32 *)tile_last, tiles, carry);

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
    // the tail kernel's input (written by the resolve kernel, dead before the big round of the same pass starts): key[1] again
    const u32 tail_cap = n / 4;
    u32 * tail_v = reinterpret_cast<u32 *>(key[1]);
    u32 * tail_slot = tail_v + tail_cap;
    u16 * tail_d = reinterpret_cast<u16 *>(tail_slot + tail_cap);
    u8 * tail_pb = reinterpret_cast<u8 *>(tail_d + tail_cap);

    u32 g = 7;            // symbols every group is known to share at least (7 per 56-bit window)
    bool deep = false;    // fall back to rank doubling
    u32 h_words[8];
    for (int pass = 0;; pass++) {
        HIP_CHECK(hipMemsetAsync(chunk_cnt, 0, ((size_t)tiles + 1) * sizeof(u32), s));
        launch(k_bwt_resolve, dim3(tiles), dim3(TR_NT), 0, s, d_in, n, V, pb, (const u32 *)hbits, (const u32 *)carry, (const u8 *)(pass ? dirty : nullptr),
               (const u32 *)d_vlc, big_slot, big_hp, big_cap, chunk_off, chunk_cnt, tail_v, tail_slot, tail_d, tail_pb, tail_cap, d_words, (u32)pass + 1u);
        launch(k_bwt_tail, dim3(tiles), dim3(WAVE), 0, s, d_in, n, V, pb, (const u32 *)d_vlc, (const u32 *)chunk_off, (const u32 *)chunk_cnt, (const u32 *)tail_v,
               (const u32 *)tail_slot, (const u16 *)tail_d, (const u8 *)tail_pb, d_words);
        HIP_CHECK(hipMemcpyAsync(h_words, d_words, sizeof h_words, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        const u32 nb = h_words[0];
        if (getenv("BZ3_BWT_TRACE")) fprintf(stderr, "[bwt] n %u pass %d depth %u: %u suffixes in groups > %d, %u through the tail kernel, %u given up, overflow %u\n", n, pass, g, nb, TR_G, h_words[4], h_words[1], h_words[3]);
        if (nb == 0) {  // no group left that is too large: done, unless the resolve kernel gave some up (counted over all passes)
            deep = h_words[1] != 0;
            break;
        }
        const char * env_rounds = getenv("BZ3_BWT_BIG_ROUNDS");  // tests only: 0 = big groups go straight to the deep path
        const int max_big = env_rounds ? atoi(env_rounds) : 1;  // one more window takes text from ~12 % of its suffixes in big groups to ~1 %; what is left is deep and goes to rank doubling
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
        launch(k_bwt_reduce_heads, dim3(tiles), dim3(BW_BLOCK), 0, s, (const u32 *)V, n, tile_last, hbits);
        launch(k_bwt_spine_max, dim3(1), dim3(SP_BLOCK), 0, s, (const u32 *)tile_last, tiles, carry);
    }

    u32 idx = 0;
    if (!deep) {
        launch(k_bwt_finish, grid(n), dim3(BW_BLOCK), 0, s, d_in, (const u8 *)pb, n, (const u32 *)d_words, d_out, d_words + 7);
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
        // every group still ambiguous shares at least h symbols: 7 per window of the big rounds, but a group the resolve kernel
        // handed back early (more than 64 members after its three steps) is only known to share the first window's 7 + 3 x 5.
        u32 m = n, h = h_words[1] ? (g < 22u ? g : 22u) : g;
        if (getenv("BZ3_BWT_TRACE")) fprintf(stderr, "[bwt] n %u deep path from depth %u\n", n, h);
        {
            const u32 tl = (u32)(((u64)m + BG_TILE - 1) / BG_TILE);
            launch(k_bg_reduce<true>, dim3(tl), dim3(BW_BLOCK), 0, s, (const u64 *)nullptr, (const u32 *)V, m, tile_head, tile_keep);
            launch(k_bg_spine, dim3(1), dim3(BG_SPINE), 0, s, tile_head, tile_keep, tl, d_words + 6);
            launch(k_bg_apply<true, false>, dim3(tl), dim3(BW_BLOCK), 0, s, (const u64 *)nullptr, (const u32 *)V, (const u32 *)nullptr, m, (const u32 *)tile_head,
                   (const u32 *)tile_keep, sa, isa, vv[0], slot[0], grp);
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
                   (const u32 *)tile_head, (const u32 *)tile_keep, sa, isa, vfree, slot[scur ^ 1], grp);
            scur ^= 1;
            vact = vfree;
            vfree = vsorted;
            if (h >= 0x40000000u) throw HipError{hipErrorUnknown, "suffix sort did not converge", __FILE__, __LINE__};
            h *= 2;
        }
        launch(k_bwt_emit, grid(n), dim3(BW_BLOCK), 0, s, d_in, (const u32 *)sa, (const u32 *)isa, n, d_out, d_words + 7);
    }
    HIP_CHECK(hipMemcpyAsync(&idx, d_words + 7, 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    tmp.release(mk);
    if (stats) *stats = st;
    return (s32)idx;
}

}  // namespace bz3
