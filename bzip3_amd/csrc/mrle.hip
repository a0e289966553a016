// mrle.hip -- bzip3's "mRLE" byte filter, encode and decode, as data-parallel scans on gfx950.
// Replaces mrlec (reference src/libbz3.c:264-301) and mrled (:303-329), both sequential byte loops.
//
// Encode, closed form.  For position i let r = i - (start of its maximal
// run) and c = in[i]:
//   gain[c] += (r == 0) ? -1 : (r % 255 != 0)                      -> symbol c is "flagged" iff gain[c] > 0
//   emitted bytes at i: unflagged: c.   flagged: [c if r == 0] [0xFF if r > 0 && r % 255 == 0]
//                                                [r % 255 if i is the last byte of its run]
// r comes from a running maximum of run-head positions (tile scan + carry), the output offsets from
// an exclusive sum of emitted-byte counts.  Every pass streams the input once, 16 bytes per lane.
//
// Decode: the stream is a 2-state automaton (S = expecting a symbol, L = inside a length sequence).
// Each byte is a map {S,L}->{S,L}; tiles compose their maps, a spine kernel threads the state through
// the tiles, then every byte knows whether it is a symbol or a length byte, how many output bytes it
// stands for, and (via a running maximum of symbol positions) which symbol it repeats.
#include "prims.hpp"
#include "sort.hpp"
#include "stages.hpp"

namespace bz3 {

constexpr int MR_BLOCK = 256;
constexpr int MR_ITEMS = 16;
constexpr int MR_TILE = MR_BLOCK * MR_ITEMS;  // 4096 bytes per workgroup

__device__ __forceinline__ void load16(const u8 * __restrict__ in, u64 base, u64 n, u8 (&b)[MR_ITEMS]) {
    if (base + MR_ITEMS <= n && ((reinterpret_cast<uintptr_t>(in) + base) & 15) == 0) {
        uint4 q = *reinterpret_cast<const uint4 *>(in + base);
        const u32 w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < MR_ITEMS; k++) b[k] = (u8)(w[k >> 2] >> (8 * (k & 3)));
    } else {
#pragma unroll
        for (int k = 0; k < MR_ITEMS; k++) b[k] = (base + k < n) ? in[base + k] : (u8)0;
    }
}

// ------------------------------------------------------------------------------------------------
// encode
// ------------------------------------------------------------------------------------------------

// tile_head[t] = 1 + (largest run-head position inside tile t), or 0 if the tile has none.
__global__ void __launch_bounds__(MR_BLOCK) k_mrle_heads(const u8 * __restrict__ in, u32 n, u32 * __restrict__ tile_head) {
    __shared__ u32 lds[MR_BLOCK / WAVE];
    const u64 base = (u64)blockIdx.x * MR_TILE + (u64)threadIdx.x * MR_ITEMS;
    u8 b[MR_ITEMS];
    load16(in, base, n, b);
    u8 prev = (base > 0 && base < n) ? in[base - 1] : (u8)0;
    u32 best = 0;
#pragma unroll
    for (int k = 0; k < MR_ITEMS; k++) {
        const u64 i = base + k;
        if (i < n && (i == 0 || b[k] != prev)) best = (u32)i + 1;
        prev = b[k];
    }
    best = block_max<MR_BLOCK>(best, lds);
    if (threadIdx.x == 0) tile_head[blockIdx.x] = best;
}

// In-place inclusive running maximum over the tile table, made exclusive: carry[t] = last head before tile t (+1).
__global__ void __launch_bounds__(1024) k_mrle_head_spine(u32 * __restrict__ tile_head, u32 tiles) {
    __shared__ u32 lds[1024 / WAVE];
    __shared__ u32 carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (u32 base = 0; base < tiles; base += 1024) {
        const u32 t = base + threadIdx.x;
        const u32 v = t < tiles ? tile_head[t] : 0u;
        u32 incl = block_incl_max<1024>(v, lds);
        const u32 carry = carry_s;
        if (incl < carry) incl = carry;
        // exclusive value for tile t = inclusive value of tile t-1
        u32 excl = __shfl_up(incl, 1u);
        __shared__ u32 wave_last[1024 / WAVE];
        if (lane_id() == WAVE - 1) wave_last[wave_id()] = incl;
        __syncthreads();
        if (lane_id() == 0) excl = wave_id() == 0 ? carry : wave_last[wave_id() - 1];
        if (t < tiles) tile_head[t] = excl;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = incl;
        __syncthreads();
    }
}

// Per-thread view of 16 consecutive bytes with their run offsets r.
struct MrleView {
    u8 b[MR_ITEMS];
    u32 r[MR_ITEMS];
    bool last[MR_ITEMS];
};

// Computes run offsets for the calling thread's 16 bytes.  carry = 1 + last head position before this tile (0 = none).
template <int BLOCK>
__device__ __forceinline__ void mrle_view(const u8 * __restrict__ in, u32 n, u32 carry, u32 * lds, MrleView & v) {
    const u64 base = (u64)blockIdx.x * MR_TILE + (u64)threadIdx.x * MR_ITEMS;
    load16(in, base, n, v.b);
    u8 prev = (base > 0 && base < n) ? in[base - 1] : (u8)0;
    u32 head1[MR_ITEMS];  // 1 + latest head position at or before item k inside this thread (0 = none)
    u32 run = 0;
#pragma unroll
    for (int k = 0; k < MR_ITEMS; k++) {
        const u64 i = base + k;
        if (i < n && (i == 0 || v.b[k] != prev)) run = (u32)i + 1;
        head1[k] = run;
        prev = v.b[k];
    }
    // running maximum across threads (exclusive for this thread)
    u32 incl = block_incl_max<BLOCK>(run, lds);
    u32 excl = __shfl_up(incl, 1u);
    __shared__ u32 wl[BLOCK / WAVE];
    if (lane_id() == WAVE - 1) wl[wave_id()] = incl;
    __syncthreads();
    if (lane_id() == 0) excl = wave_id() == 0 ? 0u : wl[wave_id() - 1];
    if (excl < carry) excl = carry;
    __syncthreads();
    const u8 next = (base + MR_ITEMS < n) ? in[base + MR_ITEMS] : (u8)0;
#pragma unroll
    for (int k = 0; k < MR_ITEMS; k++) {
        const u64 i = base + k;
        const u32 h1 = head1[k] ? head1[k] : excl;  // always >= 1 for i < n
        v.r[k] = (i < n) ? (u32)i - (h1 - 1) : 0u;
        const u8 nb = (k + 1 < MR_ITEMS) ? v.b[(k + 1) & (MR_ITEMS - 1)] : next;
        v.last[k] = (i < n) && (i + 1 == n || nb != v.b[k]);
    }
}

__global__ void __launch_bounds__(MR_BLOCK) k_mrle_gain(const u8 * __restrict__ in, u32 n, const u32 * __restrict__ carry, s32 * __restrict__ gain) {
    __shared__ u32 lds[MR_BLOCK / WAVE];
    __shared__ s32 g[256];
    g[threadIdx.x] = 0;
    __syncthreads();
    MrleView v;
    mrle_view<MR_BLOCK>(in, n, carry[blockIdx.x], lds, v);
    const u64 base = (u64)blockIdx.x * MR_TILE + (u64)threadIdx.x * MR_ITEMS;
    s32 acc = 0;
    int sym = -1;
#pragma unroll
    for (int k = 0; k < MR_ITEMS; k++) {
        if (base + k >= n) break;
        if ((int)v.b[k] != sym) {
            if (sym >= 0 && acc != 0) atomicAdd(&g[sym], acc);
            sym = v.b[k];
            acc = 0;
        }
        acc += (v.r[k] == 0) ? -1 : ((v.r[k] % 255u) != 0u ? 1 : 0);
    }
    if (sym >= 0 && acc != 0) atomicAdd(&g[sym], acc);
    __syncthreads();
    if (g[threadIdx.x] != 0) atomicAdd(&gain[threadIdx.x], g[threadIdx.x]);
}

__device__ __forceinline__ u32 mrle_emit_count(bool flagged, u32 r, bool last) {
    if (!flagged) return 1u;
    return (r == 0 ? 1u : 0u) + ((r > 0 && r % 255u == 0) ? 1u : 0u) + (last ? 1u : 0u);
}

// MODE 0: tile sums of emitted bytes.  MODE 1: write the output (tile_off = exclusive prefix of tile sums).
template <int MODE>
__global__ void __launch_bounds__(MR_BLOCK) k_mrle_emit(const u8 * __restrict__ in, u32 n, const u32 * __restrict__ carry, const s32 * __restrict__ gain,
                                                       u32 * __restrict__ tile_sum, u8 * __restrict__ out) {
    __shared__ u32 lds[MR_BLOCK / WAVE + 1];
    __shared__ u8 flag[256];
    flag[threadIdx.x] = gain[threadIdx.x] > 0;
    __syncthreads();
    MrleView v;
    mrle_view<MR_BLOCK>(in, n, carry[blockIdx.x], lds, v);
    const u64 base = (u64)blockIdx.x * MR_TILE + (u64)threadIdx.x * MR_ITEMS;
    u32 cnt = 0;
#pragma unroll
    for (int k = 0; k < MR_ITEMS; k++)
        if (base + k < n) cnt += mrle_emit_count(flag[v.b[k]], v.r[k], v.last[k]);
    u32 tot;
    u32 pre = block_excl_add<MR_BLOCK>(cnt, lds, tot);
    if (MODE == 0) {
        if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x < 32) {  // 32-byte bitmap header (:278-282)
        u8 m = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) m |= (u8)(flag[threadIdx.x * 8 + j] << j);
        out[threadIdx.x] = m;
    }
    u64 o = 32ull + tile_sum[blockIdx.x] + pre;
    // Sixteen bytes that all pass through unchanged (no flagged value among them -- every lane of a text block but a handful) leave as ONE 16-byte store at
    // whatever alignment `o` has (round 6: sixteen byte stores per lane were the kernel: 0.94 ms per 256 MiB block).
    if (base + MR_ITEMS <= n && cnt == (u32)MR_ITEMS) {
        bool plain = true;
        PackedU128 q;
        q.v[0] = q.v[1] = q.v[2] = q.v[3] = 0u;
#pragma unroll
        for (int k = 0; k < MR_ITEMS; k++) {
            plain = plain && !flag[v.b[k]];
            q.v[k >> 2] |= (u32)v.b[k] << (8 * (k & 3));
        }
        if (plain) {
            *reinterpret_cast<PackedU128 *>(out + o) = q;
            return;
        }
    }
#pragma unroll
    for (int k = 0; k < MR_ITEMS; k++) {
        if (base + k >= n) break;
        const u8 c = v.b[k];
        const u32 r = v.r[k];
        if (!flag[c]) {
            out[o++] = c;
        } else {
            if (r == 0) out[o++] = c;
            if (r > 0 && r % 255u == 0) out[o++] = 255;
            if (v.last[k]) out[o++] = (u8)(r % 255u);
        }
    }
}

void mrle_encode_size(const u8 * d_in, u32 n, MrleEncScratch & sc, Arena & tmp, hipStream_t s) {
    const u32 tiles = n ? (n + MR_TILE - 1) / MR_TILE : 1u;  // at least one workgroup: it writes the bitmap header
    sc.tiles = tiles;
    sc.carry = tmp.take<u32>(tiles + 1);
    sc.tile_sum = tmp.take<u32>(tiles + 1);
    sc.gain = tmp.take<s32>(256);
    sc.total = tmp.take<u32>(1);
    HIP_CHECK(hipMemsetAsync(sc.gain, 0, 256 * sizeof(s32), s));
    launch(k_mrle_heads, dim3(tiles), dim3(MR_BLOCK), 0, s, d_in, n, sc.carry);
    launch(k_mrle_head_spine, dim3(1), dim3(1024), 0, s, sc.carry, tiles);
    launch(k_mrle_gain, dim3(tiles), dim3(MR_BLOCK), 0, s, d_in, n, (const u32 *)sc.carry, sc.gain);
    launch(k_mrle_emit<0>, dim3(tiles), dim3(MR_BLOCK), 0, s, d_in, n, (const u32 *)sc.carry, (const s32 *)sc.gain, sc.tile_sum, (u8 *)nullptr);
    exclusive_scan_u32(sc.tile_sum, tiles, sc.total, tmp, s);
}

void mrle_encode_write(const u8 * d_in, u32 n, const MrleEncScratch & sc, u8 * d_out, hipStream_t s) {
    launch(k_mrle_emit<1>, dim3(sc.tiles), dim3(MR_BLOCK), 0, s, d_in, n, (const u32 *)sc.carry, (const s32 *)sc.gain, sc.tile_sum, d_out);
}

// ------------------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------------------
// State maps are packed as 2 bits: bit0 = next state when entering in S, bit1 = next state when
// entering in L (0 = S, 1 = L).  compose(f, g) = "f then g".
__device__ __forceinline__ u32 mr_compose(u32 f, u32 g) {
    const u32 a = (g >> (f & 1u)) & 1u;
    const u32 b = (g >> ((f >> 1) & 1u)) & 1u;
    return a | (b << 1);
}
__device__ __forceinline__ u32 mr_byte_map(u8 byte, bool flagged) { return (flagged ? 1u : 0u) | ((byte == 255) ? 2u : 0u); }

template <typename F>
__device__ __forceinline__ u32 wave_incl_compose(u32 f, F) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        u32 t = __shfl_up(f, (unsigned)d);
        if (l >= d) f = mr_compose(t, f);
    }
    return f;
}

// The data region of the encoded stream starts at byte 32; tile t covers stream bytes [32 + t*TILE, ...).
struct MrdView {
    u8 b[MR_ITEMS];
    u32 fmap[MR_ITEMS];   // inclusive composed map of this thread's items up to k
};

__device__ __forceinline__ bool mr_flagged(const u8 * __restrict__ enc, u8 c) { return (enc[c >> 3] >> (c & 7)) & 1; }

// Pass 1: per tile, the composed map and (for both entry states) 1 + last symbol position and the output count.
__global__ void __launch_bounds__(MR_BLOCK) k_mrd_tile_maps(const u8 * __restrict__ enc, u32 m, u32 * __restrict__ tile_map) {
    __shared__ u8 flag[256];
    __shared__ u32 wmap[MR_BLOCK / WAVE];
    flag[threadIdx.x] = mr_flagged(enc, (u8)threadIdx.x);
    __syncthreads();
    const u64 base = 32ull + (u64)blockIdx.x * MR_TILE + (u64)threadIdx.x * MR_ITEMS;
    u8 b[MR_ITEMS];
    load16(enc, base, m, b);
    u32 f = 2u;  // identity: S->S, L->L
#pragma unroll
    for (int k = 0; k < MR_ITEMS; k++)
        if (base + k < m) f = mr_compose(f, mr_byte_map(b[k], flag[b[k]]));
    f = wave_incl_compose(f, 0);
    if (lane_id() == WAVE - 1) wmap[wave_id()] = f;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 t = 2u;
        for (int w = 0; w < MR_BLOCK / WAVE; w++) t = mr_compose(t, wmap[w]);
        tile_map[blockIdx.x] = t;
    }
}

// Spine: thread the automaton state through the tiles (sequential composition, chunked by 1024).
__global__ void __launch_bounds__(1024) k_mrd_spine(u32 * __restrict__ tile_map, u32 tiles) {
    __shared__ u32 wl[1024 / WAVE];
    __shared__ u32 state_s;
    if (threadIdx.x == 0) state_s = 0;  // S
    __syncthreads();
    for (u32 base = 0; base < tiles; base += 1024) {
        const u32 t = base + threadIdx.x;
        u32 f = t < tiles ? tile_map[t] : 2u;
        u32 incl = wave_incl_compose(f, 0);
        if (lane_id() == WAVE - 1) wl[wave_id()] = incl;
        __syncthreads();
        u32 pre = 2u;
        for (int w = 0; w < wave_id(); w++) pre = mr_compose(pre, wl[w]);
        incl = mr_compose(pre, incl);
        // entry state of tile t = state after tiles < t
        u32 excl = __shfl_up(incl, 1u);
        if (lane_id() == 0) excl = pre;
        const u32 st0 = state_s;
        const u32 entry = (excl >> st0) & 1u;
        const u32 exit_state = (incl >> st0) & 1u;
        __syncthreads();
        if (t < tiles) tile_map[t] = entry;
        if (threadIdx.x == 1023) state_s = exit_state;
        __syncthreads();
    }
}

// Shared per-tile analysis for passes 2 and 3: state before each byte, output count of each byte,
// and 1 + position of the governing symbol (for length bytes).
struct MrdItems {
    u8 b[MR_ITEMS];
    bool is_sym[MR_ITEMS];
    u32 cnt[MR_ITEMS];
    u32 sym1[MR_ITEMS];  // 1 + stream position of the symbol this byte repeats (valid for length bytes)
};

template <int BLOCK>
__device__ __forceinline__ void mrd_analyse(const u8 * __restrict__ enc, u32 m, u32 entry_state, u32 sym_carry, const u8 * flag, u32 * lds,
                                            MrdItems & it, u32 & thread_cnt, u32 & last_len_pos1, u32 * st_end = nullptr) {
    __shared__ u32 wmap[BLOCK / WAVE];
    __shared__ u32 wl[BLOCK / WAVE];
    const u64 base = 32ull + (u64)blockIdx.x * MR_TILE + (u64)threadIdx.x * MR_ITEMS;
    load16(enc, base, m, it.b);
    u32 f = 2u;
#pragma unroll
    for (int k = 0; k < MR_ITEMS; k++)
        if (base + k < m) f = mr_compose(f, mr_byte_map(it.b[k], flag[it.b[k]]));
    u32 incl = wave_incl_compose(f, 0);
    if (lane_id() == WAVE - 1) wmap[wave_id()] = incl;
    __syncthreads();
    u32 pre = 2u;
    for (int w = 0; w < wave_id(); w++) pre = mr_compose(pre, wmap[w]);
    u32 excl = __shfl_up(incl, 1u);
    if (lane_id() == 0) excl = 2u;
    excl = mr_compose(pre, excl);
    u32 st = (excl >> entry_state) & 1u;  // state before this thread's first byte
    u32 symrun = 0;                       // 1 + latest symbol position inside this thread
    u32 local_sym[MR_ITEMS];
    thread_cnt = 0;
    last_len_pos1 = 0;
#pragma unroll
    for (int k = 0; k < MR_ITEMS; k++) {
        const u64 i = base + k;
        const bool ok = i < m;
        const u8 c = it.b[k];
        it.is_sym[k] = ok && st == 0;
        u32 cnt = 0;
        if (ok) {
            if (st == 0) {
                symrun = (u32)i + 1;
                cnt = flag[c] ? 0u : 1u;
                st = flag[c] ? 1u : 0u;
            } else {
                cnt = (c == 255) ? 255u : (u32)c + 1u;
                st = (c == 255) ? 1u : 0u;
                last_len_pos1 = (u32)i + 1;
            }
        }
        it.cnt[k] = cnt;
        local_sym[k] = symrun;
        thread_cnt += cnt;
    }
    if (st_end) *st_end = st;  // the automaton's state behind this thread's last byte
    u32 sincl = block_incl_max<BLOCK>(symrun, lds);
    u32 sexcl = __shfl_up(sincl, 1u);
    if (lane_id() == WAVE - 1) wl[wave_id()] = sincl;
    __syncthreads();
    if (lane_id() == 0) sexcl = wave_id() == 0 ? 0u : wl[wave_id() - 1];
    if (sexcl < sym_carry) sexcl = sym_carry;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MR_ITEMS; k++) it.sym1[k] = local_sym[k] ? local_sym[k] : sexcl;
}

// Pass 2: per tile output-count sum (saturated), 1 + last symbol position, 1 + last length-byte position.
__global__ void __launch_bounds__(MR_BLOCK) k_mrd_tile_counts(const u8 * __restrict__ enc, u32 m, const u32 * __restrict__ tile_entry, u32 * __restrict__ tile_cnt,
                                                              u32 * __restrict__ tile_sym, u32 * __restrict__ last_len1) {
    __shared__ u8 flag[256];
    __shared__ u32 lds[MR_BLOCK / WAVE + 1];
    flag[threadIdx.x] = mr_flagged(enc, (u8)threadIdx.x);
    __syncthreads();
    MrdItems it;
    u32 tc, ll, st_end;
    mrd_analyse<MR_BLOCK>(enc, m, tile_entry[blockIdx.x], 0u, flag, lds, it, tc, ll, &st_end);
    {  // the state behind the stream's LAST byte, for k_mrd_tail: 1 + state (0 = not written)
        const u64 b0 = 32ull + (u64)blockIdx.x * MR_TILE + (u64)threadIdx.x * MR_ITEMS;
        if (b0 < (u64)m && (u64)m <= b0 + MR_ITEMS) last_len1[1] = st_end + 1u;
    }
    u32 symrun = 0;
#pragma unroll
    for (int k = 0; k < MR_ITEMS; k++)
        if (it.is_sym[k]) symrun = (u32)(32ull + (u64)blockIdx.x * MR_TILE + (u64)threadIdx.x * MR_ITEMS + k) + 1;
    const u32 tot = block_sum<MR_BLOCK>(tc, lds);
    const u32 smax = block_max<MR_BLOCK>(symrun, lds);
    const u32 lmax = block_max<MR_BLOCK>(ll, lds);
    if (threadIdx.x == 0) {
        tile_cnt[blockIdx.x] = tot;  // <= 4096 * 256, no overflow inside a tile
        tile_sym[blockIdx.x] = smax;
        if (lmax) atomicMax(last_len1, lmax);
    }
}

// Spine 2: exclusive saturating sum of tile counts and exclusive running max of symbol positions.
__global__ void __launch_bounds__(1024) k_mrd_spine2(u32 * __restrict__ tile_cnt, u32 * __restrict__ tile_sym, u32 tiles, u32 cap, u32 * __restrict__ total) {
    __shared__ u64 lds64[1024 / WAVE + 1];
    __shared__ u32 lds[1024 / WAVE];
    __shared__ u32 wl[1024 / WAVE];
    __shared__ u64 sum_s;
    __shared__ u32 sym_s;
    if (threadIdx.x == 0) { sum_s = 0; sym_s = 0; }
    __syncthreads();
    for (u32 base = 0; base < tiles; base += 1024) {
        const u32 t = base + threadIdx.x;
        const u64 c = t < tiles ? tile_cnt[t] : 0u;
        const u32 sy = t < tiles ? tile_sym[t] : 0u;
        u64 tot;
        u64 pre = block_excl_add<1024, u64>(c, lds64, tot);
        u32 sincl = block_incl_max<1024>(sy, lds);
        u32 sexcl = __shfl_up(sincl, 1u);
        if (lane_id() == WAVE - 1) wl[wave_id()] = sincl;
        __syncthreads();
        if (lane_id() == 0) sexcl = wave_id() == 0 ? 0u : wl[wave_id() - 1];
        const u64 s0 = sum_s;
        const u32 y0 = sym_s;
        if (sexcl < y0) sexcl = y0;
        if (sincl < y0) sincl = y0;
        u64 off = s0 + pre;
        if (off > cap) off = cap;
        __syncthreads();
        if (t < tiles) { tile_cnt[t] = (u32)off; tile_sym[t] = sexcl; }
        if (threadIdx.x == 1023) { sum_s = s0 + tot; sym_s = sincl; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = (u32)(sum_s > cap ? cap : sum_s);
}

// Pass 3: fill the output.
__global__ void __launch_bounds__(MR_BLOCK) k_mrd_fill(const u8 * __restrict__ enc, u32 m, const u32 * __restrict__ tile_entry, const u32 * __restrict__ tile_off,
                                                       const u32 * __restrict__ tile_sym, u8 * __restrict__ out, u32 outlen) {
    __shared__ u8 flag[256];
    __shared__ u32 lds[MR_BLOCK / WAVE + 1];
    flag[threadIdx.x] = mr_flagged(enc, (u8)threadIdx.x);
    __syncthreads();
    MrdItems it;
    u32 tc, ll;
    mrd_analyse<MR_BLOCK>(enc, m, tile_entry[blockIdx.x], tile_sym[blockIdx.x], flag, lds, it, tc, ll);
    u32 tot;
    u32 pre = block_excl_add<MR_BLOCK>(tc, lds, tot);
    u64 o = (u64)tile_off[blockIdx.x] + pre;
    // Sixteen plain symbols (every lane of a stream that hardly used the filter) leave as ONE 16-byte store at whatever alignment `o` has (round 6)
    if (tc == (u32)MR_ITEMS && o + MR_ITEMS <= (u64)outlen) {
        bool plain = true;
        PackedU128 q;
        q.v[0] = q.v[1] = q.v[2] = q.v[3] = 0u;
#pragma unroll
        for (int k = 0; k < MR_ITEMS; k++) {
            plain = plain && it.is_sym[k] && it.cnt[k] == 1u;
            q.v[k >> 2] |= (u32)it.b[k] << (8 * (k & 3));
        }
        if (plain) {
            *reinterpret_cast<PackedU128 *>(out + o) = q;
            return;
        }
    }
#pragma unroll 1
    for (int k = 0; k < MR_ITEMS; k++) {
        const u32 cnt = it.cnt[k];
        if (cnt == 0) continue;
        const u8 c = it.is_sym[k] ? it.b[k] : enc[it.sym1[k] - 1];
        for (u32 j = 0; j < cnt && o + j < outlen; j++) out[o + j] = c;
        o += cnt;
    }
}

// Reference quirk (:320-322): a length sequence cut off by the end of the stream still adds
// (last length byte read anywhere before) + 1 copies.  `total` holds the copies counted so far.
__global__ void k_mrd_tail(const u8 * __restrict__ enc, u32 m, const u32 * __restrict__ tile_entry, const u32 * __restrict__ tile_sym, u32 tiles,
                           const u32 * __restrict__ last_len1, u8 * __restrict__ out, u32 outlen, u32 * __restrict__ total) {
    if (threadIdx.x != 0 || m <= 32) return;
    // The stream ended cleanly (in state S) almost always, and k_mrd_tile_counts has left the final state behind the length word: nothing to replay then
    // (round 6: the replay below -- up to 4096 dependent pairs of global loads on one lane -- cost every block 0.5 ms to find that out).
    if (last_len1[1] == 1u) return;
    // final automaton state and governing symbol: replay the last tile from its entry state
    u32 st = tile_entry[tiles - 1];
    u32 sym1 = tile_sym[tiles - 1];  // 1 + last symbol position before the last tile
    const u32 start = 32u + (tiles - 1) * MR_TILE;
    for (u32 i = start; i < m; i++) {
        const u8 c = enc[i];
        if (st == 0) { sym1 = i + 1; st = mr_flagged(enc, c) ? 1u : 0u; }
        else st = (c == 255) ? 1u : 0u;
    }
    if (st != 1 || sym1 == 0) return;  // stream ended cleanly
    const u32 ll = *last_len1;  // 1 + position of the last length byte read (0 = none => pc = -1)
    const s32 pc = ll ? (s32)enc[ll - 1] : -1;
    const u32 extra = (u32)(pc + 1);
    u32 t = *total;
    const u8 c = enc[sym1 - 1];
    for (u32 j = 0; j < extra && t < outlen; j++) out[t++] = c;
    *total = t;
}

void mrle_decode(const u8 * d_enc, u32 m, u8 * d_out, u32 outlen, u32 * d_total, Arena & tmp, hipStream_t s) {
    // d_total receives the number of bytes produced, capped at outlen (success iff == outlen; the
    // `maxin < 32` failure of :310 is decided by the caller).
    if (m <= 32) {
        HIP_CHECK(hipMemsetAsync(d_total, 0, 4, s));
        return;
    }
    const u32 tiles = (m - 32 + MR_TILE - 1) / MR_TILE;
    size_t mk = tmp.mark();
    u32 * tile_entry = tmp.take<u32>(tiles + 1);
    u32 * tile_cnt = tmp.take<u32>(tiles + 1);
    u32 * tile_sym = tmp.take<u32>(tiles + 1);
    u32 * last_len1 = tmp.take<u32>(2);  // [0] 1 + position of the last length byte, [1] 1 + the automaton's state behind the stream's last byte
    HIP_CHECK(hipMemsetAsync(last_len1, 0, 8, s));
    launch(k_mrd_tile_maps, dim3(tiles), dim3(MR_BLOCK), 0, s, d_enc, m, tile_entry);
    launch(k_mrd_spine, dim3(1), dim3(1024), 0, s, tile_entry, tiles);
    launch(k_mrd_tile_counts, dim3(tiles), dim3(MR_BLOCK), 0, s, d_enc, m, (const u32 *)tile_entry, tile_cnt, tile_sym, last_len1);
    launch(k_mrd_spine2, dim3(1), dim3(1024), 0, s, tile_cnt, tile_sym, tiles, outlen, d_total);
    launch(k_mrd_fill, dim3(tiles), dim3(MR_BLOCK), 0, s, d_enc, m, (const u32 *)tile_entry, (const u32 *)tile_cnt, (const u32 *)tile_sym, d_out, outlen);
    launch(k_mrd_tail, dim3(1), dim3(64), 0, s, d_enc, m, (const u32 *)tile_entry, (const u32 *)tile_sym, tiles, (const u32 *)last_len1, d_out, outlen, d_total);
    tmp.release(mk);
}

}  // namespace bz3
