// sort.hip -- device-wide exclusive scan + stable 8-bit-digit LSD radix sort for gfx950.
//
// Radix pass = three kernels (all HBM-streaming, no MFMA: byte/integer work):
//   k_rs_hist    : per-tile digit histogram, LDS-binned, written digit-major  hist[d * tiles + t]
//   scan         : device-wide exclusive scan of the digit-major table => global offset of every
//                  (digit, tile) bucket
//   k_rs_scatter : re-reads the tile, ranks each key among equal digits with wave64 ballots
//                  (match-any over the 8 digit bits + popcount below the lane), adds the per-wave
//                  running counts kept in LDS, orders the tile by digit in LDS and writes key+value out in runs.  The
//                  tile is walked wave-striped (wave w owns keys [1024w, 1024w+1024), 16 rounds of
//                  64 consecutive keys) so the order of equal digits is preserved: the pass is stable.
// Algorithmic traffic per pass and element: sizeof(K) (hist) + sizeof(K)+4 (read) + sizeof(K)+4 (write).
//
// Memory-level parallelism (round 2, found in the ISA): a load written as `if (i < n) x = p[i]` inside an unrolled loop compiles to
// a branch around the load and an `s_waitcnt vmcnt(0)` before its first use, i.e. ONE load in flight per wave and one exposed HBM
// round trip per round -- 16 (+16 for the values) per tile in the first version of k_rs_scatter, which is what held it at
// ~1 TB/s.  Every streaming kernel here therefore issues ALL loads of its tile first, branch-free (the index is clamped to n-1
// and the lane masked afterwards), and only then starts to consume them.
#include "prims.hpp"
#include "sort.hpp"

namespace bz3 {

// ---------------------------------------------------------------------------------------------
// exclusive scan
// ---------------------------------------------------------------------------------------------
constexpr int SC_BLOCK = 256;
constexpr int SC_ITEMS = 8;
constexpr int SC_TILE = SC_BLOCK * SC_ITEMS;

// The SC_ITEMS consecutive values of a thread: two 16-byte loads when the tile is whole and aligned, clamped scalar loads (all in
// flight together, see the note on memory-level parallelism above) otherwise.  Values past n read as 0.
__device__ __forceinline__ void scan_load_items(const u32 * __restrict__ data, u64 n, u64 base, u32 (&v)[SC_ITEMS]) {
    static_assert(SC_ITEMS == 8, "two uint4 per thread");
    if (base + SC_ITEMS <= n && (reinterpret_cast<uintptr_t>(data) & 15u) == 0) {
        const uint4 a = *reinterpret_cast<const uint4 *>(data + base);
        const uint4 b = *reinterpret_cast<const uint4 *>(data + base + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        const u64 last = n - 1;  // n >= 1: the scan is never launched on nothing
#pragma unroll
        for (int k = 0; k < SC_ITEMS; k++) v[k] = data[base + k < n ? base + k : last];
#pragma unroll
        for (int k = 0; k < SC_ITEMS; k++) v[k] = base + k < n ? v[k] : 0u;
    }
}

__global__ void __launch_bounds__(SC_BLOCK) k_scan_reduce(const u32 * __restrict__ data, u64 n, u32 * __restrict__ sums) {
    __shared__ u32 lds[SC_BLOCK / WAVE + 1];
    const u64 base = (u64)blockIdx.x * SC_TILE + (u64)threadIdx.x * SC_ITEMS;
    u32 v[SC_ITEMS];
    scan_load_items(data, n, base, v);
    u32 acc = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) acc += v[k];
    u32 tot = block_sum<SC_BLOCK>(acc, lds);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// Local exclusive scan of one tile; `sums` (may be null) holds the exclusive prefix of every tile.
__global__ void __launch_bounds__(SC_BLOCK) k_scan_apply(u32 * __restrict__ data, u64 n, const u32 * __restrict__ sums,
                                                         u32 * __restrict__ total_out) {
    __shared__ u32 lds[SC_BLOCK / WAVE + 1];
    const u64 base = (u64)blockIdx.x * SC_TILE + (u64)threadIdx.x * SC_ITEMS;
    u32 v[SC_ITEMS];
    scan_load_items(data, n, base, v);
    u32 acc = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) acc += v[k];
    u32 tot;
    u32 pre = block_excl_add<SC_BLOCK>(acc, lds, tot);
    if (sums) pre += sums[blockIdx.x];
    if (base + SC_ITEMS <= n && (reinterpret_cast<uintptr_t>(data) & 15u) == 0) {
        uint4 a, b;
        a.x = pre; a.y = a.x + v[0]; a.z = a.y + v[1]; a.w = a.z + v[2];
        b.x = a.w + v[3]; b.y = b.x + v[4]; b.z = b.y + v[5]; b.w = b.z + v[6];
        *reinterpret_cast<uint4 *>(data + base) = a;
        *reinterpret_cast<uint4 *>(data + base + 4) = b;
    } else {
#pragma unroll
        for (int k = 0; k < SC_ITEMS; k++) {
            if (base + k < n) data[base + k] = pre;
            pre += v[k];
        }
    }
    if (total_out && gridDim.x == 1 && threadIdx.x == 0) *total_out = tot;
}

__global__ void k_add_last(const u32 * __restrict__ sums_excl, const u32 * __restrict__ last_tile_total, u32 * out) {
    // total = exclusive prefix of the last tile + that tile's own sum
    *out = *sums_excl + *last_tile_total;
}

// Mid-sized tables in ONE launch (round 5): a workgroup of 1024, every thread a contiguous chunk of the table (16-byte accesses; a thread's lines are
// reused from L1 / L2 across its iterations), one workgroup scan over the chunk sums.  Up to 256 Ki words -- the count table of a radix pass of up to
// 1024 tiles (4 Mi keys) -- this replaces reduce + apply(1 tile) + apply: three launches of ~10 us with their gaps, of which the sorter's ~40 mid-sized
// passes per block paid ~1.5 ms (profiles/r05_call6_kernels_*: 123 k_scan_apply + 68 k_scan_reduce launches per 256 MiB block).
constexpr int SC1_BLOCK = 1024;
constexpr u64 SC1_MAX = 262144;
__global__ void __launch_bounds__(SC1_BLOCK) k_scan_single(u32 * __restrict__ data, u64 n, u32 * __restrict__ total_out) {
    __shared__ u32 lds[SC1_BLOCK / WAVE + 1];
    const u64 quads = (n + 3) / 4;                                  // the table in groups of four words
    const u64 per = (quads + SC1_BLOCK - 1) / SC1_BLOCK;            // groups per thread
    const u64 q0 = (u64)threadIdx.x * per, q1 = q0 + per < quads ? q0 + per : quads;
    const bool vec = (reinterpret_cast<uintptr_t>(data) & 15u) == 0;
    auto load4 = [&](u64 q, u32 (&v)[4]) {
        if (vec && 4 * q + 4 <= n) {
            const uint4 a = *reinterpret_cast<const uint4 *>(data + 4 * q);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = 4 * q + k < n ? data[4 * q + k] : 0u;
        }
    };
    u32 acc = 0;
    for (u64 q = q0; q < q1; q++) {
        u32 v[4];
        load4(q, v);
        acc += v[0] + v[1] + v[2] + v[3];
    }
    u32 tot;
    u32 run = block_excl_add<SC1_BLOCK>(acc, lds, tot);
    for (u64 q = q0; q < q1; q++) {
        u32 v[4];
        load4(q, v);
        u32 o[4];
        o[0] = run; o[1] = o[0] + v[0]; o[2] = o[1] + v[1]; o[3] = o[2] + v[2];
        run = o[3] + v[3];
        if (vec && 4 * q + 4 <= n) {
            *reinterpret_cast<uint4 *>(data + 4 * q) = make_uint4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (4 * q + k < n) data[4 * q + k] = o[k];
        }
    }
    if (total_out && threadIdx.x == 0) *total_out = tot;
}

static void scan_rec(u32 * d, u64 n, u32 * d_total, Arena & tmp, hipStream_t s) {
    if (n == 0) {
        if (d_total) HIP_CHECK(hipMemsetAsync(d_total, 0, 4, s));
        return;
    }
    if (n <= SC_TILE) {
        launch(k_scan_apply, dim3(1), dim3(SC_BLOCK), 0, s, d, n, (const u32 *)nullptr, d_total);
        return;
    }
    static const bool no_single = getenv("BZ3_SCAN_NO_SINGLE") != nullptr;  // (experiments: the recursive scan for every size, as up to round 4; read once)
    if (n <= SC1_MAX && !no_single) {
        launch(k_scan_single, dim3(1), dim3(SC1_BLOCK), 0, s, d, n, d_total);
        return;
    }
    const u64 tiles = (n + SC_TILE - 1) / SC_TILE;
    size_t m = tmp.mark();
    u32 * sums = tmp.take<u32>(tiles + 1);
    launch(k_scan_reduce, dim3((u32)tiles), dim3(SC_BLOCK), 0, s, (const u32 *)d, n, sums);
    scan_rec(sums, tiles, d_total, tmp, s);  // d_total = sum of all tile sums = grand total
    launch(k_scan_apply, dim3((u32)tiles), dim3(SC_BLOCK), 0, s, d, n, (const u32 *)sums, (u32 *)nullptr);
    tmp.release(m);
}

void exclusive_scan_u32(u32 * d_data, u64 n, u32 * d_total, Arena & tmp, hipStream_t s) { scan_rec(d_data, n, d_total, tmp, s); }

// ---------------------------------------------------------------------------------------------
// radix pass
// ---------------------------------------------------------------------------------------------
constexpr int RS_WAVES = RS_BLOCK / WAVE;          // 4
constexpr int RS_ROUNDS = RS_TILE / RS_BLOCK;      // 16 rounds of 64 keys per wave
constexpr int RS_WAVE_SPAN = RS_TILE / RS_WAVES;   // 1024 keys per wave

// XCD-aware tile assignment.  The dispatcher places workgroup b on XCD b % 8 (observed, MI355X_MICROARCH.md "Workgroup dispatch"; a
// speed assumption only: every tile is processed exactly once whatever the placement), and every XCD has its own L2.  A radix
// scatter writes ~16 keys of a tile to each of 256 bucket regions; the tiles that fill the neighbouring bytes of a cache line are
// the NEXT tiles.  With tile = blockIdx they sit on the other seven XCDs and every L2 writes its partial lines back on its own
// (measured in round 1: 4.6x write amplification); with a contiguous range of tiles per XCD they merge in one L2.
// The grid is 8 * ceil(tiles / 8) workgroups; the padded ones (tile >= tiles) leave at once.
__device__ __forceinline__ u32 rs_tile_of_block(u32 b, u32 tiles, bool xcd) {
    if (!xcd) return b;
    const u32 per = (tiles + 7u) / 8u;
    return (b & 7u) * per + (b >> 3);
}
inline u32 rs_grid(u32 tiles, bool xcd) { return xcd ? 8u * ((tiles + 7u) / 8u) : tiles; }

template <typename K, int BITS>
__device__ __forceinline__ u32 rs_digit(K key, int shift) {
    return (u32)(key >> shift) & ((1u << BITS) - 1u);
}

// BITS = digit width: 8 (256 bins, the default) or 9 (512 bins: the LZP predecessor build sorts its 18-bit hashes in two passes
// instead of three, round 5).  A thread owns RADIX / RS_BLOCK neighbouring digits wherever a digit needs a thread.
template <typename K, int BITS>
__global__ void __launch_bounds__(RS_BLOCK) k_rs_hist(const K * __restrict__ keys, u64 n, int shift, u32 * __restrict__ hist, u32 tiles, u32 xcd) {
    constexpr int RADIX = 1 << BITS;
    __shared__ u32 bins[RADIX];
    // same tile assignment as the scatter: the counts of neighbouring tiles are neighbouring words of the digit-major table, so
    // the tiles of one XCD fill whole lines of it in that XCD's L2 (and the scatter finds its tile's keys in the L2 that read them)
    const u32 tile = rs_tile_of_block(blockIdx.x, tiles, xcd != 0u);
    if (tile >= tiles) return;
#pragma unroll
    for (int d = threadIdx.x; d < RADIX; d += RS_BLOCK) bins[d] = 0;
    __syncthreads();
    const u64 tile_base = (u64)tile * RS_TILE;
    const u64 wbase = tile_base + (u64)wave_id() * RS_WAVE_SPAN + lane_id();
    const u64 last = n - 1;
    K key[RS_ROUNDS];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {  // all 16 loads in flight before the first LDS atomic
        const u64 i = wbase + (u64)r * WAVE;
        key[r] = keys[i < n ? i : last];
    }
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const u64 i = wbase + (u64)r * WAVE;
        if (i < n) atomicAdd(&bins[rs_digit<K, BITS>(key[r], shift)], 1u);
    }
    __syncthreads();
#pragma unroll
    for (int d = threadIdx.x; d < RADIX; d += RS_BLOCK) hist[(u64)d * tiles + tile] = bins[d];
}

// The tile is first put in digit order in LDS, then written out with consecutive lanes on consecutive destinations -- a digit's ~16
// keys of a tile leave as one run instead of 16 separate 8-byte stores.  Measured in round 3 on a 256 MiB block (full-n pass of
// 8-byte keys + 4-byte values): 1.46 ms = 4.4 TB/s of algorithmic traffic against 2.37 ms for the direct scatter
// (profiles/r03_kernel_stats_stages_256MiB_staged.txt), so the direct scatter is gone.
// Slot of a key inside the tile = start of its digit + keys of that digit in earlier waves + rank in its own wave, which is the
// stable order.  LDS: sizeof(K) * 4096 + 16 KiB + 6 KiB (54 KiB for 8-byte keys: two workgroups per CU).
//
// RAW (round 5): `offs` is the hist kernel's table as it left it -- per-tile COUNTS, not scanned -- and every workgroup derives its own
// 2^BITS offsets from it: digit d of tile t starts at (keys of smaller digits in all tiles) + (keys of digit d in earlier tiles), two
// sums over one row of the digit-major table per digit.  For a pass of up to RS_RAW_TILES tiles that is cheaper than the three to
// five launches of the recursive scan it replaces (415 k scan launches per bench run, most of them for the sorter's
// ~50 small passes per block); larger passes keep the scanned table.
template <typename K, bool IOTA, bool WKEYS, int BITS, bool RAW>
__global__ void __launch_bounds__(RS_BLOCK) k_rs_scatter(const K * __restrict__ kin, K * __restrict__ kout, const u32 * __restrict__ vin,
                                                               u32 * __restrict__ vout, u64 n, int shift, const u32 * __restrict__ offs, u32 tiles,
                                                               u32 iota_split, u32 out_base, u32 xcd) {
    constexpr int RADIX = 1 << BITS;
    constexpr int DPT = RADIX / RS_BLOCK;  // digits per thread: d = DPT * threadIdx.x + j
    static_assert(RADIX % RS_BLOCK == 0 && DPT >= 1, "a thread owns whole digits");
    __shared__ u32 cnt[RS_WAVES][RADIX];
    __shared__ u32 dstart[RADIX];   // first slot of a digit inside the tile
    __shared__ u32 gdelta[RADIX];   // global destination of a digit's first key - dstart (mod 2^32)
    __shared__ u32 scan_lds[RS_BLOCK / WAVE + 1];
    __shared__ K skey[RS_TILE];
    __shared__ u32 sval[RS_TILE];
    const u32 tile = rs_tile_of_block(blockIdx.x, tiles, xcd != 0u);
    if (tile >= tiles) return;
    const int w = wave_id(), l = lane_id();
#pragma unroll
    for (int k = 0; k < RS_WAVES; k++)
#pragma unroll
        for (int d = threadIdx.x; d < RADIX; d += RS_BLOCK) cnt[k][d] = 0;
    __syncthreads();

    const u64 tile_base = (u64)tile * RS_TILE;
    const u64 wbase = tile_base + (u64)w * RS_WAVE_SPAN + l;
    K key[RS_ROUNDS];
    u32 local[RS_ROUNDS];
    const u64 lt = lanemask_lt();
    const u64 last = n - 1;
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const u64 i = wbase + (u64)r * WAVE;
        key[r] = kin[i < n ? i : last];
    }
    // RAW: this thread's digits' rows of the count table, requested before the ranking below so that their latency hides behind it
    u32 row_total[DPT], row_before[DPT];
    if (RAW) {
#pragma unroll
        for (int j = 0; j < DPT; j++) {
            const u32 * __restrict__ row = offs + (u64)(DPT * threadIdx.x + j) * tiles;
            u32 tot = 0, bef = 0;
#pragma unroll 4
            for (u32 t = 0; t < tiles; t++) {
                const u32 c = row[t];
                tot += c;
                bef += t < tile ? c : 0u;
            }
            row_total[j] = tot;
            row_before[j] = bef;
        }
    }
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const u64 i = wbase + (u64)r * WAVE;
        const bool valid = i < n;
        const u32 d = rs_digit<K, BITS>(key[r], shift);
        u64 peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < BITS; b++) {
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
    u32 val[RS_ROUNDS];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const u64 i = wbase + (u64)r * WAVE;
        val[r] = IOTA ? (u32)i + ((u32)i >= iota_split ? 1u : 0u) : vin[i < n ? i : last];
    }
    __syncthreads();
    {
        u32 run[DPT];
        u32 mine = 0;
#pragma unroll
        for (int j = 0; j < DPT; j++) {
            const u32 d = DPT * threadIdx.x + j;
            u32 acc = 0;
#pragma unroll
            for (int k = 0; k < RS_WAVES; k++) {
                u32 c = cnt[k][d];
                cnt[k][d] = acc;
                acc += c;
            }
            run[j] = acc;
            mine += acc;
        }
        u32 total;
        u32 start = block_excl_add<RS_BLOCK>(mine, scan_lds, total);  // ends with a barrier
        u32 gbase = 0;
        if (RAW) {
            u32 rt = 0;
#pragma unroll
            for (int j = 0; j < DPT; j++) rt += row_total[j];
            u32 all;
            gbase = block_excl_add<RS_BLOCK>(rt, scan_lds, all);  // keys of smaller digits, all tiles
        }
#pragma unroll
        for (int j = 0; j < DPT; j++) {
            const u32 d = DPT * threadIdx.x + j;
            dstart[d] = start;
            if (RAW) {
                gdelta[d] = gbase + row_before[j] + out_base - start;
                gbase += row_total[j];
            } else {
                gdelta[d] = offs[(u64)d * tiles + tile] + out_base - start;
            }
            start += run[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const u32 d = rs_digit<K, BITS>(key[r], shift);
        local[r] += dstart[d] + cnt[w][d];
    }
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const u64 i = wbase + (u64)r * WAVE;
        if (i < n) {
            skey[local[r]] = key[r];
            sval[local[r]] = val[r];
        }
    }
    __syncthreads();
    const u64 left = n - tile_base;
    const u32 count = left < (u64)RS_TILE ? (u32)left : (u32)RS_TILE;
#pragma unroll
    for (int q = 0; q < RS_ROUNDS; q++) {
        const u32 j = (u32)q * RS_BLOCK + threadIdx.x;
        if (j < count) {
            const K k = skey[j];
            const u32 pos = gdelta[rs_digit<K, BITS>(k, shift)] + j;
            if (WKEYS) kout[pos] = k;
            vout[pos] = sval[j];
        }
    }
}

// Passes of up to this many tiles (512 Ki keys) skip the scan: the scatter's workgroups read the count table themselves (RAW above).
// Every workgroup reads the whole table -- tiles * 2^BITS words from L2 -- so the bound keeps that at 128 KiB (8-bit digits) per workgroup.
constexpr u32 RS_RAW_TILES = 128;

template <typename K, int BITS>
void radix_pass_bits(const K * kin, K * kout, const u32 * vin, u32 * vout, u64 n, int shift, u32 iota_split, u32 out_base, Arena & tmp,
                     hipStream_t s) {
    if (n == 0) return;
    constexpr int RADIX = 1 << BITS;
    const u32 tiles = (u32)((n + RS_TILE - 1) / RS_TILE);
    size_t m = tmp.mark();
    u32 * hist = tmp.take<u32>((size_t)tiles * RADIX);
    const u32 xcd = 1u;  // contiguous tiles per XCD (measured round 3: 2.37 against 3.58 ms per full-n pass with tile = blockIdx)
    const dim3 sgrid(rs_grid(tiles, xcd != 0u));
    launch(k_rs_hist<K, BITS>, sgrid, dim3(RS_BLOCK), 0, s, kin, n, shift, hist, tiles, xcd);
    static const bool no_raw = getenv("BZ3_RS_NO_RAW") != nullptr;  // (experiments: the scanned table for every pass, as up to round 4; read once)
    const bool raw = tiles <= RS_RAW_TILES && !no_raw;
    if (!raw) exclusive_scan_u32(hist, (u64)tiles * RADIX, nullptr, tmp, s);
    const bool iota = vin == nullptr, wkeys = kout != nullptr;
#define BZ3_RS_LAUNCH(I, W, R) \
    launch(k_rs_scatter<K, I, W, BITS, R>, sgrid, dim3(RS_BLOCK), 0, s, kin, kout, vin, vout, n, shift, (const u32 *)hist, tiles, iota_split, out_base, xcd)
#define BZ3_RS_DISPATCH(R)                                   \
    do {                                                     \
        if (iota && wkeys) BZ3_RS_LAUNCH(true, true, R);     \
        else if (iota) BZ3_RS_LAUNCH(true, false, R);        \
        else if (wkeys) BZ3_RS_LAUNCH(false, true, R);       \
        else BZ3_RS_LAUNCH(false, false, R);                 \
    } while (0)
    if (raw) BZ3_RS_DISPATCH(true);
    else BZ3_RS_DISPATCH(false);
#undef BZ3_RS_DISPATCH
#undef BZ3_RS_LAUNCH
    tmp.release(m);
}

void radix_hist_bits9(const u32 * keys, u64 n, int shift, u32 * hist, u32 tiles, hipStream_t s) {
    launch(k_rs_hist<u32, 9>, dim3(rs_grid(tiles, true)), dim3(RS_BLOCK), 0, s, keys, n, shift, hist, tiles, 1u);
}

template <typename K>
void radix_pass(const K * kin, K * kout, const u32 * vin, u32 * vout, u64 n, int shift, u32 iota_split, u32 out_base, Arena & tmp,
                hipStream_t s) {
    radix_pass_bits<K, 8>(kin, kout, vin, vout, n, shift, iota_split, out_base, tmp, s);
}

template <typename K>
int radix_sort_pairs(K * k0, K * k1, u32 * v0, u32 * v1, u64 n, int bit_lo, int bit_hi, Arena & tmp, hipStream_t s) {
    int cur = 0;
    for (int shift = bit_lo; shift < bit_hi; shift += 8) {
        if (cur == 0)
            radix_pass<K>(k0, k1, v0, v1, n, shift, 0xFFFFFFFFu, 0u, tmp, s);
        else
            radix_pass<K>(k1, k0, v1, v0, n, shift, 0xFFFFFFFFu, 0u, tmp, s);
        cur ^= 1;
    }
    return cur;
}

template void radix_pass<u8>(const u8 *, u8 *, const u32 *, u32 *, u64, int, u32, u32, Arena &, hipStream_t);
template void radix_pass<u32>(const u32 *, u32 *, const u32 *, u32 *, u64, int, u32, u32, Arena &, hipStream_t);
template void radix_pass<u64>(const u64 *, u64 *, const u32 *, u32 *, u64, int, u32, u32, Arena &, hipStream_t);
template void radix_pass_bits<u32, 9>(const u32 *, u32 *, const u32 *, u32 *, u64, int, u32, u32, Arena &, hipStream_t);
template int radix_sort_pairs<u32>(u32 *, u32 *, u32 *, u32 *, u64, int, int, Arena &, hipStream_t);
template int radix_sort_pairs<u64>(u64 *, u64 *, u32 *, u32 *, u64, int, int, Arena &, hipStream_t);

}  // namespace bz3
