// sort.hip -- device-wide exclusive scan + stable 8-bit-digit LSD radix sort for gfx950.
//
// Radix pass = three kernels (all HBM-streaming, no MFMA: byte/integer work):
//   k_rs_hist    : per-tile digit histogram, LDS-binned, written digit-major  hist[d * tiles + t]
//   scan         : device-wide exclusive scan of the digit-major table => global offset of every
//                  (digit, tile) bucket
//   k_rs_scatter : re-reads the tile, ranks each key among equal digits with wave64 ballots
//                  (match-any over the 8 digit bits + popcount below the lane), adds the per-wave
//                  running counts kept in LDS, and scatters key+value to its final slot.  The
//                  tile is walked wave-striped (wave w owns keys [1024w, 1024w+1024), 16 rounds of
//                  64 consecutive keys) so the order of equal digits is preserved: the pass is stable.
// Algorithmic traffic per pass and element: sizeof(K) (hist) + sizeof(K)+4 (read) + sizeof(K)+4 (write).
#include "prims.hpp"
#include "sort.hpp"

namespace bz3 {

// ---------------------------------------------------------------------------------------------
// exclusive scan
// ---------------------------------------------------------------------------------------------
constexpr int SC_BLOCK = 256;
constexpr int SC_ITEMS = 8;
constexpr int SC_TILE = SC_BLOCK * SC_ITEMS;

__global__ void __launch_bounds__(SC_BLOCK) k_scan_reduce(const u32 * __restrict__ data, u64 n, u32 * __restrict__ sums) {
    __shared__ u32 lds[SC_BLOCK / WAVE + 1];
    const u64 base = (u64)blockIdx.x * SC_TILE + (u64)threadIdx.x * SC_ITEMS;
    u32 acc = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++)
        if (base + k < n) acc += data[base + k];
    u32 tot = block_sum<SC_BLOCK>(acc, lds);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// Local exclusive scan of one tile; `sums` (may be null) holds the exclusive prefix of every tile.
__global__ void __launch_bounds__(SC_BLOCK) k_scan_apply(u32 * __restrict__ data, u64 n, const u32 * __restrict__ sums,
                                                         u32 * __restrict__ total_out) {
    __shared__ u32 lds[SC_BLOCK / WAVE + 1];
    const u64 base = (u64)blockIdx.x * SC_TILE + (u64)threadIdx.x * SC_ITEMS;
    u32 v[SC_ITEMS];
    u32 acc = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) {
        v[k] = (base + k < n) ? data[base + k] : 0u;
        acc += v[k];
    }
    u32 tot;
    u32 pre = block_excl_add<SC_BLOCK>(acc, lds, tot);
    if (sums) pre += sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) {
        if (base + k < n) data[base + k] = pre;
        pre += v[k];
    }
    if (total_out && gridDim.x == 1 && threadIdx.x == 0) *total_out = tot;
}

__global__ void k_add_last(const u32 * __restrict__ sums_excl, const u32 * __restrict__ last_tile_total, u32 * out) {
    // total = exclusive prefix of the last tile + that tile's own sum
    *out = *sums_excl + *last_tile_total;
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

template <typename K>
__device__ __forceinline__ u32 rs_digit(K key, int shift) {
    return (u32)(key >> shift) & 0xFFu;
}

template <typename K>
__global__ void __launch_bounds__(RS_BLOCK) k_rs_hist(const K * __restrict__ keys, u64 n, int shift, u32 * __restrict__ hist, u32 tiles) {
    __shared__ u32 bins[RS_RADIX];
    bins[threadIdx.x] = 0;
    __syncthreads();
    const u64 tile_base = (u64)blockIdx.x * RS_TILE;
    const u64 wbase = tile_base + (u64)wave_id() * RS_WAVE_SPAN + lane_id();
#pragma unroll 4
    for (int r = 0; r < RS_ROUNDS; r++) {
        const u64 i = wbase + (u64)r * WAVE;
        if (i < n) atomicAdd(&bins[rs_digit(keys[i], shift)], 1u);
    }
    __syncthreads();
    hist[(u64)threadIdx.x * tiles + blockIdx.x] = bins[threadIdx.x];
}

template <typename K, bool IOTA, bool WKEYS>
__global__ void __launch_bounds__(RS_BLOCK) k_rs_scatter(const K * __restrict__ kin, K * __restrict__ kout, const u32 * __restrict__ vin,
                                                        u32 * __restrict__ vout, u64 n, int shift, const u32 * __restrict__ offs, u32 tiles,
                                                        u32 iota_split, u32 out_base, u32 xcd) {
    __shared__ u32 cnt[RS_WAVES][RS_RADIX];
    __shared__ u32 gbase[RS_RADIX];
    const u32 tile = rs_tile_of_block(blockIdx.x, tiles, xcd != 0u);
    if (tile >= tiles) return;
    const int w = wave_id(), l = lane_id();
#pragma unroll
    for (int k = 0; k < RS_WAVES; k++) cnt[k][threadIdx.x] = 0;
    __syncthreads();

    const u64 tile_base = (u64)tile * RS_TILE;
    const u64 wbase = tile_base + (u64)w * RS_WAVE_SPAN + l;
    K key[RS_ROUNDS];
    u32 local[RS_ROUNDS];
    const u64 lt = lanemask_lt();

#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const u64 i = wbase + (u64)r * WAVE;
        const bool valid = i < n;
        key[r] = valid ? kin[i] : (K)0;
        const u32 d = rs_digit(key[r], shift);
        u64 peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const u64 bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const u32 below = (u32)__popcll(peers & lt);
        const u32 pre = valid ? cnt[w][d] : 0u;
        wave_sync();  // every lane has read the running count before a leader bumps it
        if (valid && below == 0) cnt[w][d] = pre + (u32)__popcll(peers);
        wave_sync();
        local[r] = pre + below;
    }
    __syncthreads();
    {
        const u32 d = threadIdx.x;
        u32 run = 0;
#pragma unroll
        for (int k = 0; k < RS_WAVES; k++) {
            u32 c = cnt[k][d];
            cnt[k][d] = run;
            run += c;
        }
        gbase[d] = offs[(u64)d * tiles + tile] + out_base;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const u64 i = wbase + (u64)r * WAVE;
        if (i < n) {
            const u32 d = rs_digit(key[r], shift);
            const u64 pos = (u64)gbase[d] + cnt[w][d] + local[r];
            if (WKEYS) kout[pos] = key[r];
            vout[pos] = IOTA ? (u32)i + ((u32)i >= iota_split ? 1u : 0u) : vin[i];
        }
    }
}

template <typename K>
void radix_pass(const K * kin, K * kout, const u32 * vin, u32 * vout, u64 n, int shift, u32 iota_split, u32 out_base, Arena & tmp,
                hipStream_t s) {
    if (n == 0) return;
    const u32 tiles = (u32)((n + RS_TILE - 1) / RS_TILE);
    size_t m = tmp.mark();
    u32 * hist = tmp.take<u32>((size_t)tiles * RS_RADIX);
    launch(k_rs_hist<K>, dim3(tiles), dim3(RS_BLOCK), 0, s, kin, n, shift, hist, tiles);
    exclusive_scan_u32(hist, (u64)tiles * RS_RADIX, nullptr, tmp, s);
    const bool iota = vin == nullptr, wkeys = kout != nullptr;
    static const u32 xcd = getenv("BZ3_RS_NO_XCD") ? 0u : 1u;  // experiments: BZ3_RS_NO_XCD=1 = tile = blockIdx
    const dim3 sgrid(rs_grid(tiles, xcd != 0u));
    if (iota && wkeys)
        launch(k_rs_scatter<K, true, true>, sgrid, dim3(RS_BLOCK), 0, s, kin, kout, vin, vout, n, shift, (const u32 *)hist, tiles, iota_split, out_base, xcd);
    else if (iota)
        launch(k_rs_scatter<K, true, false>, sgrid, dim3(RS_BLOCK), 0, s, kin, kout, vin, vout, n, shift, (const u32 *)hist, tiles, iota_split, out_base, xcd);
    else if (wkeys)
        launch(k_rs_scatter<K, false, true>, sgrid, dim3(RS_BLOCK), 0, s, kin, kout, vin, vout, n, shift, (const u32 *)hist, tiles, iota_split, out_base, xcd);
    else
        launch(k_rs_scatter<K, false, false>, sgrid, dim3(RS_BLOCK), 0, s, kin, kout, vin, vout, n, shift, (const u32 *)hist, tiles, iota_split, out_base, xcd);
    tmp.release(m);
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
template int radix_sort_pairs<u32>(u32 *, u32 *, u32 *, u32 *, u64, int, int, Arena &, hipStream_t);
template int radix_sort_pairs<u64>(u64 *, u64 *, u32 *, u32 *, u64, int, int, Arena &, hipStream_t);

}  // namespace bz3
