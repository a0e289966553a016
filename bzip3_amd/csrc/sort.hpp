// sort.hpp -- device-wide primitives: exclusive scan and stable LSD radix sort passes.
// Hand-written for gfx950: wave64 ballot ranking, LDS-binned per-tile digit histograms,
// digit-major global histogram + device-wide scan, stable scatter.  Used by the suffix sort
// (bwt.hip), the inverse-BWT LF/psi build (unbwt.hip) and the LZP predecessor build (lzp.hip).
#pragma once
#include "hipx.hpp"

namespace bz3 {

// Bump allocator over a device scratch buffer (the per-device workspace owned by api.cpp).
struct Arena {
    char * base = nullptr;
    size_t cap = 0, used = 0;
    template <typename T>
    T * take(size_t count) {
        size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        if (used + bytes > cap) throw HipError{hipErrorUnknown, "device workspace exhausted", __FILE__, __LINE__};
        T * p = reinterpret_cast<T *>(base + used);
        used += bytes;
        return p;
    }
    size_t mark() const { return used; }
    void release(size_t m) { used = m; }
};

constexpr int RS_TILE = 4096;   // keys per workgroup tile (4 waves x 16 rounds x 64 lanes)
constexpr int RS_BLOCK = 256;
constexpr int RS_RADIX = 256;   // 8-bit digits

inline size_t scan_temp_words(u64 n) {  // words of scratch the recursive scan needs
    size_t w = 0;
    while (n > 2048) {
        n = (n + 2047) / 2048;
        w += ((n + 63) & ~(u64)63);
    }
    return w + 64;
}
inline size_t radix_temp_bytes(u64 n, int bits = 8) {  // scratch of one pass with 2^bits bins
    u64 tiles = (n + RS_TILE - 1) / RS_TILE;
    u64 hist = tiles << bits;
    return (hist + scan_temp_words(hist) + 256) * 4 + 4096;
}

// In-place exclusive prefix sum of n u32 values; if d_total != nullptr the grand total is written there.
void exclusive_scan_u32(u32 * d_data, u64 n, u32 * d_total, Arena & tmp, hipStream_t s);

// One stable radix pass on digit (key >> shift) & 0xFF.
//   K       : u8 / u32 / u64 keys
//   vin     : values (nullptr => value of element i is  i + (i >= iota_split ? 1 : 0))
//   kout    : may be nullptr (keys not needed downstream)
//   out_base: added to every destination index (the inverse BWT uses 1: row 0 is the sentinel)
template <typename K>
void radix_pass(const K * kin, K * kout, const u32 * vin, u32 * vout, u64 n, int shift, u32 iota_split, u32 out_base,
                Arena & tmp, hipStream_t s);

// The same with digits of BITS bits (8 or 9): digit = (key >> shift) & (2^BITS - 1).
template <typename K, int BITS>
void radix_pass_bits(const K * kin, K * kout, const u32 * vin, u32 * vout, u64 n, int shift, u32 iota_split, u32 out_base,
                     Arena & tmp, hipStream_t s);

// The per-tile digit-major count table of a 9-bit pass alone (lzp.hip bins its link records with a scatter kernel of its own).
void radix_hist_bits9(const u32 * keys, u64 n, int shift, u32 * hist, u32 tiles, hipStream_t s);

// Full LSD sort over key bits [bit_lo, bit_hi) in 8-bit digits, ping-ponging (k0,v0) <-> (k1,v1).
// Returns 0 if the sorted data ends in (k0,v0), 1 if in (k1,v1).
template <typename K>
int radix_sort_pairs(K * k0, K * k1, u32 * v0, u32 * v1, u64 n, int bit_lo, int bit_hi, Arena & tmp, hipStream_t s);

}  // namespace bz3
