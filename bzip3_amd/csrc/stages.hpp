// stages.hpp -- host-side entry points of the per-stage HIP pipelines (all asynchronous on a stream
// unless stated).  One function per reference stage; api.cpp strings them together exactly in the
// order of bz3_encode_block / bz3_decode_block (reference src/libbz3.c:585-654, :656-809).
#pragma once
#include "hipx.hpp"
#include "sort.hpp"

namespace bz3 {

// ---- CRC-32C (crc32c.hip) -- replaces crc32sum, src/libbz3.c:69-72 ---------------------------
struct CrcTables {
    u32 pow2[64];  // x^(2^b) mod P
    u32 lane[64];  // x^(32*(63-l)) mod P
    u32 row;       // x^2048 mod P
    u32 pad[3];
};
void crc_build_tables(CrcTables & t);  // host
// Result is left in d_scratch[1] (d_scratch: 2 words).
void crc32c_device(const u8 * d_data, u64 n, u32 init, const CrcTables * d_tables, u32 * d_scratch, hipStream_t s);

// ---- mRLE (mrle.hip) -- replaces mrlec / mrled, src/libbz3.c:264-329 -------------------------
struct MrleEncScratch {
    u32 tiles = 0;
    u32 * carry = nullptr;
    u32 * tile_sum = nullptr;
    s32 * gain = nullptr;
    u32 * total = nullptr;  // device word: encoded size - 32
};
// Pass 1: decides the flagged symbols and the encoded size (sc.total, device).  Scratch is carved from
// `tmp` and stays valid until the caller releases it.
void mrle_encode_size(const u8 * d_in, u32 n, MrleEncScratch & sc, Arena & tmp, hipStream_t s);
// Pass 2: writes the 32-byte bitmap + encoded bytes.
void mrle_encode_write(const u8 * d_in, u32 n, const MrleEncScratch & sc, u8 * d_out, hipStream_t s);
// Decodes m stream bytes into at most outlen bytes; *d_total = bytes produced (capped at outlen).
void mrle_decode(const u8 * d_enc, u32 m, u8 * d_out, u32 outlen, u32 * d_total, Arena & tmp, hipStream_t s);

// ---- LZP (lzp.hip) -- replaces lzp_compress / lzp_decompress, src/libbz3.c:124-257 -----------
// Encode runs in three phases so that the serial "driver" kernels of many blocks can run side by side
// (one workgroup per block): prepare (grid-wide, async) -> driver batch (async) -> finish (grid-wide, sync).
struct LzDriverOut {
    u32 end_pos;     // first position handled by the tail loop (>= n - 72)
    u32 n_matches;
    u32 iterations;  // driver loop iterations          } profiling counters
    u32 evaluated;   // flagged positions it resolved   }
};
struct LzpEncodeCtx {
    bool active = false;  // false: n < 72, LZP declines (:244)
    const u8 * in = nullptr;
    u32 n = 0, nwords = 0;
    u32 *prev = nullptr, *next = nullptr, *skip = nullptr, *mstart = nullptr, *cand_bits = nullptr, *mpos = nullptr, *mlen = nullptr;
    LzDriverOut * d_res = nullptr;
};
struct LzpDriverJob {  // device addresses as integers: see prims.hpp global_ptr()
    u64 in;
    u64 prev, next, cand_bits, skip, mstart, mpos, mlen;
    u64 result;  // LzDriverOut *
    u32 n, nwords;
};
size_t lzp_encode_ctx_bytes(u64 n);
void lzp_encode_prepare(const u8 * d_in, u32 n, LzpEncodeCtx & c, Arena & tmp, hipStream_t s);
void lzp_encode_prepare(const u8 * d_in, u32 n, LzpEncodeCtx & c, Arena & ctx, Arena & tmp, hipStream_t s);  // context and transient scratch from different arenas
LzpDriverJob lzp_driver_job(const LzpEncodeCtx & c);
void lzp_driver_batch(const LzpDriverJob * h_jobs, LzpDriverJob * d_jobs, u32 njobs, hipStream_t s);
s32 lzp_encode_finish(const LzpEncodeCtx & c, u8 * d_out, Arena & tmp, hipStream_t s);  // encoded size or -1
s32 lzp_encode(const u8 * d_in, u32 n, u8 * d_out, Arena & tmp, hipStream_t s);         // one block, all three phases
// Decode: one workgroup per block, so a batch of blocks decodes concurrently (one launch, grid = jobs).
struct LzpDecodeJob {  // device addresses as integers: see prims.hpp global_ptr()
    u64 in;
    u64 out;
    u64 lut;     // 2^18 words, 16-byte aligned (the kernel zeroes them)
    u64 result;  // s32 *: decoded size or -1
    u32 n, max_out;
};
constexpr size_t LZP_LUT_WORDS = (size_t)1 << 18;
void lzp_decode_batch(const LzpDecodeJob * h_jobs, LzpDecodeJob * d_jobs, u32 njobs, hipStream_t s);  // asynchronous

// ---- BWT (bwt.hip) -- replaces libsais_bwt, include/libsais.h:4095-4121 ----------------------
// Returns the primary index (>= 1), synchronously -- or, with d_idx, leaves it in that device word and returns 0 without waiting for
// the stream (one read-back per pass remains inside: which path the block's groups take is decided on the host).  Rounds/active
// statistics are reported for profiling.
struct BwtStats {
    int rounds = 0;
    u64 sorted_elements = 0;  // sum over rounds of elements that went through the radix sorter
    int radix_passes = 0;
};
// d_outside (optional device word): receives the number of bytes of the block outside its BWT_ROUTE_KEEP most frequent byte values.
constexpr int BWT_ROUTE_KEEP = 40;
s32 bwt_forward(const u8 * d_in, u32 n, u8 * d_out, Arena & tmp, hipStream_t s, BwtStats * stats, u32 * d_idx = nullptr, u32 * d_outside = nullptr);
size_t bwt_workspace_bytes(u64 n);
void bwt_set_big_rounds(int k);  // tests only: more windows for the big groups before the deep path (k < 0: the default, 1)

// ---- inverse BWT (unbwt.hip) -- replaces libsais_unbwt, include/libsais.h:5260-5262 ----------
// Synchronous.  idx must already be validated (0 < idx <= n).
void bwt_inverse(const u8 * d_in, u32 n, u32 idx, u8 * d_out, Arena & tmp, hipStream_t s);
size_t unbwt_workspace_bytes(u64 n);

// ---- CM coder (cm.hip) -- replaces begin/encode_bytes/decode_bytes, src/libbz3.c:333-494 -----
// One workgroup (= one CU: the 145.5 KiB model fills its LDS) per block; a batch is ONE launch with
// grid = number of blocks, so the blocks of bz3_encode_blocks / bz3_decode_blocks run side by side without
// depending on how HIP maps streams to hardware queues.  Asynchronous.  Job arrays live in device memory.
constexpr u32 CM_NO_GAP = 0xFFFFFFFFu;
struct CmEncodeJob {  // device addresses as integers: see prims.hpp global_ptr()
    u64 in;
    u64 out;       // receives the coded bytes
    u64 out_size;  // u32[2] *: [0] receives the coded byte count
    u32 n;
    u32 debug;     // 0 = normal; profiling only (output invalid): 1 = coder wave alone, 2 = model waves alone
    // row-cache variants only (CM_VARIANT_ROWS*):
    u64 spill = 0;   // u16[256][256] scratch of this block: the order-1 rows that are not resident in LDS
    u64 status = 0;  // u32 *, zeroed by the caller: set to 1 when the kernel gave the block up (code it again with CM_VARIANT_FULL)
    u32 miss_base = 0, miss_shift = 0;  // give up once row misses > miss_base + (position >> miss_shift)
    // in-place coding (cm.hip CmSink): `out` lies `gap` bytes below `in` in the same buffer; CM_NO_GAP = separate buffers.
    // out_size[1] receives the index of the first coded byte that went to `side` instead (0xFFFFFFFF: none);
    // out_size[0] = 0xFFFFFFFF when `side` overflowed.
    u32 gap = CM_NO_GAP, side_cap = 0;
    u64 side = 0;
    // u32[CM_CLAIM_WORDS] *, zeroed before the launch (0: none): one word per CU in which the workgroups of that CU claim the SIMD their coder wave sits on
    // (cm.hip, "which of the two waves codes")
    u64 claim = 0;
};
constexpr u32 CM_CLAIM_WORDS = 4096;  // XCC (4 bits) | SE (3) | SH (1) | CU (4)
struct CmDecodeJob {
    u64 in;        // coded bytes; reads past in_size yield 0xFF.. like read_in (:345)
    u64 out;
    u32 in_size;
    u32 n;
    u32 debug;     // 0 = normal; 3 = the cycle-counter build's job (profiling only: counters replace the first output bytes)
    u32 pad;
    u64 spill = 0, status = 0;  // as above
    u32 miss_base = 0, miss_shift = 0;
};
// Kernel variants: the whole 145.5 KiB model in LDS (one workgroup per CU: k_cm_encode / k_cm_decode_sync), or the row-cache
// kernels (order-1 rows cached in LDS; they may give a block up, see status): ROWS = 96 rows, two workgroups per CU
// (k_cm_encode_rows / k_cm_decode_sync2), ROWS3 = 44 / 56 rows, three per CU (k_cm_encode_rows3 / k_cm_decode_sync3).
enum { CM_VARIANT_FULL = 0, CM_VARIANT_ROWS = 1, CM_VARIANT_ROWS3 = 2, CM_VARIANT_ROWS_TEST = 9 /* emulator builds only: tiny cache */ };
inline bool cm_variant_has_rows(int v) { return v != CM_VARIANT_FULL; }
inline bool cm_variant_is_test(int v) { return v == CM_VARIANT_ROWS_TEST; }
constexpr size_t CM_SPILL_BYTES = 256 * 256 * 2;
// When the row-cache kernels give a block up: row misses > CM_MISS_BASE + (position >> CM_MISS_SHIFT), i.e. beyond ~3 % of the bytes.
// A miss moves two 512-byte rows between LDS and HBM, ~2 us during which the block's coder waits: 3 % of misses cost ~60 ns per byte
// (+10 %), whereas a block that is given up is coded again by the whole-model kernel, one block per CU.  Rounds 1-3 gave up beyond
// 0.4 %, which text with digits and markup reaches (the enwik8-calibrated generator: 0.33 % at 44 rows, tools/cm_row_cache_sim.py) --
// round 4 found every block of such a batch coded twice.  Binary data misses on most bytes and is still given up within its first KiB.
constexpr u32 CM_MISS_BASE = 1024, CM_MISS_SHIFT = 5;
void cm_encode_batch(const CmEncodeJob * d_jobs, u32 njobs, hipStream_t s, int variant = CM_VARIANT_FULL);
void cm_decode_batch(const CmDecodeJob * d_jobs, u32 njobs, hipStream_t s, int variant = CM_VARIANT_FULL, bool prof = false);  // prof: the sync decoders' cycle-counter build

}  // namespace bz3
