/*
 * bz3_hip.h -- extensions of the libbz3.h C ABI that only make sense for a GPU backend.
 * Plain C, plain pointers and sizes; no torch / HIP types in any signature (device memory is passed
 * as void* addresses, e.g. torch.Tensor.data_ptr()).
 *
 *  - device-resident block entry points: same contract as bz3_encode_block(s)/bz3_decode_block(s)
 *    (reference src/libbz3.c:585-654, :656-809, :845-870) but `buffer` is a DEVICE pointer, so a
 *    pipeline that already has its data in HBM (or bench.py) skips the PCIe hops;
 *  - device selection for the multi-GPU block sharding of SURVEY.md section 8e;
 *  - per-stage entry points on host buffers, used by the parity tests to diff each HIP stage
 *    against its reference stage (crc32sum, mrlec, mrled, lzp_compress, lzp_decompress,
 *    libsais_bwt, libsais_unbwt, encode_bytes, decode_bytes);
 *  - per-stage timings of the last block a state processed.
 */
#ifndef BZ3_HIP_H
#define BZ3_HIP_H
#include <stddef.h>
#include <stdint.h>
#include "libbz3.h"
#define BZ3_HIP 1 /* this libbzip3 is the HIP implementation: the bz3_hip_* entry points exist */

#ifdef __cplusplus
extern "C" {
#endif

/* Number of usable HIP devices (0 when the runtime finds none). */
BZIP3_API int bz3_hip_device_count(void);
/* Pin every state created afterwards BY THIS PROCESS to `device` (>= 0), or restore round-robin (-1).
 * Returns 0, or -1 for an invalid device.  Environment BZ3_HIP_DEVICE=<n> has the same effect. */
BZIP3_API int bz3_hip_bind_device(int device);
/* Device a state is bound to. */
BZIP3_API int bz3_hip_state_device(struct bz3_state * state);

/* CM kernel variant (process-wide).  0 = the whole 145.5 KiB model in LDS, one block per CU; 1 / 2 = row-cache
 * kernels (the order-1 rows a block uses are cached in LDS -- 96 or 44/56 of them --, the others spill to HBM:
 * two / three blocks per CU; a block whose working set does not fit is handed back to variant 0 automatically);
 * -1 = automatic (default), by batch size: up to one block per CU variant 0, up to two per CU variant 1, beyond that variant 2.
 * Environment BZ3_HIP_CM_MODE=auto|full|rows|rows3 has the same effect.  Output bytes do not depend on the variant.
 * Returns 0, or -1 for an invalid mode. */
BZIP3_API int bz3_hip_set_cm_mode(int mode);
/* Test hook: how many more code windows the suffix sorter gives groups that are too large for its in-LDS kernels before rank doubling
 * takes them (0 = none: straight to the deep path; k < 0 = the default, 1).  Output bytes do not depend on it. */
BZIP3_API void bz3_hip_debug_bwt_big_rounds(int k);
/* Number of blocks the row-cache kernels have handed back to the full-model kernels so far (statistics). */
BZIP3_API unsigned bz3_hip_cm_blocks_given_up(void);
/* Number of blocks sent STRAIGHT to the whole-model CM kernels so far (statistics): blocks of batches that take a row-cache variant (more blocks than
 * CUs) with many live byte values (encode) or a payload that hardly shrank (decode).  A batch of at most one block per CU is never split. */
BZIP3_API unsigned bz3_hip_cm_blocks_routed_full(void);

/* The CM kernel variant (0..8, 12 as above) a batch of `blocks` blocks on `device` is coded (encode != 0) or decoded with under the
 * current mode; -1 for an invalid device. */
BZIP3_API int bz3_hip_cm_variant_for(int device, int blocks, int encode);

/* Test hook: the largest number of per-GPU groups of one batch call that have been running at the same time since the
 * last reset (a batch whose states live on G GPUs runs G groups concurrently, one host thread per GPU). */
BZIP3_API int bz3_hip_debug_peak_concurrent_groups(int reset);

/* Test hook: shape of the ring the last bz3_encode_blocks / bz3_hip_encode_blocks_device group ran its front end through:
 * blocks per window | context slots << 16 | (workspace handed back when the call ended) << 30 (0 before the first call).  The serial LZP drivers of a window run on a side stream
 * while the whole-GPU stages of the other slots' windows run on the group's stream; the shape follows the free memory. */
BZIP3_API int bz3_hip_debug_front_end_ring(void);
/* Keep-workspace mode for lean states (same as BZ3_HIP_KEEP_WS=1 in the environment, but switchable: 1 on, 0 off, -1 back to the environment):
 * a GPU-filling lean batch's workspace survives the call, the decode call that follows reuses it and carves the swap buffers of its tail windows
 * from it -- instead of a hipFree and two multi-GB hipMallocs per round trip (30-45 ms per GiB).  The memory stays with the library until
 * bz3_hip_release_cached_memory(), within the headroom rule below.  bench.py turns it on for its timed steps. */
BZIP3_API int bz3_hip_set_keep_workspace(int on);
/* Two-thread front end of the encoder (round 6; 1 on, 0 off, -1 back to the environment: BZ3_HIP_FRONT_DUO, read once).  A batch of 16 blocks or more runs the first
 * half of every block's front end (CRC, mRLE, LZP preparation) on a second host thread and stream, up to ring-slots - 1 windows ahead of the second half (LZP emission,
 * suffix sort) on the calling thread: each half stops for read-backs the host needs, and the kernels of one fill the other's bubbles.  Costs a second scratch region
 * (~30 bytes per byte of the largest block).  Output bytes do not depend on it.  bz3_hip_debug_front_end_ring() reports bit 29 when the last call took this form. */
BZIP3_API int bz3_hip_set_front_end_duo(int on);
/* Headroom: the device memory the library leaves to the host program (the caller owns its memory; the library's workspace, its pool of swap buffers and --
 * with keep-workspace -- a GPU-filling batch's whole ring are caches).  Rule: when a batch call returns, at least `bytes` of the device are free
 * (hipMemGetInfo), or the library holds nothing cached on that device.  The rings are sized for it and the rule is enforced when a call ends (idle pooled swap
 * buffers go back to the driver first, the workspace second).  Default 4 GiB; environment BZ3_HIP_WS_HEADROOM_MB; bytes < 0 = back to the environment;
 * 0 = no rule (the library may keep whatever it grew to). */
BZIP3_API void bz3_hip_set_workspace_headroom(long long bytes);
BZIP3_API size_t bz3_hip_workspace_headroom(void);
/* Statistics: how often the rule had to be enforced since the last reset -- returns the pool trims, *releases receives the workspace releases. */
BZIP3_API unsigned bz3_hip_debug_headroom_events(int reset, unsigned * releases);
/* tests only: LZP contexts the encoder's front-end ring may hold (api.hip ring_contexts_for), the arena's slack beyond a request, and the bytes the
 * library holds cached on `device` right now (workspace + idle pooled swap buffers). */
BZIP3_API size_t bz3_hip_debug_ring_contexts(size_t free_bytes, size_t have, size_t need, size_t fixed, size_t ctx_bytes, size_t cap, int lean, size_t headroom);
BZIP3_API size_t bz3_hip_debug_arena_slack(size_t bytes);
BZIP3_API unsigned bz3_hip_debug_cm_launches(int reset); /* statistics: CM kernel launches of the batch paths since the last reset */
BZIP3_API size_t bz3_hip_debug_workspace_bytes(size_t block_bytes, int which); /* 0: per-block scratch of the stages, 1: one LZP context of the encoder's ring */
BZIP3_API size_t bz3_hip_debug_cached_bytes(int device);
/* Statistics of the keep-workspace experiment (environment BZ3_HIP_KEEP_WS=1: a lean batch's workspace survives the call and the decoder's tail carves
 * its swap buffers from it): swap buffers served from the arena instead of the pool since the last reset. */
BZIP3_API int bz3_hip_debug_arena_swap_buffers(int reset);

/* Lean states (process-wide switch, read by bz3_new; environment BZ3_HIP_LEAN=1 has the same effect).  A state
 * normally owns its swap buffer (the reference's swap_buffer, bz3_bound(block_size) bytes of HBM) for life, so a
 * batch of N blocks holds 2 N block-sized buffers.  A lean state owns none: it borrows one from a per-GPU pool only
 * while its block is in the whole-GPU stages, the CM encoder works in place in the caller's buffer (input at the end
 * of the bz3_bound(size) bytes the API guarantees, coded bytes growing from the start), and the CM decoder reads a
 * staged copy of the coded payload -- about 1.2 block-sized buffers per block in flight, which is what lets 3 x 256
 * blocks of 256 MiB share one MI355X.  Output bytes, return values and error codes are the same.  Returns 0. */
BZIP3_API int bz3_hip_set_lean_states(int on);
/* Frees the per-GPU workspace and the idle pooled swap buffers (they are otherwise kept for the next call). */
BZIP3_API void bz3_hip_release_cached_memory(void);

/* Device-resident variants: `buffer` / `buffers[i]` are device addresses on the state's GPU with the
 * same capacities the host API requires (bz3_bound(size) for encode; buffer_size for decode). */
BZIP3_API int32_t bz3_hip_encode_block_device(struct bz3_state * state, void * buffer, int32_t size);
BZIP3_API int32_t bz3_hip_decode_block_device(struct bz3_state * state, void * buffer, size_t buffer_size, int32_t compressed_size,
                                              int32_t orig_size);
BZIP3_API void bz3_hip_encode_blocks_device(struct bz3_state * states[], void * buffers[], int32_t sizes[], int32_t n);
BZIP3_API void bz3_hip_decode_blocks_device(struct bz3_state * states[], void * buffers[], size_t buffer_sizes[], int32_t sizes[],
                                            int32_t orig_sizes[], int32_t n);

/* Stage timings (milliseconds) of the last block processed by `state`.  Timing a stage means waiting for the stream, so since round 4 only
 * the FIRST state of a batch (per GPU) is timed: its CRC / RLE / BWT entries are stage times, its LZP entry includes the window's driver
 * launch; for every other state of the batch CRC / BWT read 0 and RLE / LZP are launch (enqueue) times, not kernel times.  CM is the batch's
 * launch on every state. */
enum {
    BZ3_HIP_T_CRC = 0,
    BZ3_HIP_T_RLE = 1,
    BZ3_HIP_T_LZP = 2,
    BZ3_HIP_T_BWT = 3,
    BZ3_HIP_T_CM = 4,   /* CM kernel alone, measured with HIP events on the state's stream */
    BZ3_HIP_T_COPY = 5, /* host<->device and device<->device block copies */
    BZ3_HIP_T_COUNT = 8
};
BZIP3_API void bz3_hip_last_timings(struct bz3_state * state, float ms[BZ3_HIP_T_COUNT]);
/* BWT statistics of the last encoded block: doubling rounds, radix passes, elements pushed through the sorter. */
BZIP3_API void bz3_hip_last_bwt_stats(struct bz3_state * state, int32_t * rounds, int32_t * radix_passes, uint64_t * sorted_elements);

/* Single-block calls (bz3_encode_block / bz3_decode_block) that arrive from several host threads within this window are collected into
 * ONE batch per direction (the reference's own batch API is N threads with one block each, src/libbz3.c:831-856; here a batch is one
 * CM launch instead of N).  Default 200 us; 0 = no waiting (callers that arrive while a batch runs still form the next batch). */
BZIP3_API void bz3_hip_set_collect_window_us(int us);
BZIP3_API unsigned bz3_hip_debug_collected_batches(int reset, unsigned * largest); /* statistics: batches run for single-block callers */

/* ---- per-stage hooks on HOST buffers (tests / profiling).  Return values mirror the reference stage. */
BZIP3_API uint32_t bz3_hip_stage_crc32c(const uint8_t * data, size_t n, uint32_t init);             /* crc32sum        */
BZIP3_API int32_t bz3_hip_stage_mrle_encode(const uint8_t * in, int32_t n, uint8_t * out);          /* mrlec           */
BZIP3_API int bz3_hip_stage_mrle_decode(const uint8_t * in, uint8_t * out, int32_t outlen, int32_t maxin); /* mrled    */
BZIP3_API int32_t bz3_hip_stage_lzp_encode(const uint8_t * in, int32_t n, uint8_t * out);           /* lzp_compress    */
BZIP3_API int32_t bz3_hip_stage_lzp_decode(const uint8_t * in, int32_t n, uint8_t * out, int32_t max); /* lzp_decompress */
BZIP3_API int32_t bz3_hip_stage_bwt(const uint8_t * in, uint8_t * out, int32_t n);                  /* libsais_bwt     */
BZIP3_API int32_t bz3_hip_stage_unbwt(const uint8_t * in, uint8_t * out, int32_t n, int32_t idx);   /* libsais_unbwt   */
/* tests only: the two CU masks of the decoder's partition (side streams / everything else) for `cus` CUs of which `reserve` are set aside; returns the words per mask */
BZIP3_API int32_t bz3_hip_debug_cu_masks(int cus, int reserve, uint32_t * side, uint32_t * rest);
/* tests only: sort.hip's device-wide exclusive prefix sum, in place on a host buffer */
BZIP3_API int32_t bz3_hip_debug_scan_u32(uint32_t * data, uint32_t n, uint32_t * total);
/* tests only: sort.hip's stable LSD radix sort of (keys[i], i) over key bits [0, key_bits), digits of 8 or 9 bits; returns the passes run */
BZIP3_API int32_t bz3_hip_debug_sort_u32(const uint32_t * keys, uint32_t n, int key_bits, int digit_bits, uint32_t * sorted_keys, uint32_t * sorted_index);
BZIP3_API float bz3_hip_stage_last_ms(void); /* wall ms of the transform inside the last bz3_hip_stage_bwt / _unbwt call (allocations and PCIe copies excluded) */
BZIP3_API int32_t bz3_hip_stage_cm_encode(const uint8_t * in, int32_t n, uint8_t * out);            /* encode_bytes    */
BZIP3_API void bz3_hip_stage_cm_decode(const uint8_t * in, int32_t in_size, uint8_t * out, int32_t n); /* decode_bytes  */

/* Streaming file codec (the reference CLI's driver loop, src/main.c:351-407, as a pipeline): reads blocks
 * from in_fd, codes `blocks_per_batch` of them at a time on the GPU(s) while the next batch is being read and the previous
 * one written, writes the reference's file format to out_fd ("BZ3v1", u32le block size, then per block u32le coded size,
 * u32le original size, block bytes -- byte-identical to `bzip3 -e -b`, decodable by `bzip3 -d`, and vice versa).
 * Returns 0, or a BZ3_ERR_* code (of the first failing block; the blocks before it have been written), or BZ3_HIP_ERR_IO.
 * Host buffers: page-locked up to BZ3_HIP_STREAM_PINNED_MIB MiB in total (environment, read once per process; default 4096 = 4 GiB), plain
 * malloc'ed buffers beyond that or when the host refuses -- larger configurations (blocks_per_batch x block size x 3 buffers above the
 * budget) therefore mix pinned and pageable buffers and copy the pageable ones through the runtime's staging at roughly half the rate;
 * raise the variable where the host has the lockable memory. */
#define BZ3_HIP_ERR_IO (-100)
BZIP3_API int bz3_hip_encode_stream(int in_fd, int out_fd, int32_t block_size, int32_t blocks_per_batch);
BZIP3_API int bz3_hip_decode_stream(int in_fd, int out_fd, int32_t blocks_per_batch);

/* Profiling: `copies` identical CM decode jobs in one launch through the current CM kernel variant; returns the launch
 * time in milliseconds, `out` receives the n (>= 256) decoded bytes of copy 0; with BZ3_CM_DEBUG=3 `counters` (u64[16] per
 * copy, may be NULL) receives the decoder's phase cycle counters instead of valid output (bzip3_amd/csrc/api.hip). */
BZIP3_API float bz3_hip_stage_cm_decode_many(const uint8_t * in, int32_t in_size, uint8_t * out, int32_t n, int32_t copies, uint64_t * counters);

/* The same for the encoder: `copies` identical CM encode jobs in one launch; *coded = coded size of copy 0, its bytes in `out`
 * (capacity bz3_bound(n)).  BZ3_CM_DEBUG=1 / 2: coder wave / model waves alone (output invalid). */
BZIP3_API float bz3_hip_stage_cm_encode_many(const uint8_t * in, int32_t n, uint8_t * out, int32_t * coded, int32_t copies);

#ifdef __cplusplus
}
#endif
#endif
