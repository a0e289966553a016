/*
 * libbz3.h -- the C ABI of the MI355X-native bzip3 block codec (bzip3_amd).
 *
 * This header declares exactly the 14 symbols that the reference library exports
 * (kspalaiologos/bzip3 v1.5.2, include/libbz3.h:62-235) with the same names, argument
 * meaning, return values and error codes, so that the reference CLI (src/main.c) and any
 * existing binding link against libbzip3.so from this repository unchanged.  Only the
 * implementation differs: every block stage (CRC-32C, mRLE, LZP, BWT, CM coder and their
 * inverses) runs as hand-written HIP kernels on a gfx950 device; there is no CPU code path.
 * The text below is written for this implementation; citations `ref:` point at the reference
 * declaration each entry replaces.
 */
#ifndef LIBBZ3_H
#define LIBBZ3_H

#include <stddef.h>
#include <stdint.h>

/* Symbol visibility / import-export control, the same switches as the reference's header (ref: include/libbz3.h:27-41): a caller that
 * defines BZIP3_DLL_IMPORT=1 (or a build that defines BZIP3_DLL_EXPORT=1, BZIP3_VISIBLE) keeps compiling against this header.
 * (The header is written for this implementation rather than copied: the rules of this repository keep reference sources out of
 * it; tests/test_abi.py checks the 14 prototypes against the reference's header whenever /root/reference is present.) */
#ifndef BZIP3_VISIBLE
#  if defined(__GNUC__) && (__GNUC__ >= 4) && !defined(__MINGW32__)
#    define BZIP3_VISIBLE __attribute__((visibility("default")))
#  else
#    define BZIP3_VISIBLE
#  endif
#endif
#if defined(BZIP3_DLL_EXPORT) && (BZIP3_DLL_EXPORT == 1)
#  define BZIP3_API __declspec(dllexport) BZIP3_VISIBLE
#elif defined(BZIP3_DLL_IMPORT) && (BZIP3_DLL_IMPORT == 1)
#  define BZIP3_API __declspec(dllimport) BZIP3_VISIBLE
#else
#  define BZIP3_API BZIP3_VISIBLE
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* Error codes, ref: include/libbz3.h:47-55.  HIP failures (no device, out of device memory, a
 * kernel fault) are reported as BZ3_ERR_INIT by bz3_new-time paths (NULL return) and as
 * BZ3_ERR_BWT by run-time paths, the two codes the CLI treats generically. */
#define BZ3_OK 0
#define BZ3_ERR_OUT_OF_BOUNDS -1
#define BZ3_ERR_BWT -2
#define BZ3_ERR_CRC -3
#define BZ3_ERR_MALFORMED_HEADER -4
#define BZ3_ERR_TRUNCATED_DATA -5
#define BZ3_ERR_DATA_TOO_BIG -6
#define BZ3_ERR_INIT -7
#define BZ3_ERR_DATA_SIZE_TOO_SMALL -8

struct bz3_state; /* opaque; owns a HIP stream and the device-side ping-pong buffers of one block */

/* ref: libbz3.h:62 */
BZIP3_API const char * bz3_version(void);
/* ref: libbz3.h:67 -- error of the most recent call on this state */
BZIP3_API int8_t bz3_last_error(struct bz3_state * state);
/* ref: libbz3.h:72 */
BZIP3_API const char * bz3_strerror(struct bz3_state * state);

/* ref: libbz3.h:79.  block_size in [65 KiB, 511 MiB], else NULL.  Also NULL when no HIP device is
 * usable or device memory cannot be allocated.  The state is bound to one GPU (round-robin over the
 * visible devices unless bz3_hip_bind_device() / BZ3_HIP_DEVICE pins it; see bz3_hip.h). */
BZIP3_API struct bz3_state * bz3_new(int32_t block_size);
/* ref: libbz3.h:84 */
BZIP3_API void bz3_free(struct bz3_state * state);
/* ref: libbz3.h:89 -- n + n/50 + 32 */
BZIP3_API size_t bz3_bound(size_t input_size);

/* Frame API (13-byte "BZ3v1" header with block count), ref: libbz3.h:101, :110. */
BZIP3_API int bz3_compress(uint32_t block_size, const uint8_t * in, uint8_t * out, size_t in_size, size_t * out_size);
BZIP3_API int bz3_decompress(const uint8_t * in, uint8_t * out, size_t in_size, size_t * out_size);

/* ref: libbz3.h:167.  Bytes (host + device) a bz3_new(block_size) allocates eagerly; 0 for an
 * invalid block size.  The suffix-sort workspace is per device, shared by all states and not counted. */
BZIP3_API size_t bz3_min_memory_needed(int32_t block_size);

/* ref: libbz3.h:176.  In place on a HOST buffer of at least bz3_bound(size) bytes.  Returns the
 * encoded size or -1 (bz3_last_error tells why). */
BZIP3_API int32_t bz3_encode_block(struct bz3_state * state, uint8_t * buffer, int32_t size);
/* ref: libbz3.h:194.  In place on a HOST buffer; returns the decoded size or -1. */
BZIP3_API int32_t bz3_decode_block(struct bz3_state * state, uint8_t * buffer, size_t buffer_size, int32_t compressed_size,
                                   int32_t orig_size);

/* ref: libbz3.h:206, :212.  n independent (state, buffer) pairs.  The reference forks one pthread per
 * block; here all blocks are enqueued on their states' streams (one CU per block for the CM coder, the
 * whole GPU for each block's BWT in turn) and joined once.  States may live on different GPUs. */
BZIP3_API void bz3_encode_blocks(struct bz3_state * states[], uint8_t * buffers[], int32_t sizes[], int32_t n);
BZIP3_API void bz3_decode_blocks(struct bz3_state * states[], uint8_t * buffers[], size_t buffer_sizes[], int32_t sizes[],
                                 int32_t orig_sizes[], int32_t n);

/* ref: libbz3.h:235 -- 1 / 0 / -1 exactly like the reference. */
BZIP3_API int bz3_orig_size_sufficient_for_decode(const uint8_t * block, size_t block_size, int32_t orig_size);

#ifdef __cplusplus
}
#endif
#endif
