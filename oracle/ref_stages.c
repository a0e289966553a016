/*
 * oracle/ref_stages.c -- TEST INFRASTRUCTURE.  Thin exported wrappers around the REAL reference's
 * static stage functions, so each restated oracle function and each HIP kernel can be diffed
 * against its own stage (crc, mrlec, lzp_compress, libsais_bwt, encode_bytes, ...), not only end
 * to end.  Same inclusion pattern as the reference's own fuzz harnesses
 * (examples/fuzz-round-trip.c:43-44).  The reference sources are compiled where they lie under
 * /root/reference; nothing is copied into this repository.  Output: oracle/_ref/ (git-ignored).
 */
#include "src/libbz3.c" /* resolved through -I/root/reference by oracle/Makefile */

#define REF_API __attribute__((visibility("default")))

REF_API u32 ref_crc32(u32 init, u8 * buf, size_t n) { return crc32sum(init, buf, n); }
REF_API s32 ref_mrlec(u8 * in, s32 n, u8 * out) { return mrlec(in, n, out); }
REF_API int ref_mrled(u8 * in, u8 * out, s32 outlen, s32 maxin) { return mrled(in, out, outlen, maxin); }
REF_API s32 ref_lzp_compress(const u8 * in, u8 * out, s32 n) {
    s32 * lut = calloc(1 << LZP_DICTIONARY, sizeof(s32));
    s32 r = lzp_compress(in, out, n, lut);
    free(lut);
    return r;
}
REF_API s32 ref_lzp_decompress(const u8 * in, u8 * out, s32 n, s32 max) {
    s32 * lut = calloc(1 << LZP_DICTIONARY, sizeof(s32));
    s32 r = lzp_decompress(in, out, n, max, lut);
    free(lut);
    return r;
}
REF_API s32 ref_bwt(const u8 * T, u8 * U, s32 n) {
    s32 * A = calloc((size_t)n + 128, sizeof(s32));
    s32 r = libsais_bwt(T, U, A, n, 0, NULL);
    free(A);
    return r;
}
REF_API s32 ref_unbwt(const u8 * T, u8 * U, s32 n, s32 idx) {
    s32 * A = calloc((size_t)n + 128, sizeof(s32));
    s32 r = libsais_unbwt(T, U, A, n, NULL, idx);
    free(A);
    return r;
}
REF_API s32 ref_cm_encode(u8 * in, s32 n, u8 * out) {
    state * s = malloc(sizeof(state));
    begin(s);
    s->out_queue = out;
    s->output_ptr = 0;
    encode_bytes(s, in, n);
    s32 r = s->output_ptr;
    free(s);
    return r;
}
REF_API void ref_cm_decode(u8 * in, s32 insize, u8 * out, s32 n) {
    state * s = malloc(sizeof(state));
    begin(s);
    s->in_queue = in;
    s->input_ptr = 0;
    s->input_max = insize;
    decode_bytes(s, out, n);
    free(s);
}
