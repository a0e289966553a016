/*
 * oracle/bz3_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded CPU restatement of the bzip3 block codec
 * (kspalaiologos/bzip3 v1.5.2).  It exists so that tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg can check the HIP path bit-for-bit.  Nothing
 * under bzip3_amd/ may include, link or call this file.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks every function below
 *   (a) stage by stage against the reference's own static functions, compiled from
 *       /root/reference by oracle/Makefile into oracle/_ref/libbz3ref_stages.so, and
 *   (b) against the reference's only known-answer vector, examples/shakespeare.txt.bz3
 *       (committed as tests/golden/shakespeare.txt.bz3; plaintext md5
 *       d2028225a89d8b0b3093dddb720da91f).
 *
 * Every function cites the reference lines whose behaviour it restates
 * (paths relative to /root/reference).  The code is written from the rules in
 * SURVEY.md sections 7/8, in the same "closed form" the GPU kernels use (runs,
 * visited positions, LF mapping), not transcribed from the reference.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t s32;
typedef int64_t s64;

/* error codes: include/libbz3.h:47-55 */
enum {
    ORC_OK = 0,
    ORC_ERR_OUT_OF_BOUNDS = -1,
    ORC_ERR_BWT = -2,
    ORC_ERR_CRC = -3,
    ORC_ERR_MALFORMED_HEADER = -4,
    ORC_ERR_TRUNCATED_DATA = -5,
    ORC_ERR_DATA_TOO_BIG = -6,
    ORC_ERR_INIT = -7,
    ORC_ERR_DATA_SIZE_TOO_SMALL = -8
};

static u32 ld_le32(const u8 * p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }
static void st_le32(u8 * p, u32 v) {
    p[0] = (u8)v;
    p[1] = (u8)(v >> 8);
    p[2] = (u8)(v >> 16);
    p[3] = (u8)(v >> 24);
}
static u32 ld_be32(const u8 * p) { return ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | (u32)p[3]; }
/* raw 4-byte equality probe (the reference compares native u32 loads; equality is endian-free) */
static int eq4(const u8 * a, const u8 * b) { return memcmp(a, b, 4) == 0; }

/* bz3_bound: src/libbz3.c:510 */
ORC_API size_t orc_bound(size_t n) { return n + n / 50 + 32; }

/* ------------------------------------------------------------------------------------------
 * CRC-32C, reflected polynomial 0x82F63B78, caller-supplied start state, no final xor.
 * Restates crc32sum + crc32Table (src/libbz3.c:37-72); the block codec calls it with state 1
 * (src/libbz3.c:593, :686, :803).  Bitwise form: one conditional xor per message bit.
 * ------------------------------------------------------------------------------------------ */
ORC_API u32 orc_crc32c(u32 state, const u8 * buf, size_t n) {
    for (size_t i = 0; i < n; i++) {
        state ^= buf[i];
        for (int k = 0; k < 8; k++) state = (state >> 1) ^ (0x82F63B78u & (0u - (state & 1u)));
    }
    return state;
}

/* ------------------------------------------------------------------------------------------
 * mRLE encoder.  Restates mrlec (src/libbz3.c:264-301) in run form:
 *   gain[c] = sum over maximal runs (symbol c, length L) of  L - 2 - floor((L-1)/255);
 *   header  = 32-byte bitmap, bit c set iff gain[c] > 0;
 *   a run of a flagged symbol becomes  c, 0xFF x ceil(L/255)-1, (L-1) mod 255 ;
 *   a run of an unflagged symbol is copied verbatim.
 * `out` needs n + 32 bytes.  Returns the encoded size.
 * ------------------------------------------------------------------------------------------ */
ORC_API s32 orc_mrle_encode(const u8 * in, s32 n, u8 * out) {
    s64 gain[256];
    memset(gain, 0, sizeof gain);
    for (s32 i = 0; i < n;) {
        s32 j = i + 1;
        while (j < n && in[j] == in[i]) j++;
        s32 L = j - i;
        gain[in[i]] += (s64)L - 2 - (L - 1) / 255;
        i = j;
    }
    memset(out, 0, 32);
    for (int c = 0; c < 256; c++)
        if (gain[c] > 0) out[c >> 3] |= (u8)(1u << (c & 7));
    s32 op = 32;
    for (s32 i = 0; i < n;) {
        s32 j = i + 1;
        while (j < n && in[j] == in[i]) j++;
        s32 L = j - i;
        u8 c = in[i];
        if (gain[c] > 0) {
            out[op++] = c;
            for (; L > 255; L -= 255) out[op++] = 255;
            out[op++] = (u8)(L - 1);
        } else {
            memset(out + op, c, (size_t)L);
            op += L;
        }
        i = j;
    }
    return op;
}

/* ------------------------------------------------------------------------------------------
 * mRLE decoder.  Restates mrled (src/libbz3.c:303-329), including its behaviour on truncated
 * input: a length sequence cut off by `maxin` re-uses the last length byte that was read
 * (initially -1), exactly like the reference's `pc` variable (:320-322).
 * Returns 0 on success (exactly outlen bytes produced), 1 otherwise.
 * ------------------------------------------------------------------------------------------ */
ORC_API int orc_mrle_decode(const u8 * in, u8 * out, s32 outlen, s32 maxin) {
    if (maxin < 32) return 1;
    s32 ip = 32, op = 0, last = -1;
    while (op < outlen && ip < maxin) {
        u8 c = in[ip++];
        if ((in[c >> 3] >> (c & 7)) & 1) {
            s32 run = 0;
            while (ip < maxin) {
                last = in[ip++];
                if (last != 255) break;
                run += 255;
            }
            run += last + 1;
            for (; run > 0 && op < outlen; run--) out[op++] = c;
        } else {
            out[op++] = c;
        }
    }
    return op != outlen;
}

/* ------------------------------------------------------------------------------------------
 * LZP.  Shared pieces: 2^18-entry table of "last visited position whose preceding 4 bytes hash
 * here"; hash of the big-endian 4-byte context (src/libbz3.c:84-87, :135, :138).
 * ------------------------------------------------------------------------------------------ */
#define ORC_LZP_BITS 18
#define ORC_LZP_MIN 40
#define ORC_LZP_ESC 0xF2

static u32 lzp_slot(const u8 * at) { /* hash of the 4 bytes preceding `at` */
    u32 ctx = ld_be32(at - 4);
    return ((ctx >> 15) ^ ctx ^ (ctx >> 3)) & ((1u << ORC_LZP_BITS) - 1);
}

/* LZP encoder.  Restates lzp_compress -> lzp_encode_block (src/libbz3.c:243-249, :124-198).
 * `out` needs n + 16 bytes.  Returns the encoded size, or -1 when n < 72 or when the output
 * reaches n - 8 bytes (:128, :197). */
ORC_API s32 orc_lzp_encode(const u8 * in, s32 n, u8 * out) {
    if (n < ORC_LZP_MIN + 32) return -1;
    s32 * tab = (s32 *)calloc((size_t)1 << ORC_LZP_BITS, sizeof(s32));
    if (!tab) return -1;
    const s32 main_end = n - ORC_LZP_MIN - 32; /* main loop visits positions < n-72 (:137) */
    const s32 out_lim = n - 8;                 /* :128 */
    s32 ip = 4, op = 4, heur = 0;
    memcpy(out, in, 4);
    while (ip < main_end && op < out_lim) {
        u32 h = lzp_slot(in + ip);
        s32 cand = tab[h];
        tab[h] = ip; /* inserted before testing (:139-140) */
        int took_match = 0;
        if (cand > 0 && eq4(in + ip + ORC_LZP_MIN - 4, in + cand + ORC_LZP_MIN - 4) && eq4(in + ip, in + cand)) {
            /* heuristic early-out (:145): re-test the word where the last short candidate failed */
            if (!(heur > ip && !eq4(in + heur, in + cand + (heur - ip)))) {
                s32 len = 4;
                while (ip + len < main_end && eq4(in + ip + len, in + cand + len)) len += 4; /* :148-150 */
                if (len < ORC_LZP_MIN) {
                    if (heur < ip + len) heur = ip + len; /* :152-155 */
                } else {
                    len += in[ip + len] == in[cand + len]; /* :157-159 */
                    len += in[ip + len] == in[cand + len];
                    len += in[ip + len] == in[cand + len];
                    ip += len;
                    out[op++] = ORC_LZP_ESC;
                    len -= ORC_LZP_MIN;
                    while (len >= 254) { /* :167-171 */
                        len -= 254;
                        out[op++] = 254;
                        if (op >= out_lim) break;
                    }
                    out[op++] = (u8)len;
                    took_match = 1;
                }
            }
        }
        if (!took_match) {
            u8 b = in[ip++];
            out[op++] = b;
            if (b == ORC_LZP_ESC && cand > 0) out[op++] = 255; /* escape only if the slot was live (:176-181) */
        }
    }
    while (ip < n && op < out_lim) { /* tail: no matching, still inserts and escapes (:187-195) */
        u32 h = lzp_slot(in + ip);
        s32 cand = tab[h];
        tab[h] = ip;
        u8 b = in[ip++];
        out[op++] = b;
        if (b == ORC_LZP_ESC && cand > 0) out[op++] = 255;
    }
    free(tab);
    return op >= out_lim ? -1 : op;
}

/* LZP decoder.  Restates lzp_decompress -> lzp_decode_block (src/libbz3.c:251-257, :200-241).
 * `out` must hold `max` bytes.  Returns decoded size or -1. */
ORC_API s32 orc_lzp_decode(const u8 * in, s32 n, u8 * out, s32 max) {
    if (n < 4) return -1;
    s32 * tab = (s32 *)calloc((size_t)1 << ORC_LZP_BITS, sizeof(s32));
    if (!tab) return -1;
    s32 ip = 4, op = 4, rc = 0;
    memcpy(out, in, 4);
    while (ip < n && op < max) {
        u32 h = lzp_slot(out + op);
        s32 cand = tab[h];
        tab[h] = op;
        if (in[ip] == ORC_LZP_ESC && cand > 0) {
            ip++;
            if (ip == n) { rc = -1; break; } /* :215 */
            if (in[ip] != 255) {
                s32 len = ORC_LZP_MIN;
                for (;;) { /* :218-222 */
                    if (ip == n) { rc = -1; break; }
                    u8 b = in[ip++];
                    len += b;
                    if (b != 254) break;
                }
                if (rc) break;
                s64 stop = (s64)op + len;
                if (stop > max) stop = max;
                s32 src = cand;
                while (op < stop) out[op++] = out[src++]; /* may self-overlap (:228) */
            } else {
                ip++;
                out[op++] = ORC_LZP_ESC;
            }
        } else {
            out[op++] = in[ip++];
        }
    }
    free(tab);
    return rc ? -1 : op;
}

/* ------------------------------------------------------------------------------------------
 * Suffix array by prefix doubling with LSD radix sorting of rank pairs (the same scheme the
 * GPU path uses).  Order: plain suffix order, a suffix that is a proper prefix of another
 * sorts first.  Ranks are 1-based; "past the end" is rank 0.
 * ------------------------------------------------------------------------------------------ */
static void radix_pass16(const s32 * src, s32 * dst, const u32 * key, s32 n, int shift, u32 * cnt) {
    memset(cnt, 0, 65537 * sizeof(u32));
    for (s32 i = 0; i < n; i++) cnt[((key[src[i]] >> shift) & 0xFFFF) + 1]++;
    for (int d = 0; d < 65536; d++) cnt[d + 1] += cnt[d];
    for (s32 i = 0; i < n; i++) dst[cnt[(key[src[i]] >> shift) & 0xFFFF]++] = src[i];
}

static int orc_suffix_array(const u8 * T, s32 n, s32 * SA) {
    u32 * rk = (u32 *)malloc((size_t)n * 4);
    u32 * k2 = (u32 *)malloc((size_t)n * 4);
    s32 * tmp = (s32 *)malloc((size_t)n * 4);
    u32 * cnt = (u32 *)malloc(65537 * sizeof(u32));
    if (!rk || !k2 || !tmp || !cnt) { free(rk); free(k2); free(tmp); free(cnt); return -1; }
    /* order-2 start: 9 bits per symbol so that "no symbol" (0) < byte 0x00 (1) */
    for (s32 i = 0; i < n; i++) rk[i] = ((u32)T[i] + 1) * 257 + (i + 1 < n ? (u32)T[i + 1] + 1 : 0);
    for (s32 i = 0; i < n; i++) SA[i] = i;
    radix_pass16(SA, tmp, rk, n, 0, cnt);
    radix_pass16(tmp, SA, rk, n, 16, cnt);
    /* densify */
    {
        u32 r = 0, prev = 0;
        for (s32 i = 0; i < n; i++) {
            u32 k = rk[SA[i]];
            if (i == 0 || k != prev) r = (u32)i + 1;
            prev = k;
            k2[SA[i]] = r;
        }
        memcpy(rk, k2, (size_t)n * 4);
    }
    for (s32 h = 2;; h *= 2) {
        int all_unique = 1;
        for (s32 i = 0; i + 1 < n; i++)
            if (rk[SA[i]] == rk[SA[i + 1]]) { all_unique = 0; break; }
        if (all_unique) break;
        for (s32 i = 0; i < n; i++) k2[i] = (i + h < n) ? rk[i + h] : 0;
        radix_pass16(SA, tmp, k2, n, 0, cnt);
        radix_pass16(tmp, SA, k2, n, 16, cnt);
        radix_pass16(SA, tmp, rk, n, 0, cnt);
        radix_pass16(tmp, SA, rk, n, 16, cnt);
        /* new ranks = 1 + position of the head of each (rk, k2) group */
        u32 r = 0;
        for (s32 i = 0; i < n; i++) {
            if (i == 0 || rk[SA[i]] != rk[SA[i - 1]] || k2[SA[i]] != k2[SA[i - 1]]) r = (u32)i + 1;
            ((u32 *)tmp)[SA[i]] = r;
        }
        memcpy(rk, tmp, (size_t)n * 4);
        if (h > n) break;
    }
    free(rk); free(k2); free(tmp); free(cnt);
    return 0;
}

/* Forward BWT.  Restates what libsais_bwt returns (include/libsais.h:4095-4121) by definition:
 *   idx = 1 + (row of suffix 0);  U = T[n-1] followed by T[SA[i]-1] for every SA[i] != 0.
 * Returns idx (>= 1; n for n <= 1) or -1. */
ORC_API s32 orc_bwt(const u8 * T, u8 * U, s32 n) {
    if (n < 0) return -1;
    if (n <= 1) {
        if (n == 1) U[0] = T[0];
        return n;
    }
    s32 * SA = (s32 *)malloc((size_t)n * 4);
    if (!SA || orc_suffix_array(T, n, SA) != 0) { free(SA); return -1; }
    s32 idx = -1, w = 0;
    U[w++] = T[n - 1];
    for (s32 i = 0; i < n; i++) {
        if (SA[i] == 0) idx = i + 1;
        else U[w++] = T[SA[i] - 1];
    }
    free(SA);
    return idx;
}

/* Inverse BWT.  Same function as libsais_unbwt (include/libsais.h:5260-5262, validation :5210-5232)
 * computed through the LF mapping: insert a virtual sentinel row at `idx`,
 * LF(r) = C[L[r]] + rank of r among equal symbols, T[n-1-k] = L[LF^k(0)].
 * Returns 0, or -1 for an index outside (0, n]. */
ORC_API s32 orc_unbwt(const u8 * U, u8 * T, s32 n, s32 idx) {
    /* Restates libsais_unbwt (include/libsais.h:5260-5262 -> :5210-5232, init :4593-4617, decode :4618-4636,
     * :5133-5156) for ANY input, not only for a genuine BWT.  Rows 0..n: row 0 is the empty suffix, row idx is the
     * row whose preceding character is the (virtual) sentinel, L[r] = U[r < idx ? r : r - 1] otherwise.
     *   LF(r)  = C[L[r]] + #{r' < r : L[r'] = L[r]}  (C[c] = 1 + #{bytes < c}),  LF(idx) = 0;   psi = LF^-1, psi(0) = idx.
     * The reference chases the bigram table P = psi o psi from p = idx for n/2 steps and emits (F[p], F[psi(p)]) per
     * step (:4618-4636), which is F[psi^j(idx)] byte by byte.  psi(0) = idx puts rows 0 and idx on one cycle, so the
     * chase always runs into E = LF(0) (the suffix made of the last character alone) after D <= n bytes; for a genuine
     * BWT D = n.  For anything else the table holds zeros there: P[E] was never written (the caller zeroes the work
     * array, src/libbz3.c:756), P[LF(E)] = 0 legitimately and P[0] = 0, so the chase is stuck at p = 0 and repeats the
     * bigram of the first non-empty bucket (fastbits[0], :4534-4553).  If E itself is hit on an even step its bucket
     * lookup (:4626-4631) yields the pair (last character, 0x00).  Finally U[n-1] = first BWT byte (:5155). */
    if (n < 0) return -1;
    if (n <= 1) {
        if (idx != n) return -1;
        if (n == 1) T[0] = U[0];
        return 0;
    }
    if (idx <= 0 || idx > n) return -1;
    u32 C[257], cum[257];
    memset(C, 0, sizeof C);
    for (s32 i = 0; i < n; i++) C[U[i] + 1]++;
    C[0] = 1; /* row 0 of F is the sentinel */
    for (int c = 0; c < 256; c++) C[c + 1] += C[c];
    memcpy(cum, C, sizeof cum); /* cum[c] = first row whose suffix starts with c; cum[256] = n + 1 */
    u32 * psi = (u32 *)malloc(((size_t)n + 1) * 4);
    u8 * F = (u8 *)malloc((size_t)n + 1);
    if (!psi || !F) { free(psi); free(F); return -1; }
    psi[0] = (u32)idx;
    for (s32 r = 0; r <= n; r++) {
        if (r == idx) continue; /* the sentinel row maps to F-row 0: psi[0] = idx */
        u8 c = U[r < idx ? r : r - 1];
        psi[C[c]++] = (u32)r;
    }
    F[0] = 0;
    for (int c = 0; c < 256; c++)
        for (u32 r = cum[c]; r < cum[c + 1]; r++) F[r] = (u8)c;
    const u8 lastc = U[0];
    const u32 E = cum[lastc];
    const s32 limit = 2 * (n / 2);
    s32 j = 0;
    u32 r = (u32)idx;
    while (j < limit && r != 0) {
        T[j++] = F[r];
        r = psi[r];
    }
    if (j < limit) { /* the chain ended after D = j bytes (not a genuine BWT) */
        if (j & 1) {
            /* E was hit on an even step, and E lies in no bucket.  The lookup (:4626-4631) starts at fastbits[E >> shift],
             * the first non-empty bucket that reaches into E's group of 2^shift rows: if that is a bucket before E (row
             * E-1 is in the same group) the scan walks up to the empty-or-not bucket (lastc, 0x00) right behind the
             * reserved row; otherwise it is already the bucket of row E+1.  (E = n leaves fastbits unset: undefined.) */
            int shift = 0;
            while ((n >> shift) > (1 << 17)) shift++;
            const int same_group = E >= 2 && ((E - 1) >> shift) == (E >> shift);
            if (same_group || E + 1 > (u32)n) {
                T[j++] = 0;
            } else {
                T[j - 1] = F[E + 1];
                T[j++] = F[psi[E + 1]];
            }
        }
        const u32 q0 = (E == 1) ? 2u : 1u; /* smallest row that has a bigram */
        const u8 hi = F[q0], lo = F[psi[q0]];
        for (; j < limit; j++) T[j] = (j & 1) ? lo : hi;
    }
    T[n - 1] = lastc;
    free(psi);
    free(F);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * CM model + binary arithmetic coder.  Restates state/begin/encode_bytes/decode_bytes
 * (src/libbz3.c:333-494).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    u16 c0[256];      /* order-0, indexed by tree node              (rate 2) */
    u16 c1[256][256]; /* order-1, [previous byte][tree node]        (rate 4) */
    u16 c2[512][17];  /* interpolated APM rows, [2*node + runflag]  (rate 6) */
} orc_cm_model;

static void cm_reset(orc_cm_model * m) { /* begin(): :350-358 */
    for (int i = 0; i < 256; i++) m->c0[i] = 32768;
    for (int i = 0; i < 256; i++)
        for (int j = 0; j < 256; j++) m->c1[i][j] = 32768;
    for (int r = 0; r < 512; r++)
        for (int k = 0; k < 17; k++) m->c2[r][k] = (u16)((k << 12) - (k == 16));
}

typedef struct {
    u16 *a, *b, *lo, *hi; /* the four counters an event updates */
    u32 p18;              /* 18-bit probability that the bit is 1 */
} cm_probe;

static cm_probe cm_predict(orc_cm_model * m, u32 node, u32 prev1, u32 prev2, int runflag) { /* :377-388 */
    cm_probe q;
    q.a = &m->c0[node];
    q.b = &m->c1[prev1][node];
    int p = (((int)*q.a + (int)*q.b) * 7 + 2 * (int)m->c1[prev2][node]) >> 4;
    int j = p >> 12;
    q.lo = &m->c2[2 * node + (u32)runflag][j];
    q.hi = q.lo + 1;
    int x1 = *q.lo, x2 = *q.hi;
    int ssep = x1 + (((x2 - x1) * (p & 4095)) >> 12); /* signed product, arithmetic shift (:385) */
    q.p18 = (u32)(ssep * 3 + p);
    return q;
}

static void cm_learn(const cm_probe * q, int bit) { /* update0/update1: :347-348, :396-399, :411-414 */
    if (bit) {
        *q->a += (u16)((*q->a ^ 65535) >> 2);
        *q->b += (u16)((*q->b ^ 65535) >> 4);
        *q->lo += (u16)((*q->lo ^ 65535) >> 6);
        *q->hi += (u16)((*q->hi ^ 65535) >> 6);
    } else {
        *q->a -= *q->a >> 2;
        *q->b -= *q->b >> 4;
        *q->lo -= *q->lo >> 6;
        *q->hi -= *q->hi >> 6;
    }
}

/* Encode n bytes; `out` must hold the coded size (<= ~1.01 n + 16).  Returns coded size. (:360-433) */
ORC_API s32 orc_cm_encode(const u8 * in, s32 n, u8 * out) {
    orc_cm_model * m = (orc_cm_model *)malloc(sizeof *m);
    if (!m) return -1;
    cm_reset(m);
    u32 low = 0, high = 0xFFFFFFFFu, prev1 = 0, prev2 = 0, run = 0;
    s32 op = 0;
    for (s32 i = 0; i < n; i++) {
        run = (prev1 == prev2) ? run + 1 : 0; /* :367-372 */
        int runflag = run > 2;
        u32 node = 1, sym = in[i];
        for (int k = 7; k >= 0; k--) {
            int bit = (sym >> k) & 1;
            cm_probe q = cm_predict(m, node, prev1, prev2, runflag);
            u32 mid = low + (u32)(((u64)(high - low) * q.p18) >> 18);
            if (bit) high = mid; else low = mid + 1;
            while ((low ^ high) < (1u << 24)) { /* :390-394 */
                out[op++] = (u8)(low >> 24);
                low <<= 8;
                high = (high << 8) | 0xFF;
            }
            cm_learn(&q, bit);
            node = node * 2 + (u32)bit;
        }
        prev2 = prev1;
        prev1 = node & 255;
    }
    for (int k = 0; k < 4; k++) { /* flush (:425-432) */
        out[op++] = (u8)(low >> 24);
        low <<= 8;
    }
    free(m);
    return op;
}

/* Decode n bytes from `in` (insize bytes; reads past the end return 0xFFFFFFFF like read_in's -1,
 * :345).  (:435-494) */
ORC_API void orc_cm_decode(const u8 * in, s32 insize, u8 * out, s32 n) {
    orc_cm_model * m = (orc_cm_model *)malloc(sizeof *m);
    if (!m) return;
    cm_reset(m);
    u32 low = 0, high = 0xFFFFFFFFu, code = 0, prev1 = 0, prev2 = 0, run = 0;
    s32 ip = 0;
#define ORC_NEXT() (ip < insize ? (u32)in[ip++] : 0xFFFFFFFFu)
    for (int k = 0; k < 4; k++) code = (code << 8) + ORC_NEXT();
    for (s32 i = 0; i < n; i++) {
        run = (prev1 == prev2) ? run + 1 : 0;
        int runflag = run > 2;
        u32 node = 1;
        while (node < 256) {
            cm_probe q = cm_predict(m, node, prev1, prev2, runflag);
            u32 mid = low + (u32)(((u64)(high - low) * q.p18) >> 18);
            int bit = code <= mid;
            if (bit) high = mid; else low = mid + 1;
            while ((low ^ high) < (1u << 24)) {
                low <<= 8;
                high = (high << 8) | 0xFF;
                code = (code << 8) + ORC_NEXT();
            }
            cm_learn(&q, bit);
            node = node * 2 + (u32)bit;
        }
        prev2 = prev1;
        prev1 = node & 255;
        out[i] = (u8)prev1;
    }
#undef ORC_NEXT
    free(m);
}

/* ------------------------------------------------------------------------------------------
 * Block encode / decode.  Restate bz3_encode_block (src/libbz3.c:585-654) and
 * bz3_decode_block (src/libbz3.c:656-809) including every validation step and error code.
 * `*err` receives the value bz3_last_error() would report afterwards (ORC_OK when the reference
 * leaves last_error untouched on that path; callers start from a fresh state).
 * ------------------------------------------------------------------------------------------ */

/* buffer must hold orc_bound(max(size, 64)) bytes.  Returns new size or -1. */
ORC_API s32 orc_encode_block(u8 * buffer, s32 size, s32 block_size, s32 * err) {
    *err = ORC_OK;
    if (size > block_size) { *err = ORC_ERR_DATA_TOO_BIG; return -1; }
    u32 crc = orc_crc32c(1, buffer, (size_t)size);
    if (size < 64) { /* stored block (:596-601) */
        memmove(buffer + 8, buffer, (size_t)size);
        st_le32(buffer, crc);
        st_le32(buffer + 4, 0xFFFFFFFFu);
        return size + 8;
    }
    size_t cap = orc_bound((size_t)size) + 64;
    u8 * cur = (u8 *)malloc(cap);
    u8 * alt = (u8 *)malloc(cap);
    if (!cur || !alt) { free(cur); free(alt); *err = ORC_ERR_INIT; return -1; }
    memcpy(cur, buffer, (size_t)size);
    s32 n = size, model = 0, lzp_size = 0, rle_size;
    rle_size = orc_mrle_encode(cur, n, alt);
    if (rle_size < n) { u8 * t = cur; cur = alt; alt = t; n = rle_size; model |= 4; } /* :609-614 */
    lzp_size = orc_lzp_encode(cur, n, alt);
    if (lzp_size > 0 && lzp_size < n) { u8 * t = cur; cur = alt; alt = t; n = lzp_size; model |= 2; } /* :616-621 */
    s32 idx = orc_bwt(cur, alt, n);
    if (idx < 0) { free(cur); free(alt); *err = ORC_ERR_BWT; return -1; }
    s32 words = 2 + ((model & 2) != 0) + ((model & 4) != 0);
    s32 coded = orc_cm_encode(alt, n, buffer + words * 4 + 1);
    st_le32(buffer, crc);
    st_le32(buffer + 4, (u32)idx);
    buffer[8] = (u8)model;
    s32 w = 0;
    if (model & 2) st_le32(buffer + 9 + 4 * w++, (u32)lzp_size); /* lzp first, then rle (:646-647) */
    if (model & 4) st_le32(buffer + 9 + 4 * w++, (u32)rle_size);
    free(cur); free(alt);
    return coded + words * 4 + 1;
}

static int sizes_fit(size_t buffer_size, s32 lzp_size, s32 rle_size, s32 orig_size) { /* :114-122 */
    size_t a = lzp_size < 0 ? 0 : (size_t)lzp_size, b = rle_size < 0 ? 0 : (size_t)rle_size,
           c = orig_size < 0 ? 0 : (size_t)orig_size;
    return a <= buffer_size && b <= buffer_size && c <= buffer_size;
}

/* Returns decoded size or -1. */
ORC_API s32 orc_decode_block(u8 * buffer, size_t buffer_size, s32 comp_size, s32 orig_size, s32 block_size, s32 * err) {
    *err = ORC_OK;
    const size_t bound = orc_bound((size_t)block_size);
    if (buffer_size < 9 || buffer_size < (size_t)comp_size) { /* s32 -> size_t like the reference's comparison */ *err = ORC_ERR_DATA_SIZE_TOO_SMALL; return -1; } /* :658 */
    u32 crc = ld_le32(buffer);
    s32 idx = (s32)ld_le32(buffer + 4);
    if (comp_size < 0 || (size_t)comp_size > bound) { *err = ORC_ERR_MALFORMED_HEADER; return -1; } /* :667 */
    if (idx == -1) { /* stored block (:672-692) */
        if (comp_size - 8 > 64 || comp_size < 8) { *err = ORC_ERR_MALFORMED_HEADER; return -1; }
        if ((size_t)(comp_size - 8) > buffer_size) { *err = ORC_ERR_DATA_SIZE_TOO_SMALL; return -1; }
        memmove(buffer, buffer + 8, (size_t)(comp_size - 8));
        if (orc_crc32c(1, buffer, (size_t)(comp_size - 8)) != crc) { *err = ORC_ERR_CRC; return -1; }
        return comp_size - 8;
    }
    s32 model = (int8_t)buffer[8];
    size_t need = 9 + (size_t)((model & 2) * 4) + (size_t)((model & 4) * 4); /* 9/17/25/33 (:697) */
    if (buffer_size < need) { *err = ORC_ERR_DATA_SIZE_TOO_SMALL; return -1; }
    s32 lzp_size = -1, rle_size = -1, w = 0;
    if (model & 2) lzp_size = (s32)ld_le32(buffer + 9 + 4 * w++);
    if (model & 4) rle_size = (s32)ld_le32(buffer + 9 + 4 * w++);
    w += 2;
    comp_size -= w * 4 + 1;
    if (((model & 2) && (lzp_size < 0 || (size_t)lzp_size > bound)) ||
        ((model & 4) && (rle_size < 0 || (size_t)rle_size > bound))) { *err = ORC_ERR_MALFORMED_HEADER; return -1; }
    if (orig_size < 0 || (size_t)orig_size > bound) { *err = ORC_ERR_MALFORMED_HEADER; return -1; }
    s32 n = (model & 2) ? lzp_size : (model & 4) ? rle_size : orig_size; /* :724-729 */
    if (!sizes_fit(buffer_size, lzp_size, rle_size, orig_size)) { *err = ORC_ERR_DATA_SIZE_TOO_SMALL; return -1; }

    u8 * cur = (u8 *)malloc(bound + 64);
    u8 * alt = (u8 *)malloc(bound + 64);
    if (!cur || !alt) { free(cur); free(alt); *err = ORC_ERR_INIT; return -1; }
    s32 result = -1;
    orc_cm_decode(buffer + w * 4 + 1, comp_size, cur, n);
    if (idx > n) { *err = ORC_ERR_MALFORMED_HEADER; goto done; } /* :750 */
    if (orc_unbwt(cur, alt, n, idx) < 0) { *err = ORC_ERR_BWT; goto done; }
    { u8 * t = cur; cur = alt; alt = t; }
    s32 have = n;
    if (model & 2) {
        have = orc_lzp_decode(cur, lzp_size, alt, (s32)bound);
        if (have == -1) { *err = ORC_ERR_CRC; goto done; } /* :769-771 */
        if ((size_t)have > buffer_size) { *err = ORC_ERR_DATA_SIZE_TOO_SMALL; goto done; }
        u8 * t = cur; cur = alt; alt = t;
    }
    if (model & 4) {
        if (orc_mrle_decode(cur, alt, orig_size, have)) { *err = ORC_ERR_CRC; goto done; } /* :785-789 */
        have = orig_size;
        u8 * t = cur; cur = alt; alt = t;
    }
    if (have > block_size || have < 0) { *err = ORC_ERR_MALFORMED_HEADER; goto done; } /* :796 */
    memcpy(buffer, cur, (size_t)have);
    if (orc_crc32c(1, buffer, (size_t)have) != crc) { *err = ORC_ERR_CRC; goto done; }
    result = have;
done:
    free(cur); free(alt);
    return result;
}
