// Offline experiment for the CM decoder's guess-ahead (cm.hip, "sync" decoder): how often would different guesses for the next byte
// be right?  Runs the CM model (the same restatement of src/libbz3.c:333-494 as oracle/bz3_oracle.c) over a file of BWT output and
// counts, per byte, whether it equals: the previous byte ("repeat", the guess the decoder makes today), the model's own greedy
// most-probable path, the last byte that differed from the current run's byte ("MRU2"), the byte that followed the previous byte
// value last time, and unions of those.  CPU only, test/analysis infrastructure, not linked into anything.
//   gcc -O2 -o /tmp/cm_guess_stats tools/cm_guess_stats.c && /tmp/cm_guess_stats <file with BWT output>
// Result on 32 MiB of the bench's text after LZP + BWT (33,155,918 bytes): repeat 64.8 %, greedy MAP 67.2 %, repeat|MRU2 76.2 %,
// repeat|MAP|MRU2 79.0 % -- a second speculative table for MRU2 would turn a third of the wrong guesses into right ones; the
// model's own prediction is barely better than "repeat" and not worth computing.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
typedef uint8_t u8; typedef uint16_t u16; typedef uint32_t u32; typedef int32_t s32; typedef uint64_t u64;
typedef struct { u16 c0[256]; u16 c1[256][256]; u16 c2[512][17]; } model;
static void reset(model * m) {
    for (int i = 0; i < 256; i++) m->c0[i] = 32768;
    for (int i = 0; i < 256; i++) for (int j = 0; j < 256; j++) m->c1[i][j] = 32768;
    for (int r = 0; r < 512; r++) for (int k = 0; k < 17; k++) m->c2[r][k] = (u16)((k << 12) - (k == 16));
}
typedef struct { u16 *a, *b, *lo, *hi; u32 p18; } probe;
static probe predict(model * m, u32 node, u32 p1, u32 p2, int rf) {
    probe q; q.a = &m->c0[node]; q.b = &m->c1[p1][node];
    int p = (((int)*q.a + (int)*q.b) * 7 + 2 * (int)m->c1[p2][node]) >> 4;
    int j = p >> 12; q.lo = &m->c2[2 * node + (u32)rf][j]; q.hi = q.lo + 1;
    int x1 = *q.lo, x2 = *q.hi; int ssep = x1 + (((x2 - x1) * (p & 4095)) >> 12);
    q.p18 = (u32)(ssep * 3 + p); return q;
}
static void learn(const probe * q, int bit) {
    if (bit) { *q->a += (u16)((*q->a ^ 65535) >> 2); *q->b += (u16)((*q->b ^ 65535) >> 4); *q->lo += (u16)((*q->lo ^ 65535) >> 6); *q->hi += (u16)((*q->hi ^ 65535) >> 6); }
    else { *q->a -= *q->a >> 2; *q->b -= *q->b >> 4; *q->lo -= *q->lo >> 6; *q->hi -= *q->hi >> 6; }
}
int main(int argc, char ** argv) {
    FILE * f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    u8 * in = malloc(n); if (fread(in, 1, n, f) != (size_t)n) return 1; fclose(f);
    model * m = malloc(sizeof *m); reset(m);
    u32 prev1 = 0, prev2 = 0, run = 0, mru2 = 0;
    u64 rep = 0, map = 0, map6 = 0, either = 0, rep_or_mru2 = 0, map_or_mru2 = 0, any3 = 0, o1 = 0, rep_or_o1 = 0;
    static u8 succ[256];  // last successor seen after each byte value (an order-1 "what followed this byte last time" guess)
    u64 seg_rep[8] = {0}, seg_map[8] = {0}, seg_n[8] = {0};
    for (long i = 0; i < n; i++) {
        run = (prev1 == prev2) ? run + 1 : 0; int rf = run > 2;
        u32 sym = in[i];
        // greedy MAP descent (no learning)
        u32 node = 1;
        for (int k = 0; k < 8; k++) { probe q = predict(m, node, prev1, prev2, rf); node = node * 2 + (q.p18 >= (1u << 17)); }
        u32 g_map = node & 255;
        // top six bits greedy, last two bits from the repeat guess
        u32 g_map6 = (g_map & 0xFC) | (prev1 & 3);
        u32 g_o1 = succ[prev1];
        int h_rep = sym == prev1, h_map = sym == g_map, h_mru2 = sym == mru2, h_o1 = sym == g_o1;
        rep += h_rep; map += h_map; map6 += sym == g_map6; either += h_rep | h_map; rep_or_mru2 += h_rep | h_mru2; map_or_mru2 += h_map | h_mru2;
        any3 += h_rep | h_map | h_mru2; o1 += h_o1; rep_or_o1 += h_rep | h_o1;
        int s = (int)(i * 8 / n); seg_rep[s] += h_rep; seg_map[s] += h_map; seg_n[s]++;
        // real coding step
        node = 1;
        for (int k = 7; k >= 0; k--) { int bit = (sym >> k) & 1; probe q = predict(m, node, prev1, prev2, rf); learn(&q, bit); node = node * 2 + bit; }
        succ[prev1] = (u8)sym;
        if (sym != prev1) mru2 = prev1;
        prev2 = prev1; prev1 = sym;
    }
    double N = (double)n;
    printf("n=%ld\nrepeat            %.4f\ngreedy MAP        %.4f\nMAP top6 + rep    %.4f\nrepeat|MAP        %.4f\nrepeat|MRU2       %.4f\nMAP|MRU2          %.4f\nrepeat|MAP|MRU2   %.4f\norder-1 successor %.4f\nrepeat|o1succ     %.4f\n",
           n, rep / N, map / N, map6 / N, either / N, rep_or_mru2 / N, map_or_mru2 / N, any3 / N, o1 / N, rep_or_o1 / N);
    for (int s = 0; s < 8; s++) printf("eighth %d: repeat %.4f  MAP %.4f\n", s, seg_rep[s] / (double)seg_n[s], seg_map[s] / (double)seg_n[s]);
    return 0;
}
