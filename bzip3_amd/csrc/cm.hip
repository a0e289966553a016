// cm.hip -- bzip3's context-mixing model + 32-bit binary arithmetic coder, one workgroup per block.
// Replaces begin / encode_bytes / decode_bytes (reference src/libbz3.c:333-494).
//
// The whole model -- C0 u16[256], C1 u16[256][256], C2 u16[512][17] = 148,992 B -- lives in the
// 160 KiB LDS of ONE compute unit for the lifetime of the block, so the 5 table reads and 4 counter
// updates per coded bit never leave the CU; HBM traffic is the algorithmic minimum (read n bytes,
// write the coded bytes, or the reverse).  Blocks are independent, so a batch runs one block per CU.
//
// Exact facts used:
//  * encode: all 8 tree nodes of a byte are known up front (the encoder knows the byte), they touch
//    disjoint counters, so 8 lanes evaluate and update them at once; the (low, high) recurrence of the
//    coder is inherently serial and is kept wave-uniform (scalar registers).
//  * decode: the bit is unknown until decoded, but the 255 nodes of the NEXT byte depend only on state
//    that is fixed once the previous byte is known, so 256 lanes (one tree node each) pre-evaluate every
//    node's 18-bit probability; the 8 serial decisions then only pick values out of registers
//    (v_readlane), and the 8 lanes on the decoded path update their counters in parallel.
// All arithmetic is integer and matches the reference bit for bit, including the signed interpolation
// `x1 + (((x2 - x1) * (p & 4095)) >> 12)` with an arithmetic shift of a possibly negative product.
#include "prims.hpp"
#include "stages.hpp"

namespace bz3 {

constexpr int CM_C2_STRIDE = 17;

struct CmLds {
    u16 c1[256 * 256];
    u16 c0[256];
    u16 c2[512 * CM_C2_STRIDE];
};

__device__ __forceinline__ u32 cm_readlane(u32 v, int lane) {
#ifdef BZ3_EMU
    return __shfl(v, lane);
#else
    return (u32)__builtin_amdgcn_readlane((int)v, lane);
#endif
}

// Tells the compiler a value is wave-uniform so that it lives in scalar registers and branches on it
// are scalar branches (the serial coder recurrences run entirely on the scalar unit).
__device__ __forceinline__ u32 cm_uniform(u32 v) {
#ifdef BZ3_EMU
    return v;
#else
    return (u32)__builtin_amdgcn_readfirstlane((int)v);
#endif
}

__device__ __forceinline__ void cm_model_init(CmLds & m) {  // begin(): :350-358
    for (int i = threadIdx.x; i < 256 * 256; i += blockDim.x) m.c1[i] = 32768;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) m.c0[i] = 32768;
    for (int i = threadIdx.x; i < 512 * CM_C2_STRIDE; i += blockDim.x) {
        const int k = i % CM_C2_STRIDE;
        m.c2[i] = (u16)((k << 12) - (k == 16));
    }
    __syncthreads();
}

struct CmProbe {
    u32 p0, p1;   // C0[node], C1[c1][node]
    u32 x1, x2;   // C2 row cells j, j+1
    u32 c2off;    // index of cell j in c2[]
    u32 p18;      // 18-bit probability of a 1 bit
};

__device__ __forceinline__ CmProbe cm_probe(const CmLds & m, u32 node, u32 c1, u32 c2, u32 f) {  // :377-388
    CmProbe q;
    q.p0 = m.c0[node];
    q.p1 = m.c1[c1 * 256 + node];
    const u32 p2 = m.c1[c2 * 256 + node];
    const int p = (int)(((q.p0 + q.p1) * 7u + 2u * p2) >> 4);
    const int j = p >> 12;
    q.c2off = (2u * node + f) * CM_C2_STRIDE + (u32)j;
    q.x1 = m.c2[q.c2off];
    q.x2 = m.c2[q.c2off + 1];
    const int ssep = (int)q.x1 + ((((int)q.x2 - (int)q.x1) * (p & 4095)) >> 12);
    q.p18 = (u32)(ssep * 3 + p);
    return q;
}

__device__ __forceinline__ void cm_learn(CmLds & m, const CmProbe & q, u32 node, u32 c1, u32 bit) {  // :347-348, :396-399, :411-414
    u32 a = q.p0, b = q.p1, lo = q.x1, hi = q.x2;
    if (bit) {
        a += (a ^ 65535u) >> 2;
        b += (b ^ 65535u) >> 4;
        lo += (lo ^ 65535u) >> 6;
        hi += (hi ^ 65535u) >> 6;
    } else {
        a -= a >> 2;
        b -= b >> 4;
        lo -= lo >> 6;
        hi -= hi >> 6;
    }
    m.c0[node] = (u16)a;
    m.c1[c1 * 256 + node] = (u16)b;
    m.c2[q.c2off] = (u16)lo;
    m.c2[q.c2off + 1] = (u16)hi;
}

// ------------------------------------------------------------------------------------------------
// encode: two waves.  Wave 1 ("model") walks the block: lanes 0..7 own the 8 tree levels of the current
// byte, evaluate their node, update the counters and push the 18-bit probabilities into an LDS ring.
// Wave 0 ("coder") drains the ring and runs the serial (low, range) recurrence in scalar registers.
// The two overlap; the block finishes at the pace of the slower one (the coder: ~12 scalar
// instructions per coded bit).
// ------------------------------------------------------------------------------------------------
constexpr u32 CM_RING = 128;  // bytes of look-ahead between the model wave and the coder wave

__device__ __forceinline__ u32 lds_peek(const u32 * p) { return *reinterpret_cast<const volatile u32 *>(p); }
__device__ __forceinline__ void lds_poke(u32 * p, u32 v) { *reinterpret_cast<volatile u32 *>(p) = v; }

__global__ void __launch_bounds__(128) k_cm_encode(const u8 * __restrict__ in, u32 n, u8 * __restrict__ out, u32 * __restrict__ out_size) {
    __shared__ CmLds m;
    __shared__ u32 ring[CM_RING * 8];
    __shared__ u32 s_prod, s_cons;
    if (threadIdx.x == 0) { s_prod = 0; s_cons = 0; }
    cm_model_init(m);
    const int lane = lane_id();
    if (cm_uniform((u32)wave_id()) == 1) {
        // ---- model wave --------------------------------------------------------------------------
        const u32 k = (u32)lane & 7u;
        u32 c1 = 0, c2 = 0, run = 0, cons_seen = 0;
        for (u32 base = 0; base < n; base += 64) {
            const u32 mine = (base + lane < n) ? in[base + lane] : 0u;
            const u32 cnt = (n - base < 64u) ? n - base : 64u;
            for (u32 t = 0; t < cnt; t++) {
                const u32 i = base + t;
                while (i - cons_seen >= CM_RING) {  // ring full: wait for the coder
                    cons_seen = lds_peek(&s_cons);
                    if (i - cons_seen >= CM_RING) BZ3_SPIN_PAUSE();
                }
                const u32 c = cm_readlane(mine, (int)t);
                run = (c1 == c2) ? run + 1 : 0;  // :367-372
                const u32 f = run > 2 ? 1u : 0u;
                const u32 node = (1u << k) | (c >> (8 - k));
                const u32 bit = (c >> (7 - k)) & 1u;
                CmProbe q = cm_probe(m, node, c1, c2, f);
                wave_sync();
                if (lane < 8) {
                    cm_learn(m, q, node, c1, bit);
                    ring[(i & (CM_RING - 1)) * 8 + k] = q.p18;
                }
                wave_sync();
                c2 = c1;
                c1 = c;
                if ((i & 3u) == 3u || i + 1 == n) {
                    lds_release();
                    if (lane == 0) lds_poke(&s_prod, i + 1);
                }
            }
        }
        return;
    }
    // ---- coder wave: high == low + range throughout (:388-394 in (low, range) form) -----------------
    u32 low = 0, range = 0xFFFFFFFFu, op = 0, prod_seen = 0;
    for (u32 base = 0; base < n; base += 64) {
        const u32 mine = (base + lane < n) ? in[base + lane] : 0u;
        const u32 cnt = (n - base < 64u) ? n - base : 64u;
        for (u32 t = 0; t < cnt; t++) {
            const u32 i = base + t;
            while (prod_seen <= i) {
                prod_seen = lds_peek(&s_prod);
                if (prod_seen <= i) BZ3_SPIN_PAUSE();
            }
            lds_acquire();
            const u32 ev = ring[(i & (CM_RING - 1)) * 8 + ((u32)lane & 7u)];
            const u32 c = cm_readlane(mine, (int)t);
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                const u32 p18 = cm_readlane(ev, kk);
                const u32 tt = (u32)(((u64)range * p18) >> 18);
                if ((c >> (7 - kk)) & 1u) {
                    range = tt;  // high = mid
                } else {
                    low += tt + 1;  // low = mid + 1
                    range -= tt + 1;
                }
                if (range < (1u << 24)) {  // necessary for (low ^ high) < 2^24; the exact test follows
                    while ((low ^ (low + range)) < (1u << 24)) {
                        if (lane == 0) out[op] = (u8)(low >> 24);
                        op++;
                        low <<= 8;
                        range = (range << 8) | 0xFFu;
                    }
                }
            }
            if ((i & 15u) == 15u && lane == 0) lds_poke(&s_cons, i + 1);
        }
    }
    if (lane == 0) {  // flush (:425-432)
        for (int j = 0; j < 4; j++) {
            out[op + j] = (u8)(low >> 24);
            low <<= 8;
        }
        *out_size = op + 4;
    }
}

// ------------------------------------------------------------------------------------------------
// decode: five waves.  Waves 1..4 hold one tree node per lane (node = thread - 64): before every byte
// they evaluate all 255 probabilities into an LDS table; wave 0 loads the table into registers (4 per
// lane), makes the 8 serial decisions with v_readlane + scalar arithmetic, and publishes the byte; the
// 8 lanes whose node lies on the decoded path then update their counters.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(320) k_cm_decode(const u8 * __restrict__ in, u32 in_size, u8 * __restrict__ out, u32 n) {
    __shared__ CmLds m;
    __shared__ u32 ptab[256];
    __shared__ u32 s_byte;
    cm_model_init(m);
    const int lane = lane_id();
    const bool coder = cm_uniform((u32)wave_id()) == 0;
    const u32 node = threadIdx.x - 64u;  // model lanes only
    u32 low = 0, range = 0xFFFFFFFFu, x = 0, c1 = 0, c2 = 0, run = 0;  // x = code - low
    u32 ip = 0, ibase = 0;
    u32 window = (coder && ibase + lane < in_size) ? in[ibase + lane] : 0xFFFFFFFFu;
#define CM_NEXT_BYTE(dst)                                                              \
    do {                                                                               \
        if (ip - ibase >= 64u) {                                                       \
            ibase += 64u;                                                              \
            window = (ibase + lane < in_size) ? in[ibase + lane] : 0xFFFFFFFFu;        \
        }                                                                              \
        dst = cm_readlane(window, (int)(ip - ibase));                                  \
        ip++;                                                                          \
    } while (0)
    if (coder) {
        for (int j = 0; j < 4; j++) {  // :438-441; bytes past the end read as -1 (:345)
            u32 b;
            CM_NEXT_BYTE(b);
            x = (x << 8) + b;
        }
    }
    u32 staged = 0;
    CmProbe q;
    q.p0 = q.p1 = q.x1 = q.x2 = q.c2off = q.p18 = 0;
    for (u32 i = 0; i < n; i++) {
        run = (c1 == c2) ? run + 1 : 0;
        const u32 f = run > 2 ? 1u : 0u;
        if (!coder) {
            q = cm_probe(m, node, c1, c2, f);
            ptab[node] = q.p18;
        }
        __syncthreads();
        if (coder) {
            const u32 p0 = ptab[lane], p1 = ptab[lane + 64], p2 = ptab[lane + 128], p3 = ptab[lane + 192];
            u32 ctx = 1;
#pragma unroll
            for (int lvl = 0; lvl < 8; lvl++) {  // :453-489
                u32 p18;
                const int src = (int)(ctx & 63u);
                if (lvl < 6) p18 = cm_readlane(p0, src);
                else if (lvl == 6) p18 = cm_readlane(p1, src);
                else {
                    const u32 a = cm_readlane(p2, src), b = cm_readlane(p3, src);
                    p18 = (ctx & 64u) ? b : a;
                }
                const u32 tt = (u32)(((u64)range * p18) >> 18);  // mid - low (:464)
                const bool bit = x <= tt;                        // code <= mid
#ifdef BZ3_EMU
                ctx = ctx * 2 + (bit ? 1u : 0u);
#else
                // ctx = 2*ctx + bit on the scalar unit (keeps the node index out of vector registers)
                asm volatile("s_cmp_le_u32 %1, %2\n\ts_addc_u32 %0, %0, %0" : "+s"(ctx) : "s"(x), "s"(tt) : "scc");
#endif
                if (bit) {
                    range = tt;
                } else {
                    low += tt + 1;
                    x -= tt + 1;
                    range -= tt + 1;
                }
                if (range < (1u << 24)) {
                    while ((low ^ (low + range)) < (1u << 24)) {  // :470-474
                        low <<= 8;
                        range = (range << 8) | 0xFFu;
                        u32 b;
                        CM_NEXT_BYTE(b);
                        x = (x << 8) + b;
                    }
                }
            }
            const u32 c = ctx & 255u;
            if (lane == 0) s_byte = c;
            if ((u32)lane == (i & 63u)) staged = c;
            if ((i & 63u) == 63u || i + 1 == n) {
                const u32 first = i & ~63u;
                if (first + lane <= i) out[first + lane] = (u8)staged;
            }
        }
        __syncthreads();
        const u32 c = cm_uniform(s_byte);
        if (!coder && node != 0) {
            const int lvl = 31 - __clz((int)node);
            if (((256u | c) >> (8 - lvl)) == node) cm_learn(m, q, node, c1, (c >> (7 - lvl)) & 1u);
        }
        c2 = c1;
        c1 = c;
    }
#undef CM_NEXT_BYTE
}

void cm_encode(const u8 * d_in, u32 n, u8 * d_out, u32 * d_out_size, hipStream_t s) {
    launch(k_cm_encode, dim3(1), dim3(128), 0, s, d_in, n, d_out, d_out_size);
}

void cm_decode(const u8 * d_in, u32 in_size, u8 * d_out, u32 n, hipStream_t s) {
    launch(k_cm_decode, dim3(1), dim3(320), 0, s, d_in, in_size, d_out, n);
}

}  // namespace bz3
