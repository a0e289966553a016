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
//    that is fixed once the previous byte is known, so 64 lanes x 4 nodes pre-evaluate every node's
//    18-bit probability; the 8 serial decisions then only pick values out of registers
//    (v_readlane), and the 8 nodes on the decoded path are updated in parallel.
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
// encode: one wave.  Lanes 0..7 own the 8 tree levels of the current byte; the coder state is uniform.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_cm_encode(const u8 * __restrict__ in, u32 n, u8 * __restrict__ out, u32 * __restrict__ out_size) {
    __shared__ CmLds m;
    cm_model_init(m);
    const int lane = lane_id();
    const u32 k = (u32)lane & 7u;
    u32 low = 0, high = 0xFFFFFFFFu, c1 = 0, c2 = 0, run = 0, op = 0;
    for (u32 base = 0; base < n; base += 64) {
        const u32 mine = (base + lane < n) ? in[base + lane] : 0u;
        const u32 cnt = (n - base < 64u) ? n - base : 64u;
        for (u32 t = 0; t < cnt; t++) {
            const u32 c = cm_readlane(mine, (int)t);
            run = (c1 == c2) ? run + 1 : 0;  // :367-372
            const u32 f = run > 2 ? 1u : 0u;
            const u32 node = (1u << k) | (c >> (8 - k));
            const u32 bit = (c >> (7 - k)) & 1u;
            CmProbe q = cm_probe(m, node, c1, c2, f);
            wave_sync();
            if (lane < 8) cm_learn(m, q, node, c1, bit);
            wave_sync();
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                const u32 p18 = cm_readlane(q.p18, kk);
                const u32 b = (c >> (7 - kk)) & 1u;
                const u32 mid = low + (u32)(((u64)(high - low) * p18) >> 18);  // :388, :402
                if (b) high = mid; else low = mid + 1;
                while ((low ^ high) < (1u << 24)) {  // :390-394
                    if (lane == 0) out[op] = (u8)(low >> 24);
                    op++;
                    low <<= 8;
                    high = (high << 8) | 0xFFu;
                }
            }
            c2 = c1;
            c1 = c;
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
// decode: one wave.  Lane l owns tree nodes l, l+64, l+128, l+192.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_cm_decode(const u8 * __restrict__ in, u32 in_size, u8 * __restrict__ out, u32 n) {
    __shared__ CmLds m;
    cm_model_init(m);
    const int lane = lane_id();
    u32 low = 0, high = 0xFFFFFFFFu, code = 0, c1 = 0, c2 = 0, run = 0;
    u32 ip = 0;       // next input byte index
    u32 ibase = 0;    // input window [ibase, ibase + 64) is held one byte per lane
    u32 window = (ibase + lane < in_size) ? in[ibase + lane] : 0xFFFFFFFFu;
#define CM_NEXT_BYTE(dst)                                                              \
    do {                                                                               \
        if (ip - ibase >= 64u) {                                                       \
            ibase += 64u;                                                              \
            window = (ibase + lane < in_size) ? in[ibase + lane] : 0xFFFFFFFFu;        \
        }                                                                              \
        dst = cm_readlane(window, (int)(ip - ibase));                                  \
        ip++;                                                                          \
    } while (0)
    for (int j = 0; j < 4; j++) {  // :438-441; bytes past the end read as -1 (:345)
        u32 b;
        CM_NEXT_BYTE(b);
        code = (code << 8) + b;
    }
    u32 staged = 0;
    for (u32 i = 0; i < n; i++) {
        run = (c1 == c2) ? run + 1 : 0;
        const u32 f = run > 2 ? 1u : 0u;
        // phase 1: probabilities of all 255 nodes
        CmProbe q0 = cm_probe(m, (u32)lane, c1, c2, f);
        CmProbe q1 = cm_probe(m, (u32)lane + 64u, c1, c2, f);
        CmProbe q2 = cm_probe(m, (u32)lane + 128u, c1, c2, f);
        CmProbe q3 = cm_probe(m, (u32)lane + 192u, c1, c2, f);
        // phase 2: 8 serial binary decisions (:453-489)
        u32 ctx = 1;
#pragma unroll
        for (int lvl = 0; lvl < 8; lvl++) {
            u32 p18;
            const int src = (int)(ctx & 63u);
            if (lvl < 6) p18 = cm_readlane(q0.p18, src);
            else if (lvl == 6) p18 = cm_readlane(q1.p18, src);
            else {
                const u32 a = cm_readlane(q2.p18, src), b = cm_readlane(q3.p18, src);
                p18 = (ctx & 64u) ? b : a;
            }
            const u32 mid = low + (u32)(((u64)(high - low) * p18) >> 18);  // :464
            const u32 bit = code <= mid ? 1u : 0u;
            if (bit) high = mid; else low = mid + 1;
            while ((low ^ high) < (1u << 24)) {  // :470-474
                low <<= 8;
                high = (high << 8) | 0xFFu;
                u32 b;
                CM_NEXT_BYTE(b);
                code = (code << 8) + b;
            }
            ctx = ctx * 2 + bit;
        }
        const u32 c = ctx & 255u;
        // phase 3: update the 8 nodes on the decoded path (one per tree level)
        wave_sync();
        {
            const u32 full = 256u | c;
#define CM_UPDATE_IF_ON_PATH(q, node_expr)                                   \
    do {                                                                     \
        const u32 node = (node_expr);                                        \
        if (node != 0) {                                                     \
            const int lvl = 31 - __clz(node);                                \
            if ((full >> (8 - lvl)) == node) cm_learn(m, q, node, c1, (c >> (7 - lvl)) & 1u); \
        }                                                                    \
    } while (0)
            CM_UPDATE_IF_ON_PATH(q0, (u32)lane);
            CM_UPDATE_IF_ON_PATH(q1, (u32)lane + 64u);
            CM_UPDATE_IF_ON_PATH(q2, (u32)lane + 128u);
            CM_UPDATE_IF_ON_PATH(q3, (u32)lane + 192u);
#undef CM_UPDATE_IF_ON_PATH
        }
        wave_sync();
        c2 = c1;
        c1 = c;
        if ((u32)lane == (i & 63u)) staged = c;
        if ((i & 63u) == 63u || i + 1 == n) {
            const u32 first = i & ~63u;
            if (first + lane <= i) out[first + lane] = (u8)staged;
        }
    }
#undef CM_NEXT_BYTE
}

void cm_encode(const u8 * d_in, u32 n, u8 * d_out, u32 * d_out_size, hipStream_t s) {
    launch(k_cm_encode, dim3(1), dim3(64), 0, s, d_in, n, d_out, d_out_size);
}

void cm_decode(const u8 * d_in, u32 in_size, u8 * d_out, u32 n, hipStream_t s) {
    launch(k_cm_decode, dim3(1), dim3(64), 0, s, d_in, in_size, d_out, n);
}

}  // namespace bz3
