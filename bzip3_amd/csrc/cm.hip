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
// byte, evaluate their node, update the counters and push one 16-byte event per coded bit into an LDS ring.
// Wave 0 ("coder") drains the ring and runs the serial range recurrence.
//
// Coder formulation (exact).  With high == low + range, the reference's update
//     mid = low + ((range * P) >> 18);   bit ? high = mid : low = mid + 1                  (:388, :402)
// is   bit = 1:  range' = (range * P) >> 18                       low' = low
//      bit = 0:  range' = (range * (2^18 - P) - 1) >> 18          low' = low + (range - range')
// (range - ((range*P)>>18) - 1 == ((range*(2^18-P)) - 1) >> 18 for every range, P < 2^18), so the model wave
// ships (-s, -s, M, s) with M = P or 2^18 - P and the coder needs one v_mad_u64_u32, one 64-bit shift, one
// v_sub and one v_mad per bit -- no bit test, no selects.  The coder state is deliberately kept in VECTOR
// registers (seeded through an opaque v_mov): the events arrive in VGPRs from LDS broadcast reads, and moving
// them to the scalar unit would cost a v_readfirstlane per operand.  A single wave issues one instruction every ~5.4
// cycles (8.5 when dependent; profiles/r01_ubench_single_wave.txt), so instruction count is the currency.
// ------------------------------------------------------------------------------------------------
constexpr u32 CM_RING = 64;  // bytes of look-ahead between the model wave and the coder wave (8 KiB of LDS)

__device__ __forceinline__ u32 lds_peek(const u32 * p) { return *reinterpret_cast<const volatile u32 *>(p); }
__device__ __forceinline__ void lds_poke(u32 * p, u32 v) { *reinterpret_cast<volatile u32 *>(p) = v; }

__global__ void __launch_bounds__(128) k_cm_encode(const CmEncodeJob * __restrict__ jobs) {
    // one workgroup per block: blockIdx.x selects the job
    const u8 * __restrict__ in = jobs[blockIdx.x].in;
    const u32 n = jobs[blockIdx.x].n;
    u8 * __restrict__ out = jobs[blockIdx.x].out;
    u32 * __restrict__ out_size = jobs[blockIdx.x].out_size;
    __shared__ CmLds m;
    __shared__ uint4 ring[CM_RING * 8];
    __shared__ u32 s_prod, s_cons;
    if (threadIdx.x == 0) { s_prod = 0; s_cons = 0; }
    cm_model_init(m);
    const int lane = lane_id();
    const u32 * __restrict__ in32 = reinterpret_cast<const u32 *>(in);  // block buffers are 256-byte aligned
    if (cm_uniform((u32)wave_id()) == 1) {
        // ---- model wave --------------------------------------------------------------------------
        const u32 k = (u32)lane & 7u;
        u32 c1 = 0, c2 = 0, run = 0, cons_seen = 0, word = 0;
        for (u32 i = 0; i < n; i++) {
            while (i - cons_seen >= CM_RING) {  // ring full: wait for the coder
                cons_seen = lds_peek(&s_cons);
                if (i - cons_seen >= CM_RING) BZ3_SPIN_PAUSE();
            }
            if ((i & 3u) == 0) word = cm_uniform(in32[i >> 2]);  // wave-uniform address: a scalar load
            const u32 c = (word >> ((i & 3u) * 8u)) & 0xFFu;
            run = (c1 == c2) ? run + 1 : 0;  // :367-372
            const u32 f = run > 2 ? 1u : 0u;
            const u32 node = (1u << k) | (c >> (8 - k));
            const u32 bit = (c >> (7 - k)) & 1u;
            CmProbe q = cm_probe(m, node, c1, c2, f);
            wave_sync();
            if (lane < 8) {
                cm_learn(m, q, node, c1, bit);
                const u32 neg = bit ? 0u : 0xFFFFFFFFu;
                ring[(i & (CM_RING - 1)) * 8 + k] = make_uint4(neg, neg, bit ? q.p18 : (1u << 18) - q.p18, bit ^ 1u);
            }
            wave_sync();
            c2 = c1;
            c1 = c;
            if ((i & 3u) == 3u || i + 1 == n) {
                lds_release();
                if (lane == 0) lds_poke(&s_prod, i + 1);
            }
        }
        return;
    }
    // ---- coder wave (every lane carries the same state; lane 0 stores) ---------------------------------
    u32 vzero;
#ifdef BZ3_EMU
    vzero = 0;
#else
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));  // opaque zero: keeps the recurrence on the vector ALU
#endif
    u32 range = 0xFFFFFFFFu ^ vzero, op = 0, prod_seen = 0;
    u64 low = vzero;  // only the low 32 bits are meaningful
    for (u32 i = 0; i < n; i++) {
        while (prod_seen <= i) {
            prod_seen = lds_peek(&s_prod);
            if (prod_seen <= i) BZ3_SPIN_PAUSE();
        }
        lds_acquire();
        const uint4 * __restrict__ evp = &ring[(i & (CM_RING - 1)) * 8];
        uint4 ev[8];
#pragma unroll
        for (int kk = 0; kk < 8; kk++) ev[kk] = evp[kk];  // same address in every lane: LDS broadcast reads
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
            const uint4 e = ev[kk];
            const u64 prod = (u64)range * e.z + (((u64)e.y << 32) | e.x);
            const u32 r2 = (u32)(prod >> 18);
            low += (u64)(range - r2) * e.w;
            range = r2;
            if (__ballot(range < (1u << 24)) != 0ull) {  // necessary for (low ^ high) < 2^24; the exact test follows
                u32 lo32 = (u32)low;
                while (__ballot((lo32 ^ (lo32 + range)) < (1u << 24)) != 0ull) {  // :390-394
                    if (lane == 0) out[op] = (u8)(lo32 >> 24);
                    op++;
                    lo32 <<= 8;
                    range = (range << 8) | 0xFFu;
                }
                low = lo32;
            }
        }
        if ((i & 15u) == 15u && lane == 0) lds_poke(&s_cons, i + 1);
    }
    if (lane == 0) {  // flush (:425-432)
        u32 lo32 = (u32)low;
        for (int j = 0; j < 4; j++) {
            out[op + j] = (u8)(lo32 >> 24);
            lo32 <<= 8;
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
__global__ void __launch_bounds__(320) k_cm_decode(const CmDecodeJob * __restrict__ jobs) {
    const u8 * __restrict__ in = jobs[blockIdx.x].in;
    const u32 in_size = jobs[blockIdx.x].in_size;
    u8 * __restrict__ out = jobs[blockIdx.x].out;
    const u32 n = jobs[blockIdx.x].n;
    __shared__ CmLds m;
    __shared__ u32 ptab[256];
    __shared__ u32 s_byte;
    cm_model_init(m);
    const int lane = lane_id();
    const bool coder = cm_uniform((u32)wave_id()) == 0;
    const u32 node = threadIdx.x - 64u;  // model lanes only
    u32 low = 0, range = 0xFFFFFFFFu, code = 0, c1 = 0, c2 = 0, run = 0;
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
            code = (code << 8) + b;
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
                const u32 mid = low + (u32)(((u64)range * p18) >> 18);  // :464 (high == low + range)
                // NB: the comparison must be on absolute values: a truncated stream feeds -1 bytes (:345) and
                // can push `code` below `low`, where the reference still decodes a 1.
                const bool bit = code <= mid;
#ifdef BZ3_EMU
                ctx = ctx * 2 + (bit ? 1u : 0u);
#else
                // ctx = 2*ctx + bit on the scalar unit (keeps the node index out of vector registers)
                asm volatile("s_cmp_le_u32 %1, %2\n\ts_addc_u32 %0, %0, %0" : "+s"(ctx) : "s"(code), "s"(mid) : "scc");
#endif
                if (bit) {
                    range = mid - low;
                } else {
                    range -= mid - low + 1;
                    low = mid + 1;
                }
                if (range < (1u << 24)) {
                    while ((low ^ (low + range)) < (1u << 24)) {  // :470-474
                        low <<= 8;
                        range = (range << 8) | 0xFFu;
                        u32 b;
                        CM_NEXT_BYTE(b);
                        code = (code << 8) + b;
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

void cm_encode_batch(const CmEncodeJob * d_jobs, u32 njobs, hipStream_t s) {
    if (njobs) launch(k_cm_encode, dim3(njobs), dim3(128), 0, s, d_jobs);
}

void cm_decode_batch(const CmDecodeJob * d_jobs, u32 njobs, hipStream_t s) {
    if (njobs) launch(k_cm_decode, dim3(njobs), dim3(320), 0, s, d_jobs);
}

}  // namespace bz3
