// Decoder self-attention core: out = dropout(softmax(q k^T * scale)) v for head_dim 32, fp32, sequences of at most 320
// (models/arctic_transformer.py:351,374-376: nn.MultiheadAttention over the 300 queries, 8 heads of 32, batch = frames).
// The vendor's fused path spends 77 + 112 + 100 us per decoder layer on it at 32 frames (attn_fwd, bwd_kernel_dk_dv,
// bwd_kernel_dq, profiles/r05_layers_kernel_stats_fused.csv) — for 256 independent 300 x 300 x 32 problems that fit a CU's LDS.
//
// All three kernels: a workgroup of 12 wavefronts (three per SIMD; with 10 two SIMDs carry three and two carry two: 201 us forward +
// backward against 167 with 12 or 16, 174 with 8) owns one (batch, head) pair — its 16-row tiles dealt to the wavefronts, up to two
// each at 300 rows (two workgroups per pair re-load the pair's operands and measured 8 % slower: MSDA_ATTN_HALVES);
// the pair's other operand(s) sit in LDS row-major ([rows][36 floats]: 16-byte aligned rows, the 16 lanes of a ds_read_b128 on
// different bank groups).  Every product runs on v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate: the arithmetic of an fmaf
// chain), and the score tile never changes layout between the two products it takes part in:
//   * forward and dQ work on S^T tiles (rows = keys, columns = queries).  The accumulator of a 16 x 16 tile gives lane
//     (c = lane & 15, r = lane >> 4) the entries [4r .. 4r+3][c]; the following product sums over KEYS, and a sum does not care
//     in which order its terms arrive: MFMA step v of a tile takes "k index r" to mean key 4r + v, so the B operand is the
//     accumulator register v as it stands and the A operand is row 4r + v of V (or K) in LDS.  No transposition through LDS,
//     no shuffles.
//   * dK / dV work on S tiles (rows = queries, columns = keys) and sum over QUERIES the same way.
//   * the head dimension is relabelled likewise (step (half, v): k index r = channel 16 half + 4r + v), so the operand whose
//     row index is the lane's column reads four steps with one ds_read_b128 / one float4 global load.
// Softmax statistics: the forward keeps log-sum-exp per (pair, query); the backward recomputes the probabilities from it.
// Dropout: keep(seed, pair, query, key) is a 32-bit integer hash (at_hash) compared with p * 2^32 — recomputed in the backward, no mask
// tensor; the seed is READ FROM DEVICE MEMORY (the caller draws it with torch's generator: reproducible under manual_seed,
// safe under HIP-graph capture).  This is the kernel's own random stream, not nn.functional.dropout's.
#include <math.h>

#include "msda_common.h"
#include "msda_launch.h"

namespace msda {

#ifndef MSDA_AT_WAVES
#define MSDA_AT_WAVES 12
#endif
constexpr int kAtRow = 36, kAtWaves = MSDA_AT_WAVES, kAtBlock = kAtWaves * 64, kAtMaxTiles = 20, kAtMaxLen = 16 * kAtMaxTiles;
using at_f4 = __attribute__((ext_vector_type(4))) float;

struct AtView { float *p; long long sn, sl; };          // element (n, h, l, d) at p + n*sn + h*32 + l*sl + d

// 32 bits per (pair, query, key): the counter (query << 16) + key + mix — mix = seed and pair, uniform — through the two
// multiply-xorshift rounds of the usual 32-bit finalizer; its last xor-shift is left out (it only folds high bits into low ones,
// and the result is compared with a threshold).  v_mul_lo_u32 is a quarter-rate instruction and the vector unit is as busy as the
// matrix pipe in these kernels: a lane adds its part of the counter once per tile, an element one constant.
__device__ __forceinline__ unsigned at_mix(unsigned long long seed, unsigned pair)
{
    return (unsigned)seed + pair * 0x9E3779B1u + (unsigned)(seed >> 32) * 0x85EBCA6Bu;
}
__device__ __forceinline__ unsigned at_hash(unsigned counter)
{
    unsigned x = counter;
    x ^= x >> 16; x *= 0x85EBCA6Bu;
    x ^= x >> 13; x *= 0xC2B2AE35u;
    return x;
}

__device__ __forceinline__ float4 at_ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 at_zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// rows [0, L) of two [L][32] slices (row strides sl0 / sl1) -> dst0 / dst1 [Lp][kAtRow]; rows [L, Lp) zero.  Lp * 8 float4 per
// slice on 640 threads: at most four rounds (Lp <= 320); all the loads of a thread are issued before its first LDS store.
__device__ __forceinline__ void at_load_rows2(float *dst0, const float *src0, long long sl0, float *dst1, const float *src1,
                                              long long sl1, int L, int Lp)
{
    constexpr int R = (kAtMaxLen * 8 + kAtBlock - 1) / kAtBlock;
    float4 v0[R], v1[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int i = threadIdx.x + k * kAtBlock, row = i >> 3, c = (i & 7) * 4;
        const bool live = row < L;
        v0[k] = live ? at_ld4(src0 + (long long)row * sl0 + c) : at_zero4();
        v1[k] = live ? at_ld4(src1 + (long long)row * sl1 + c) : at_zero4();
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int i = threadIdx.x + k * kAtBlock, row = i >> 3, c = (i & 7) * 4;
        if (row < Lp) {
            *reinterpret_cast<float4 *>(dst0 + row * kAtRow + c) = v0[k];
            *reinterpret_cast<float4 *>(dst1 + row * kAtRow + c) = v1[k];
        }
    }
}

#define AT_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// acc += A-rows(tile) . breg over the 32 channels: `arow` = LDS row of this lane's A row (channels 4r.. of each half at +0, +16)
__device__ __forceinline__ at_f4 at_dot32(const float *arow, const float4 &b0, const float4 &b1)
{
    const float4 a0 = at_ld4(arow), a1 = at_ld4(arow + 16);
    at_f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = AT_MFMA(a0.x, b0.x, acc); acc = AT_MFMA(a0.y, b0.y, acc); acc = AT_MFMA(a0.z, b0.z, acc); acc = AT_MFMA(a0.w, b0.w, acc);
    acc = AT_MFMA(a1.x, b1.x, acc); acc = AT_MFMA(a1.y, b1.y, acc); acc = AT_MFMA(a1.z, b1.z, acc); acc = AT_MFMA(a1.w, b1.w, acc);
    return acc;                                      // (one chain: the sum's association is the channel order)
}

// the same for this tile and the next one (16 rows further), the two accumulators' chains interleaved
__device__ __forceinline__ void at_dot32x2(const float *arow, const float4 &b0, const float4 &b1, at_f4 &acc0, at_f4 &acc1)
{
    const float4 a0 = at_ld4(arow), a1 = at_ld4(arow + 16), c0 = at_ld4(arow + 16 * kAtRow), c1 = at_ld4(arow + 16 * kAtRow + 16);
    at_f4 x = {0.f, 0.f, 0.f, 0.f}, y = x;
    x = AT_MFMA(a0.x, b0.x, x); y = AT_MFMA(c0.x, b0.x, y); x = AT_MFMA(a0.y, b0.y, x); y = AT_MFMA(c0.y, b0.y, y);
    x = AT_MFMA(a0.z, b0.z, x); y = AT_MFMA(c0.z, b0.z, y); x = AT_MFMA(a0.w, b0.w, x); y = AT_MFMA(c0.w, b0.w, y);
    x = AT_MFMA(a1.x, b1.x, x); y = AT_MFMA(c1.x, b1.x, y); x = AT_MFMA(a1.y, b1.y, x); y = AT_MFMA(c1.y, b1.y, y);
    x = AT_MFMA(a1.z, b1.z, x); y = AT_MFMA(c1.z, b1.z, y); x = AT_MFMA(a1.w, b1.w, x); y = AT_MFMA(c1.w, b1.w, y);
    acc0 = x; acc1 = y;
}

// o[half] += X^T-rows . w over the tile's 16 rows: X = LDS tile base (row 16t), w = accumulator-layout weights of this lane
__device__ __forceinline__ void at_accum_t(const float *xt, int r, int c, const at_f4 &w, at_f4 &o0, at_f4 &o1)
{
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const float *row = xt + (4 * r + v) * kAtRow + c;
        o0 = AT_MFMA(row[0], w[v], o0);
        o1 = AT_MFMA(row[16], w[v], o1);
    }
}

__device__ __forceinline__ float at_rsum(float x)               // over the four lanes c, c + 16, c + 32, c + 48
{
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    return x;
}
__device__ __forceinline__ float at_rmax(float x)
{
    x = fmaxf(x, __shfl_xor(x, 16));
    x = fmaxf(x, __shfl_xor(x, 32));
    return x;
}

struct AtArgs {
    AtView q, k, v, o, go, gq, gk, gv;
    float *lse;                         // [N*H][Lq]
    const unsigned long long *seed;     // device; null when thresh == 0
    int H, Lq, Lk, halves;
    float scale, scale2, keep_scale;    // scale2 = scale * log2(e): scores are kept in base 2; keep_scale = 1 / (1 - p)
    unsigned thresh;                    // keep iff hash >= thresh (p * 2^32; 0: no dropout)
};

__global__ __launch_bounds__(kAtBlock) void attn32_fwd_kernel(const AtArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float at_smem[];
    const int Lkp = (a.Lk + 15) & ~15, ntk = Lkp >> 4, ntq = (a.Lq + 15) >> 4;
    float *Ks = at_smem, *Vs = at_smem + Lkp * kAtRow;
    const int pair = (int)blockIdx.x / a.halves, part = (int)blockIdx.x % a.halves, n = pair / a.H, h = pair % a.H;
    at_load_rows2(Ks, a.k.p + n * a.k.sn + h * 32, a.k.sl, Vs, a.v.p + n * a.v.sn + h * 32, a.v.sl, a.Lk, Lkp);
    const unsigned mix = at_mix(a.thresh ? a.seed[0] : 0ull, (unsigned)pair);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, r = lane >> 4;
    const int per = (ntq + a.halves - 1) / a.halves, t_end = min(ntq, (part + 1) * per);
    const float *qp = a.q.p + n * a.q.sn + h * 32;
    float *op = a.o.p + n * a.o.sn + h * 32;
    for (int tq = part * per + wave; tq < t_end; tq += kAtWaves) {
        const int qi = tq * 16 + c;
        const bool qok = qi < a.Lq;
        float4 q0 = at_zero4(), q1 = at_zero4();
        if (qok) { q0 = at_ld4(qp + (long long)qi * a.q.sl + 4 * r); q1 = at_ld4(qp + (long long)qi * a.q.sl + 16 + 4 * r); }
        q0.x *= a.scale2; q0.y *= a.scale2; q0.z *= a.scale2; q0.w *= a.scale2;
        q1.x *= a.scale2; q1.y *= a.scale2; q1.z *= a.scale2; q1.w *= a.scale2;
        // S^T tiles: s[t][v] = score (times log2 e) of key 16 t + 4 r + v for query qi
        at_f4 s[kAtMaxTiles];
        float m = -INFINITY;
        // (all the products first — independent chains, the matrix pipe runs them back to back — then the row maximum)
#pragma unroll
        for (int t = 0; t < kAtMaxTiles; t += 2) {               // two tiles' chains interleaved (a dependent MFMA waits 40 cycles, an issue takes 32)
            if (t + 1 < ntk) at_dot32x2(Ks + (16 * t + c) * kAtRow + 4 * r, q0, q1, s[t], s[t + 1]);
            else if (t < ntk) s[t] = at_dot32(Ks + (16 * t + c) * kAtRow + 4 * r, q0, q1);
        }
#pragma unroll
        for (int t = 0; t < kAtMaxTiles; ++t) {
            if (t < ntk) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    if (t == ntk - 1) s[t][v] = 16 * t + 4 * r + v < a.Lk ? s[t][v] : -INFINITY;      // (only the last tile has padding keys)
                    m = fmaxf(m, s[t][v]);
                }
            }
        }
        m = at_rmax(m);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < kAtMaxTiles; ++t) {
            if (t < ntk) {
#pragma unroll
                for (int v = 0; v < 4; ++v) { s[t][v] = __builtin_amdgcn_exp2f(s[t][v] - m); sum += s[t][v]; }
            }
        }
        sum = at_rsum(sum);
        if (qok && r == 0) a.lse[(long long)pair * a.Lq + qi] = (m + __builtin_amdgcn_logf(sum)) * 0.6931471805599453f;   // natural log
        const float inv = a.keep_scale / sum;
        at_f4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
        const unsigned ctr = ((unsigned)qi << 16) + 4 * r + mix;
#pragma unroll
        for (int t = 0; t < kAtMaxTiles; ++t) {
            if (t < ntk) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const bool keep = at_hash(ctr + (16 * t + v)) >= a.thresh;      // (thresh 0: always)
                    s[t][v] = keep ? s[t][v] * inv : 0.f;
                }
                at_accum_t(Vs + 16 * t * kAtRow, r, c, s[t], o0, o1);       // O^T[channel][query] += V^T . P
            }
        }
        if (qok) {
            float *orow = op + (long long)qi * a.o.sl + 4 * r;
            *reinterpret_cast<float4 *>(orow) = make_float4(o0[0], o0[1], o0[2], o0[3]);
            *reinterpret_cast<float4 *>(orow + 16) = make_float4(o1[0], o1[1], o1[2], o1[3]);
        }
    }
}

// dK, dV: a wavefront owns 16 keys and walks the query tiles
__global__ __launch_bounds__(kAtBlock) void attn32_bwd_kv_kernel(const AtArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float at_smem[];
    const int Lqp = (a.Lq + 15) & ~15, ntq = Lqp >> 4, ntk = (a.Lk + 15) >> 4;
    float *Qs = at_smem, *Gs = at_smem + Lqp * kAtRow, *lse_s = Gs + Lqp * kAtRow, *del_s = lse_s + Lqp;
    const int pair = (int)blockIdx.x / a.halves, part = (int)blockIdx.x % a.halves, n = pair / a.H, h = pair % a.H;
    at_load_rows2(Qs, a.q.p + n * a.q.sn + h * 32, a.q.sl, Gs, a.go.p + n * a.go.sn + h * 32, a.go.sl, a.Lq, Lqp);
    const unsigned mix = at_mix(a.thresh ? a.seed[0] : 0ull, (unsigned)pair);
    __syncthreads();
    // per query: log-sum-exp (+inf for the padding rows: their probabilities vanish) and delta = <dO, O>
    for (int qi = threadIdx.x; qi < Lqp; qi += kAtBlock) {
        float d = 0.f;
        if (qi < a.Lq) {
            const float *orow = a.o.p + n * a.o.sn + h * 32 + (long long)qi * a.o.sl, *grow = Gs + qi * kAtRow;
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                const float4 ov = at_ld4(orow + 4 * k4), gv = at_ld4(grow + 4 * k4);
                d += ov.x * gv.x + ov.y * gv.y + ov.z * gv.z + ov.w * gv.w;
            }
        }
        del_s[qi] = d;
        lse_s[qi] = qi < a.Lq ? a.lse[(long long)pair * a.Lq + qi] * 1.4426950408889634f : INFINITY;      // base 2, like the scores
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, r = lane >> 4;
    const int per = (ntk + a.halves - 1) / a.halves, t_end = min(ntk, (part + 1) * per);
    for (int tk = part * per + wave; tk < t_end; tk += kAtWaves) {
        const int key = tk * 16 + c;
        const bool kok = key < a.Lk;
        float4 k0 = at_zero4(), k1 = at_zero4(), v0 = at_zero4(), v1 = at_zero4();
        if (kok) {
            const float *kr = a.k.p + n * a.k.sn + h * 32 + (long long)key * a.k.sl + 4 * r;
            const float *vr = a.v.p + n * a.v.sn + h * 32 + (long long)key * a.v.sl + 4 * r;
            k0 = at_ld4(kr); k1 = at_ld4(kr + 16); v0 = at_ld4(vr); v1 = at_ld4(vr + 16);
        }
        k0.x *= a.scale2; k0.y *= a.scale2; k0.z *= a.scale2; k0.w *= a.scale2;
        k1.x *= a.scale2; k1.y *= a.scale2; k1.z *= a.scale2; k1.w *= a.scale2;
        const unsigned ctr = (unsigned)key + ((unsigned)(4 * r) << 16) + mix;
        at_f4 dv0 = {0.f, 0.f, 0.f, 0.f}, dv1 = dv0, dk0 = dv0, dk1 = dv0;
        // S and dP tiles: entry v = (query 16 t + 4 r + v, key).  The next tile's two products are issued before this tile's
        // exponentials and hashes: the matrix pipe works through them while the vector unit is busy.
        at_f4 s = at_dot32(Qs + c * kAtRow + 4 * r, k0, k1), dp = at_dot32(Gs + c * kAtRow + 4 * r, v0, v1);
#pragma unroll 1
        for (int t = 0; t < ntq; ++t) {
            const int tn = min(t + 1, ntq - 1);
            const at_f4 s_next = at_dot32(Qs + (16 * tn + c) * kAtRow + 4 * r, k0, k1);
            const at_f4 dp_next = at_dot32(Gs + (16 * tn + c) * kAtRow + 4 * r, v0, v1);
            __builtin_amdgcn_sched_barrier(0);
            const float4 ls = at_ld4(lse_s + 16 * t + 4 * r), dl = at_ld4(del_s + 16 * t + 4 * r);
            const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, dlv[4] = {dl.x, dl.y, dl.z, dl.w};
            at_f4 pd, ds;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float p = __builtin_amdgcn_exp2f(s[v] - lsv[v]);
                const bool keep = at_hash(ctr + ((unsigned)(16 * t + v) << 16)) >= a.thresh;
                pd[v] = keep ? p * a.keep_scale : 0.f;
                ds[v] = p * ((keep ? dp[v] * a.keep_scale : 0.f) - dlv[v]);
            }
            at_accum_t(Gs + 16 * t * kAtRow, r, c, pd, dv0, dv1);           // dV^T[channel][key] += dO^T . P_drop
            at_accum_t(Qs + 16 * t * kAtRow, r, c, ds, dk0, dk1);           // dK^T[channel][key] += Q^T . dS
            s = s_next; dp = dp_next;
        }
        if (kok) {
            float *gvr = a.gv.p + n * a.gv.sn + h * 32 + (long long)key * a.gv.sl + 4 * r;
            float *gkr = a.gk.p + n * a.gk.sn + h * 32 + (long long)key * a.gk.sl + 4 * r;
            *reinterpret_cast<float4 *>(gvr) = make_float4(dv0[0], dv0[1], dv0[2], dv0[3]);
            *reinterpret_cast<float4 *>(gvr + 16) = make_float4(dv1[0], dv1[1], dv1[2], dv1[3]);
            *reinterpret_cast<float4 *>(gkr) = make_float4(dk0[0] * a.scale, dk0[1] * a.scale, dk0[2] * a.scale, dk0[3] * a.scale);
            *reinterpret_cast<float4 *>(gkr + 16) = make_float4(dk1[0] * a.scale, dk1[1] * a.scale, dk1[2] * a.scale, dk1[3] * a.scale);
        }
    }
}

// dQ: a wavefront owns 16 queries and walks the key tiles (S^T tiles, as the forward)
__global__ __launch_bounds__(kAtBlock) void attn32_bwd_q_kernel(const AtArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float at_smem[];
    const int Lkp = (a.Lk + 15) & ~15, ntk = Lkp >> 4, ntq = (a.Lq + 15) >> 4;
    float *Ks = at_smem, *Vs = at_smem + Lkp * kAtRow;
    const int pair = (int)blockIdx.x / a.halves, part = (int)blockIdx.x % a.halves, n = pair / a.H, h = pair % a.H;
    at_load_rows2(Ks, a.k.p + n * a.k.sn + h * 32, a.k.sl, Vs, a.v.p + n * a.v.sn + h * 32, a.v.sl, a.Lk, Lkp);
    const unsigned mix = at_mix(a.thresh ? a.seed[0] : 0ull, (unsigned)pair);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, r = lane >> 4;
    const int per = (ntq + a.halves - 1) / a.halves, t_end = min(ntq, (part + 1) * per);
    for (int tq = part * per + wave; tq < t_end; tq += kAtWaves) {
        const int qi = tq * 16 + c;
        const bool qok = qi < a.Lq;
        float4 q0 = at_zero4(), q1 = at_zero4(), g0 = at_zero4(), g1 = at_zero4(), o0 = at_zero4(), o1 = at_zero4();
        float lse = INFINITY;
        if (qok) {
            const float *qr = a.q.p + n * a.q.sn + h * 32 + (long long)qi * a.q.sl + 4 * r;
            const float *gr = a.go.p + n * a.go.sn + h * 32 + (long long)qi * a.go.sl + 4 * r;
            const float *orw = a.o.p + n * a.o.sn + h * 32 + (long long)qi * a.o.sl + 4 * r;
            q0 = at_ld4(qr); q1 = at_ld4(qr + 16); g0 = at_ld4(gr); g1 = at_ld4(gr + 16); o0 = at_ld4(orw); o1 = at_ld4(orw + 16);
            lse = a.lse[(long long)pair * a.Lq + qi] * 1.4426950408889634f;
        }
        q0.x *= a.scale2; q0.y *= a.scale2; q0.z *= a.scale2; q0.w *= a.scale2;
        q1.x *= a.scale2; q1.y *= a.scale2; q1.z *= a.scale2; q1.w *= a.scale2;
        const unsigned ctr = ((unsigned)qi << 16) + 4 * r + mix;
        const float delta = at_rsum(g0.x * o0.x + g0.y * o0.y + g0.z * o0.z + g0.w * o0.w + g1.x * o1.x + g1.y * o1.y + g1.z * o1.z +
                                    g1.w * o1.w);
        at_f4 dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = dq0;
        // S^T and dP^T tiles: entry v = (key 16 t + 4 r + v, query qi); the next tile's products issued ahead, as in the dK / dV kernel
        at_f4 s = at_dot32(Ks + c * kAtRow + 4 * r, q0, q1), dp = at_dot32(Vs + c * kAtRow + 4 * r, g0, g1);
#pragma unroll 1
        for (int t = 0; t < ntk; ++t) {
            const int tn = min(t + 1, ntk - 1);
            const at_f4 s_next = at_dot32(Ks + (16 * tn + c) * kAtRow + 4 * r, q0, q1);
            const at_f4 dp_next = at_dot32(Vs + (16 * tn + c) * kAtRow + 4 * r, g0, g1);
            __builtin_amdgcn_sched_barrier(0);
            at_f4 ds;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float p = __builtin_amdgcn_exp2f(s[v] - lse);
                if (t == ntk - 1) p = 16 * t + 4 * r + v < a.Lk ? p : 0.f;      // (uniform: only the last tile has padding keys)
                const bool keep = at_hash(ctr + (16 * t + v)) >= a.thresh;
                ds[v] = p * ((keep ? dp[v] * a.keep_scale : 0.f) - delta);
            }
            at_accum_t(Ks + 16 * t * kAtRow, r, c, ds, dq0, dq1);           // dQ^T[channel][query] += K^T . dS^T
            s = s_next; dp = dp_next;
        }
        if (qok) {
            float *gq = a.gq.p + n * a.gq.sn + h * 32 + (long long)qi * a.gq.sl + 4 * r;
            *reinterpret_cast<float4 *>(gq) = make_float4(dq0[0] * a.scale, dq0[1] * a.scale, dq0[2] * a.scale, dq0[3] * a.scale);
            *reinterpret_cast<float4 *>(gq + 16) = make_float4(dq1[0] * a.scale, dq1[1] * a.scale, dq1[2] * a.scale, dq1[3] * a.scale);
        }
    }
}

static int at_allow_lds(const void *fn, size_t bytes)
{
    if (bytes <= 64 * 1024) return MSDA_OK;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return e == hipSuccess ? MSDA_OK : set_error(MSDA_ERR_LAUNCH, hipGetErrorString(e));
}

static bool at_view_ok(const AtView &v)
{
    return v.p != nullptr && ((uintptr_t)v.p & 15) == 0 && (v.sn & 3) == 0 && (v.sl & 3) == 0;
}

}  // namespace msda

using msda::AtArgs;
using msda::AtView;

extern "C" {

int msda_attn32_supported(int Lq, int Lk, int head_dim)
{
    return head_dim == 32 && Lq >= 1 && Lk >= 1 && Lq <= msda::kAtMaxLen && Lk <= msda::kAtMaxLen;
}

static int at_fill(AtArgs &a, const char *who, int N, int H, int Lq, int Lk, float scale, float dropout_p, const unsigned long long *seed)
{
    if (N <= 0 || H <= 0 || !msda_attn32_supported(Lq, Lk, 32)) return msda::set_error(MSDA_ERR_ARGUMENT, who);
    if (!(dropout_p >= 0.f && dropout_p < 1.f) || (dropout_p > 0.f && seed == nullptr)) return msda::set_error(MSDA_ERR_ARGUMENT, who);
    a.H = H; a.Lq = Lq; a.Lk = Lk; a.halves = msda::tuning_int("MSDA_ATTN_HALVES", 1) == 2 ? 2 : 1; a.scale = scale;
    a.scale2 = scale * 1.4426950408889634f;
    a.keep_scale = 1.f / (1.f - dropout_p);
    a.thresh = dropout_p > 0.f ? (unsigned)fmin(4294967295.0, (double)dropout_p * 4294967296.0) : 0u;
    a.seed = seed;
    return MSDA_OK;
}

int msda_attn32_forward_f32(const float *q, long long q_sn, long long q_sl, const float *k, long long k_sn, long long k_sl,
                            const float *v, long long v_sn, long long v_sl, int N, int H, int Lq, int Lk, float scale,
                            float dropout_p, const unsigned long long *seed, float *out, long long o_sn, long long o_sl, float *lse,
                            msda_stream_t stream)
{
    AtArgs a{};
    if (int rc = at_fill(a, "msda_attn32_forward_f32: head_dim 32, 1 <= Lq, Lk <= 320, 0 <= p < 1 (seed required for p > 0)", N, H, Lq, Lk,
                         scale, dropout_p, seed)) return rc;
    a.q = AtView{const_cast<float *>(q), q_sn, q_sl}; a.k = AtView{const_cast<float *>(k), k_sn, k_sl};
    a.v = AtView{const_cast<float *>(v), v_sn, v_sl}; a.o = AtView{out, o_sn, o_sl}; a.lse = lse;
    if (!msda::at_view_ok(a.q) || !msda::at_view_ok(a.k) || !msda::at_view_ok(a.v) || !msda::at_view_ok(a.o) || lse == nullptr)
        return msda::set_error(MSDA_ERR_ARGUMENT, "msda_attn32_forward_f32: 16-byte aligned tensors, strides multiples of 4 floats");
    const size_t lds = (size_t)2 * ((Lk + 15) & ~15) * msda::kAtRow * sizeof(float);
    if (int rc = msda::at_allow_lds(reinterpret_cast<const void *>(msda::attn32_fwd_kernel), lds)) return rc;
    hipLaunchKernelGGL(msda::attn32_fwd_kernel, dim3((unsigned)(N * H * a.halves)), dim3(msda::kAtBlock), lds, (hipStream_t)stream, a);
    return msda::check_launch("msda attention forward (head_dim 32)");
}

int msda_attn32_backward_f32(const float *q, long long q_sn, long long q_sl, const float *k, long long k_sn, long long k_sl,
                             const float *v, long long v_sn, long long v_sl, const float *out, long long o_sn, long long o_sl,
                             const float *lse, const float *grad_out, long long go_sn, long long go_sl, int N, int H, int Lq, int Lk,
                             float scale, float dropout_p, const unsigned long long *seed, float *grad_q, long long gq_sn,
                             long long gq_sl, float *grad_k, long long gk_sn, long long gk_sl, float *grad_v, long long gv_sn,
                             long long gv_sl, msda_stream_t stream)
{
    AtArgs a{};
    if (int rc = at_fill(a, "msda_attn32_backward_f32: head_dim 32, 1 <= Lq, Lk <= 320, 0 <= p < 1 (seed required for p > 0)", N, H, Lq, Lk,
                         scale, dropout_p, seed)) return rc;
    a.q = AtView{const_cast<float *>(q), q_sn, q_sl}; a.k = AtView{const_cast<float *>(k), k_sn, k_sl};
    a.v = AtView{const_cast<float *>(v), v_sn, v_sl}; a.o = AtView{const_cast<float *>(out), o_sn, o_sl};
    a.go = AtView{const_cast<float *>(grad_out), go_sn, go_sl}; a.lse = const_cast<float *>(lse);
    a.gq = AtView{grad_q, gq_sn, gq_sl}; a.gk = AtView{grad_k, gk_sn, gk_sl}; a.gv = AtView{grad_v, gv_sn, gv_sl};
    for (const AtView *w : {&a.q, &a.k, &a.v, &a.o, &a.go, &a.gq, &a.gk, &a.gv})
        if (!msda::at_view_ok(*w))
            return msda::set_error(MSDA_ERR_ARGUMENT, "msda_attn32_backward_f32: 16-byte aligned tensors, strides multiples of 4 floats");
    if (lse == nullptr) return msda::set_error(MSDA_ERR_ARGUMENT, "msda_attn32_backward_f32: lse required");
    const int Lqp = (Lq + 15) & ~15, Lkp = (Lk + 15) & ~15;
    const size_t lds_kv = ((size_t)2 * Lqp * msda::kAtRow + 2 * Lqp) * sizeof(float), lds_q = (size_t)2 * Lkp * msda::kAtRow * sizeof(float);
    if (int rc = msda::at_allow_lds(reinterpret_cast<const void *>(msda::attn32_bwd_kv_kernel), lds_kv)) return rc;
    if (int rc = msda::at_allow_lds(reinterpret_cast<const void *>(msda::attn32_bwd_q_kernel), lds_q)) return rc;
    const dim3 grid((unsigned)(N * H * a.halves));
    hipLaunchKernelGGL(msda::attn32_bwd_kv_kernel, grid, dim3(msda::kAtBlock), lds_kv, (hipStream_t)stream, a);
    if (int rc = msda::check_launch("msda attention backward (dK, dV)")) return rc;
    hipLaunchKernelGGL(msda::attn32_bwd_q_kernel, grid, dim3(msda::kAtBlock), lds_q, (hipStream_t)stream, a);
    return msda::check_launch("msda attention backward (dQ)");
}

}  // extern "C"
