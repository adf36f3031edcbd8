// D = 32 forward and backward role A for LARGE problems: the coarse levels of the pyramid served from LDS.
// Device code only; included by msda_d32.hip after its row / DPP / prologue helpers.
//
// Why: on large problems the forward and role A sit at the vector memory path's ROW-REQUEST ceiling (~0.4 rows per clock
// per CU, profiles/r02_notes.md §2) although most of their taps land on a few hundred pixels: every level receives the same
// number of sampling points, so at the training shape (28/14/7/4) the three coarse levels — 261 pixels, 33 KB per
// (batch, head) pair in fp32 — take 75 % of all taps, at 48/24/12/6 the two coarsest (180 pixels, 23 KB) take 50 %.
// Here a workgroup owns queries of ONE (batch, head) pair, copies the levels that fit a stage of `stage_rows` rows into
// LDS once (the smallest levels first), and then serves those levels' taps with ds_read_b128 — a separate pipe from the
// global loads of the remaining level(s), so the two overlap — for `qw` queries in sub-batches of 64.
// Replaces (large problems only) ms_deform_im2col_cuda.cuh:237-299 / :33-84 and the query-major half of :301-403.
//
// Results are bit-identical to fwd_d32_kernel / bwd_query_body: the same expressions, the same FMA order per item
// (points in order, taps 00 01 10 11); only where a row comes from differs.
//
// Record of a sampling point: 32 B, written once by the point's lane and read (broadcast) by the 8 lanes of its item —
//   off[4]: BYTE offset of each tap's row, ready for the load: inside the stage for a staged level (an absent tap points at
//           the stage's extra all-zero row), inside the batch element's slice of `value` otherwise (an absent tap gets an
//           offset past the buffer descriptor, for which the hardware returns zeros) — so the gather lanes spend ONE add
//           per tap (their 16-byte column) and no select at all: these kernels are bound by instruction issue
//           (~one wave64 instruction per clock per CU) as much as by row requests;
//   forward: w[4], the bilinear weights times the attention weight;   role A: lh, lw, attention weight.
// Role A writes a point's three sums back over its own record (only the item's lanes ever read it): no result array.
#pragma once
#include <type_traits>

namespace msda {

constexpr int kLBlock = 512;                       // 8 wavefronts, each on its own octets (8 queries) of the pair
constexpr int kLWaves = kLBlock / kWave;
constexpr int kLItems = kLWaves * 8;
constexpr int kLRec = 32;                          // bytes per point record
constexpr int kLStageBytes = 36 * 1024;            // stage budget: 288 fp32 rows / 576 bf16 rows

struct alignas(16) LdsLevel { int H, W, start, stage_row; };          // stage_row: first row in the stage; -1: the level stays in
                                                                      // global memory; -2: it does not fit in S (contributes nothing)

// LDS layout: [level table: (kMaxLevels + 1) x 16 B] [stage: stage_rows rows + one all-zero row] [records: 64 x item_stride]
template <typename VT>
__host__ __device__ inline size_t lds_variant_bytes(int stage_rows, int LP)
{
    return (size_t)(kMaxLevels + 1) * sizeof(LdsLevel) + (size_t)(stage_rows + 1) * kD * sizeof(VT) + (size_t)kLItems * (LP * kLRec + kItemPad);
}

// Which levels go to the stage and where: the levels in order of pixel count (ties: lower index first), as many as fit.
// Lane k < L works out level k (every lane loads the L shapes itself: independent loads, one round trip — a serial walk by
// one thread cost a workgroup 2-3 us before its first useful load): level k is staged iff the pixels of all levels that
// come before it in that order plus its own fit the stage; then it starts after them.
__device__ __forceinline__ void lds_level_table(const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start, int S,
                                                int L, int stage_rows, LdsLevel *tab)
{
    const int k = threadIdx.x;
    if (k >= L) return;
    const long long H = shapes[2 * k], W = shapes[2 * k + 1], st = level_start[k];
    const bool fits = level_fits(H, W, st, S);
    const int hw = fits ? (int)(H * W) : 0;
    int before = 0;
    for (int i = 0; i < L; ++i) {
        const long long Hi = shapes[2 * i], Wi = shapes[2 * i + 1];
        const int hwi = level_fits(Hi, Wi, level_start[i], S) ? (int)(Hi * Wi) : 0;
        if (hwi > 0 && (hwi < hw || (hwi == hw && i < k))) before += hwi;
    }
    LdsLevel t;
    t.H = (int)H; t.W = (int)W; t.start = (int)st;
    t.stage_row = !fits ? -2 : (before + hw <= stage_rows ? before : -1);
    tab[k] = t;
}

// Copies the staged levels' rows of pair (b, m) into LDS: 8 lanes x 16 B (8 B for bf16) per row, 64 rows per trip.
template <typename VT>
__device__ __forceinline__ void lds_stage_rows(const VT *__restrict__ value, const LdsLevel *tab, int L, int S, int M, int b, int m,
                                               VT *stage)
{
    const int tid = threadIdx.x, rr = tid >> 3, j = tid & 7;
    for (int l = 0; l < L; ++l) {
        const LdsLevel t = tab[l];
        if (t.stage_row < 0) continue;                                       // uniform
        const int hw = t.H * t.W;
        const VT *src = value + ((long long)(b * S + t.start) * M + m) * kD + j * 4;
        VT *dst = stage + (long long)t.stage_row * kD + j * 4;
        constexpr int RPT = kLBlock / 8;                                     // rows per trip
        for (int r0 = 0; r0 < hw; r0 += 4 * RPT) {                           // 4 loads in flight per thread
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {                                    // (unconditional: the last trip re-reads the last row)
                const int r = min(r0 + u * RPT + rr, hw - 1);
                v[u] = Row<VT>::load(src + (long long)r * M * kD);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + u * RPT + rr;
                if (r < hw) Row<VT>::store(dst + r * kD, v[u]);
            }
        }
    }
}

// (BufRow<>, kBufOob, kBufDword3: msda_d32.hip — the tiled kernels load through buffer descriptors too)

// One sampling point -> the offsets half of its record and its geometry.  Branch-free on purpose (selects): the point
// lanes of a wavefront belong to different levels and items.  `zero_off`: byte offset of the stage's zero row.
template <typename VT>
__device__ __forceinline__ uint4 lds_point_offsets(float x, float y, const LdsLevel &t, int m, int M, bool ok, unsigned zero_off,
                                                  PointGeom<float> &g)
{
    g = point_geom<float>(x, y, t.H, t.W);                                 // lh = lw = 0 and no tap valid when outside
    const bool use = ok && t.stage_row != -2 && g.inside;
    const bool staged = t.stage_row >= 0;
    const int pix = g.h0 * t.W + g.w0;
    constexpr int RB = kD * (int)sizeof(VT);                               // bytes per row
    const int rs = staged ? RB : M * RB;                                   // next pixel, next image row
    const int base = staged ? (t.stage_row + pix) * RB : ((t.start + pix) * M + m) * RB;
    const unsigned none = staged ? zero_off : kBufOob;
    return make_uint4(use && g.ok00 ? (unsigned)base : none, use && g.ok01 ? (unsigned)(base + rs) : none,
                      use && g.ok10 ? (unsigned)(base + t.W * rs) : none, use && g.ok11 ? (unsigned)(base + t.W * rs + rs) : none);
}

// The sixteen rows of a trip (4 points x 4 taps), all loads issued back to back: ds_read_b128 from the stage (STAGED), or
// buffer_load_dwordx4 from `value`.  joff = this lane's 16-byte (8-byte for bf16) column inside a row.  The callers keep
// the arithmetic that consumes the rows INSIDE the staged / global branch (a join after the loads made the compiler
// shuffle all sixteen results into common registers, one wait per row).
template <typename VT, bool STAGED>
__device__ __forceinline__ void lds_trip_rows(const unsigned char *stage_b, __amdgpu_buffer_rsrc_t buf, unsigned joff,
                                              const uint4 (&o)[4], float4 (&v)[4][4])
{
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const unsigned ou[4] = {o[u].x, o[u].y, o[u].z, o[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (STAGED) v[u][k] = Row<VT>::load(reinterpret_cast<const VT *>(stage_b + (ou[k] + joff)));
            else        v[u][k] = BufRow<VT>::load(buf, ou[k] + joff);      // (kBufOob + joff stays out of range)
        }
    }
}

// Record offsets of the trip's points p0 .. p0+3; `np` of them (1..4) belong to the level, the others read no tap.
__device__ __forceinline__ void lds_trip_offsets(const unsigned char *rb, int p0, int np, unsigned none, uint4 (&o)[4])
{
    if (np >= 4) {                                                           // uniform; the usual case (P a multiple of 4)
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = *reinterpret_cast<const uint4 *>(rb + (p0 + u) * kLRec);
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            o[u] = make_uint4(none, none, none, none);
            if (u < np) o[u] = *reinterpret_cast<const uint4 *>(rb + (p0 + u) * kLRec);
        }
    }
}

// Workgroup id -> (batch, head, first query, end query).  Chunks of one pair are neighbours (their level-0 rows share an L2).
__device__ __forceinline__ void lds_block_range(int bid, int chunks, int qw, int M, int Lq, int &b, int &m, int &q0, int &q1)
{
    const int pr = bid / chunks, c = bid - pr * chunks;
    b = pr / M; m = pr - b * M;
    q0 = c * qw; q1 = min(Lq, q0 + qw);
}

// ------------------------------------------------------------------------------------------
// Both kernels: after the stage is filled (ONE workgroup barrier) every wavefront works on its own — it takes octets (8
// queries of the pair) w, w + 8, ... of the workgroup's range, turns their sampling points into records in its own slice of
// LDS (LDS operations of one wavefront execute in issue order: no barrier between its writes and its reads), gathers, and
// stores.  No workgroup barrier inside the loop, so the wavefronts of a CU drift apart and one's wait for global rows is
// another's LDS phase; the next octet's locations are loaded before the current octet's gather (their latency hides
// behind it).
// ------------------------------------------------------------------------------------------
// NS = point slots per lane and octet = ceil(8 * L*P / 64), 1..4 (L*P <= 32); a template parameter so that the locations
// fetched ahead for the next octet take no more registers than the geometry needs (L*P = 16: 2 slots)

// forward
template <typename VT, bool FUSED, int NS>
__global__ __launch_bounds__(kLBlock, 4) void fwd_d32_lds_kernel(
    const VT *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ loc, const float *__restrict__ attn, int S, int M, int L, int Lq, int P, int p_shift, int lp_shift,
    int chunks, int qw, int stage_rows, VT *__restrict__ out, const PrologueIn pro, int xcd,
    const MaskOut mo = MaskOut{nullptr, nullptr, 0})
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsLevel *tab = reinterpret_cast<LdsLevel *>(smem);
    VT *stage = reinterpret_cast<VT *>(smem + (kMaxLevels + 1) * sizeof(LdsLevel));
    const int LP = L * P, item_stride = LP * kLRec + kItemPad;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, grp = lane >> 3, j = lane & 7;
    unsigned char *recs = reinterpret_cast<unsigned char *>(stage) + (size_t)(stage_rows + 1) * kD * sizeof(VT) + wave * 8 * item_stride;
    int b, m, q0, q1;
    const int lbid = xcd ? xcd_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    lds_block_range(lbid, chunks, qw, M, Lq, b, m, q0, q1);
    lds_level_table(shapes, level_start, S, L, stage_rows, tab);
    // Per-point range masks for role B of this node's backward (MaskHeader, msda_d32.hip): the ranges' first pixels [L][W + 1]
    // — the very expression role B cuts its ranges with — and W / (H*W) per level.  (uniform: mo.masks is a kernel argument)
    const int LW = mo.W;
    int *bnd = reinterpret_cast<int *>(smem + lds_variant_bytes<VT>(stage_rows, LP));
    float *bscale = reinterpret_cast<float *>(bnd + L * (LW + 1));
    if (mo.masks) {
        for (int i = tid; i < L * (LW + 1); i += kLBlock) {
            const int l = i / (LW + 1), t = i - l * (LW + 1);
            const long long H = shapes[2 * l], Wd = shapes[2 * l + 1];
            const int HW = level_fits(H, Wd, level_start[l], S) ? (int)(H * Wd) : 0;
            bnd[i] = (int)((unsigned)(t * HW) / (unsigned)LW);
            if (t == 0) bscale[l] = HW > 0 ? (float)LW / (float)HW : 0.f;
        }
        if (lbid == 0 && tid == 0) {
            MaskHeader h;
            h.magic = kMaskMagic; h.W = LW; h.L = L; h.NP = Lq * P; h.pairs = (int)gridDim.x / chunks;
#pragma unroll
            for (int k = 0; k < 11; ++k) h.pad[k] = 0;
            *mo.hdr = h;
        }
    }
    const size_t mask_base = (size_t)(lbid / chunks) * L * ((size_t)Lq * P);       // this pair's [L][Lq*P] bytes
    __syncthreads();
    lds_stage_rows<VT>(value, tab, L, S, M, b, m, stage);
    if (tid < 8) Row<VT>::store(stage + stage_rows * kD + tid * 4, make_float4(0.f, 0.f, 0.f, 0.f));
    const __amdgpu_buffer_rsrc_t vbuf = uniform_rsrc(value + (long long)b * S * M * kD, (long long)S * M * kD * (long long)sizeof(VT));
    const float2 *loc2 = reinterpret_cast<const float2 *>(loc);
    const unsigned char *rb = recs + grp * item_stride;
    const unsigned char *stage_b = reinterpret_cast<const unsigned char *>(stage);
    const unsigned zero_off = (unsigned)(stage_rows * kD * sizeof(VT)), joff = (unsigned)(j * 4 * sizeof(VT));

    // the octet's points, one per lane and slot: loc / attn (FUSED: raw offset / logit) of point `idx` of the octet at qo
    float2 pxy[NS]; float pa[NS];
    auto fetch = [&](int qo) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int idx = lane + k * kWave;
            if (idx >= 8 * LP) break;                                        // uniform
            const int i2 = fdiv(idx, LP, lp_shift), pt = idx - i2 * LP;
            const long long row = (long long)b * Lq + min(qo + i2, q1 - 1);
            const long long e = (row * M + m) * LP + pt;
            pxy[k] = loc2[FUSED ? e + row * pro.off_pad : e];
            pa[k] = attn[FUSED ? e + row * pro.log_pad : e];
        }
    };
    const int step = 8 * kLWaves;
    int qo = q0 + wave * 8;
    if (qo < q1) fetch(qo);
    __syncthreads();                                             // stage, zero row and level table are in LDS
    for (; qo < q1; qo += step) {
        // ---- records of this octet (one lane per sampling point) ----
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int idx = lane + k * kWave;
            if (idx >= 8 * LP) break;
            const int i2 = fdiv(idx, LP, lp_shift), pt = idx - i2 * LP, l = fdiv(pt, P, p_shift);
            const bool live = qo + i2 < q1;
            float2 xy = pxy[k];
            float a = pa[k];
            const LdsLevel t = tab[l];
            if (FUSED) {
                const long long row = (long long)b * Lq + min(qo + i2, q1 - 1), e = (row * M + m) * LP + pt;
                const float2 rp = reinterpret_cast<const float2 *>(pro.ref)[row * L + l];
                xy = make_float2(rp.x + xy.x / (float)t.W, rp.y + xy.y / (float)t.H);
                const float ex = expf(a - group_max(a, LP));
                a = ex / group_sum(ex, LP);
                if (live) {
                    reinterpret_cast<float2 *>(pro.loc_out)[e] = xy;
                    pro.attn_out[e] = a;
                }
            }
            PointGeom<float> g;
            const uint4 off = lds_point_offsets<VT>(xy.x, xy.y, t, m, M, live, zero_off, g);
            const float hh = 1.f - g.lh, hw = 1.f - g.lw;
            unsigned char *slot = recs + i2 * item_stride + pt * kLRec;
            *reinterpret_cast<uint4 *>(slot) = off;
            // (outside the map lh = lw = 0 and every tap reads zeros: the weight only has to be finite -> 0)
            const float aw = g.inside ? a : 0.f;
            *reinterpret_cast<float4 *>(slot + 16) = make_float4(hh * hw * aw, hh * g.lw * aw, g.lh * hw * aw, g.lh * g.lw * aw);
            if (mo.masks) {                                                  // (uniform) bit t: a tap of this point may land in range t
                const int *bl = bnd + l * (LW + 1);
                const int HW = bl[LW];
                unsigned mk = 0;
                if (g.inside && HW > 0) {
                    const int pix = g.h0 * t.W + g.w0;                       // tap (h0, w0); its neighbours: +1, +W, +W+1
                    const int pmin = max(pix, 0), pmax = min(pix + t.W + 1, HW - 1);
                    // range of a pixel: ranges start at floor(k * HW / W); a float estimate, made exact against the table
                    const float sc = bscale[l];
                    int t0 = min(max((int)((float)(pmin + 1) * sc), 0), LW - 1);
                    int t1 = min(max((int)((float)(pmax + 1) * sc), 0), LW - 1);
                    t0 += (int)(pmin >= bl[t0 + 1]) - (int)(pmin < bl[t0]);
                    t1 += (int)(pmax >= bl[t1 + 1]) - (int)(pmax < bl[t1]);
                    mk = (2u << t1) - (1u << t0);
                }
                if (live) mo.masks[mask_base + (size_t)(l * (Lq * P) + (qo + i2) * P + (pt - l * P))] = (uint8_t)mk;
            }
        }
        if (qo + step < q1) fetch(qo + step);                    // next octet's locations: in flight during the gather
        __builtin_amdgcn_wave_barrier();
        // ---- gather: 8 lanes x float4 per item, level by level, up to 4 points per trip ----
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
        for (int l = 0; l < L; ++l) {
            const LdsLevel t = tab[l];
            const bool staged = __builtin_amdgcn_readfirstlane(t.stage_row) >= 0;
            const unsigned none = staged ? zero_off : kBufOob;
#pragma unroll 1
            for (int p0 = l * P; p0 < (l + 1) * P; p0 += 4) {
                const int np = min(4, (l + 1) * P - p0);
                auto trip = [&](auto staged_c) {
                    uint4 o[4]; float4 v[4][4];
                    lds_trip_offsets(rb, p0, np, none, o);
                    lds_trip_rows<VT, decltype(staged_c)::value>(stage_b, vbuf, joff, o, v);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {                            // (slots past np: rows are zeros, any finite weight)
                        const float4 w = *reinterpret_cast<const float4 *>(rb + min(p0 + u, LP - 1) * kLRec + 16);
                        fma4(acc, w.x, v[u][0]); fma4(acc, w.y, v[u][1]); fma4(acc, w.z, v[u][2]); fma4(acc, w.w, v[u][3]);
                    }
                };
                if (staged) trip(std::true_type{}); else trip(std::false_type{});
            }
        }
        if (qo + grp < q1) Row<VT>::store(out + (((long long)b * Lq + qo + grp) * M + m) * kD + j * 4, acc);
        __builtin_amdgcn_wave_barrier();                         // the records are rewritten for the next octet
    }
}

// ------------------------------------------------------------------------------------------
// backward, role A (grad_sampling_loc / grad_attn_weight); FUSED: the gradients of the raw offsets / logits, and the
// reference-point gradient PER HEAD into pro.grad_ref ([N, Lq, M, L, 2] here: a workgroup sees one head only — the caller
// adds the heads in a fixed order, ref_heads_reduce_kernel).
// ------------------------------------------------------------------------------------------
template <typename VT, bool FUSED, int NS>
__device__ __forceinline__ void bwd_query_lds_body(
    const VT *__restrict__ grad_out, const VT *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ level_start, const float *__restrict__ loc, const float *__restrict__ attn, int S, int M, int L,
    int Lq, int P, int p_shift, int lp_shift, int chunks, int qw, int stage_rows, float *__restrict__ grad_loc,
    float *__restrict__ grad_attn, const PrologueOut pro, int block, unsigned char *smem)
{
    LdsLevel *tab = reinterpret_cast<LdsLevel *>(smem);
    VT *stage = reinterpret_cast<VT *>(smem + (kMaxLevels + 1) * sizeof(LdsLevel));
    const int LP = L * P, item_stride = LP * kLRec + kItemPad;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, grp = lane >> 3, j = lane & 7;
    unsigned char *recs = reinterpret_cast<unsigned char *>(stage) + (size_t)(stage_rows + 1) * kD * sizeof(VT) + wave * 8 * item_stride;
    int b, m, q0, q1;
    lds_block_range(block, chunks, qw, M, Lq, b, m, q0, q1);
    lds_level_table(shapes, level_start, S, L, stage_rows, tab);
    __syncthreads();
    lds_stage_rows<VT>(value, tab, L, S, M, b, m, stage);
    if (tid < 8) Row<VT>::store(stage + stage_rows * kD + tid * 4, make_float4(0.f, 0.f, 0.f, 0.f));
    const __amdgpu_buffer_rsrc_t vbuf = uniform_rsrc(value + (long long)b * S * M * kD, (long long)S * M * kD * (long long)sizeof(VT));
    const float2 *loc2 = reinterpret_cast<const float2 *>(loc);
    unsigned char *rb = recs + grp * item_stride;
    const unsigned char *stage_b = reinterpret_cast<const unsigned char *>(stage);
    const unsigned zero_off = (unsigned)(stage_rows * kD * sizeof(VT)), joff = (unsigned)(j * 4 * sizeof(VT));

    float2 pxy[NS]; float pa[NS];
    auto fetch = [&](int qo) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int idx = lane + k * kWave;
            if (idx >= 8 * LP) break;
            const int i2 = fdiv(idx, LP, lp_shift), pt = idx - i2 * LP;
            const long long e = (((long long)b * Lq + min(qo + i2, q1 - 1)) * M + m) * LP + pt;
            pxy[k] = loc2[e];
            pa[k] = attn[e];
        }
    };
    const int step = 8 * kLWaves;
    int qo = q0 + wave * 8;
    if (qo < q1) fetch(qo);
    __syncthreads();
    for (; qo < q1; qo += step) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int idx = lane + k * kWave;
            if (idx >= 8 * LP) break;
            const int i2 = fdiv(idx, LP, lp_shift), pt = idx - i2 * LP, l = fdiv(pt, P, p_shift);
            PointGeom<float> g;
            const uint4 off = lds_point_offsets<VT>(pxy[k].x, pxy[k].y, tab[l], m, M, qo + i2 < q1, zero_off, g);
            unsigned char *slot = recs + i2 * item_stride + pt * kLRec;
            *reinterpret_cast<uint4 *>(slot) = off;
            *reinterpret_cast<float4 *>(slot + 16) = make_float4(g.lh, g.lw, g.inside ? pa[k] : 0.f, 0.f);
        }
        // (the item's grad_out row: its latency hides behind the records phase above and the first trip's loads below)
        float4 g4 = Row<VT>::load(grad_out + (((long long)b * Lq + min(qo + grp, q1 - 1)) * M + m) * kD + j * 4);
        if (qo + grp >= q1) g4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qo + step < q1) fetch(qo + step);
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (int l = 0; l < L; ++l) {
            const LdsLevel t = tab[l];
            const bool staged = __builtin_amdgcn_readfirstlane(t.stage_row) >= 0;
            const unsigned none = staged ? zero_off : kBufOob;
#pragma unroll 1
            for (int p0 = l * P; p0 < (l + 1) * P; p0 += 4) {
                const int np = min(4, (l + 1) * P - p0);
                auto trip = [&](auto staged_c) {
                    uint4 o[4]; float4 v[4][4];
                    lds_trip_offsets(rb, p0, np, none, o);
                    lds_trip_rows<VT, decltype(staged_c)::value>(stage_b, vbuf, joff, o, v);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float4 f = *reinterpret_cast<const float4 *>(rb + min(p0 + u, LP - 1) * kLRec + 16);    // lh lw a -
                        const float d1 = dot4(g4, v[u][0]), d2 = dot4(g4, v[u][1]), d3 = dot4(g4, v[u][2]), d4 = dot4(g4, v[u][3]);
                        float s_a, s_x, s_y;
                        tap_sums(f.x, f.y, f.z, d1, d2, d3, d4, s_a, s_x, s_y);
                        s_a = octlane_sum(s_a); s_x = octlane_sum(s_x); s_y = octlane_sum(s_y);
                        // the point's record has been consumed (by this item's lanes only): its slot takes the result
                        if (j == 0 && u < np) *reinterpret_cast<float4 *>(rb + (p0 + u) * kLRec) = make_float4(s_x, s_y, s_a, 0.f);
                    }
                };
                if (staged) trip(std::true_type{}); else trip(std::false_type{});
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- write-out: an item's L*P points are L*P consecutive lanes -> contiguous runs of both gradient tensors ----
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int idx = lane + k * kWave;
            if (idx >= 8 * LP) break;
            const int i2 = fdiv(idx, LP, lp_shift), pt = idx - i2 * LP, l = fdiv(pt, P, p_shift);
            if (qo + i2 >= q1) continue;                                      // whole items drop out (their lanes together)
            const long long row = (long long)b * Lq + qo + i2, e = (row * M + m) * LP + pt;
            const float4 r = *reinterpret_cast<const float4 *>(recs + i2 * item_stride + pt * kLRec);
            const LdsLevel t = tab[l];
            const float gx = r.x * (float)t.W, gy = r.y * (float)t.H;
            if (FUSED) {
                reinterpret_cast<float2 *>(grad_loc)[e + row * pro.off_pad] = make_float2(r.x, r.y);
                const float a = attn[e];
                const float dot = group_sum(a * r.z, LP);
                grad_attn[e + row * pro.log_pad] = a * (r.z - dot);
                const float sx = group_sum(gx, P), sy = group_sum(gy, P);
                if ((pt & (P - 1)) == 0) reinterpret_cast<float2 *>(pro.grad_ref)[(row * M + m) * L + l] = make_float2(sx, sy);
            } else {
                reinterpret_cast<float2 *>(grad_loc)[e] = make_float2(gx, gy);
                grad_attn[e] = r.z;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename VT, bool FUSED, int NS>
__global__ __launch_bounds__(kLBlock, 4) void bwd_query_d32_lds_kernel(
    const VT *__restrict__ grad_out, const VT *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ level_start, const float *__restrict__ loc, const float *__restrict__ attn, int S, int M, int L,
    int Lq, int P, int p_shift, int lp_shift, int chunks, int qw, int stage_rows, float *__restrict__ grad_loc,
    float *__restrict__ grad_attn, const PrologueOut pro, int xcd)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bwd_query_lds_body<VT, FUSED, NS>(grad_out, value, shapes, level_start, loc, attn, S, M, L, Lq, P, p_shift, lp_shift, chunks, qw,
                                      stage_rows, grad_loc, grad_attn, pro,
                                      xcd ? xcd_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x, smem);
}

// grad_ref[N*Lq, L, 2] = sum over the M heads of heads[N*Lq, M, L, 2], heads in order (reproducible).
static __global__ __launch_bounds__(256) void ref_heads_reduce_kernel(const float2 *__restrict__ heads, int rows, int M, int L,
                                                               float2 *__restrict__ grad_ref)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)rows * L) return;
    const long long row = i / L;
    const int l = (int)(i - row * L);
    float2 acc = make_float2(0.f, 0.f);
    for (int mm = 0; mm < M; ++mm) { const float2 v = heads[(row * M + mm) * L + l]; acc.x += v.x; acc.y += v.y; }
    grad_ref[i] = acc;
}

}  // namespace msda
