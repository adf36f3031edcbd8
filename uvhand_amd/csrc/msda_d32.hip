// D = 32, fp32 multi-scale deformable attention for MI355X (gfx950) — the shape the
// UVHand transformers use (d_model 256 / 8 heads; util/settings.py:102-120).
//
// Work decomposition (not the reference's one-thread-per-channel / 32-thread blocks,
// ms_deform_im2col_cuda.cuh:237-299, :301-403):
//
//   * item  = one (batch, query, head); its output row is 32 floats = 128 B.
//   * octet = 8 consecutive items = 64 lanes: 8 lanes per item, one float4 (16 B) per
//     lane, so every tap is one 128-B line per item and a wavefront's store of its
//     octet is one contiguous 1-KiB write (for M = 8 an octet is exactly one query).
//   * a 256-thread workgroup first turns the sampling locations of its items into
//     "tap records" in LDS — ONE lane per sampling point does the floor / bounds /
//     bilinear-weight arithmetic once, reading loc and attn coalesced from HBM —
//     and then the gather lanes only read records (broadcast ds_read_b128) and rows.
//     The reference recomputes that arithmetic in every one of the 32 channel threads.
//   * SPLIT = 4: the 4 wavefronts of a workgroup share ONE octet and split its L*P
//     points, so that small problems (the 300-query decoder shape has only 600
//     octets) still put ~2400 wavefronts with 16 row loads each in flight;
//     SPLIT = 1: one octet per wavefront, no cross-wave reduction (encoder regime,
//     large batches).
#include <cstdlib>
#include <cstring>

#include "msda_common.h"
#include "msda_launch.h"

namespace msda {

constexpr int kD = 32;
constexpr int kBlock = 256;
constexpr int kRecBytes = 32;      // one tap record
constexpr int kItemPad = 16;       // bytes: shifts consecutive items by one 16-B bank slot so the
                                   // 8 per-item broadcast reads of a wavefront do not collide
constexpr int kLvBytes = kMaxLevels * 16;

struct alignas(16) LevelInfo { int H, W, start, pad; };

// forward record: element offsets of the 4 taps (-1 = tap outside the map) and their
// weights already multiplied by the attention weight.
struct alignas(16) FwdRec { int off[4]; float w[4]; };
// backward record: same offsets, the two fractions and the attention weight.
struct alignas(16) BwdRec { int off[4]; float lh, lw, a, pad; };

__device__ __forceinline__ void tap_offsets(const PointGeom<float> &g, const LevelInfo &lv, int b,
                                            int m, int S, int M, int off[4])
{
    const int row = M * kD;                                            // elements per pixel
    const int base = ((b * S + lv.start) * M + m) * kD + (g.h0 * lv.W + g.w0) * row;
    off[0] = g.ok00 ? base : -1;
    off[1] = g.ok01 ? base + row : -1;
    off[2] = g.ok10 ? base + lv.W * row : -1;
    off[3] = g.ok11 ? base + lv.W * row + row : -1;
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

__device__ __forceinline__ void fma4(float4 &acc, float w, const float4 &v)
{
    acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y);
    acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
}
__device__ __forceinline__ float dot4(const float4 &a, const float4 &b)
{
    return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

// Sum over the 8 lanes that share an item (lanes 8k..8k+7).
__device__ __forceinline__ float octlane_sum(float x)
{
    x += __shfl_xor(x, 1, kWave);
    x += __shfl_xor(x, 2, kWave);
    x += __shfl_xor(x, 4, kWave);
    return x;
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int SPLIT>
__global__ __launch_bounds__(kBlock) void fwd_d32_kernel(
    const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ level_start, const float *__restrict__ loc,
    const float *__restrict__ attn, int S, int M, int L, int Lq, int P, int items,
    float *__restrict__ out)
{
    constexpr int IPW = 32 / SPLIT;                       // items per workgroup
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LevelInfo *lv = reinterpret_cast<LevelInfo *>(smem);
    unsigned char *recs = smem + kLvBytes;
    const int LP = L * P;
    const int item_stride = LP * kRecBytes + kItemPad;
    const int tid = threadIdx.x;
    const int item0 = blockIdx.x * IPW;

    if (tid < L) {
        LevelInfo li;
        li.H = (int)shapes[2 * tid]; li.W = (int)shapes[2 * tid + 1];
        li.start = (int)level_start[tid]; li.pad = 0;
        lv[tid] = li;
    }
    __syncthreads();

    // ---- one lane per sampling point: loc/attn coalesced from HBM -> tap records in LDS ----
    for (int idx = tid; idx < IPW * LP; idx += kBlock) {
        const int il = idx / LP, pt = idx - il * LP;
        const int item = item0 + il;
        FwdRec r;
        r.off[0] = r.off[1] = r.off[2] = r.off[3] = -1;
        r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0.f;
        if (item < items) {
            const float2 xy = reinterpret_cast<const float2 *>(loc)[(long long)item0 * LP + idx];
            const float a = attn[(long long)item0 * LP + idx];
            const LevelInfo li = lv[pt / P];
            const PointGeom<float> g = point_geom<float>(xy.x, xy.y, li.H, li.W);
            if (g.inside) {
                tap_offsets(g, li, item / (Lq * M), item % M, S, M, r.off);
                const float hh = 1.f - g.lh, hw = 1.f - g.lw;
                r.w[0] = hh * hw * a; r.w[1] = hh * g.lw * a;
                r.w[2] = g.lh * hw * a; r.w[3] = g.lh * g.lw * a;
            }
        }
        *reinterpret_cast<FwdRec *>(recs + il * item_stride + pt * kRecBytes) = r;
    }
    __syncthreads();

    // ---- gather: 8 lanes x float4 per item, 8 items per wavefront ----
    const int wave = tid >> 6, lane = tid & 63, grp = lane >> 3, j = lane & 7;
    const int il = (SPLIT == 1 ? wave * 8 : 0) + grp;
    const unsigned char *rb = recs + il * item_stride;
    const float *vb = value + j * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int p = (SPLIT == 1 ? 0 : wave); p < LP; p += SPLIT) {
        const int4 off = *reinterpret_cast<const int4 *>(rb + p * kRecBytes);
        const float4 w = *reinterpret_cast<const float4 *>(rb + p * kRecBytes + 16);
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0, v2 = v0, v3 = v0;
        if (off.x >= 0) v0 = ld4(vb + off.x);
        if (off.y >= 0) v1 = ld4(vb + off.y);
        if (off.z >= 0) v2 = ld4(vb + off.z);
        if (off.w >= 0) v3 = ld4(vb + off.w);
        fma4(acc, w.x, v0); fma4(acc, w.y, v1); fma4(acc, w.z, v2); fma4(acc, w.w, v3);
    }

    if (SPLIT == 1) {
        if (item0 + il < items)
            *reinterpret_cast<float4 *>(out + (long long)(item0 + il) * kD + j * 4) = acc;
    } else {
        // the 4 wavefronts hold partial sums of the same octet: combine through LDS in a
        // fixed order, then one coalesced 1-KiB store.
        float *red = reinterpret_cast<float *>(recs + IPW * item_stride);
        reinterpret_cast<float4 *>(red)[wave * 64 + lane] = acc;
        __syncthreads();
        const float s = ((red[tid] + red[256 + tid]) + red[512 + tid]) + red[768 + tid];
        if (item0 + (tid >> 5) < items) out[(long long)item0 * kD + tid] = s;
    }
}

// ------------------------------------------------------------------------------------------
// backward, role A — query-major: grad_sampling_loc and grad_attn_weight.
// Same tiling as the forward; each lane forms the four tap dot products <grad_out, v_k> over its
// 4 channels, the three per-point sums are reduced over the item's 8 lanes and staged in LDS so
// that the workgroup writes both gradient tensors as contiguous runs.
// ATOMIC = true additionally scatters grad_value with global float atomics (the v1 scheme, kept
// for A/B measurements: MSDA_BWD_MODE=atomic); the default leaves grad_value to role B.
// ------------------------------------------------------------------------------------------
template <int SPLIT, bool ATOMIC>
__global__ __launch_bounds__(kBlock) void bwd_query_d32_kernel(
    const float *__restrict__ grad_out, const float *__restrict__ value,
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ loc, const float *__restrict__ attn, int S, int M, int L, int Lq,
    int P, int items, float *__restrict__ grad_value, float *__restrict__ grad_loc,
    float *__restrict__ grad_attn)
{
    constexpr int IPW = 32 / SPLIT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LevelInfo *lv = reinterpret_cast<LevelInfo *>(smem);
    unsigned char *recs = smem + kLvBytes;
    const int LP = L * P;
    const int item_stride = LP * kRecBytes + kItemPad;
    float4 *res = reinterpret_cast<float4 *>(recs + IPW * item_stride);   // [IPW*LP] (gx, gy, ga, -)
    const int tid = threadIdx.x;
    const int item0 = blockIdx.x * IPW;

    if (tid < L) {
        LevelInfo li;
        li.H = (int)shapes[2 * tid]; li.W = (int)shapes[2 * tid + 1];
        li.start = (int)level_start[tid]; li.pad = 0;
        lv[tid] = li;
    }
    __syncthreads();

    for (int idx = tid; idx < IPW * LP; idx += kBlock) {
        const int il = idx / LP, pt = idx - il * LP;
        const int item = item0 + il;
        BwdRec r;
        r.off[0] = r.off[1] = r.off[2] = r.off[3] = -1;
        r.lh = r.lw = r.a = r.pad = 0.f;
        if (item < items) {
            const float2 xy = reinterpret_cast<const float2 *>(loc)[(long long)item0 * LP + idx];
            const LevelInfo li = lv[pt / P];
            const PointGeom<float> g = point_geom<float>(xy.x, xy.y, li.H, li.W);
            if (g.inside) {
                tap_offsets(g, li, item / (Lq * M), item % M, S, M, r.off);
                r.lh = g.lh; r.lw = g.lw;
                r.a = attn[(long long)item0 * LP + idx];
            }
        }
        *reinterpret_cast<BwdRec *>(recs + il * item_stride + pt * kRecBytes) = r;
    }
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63, grp = lane >> 3, j = lane & 7;
    const int il = (SPLIT == 1 ? wave * 8 : 0) + grp;
    const unsigned char *rb = recs + il * item_stride;
    const float *vb = value + j * 4;
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (item0 + il < items) g4 = ld4(grad_out + (long long)(item0 + il) * kD + j * 4);

#pragma unroll 4
    for (int p = (SPLIT == 1 ? 0 : wave); p < LP; p += SPLIT) {
        const int4 off = *reinterpret_cast<const int4 *>(rb + p * kRecBytes);
        const float4 f = *reinterpret_cast<const float4 *>(rb + p * kRecBytes + 16);   // lh lw a -
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0, v2 = v0, v3 = v0;
        if (off.x >= 0) v0 = ld4(vb + off.x);
        if (off.y >= 0) v1 = ld4(vb + off.y);
        if (off.z >= 0) v2 = ld4(vb + off.z);
        if (off.w >= 0) v3 = ld4(vb + off.w);
        const float lh = f.x, lw = f.y, a = f.z, hh = 1.f - lh, hw = 1.f - lw;
        const float k1 = hh * hw, k2 = hh * lw, k3 = lh * hw, k4 = lh * lw;
        if (ATOMIC) {
            float *gvb = grad_value + j * 4;
            if (off.x >= 0) { const float c = k1 * a; float *d = gvb + off.x;
                atomicAdd(d, c * g4.x); atomicAdd(d + 1, c * g4.y); atomicAdd(d + 2, c * g4.z); atomicAdd(d + 3, c * g4.w); }
            if (off.y >= 0) { const float c = k2 * a; float *d = gvb + off.y;
                atomicAdd(d, c * g4.x); atomicAdd(d + 1, c * g4.y); atomicAdd(d + 2, c * g4.z); atomicAdd(d + 3, c * g4.w); }
            if (off.z >= 0) { const float c = k3 * a; float *d = gvb + off.z;
                atomicAdd(d, c * g4.x); atomicAdd(d + 1, c * g4.y); atomicAdd(d + 2, c * g4.z); atomicAdd(d + 3, c * g4.w); }
            if (off.w >= 0) { const float c = k4 * a; float *d = gvb + off.w;
                atomicAdd(d, c * g4.x); atomicAdd(d + 1, c * g4.y); atomicAdd(d + 2, c * g4.z); atomicAdd(d + 3, c * g4.w); }
        }
        const float d1 = dot4(g4, v0), d2 = dot4(g4, v1), d3 = dot4(g4, v2), d4 = dot4(g4, v3);
        float s_a = k1 * d1 + k2 * d2 + k3 * d3 + k4 * d4;
        float s_x = a * (hh * (d2 - d1) + lh * (d4 - d3));
        float s_y = a * (hw * (d3 - d1) + lw * (d4 - d2));
        s_a = octlane_sum(s_a); s_x = octlane_sum(s_x); s_y = octlane_sum(s_y);
        if (j == 0) res[il * LP + p] = make_float4(s_x, s_y, s_a, 0.f);
    }
    __syncthreads();

    // ---- coalesced write-out of grad_sampling_loc / grad_attn_weight for the workgroup's items ----
    for (int idx = tid; idx < IPW * LP; idx += kBlock) {
        const int il2 = idx / LP, pt = idx - il2 * LP;
        if (item0 + il2 < items) {
            const LevelInfo li = lv[pt / P];
            const float4 r = res[idx];
            reinterpret_cast<float2 *>(grad_loc)[(long long)item0 * LP + idx] =
                make_float2(r.x * (float)li.W, r.y * (float)li.H);
            grad_attn[(long long)item0 * LP + idx] = r.z;
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward, role B — destination-major: grad_value with no float atomics at all.
//
// grad_value[b, s, m, :] = sum over the taps that land on pixel s of (bilinear weight * attention
// weight) * grad_out[b, q, m, :] — a sparse-matrix x dense-matrix product whose sparse factor is
// only known at run time.  Float atomics are the wrong tool for it on this chip: global float
// atomics run at ~1.3 TB/s of added bytes chip-wide (MI355X_MICROARCH.md), and LDS float atomics
// are slower still — ds_add_f32 measured ~200 cycles per wavefront instruction whatever the
// address pattern (tools/micro/lds_atomic_bench.hip), i.e. ~0.8 TB/s chip-wide, while LDS
// INTEGER atomics are >6x faster.  So each workgroup sorts instead of scattering:
//
//   a workgroup owns the rows of (batch b, head m, level l, pixel range [px0, px1)) and
//   1. scans the level's Lq*P sampling points of (b, m), one lane per point, and counts the taps
//      that land on each of its rows        (LDS integer atomics: histogram)
//   2. prefix-sums the histogram            (row -> segment of the record array)
//   3. scans again and drops a record {weight, query} into the row's segment
//      (counting sort, LDS integer atomics for the cursor)
//   4. gathers: SLOTS x 8 lanes per row walk the row's segment, each record one coalesced 128-B
//      read of grad_out, accumulate in registers, combine the SLOTS partial sums with cross-lane
//      shuffles, and store the row once (zeros included) — no zero-fill pass, no atomics, every
//      element of grad_value written exactly once by exactly one workgroup.
//
// Every level is cut into the same number W of pixel ranges: all levels receive the same number
// of points, so equal counts of workgroups per level balance the gather work even though a coarse
// level has 4x fewer pixels.  W comes from the host (it only needs S and the batch size); the
// ranges come from spatial_shapes on the device.
// MULTIPASS (Lq*P too large for the record array): queries are processed in chunks and the rows
// accumulate in an LDS tile owned by the same lanes in every pass, flushed once at the end.
// ------------------------------------------------------------------------------------------
constexpr int kSBlock = 512;
constexpr int kSWaves = kSBlock / kWave;
struct alignas(8) SRec { float w; int q; };

struct TapSet { int dest[4]; float w[4]; };      // dest < 0: tap not in this workgroup's range

__device__ __forceinline__ bool point_taps(const float *__restrict__ loc, const float *__restrict__ attn,
                                           long long pi, int H, int Wd, int px0, int npx, TapSet &t)
{
    const float2 xy = reinterpret_cast<const float2 *>(loc)[pi];
    const PointGeom<float> g = point_geom<float>(xy.x, xy.y, H, Wd);
    if (!g.inside) return false;
    const int pix = g.h0 * Wd + g.w0 - px0;                  // range-local index of tap (h0, w0)
    const int p01 = pix + 1, p10 = pix + Wd, p11 = pix + Wd + 1;
    t.dest[0] = (g.ok00 && pix >= 0 && pix < npx) ? pix : -1;
    t.dest[1] = (g.ok01 && p01 >= 0 && p01 < npx) ? p01 : -1;
    t.dest[2] = (g.ok10 && p10 >= 0 && p10 < npx) ? p10 : -1;
    t.dest[3] = (g.ok11 && p11 >= 0 && p11 < npx) ? p11 : -1;
    if ((t.dest[0] & t.dest[1] & t.dest[2] & t.dest[3]) < 0) return false;   // all four are -1
    const float a = attn[pi];
    const float hh = 1.f - g.lh, hw = 1.f - g.lw;
    t.w[0] = hh * hw * a; t.w[1] = hh * g.lw * a; t.w[2] = g.lh * hw * a; t.w[3] = g.lh * g.lw * a;
    return true;
}

__device__ __forceinline__ float4 shfl_xor4(const float4 &v, int m)
{
    return make_float4(__shfl_xor(v.x, m, kWave), __shfl_xor(v.y, m, kWave), __shfl_xor(v.z, m, kWave),
                       __shfl_xor(v.w, m, kWave));
}
__device__ __forceinline__ void add4(float4 &a, const float4 &b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// Step 4.  SLOTS lanes-groups of 8 lanes share one row's segment; 8/SLOTS rows per wavefront pass.
template <int SLOTS, bool MULTIPASS>
__device__ __forceinline__ void gather_rows(const float *__restrict__ go_base, float *__restrict__ gv_base,
                                            const int *cnt, const int *start, const SRec *rec, float *tile,
                                            int npx, int row_stride, bool first_pass)
{
    constexpr int DPW = 8 / SLOTS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int dsub = lane / (SLOTS * 8), slot = (lane >> 3) % SLOTS, j = lane & 7;
    for (int d0 = wave * DPW; d0 < npx; d0 += kSWaves * DPW) {
        const int d = d0 + dsub;
        float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
        if (d < npx) {
            const int n = cnt[d];
            const SRec *r = rec + start[d];
            int i = slot;
            for (; i + SLOTS < n; i += 2 * SLOTS) {                 // two records in flight per lane
                const SRec r0 = r[i], r1 = r[i + SLOTS];
                const float4 g0 = ld4(go_base + (long long)r0.q * row_stride);
                const float4 g1 = ld4(go_base + (long long)r1.q * row_stride);
                fma4(acc0, r0.w, g0); fma4(acc1, r1.w, g1);
            }
            if (i < n) {
                const SRec r0 = r[i];
                fma4(acc0, r0.w, ld4(go_base + (long long)r0.q * row_stride));
            }
        }
        add4(acc0, acc1);
        if (SLOTS >= 2) add4(acc0, shfl_xor4(acc0, 8));
        if (SLOTS >= 4) add4(acc0, shfl_xor4(acc0, 16));
        if (SLOTS >= 8) add4(acc0, shfl_xor4(acc0, 32));
        if (d < npx && slot == 0) {
            if (MULTIPASS) {
                float4 *t = reinterpret_cast<float4 *>(tile) + d * 8 + j;   // same lane owns it in every pass
                if (first_pass) *t = acc0; else { float4 o = *t; add4(o, acc0); *t = o; }
            } else {
                *reinterpret_cast<float4 *>(gv_base + (long long)d * row_stride) = acc0;
            }
        }
    }
}

template <bool MULTIPASS>
__global__ __launch_bounds__(kSBlock) void bwd_value_d32_kernel(
    const float *__restrict__ grad_out, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ level_start, const float *__restrict__ loc,
    const float *__restrict__ attn, int S, int M, int L, int Lq, int P, int W, int tp_cap,
    int np_chunk, float *__restrict__ grad_value)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS: [tile: tp_cap*32 floats if MULTIPASS] [cnt tp_cap] [start tp_cap] [cursor tp_cap] [wsum 16] [rec ...]
    float *tile = reinterpret_cast<float *>(smem);
    int *cnt = reinterpret_cast<int *>(smem + (MULTIPASS ? (size_t)tp_cap * kD * 4 : 0));
    int *start = cnt + tp_cap;
    int *cursor = start + tp_cap;
    int *wsum = cursor + tp_cap;
    SRec *rec = reinterpret_cast<SRec *>(wsum + 16);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int bid = blockIdx.x;
    const int ti = bid % W, l = (bid / W) % L, pr = bid / (W * L);          // uniform -> scalar loads below
    const int H = (int)shapes[2 * l], Wd = (int)shapes[2 * l + 1], lstart = (int)level_start[l];
    const int HW = H * Wd;
    const int px0 = (int)((long long)ti * HW / W), px1 = (int)((long long)(ti + 1) * HW / W);
    const int npx = px1 - px0;
    if (npx <= 0 || npx > tp_cap) return;                                    // empty range (uniform)
    const int b = pr / M, m = pr % M;
    const int NP = Lq * P;
    const long long item_base = (long long)b * Lq * M + m;                   // item(q) = item_base + q*M
    const int row_stride = M * kD;
    const float *go_base = grad_out + item_base * kD + (lane & 7) * 4;
    float *gv_base = grad_value + ((long long)(b * S + lstart + px0) * M + m) * kD + (lane & 7) * 4;

    for (int c0 = 0; c0 < NP; c0 += np_chunk) {
        const int c1 = min(NP, c0 + np_chunk);
        for (int i = tid; i < npx; i += kSBlock) cnt[i] = 0;
        __syncthreads();
        // ---- 1. histogram of taps per row ----
        for (int idx = c0 + tid; idx < c1; idx += kSBlock) {
            const int q = idx / P, p = idx - q * P;
            TapSet t;
            if (point_taps(loc, attn, ((item_base + (long long)q * M) * L + l) * P + p, H, Wd, px0, npx, t)) {
#pragma unroll
                for (int k = 0; k < 4; ++k) if (t.dest[k] >= 0) atomicAdd(&cnt[t.dest[k]], 1);
            }
        }
        __syncthreads();
        // ---- 2. exclusive prefix sum over the rows (512 threads x CH consecutive rows) ----
        const int CH = (npx + kSBlock - 1) / kSBlock;
        const int r0 = tid * CH;
        int mine = 0;
        for (int k = 0; k < CH; ++k) if (r0 + k < npx) mine += cnt[r0 + k];
        int incl = mine;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) { const int y = __shfl_up(incl, o, kWave); if (lane >= o) incl += y; }
        if (lane == kWave - 1) wsum[wave] = incl;
        __syncthreads();
        int excl = incl - mine, total = 0;
        for (int w2 = 0; w2 < kSWaves; ++w2) { const int v = wsum[w2]; if (w2 < wave) excl += v; total += v; }
        for (int k = 0; k < CH; ++k) if (r0 + k < npx) { const int c = cnt[r0 + k]; start[r0 + k] = excl; cursor[r0 + k] = excl; excl += c; }
        __syncthreads();
        // ---- 3. counting sort: records into their row's segment ----
        for (int idx = c0 + tid; idx < c1; idx += kSBlock) {
            const int q = idx / P, p = idx - q * P;
            TapSet t;
            if (point_taps(loc, attn, ((item_base + (long long)q * M) * L + l) * P + p, H, Wd, px0, npx, t)) {
#pragma unroll
                for (int k = 0; k < 4; ++k) if (t.dest[k] >= 0) {
                    SRec r; r.w = t.w[k]; r.q = q;
                    rec[atomicAdd(&cursor[t.dest[k]], 1)] = r;
                }
            }
        }
        __syncthreads();
        // ---- 4. gather; lanes per row chosen from the mean segment length (uniform) ----
        const int mean2 = (2 * total) / npx;                                  // 2 x mean records per row
        const bool first = (c0 == 0);
        if (mean2 <= 3)       gather_rows<1, MULTIPASS>(go_base, gv_base, cnt, start, rec, tile, npx, row_stride, first);
        else if (mean2 <= 7)  gather_rows<2, MULTIPASS>(go_base, gv_base, cnt, start, rec, tile, npx, row_stride, first);
        else if (mean2 <= 15) gather_rows<4, MULTIPASS>(go_base, gv_base, cnt, start, rec, tile, npx, row_stride, first);
        else                  gather_rows<8, MULTIPASS>(go_base, gv_base, cnt, start, rec, tile, npx, row_stride, first);
        __syncthreads();
    }

    if (MULTIPASS) {
        // ---- flush the LDS tile: each row once, coalesced ----
        for (int i = tid; i < npx * 8; i += kSBlock) {
            const int d = i >> 3, jj = i & 7;
            const float4 v = NP > 0 ? reinterpret_cast<const float4 *>(tile)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(grad_value + ((long long)(b * S + lstart + px0 + d) * M + m) * kD + jj * 4) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
bool d32_supported(int N, int S, int M, int D, int L, int Lq, int P)
{
    if (D != kD || L > kMaxLevels || L * P > 32) return false;
    const long long items = (long long)N * Lq * M;
    if ((long long)N * S * M * kD >= (1LL << 31)) return false;        // int32 element offsets
    if (items * L * P * 2 >= (1LL << 31) || items >= (1LL << 30)) return false;
    return true;
}

static int pick_split(int items, int LP)
{
    const int octets = (items + 7) / 8;
    return (octets <= 4096 && LP >= 4) ? 4 : 1;
}

int launch_fwd_d32(const float *value, const int64_t *shapes, const int64_t *level_start,
                   const float *loc, const float *attn, int N, int S, int M, int L, int Lq, int P,
                   float *out, hipStream_t stream)
{
    const int items = N * Lq * M, LP = L * P;
    const int item_stride = LP * kRecBytes + kItemPad;
    if (pick_split(items, LP) == 4) {
        const size_t lds = kLvBytes + 8 * item_stride + 4096;
        hipLaunchKernelGGL(fwd_d32_kernel<4>, dim3((items + 7) / 8), dim3(kBlock), lds, stream, value,
                           shapes, level_start, loc, attn, S, M, L, Lq, P, items, out);
    } else {
        const size_t lds = kLvBytes + 32 * item_stride;
        hipLaunchKernelGGL(fwd_d32_kernel<1>, dim3((items + 31) / 32), dim3(kBlock), lds, stream,
                           value, shapes, level_start, loc, attn, S, M, L, Lq, P, items, out);
    }
    return check_launch("msda forward (d32)");
}

// role-B sizing: single pass while 4*Lq*P records (8 B) fit beside the histogram in 64 KB of LDS
constexpr int kSingleMaxPoints = 1536;      // 48 KB of records
constexpr int kSingleMaxRows = 1280;        // 15 KB of histogram / prefix / cursor
constexpr int kMultiRows = 256;             // 32 KB LDS tile
constexpr int kMultiChunkPoints = 3072;     // 96 KB of records per pass

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

int launch_bwd_d32(const float *grad_out, const float *value, const int64_t *shapes,
                   const int64_t *level_start, const float *loc, const float *attn, int N, int S,
                   int M, int L, int Lq, int P, float *grad_value, float *grad_loc, float *grad_attn,
                   hipStream_t stream)
{
    const int items = N * Lq * M, LP = L * P;
    const int item_stride = LP * kRecBytes + kItemPad;
    // tuning / A-B knobs, read once per process: MSDA_BWD_MODE=atomic selects the v1 global-atomic
    // scatter; MSDA_BWD_WGS the number of role-B workgroups to aim for on small problems.
    static const bool atomic_mode = [] { const char *v = getenv("MSDA_BWD_MODE"); return v && !strcmp(v, "atomic"); }();
    static const int target_wgs = [] { int v = env_int("MSDA_BWD_WGS", 512); return v < 1 ? 1 : v; }();
    const int split = pick_split(items, LP);

    if (atomic_mode) {
        hipError_t e = hipMemsetAsync(grad_value, 0, sizeof(float) * (size_t)N * S * M * kD, stream);
        if (e != hipSuccess) return set_error(MSDA_ERR_LAUNCH, hipGetErrorString(e));
    } else {
        // role B: counting sort + gather, W pixel ranges per level (see the kernel's header)
        const int NP = Lq * P;
        const bool multipass = NP > kSingleMaxPoints;
        const int pairs_levels = N * M * L;
        int W, tp_cap, np_chunk;
        if (!multipass) {
            W = max(ceil_div(S, kSingleMaxRows), ceil_div(target_wgs, pairs_levels));
            W = max(1, min(W, max(1, S / 16)));
            tp_cap = ceil_div(S, W);
            np_chunk = max(NP, 1);
        } else {
            tp_cap = kMultiRows;
            W = ceil_div(S, tp_cap);
            np_chunk = kMultiChunkPoints;
        }
        const long long blocks = (long long)pairs_levels * W;
        if (blocks > 0x7fffffffLL) return set_error(MSDA_ERR_ARGUMENT, "msda backward: too many tiles");
        const size_t lds = (multipass ? (size_t)tp_cap * kD * 4 : 0) + (3 * (size_t)tp_cap + 16) * 4 +
                           (size_t)4 * np_chunk * sizeof(SRec);
        if (multipass) {
            static thread_local size_t granted = 0;
            if (lds > granted) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(bwd_value_d32_kernel<true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return set_error(MSDA_ERR_LAUNCH, hipGetErrorString(e));
                granted = lds;
            }
            hipLaunchKernelGGL(bwd_value_d32_kernel<true>, dim3((unsigned)blocks), dim3(kSBlock), lds, stream,
                               grad_out, shapes, level_start, loc, attn, S, M, L, Lq, P, W, tp_cap, np_chunk, grad_value);
        } else {
            hipLaunchKernelGGL(bwd_value_d32_kernel<false>, dim3((unsigned)blocks), dim3(kSBlock), lds, stream,
                               grad_out, shapes, level_start, loc, attn, S, M, L, Lq, P, W, tp_cap, np_chunk, grad_value);
        }
        if (int rc = check_launch("msda backward (d32, grad_value sort+gather)")) return rc;
    }
    if (split == 4) {
        const size_t lds = kLvBytes + 8 * item_stride + 8 * LP * 16;
        if (atomic_mode)
            hipLaunchKernelGGL((bwd_query_d32_kernel<4, true>), dim3((items + 7) / 8), dim3(kBlock), lds, stream,
                               grad_out, value, shapes, level_start, loc, attn, S, M, L, Lq, P, items,
                               grad_value, grad_loc, grad_attn);
        else
            hipLaunchKernelGGL((bwd_query_d32_kernel<4, false>), dim3((items + 7) / 8), dim3(kBlock), lds, stream,
                               grad_out, value, shapes, level_start, loc, attn, S, M, L, Lq, P, items,
                               grad_value, grad_loc, grad_attn);
    } else {
        const size_t lds = kLvBytes + 32 * item_stride + 32 * LP * 16;
        if (atomic_mode)
            hipLaunchKernelGGL((bwd_query_d32_kernel<1, true>), dim3((items + 31) / 32), dim3(kBlock), lds, stream,
                               grad_out, value, shapes, level_start, loc, attn, S, M, L, Lq, P, items,
                               grad_value, grad_loc, grad_attn);
        else
            hipLaunchKernelGGL((bwd_query_d32_kernel<1, false>), dim3((items + 31) / 32), dim3(kBlock), lds, stream,
                               grad_out, value, shapes, level_start, loc, attn, S, M, L, Lq, P, items,
                               grad_value, grad_loc, grad_attn);
    }
    return check_launch("msda backward (d32, query-major)");
}

}  // namespace msda
