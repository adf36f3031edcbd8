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
// backward (v1): query-major; grad_value by global float atomics
// ------------------------------------------------------------------------------------------
template <int SPLIT>
__global__ __launch_bounds__(kBlock) void bwd_d32_kernel(
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
    float *gvb = grad_value + j * 4;
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (item0 + il < items) g4 = ld4(grad_out + (long long)(item0 + il) * kD + j * 4);

#pragma unroll 2
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
        if (off.x >= 0) { const float c = k1 * a; float *d = gvb + off.x;
            atomicAdd(d, c * g4.x); atomicAdd(d + 1, c * g4.y); atomicAdd(d + 2, c * g4.z); atomicAdd(d + 3, c * g4.w); }
        if (off.y >= 0) { const float c = k2 * a; float *d = gvb + off.y;
            atomicAdd(d, c * g4.x); atomicAdd(d + 1, c * g4.y); atomicAdd(d + 2, c * g4.z); atomicAdd(d + 3, c * g4.w); }
        if (off.z >= 0) { const float c = k3 * a; float *d = gvb + off.z;
            atomicAdd(d, c * g4.x); atomicAdd(d + 1, c * g4.y); atomicAdd(d + 2, c * g4.z); atomicAdd(d + 3, c * g4.w); }
        if (off.w >= 0) { const float c = k4 * a; float *d = gvb + off.w;
            atomicAdd(d, c * g4.x); atomicAdd(d + 1, c * g4.y); atomicAdd(d + 2, c * g4.z); atomicAdd(d + 3, c * g4.w); }
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

int launch_bwd_d32(const float *grad_out, const float *value, const int64_t *shapes,
                   const int64_t *level_start, const float *loc, const float *attn, int N, int S,
                   int M, int L, int Lq, int P, float *grad_value, float *grad_loc, float *grad_attn,
                   hipStream_t stream)
{
    const int items = N * Lq * M, LP = L * P;
    const int item_stride = LP * kRecBytes + kItemPad;
    hipError_t e = hipMemsetAsync(grad_value, 0, sizeof(float) * (size_t)N * S * M * kD, stream);
    if (e != hipSuccess) return set_error(MSDA_ERR_LAUNCH, hipGetErrorString(e));
    if (pick_split(items, LP) == 4) {
        const size_t lds = kLvBytes + 8 * item_stride + 8 * LP * 16;
        hipLaunchKernelGGL(bwd_d32_kernel<4>, dim3((items + 7) / 8), dim3(kBlock), lds, stream,
                           grad_out, value, shapes, level_start, loc, attn, S, M, L, Lq, P, items,
                           grad_value, grad_loc, grad_attn);
    } else {
        const size_t lds = kLvBytes + 32 * item_stride + 32 * LP * 16;
        hipLaunchKernelGGL(bwd_d32_kernel<1>, dim3((items + 31) / 32), dim3(kBlock), lds, stream,
                           grad_out, value, shapes, level_start, loc, attn, S, M, L, Lq, P, items,
                           grad_value, grad_loc, grad_attn);
    }
    return check_launch("msda backward (d32)");
}

}  // namespace msda
