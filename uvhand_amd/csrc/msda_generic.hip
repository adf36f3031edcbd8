// Generic multi-scale deformable attention kernels: any channel count D, any
// (L, P), float or double — or bf16 rows (value / out / grad_out as bf16 bits,
// fp32 arithmetic, fp32 grad_value) for the shapes the D = 32 family does not take.  One 64-lane wavefront owns one (batch, query, head)
// item; lanes stride over the D channels, so a tap is read as contiguous
// 4*64-byte (or 8*64-byte) runs and the three per-point gradients are reduced
// across the wavefront with cross-lane shuffles — no LDS, no barriers.
//
// This is the family that serves the reference test's channel sweep
// (UVHand models/ops/test.py:85: D in {30, 64, 71, 1025, 2048, 3096}) and the
// fp64 gradcheck; the model shape (D = 32, fp32) takes the tiled kernels in
// msda_d32.hip instead.  It replaces, with one kernel per direction, the
// reference's forward kernel (ms_deform_im2col_cuda.cuh:237-299) and its six
// backward variants (:301-920), whose selection by D (:975-1319) is not needed
// when lanes stride over channels.
#include "msda_common.h"
#include "msda_launch.h"

namespace msda {

constexpr int kGenericBlock = 256;                       // 4 wavefronts = 4 items per block
constexpr int kGenericItemsPerBlock = kGenericBlock / kWave;

// One row element: T itself, or bf16 bits widened to / rounded from float (round-to-nearest-even, like the D = 32 family).
template <typename T, typename VT>
__device__ __forceinline__ T ld_elem(const VT *p)
{
    if constexpr (sizeof(VT) == sizeof(T)) return *p;
    else return __uint_as_float((uint32_t)*p << 16);
}
template <typename T, typename VT>
__device__ __forceinline__ void st_elem(VT *p, T v)
{
    if constexpr (sizeof(VT) == sizeof(T)) *p = v;
    else *p = __builtin_bit_cast(unsigned short, static_cast<__bf16>(v));
}

template <typename T, typename VT>
__global__ __launch_bounds__(kGenericBlock) void fwd_generic_kernel(
    const VT *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ level_start, const T *__restrict__ loc,
    const T *__restrict__ attn, int S, int M, int D, int L, int Lq, int P, long long items,
    VT *__restrict__ out)
{
    const int lane = threadIdx.x & (kWave - 1);
    long long item = (long long)blockIdx.x * kGenericItemsPerBlock + (threadIdx.x >> 6);
    if (item >= items) return;                            // whole wavefront leaves together
    const int m = (int)(item % M);
    const long long b = item / ((long long)M * Lq);
    const int LP = L * P;
    const T *lp = loc + item * LP * 2;
    const T *ap = attn + item * LP;
    const long long ws = (long long)M * D;                // elements between horizontally adjacent pixels

    for (int c0 = 0; c0 < D; c0 += kWave) {
        const int c = c0 + lane;
        const bool live = c < D;
        T acc = 0;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            if (!level_fits(shapes[2 * l], shapes[2 * l + 1], level_start[l], S)) continue;
            const VT *v = value + ((b * S + level_start[l]) * M + m) * D + (live ? c : 0);
            for (int p = 0; p < P; ++p) {
                const int k = l * P + p;
                const PointGeom<T> g = point_geom<T>(lp[2 * k], lp[2 * k + 1], H, W);
                if (!g.inside) continue;                  // wave-uniform: loc is per item
                const T a = ap[k];
                const T hh = 1 - g.lh, hw = 1 - g.lw;
                const long long r0 = ((long long)g.h0 * W + g.w0) * ws;
                const long long r1 = r0 + (long long)W * ws;
                T v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                if (live) {
                    if (g.ok00) v1 = ld_elem<T>(v + r0);
                    if (g.ok01) v2 = ld_elem<T>(v + r0 + ws);
                    if (g.ok10) v3 = ld_elem<T>(v + r1);
                    if (g.ok11) v4 = ld_elem<T>(v + r1 + ws);
                }
                acc += (hh * hw * v1 + hh * g.lw * v2 + g.lh * hw * v3 + g.lh * g.lw * v4) * a;
            }
        }
        if (live) st_elem<T>(out + item * D + c, acc);
    }
}

// Backward: grad_value by hardware float atomics (the destination rows are shared
// between items), grad_sampling_loc / grad_attn_weight by a wavefront reduction over
// the channels, written once per point (zero for a point outside the map, as the
// reference's shared-memory variants produce, ms_deform_im2col_cuda.cuh:365-393).
// grad_value must be zero on entry (the launcher enqueues the memset).
// SCATTER = false (deterministic mode): grad_value is left to bwd_generic_value_det_kernel below.
template <typename T, typename VT, bool SCATTER>
__global__ __launch_bounds__(kGenericBlock) void bwd_generic_kernel(
    const VT *__restrict__ grad_out, const VT *__restrict__ value,
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const T *__restrict__ loc, const T *__restrict__ attn, int S, int M, int D, int L, int Lq,
    int P, long long items, T *__restrict__ grad_value, T *__restrict__ grad_loc,
    T *__restrict__ grad_attn)
{
    const int lane = threadIdx.x & (kWave - 1);
    long long item = (long long)blockIdx.x * kGenericItemsPerBlock + (threadIdx.x >> 6);
    if (item >= items) return;
    const int m = (int)(item % M);
    const long long b = item / ((long long)M * Lq);
    const int LP = L * P;
    const T *lp = loc + item * LP * 2;
    const T *ap = attn + item * LP;
    const VT *go = grad_out + item * D;
    T *gl = grad_loc + item * LP * 2;
    T *ga = grad_attn + item * LP;
    const long long ws = (long long)M * D;

    for (int l = 0; l < L; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const long long lvl = ((b * S + level_start[l]) * M + m) * D;
        const bool fits = level_fits(shapes[2 * l], shapes[2 * l + 1], level_start[l], S);
        for (int p = 0; p < P; ++p) {
            const int k = l * P + p;
            const PointGeom<T> g = point_geom<T>(lp[2 * k], lp[2 * k + 1], H, W);
            T s_attn = 0, s_x = 0, s_y = 0;
            if (g.inside && fits) {
                const T a = ap[k];
                const T hh = 1 - g.lh, hw = 1 - g.lw;
                const T k1 = hh * hw, k2 = hh * g.lw, k3 = g.lh * hw, k4 = g.lh * g.lw;
                const long long r0 = lvl + ((long long)g.h0 * W + g.w0) * ws;
                const long long r1 = r0 + (long long)W * ws;
                for (int c = lane; c < D; c += kWave) {
                    const T top = ld_elem<T>(go + c), tv = top * a;
                    T v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                    if (g.ok00) { v1 = ld_elem<T>(value + r0 + c);      if (SCATTER) atomicAdd(grad_value + r0 + c, k1 * tv); }
                    if (g.ok01) { v2 = ld_elem<T>(value + r0 + ws + c); if (SCATTER) atomicAdd(grad_value + r0 + ws + c, k2 * tv); }
                    if (g.ok10) { v3 = ld_elem<T>(value + r1 + c);      if (SCATTER) atomicAdd(grad_value + r1 + c, k3 * tv); }
                    if (g.ok11) { v4 = ld_elem<T>(value + r1 + ws + c); if (SCATTER) atomicAdd(grad_value + r1 + ws + c, k4 * tv); }
                    s_attn += top * (k1 * v1 + k2 * v2 + k3 * v3 + k4 * v4);
                    s_x += (hh * (v2 - v1) + g.lh * (v4 - v3)) * tv;    // d/dw of the bilinear form
                    s_y += (hw * (v3 - v1) + g.lw * (v4 - v2)) * tv;    // d/dh
                }
                s_attn = wave_sum(s_attn);
                s_x = wave_sum(s_x) * (T)W;                // pixel -> normalised coordinate
                s_y = wave_sum(s_y) * (T)H;
            }
            if (lane == 0) { gl[2 * k] = s_x; gl[2 * k + 1] = s_y; ga[k] = s_attn; }
        }
    }
}

// Deterministic grad_value for any D (MSDA_FLAG_DETERMINISTIC outside the D = 32 family; the reference has nothing like it:
// every one of its backward variants ends in atomicAdd, ms_deform_im2col_cuda.cuh:125-152, 845-920).  Destination-major and
// brute force: one wavefront owns one (batch, pixel, head) row; its lanes test 64 sampling points of the pixel's level at
// a time against the pixel, and every hit — taken in (query, point) order, so the sum's association is a pure function of
// the inputs — adds weight * attention * grad_out row into the lanes' channel accumulators.  Work is rows x Lq*P point tests:
// fine for the shapes that take this family (test sweeps, odd D), and it needs neither a zero-fill nor scratch.
constexpr int kDetChunk = 4;                              // channels per lane and pass: 256 channels per sweep over the points
template <typename T, typename VT>
__global__ __launch_bounds__(kGenericBlock) void bwd_generic_value_det_kernel(
    const VT *__restrict__ grad_out, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const T *__restrict__ loc, const T *__restrict__ attn, int S, int M, int D, int L, int Lq, int P, long long rows,
    T *__restrict__ grad_value)
{
    const int lane = threadIdx.x & (kWave - 1);
    // grid-stride over the rows: a launch holds fewer than 2^32 threads, a 2048 x 4096 map with 8 heads has 2^26 rows
    for (long long row = (long long)blockIdx.x * kGenericItemsPerBlock + (threadIdx.x >> 6); row < rows;
         row += (long long)gridDim.x * kGenericItemsPerBlock) {   // (whole wavefronts leave together)
    const int m = (int)(row % M);
    const int sp = (int)((row / M) % S);
    const long long b = row / ((long long)M * S);
    T *gv = grad_value + row * D;
    int l = -1, H = 0, W = 0, pix = 0;
    for (int k = 0; k < L && l < 0; ++k) {                // the (first) level that covers this pixel
        const long long hk = shapes[2 * k], wk = shapes[2 * k + 1], st = level_start[k];
        if (level_fits(hk, wk, st, S) && sp >= st && sp < st + hk * wk) { l = k; H = (int)hk; W = (int)wk; pix = sp - (int)st; }
    }
    if (l < 0) {                                          // a pixel no level covers: zeros (include/msda.h)
        for (int c = lane; c < D; c += kWave) gv[c] = 0;
        continue;
    }
    const int h = pix / W, w = pix - h * W, LP = L * P;
    const long long NP = (long long)Lq * P;
    for (int c0 = 0; c0 < D; c0 += kWave * kDetChunk) {
        T acc[kDetChunk];
#pragma unroll
        for (int k = 0; k < kDetChunk; ++k) acc[k] = 0;
        for (long long base = 0; base < NP; base += kWave) {
            const long long idx = base + lane;
            bool hit = false;
            T wt = 0, a = 0;
            long long q = 0;
            if (idx < NP) {
                q = idx / P;
                const long long e = ((b * Lq + q) * M + m) * LP + (long long)l * P + (idx - q * P);
                const PointGeom<T> g = point_geom<T>(loc[2 * e], loc[2 * e + 1], H, W);
                const int dh = h - g.h0, dw = w - g.w0;   // the pixel is inside the map: a matching tap is a valid one
                if (g.inside && (dh == 0 || dh == 1) && (dw == 0 || dw == 1)) {
                    hit = true;
                    wt = (dh ? g.lh : 1 - g.lh) * (dw ? g.lw : 1 - g.lw);
                    a = attn[e];
                }
            }
            unsigned long long mask = __ballot(hit);
            while (mask) {                                // uniform loop: hits in point order
                const int src = __builtin_ctzll(mask);
                mask &= mask - 1;
                const long long qs = __shfl(q, src, kWave);
                const T ws = __shfl(wt, src, kWave), as = __shfl(a, src, kWave);
                const VT *go = grad_out + ((b * Lq + qs) * M + m) * D;
#pragma unroll
                for (int k = 0; k < kDetChunk; ++k) {
                    const int c = c0 + k * kWave + lane;
                    if (c < D) acc[k] += ws * (ld_elem<T>(go + c) * as);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < kDetChunk; ++k) {
            const int c = c0 + k * kWave + lane;
            if (c < D) gv[c] = acc[k];
        }
    }
    }
}

template <typename T, typename VT>
int launch_fwd_generic(const VT *value, const int64_t *shapes, const int64_t *level_start,
                       const T *loc, const T *attn, int N, int S, int M, int D, int L, int Lq, int P,
                       VT *out, hipStream_t stream)
{
    const long long items = (long long)N * Lq * M;
    const long long blocks = (items + kGenericItemsPerBlock - 1) / kGenericItemsPerBlock;
    if (blocks * kGenericBlock > 0xffffffffLL) return set_error(MSDA_ERR_ARGUMENT, "N*Lq*M too large for one launch");
    hipLaunchKernelGGL((fwd_generic_kernel<T, VT>), dim3((unsigned)blocks), dim3(kGenericBlock), 0, stream,
                       value, shapes, level_start, loc, attn, S, M, D, L, Lq, P, items, out);
    return check_launch("msda forward (generic)");
}

template <typename T, typename VT>
int launch_bwd_generic(const VT *grad_out, const VT *value, const int64_t *shapes,
                       const int64_t *level_start, const T *loc, const T *attn, int N, int S, int M,
                       int D, int L, int Lq, int P, T *grad_value, T *grad_loc, T *grad_attn,
                       hipStream_t stream, bool deterministic)
{
    const long long items = (long long)N * Lq * M;
    const long long blocks = (items + kGenericItemsPerBlock - 1) / kGenericItemsPerBlock;
    if (blocks * kGenericBlock > 0xffffffffLL) return set_error(MSDA_ERR_ARGUMENT, "N*Lq*M too large for one launch");
    if (deterministic) {
        const long long rows = (long long)N * S * M;
        // brute force (rows x Lq*P point tests): bounded, so that a flag set for the whole program (torch.use_deterministic_
        // algorithms) cannot silently turn an encoder-sized fp64 / odd-D call into seconds of work — refuse it like a framework
        // refuses an op that has no deterministic form (include/msda.h, MSDA_FLAG_DETERMINISTIC)
        if ((double)rows * (double)Lq * (double)P > 68719476736.0)       // 2^36 point tests, ~50 ms
            return set_error(MSDA_ERR_ARGUMENT, "msda backward: MSDA_FLAG_DETERMINISTIC outside the D = 32 family tests every sampling point "
                                                "against every pixel row (N*S*M x Lq*P > 2^36 here): use D = 32 fp32 / bf16 tensors or clear the flag");
        long long rblocks = (rows + kGenericItemsPerBlock - 1) / kGenericItemsPerBlock;
        if (rblocks > (1LL << 22)) rblocks = 1LL << 22;                  // the kernel strides over the rows
        hipLaunchKernelGGL((bwd_generic_kernel<T, VT, false>), dim3((unsigned)blocks), dim3(kGenericBlock), 0, stream,
                           grad_out, value, shapes, level_start, loc, attn, S, M, D, L, Lq, P, items,
                           grad_value, grad_loc, grad_attn);
        if (int rc = check_launch("msda backward (generic, grad_loc / grad_attn)")) return rc;
        hipLaunchKernelGGL((bwd_generic_value_det_kernel<T, VT>), dim3((unsigned)rblocks), dim3(kGenericBlock), 0, stream,
                           grad_out, shapes, level_start, loc, attn, S, M, D, L, Lq, P, rows, grad_value);
        return check_launch("msda backward (generic, deterministic grad_value)");
    }
    hipError_t e = hipMemsetAsync(grad_value, 0, sizeof(T) * (size_t)N * S * M * D, stream);
    if (e != hipSuccess) return set_error(MSDA_ERR_LAUNCH, hipGetErrorString(e));
    hipLaunchKernelGGL((bwd_generic_kernel<T, VT, true>), dim3((unsigned)blocks), dim3(kGenericBlock), 0, stream,
                       grad_out, value, shapes, level_start, loc, attn, S, M, D, L, Lq, P, items,
                       grad_value, grad_loc, grad_attn);
    return check_launch("msda backward (generic)");
}

#define MSDA_GENERIC_INST(T, VT)                                                                                        \
    template int launch_fwd_generic<T, VT>(const VT *, const int64_t *, const int64_t *, const T *, const T *, int, int, int, \
                                           int, int, int, int, VT *, hipStream_t);                                            \
    template int launch_bwd_generic<T, VT>(const VT *, const VT *, const int64_t *, const int64_t *, const T *, const T *,    \
                                           int, int, int, int, int, int, int, T *, T *, T *, hipStream_t, bool);
MSDA_GENERIC_INST(float, float)
MSDA_GENERIC_INST(double, double)
MSDA_GENERIC_INST(float, uint16_t)       // bf16 rows, fp32 grad_value
#undef MSDA_GENERIC_INST

}  // namespace msda
