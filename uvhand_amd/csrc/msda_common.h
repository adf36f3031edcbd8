// Shared device helpers for the MI355X (gfx950) multi-scale deformable attention kernels.
//
// Geometry of one sampling point, exactly as the reference computes it
// (UVHand models/ops/src/cuda/ms_deform_im2col_cuda.cuh:285-288 for the pixel
// coordinate and the open-interval test, :38-45 / :97-104 for floor / fractions,
// :56-78 for the four tap guards).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace msda {

constexpr int kWave = 64;          // CDNA wavefront width (hard-coded on purpose)
constexpr int kMaxLevels = 16;     // fast-path limit (level table lives in LDS)

template <typename F>
struct PointGeom {
    int h0, w0;        // top-left tap (may be -1)
    F lh, lw;          // fractional parts: weights of the +1 taps
    bool inside;       // passes the reference's (-1, H) x (-1, W) test
    bool ok00, ok01, ok10, ok11;   // tap guards (h0,w0) (h0,w1) (h1,w0) (h1,w1)
};

template <typename F>
__device__ __forceinline__ F floor_t(F x);
template <>
__device__ __forceinline__ float floor_t<float>(float x) { return floorf(x); }
template <>
__device__ __forceinline__ double floor_t<double>(double x) { return floor(x); }

template <typename F>
__device__ __forceinline__ PointGeom<F> point_geom(F loc_x, F loc_y, int H, int W)
{
    PointGeom<F> g;
    const F h_im = loc_y * (F)H - (F)0.5;
    const F w_im = loc_x * (F)W - (F)0.5;
    g.inside = (h_im > (F)-1) && (w_im > (F)-1) && (h_im < (F)H) && (w_im < (F)W);
    const F hf = floor_t<F>(h_im), wf = floor_t<F>(w_im);
    // NaN / out-of-range coordinates never reach the integer conversion's result:
    // every consumer tests `inside` first.
    g.h0 = g.inside ? (int)hf : 0;
    g.w0 = g.inside ? (int)wf : 0;
    g.lh = g.inside ? h_im - hf : (F)0;
    g.lw = g.inside ? w_im - wf : (F)0;
    const bool h0ok = g.h0 >= 0, h1ok = g.h0 + 1 <= H - 1;
    const bool w0ok = g.w0 >= 0, w1ok = g.w0 + 1 <= W - 1;
    g.ok00 = g.inside && h0ok && w0ok;
    g.ok01 = g.inside && h0ok && w1ok;
    g.ok10 = g.inside && h1ok && w0ok;
    g.ok11 = g.inside && h1ok && w1ok;
    return g;
}

// A level is usable only if its H*W pixels lie inside the S pixel rows of a batch element.  Every kernel treats the
// points of a level that does not fit as outside the map, so nothing is ever read or written beyond the tensors,
// whatever spatial_shapes / level_start_index hold (the reference trusts them: ms_deform_im2col_cuda.cuh:274-283).
__device__ __forceinline__ bool level_fits(long long H, long long W, long long start, int S)
{
    return H > 0 && W > 0 && start >= 0 && H <= (long long)S && W <= (long long)S && start + H * W <= (long long)S;
}

// Sum over the 64 lanes of a wavefront; every lane gets the total.
template <typename F>
__device__ __forceinline__ F wave_sum(F x)
{
#pragma unroll
    for (int s = kWave / 2; s > 0; s >>= 1) x += __shfl_xor(x, s, kWave);
    return x;
}

}  // namespace msda
