/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's multi-scale deformable
 * attention forward / backward.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library, and only as the checker.
 * The product (uvhand_amd/) never links, imports or calls it.
 *
 * Parity pin: checked against golden vectors produced by importing the
 * reference's own pure-PyTorch fallback in the build container
 * (tests/golden/gen_golden.py -> tests/golden/ npz files; see tests/test_oracle.py).
 *
 * What each function follows (paths relative to /root/reference/):
 *   sample_fwd_*   models/ops/src/cuda/ms_deform_im2col_cuda.cuh:33-84
 *                  (4 guarded taps, zero padding, w1..w4 = hh*hw, hh*lw, lh*hw, lh*lw)
 *   forward        models/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299
 *                  (h_im = loc_y*H - 0.5, w_im = loc_x*W - 0.5, open-interval
 *                   test at :288, sum over levels then points)
 *                  == functions/ms_deform_attn_func.py:42-62
 *                  (grid_sample bilinear / zeros / align_corners=False)
 *   backward       models/ops/src/cuda/ms_deform_im2col_cuda.cuh:87-159
 *                  (per-tap grad_value accumulation, grad_h_weight/grad_w_weight,
 *                   W* / H* scaling of grad_sampling_loc at :157-158) and the
 *                  per-(b,q,m) channel reduction of :301-403.
 *
 * Layouts (row-major, contiguous), as the reference reads them:
 *   value[N,S,M,D]  shapes[L,2] int64 (H,W)  level_start[L] int64
 *   loc[N,Lq,M,L,P,2] (x,y) normalised   attn[N,Lq,M,L,P]   out[N,Lq,M*D]
 *
 * Parallelism (OpenMP, optional): forward over (b,q,m) items; backward over
 * (b,m) pairs, each of which owns grad_value[b,:,m,:] exclusively, so the
 * accumulation order -- and therefore the result -- is fixed (q, then l, then p,
 * then tap), unlike the reference's atomicAdd order.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int msda_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void msda_oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

#define DEFINE_ORACLE(T, SUF, FLOOR)                                                         \
                                                                                             \
void msda_oracle_forward_##SUF(const T *value, const int64_t *shapes,                        \
                               const int64_t *level_start, const T *loc, const T *attn,      \
                               int N, int S, int M, int D, int L, int Lq, int P, T *out)     \
{                                                                                            \
    const int64_t items = (int64_t)N * Lq * M;                                               \
    _Pragma("omp parallel for schedule(static)")                                             \
    for (int64_t it = 0; it < items; ++it) {                                                 \
        const int m = (int)(it % M);                                                         \
        const int b = (int)(it / ((int64_t)M * Lq));                                         \
        T *o = out + it * D;                                                                 \
        for (int c = 0; c < D; ++c) o[c] = 0;                                                \
        const T *lp = loc + it * L * P * 2;                                                  \
        const T *ap = attn + it * L * P;                                                     \
        for (int l = 0; l < L; ++l) {                                                        \
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                    \
            const T *v = value + ((int64_t)b * S + level_start[l]) * M * D;                  \
            const int64_t ws = (int64_t)M * D, hs = (int64_t)W * ws;                         \
            for (int p = 0; p < P; ++p, lp += 2, ap += 1) {                                  \
                const T w_im = lp[0] * W - (T)0.5;                                           \
                const T h_im = lp[1] * H - (T)0.5;                                           \
                if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) continue;             \
                const int h0 = (int)FLOOR(h_im), w0 = (int)FLOOR(w_im);                      \
                const int h1 = h0 + 1, w1 = w0 + 1;                                          \
                const T lh = h_im - h0, lw = w_im - w0, hh = 1 - lh, hw = 1 - lw;            \
                const T k1 = hh * hw, k2 = hh * lw, k3 = lh * hw, k4 = lh * lw;              \
                const int ok1 = (h0 >= 0 && w0 >= 0), ok2 = (h0 >= 0 && w1 <= W - 1);        \
                const int ok3 = (h1 <= H - 1 && w0 >= 0), ok4 = (h1 <= H - 1 && w1 <= W - 1);\
                const T a = ap[0];                                                           \
                for (int c = 0; c < D; ++c) {                                                \
                    const int64_t base = (int64_t)m * D + c;                                 \
                    const T v1 = ok1 ? v[h0 * hs + w0 * ws + base] : 0;                      \
                    const T v2 = ok2 ? v[h0 * hs + w1 * ws + base] : 0;                      \
                    const T v3 = ok3 ? v[h1 * hs + w0 * ws + base] : 0;                      \
                    const T v4 = ok4 ? v[h1 * hs + w1 * ws + base] : 0;                      \
                    o[c] += (k1 * v1 + k2 * v2 + k3 * v3 + k4 * v4) * a;                     \
                }                                                                            \
            }                                                                                \
        }                                                                                    \
    }                                                                                        \
}                                                                                            \
                                                                                             \
void msda_oracle_backward_##SUF(const T *grad_out, const T *value, const int64_t *shapes,    \
                                const int64_t *level_start, const T *loc, const T *attn,     \
                                int N, int S, int M, int D, int L, int Lq, int P,            \
                                T *grad_value, T *grad_loc, T *grad_attn)                    \
{                                                                                            \
    memset(grad_value, 0, sizeof(T) * (size_t)N * S * M * D);                                \
    const int pairs = N * M;                                                                 \
    _Pragma("omp parallel for schedule(dynamic, 1)")                                         \
    for (int pr = 0; pr < pairs; ++pr) {                                                     \
        const int b = pr / M, m = pr % M;                                                    \
        for (int q = 0; q < Lq; ++q) {                                                       \
            const int64_t it = ((int64_t)b * Lq + q) * M + m;                                \
            const T *go = grad_out + it * D;                                                 \
            const T *lp = loc + it * L * P * 2;                                              \
            const T *ap = attn + it * L * P;                                                 \
            T *gl = grad_loc + it * L * P * 2;                                               \
            T *ga = grad_attn + it * L * P;                                                  \
            for (int l = 0; l < L; ++l) {                                                    \
                const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                \
                const int64_t lvl = ((int64_t)b * S + level_start[l]) * M * D;               \
                const T *v = value + lvl;                                                    \
                T *gv = grad_value + lvl;                                                    \
                const int64_t ws = (int64_t)M * D, hs = (int64_t)W * ws;                     \
                for (int p = 0; p < P; ++p, lp += 2, ap += 1, gl += 2, ga += 1) {            \
                    gl[0] = 0; gl[1] = 0; ga[0] = 0;                                         \
                    const T w_im = lp[0] * W - (T)0.5;                                       \
                    const T h_im = lp[1] * H - (T)0.5;                                       \
                    if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) continue;         \
                    const int h0 = (int)FLOOR(h_im), w0 = (int)FLOOR(w_im);                  \
                    const int h1 = h0 + 1, w1 = w0 + 1;                                      \
                    const T lh = h_im - h0, lw = w_im - w0, hh = 1 - lh, hw = 1 - lw;        \
                    const T k1 = hh * hw, k2 = hh * lw, k3 = lh * hw, k4 = lh * lw;          \
                    const int ok1 = (h0 >= 0 && w0 >= 0), ok2 = (h0 >= 0 && w1 <= W - 1);    \
                    const int ok3 = (h1 <= H - 1 && w0 >= 0);                                \
                    const int ok4 = (h1 <= H - 1 && w1 <= W - 1);                            \
                    const T a = ap[0];                                                       \
                    T s_attn = 0, s_x = 0, s_y = 0;                                          \
                    for (int c = 0; c < D; ++c) {                                            \
                        const int64_t base = (int64_t)m * D + c;                             \
                        const T top = go[c], tv = top * a;                                   \
                        T gh = 0, gw = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;                    \
                        if (ok1) { const int64_t i = h0 * hs + w0 * ws + base; v1 = v[i];    \
                                   gh -= hw * v1; gw -= hh * v1; gv[i] += k1 * tv; }         \
                        if (ok2) { const int64_t i = h0 * hs + w1 * ws + base; v2 = v[i];    \
                                   gh -= lw * v2; gw += hh * v2; gv[i] += k2 * tv; }         \
                        if (ok3) { const int64_t i = h1 * hs + w0 * ws + base; v3 = v[i];    \
                                   gh += hw * v3; gw -= lh * v3; gv[i] += k3 * tv; }         \
                        if (ok4) { const int64_t i = h1 * hs + w1 * ws + base; v4 = v[i];    \
                                   gh += lw * v4; gw += lh * v4; gv[i] += k4 * tv; }         \
                        s_attn += top * (k1 * v1 + k2 * v2 + k3 * v3 + k4 * v4);             \
                        s_x += W * gw * tv;                                                  \
                        s_y += H * gh * tv;                                                  \
                    }                                                                        \
                    gl[0] = s_x; gl[1] = s_y; ga[0] = s_attn;                                \
                }                                                                            \
            }                                                                                \
        }                                                                                    \
    }                                                                                        \
}

DEFINE_ORACLE(float, f32, floorf)
DEFINE_ORACLE(double, f64, floor)
