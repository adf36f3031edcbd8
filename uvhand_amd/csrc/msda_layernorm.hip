// Residual add + LayerNorm of the transformer layers around the op, fp32, for MI355X (gfx950).
//
// The reference's encoder / decoder layers follow every attention and FFN block with
//     x = x + dropout(x2);  x = norm(x)                (UVHand models/arctic_transformer.py:279-282, 294-295, 366-368,
//                                                       377-378, 385-386)
// which stock PyTorch runs as an add kernel (read 2, write 1) and a LayerNorm kernel (read 1, write 1) — and the
// LayerNorm backward re-reads the sum.  Here:
//
//   forward   y = LayerNorm(x + r) * gamma + beta   one wavefront per row: lane i holds channels 4i..4i+3 (+256k), the
//             row is read once (x and r, coalesced float4), mean and variance are wavefront reductions over registers
//             (two-pass, like the framework's rowwise moments), y / mean / rstd are written; the sum itself is NOT
//             stored — the backward recomputes it from x and r, which autograd keeps alive anyway.
//   backward  ds = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma   (ds is the gradient of BOTH x and r),
//             dgamma / dbeta as per-workgroup partial column sums in registers, combined by a second fixed-order
//             stage: no float atomics, bitwise reproducible.
//
// Dropout (training, p > 0) stays PyTorch's own F.dropout on r before the call, so the Philox stream and the masks are
// exactly the framework's; only the add and the normalisation are fused.
#include "msda_common.h"
#include "msda_launch.h"

namespace msda {

constexpr int kLnBlock = 256;                     // 4 wavefronts = 4 rows in flight per workgroup
constexpr int kLnWaves = kLnBlock / kWave;
constexpr int kLnMaxVec = 4;                      // float4 per lane: d <= 1024

__device__ __forceinline__ float4 ld_f4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

template <int NV>
__global__ __launch_bounds__(kLnBlock) void add_layernorm_fwd_kernel(
    const float *__restrict__ x, const float *__restrict__ res, const float *__restrict__ gamma,
    const float *__restrict__ beta, long long rows, int d, float eps, float *__restrict__ y,
    float *__restrict__ mean_out, float *__restrict__ rstd_out)
{
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const long long row = (long long)blockIdx.x * kLnWaves + wave;
    if (row >= rows) return;                                             // whole wavefront leaves together
    const float *xr = x + row * d, *rr = res ? res + row * d : nullptr;
    float4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = (k * kWave + lane) * 4;
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < d) {
            v[k] = ld_f4(xr + c);
            if (rr) { const float4 t = ld_f4(rr + c); v[k].x += t.x; v[k].y += t.y; v[k].z += t.z; v[k].w += t.w; }
            sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        }
    }
    const float mean = wave_sum(sum) / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = (k * kWave + lane) * 4;
        if (c < d) {
            const float a = v[k].x - mean, b = v[k].y - mean, e = v[k].z - mean, f = v[k].w - mean;
            sq += (a * a + b * b) + (e * e + f * f);
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)d + eps);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = (k * kWave + lane) * 4;
        if (c < d) {
            const float4 g = ld_f4(gamma + c), b = ld_f4(beta + c);
            float4 o;
            o.x = (v[k].x - mean) * rstd * g.x + b.x; o.y = (v[k].y - mean) * rstd * g.y + b.y;
            o.z = (v[k].z - mean) * rstd * g.z + b.z; o.w = (v[k].w - mean) * rstd * g.w + b.w;
            *reinterpret_cast<float4 *>(y + row * d + c) = o;
        }
    }
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// Workgroup w owns the rows [w * rows_per_wg, (w+1) * rows_per_wg); its 4 wavefronts take them round-robin and keep
// the dgamma / dbeta column sums of their rows in registers; partial[w][2][d] is the workgroup's total.
template <int NV>
__global__ __launch_bounds__(kLnBlock) void add_layernorm_bwd_kernel(
    const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ res,
    const float *__restrict__ gamma, const float *__restrict__ mean_in, const float *__restrict__ rstd_in,
    long long rows, int d, int rows_per_wg, float *__restrict__ ds, float *__restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) float red[kLnWaves][2][kLnMaxVec * kWave * 4];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const long long r0 = (long long)blockIdx.x * rows_per_wg;
    const long long r1 = r0 + rows_per_wg < rows ? r0 + rows_per_wg : rows;
    float4 gm[NV], dg[NV], db[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = (k * kWave + lane) * 4;
        gm[k] = c < d ? ld_f4(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        dg[k] = db[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (long long row = r0 + wave; row < r1; row += kLnWaves) {
        const float mean = mean_in[row], rstd = rstd_in[row];
        const float *xr = x + row * d, *rr = res ? res + row * d : nullptr, *gr = dy + row * d;
        float4 xh[NV], g[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = (k * kWave + lane) * 4;
            xh[k] = g[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < d) {
                float4 v = ld_f4(xr + c);
                if (rr) { const float4 t = ld_f4(rr + c); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
                const float4 o = ld_f4(gr + c);
                xh[k] = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
                g[k] = make_float4(o.x * gm[k].x, o.y * gm[k].y, o.z * gm[k].z, o.w * gm[k].w);
                s1 += (g[k].x + g[k].y) + (g[k].z + g[k].w);
                s2 += (g[k].x * xh[k].x + g[k].y * xh[k].y) + (g[k].z * xh[k].z + g[k].w * xh[k].w);
                dg[k].x += o.x * xh[k].x; dg[k].y += o.y * xh[k].y; dg[k].z += o.z * xh[k].z; dg[k].w += o.w * xh[k].w;
                db[k].x += o.x; db[k].y += o.y; db[k].z += o.z; db[k].w += o.w;
            }
        }
        const float c1 = wave_sum(s1) / (float)d, c2 = wave_sum(s2) / (float)d;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = (k * kWave + lane) * 4;
            if (c < d) {
                float4 o;
                o.x = rstd * (g[k].x - c1 - xh[k].x * c2); o.y = rstd * (g[k].y - c1 - xh[k].y * c2);
                o.z = rstd * (g[k].z - c1 - xh[k].z * c2); o.w = rstd * (g[k].w - c1 - xh[k].w * c2);
                *reinterpret_cast<float4 *>(ds + row * d + c) = o;
            }
        }
    }
    // the 4 wavefronts' column sums, combined in a fixed order
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        *reinterpret_cast<float4 *>(&red[wave][0][(k * kWave + lane) * 4]) = dg[k];
        *reinterpret_cast<float4 *>(&red[wave][1][(k * kWave + lane) * 4]) = db[k];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * d; i += kLnBlock) {
        const int which = i >= d, c = which ? i - d : i;
        float t = red[0][which][c];
#pragma unroll
        for (int w = 1; w < kLnWaves; ++w) t += red[w][which][c];
        partial[((long long)blockIdx.x * 2 + which) * d + c] = t;
    }
}

// dgamma[c] / dbeta[c] = sum over the workgroups' partials: a 256-thread workgroup takes 16 columns, 16 threads per
// column each summing every 16th partial (independent load chains), combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void layernorm_param_reduce_kernel(const float *__restrict__ partial, int nwg, int d,
                                                                    float *__restrict__ dgamma, float *__restrict__ dbeta)
{
    __shared__ float red[16][17];
    const int cl = (int)threadIdx.x & 15, part = (int)threadIdx.x >> 4;
    const int col = (int)blockIdx.x * 16 + cl, which = (int)blockIdx.y;
    float t0 = 0.f, t1 = 0.f;
    if (col < d) {
        int w = part;
        for (; w + 16 < nwg; w += 32) {
            t0 += partial[((long long)w * 2 + which) * d + col];
            t1 += partial[((long long)(w + 16) * 2 + which) * d + col];
        }
        if (w < nwg) t0 += partial[((long long)w * 2 + which) * d + col];
    }
    red[part][cl] = t0 + t1;
    __syncthreads();
    if (part == 0 && col < d) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][cl];
        (which ? dbeta : dgamma)[col] = s;
    }
}

static int ln_bwd_wgs(long long rows)
{
    long long w = (rows + 15) / 16;                   // at least ~16 rows per workgroup
    if (w > 1024) w = 1024;
    return (int)(w < 1 ? 1 : w);
}

size_t add_layernorm_workspace_bytes(long long rows, int d) { return (size_t)ln_bwd_wgs(rows) * 2 * d * sizeof(float); }

int launch_add_layernorm_fwd(const float *x, const float *res, const float *gamma, const float *beta, long long rows, int d,
                             float eps, float *y, float *mean, float *rstd, hipStream_t stream)
{
    const dim3 grid((unsigned)((rows + kLnWaves - 1) / kLnWaves)), block(kLnBlock);
#define MSDA_LN_F(NV) hipLaunchKernelGGL((add_layernorm_fwd_kernel<NV>), grid, block, 0, stream, x, res, gamma, beta, rows, d, eps, y, mean, rstd)
    if (d <= 256) MSDA_LN_F(1); else if (d <= 512) MSDA_LN_F(2); else MSDA_LN_F(4);
#undef MSDA_LN_F
    return check_launch("msda add+layernorm forward");
}

int launch_add_layernorm_bwd(const float *dy, const float *x, const float *res, const float *gamma, const float *mean,
                             const float *rstd, long long rows, int d, float *ds, float *dgamma, float *dbeta, float *workspace,
                             hipStream_t stream)
{
    const int nwg = ln_bwd_wgs(rows);
    const int rows_per_wg = (int)((rows + nwg - 1) / nwg);
#define MSDA_LN_B(NV) hipLaunchKernelGGL((add_layernorm_bwd_kernel<NV>), dim3(nwg), dim3(kLnBlock), 0, stream, dy, x, res, gamma, mean, rstd, rows, d, rows_per_wg, ds, workspace)
    if (d <= 256) MSDA_LN_B(1); else if (d <= 512) MSDA_LN_B(2); else MSDA_LN_B(4);
#undef MSDA_LN_B
    if (int rc = check_launch("msda add+layernorm backward")) return rc;
    hipLaunchKernelGGL(layernorm_param_reduce_kernel, dim3((d + 15) / 16, 2), dim3(256), 0, stream, workspace, nwg, d, dgamma, dbeta);
    return check_launch("msda layernorm parameter gradients");
}

// ---- FFN of the layers (models/arctic_transformer.py:283-287, :366-370): linear2(dropout(relu(linear1(x)))) ----------------
// Backward of dropout(relu(h)) in ONE pass, in place on the incoming gradient: with a = dropout(relu(h)) = relu(h) * keep / (1 - p)
// (what the forward saved as linear2's input anyway), a > 0 exactly where h > 0 and the element was kept, so
//     grad_h = grad_a * scale * (a > 0),   scale = 1 / (1 - p)   (1 without dropout)
// needs neither the dropout mask nor relu's output: PyTorch runs masked_scale (grad, mask -> tmp) and threshold_backward
// (tmp, relu output -> grad_h), five tensor passes over [rows, d_ffn] against three here.
__global__ __launch_bounds__(256) void relu_dropout_bwd_kernel(float *__restrict__ grad, const float *__restrict__ act, float scale,
                                                                long long n4)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 a = reinterpret_cast<const float4 *>(act)[i];
        float4 g = reinterpret_cast<float4 *>(grad)[i];
        g.x = a.x > 0.f ? g.x * scale : 0.f; g.y = a.y > 0.f ? g.y * scale : 0.f;
        g.z = a.z > 0.f ? g.z * scale : 0.f; g.w = a.w > 0.f ? g.w * scale : 0.f;
        reinterpret_cast<float4 *>(grad)[i] = g;
    }
}

int launch_relu_dropout_bwd(float *grad, const float *act, float scale, long long n, hipStream_t stream)
{
    const long long n4 = n / 4;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;                                    // grid-stride: ~32 workgroups per CU at most
    hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, grad, act, scale, n4);
    return check_launch("msda relu+dropout backward");
}

}  // namespace msda
