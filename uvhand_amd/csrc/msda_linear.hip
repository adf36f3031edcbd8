// Weight / bias gradient of the nn.Linear layers that bracket the sampling kernel
// (UVHand models/ops/modules/ms_deform_attn.py:96,100,101,139: value_proj, sampling_offsets,
// attention_weights, output_proj):
//
//     dW[N, K] = dY[M, N]^T . X[M, K]          db[N] = sum_m dY[m, :]
//
// M is the number of rows that went through the layer (N_batch*Lq = 600 at the 300-query decoder
// shape), N, K <= a few hundred.  Stock PyTorch-ROCm sends this to hipBLASLt, which runs it as ONE
// 256x256 macro-tile on one CU: 141 us at M = 600, N = K = 256 (0.6 TFLOP/s, tools/gemm_baseline.py) —
// three of these are ~60 % of the module's fwd+bwd time.  The forward GEMM and the input gradient are
// fine in the library (38-97 TFLOP/s) and stay there.
//
// Here: fp32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, no reduced precision), split over
// the reduction dimension M so that a small output still fills the chip, and a fixed-order second
// stage so the result is bitwise reproducible (no float atomics):
//   stage 1  grid (N/64 * K/64 tiles, SPLITS): each workgroup owns a 64x64 tile of dW and a chunk of
//            rows; both operands are row-major with the reduction index as the row, so 32-row stages
//            are straight coalesced copies into LDS (At[kk][i], Bs[kk][j]) and every MFMA operand
//            read is 32 consecutive floats (conflict-free ds_read_b32); 4 wavefronts = 2x2 sub-tiles
//            of 32x32; next stage's global loads are issued before the current stage's MFMAs.
//            Column sums of dY (the bias gradient) ride along in the tiles with k0 == 0.
//   stage 2  sums the SPLITS partial slabs in split order (skipped when SPLITS == 1).
//
// Padding mask (modules/ms_deform_attn.py:97-98: value.masked_fill(padding_mask[..., None], 0) after
// value_proj): instead of two full passes over [N, S, 256] (the masked_fill and its backward), the forward
// zeroes only the masked ROWS of the GEMM output in place (zero_masked_rows_kernel), and the backward passes
// the row mask to stage 1, which stages zeros for masked rows of dY — the weight and bias gradients of the
// masked product without materialising it; the input gradient's masked rows are zeroed the same way.
#include "msda_common.h"
#include "msda_launch.h"

namespace msda {

constexpr int kWgTile = 64;        // output tile is kWgTile x kWgTile
constexpr int kWgStage = 32;       // reduction rows per LDS stage (64 measured: no faster)
constexpr int kWgBlock = 256;      // 4 wavefronts
using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ __launch_bounds__(kWgBlock) void linear_wgrad_partial_kernel(
    const float *__restrict__ dY, const float *__restrict__ X, const uint8_t *__restrict__ row_mask, int M, int N, int K,
    int chunk, long long slab, float *__restrict__ out_w, float *__restrict__ out_b)
{
    // `slab` = elements between consecutive splits' partial results (0 when there is one split and the
    // results go straight to dW / db)
    __shared__ __attribute__((aligned(16))) float At[kWgStage][kWgTile];
    __shared__ __attribute__((aligned(16))) float Bs[kWgStage][kWgTile];
    const int tiles_k = (K + kWgTile - 1) / kWgTile;
    const int n0 = ((int)blockIdx.x / tiles_k) * kWgTile, k0 = ((int)blockIdx.x % tiles_k) * kWgTile;
    const int split = blockIdx.y;
    const int m_begin = split * chunk, m_end = min(M, m_begin + chunk);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i0 = (wave >> 1) * 32, j0 = (wave & 1) * 32, h = lane >> 5, c = lane & 31;

    // global -> register staging: kWgStage rows x 64 floats per operand = kLd float4 per thread and operand
    constexpr int kLd = kWgStage / 16;
    const int lrow = tid >> 4, lcol = (tid & 15) * 4;                 // rows lrow, lrow + 16, ...
    const bool a_ok = n0 + lcol < N, b_ok = k0 + lcol < K;            // N, K are multiples of 4 (host check)
    float4 ra[kLd], rb[kLd];
    uint8_t dead[kLd];                                                 // row is masked: stage zeros for it
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_stage = [&](int m0) {
#pragma unroll
        for (int r = 0; r < kLd; ++r) {
            const int m = m0 + lrow + 16 * r;
            // the mask byte travels with the row loads (no dependent load) and is applied at the LDS store
            dead[r] = (row_mask && m < m_end) ? row_mask[m] : (uint8_t)0;
            ra[r] = (a_ok && m < m_end) ? *reinterpret_cast<const float4 *>(dY + (long long)m * N + n0 + lcol) : zero;
            rb[r] = (b_ok && m < m_end) ? *reinterpret_cast<const float4 *>(X + (long long)m * K + k0 + lcol) : zero;
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;                                                   // column sum of dY (threads 0..63, k0 == 0 tiles)
    const bool do_bias = out_b != nullptr && k0 == 0;

    load_stage(m_begin);
    for (int m0 = m_begin; m0 < m_end; m0 += kWgStage) {
        __syncthreads();                                                // previous stage's reads are done
#pragma unroll
        for (int r = 0; r < kLd; ++r) {
            *reinterpret_cast<float4 *>(&At[lrow + 16 * r][lcol]) = dead[r] ? zero : ra[r];
            *reinterpret_cast<float4 *>(&Bs[lrow + 16 * r][lcol]) = rb[r];
        }
        __syncthreads();
        if (m0 + kWgStage < m_end) load_stage(m0 + kWgStage);          // in flight during the MFMAs below
#pragma unroll
        for (int s = 0; s < kWgStage / 2; ++s)                          // lane half h supplies reduction row 2s + h
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(At[2 * s + h][i0 + c], Bs[2 * s + h][j0 + c], acc, 0, 0, 0);
        if (do_bias && tid < kWgTile) {
#pragma unroll
            for (int kk = 0; kk < kWgStage; ++kk) bsum += At[kk][tid];
        }
    }

    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
    float *ow = out_w + (long long)split * slab;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = n0 + i0 + (r & 3) + 8 * (r >> 2) + 4 * h, j = k0 + j0 + c;
        if (i < N && j < K) ow[(long long)i * K + j] = acc[r];
    }
    if (do_bias && tid < kWgTile && n0 + tid < N) out_b[(long long)split * slab + n0 + tid] = bsum;
}

// Fixed-order sum of the partial slabs (each slab = [N*K weight partials][N bias partials]):
// out[e] = part[0][e] + part[1][e] + ...; elements below nw go to dW, the rest to db.
__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(const float *__restrict__ part, int splits, long long nw,
                                                                   long long n, float *__restrict__ dW,
                                                                   float *__restrict__ db)
{
    const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= n || (e >= nw && db == nullptr)) return;                   // nw and n are multiples of 4
    float4 s = *reinterpret_cast<const float4 *>(part + e);
    for (int k = 1; k < splits; ++k) {
        const float4 v = *reinterpret_cast<const float4 *>(part + (long long)k * n + e);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4 *>(e < nw ? dW + e : db + (e - nw)) = s;
}

// x[r, :] = 0 for every row with mask[r] != 0; one wavefront per row, float4 per lane.  Unmasked rows
// cost one byte of mask.
__global__ __launch_bounds__(256) void zero_masked_rows_kernel(float *__restrict__ x, const uint8_t *__restrict__ mask,
                                                                long long rows, int cols)
{
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows || mask[r] == 0) return;
    float4 *row = reinterpret_cast<float4 *>(x + r * cols);
    for (int c = threadIdx.x & 63; c < cols / 4; c += 64) row[c] = make_float4(0.f, 0.f, 0.f, 0.f);
}

int launch_zero_masked_rows(float *x, const uint8_t *mask, long long rows, int cols, hipStream_t stream)
{
    if (rows == 0) return MSDA_OK;
    hipLaunchKernelGGL(zero_masked_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, mask, rows, cols);
    return check_launch("msda zero masked rows");
}

static int wgrad_splits(int M, int N, int K)
{
    const int tiles = ((N + kWgTile - 1) / kWgTile) * ((K + kWgTile - 1) / kWgTile);
    int splits = (512 + tiles - 1) / tiles;                             // aim at ~512 workgroups
    const int max_splits = (M + 2 * kWgStage - 1) / (2 * kWgStage);     // at least two stages per chunk
    if (splits > max_splits) splits = max_splits;
    return splits < 1 ? 1 : splits;
}

size_t linear_wgrad_workspace_bytes(int M, int N, int K)
{
    const int splits = wgrad_splits(M, N, K);
    return splits <= 1 ? 0 : sizeof(float) * (size_t)splits * ((size_t)N * K + (size_t)N);
}

int launch_linear_wgrad(const float *dY, const float *X, const uint8_t *row_mask, int M, int N, int K, float *dW, float *db,
                        float *workspace, hipStream_t stream)
{
    const int splits = wgrad_splits(M, N, K);
    int chunk = (M + splits - 1) / splits;
    chunk = ((chunk + kWgStage - 1) / kWgStage) * kWgStage;
    const int tiles = ((N + kWgTile - 1) / kWgTile) * ((K + kWgTile - 1) / kWgTile);
    const dim3 grid((unsigned)tiles, (unsigned)splits);
    if (splits == 1) {
        hipLaunchKernelGGL(linear_wgrad_partial_kernel, grid, dim3(kWgBlock), 0, stream, dY, X, row_mask, M, N, K, chunk, 0LL, dW,
                           db);
        return check_launch("msda linear wgrad");
    }
    if (workspace == nullptr) return set_error(MSDA_ERR_ARGUMENT, "msda linear wgrad: workspace required");
    const long long nw = (long long)N * K, slab = nw + N;               // per split: weight partials, then bias partials
    hipLaunchKernelGGL(linear_wgrad_partial_kernel, grid, dim3(kWgBlock), 0, stream, dY, X, row_mask, M, N, K, chunk, slab,
                       workspace,
                       db ? workspace + nw : nullptr);
    if (int rc = check_launch("msda linear wgrad (partial)")) return rc;
    hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((unsigned)((slab / 4 + 255) / 256)), dim3(256), 0, stream, workspace,
                       splits, nw, slab, dW, db);
    return check_launch("msda linear wgrad (reduce)");
}

}  // namespace msda
