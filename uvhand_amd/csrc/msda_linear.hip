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
//   stage 1  1-D grid of (N/64 * K/64 tiles) x SPLITS workgroups (~3 per CU): each owns a 64x64 tile of dW
//            and a chunk of rows; both operands are row-major with the reduction index as the row, so
//            32-row stages are straight coalesced copies into LDS (At[kk][i], Bs[kk][j]) and every MFMA
//            operand read is 32 consecutive floats; 4 wavefronts = 2x2 sub-tiles of 32x32; the stage's
//            32 operand reads are issued ahead of its 16 MFMAs; LDS is double-buffered (the stage loaded
//            during the MFMAs is stored behind them, one barrier per stage) and the global loads run two
//            stages ahead.  Column sums of dY (the bias gradient) ride along in the tiles with k0 == 0.
//   stage 2  sums the SPLITS partial slabs in a fixed association (skipped when SPLITS == 1).
// 9.0 us at M = 600 (N = K = 256), 21.5 us at M = 6120, 53.7 us at M = 33440 (82 TFLOP/s; hipBLASLt + the
// bias reduction: 152 / 62 / 170 us) — tools/wgrad_time.py.
//
// Padding mask (modules/ms_deform_attn.py:97-98: value.masked_fill(padding_mask[..., None], 0) after
// value_proj): instead of two full passes over [N, S, 256] (the masked_fill and its backward), the forward
// zeroes only the masked ROWS of the GEMM output in place (zero_masked_rows_kernel), and the backward passes
// the row mask to stage 1, which stages zeros for masked rows of dY — the weight and bias gradients of the
// masked product without materialising it; the input gradient's masked rows are zeroed the same way.
#include <cstdlib>

#include "msda_common.h"
#include "msda_launch.h"

namespace msda {

// (output tile, M-split) of a workgroup in the 1-D grid of tiles*splits.  All the tiles of one split read the
// same rows of dY and X (each row 4x at N = K = 256 with 64-wide tiles): numbered naively they land on
// different XCDs (workgroup id mod 8) and every XCD's L2 fetches those rows from HBM again — the kernel was
// bandwidth-bound at ~4.7 TB/s that way.  With splits a multiple of 8, XCD x takes the splits x, x+8, ...
// whole, so a split's rows are fetched into ONE L2 and its other tiles hit there.
__device__ __forceinline__ void tile_and_split(int id, int tiles, int splits, int &tile, int &split)
{
    if ((splits & 7) == 0) {
        const int x = id & 7, slot = id >> 3;
        split = (slot / tiles) * 8 + x;
        tile = slot % tiles;
    } else {
        tile = id % tiles;
        split = id / tiles;
    }
}

// v or zeros, component-wise selects (a float4 ?: makes the compiler select between two addresses and
// park both operands in scratch memory)
__device__ __forceinline__ float4 keep4(const float4 &v, bool keep)
{
    return make_float4(keep ? v.x : 0.f, keep ? v.y : 0.f, keep ? v.z : 0.f, keep ? v.w : 0.f);
}

constexpr int kWgTile = 64;        // output tile is kWgTile x kWgTile
constexpr int kWgStage = 32;       // reduction rows per LDS stage (64 measured: no faster)
constexpr int kWgBlock = 256;      // 4 wavefronts
#ifndef MSDA_BF_STAGE
#define MSDA_BF_STAGE 32            // reduction rows per LDS stage of the bf16-MFMA kernel (64 measured: no faster)
#endif
using f32x16 = __attribute__((ext_vector_type(16))) float;

// Operand storage: float, or bfloat16 bits (uint16_t) widened to fp32 on the way into LDS — the products and the
// accumulation are the same exact fp32 MFMA chain either way (bf16 operands under autocast: half the operand bytes,
// and a weight gradient that is NOT rounded to bf16 before it reaches the fp32 master parameter).
template <typename OT> __device__ __forceinline__ float4 ld_operand4(const OT *p);
template <> __device__ __forceinline__ float4 ld_operand4<float>(const float *p) { return *reinterpret_cast<const float4 *>(p); }
template <> __device__ __forceinline__ float4 ld_operand4<uint16_t>(const uint16_t *p)
{
    const uint2 u = *reinterpret_cast<const uint2 *>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}

template <typename OT>
__global__ __launch_bounds__(kWgBlock) void linear_wgrad_partial_kernel(
    const OT *__restrict__ dY, const OT *__restrict__ X, const uint8_t *__restrict__ row_mask, int M, int N, int K,
    int chunk, int tiles, int splits, long long slab, float *__restrict__ out_w, float *__restrict__ out_b)
{
    // `slab` = elements between consecutive splits' partial results (0 when there is one split and the
    // results go straight to dW / db)
    __shared__ __attribute__((aligned(16))) float At[2][kWgStage][kWgTile];
    __shared__ __attribute__((aligned(16))) float Bs[2][kWgStage][kWgTile];
    const int tiles_k = (K + kWgTile - 1) / kWgTile;
    int tile, split;
    tile_and_split((int)blockIdx.x, tiles, splits, tile, split);
    const int n0 = (tile / tiles_k) * kWgTile, k0 = (tile % tiles_k) * kWgTile;
    const int m_begin = split * chunk, m_end = min(M, m_begin + chunk);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i0 = (wave >> 1) * 32, j0 = (wave & 1) * 32, h = lane >> 5, c = lane & 31;

    // global -> register staging: kWgStage rows x 64 floats per operand = kLd float4 per thread and operand
    constexpr int kLd = kWgStage / 16, kRowStep = 16;
    const int lrow = tid >> 4, lcol = (tid & 15) * 4;                 // rows lrow, lrow + 16, ...
    const bool a_ok = n0 + lcol < N, b_ok = k0 + lcol < K;            // N, K are multiples of 4 (host check)
    float4 ra[kLd], rb[kLd];
    unsigned mk[kLd];                                                  // mask byte of each staged row (0 = live)
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    // (a macro, not a lambda: capturing ra / rb by reference parks them in scratch memory)
#define MSDA_LOAD_STAGE(m0_)                                                                                        \
    do {                                                                                                            \
        _Pragma("unroll") for (int r = 0; r < kLd; ++r) {                                                           \
            const int m = (m0_) + lrow + kRowStep * r;                                                              \
            /* the mask byte travels with the row loads (nothing waits on it) and is applied at the LDS store */    \
            mk[r] = 0;                                                                                              \
            if (row_mask != nullptr && m < m_end) mk[r] = row_mask[m];                                              \
            ra[r] = zero; rb[r] = zero;                                                                             \
            if (a_ok && m < m_end) ra[r] = ld_operand4<OT>(dY + (long long)m * N + n0 + lcol);                      \
            if (b_ok && m < m_end) rb[r] = ld_operand4<OT>(X + (long long)m * K + k0 + lcol);                       \
        }                                                                                                           \
    } while (0)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // bias gradient (k0 == 0 tiles): thread t sums column t & 63 over the 8 stage rows 8*(t >> 6) ...; the four
    // partial sums of a column are combined through LDS at the end (fixed association)
    float bsum = 0.f;
    const bool do_bias = out_b != nullptr && k0 == 0;
    const int bcol = tid & (kWgTile - 1), brow = (tid >> 6) * (kWgStage / 4);

    // LDS is double-buffered: the stage being multiplied and the next one being filled, ONE barrier per stage
    MSDA_LOAD_STAGE(m_begin);
#pragma unroll
    for (int r = 0; r < kLd; ++r) {
        *reinterpret_cast<float4 *>(&At[0][lrow + 16 * r][lcol]) = keep4(ra[r], mk[r] == 0);
        *reinterpret_cast<float4 *>(&Bs[0][lrow + 16 * r][lcol]) = rb[r];
    }
    MSDA_LOAD_STAGE(m_begin + kWgStage);                                // rows past m_end load as zeros
    __syncthreads();
    int cur = 0;
    for (int m0 = m_begin; m0 < m_end; m0 += kWgStage) {
        // all operand reads of the stage first, then the MFMA chain: the LDS latency is paid once per stage,
        // not once per MFMA (lane half h supplies reduction row 2s + h)
        float av[kWgStage / 2], bv[kWgStage / 2];
#pragma unroll
        for (int s = 0; s < kWgStage / 2; ++s) { av[s] = At[cur][2 * s + h][i0 + c]; bv[s] = Bs[cur][2 * s + h][j0 + c]; }
        if (do_bias) {
#pragma unroll
            for (int kk = 0; kk < kWgStage / 4; ++kk) bsum += At[cur][brow + kk][bcol];
        }
        __builtin_amdgcn_sched_barrier(0);               // keep the reads ahead of the chain (the scheduler sinks them back)
#pragma unroll
        for (int s = 0; s < kWgStage / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0);
        // the next stage (loaded while the MFMAs ran) goes into the other buffer, whose last readers passed
        // the barrier of the previous iteration; then the loads of the stage after it are issued
        if (m0 + kWgStage < m_end) {
#pragma unroll
            for (int r = 0; r < kLd; ++r) {
                *reinterpret_cast<float4 *>(&At[cur ^ 1][lrow + 16 * r][lcol]) = keep4(ra[r], mk[r] == 0);
                *reinterpret_cast<float4 *>(&Bs[cur ^ 1][lrow + 16 * r][lcol]) = rb[r];
            }
            if (m0 + 2 * kWgStage < m_end) MSDA_LOAD_STAGE(m0 + 2 * kWgStage);
        }
        __syncthreads();
        cur ^= 1;
    }

    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
    float *ow = out_w + (long long)split * slab;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = n0 + i0 + (r & 3) + 8 * (r >> 2) + 4 * h, j = k0 + j0 + c;
        if (i < N && j < K) ow[(long long)i * K + j] = acc[r];
    }
    if (do_bias) {                                                      // uniform per workgroup
        float *bpart = &At[0][0][0];                                    // the stage buffers are free now (barrier above)
        bpart[tid] = bsum;
        __syncthreads();
        if (tid < kWgTile && n0 + tid < N)
            out_b[(long long)split * slab + n0 + tid] = (bpart[tid] + bpart[tid + 64]) + (bpart[tid + 128] + bpart[tid + 192]);
    }
}

#undef MSDA_LOAD_STAGE

// ---- few rows: 32 x 32 output tiles, 16x16x4 MFMA --------------------------------------------------------------------------
// At M = 600 the kernel above is 128 workgroups of three stages each plus the reduction launch: 9 us however it is split.
// Here a workgroup owns a 32 x 32 tile (64 of them at N = K = 256), each wavefront a 16 x 16 block whose MFMA chain is a
// quarter as long per row, and ~512 workgroups in all: M = 600 7.5 us (8.9), 2400 11.6 (15.3) including the
// reduction.  (One workgroup per tile walking ALL 600 rows, no slabs and no second launch: 10.0 us — ten stages whose loads
// are two stages ahead of sixteen short MFMAs each is a latency chain; kept only below 128 rows.)  Stages of 64 rows, LDS rows of 48 floats (the four lane groups of a 16x16x4 operand read rows 4s .. 4s+3: with
// 48 the two groups of a 32-lane phase fall on disjoint banks).  fp32 operands.
constexpr int kSmTile = 32, kSmStage = 64, kSmRow = 48;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(kWgBlock) void linear_wgrad_small_kernel(
    const float *__restrict__ dY, const float *__restrict__ X, const uint8_t *__restrict__ row_mask, int M, int N, int K,
    int chunk, int tiles, int splits, long long slab, float *__restrict__ out_w, float *__restrict__ out_b)
{
    __shared__ __attribute__((aligned(16))) float At[2][kSmStage][kSmRow];
    __shared__ __attribute__((aligned(16))) float Bs[2][kSmStage][kSmRow];
    const int tiles_k = (K + kSmTile - 1) / kSmTile;
    int tile, split;
    tile_and_split((int)blockIdx.x, tiles, splits, tile, split);
    const int n0 = (tile / tiles_k) * kSmTile, k0 = (tile % tiles_k) * kSmTile;
    const int m_begin = split * chunk, m_end = min(M, m_begin + chunk);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i0 = (wave >> 1) * 16, j0 = (wave & 1) * 16, g = lane >> 4, c = lane & 15;

    // staging: 64 rows x 32 floats per operand = 2 float4 per thread and operand (rows lrow, lrow + 32)
    const int lrow = tid >> 3, lcol = (tid & 7) * 4;
    const bool a_ok = n0 + lcol < N, b_ok = k0 + lcol < K;
    float4 ra[2], rb[2];
    unsigned mk[2];
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#define MSDA_LOAD_SM(m0_)                                                                                           \
    do {                                                                                                            \
        _Pragma("unroll") for (int r = 0; r < 2; ++r) {                                                             \
            const int m = (m0_) + lrow + 32 * r;                                                                    \
            mk[r] = 0;                                                                                              \
            if (row_mask != nullptr && m < m_end) mk[r] = row_mask[m];                                              \
            ra[r] = zero; rb[r] = zero;                                                                             \
            if (a_ok && m < m_end) ra[r] = *reinterpret_cast<const float4 *>(dY + (long long)m * N + n0 + lcol);    \
            if (b_ok && m < m_end) rb[r] = *reinterpret_cast<const float4 *>(X + (long long)m * K + k0 + lcol);     \
        }                                                                                                           \
    } while (0)
#define MSDA_STORE_SM(buf_)                                                                                         \
    do {                                                                                                            \
        _Pragma("unroll") for (int r = 0; r < 2; ++r) {                                                             \
            *reinterpret_cast<float4 *>(&At[buf_][lrow + 32 * r][lcol]) = keep4(ra[r], mk[r] == 0);                 \
            *reinterpret_cast<float4 *>(&Bs[buf_][lrow + 32 * r][lcol]) = rb[r];                                    \
        }                                                                                                           \
    } while (0)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // bias gradient (k0 == 0 tiles): thread t sums column t & 31 over the 8 stage rows 8 (t >> 5) ...
    float bsum = 0.f;
    const bool do_bias = out_b != nullptr && k0 == 0;
    const int bcol = tid & 31, brow = (tid >> 5) * 8;

    MSDA_LOAD_SM(m_begin);
    MSDA_STORE_SM(0);
    MSDA_LOAD_SM(m_begin + kSmStage);
    __syncthreads();
    int cur = 0;
    for (int m0 = m_begin; m0 < m_end; m0 += kSmStage) {
        float av[kSmStage / 4], bv[kSmStage / 4];
#pragma unroll
        for (int s = 0; s < kSmStage / 4; ++s) { av[s] = At[cur][4 * s + g][i0 + c]; bv[s] = Bs[cur][4 * s + g][j0 + c]; }
        if (do_bias) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) bsum += At[cur][brow + kk][bcol];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < kSmStage / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc, 0, 0, 0);
        if (m0 + kSmStage < m_end) {
            MSDA_STORE_SM(cur ^ 1);
            if (m0 + 2 * kSmStage < m_end) MSDA_LOAD_SM(m0 + 2 * kSmStage);
        }
        __syncthreads();
        cur ^= 1;
    }
#undef MSDA_LOAD_SM
#undef MSDA_STORE_SM
    // C/D layout of the 16x16 MFMA: column = lane & 15, row = 4 * (lane >> 4) + reg
    float *ow = out_w + (long long)split * slab;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = n0 + i0 + 4 * g + r, j = k0 + j0 + c;
        if (i < N && j < K) ow[(long long)i * K + j] = acc[r];
    }
    if (do_bias) {
        float *bpart = &At[0][0][0];                                    // the stage buffers are free now (barrier above)
        bpart[tid] = bsum;
        __syncthreads();
        if (tid < kSmTile && n0 + tid < N) {
            float t = bpart[tid];
#pragma unroll
            for (int q = 1; q < 8; ++q) t += bpart[tid + 32 * q];
            out_b[(long long)split * slab + n0 + tid] = t;
        }
    }
}

// ---- bf16 operands on the bf16 MFMA (v_mfma_f32_32x32x16_bf16: 16x the rate of the f32-input form the kernel above uses) ----
// Under autocast dY and X arrive as bf16.  A product of two bf16 values is exact in fp32 and the MFMA accumulates in fp32, so
// this loses nothing the widening kernel keeps; it only stops spending 15/16 of the matrix cores' time.
// Both operands have the REDUCTION index as their row ([M, N] and [M, K] row-major) while the MFMA wants, per lane, 8
// consecutive reduction steps of one output row / column — a transposed read.  The stage is therefore kept in LDS as 4-row x
// 16-column blocks of 128 B ([m / 4][col / 16][4][16] bf16), which is exactly what one 16-lane group of ds_read_b64_tr_b16
// consumes: the hardware delivers each lane its column of the block's four rows.  A 32-lane half reads two neighbouring
// blocks = 256 contiguous bytes: conflict-free.  Rows of blocks are skewed by 16 B so that the global->LDS copy (8 lanes = 8
// rows of one 16-byte column chunk) does not land on one bank group.
constexpr int kBfStage = MSDA_BF_STAGE;                    // reduction rows per LDS stage
constexpr int kBfRows = kBfStage / 32;                     // rows per thread and stage of the global -> LDS copy
constexpr int kBfBlkRow = (kWgTile / 16) * 128 + 16;       // bytes per row of blocks: 4 blocks + skew
constexpr int kBfOperand = (kBfStage / 4) * kBfBlkRow;     // bytes per operand and buffer
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using i16x4 = __attribute__((ext_vector_type(4))) short;
using i16x8 = __attribute__((ext_vector_type(8))) short;

__device__ __forceinline__ i16x4 lds_read_tr16(const unsigned char *p)
{
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3))) *)p);
}

__global__ __launch_bounds__(kWgBlock) void linear_wgrad_partial_bf16_kernel(
    const uint16_t *__restrict__ dY, const uint16_t *__restrict__ X, const uint8_t *__restrict__ row_mask, int M, int N, int K,
    int chunk, int tiles, int splits, long long slab, float *__restrict__ out_w, float *__restrict__ out_b)
{
    __shared__ __attribute__((aligned(16))) unsigned char Ab[2][kBfOperand];
    __shared__ __attribute__((aligned(16))) unsigned char Bb[2][kBfOperand];
    const int tiles_k = (K + kWgTile - 1) / kWgTile;
    int tile, split;
    tile_and_split((int)blockIdx.x, tiles, splits, tile, split);
    const int n0 = (tile / tiles_k) * kWgTile, k0 = (tile % tiles_k) * kWgTile;
    const int m_begin = split * chunk, m_end = min(M, m_begin + chunk);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i0 = (wave >> 1) * 32, j0 = (wave & 1) * 32, h = lane >> 5, c = lane & 31;

    // global -> registers: 16-byte chunks (8 bf16) of kBfRows rows per thread and operand; 8 consecutive lanes take 8 rows
    // of the same column chunk, a wavefront 8 whole 128-byte rows (rows lrow, lrow + 32, ...)
    const int lrow = (tid & 7) + 8 * wave, lchunk = (tid >> 3) & 7;       // stage row 0..31, columns 8*lchunk ..
    const bool a_ok = n0 + 8 * lchunk < N, b_ok = k0 + 8 * lchunk < K;    // N, K are multiples of 8 (host check)
    const int st_off = (lrow >> 2) * kBfBlkRow + (lchunk >> 1) * 128 + (lrow & 3) * 32 + (lchunk & 1) * 16;
    uint4 ra[kBfRows], rb[kBfRows];
    unsigned mk[kBfRows];
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
#define MSDA_LOAD_STAGE_BF(m0_)                                                                                      \
    do {                                                                                                            \
        _Pragma("unroll") for (int r = 0; r < kBfRows; ++r) {                                                       \
            const int m = (m0_) + lrow + 32 * r;                                                                    \
            mk[r] = 0;                                                                                              \
            if (row_mask != nullptr && m < m_end) mk[r] = row_mask[m];                                              \
            ra[r] = zero; rb[r] = zero;                                                                             \
            if (a_ok && m < m_end) ra[r] = *reinterpret_cast<const uint4 *>(dY + (long long)m * N + n0 + 8 * lchunk); \
            if (b_ok && m < m_end) rb[r] = *reinterpret_cast<const uint4 *>(X + (long long)m * K + k0 + 8 * lchunk);  \
        }                                                                                                           \
    } while (0)
#define MSDA_STORE_STAGE_BF(buf_)                                                                                    \
    do {                                                                                                            \
        _Pragma("unroll") for (int r = 0; r < kBfRows; ++r) {                                                       \
            const uint4 am = mk[r] == 0 ? ra[r] : zero;                                                             \
            *reinterpret_cast<uint4 *>(&Ab[buf_][st_off + 8 * r * kBfBlkRow]) = make_uint4(am.x, am.y, am.z, am.w); \
            *reinterpret_cast<uint4 *>(&Bb[buf_][st_off + 8 * r * kBfBlkRow]) = rb[r];                              \
            if (do_bias) {                                                                                          \
                bs[0] += __uint_as_float(am.x << 16); bs[1] += __uint_as_float(am.x & 0xffff0000u);                 \
                bs[2] += __uint_as_float(am.y << 16); bs[3] += __uint_as_float(am.y & 0xffff0000u);                 \
                bs[4] += __uint_as_float(am.z << 16); bs[5] += __uint_as_float(am.z & 0xffff0000u);                 \
                bs[6] += __uint_as_float(am.w << 16); bs[7] += __uint_as_float(am.w & 0xffff0000u);                 \
            }                                                                                                       \
        }                                                                                                           \
    } while (0)

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // bias gradient (k0 == 0 tiles): every thread keeps the column sums of ITS 8 columns over the rows it stages
    float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool do_bias = out_b != nullptr && k0 == 0;

    // transposed-read addresses of this lane: block row 2h (+1), column block of output row / column (c >> 4), in-block
    // row (lane & 15) >> 2 and 8-byte piece lane & 3
    const int rd = (2 * h) * kBfBlkRow + ((lane & 15) >> 2) * 32 + (lane & 3) * 8;
    const int a_rd = rd + ((i0 + (c & 16)) >> 4) * 128, b_rd = rd + ((j0 + (c & 16)) >> 4) * 128;

    MSDA_LOAD_STAGE_BF(m_begin);
    MSDA_STORE_STAGE_BF(0);
    MSDA_LOAD_STAGE_BF(m_begin + kBfStage);                             // rows past m_end load as zeros
    __syncthreads();
    int cur = 0;
    for (int m0 = m_begin; m0 < m_end; m0 += kBfStage) {
        const unsigned char *A = Ab[cur], *B = Bb[cur];
        constexpr int KS = kBfStage / 16;                               // MFMAs (k = 16 each) per stage
        i16x4 a[2 * KS], b[2 * KS];
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[2 * s + t] = lds_read_tr16(A + a_rd + (4 * s + t) * kBfBlkRow);
                b[2 * s + t] = lds_read_tr16(B + b_rd + (4 * s + t) * kBfBlkRow);
            }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const i16x8 av = {a[2 * s].x, a[2 * s].y, a[2 * s].z, a[2 * s].w, a[2 * s + 1].x, a[2 * s + 1].y, a[2 * s + 1].z, a[2 * s + 1].w};
            const i16x8 bv = {b[2 * s].x, b[2 * s].y, b[2 * s].z, b[2 * s].w, b[2 * s + 1].x, b[2 * s + 1].y, b[2 * s + 1].z, b[2 * s + 1].w};
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc, 0, 0, 0);
        }
        if (m0 + kBfStage < m_end) {
            MSDA_STORE_STAGE_BF(cur ^ 1);
            if (m0 + 2 * kBfStage < m_end) MSDA_LOAD_STAGE_BF(m0 + 2 * kBfStage);
        }
        __syncthreads();
        cur ^= 1;
    }

    float *ow = out_w + (long long)split * slab;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = n0 + i0 + (r & 3) + 8 * (r >> 2) + 4 * h, j = k0 + j0 + c;
        if (i < N && j < K) ow[(long long)i * K + j] = acc[r];
    }
    if (do_bias) {                                                      // uniform per workgroup
        // thread t holds columns 8*lchunk .. +7 over its rows: the 32 threads of a column chunk are
        // tid = (tid & 7) + 8*lchunk + 64*wave; summed in a fixed order through LDS (the stage buffers are free now)
        float (*bred)[8] = reinterpret_cast<float (*)[8]>(&Ab[0][0]);
        static_assert(2 * kBfOperand >= kWgBlock * 8 * (int)sizeof(float), "bias partials fit the A buffers");
#pragma unroll
        for (int k = 0; k < 8; ++k) bred[tid][k] = bs[k];
        __syncthreads();
        if (tid < kWgTile && n0 + tid < N) {
            const int ch = tid >> 3, k = tid & 7;
            float sum = 0.f;
            for (int w = 0; w < 4; ++w)
                for (int r = 0; r < 8; ++r) sum += bred[r + 8 * ch + 64 * w][k];
            out_b[(long long)split * slab + n0 + tid] = sum;
        }
    }
}
#undef MSDA_LOAD_STAGE_BF
#undef MSDA_STORE_STAGE_BF

// Fixed-order sum of the partial slabs (each slab = [N*K weight partials][N bias partials]):
// out[e] = sum_k part[k][e]; elements below nw go to dW, the rest to db.  A workgroup takes 64 float4 columns;
// its 4 wavefronts each sum a quarter of the splits (4 independent loads in flight per lane), and the four
// partial sums are combined through LDS in wavefront order — the association is fixed, so the result is
// bitwise reproducible, and the chain of dependent loads is splits/4 long instead of splits.
__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(const float *__restrict__ part, int splits, long long nw,
                                                                   long long n, float *__restrict__ dW,
                                                                   float *__restrict__ db)
{
    __shared__ float4 sums[4][64];
    const int g = threadIdx.x >> 6, col = threadIdx.x & 63;
    const long long e = ((long long)blockIdx.x * 64 + col) * 4;
    const bool live = e < n && !(e >= nw && db == nullptr);              // nw and n are multiples of 4
    const int per = (splits + 3) / 4, k0 = g * per, k1 = min(splits, k0 + per);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        int k = k0;
        for (; k + 4 <= k1; k += 4) {
            const float4 v0 = *reinterpret_cast<const float4 *>(part + (long long)k * n + e);
            const float4 v1 = *reinterpret_cast<const float4 *>(part + (long long)(k + 1) * n + e);
            const float4 v2 = *reinterpret_cast<const float4 *>(part + (long long)(k + 2) * n + e);
            const float4 v3 = *reinterpret_cast<const float4 *>(part + (long long)(k + 3) * n + e);
            s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
            s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
            s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
            s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
        }
        for (; k < k1; ++k) {
            const float4 v = *reinterpret_cast<const float4 *>(part + (long long)k * n + e);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    sums[g][col] = s;
    __syncthreads();
    if (g == 0 && live) {
#pragma unroll
        for (int w = 1; w < 4; ++w) { const float4 v = sums[w][col]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        *reinterpret_cast<float4 *>(e < nw ? dW + e : db + (e - nw)) = s;
    }
}

// The same second stage for up to four weight gradients in ONE launch (blockIdx.y = problem): the module's backward has three
// of them (output_proj, the merged offsets / attention projection, value_proj), and at decoder sizes three 3-us launches with a
// dependent boundary each weigh as much as one of the first stages.  Same arithmetic, same order per problem: bitwise equal
// to linear_wgrad_reduce_kernel.
struct ReduceBatch { const float *part[4]; float *dW[4]; float *db[4]; long long nw[4], n[4]; int splits[4]; };
__global__ __launch_bounds__(256) void linear_wgrad_reduce_batched_kernel(const ReduceBatch rb)
{
    __shared__ float4 sums[4][64];
    const int p = (int)blockIdx.y;
    const float *__restrict__ part = rb.part[p];
    float *__restrict__ dW = rb.dW[p], *__restrict__ db = rb.db[p];
    const long long nw = rb.nw[p], n = rb.n[p];
    const int splits = rb.splits[p];
    const int g = threadIdx.x >> 6, col = threadIdx.x & 63;
    const long long e = ((long long)blockIdx.x * 64 + col) * 4;
    const bool live = splits > 1 && e < n && !(e >= nw && db == nullptr);
    const int per = (splits + 3) / 4, k0 = g * per, k1 = min(splits, k0 + per);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        int k = k0;
        for (; k + 4 <= k1; k += 4) {
            const float4 v0 = *reinterpret_cast<const float4 *>(part + (long long)k * n + e);
            const float4 v1 = *reinterpret_cast<const float4 *>(part + (long long)(k + 1) * n + e);
            const float4 v2 = *reinterpret_cast<const float4 *>(part + (long long)(k + 2) * n + e);
            const float4 v3 = *reinterpret_cast<const float4 *>(part + (long long)(k + 3) * n + e);
            s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
            s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
            s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
            s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
        }
        for (; k < k1; ++k) {
            const float4 v = *reinterpret_cast<const float4 *>(part + (long long)k * n + e);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    sums[g][col] = s;
    __syncthreads();
    if (g == 0 && live) {
#pragma unroll
        for (int w = 1; w < 4; ++w) { const float4 v = sums[w][col]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        *reinterpret_cast<float4 *>(e < nw ? dW + e : db + (e - nw)) = s;
    }
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

// Up to four fp32 -> bf16 conversions in ONE launch (the weights and biases of value_proj / output_proj under autocast: four
// 3 us kernels with a dependent boundary each became a visible share of a 110 us module step).  Round to nearest even, NaN
// stays NaN (v_cvt_pk_bf16_f32).  Segment k: n[k] elements, a multiple of 2, 8-byte aligned source / 4-byte aligned destination.
struct CastSegs { const float *src[4]; uint16_t *dst[4]; long long n[4]; };
__global__ __launch_bounds__(256) void cast_bf16_multi_kernel(const CastSegs segs)
{
    const int k = (int)blockIdx.y;
    const long long n2 = segs.n[k] / 2, stride = (long long)gridDim.x * 256;
    const float2 *s2 = reinterpret_cast<const float2 *>(segs.src[k]);
    unsigned *d2 = reinterpret_cast<unsigned *>(segs.dst[k]);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) {
        const float2 v = s2[i];
        const __bf16 a = static_cast<__bf16>(v.x), b = static_cast<__bf16>(v.y);
        d2[i] = (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
    }
}

int launch_cast_bf16_multi(int count, const float *const *src, uint16_t *const *dst, const long long *n, hipStream_t stream)
{
    CastSegs segs;
    long long most = 0;
    for (int k = 0; k < 4; ++k) {
        segs.src[k] = k < count ? src[k] : nullptr; segs.dst[k] = k < count ? dst[k] : nullptr; segs.n[k] = k < count ? n[k] : 0;
        if (segs.n[k] > most) most = segs.n[k];
    }
    if (most == 0) return MSDA_OK;
    long long blocks = (most / 2 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(cast_bf16_multi_kernel, dim3((unsigned)blocks, (unsigned)count), dim3(256), 0, stream, segs);
    return check_launch("msda fp32 -> bf16 (multi)");
}

int launch_zero_masked_rows(float *x, const uint8_t *mask, long long rows, int cols, hipStream_t stream)
{
    if (rows == 0) return MSDA_OK;
    hipLaunchKernelGGL(zero_masked_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, mask, rows, cols);
    return check_launch("msda zero masked rows");
}


// Kernel and number of M-splits.  Splits: enough workgroups to fill the chip (a few per CU: the kernels hide their barriers and
// LDS round trips behind other wavefronts' MFMAs), at least two stages per chunk, and — when there are 8 or more — a multiple
// of 8 so that whole splits go to one XCD (tile_and_split).  fp32 operands with at most 4096 rows take the 32-tile kernel
// (~512 workgroups; one per tile, without slabs, below 128 rows); at 6120 rows it is level at N = 256 (20.0 against 21.3 us)
// and behind at N = 384 (30.1 against 25.0).
struct WgradPlan { bool small; int splits; };
static WgradPlan wgrad_plan(int M, int N, int K, bool fp32_operands)
{
    static const int target = tuning_int("MSDA_WGRAD_WGS", 768);
    static const int small_rows = tuning_int("MSDA_WGRAD_SMALL_ROWS", 4096);
    static const int small_target = tuning_int("MSDA_WGRAD_SMALL_WGS", 512);
    static const int small_single = tuning_int("MSDA_WGRAD_SMALL_SINGLE", 128);
    WgradPlan pl;
    pl.small = fp32_operands && M <= small_rows;
    const int tile = pl.small ? kSmTile : kWgTile, stage = pl.small ? kSmStage : kWgStage;
    const int tiles = ((N + tile - 1) / tile) * ((K + tile - 1) / tile);
    int splits = ((pl.small ? small_target : target) + tiles - 1) / tiles;
    if (pl.small && M <= small_single) splits = 1;
    const int max_splits = (M + 2 * stage - 1) / (2 * stage);
    if (splits > max_splits) splits = max_splits;
    if (splits >= 8) splits = (splits + 4) & ~7;                        // nearest multiple of 8
    if (splits > max_splits) splits -= 8;
    pl.splits = splits < 1 ? 1 : splits;
    return pl;
}

size_t linear_wgrad_workspace_bytes(int M, int N, int K)
{
    // one size for both operand types: the larger of the two plans
    const int s32 = wgrad_plan(M, N, K, true).splits, s16 = wgrad_plan(M, N, K, false).splits;
    const int splits = s32 > s16 ? s32 : s16;
    return splits <= 1 ? 0 : sizeof(float) * (size_t)splits * ((size_t)N * K + (size_t)N);
}

// `defer` (may be null): instead of launching the second stage, describe it there (splits == 0: nothing to reduce)
struct DeferredReduce { const float *part; float *dW, *db; long long nw, n; int splits; };
template <typename OT>
static int launch_linear_wgrad_t(const OT *dY, const OT *X, const uint8_t *row_mask, int M, int N, int K, float *dW, float *db,
                                 float *workspace, hipStream_t stream, DeferredReduce *defer = nullptr)
{
    if (defer) defer->splits = 0;
    const WgradPlan pl = wgrad_plan(M, N, K, sizeof(OT) == 4);
    const int splits = pl.splits;
    int chunk = (M + splits - 1) / splits;
    constexpr int stage16 = kBfStage > kWgStage ? kBfStage : kWgStage;
    const int stage = pl.small ? kSmStage : sizeof(OT) == 2 ? stage16 : kWgStage;
    chunk = ((chunk + stage - 1) / stage) * stage;
    const int tile = pl.small ? kSmTile : kWgTile;
    const int tiles = ((N + tile - 1) / tile) * ((K + tile - 1) / tile);
    const dim3 grid((unsigned)(tiles * splits));
    void (*partial)(const OT *, const OT *, const uint8_t *, int, int, int, int, int, int, long long, float *, float *) =
        linear_wgrad_partial_kernel<OT>;
    if constexpr (sizeof(OT) == 2) {
        // bf16 operands: the bf16-MFMA kernel whenever whole 16-byte chunks of 8 columns can be moved (else the widening one)
        static const int use_bf16_mfma = tuning_int("MSDA_WGRAD_BF16_MFMA", 1);
        if (use_bf16_mfma && N % 8 == 0 && K % 8 == 0 && (((uintptr_t)dY | (uintptr_t)X) & 15) == 0)
            partial = linear_wgrad_partial_bf16_kernel;
    } else {
        if (pl.small) partial = linear_wgrad_small_kernel;
    }
    if (splits == 1) {
        hipLaunchKernelGGL(partial, grid, dim3(kWgBlock), 0, stream, dY, X, row_mask, M, N, K, chunk, tiles, splits, 0LL, dW, db);
        return check_launch("msda linear wgrad");
    }
    if (workspace == nullptr) return set_error(MSDA_ERR_ARGUMENT, "msda linear wgrad: workspace required");
    const long long nw = (long long)N * K, slab = nw + N;               // per split: weight partials, then bias partials
    hipLaunchKernelGGL(partial, grid, dim3(kWgBlock), 0, stream, dY, X, row_mask, M, N, K, chunk, tiles, splits, slab, workspace,
                       db ? workspace + nw : nullptr);
    if (int rc = check_launch("msda linear wgrad (partial)")) return rc;
    if (defer) { *defer = DeferredReduce{workspace, dW, db, nw, slab, splits}; return MSDA_OK; }
    hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((unsigned)((slab / 4 + 63) / 64)), dim3(256), 0, stream, workspace,
                       splits, nw, slab, dW, db);
    return check_launch("msda linear wgrad (reduce)");
}

int launch_linear_wgrad(const float *dY, const float *X, const uint8_t *row_mask, int M, int N, int K, float *dW, float *db,
                        float *workspace, hipStream_t stream)
{
    return launch_linear_wgrad_t<float>(dY, X, row_mask, M, N, K, dW, db, workspace, stream);
}
// `count` (1..4) weight gradients, operands fp32 or bf16 per problem (bf16[p] != 0): their first stages one after the other
// (independent launches), then ONE second stage.
int launch_linear_wgrad_multi(int count, const void *const *dY, const void *const *X, const int *bf16, const uint8_t *const *row_mask,
                              const int *M, const int *N, const int *K, float *const *dW, float *const *db, float *const *workspace,
                              hipStream_t stream)
{
    ReduceBatch rb;
    long long most = 0;
    int pending = 0;
    for (int p = 0; p < 4; ++p) { rb.part[p] = nullptr; rb.dW[p] = rb.db[p] = nullptr; rb.nw[p] = rb.n[p] = 0; rb.splits[p] = 0; }
    for (int p = 0; p < count; ++p) {
        DeferredReduce d;
        const uint8_t *mask = row_mask ? row_mask[p] : nullptr;
        float *bias = db ? db[p] : nullptr;
        const int rc = bf16 && bf16[p]
            ? launch_linear_wgrad_t<uint16_t>(static_cast<const uint16_t *>(dY[p]), static_cast<const uint16_t *>(X[p]), mask, M[p], N[p],
                                              K[p], dW[p], bias, workspace[p], stream, &d)
            : launch_linear_wgrad_t<float>(static_cast<const float *>(dY[p]), static_cast<const float *>(X[p]), mask, M[p], N[p], K[p],
                                           dW[p], bias, workspace[p], stream, &d);
        if (rc) return rc;
        if (d.splits > 1) {
            rb.part[p] = d.part; rb.dW[p] = d.dW; rb.db[p] = d.db; rb.nw[p] = d.nw; rb.n[p] = d.n; rb.splits[p] = d.splits;
            if (d.n > most) most = d.n;
            ++pending;
        }
    }
    if (pending == 0) return MSDA_OK;
    hipLaunchKernelGGL(linear_wgrad_reduce_batched_kernel, dim3((unsigned)((most / 4 + 63) / 64), (unsigned)count), dim3(256), 0, stream, rb);
    return check_launch("msda linear wgrad (batched reduce)");
}

int launch_linear_wgrad_bf16(const uint16_t *dY, const uint16_t *X, const uint8_t *row_mask, int M, int N, int K, float *dW,
                             float *db, float *workspace, hipStream_t stream)
{
    return launch_linear_wgrad_t<uint16_t>(dY, X, row_mask, M, N, K, dW, db, workspace, stream);
}

}  // namespace msda
