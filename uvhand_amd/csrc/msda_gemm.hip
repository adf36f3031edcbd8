// Forward and input gradient of the fp32 nn.Linear layers that bracket the sampling kernel
// (UVHand models/ops/modules/ms_deform_attn.py:96,100,101,139: value_proj, sampling_offsets,
// attention_weights, output_proj):
//
//     forward   y[M, N]  = x[M, K] . W[N, K]^T + b[N]          (B operand K-major: "NT")
//     dgrad     dx[M, K] = dy[M, N] . W[N, K]                  (B operand row-major over the reduction: "NN")
//
// both as   C[M, Nc] = A[M, Kr] . B  (+ bias) , rows with row_mask != 0 written as zeros (the padding mask of value_proj and of
// its input gradient, modules/ms_deform_attn.py:97-98, without a second pass).
//
// Why not the vendor BLAS: at the shapes of this module (K = 256, N <= 384, M = 600 ... 33440) a GEMM through torch costs
// ~27 us of HOST time (dispatcher, heuristics lookup, workspace, hipBLASLt launch path) — six of them are half of the module's
// eager step at the headline decoder shape (tools/exp_module_hostprof.py) — and its kernels run at 50-88 TFLOP/s of the
// 157 TFLOP/s fp32 MFMA peak.  One plain launch through the C ABI costs ~5 us.
//
// Kernel: fp32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32 fma chain), 64 x 64 output tile per 256-thread workgroup, 4 wavefronts
// as 2 x 2 blocks of 32 x 32, reduction in stages of 32.  An fp32 MFMA occupies the pipe for 64 clocks and needs ONE float per
// lane and operand, so operand delivery is light and SMALL tiles are affordable: at M = 33440, N = 256 they give 2092
// workgroups — 8.2 per CU, so the last round is 9 against 8.2 (91 %) where 128 x 128 tiles would give 3 rounds against 2.05.
//   * A (and the NT B operand) are K-major in memory exactly as the MFMA wants them per lane: stages are straight 16-byte row
//     segment copies into LDS rows of 36 floats (16-byte aligned, and 36/4 odd: the 16 lanes of one ds_read_b128 phase hit 16
//     different 4-bank groups); a lane reads its 4 consecutive k of a row with ONE ds_read_b128 and feeds them to 4 MFMAs —
//     lane half h owns k = 8i + 4h + t of the stage in MFMA t of group i, the same for both operands.
//   * the NN B operand (W[k][n], n contiguous) is staged as it lies and read with one ds_read_b32 per MFMA (32 consecutive
//     floats per half wavefront).
//   * LDS double-buffered, one barrier per stage, global loads two stages ahead (as the weight-gradient kernel, msda_linear.hip).
//   * XCD-aware numbering: the tiles of one 64-row panel of A (consecutive logical ids) run on ONE XCD, so the panel is
//     fetched into one L2; the weights (<= 393 KB) live in every L2.
// Results: fixed summation order, bitwise reproducible.
#include <cstdlib>

#include "msda_common.h"
#include "msda_launch.h"

namespace msda {

constexpr int kGmBlock = 256;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// One MFMA block of a wavefront: BLK x BLK outputs, lanes in 64 / BLK groups g, lane (c, g) supplying row / column c and the
// reduction index g of each instruction (32x32x2: 2 groups; 16x16x4: 4 groups).
template <int BLK> struct MfmaBlock;
template <> struct MfmaBlock<32> {
    using Acc = f32x16;
    static constexpr int kRegs = 16;
    static __device__ __forceinline__ Acc mma(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }
};
template <> struct MfmaBlock<16> {
    using Acc = f32x4;
    static constexpr int kRegs = 4;
    static __device__ __forceinline__ Acc mma(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int r, int g) { return 4 * g + r; }
};

// TILE x TILE outputs per workgroup (4 wavefronts, 2 x 2 blocks of TILE/2), reduction in stages of STAGE:
//   TILE 64, STAGE 32  (32x32x2 MFMA)  the throughput shape;
//   TILE 32, STAGE 64  (16x16x4 MFMA)  few rows: 4x the workgroups, each wavefront's MFMA chain a quarter as long.
template <bool B_KMAJOR, int TILE, int STAGE>
__global__ __launch_bounds__(kGmBlock) void linear_rows_kernel(
    const float *__restrict__ A, const float *__restrict__ B, const float *__restrict__ bias,
    const uint8_t *__restrict__ row_mask, float *__restrict__ C, long long M, int Nc, int Kr, int tiles_n, long long tiles)
{
    constexpr int BLK = TILE / 2, G = 64 / BLK;                 // lane groups of the MFMA
    constexpr int NI = STAGE / (4 * G);                         // 16-byte operand reads per lane, operand and stage
    constexpr int kRowK = STAGE + 4;                            // LDS row of a K-major operand (floats): 16-byte aligned, /4 odd
    constexpr int kRowN = TILE + 4;                             // LDS row of the NN B operand
    using MB = MfmaBlock<BLK>;
    __shared__ __attribute__((aligned(16))) float As[2][TILE * kRowK];
    __shared__ __attribute__((aligned(16))) float Bs[2][B_KMAJOR ? TILE * kRowK : STAGE * kRowN];
    // workgroup id -> logical tile: XCD x (= id mod 8) takes the contiguous range [x * per, (x + 1) * per)
    const long long per = (tiles + 7) >> 3;
    const long long logical = (long long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((long long)(blockIdx.x >> 3) >= per || logical >= tiles) return;
    const long long m0 = (logical / tiles_n) * TILE;
    const int n0 = (int)(logical % tiles_n) * TILE;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i0 = (wave >> 1) * BLK, j0 = (wave & 1) * BLK, g = lane / BLK, c = lane % BLK;

    // staging: every thread moves kLdK float4 of each K-major operand (TILE rows x STAGE floats) and kLdN of the NN B operand
    // (STAGE rows x TILE floats) per stage
    constexpr int kPerRowK = STAGE / 4, kStepK = kGmBlock / kPerRowK, kLdK = TILE / kStepK;
    constexpr int kPerRowN = TILE / 4, kStepN = kGmBlock / kPerRowN, kLdN = STAGE / kStepN;
    static_assert(kLdK >= 1 && kLdN >= 1 && NI >= 2 && NI % 2 == 0, "tile / stage combination");
    constexpr int kLdB = B_KMAJOR ? kLdK : kLdN;
    const int krow = tid / kPerRowK, kcol = (tid % kPerRowK) * 4;
    const int nrow = tid / kPerRowN, ncol = (tid % kPerRowN) * 4;
    float4 ra[kLdK], rb[kLdB];
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#define MSDA_GM_LOAD(k0_)                                                                                           \
    do {                                                                                                            \
        _Pragma("unroll") for (int r = 0; r < kLdK; ++r) {                                                          \
            const long long m = m0 + krow + kStepK * r;                                                             \
            ra[r] = zero;                                                                                           \
            if (m < M && (k0_) + kcol < Kr) ra[r] = *reinterpret_cast<const float4 *>(A + m * Kr + (k0_) + kcol);   \
        }                                                                                                           \
        _Pragma("unroll") for (int r = 0; r < kLdB; ++r) {                                                          \
            rb[r] = zero;                                                                                           \
            if (B_KMAJOR) {                                                                                         \
                const int n = n0 + krow + kStepK * r;                                                               \
                if (n < Nc && (k0_) + kcol < Kr)                                                                    \
                    rb[r] = *reinterpret_cast<const float4 *>(B + (long long)n * Kr + (k0_) + kcol);                \
            } else {                                                                                                \
                const int k = (k0_) + nrow + kStepN * r;                                                            \
                if (k < Kr && n0 + ncol < Nc)                                                                       \
                    rb[r] = *reinterpret_cast<const float4 *>(B + (long long)k * Nc + n0 + ncol);                   \
            }                                                                                                       \
        }                                                                                                           \
    } while (0)
#define MSDA_GM_STORE(buf_)                                                                                         \
    do {                                                                                                            \
        _Pragma("unroll") for (int r = 0; r < kLdK; ++r)                                                            \
            *reinterpret_cast<float4 *>(&As[buf_][(krow + kStepK * r) * kRowK + kcol]) = ra[r];                     \
        _Pragma("unroll") for (int r = 0; r < kLdB; ++r) {                                                          \
            if (B_KMAJOR) *reinterpret_cast<float4 *>(&Bs[buf_][(krow + kStepK * r) * kRowK + kcol]) = rb[r];       \
            else          *reinterpret_cast<float4 *>(&Bs[buf_][(nrow + kStepN * r) * kRowN + ncol]) = rb[r];       \
        }                                                                                                           \
    } while (0)
    // operand fragments of a stage: lane (c, g) owns reduction indices 4 G i + 4 g + t of the stage, t = 0..3 feeding 4 MFMAs
#define MSDA_GM_FRAGS(buf_, av_, bv_)                                                                               \
    do {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < NI; ++i)                                                              \
            av_[i] = *reinterpret_cast<const float4 *>(&As[buf_][(i0 + c) * kRowK + 4 * G * i + 4 * g]);            \
        if (B_KMAJOR) {                                                                                             \
            _Pragma("unroll") for (int i = 0; i < NI; ++i)                                                          \
                bv_[i] = *reinterpret_cast<const float4 *>(&Bs[buf_][(j0 + c) * kRowK + 4 * G * i + 4 * g]);        \
        } else {                                                                                                    \
            _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                                        \
                const float *bp = &Bs[buf_][(4 * G * i + 4 * g) * kRowN + j0 + c];                                  \
                bv_[i] = make_float4(bp[0], bp[kRowN], bp[2 * kRowN], bp[3 * kRowN]);                               \
            }                                                                                                       \
        }                                                                                                           \
    } while (0)
#define MSDA_GM_MMA(av_, bv_, from_, to_)                                                                           \
    do {                                                                                                            \
        _Pragma("unroll") for (int i = (from_); i < (to_); ++i) {                                                   \
            acc = MB::mma(av_[i].x, bv_[i].x, acc);                                                                 \
            acc = MB::mma(av_[i].y, bv_[i].y, acc);                                                                 \
            acc = MB::mma(av_[i].z, bv_[i].z, acc);                                                                 \
            acc = MB::mma(av_[i].w, bv_[i].w, acc);                                                                 \
        }                                                                                                           \
    } while (0)

    typename MB::Acc acc;
#pragma unroll
    for (int r = 0; r < MB::kRegs; ++r) acc[r] = 0.f;
    float4 av[NI], bv[NI], avn[NI], bvn[NI];
    MSDA_GM_LOAD(0);
    MSDA_GM_STORE(0);
    MSDA_GM_LOAD(STAGE);                                                // past Kr: zeros
    __syncthreads();
    MSDA_GM_FRAGS(0, av, bv);
    // Software-pipelined over the stages so that a wavefront's MFMA chain never stops for its own LDS traffic (with every
    // resident wavefront of a SIMD sharing one MFMA pipe they fall into step, and a gap in one is a gap in all):
    //   start of stage s   the registers loaded during stage s-1 (stage s+1) go to the other LDS buffer — its last readers are
    //                      behind the barrier of stage s-1 (which also drains their LDS reads) — and the global loads of
    //                      stage s+2 are issued;
    //   first half of the MFMAs; ONE barrier; the fragment reads of stage s+1 are issued; second half of the MFMAs.
    int cur = 0;
    for (int k0 = 0; k0 < Kr; k0 += STAGE) {
        const bool more = k0 + STAGE < Kr;                              // uniform
        if (more) {
            MSDA_GM_STORE(cur ^ 1);
            if (k0 + 2 * STAGE < Kr) MSDA_GM_LOAD(k0 + 2 * STAGE);
        }
        MSDA_GM_MMA(av, bv, 0, NI / 2);
        if (more) {
            __syncthreads();
            MSDA_GM_FRAGS(cur ^ 1, avn, bvn);
        }
        MSDA_GM_MMA(av, bv, NI / 2, NI);
        if (more) {
#pragma unroll
            for (int i = 0; i < NI; ++i) { av[i] = avn[i]; bv[i] = bvn[i]; }
        }
        cur ^= 1;
    }
#undef MSDA_GM_LOAD
#undef MSDA_GM_STORE
#undef MSDA_GM_FRAGS
#undef MSDA_GM_MMA

    // C/D layout: column = c, rows MB::row(reg, g); the mask bytes of the lane's rows are fetched together, ahead of the stores
    const int j = n0 + j0 + c;
    if (j >= Nc) return;
    const float bj = bias != nullptr ? bias[j] : 0.f;
    unsigned dead = 0;
    if (row_mask != nullptr) {
#pragma unroll
        for (int r = 0; r < MB::kRegs; ++r) {
            const long long i = m0 + i0 + MB::row(r, g);
            if (i < M && row_mask[i] != 0) dead |= 1u << r;
        }
    }
#pragma unroll
    for (int r = 0; r < MB::kRegs; ++r) {
        const long long i = m0 + i0 + MB::row(r, g);
        if (i < M) C[i * Nc + j] = ((dead >> r) & 1u) ? 0.f : acc[r] + bj;
    }
}

#ifdef MSDA_TUNING
static int gemm_knob(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
#endif

template <bool B_KMAJOR>
static int launch_linear_rows_t(const float *A, const float *B, const float *bias, const uint8_t *row_mask, float *C, long long M,
                                int Nc, int Kr, hipStream_t stream)
{
    if (M == 0) return MSDA_OK;
    // the small tile while the 64 x 64 tiling leaves most of the chip's 1024 SIMDs without a wavefront
    int tile = ((M + 63) / 64) * ((Nc + 63) / 64) * 4 <= 512 ? 32 : 64;
#ifdef MSDA_TUNING
    tile = gemm_knob("MSDA_GEMM_TILE", tile);
#endif
    const int tiles_n = (Nc + tile - 1) / tile;
    const long long tiles = ((M + tile - 1) / tile) * tiles_n;
    const long long grid = 8 * ((tiles + 7) / 8);
    if (grid * kGmBlock > 0xffffffffLL) return set_error(MSDA_ERR_ARGUMENT, "linear: too many rows for one launch");   // < 2^32 threads
    if (tile == 32)
        hipLaunchKernelGGL((linear_rows_kernel<B_KMAJOR, 32, 64>), dim3((unsigned)grid), dim3(kGmBlock), 0, stream, A, B, bias,
                           row_mask, C, M, Nc, Kr, tiles_n, tiles);
    else
        hipLaunchKernelGGL((linear_rows_kernel<B_KMAJOR, 64, 32>), dim3((unsigned)grid), dim3(kGmBlock), 0, stream, A, B, bias,
                           row_mask, C, M, Nc, Kr, tiles_n, tiles);
    return check_launch("linear_rows_kernel");
}

int launch_linear_forward(const float *x, const float *w, const float *bias, const uint8_t *row_mask, long long rows,
                          int out_features, int in_features, float *y, hipStream_t stream)
{
    return launch_linear_rows_t<true>(x, w, bias, row_mask, y, rows, out_features, in_features, stream);
}

int launch_linear_dgrad(const float *grad_out, const float *w, const uint8_t *row_mask, long long rows, int out_features,
                        int in_features, float *grad_in, hipStream_t stream)
{
    return launch_linear_rows_t<false>(grad_out, w, nullptr, row_mask, grad_in, rows, in_features, out_features, stream);
}

}  // namespace msda
