/*
 * msda.h — C ABI of libmsda_hip.so: multi-scale deformable attention
 * forward / backward for AMD MI355X (gfx950), hand-written HIP.
 *
 * This is the drop-in boundary for the reference's native op.  Each entry point
 * replaces one function of the reference's pybind module
 * `MultiScaleDeformableAttention` (paths relative to the UVHand repo):
 *
 *   msda_forward_*   <-  ms_deform_attn_forward   models/ops/src/vision.cpp:14,
 *                        models/ops/src/ms_deform_attn.h:20-39, which dispatches to
 *                        ms_deform_attn_cuda_forward  models/ops/src/cuda/ms_deform_attn_cuda.cu:20-80
 *                        (kernel: models/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299)
 *   msda_backward_*  <-  ms_deform_attn_backward  models/ops/src/vision.cpp:15,
 *                        models/ops/src/ms_deform_attn.h:42-61, which dispatches to
 *                        ms_deform_attn_cuda_backward models/ops/src/cuda/ms_deform_attn_cuda.cu:83-153
 *                        (kernels: models/ops/src/cuda/ms_deform_im2col_cuda.cuh:301-920)
 *
 * The reference passes at::Tensor; this ABI passes what those tensors are: raw
 * DEVICE pointers into caller-owned, contiguous, row-major storage plus sizes:
 *
 *   value          [N, S, M, D]           T      S = sum_l H_l*W_l
 *   spatial_shapes [L, 2]  (H_l, W_l)     int64  device memory (read in-kernel, as the
 *   level_start    [L]                    int64  reference does: im2col_cuda.cuh:274-277)
 *   sampling_loc   [N, Lq, M, L, P, 2]    TL     (x, y), normalised to the level's extent
 *   attn_weight    [N, Lq, M, L, P]       TL
 *   out / grad_out [N, Lq, M*D]           T
 *   grad_value     [N, S, M, D]           T
 *   grad_sampling_loc, grad_attn_weight   TL     same shapes as sampling_loc / attn_weight
 *
 *   suffix  T        TL      accumulation
 *   f32     float    float   float            (the reference's float instantiation)
 *   f64     double   double  double           (the reference's double instantiation; gradcheck)
 *   bf16    bf16     float   float            (new capability, no reference counterpart: bf16 storage
 *                                              of the value-like tensors, D = 32 family only)
 *
 * Semantics kept from the reference: the batch is processed whole (im2col_step
 * only chunks the reference's launches; results do not depend on it, the host
 * wrapper validates it); every output element is written by the call — the
 * caller does NOT need to zero out / grad_* beforehand (the reference zero-fills
 * them on the host side, ms_deform_attn_cuda.cu:54,121-123).
 *
 * Ownership: the library allocates nothing and frees nothing, reads no environment
 * variable, and keeps no process-wide mutable state that any computation depends on: the error string and
 * the test hook msda_force_path() are per thread (msda_launch_count() is a diagnostic counter).  It never synchronises the
 * device: all work (including the zero-fill of grad_value where a kernel needs
 * it) is enqueued on `stream` (a hipStream_t; NULL = the default stream).
 * Re-entrant: forward and backward may be called concurrently from different
 * host threads (the backward arrives on PyTorch's autograd thread).
 *
 * Alignment: tensors obtained from an allocator always qualify.  The D = 32 kernels move rows as 16-byte
 * (fp32) / 8-byte (bf16) vectors and (x, y) pairs as 8 bytes; an fp32 call whose tensors are only
 * element-aligned (a contiguous view at an odd offset) is served by the generic kernels, the bf16,
 * fused-prologue and weight-gradient entry points return MSDA_ERR_ARGUMENT for such pointers.
 *
 * Errors: every function returns 0 on success, non-zero on failure, in which
 * case msda_last_error() describes it.  Unlike the reference, which only
 * printf()s a failed launch (im2col_cuda.cuh:948-952, 1321-1325), launch errors
 * are returned.
 */
#ifndef MSDA_H_
#define MSDA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSDA_OK            0
#define MSDA_ERR_ARGUMENT  1   /* null pointer, non-positive size, index range beyond int32 */
#define MSDA_ERR_LAUNCH    2   /* HIP runtime reported an error at enqueue time */

/* Which kernel family a call would use for the given geometry (for tests/bench):
 * 0 = generic (any D, any dtype), 1 = D=32 fast path. */
#define MSDA_PATH_GENERIC  0
#define MSDA_PATH_D32      1

typedef void *msda_stream_t; /* hipStream_t */

int msda_forward_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start,
                     const float *sampling_loc, const float *attn_weight,
                     int N, int S, int M, int D, int L, int Lq, int P,
                     float *out, msda_stream_t stream);

int msda_backward_f32(const float *grad_out, const float *value, const int64_t *spatial_shapes,
                      const int64_t *level_start, const float *sampling_loc, const float *attn_weight,
                      int N, int S, int M, int D, int L, int Lq, int P,
                      float *grad_value, float *grad_sampling_loc, float *grad_attn_weight,
                      msda_stream_t stream);

int msda_forward_f64(const double *value, const int64_t *spatial_shapes, const int64_t *level_start,
                     const double *sampling_loc, const double *attn_weight,
                     int N, int S, int M, int D, int L, int Lq, int P,
                     double *out, msda_stream_t stream);

int msda_backward_f64(const double *grad_out, const double *value, const int64_t *spatial_shapes,
                      const int64_t *level_start, const double *sampling_loc, const double *attn_weight,
                      int N, int S, int M, int D, int L, int Lq, int P,
                      double *grad_value, double *grad_sampling_loc, double *grad_attn_weight,
                      msda_stream_t stream);

/* bf16 tensors are passed as uint16_t* (raw bfloat16 bits); all arithmetic and accumulation is fp32, one rounding at
 * the final store.  D = 32 geometries take the tiled kernels, every other D (or rows at an odd element offset) the
 * generic ones.  msda_backward_bf16 (bf16 grad_value) exists for the D = 32 family only and returns MSDA_ERR_ARGUMENT
 * elsewhere — msda_backward_bf16_gv32 below serves every geometry (msda_path_for(2, M, D, L, P) tells which is which). */
int msda_forward_bf16(const uint16_t *value, const int64_t *spatial_shapes, const int64_t *level_start,
                      const float *sampling_loc, const float *attn_weight,
                      int N, int S, int M, int D, int L, int Lq, int P,
                      uint16_t *out, msda_stream_t stream);

int msda_backward_bf16(const uint16_t *grad_out, const uint16_t *value, const int64_t *spatial_shapes,
                       const int64_t *level_start, const float *sampling_loc, const float *attn_weight,
                       int N, int S, int M, int D, int L, int Lq, int P,
                       uint16_t *grad_value, float *grad_sampling_loc, float *grad_attn_weight,
                       msda_stream_t stream);

/* bf16 rows in, fp32 grad_value out.  Same as msda_backward_bf16 except that grad_value is float[N,S,M,D]:
 * nothing is rounded between the query chunks ("passes") of a long backward, which also lets those passes
 * accumulate in place in global memory instead of in an LDS tile (cfg-4 encoder regime: 2x faster than
 * msda_backward_bf16), and a caller whose `value` parameter is fp32 needs no conversion of the result.
 * msda_backward_passes(Lq, P) = number of passes the D = 32 backward takes (1 = single pass). */
int msda_backward_bf16_gv32(const uint16_t *grad_out, const uint16_t *value, const int64_t *spatial_shapes,
                            const int64_t *level_start, const float *sampling_loc, const float *attn_weight,
                            int N, int S, int M, int D, int L, int Lq, int P,
                            float *grad_value, float *grad_sampling_loc, float *grad_attn_weight,
                            msda_stream_t stream);
int msda_backward_passes(int Lq, int P);

/* ---- Backward with flags and caller-provided scratch (D = 32 family) -----------------------------------
 * flags = 0: exactly msda_backward_*.
 * flags & MSDA_FLAG_DETERMINISTIC: grad_value is bitwise reproducible run to run (grad_sampling_loc and
 * grad_attn_weight always are).  The default kernels order the contributions to a pixel by the rank an LDS
 * integer atomic returned, the reference by the arrival of its atomicAdds (ms_deform_im2col_cuda.cuh:125-152);
 * both differ in the last bits between runs.  With the flag the D = 32 kernels run the same sort + gather with one
 * counter per (pixel row, WAVEFRONT) — eight 16-bit counters packed in four LDS words per row — so that a row's
 * records end up wavefront-major in sampling-point order, a pure function of the inputs (uvhand_amd/csrc/
 * msda_d32_value.h, DET), inside the same single launch as the default mode.  It needs no scratch: `workspace` may be NULL
 * (msda_backward_workspace_bytes() says what a call can use; with this flag alone that is nothing).
 * Shapes that are inconsistent with S (a level whose pixels do not lie in [0, S)) never cause an
 * out-of-range access on this path: such a level contributes nothing and pixels no level covers get zeros.
 * Outside the D = 32 family (any D, fp64, element-aligned views) the flag selects a destination-major kernel that
 * adds a pixel's contributions in (query, point) order — no atomics, no scratch, rows x Lq*P point tests of work; a call
 * with N*S*M x Lq*P > 2^36 is refused (MSDA_ERR_ARGUMENT) rather than run for seconds.  A pixel that several levels cover
 * (overlapping level_start ranges: not something the reference's callers produce) is credited to the first such level there,
 * to every level by the default kernels: overlapping levels are unsupported in every deterministic path.
 * Cost on the D = 32 family: 3-15 % over the default backward (profiles/r03_notes.md section 4).  Replaces the same reference functions as msda_backward_*. */
#define MSDA_FLAG_DETERMINISTIC 1u
/* msda_backward_workspace_bytes only: the size is asked for a msda_backward_prologue_* call (on large problems its
 * grad_sampling_loc / grad_attn_weight workgroups see one head each and leave the reference-point gradient per head in
 * the scratch; without scratch the call runs the kernels that need none). */
#define MSDA_FLAG_PROLOGUE 2u
/* the workspace of this backward call STARTS WITH the table a msda_forward_ws_* / msda_forward_prologue_ws_* call of the same
 * geometry filled from the same sampling locations / attention weights (below); any scratch the call uses follows it, at the
 * table's size rounded up to 256 bytes (msda_backward_workspace_bytes with this flag = both).  The table is ignored where the
 * backward's plan reads none (msda_forward_workspace_bytes() == 0, the deterministic flag), where the buffer is too small,
 * or where its stamp says that the forward did not write it. */
#define MSDA_FLAG_FORWARD_TABLE 4u
/* grad_value of EVERY level through the sort + gather kernels, none as a dense product on the matrix cores (large problems,
 * levels of at most 64 pixels: uvhand_amd/csrc/msda_d32_dense.h).  The results agree to fp32 summation order either way while
 * grad_out is finite.  A non-finite grad_out row differs: in the dense product a zero weight still multiplies every query's
 * row (0 x Inf = NaN), so ALL pixels of that (batch, head)'s dense levels become NaN, where the reference's atomicAdd
 * (ms_deform_im2col_cuda.cuh:125-152) — and this flag — poison only the pixels that query's taps land on.  Costs the dense
 * levels' speed-up (cfg-4 encoder backward +8 %); GradScaler-style loops that discard a non-finite step do not need it. */
#define MSDA_FLAG_EXACT_NONFINITE 8u
unsigned long long msda_backward_workspace_bytes(int N, int S, int M, int D, int L, int Lq, int P, unsigned flags);
/* 1 if a backward of this geometry honours MSDA_FLAG_DETERMINISTIC, 0 if it would be refused (MSDA_ERR_ARGUMENT): outside
 * the D = 32 family (elem_bytes 8 = fp64, any D != 32, ...) the deterministic kernel is a brute-force test of every sampling
 * point against every pixel row, bounded at N*S*M x Lq*P <= 2^36.  Hosts that run under
 * torch.use_deterministic_algorithms(True, warn_only=True) ask first, warn, and clear the flag.  Pure host logic. */
int msda_deterministic_supported(int elem_bytes, int N, int S, int M, int D, int L, int Lq, int P);
int msda_backward_ws_f32(const float *grad_out, const float *value, const int64_t *spatial_shapes,
                         const int64_t *level_start, const float *sampling_loc, const float *attn_weight,
                         int N, int S, int M, int D, int L, int Lq, int P,
                         float *grad_value, float *grad_sampling_loc, float *grad_attn_weight,
                         void *workspace, unsigned long long workspace_bytes, unsigned flags, msda_stream_t stream);
int msda_backward_ws_f64(const double *grad_out, const double *value, const int64_t *spatial_shapes,
                         const int64_t *level_start, const double *sampling_loc, const double *attn_weight,
                         int N, int S, int M, int D, int L, int Lq, int P,
                         double *grad_value, double *grad_sampling_loc, double *grad_attn_weight,
                         void *workspace, unsigned long long workspace_bytes, unsigned flags, msda_stream_t stream);
int msda_backward_ws_bf16(const uint16_t *grad_out, const uint16_t *value, const int64_t *spatial_shapes,
                          const int64_t *level_start, const float *sampling_loc, const float *attn_weight,
                          int N, int S, int M, int D, int L, int Lq, int P,
                          uint16_t *grad_value, float *grad_sampling_loc, float *grad_attn_weight,
                          void *workspace, unsigned long long workspace_bytes, unsigned flags, msda_stream_t stream);
int msda_backward_ws_bf16_gv32(const uint16_t *grad_out, const uint16_t *value, const int64_t *spatial_shapes,
                               const int64_t *level_start, const float *sampling_loc, const float *attn_weight,
                               int N, int S, int M, int D, int L, int Lq, int P,
                               float *grad_value, float *grad_sampling_loc, float *grad_attn_weight,
                               void *workspace, unsigned long long workspace_bytes, unsigned flags, msda_stream_t stream);

/* ---- Forward that prepares its backward (D = 32 family) -----------------------------------------------------
 * The reference's backward re-derives every sampling point's geometry from sampling_loc / attn_weight
 * (ms_deform_im2col_cuda.cuh:340-371, inside every one of its 32 channel threads).  Here the FORWARD already computes it once
 * per point; msda_forward_ws_* additionally leaves in `workspace` what the backward of the same autograd node would otherwise
 * work out again (pass the same buffer as `workspace` with MSDA_FLAG_FORWARD_TABLE to msda_backward_ws_*; for the
 * fused-prologue pair: msda_forward_prologue_ws_* and msda_backward_prologue_ws_f32 / _bf16_gv32):
 *   small problems (every workgroup of the backward launch resident at once — the 300-query decoder shape): a level-major
 *     POINT TABLE, 16 bytes per sampling point, entry ((b*M + m)*L + l) * Lq*P + q*P + p = {tap validity bits << 24 | pixel
 *     index of the top-left tap + W + 1, the two bilinear fractions, the attention weight}, and behind it a header with the
 *     pixel range of every grad_value workgroup;
 *   large problems whose grad_value pass cuts the levels into W <= 8 pixel ranges (the encoder shapes): RANGE MASKS, one
 *     byte per sampling point, level-major — bit t set iff a tap of the point may land in range t — which turn every range's
 *     strided scan of sampling_loc into a coalesced byte scan.
 * grad_sampling_loc and grad_attn_weight are bit-identical with and without the buffer; grad_value is equal up to the order
 * in which a pixel's contributions are summed (as between any two runs of the default kernels).  Each buffer carries a stamp
 * that the forward writes and the backward checks: a forward call that cannot fill the buffer (rows that are not 16-byte
 * aligned take the generic kernels) clears it, and the backward then ignores the buffer.
 * msda_forward_workspace_bytes(): the buffer's size for a geometry, 0 where the backward's plan reads none; flags:
 * MSDA_FLAG_PROLOGUE for the fused-prologue pair.  workspace NULL / too small / unaligned (16 bytes): exactly msda_forward_*.
 * The caller owns the buffer and keeps it, unmodified, with the tensors it saves for the backward. */
unsigned long long msda_forward_workspace_bytes(int N, int S, int M, int D, int L, int Lq, int P, unsigned flags);
int msda_forward_ws_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start,
                        const float *sampling_loc, const float *attn_weight,
                        int N, int S, int M, int D, int L, int Lq, int P,
                        float *out, void *workspace, unsigned long long workspace_bytes, msda_stream_t stream);
int msda_forward_ws_bf16(const uint16_t *value, const int64_t *spatial_shapes, const int64_t *level_start,
                         const float *sampling_loc, const float *attn_weight,
                         int N, int S, int M, int D, int L, int Lq, int P,
                         uint16_t *out, void *workspace, unsigned long long workspace_bytes, msda_stream_t stream);

/* ---- Fused module prologue (fp32, D = 32 family) ---------------------------------
 * The module computes  attn = softmax(logits) over the L*P points of a (query, head)  and
 * sampling_loc = reference_point + offset / (W_l, H_l)  (models/ops/modules/ms_deform_attn.py:101-108,
 * :110-128 after the 21-keypoint mean) with elementwise PyTorch kernels before calling the op.  These
 * entry points take the RAW tensors instead and do that arithmetic in the kernels' point lanes:
 *   forward : reference_points[N,Lq,L,2], sampling_offsets[N,Lq,M,L,P,2] (pixels), attn_logits[N,Lq,M,L*P]
 *             -> out, plus sampling_loc_out / attn_weight_out (what the unfused path would have been
 *             given; saved for the backward)
 *   backward: takes those saved tensors; returns grad_value and the gradients of the RAW tensors
 *             (offsets, logits — softmax backward included — and reference points, summed over heads
 *             and points).
 * The module produces offsets and logits with two nn.Linear layers on the same input (:100-101); run as
 * ONE GEMM their outputs are column blocks of a [N*Lq, ld] matrix.  ld_offsets / ld_logits are the floats
 * between consecutive (batch, query) rows of those two tensors (0 = dense: 2*M*L*P and M*L*P); the raw
 * gradients are written with ld_grad_offsets / ld_grad_logits the same way, so the backward of that one
 * GEMM reads them in place.  Offsets need an even stride and an 8-byte aligned base.
 * msda_prologue_supported() != 0 iff the geometry qualifies (D = 32 family, L*P and P powers of two,
 * whole queries per workgroup); otherwise callers compose the plain entry points as the reference does. */
int msda_prologue_supported(int N, int S, int M, int D, int L, int Lq, int P);
int msda_forward_prologue_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start,
                              const float *reference_points, const float *sampling_offsets, const float *attn_logits,
                              int N, int S, int M, int D, int L, int Lq, int P, long long ld_offsets, long long ld_logits,
                              float *out, float *sampling_loc_out, float *attn_weight_out, msda_stream_t stream);
int msda_backward_prologue_f32(const float *grad_out, const float *value, const int64_t *spatial_shapes,
                               const int64_t *level_start, const float *sampling_loc, const float *attn_weight,
                               int N, int S, int M, int D, int L, int Lq, int P, long long ld_grad_offsets,
                               long long ld_grad_logits, float *grad_value, float *grad_sampling_offsets,
                               float *grad_attn_logits, float *grad_reference_points, msda_stream_t stream);
/* msda_forward_prologue_f32 that also leaves the point table (msda_forward_workspace_bytes(..., MSDA_FLAG_PROLOGUE)) */
int msda_forward_prologue_ws_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start,
                                 const float *reference_points, const float *sampling_offsets, const float *attn_logits,
                                 int N, int S, int M, int D, int L, int Lq, int P, long long ld_offsets, long long ld_logits,
                                 float *out, float *sampling_loc_out, float *attn_weight_out, void *workspace,
                                 unsigned long long workspace_bytes, msda_stream_t stream);
/* same with flags / scratch (MSDA_FLAG_DETERMINISTIC, MSDA_FLAG_FORWARD_TABLE, msda_backward_workspace_bytes) */
int msda_backward_prologue_ws_f32(const float *grad_out, const float *value, const int64_t *spatial_shapes,
                                  const int64_t *level_start, const float *sampling_loc, const float *attn_weight,
                                  int N, int S, int M, int D, int L, int Lq, int P, long long ld_grad_offsets,
                                  long long ld_grad_logits, float *grad_value, float *grad_sampling_offsets,
                                  float *grad_attn_logits, float *grad_reference_points, void *workspace,
                                  unsigned long long workspace_bytes, unsigned flags, msda_stream_t stream);

/* bf16 rows through the fused prologue (value / out / grad_out in bf16; offsets, logits, reference points and EVERY
 * gradient in fp32 — grad_value too: its passes accumulate in fp32 and a caller whose value_proj runs in fp32 wants it so).
 * Same geometry rule (msda_prologue_supported), same row strides, same flags / workspace as the f32 entry points. */
int msda_forward_prologue_bf16(const uint16_t *value, const int64_t *spatial_shapes, const int64_t *level_start,
                               const float *reference_points, const float *sampling_offsets, const float *attn_logits,
                               int N, int S, int M, int D, int L, int Lq, int P, long long ld_offsets, long long ld_logits,
                               uint16_t *out, float *sampling_loc_out, float *attn_weight_out, msda_stream_t stream);
int msda_forward_prologue_ws_bf16(const uint16_t *value, const int64_t *spatial_shapes, const int64_t *level_start,
                                  const float *reference_points, const float *sampling_offsets, const float *attn_logits,
                                  int N, int S, int M, int D, int L, int Lq, int P, long long ld_offsets, long long ld_logits,
                                  uint16_t *out, float *sampling_loc_out, float *attn_weight_out, void *workspace,
                                  unsigned long long workspace_bytes, msda_stream_t stream);
int msda_backward_prologue_bf16_gv32(const uint16_t *grad_out, const uint16_t *value, const int64_t *spatial_shapes,
                                     const int64_t *level_start, const float *sampling_loc, const float *attn_weight,
                                     int N, int S, int M, int D, int L, int Lq, int P, long long ld_grad_offsets,
                                     long long ld_grad_logits, float *grad_value, float *grad_sampling_offsets,
                                     float *grad_attn_logits, float *grad_reference_points, void *workspace,
                                     unsigned long long workspace_bytes, unsigned flags, msda_stream_t stream);

/* ---- Bracketing projections ----------------------------------------------------
 * Weight and bias gradient of an fp32 nn.Linear  y[M,N] = x[M,K] . W[N,K]^T + b[N]:
 *     grad_weight[N,K] = grad_out[M,N]^T . input[M,K]        grad_bias[N] = sum_m grad_out[m,:]
 * i.e. the GEMM autograd issues for value_proj / sampling_offsets / attention_weights / output_proj
 * (models/ops/modules/ms_deform_attn.py:96,100,101,139), as a split-M fp32-MFMA kernel with a
 * fixed-order reduction (bitwise reproducible).  N and K must be multiples of 4; grad_bias may be
 * NULL; `workspace` must hold msda_linear_wgrad_workspace_bytes(M, N, K) bytes (may be 0 -> NULL ok).
 * The forward GEMM and the input gradient: msda_linear_forward_f32 / msda_linear_dgrad_f32 below. */
unsigned long long msda_linear_wgrad_workspace_bytes(int M, int N, int K);
int msda_linear_wgrad_f32(const float *grad_out, const float *input, int M, int N, int K,
                          float *grad_weight, float *grad_bias, void *workspace, msda_stream_t stream);

/* Padding mask of value_proj (models/ops/modules/ms_deform_attn.py:97-98:
 * value = value.masked_fill(input_padding_mask[..., None], 0)) without the two full passes over
 * [N, S, C] that masked_fill and its backward cost:
 *   msda_zero_masked_rows_f32     x[r, :] = 0 where row_mask[r] != 0, in place (forward: on the GEMM output;
 *                                 backward: on the input gradient) — touches the mask and the masked rows only;
 *   msda_linear_wgrad_masked_f32  msda_linear_wgrad_f32 with the rows of grad_out where row_mask[r] != 0
 *                                 taken as zero (row_mask may be NULL = no mask).
 * row_mask: one byte per row (a torch.bool tensor's storage).  cols must be a multiple of 4, x 16-byte aligned. */
int msda_linear_wgrad_masked_f32(const float *grad_out, const float *input, const uint8_t *row_mask, int M, int N, int K,
                                 float *grad_weight, float *grad_bias, void *workspace, msda_stream_t stream);
/* bf16 operands (what the layer sees under torch.autocast(bfloat16)), fp32 products, accumulation and results: the weight
 * gradient reaches the fp32 master parameter without a rounding to bf16 in between, at half the operand bytes.  Same
 * workspace size as the f32 entry point; grad_out / input 8-byte aligned. */
int msda_linear_wgrad_masked_bf16(const uint16_t *grad_out, const uint16_t *input, const uint8_t *row_mask, int M, int N, int K,
                                  float *grad_weight, float *grad_bias, void *workspace, msda_stream_t stream);
/* `count` (1..4) weight gradients in one call: the first stages one after the other, then ONE fixed-order second stage for
 * all of them (the module's backward has three: output_proj, the merged projection, value_proj — at decoder sizes the three
 * separate second stages cost as much as a first stage).  Host arrays of `count` entries; row_mask / grad_bias may be NULL
 * (array) or hold NULL entries; workspace[p] as for msda_linear_wgrad_f32 with that problem's sizes.  Results are bitwise
 * those of `count` msda_linear_wgrad_masked_{f32,bf16} calls.
 *   msda_linear_wgrad_multi      operands fp32 or bf16 PER PROBLEM: operands_bf16[p] != 0 -> grad_out[p] / input[p] are bf16
 *                                (uint16_t), as under autocast, where the merged projection stays fp32 and the other two are bf16;
 *                                operands_bf16 == NULL: all fp32;
 *   msda_linear_wgrad_multi_f32  the all-fp32 spelling. */
int msda_linear_wgrad_multi(int count, const void *const *grad_out, const void *const *input, const int *operands_bf16,
                            const uint8_t *const *row_mask, const int *M, const int *N, const int *K, float *const *grad_weight,
                            float *const *grad_bias, void *const *workspace, msda_stream_t stream);
int msda_linear_wgrad_multi_f32(int count, const float *const *grad_out, const float *const *input, const uint8_t *const *row_mask,
                                const int *M, const int *N, const int *K, float *const *grad_weight, float *const *grad_bias,
                                void *const *workspace, msda_stream_t stream);
int msda_zero_masked_rows_f32(float *x, const uint8_t *row_mask, long long rows, int cols, msda_stream_t stream);
/* Up to four fp32 -> bf16 conversions (round to nearest even) in ONE launch: the weights and biases of value_proj / output_proj
 * that torch.autocast(bfloat16) casts on every call (models/ops/modules/ms_deform_attn.py:96,139 under the reference's --amp) —
 * four ~3 us kernels otherwise.  src / dst / n: host arrays of `count` (1..4) device pointers and element counts (each a
 * multiple of 2; sources 8-byte, destinations 4-byte aligned). */
int msda_cast_bf16_multi_f32(int count, const float *const *src, uint16_t *const *dst, const long long *n, msda_stream_t stream);

/* Forward and input gradient of the same fp32 layers (nn.Linear semantics, models/ops/modules/ms_deform_attn.py:96,100,101,139
 * and what autograd derives for them):
 *     output[rows, out]    = input[rows, in] . weight[out, in]^T + bias[out]        (bias may be NULL)
 *     grad_input[rows, in] = grad_out[rows, out] . weight[out, in]
 * as one plain fp32-MFMA launch each (exact fp32 products, fixed summation order: bitwise reproducible) — at this module's
 * shapes the vendor BLAS path costs more on the host (~27 us per GEMM through torch) than these kernels run for.  row_mask
 * (may be NULL; one byte per row): rows with a non-zero byte are WRITTEN AS ZEROS — value.masked_fill(padding_mask, 0) after
 * value_proj (modules/ms_deform_attn.py:97-98) and its backward, without a second pass.  out_features and in_features must
 * be multiples of 4, all operands 16-byte aligned and contiguous. */
int msda_linear_forward_f32(const float *input, const float *weight, const float *bias, const uint8_t *row_mask, long long rows,
                            int out_features, int in_features, float *output, msda_stream_t stream);
int msda_linear_dgrad_f32(const float *grad_out, const float *weight, const uint8_t *row_mask, long long rows, int out_features,
                          int in_features, float *grad_input, msda_stream_t stream);

/* Thread-local description of the last failure on the calling thread ("" if none). */
/* ---- Residual add + LayerNorm of the layers around the op --------------------------
 * y = LayerNorm(x + residual) * gamma + beta over rows of width d (fp32, d a multiple of 4, <= 1024) — the
 * `x = x + dropout(x2); x = norm(x)` pairs of the reference's encoder / decoder layers
 * (models/arctic_transformer.py:279-282, 294-295, 366-368, 377-378, 385-386) as one pass instead of an add kernel
 * and a LayerNorm kernel; `residual` may be NULL (plain LayerNorm).  The forward also returns mean[rows] and
 * rstd[rows]; the backward recomputes x + residual, writes grad_sum[rows, d] (the gradient of x and of residual
 * alike) and grad_gamma / grad_beta through per-workgroup partial sums in `workspace`
 * (msda_add_layernorm_workspace_bytes) combined in a fixed order — reproducible, no float atomics.  Dropout is not
 * part of it: the caller applies the framework's dropout to `residual` first, so its random stream is untouched. */
unsigned long long msda_add_layernorm_workspace_bytes(long long rows, int d);
int msda_add_layernorm_forward_f32(const float *x, const float *residual, const float *gamma, const float *beta, long long rows,
                                   int d, float eps, float *y, float *mean, float *rstd, msda_stream_t stream);
int msda_add_layernorm_backward_f32(const float *grad_y, const float *x, const float *residual, const float *gamma,
                                    const float *mean, const float *rstd, long long rows, int d, float *grad_sum,
                                    float *grad_gamma, float *grad_beta, void *workspace, msda_stream_t stream);

/* ---- FFN of the layers (models/arctic_transformer.py:283-287, :366-370) ---------------------------
 * linear2(dropout(relu(linear1(x)))): with act = dropout(relu(h)) — the tensor linear2 consumed, saved for its weight gradient
 * anyway — the gradient with respect to h is grad * scale * (act > 0), scale = 1 / (1 - p): one in-place pass over
 * `grad` [n floats, n a multiple of 4, both pointers 16-byte aligned] instead of PyTorch's masked_scale + threshold_backward,
 * and neither the dropout mask nor relu's output has to be kept. */
int msda_relu_dropout_backward_f32(float *grad, const float *act, float scale, long long n, msda_stream_t stream);

/* ---- Decoder self-attention core (models/arctic_transformer.py:351,374-376) -------------------------------
 * nn.MultiheadAttention over the 300 queries (8 heads of 32, batch = frames) computes, between its in- and out-projections,
 *     out = dropout(softmax(q k^T * scale), p) v          for N*H independent (batch, head) problems.
 * These entry points are that core for head_dim 32, fp32, 1 <= Lq, Lk <= 320 (msda_attn32_supported), with neither the
 * [N*H, Lq, Lk] score tensor nor a dropout mask in memory: the forward keeps log-sum-exp per (batch, head, query) [N*H, Lq],
 * the backward (two launches: dK/dV, dQ) recomputes the probabilities and the mask from it.
 * Tensors are views with the head's 32 channels contiguous: element (n, h, l, d) at p + n*sn + h*32 + l*sl + d (strides in
 * floats, multiples of 4; 16-byte aligned bases) — the column blocks of a packed in-projection output [L, N, 3E] are such views.
 * Dropout: keep(seed, pair, query, key) is an integer hash compared with p * 2^32 — the kernel's own random stream.  `seed` is a
 * DEVICE pointer to one 64-bit value (drawn by the caller with its generator; the same value must reach the backward); NULL is
 * allowed when dropout_p == 0.  No attention mask / key-padding mask (the reference's decoder passes none). */
int msda_attn32_supported(int Lq, int Lk, int head_dim);
int msda_attn32_forward_f32(const float *q, long long q_sn, long long q_sl, const float *k, long long k_sn, long long k_sl,
                            const float *v, long long v_sn, long long v_sl, int N, int H, int Lq, int Lk, float scale,
                            float dropout_p, const unsigned long long *seed, float *out, long long o_sn, long long o_sl, float *lse,
                            msda_stream_t stream);
int msda_attn32_backward_f32(const float *q, long long q_sn, long long q_sl, const float *k, long long k_sn, long long k_sl,
                             const float *v, long long v_sn, long long v_sl, const float *out, long long o_sn, long long o_sl,
                             const float *lse, const float *grad_out, long long go_sn, long long go_sl, int N, int H, int Lq, int Lk,
                             float scale, float dropout_p, const unsigned long long *seed, float *grad_q, long long gq_sn,
                             long long gq_sl, float *grad_k, long long gk_sn, long long gk_sl, float *grad_v, long long gv_sn,
                             long long gv_sl, msda_stream_t stream);

/* ---- Transformer input assembly -----------------------------------------------------
 * The flatten block of DeformableTransformer.forward (models/arctic_transformer.py:162-173): per level
 * src_l[N,C,H,W] -> rows [level_start_l, level_start_l + H*W) of src_flatten[N,S,C], and pos_l the same way with
 * level_embed[l][C] added — one tiled-transpose launch for all levels and both tensors instead of the reference's
 * strided torch.cat copies and the add.  `src_levels` / `pos_levels` are HOST arrays of L device pointers, `heights` /
 * `widths` host arrays (the caller knows the feature-map shapes as Python ints; nothing is read back from the device);
 * either tensor family may be NULL.  msda_unflatten_levels_f32 is the inverse copy (the backward: flattened gradient rows
 * back into per-level NCHW gradients, and — with grad_level_embed and a workspace of msda_unflatten_workspace_bytes — the
 * level-embedding gradient as per-tile column sums combined in a fixed order).  fp32, C a multiple of 4, L <= 16. */
int msda_flatten_levels_f32(int L, const float *const *src_levels, const float *const *pos_levels, const float *level_embed,
                            const int *heights, const int *widths, int N, int C, float *src_flatten, float *pos_flatten,
                            msda_stream_t stream);
int msda_unflatten_levels_f32(int L, float *const *grad_src_levels, float *const *grad_pos_levels, const int *heights,
                              const int *widths, int N, int C, const float *grad_src_flatten, const float *grad_pos_flatten,
                              float *grad_level_embed /* [L, C] or NULL */, void *workspace, msda_stream_t stream);
unsigned long long msda_unflatten_workspace_bytes(int L, const int *heights, const int *widths, int N, int C);

const char *msda_last_error(void);

/* Library/ABI version (major*100 + minor) and the kernel family a geometry maps to.  MSDA_ABI_VERSION is what a binding
 * compiled against THIS header expects msda_version() to return at run time (uvhand_amd/_ext.py compares the two);
 * it changes whenever a declaration in this file does. */
#define MSDA_ABI_VERSION 115
int msda_version(void);
int msda_path_for(int elem_bytes, int M, int D, int L, int P);

/* Diagnostics (tests, bench, tools/ktime.py --sweep): which launch plan a call of this geometry takes, as text, e.g.
 * "fwd=lds(chunks=2,qw=528,...) bwd=fused_lds(acc=wide,W=2,...)" — the same host-side plan functions the launchers run,
 * nothing is launched.  row_bytes 4 (fp32 rows) or 2 (bf16 rows; grad_value_bytes then 2 or 4); flags as for
 * msda_backward_workspace_bytes (MSDA_FLAG_PROLOGUE: the fused-prologue entry points; has_workspace: whether the caller
 * passes the scratch that call can use).  "generic" outside the D = 32 family.  Returns the length written (NUL-terminated,
 * truncated to buf_len - 1).  No reference counterpart: the reference's dispatch is the switch over `channels` at
 * models/ops/src/cuda/ms_deform_im2col_cuda.cuh:971-1320. */
int msda_describe_plan(int row_bytes, int grad_value_bytes, int N, int S, int M, int D, int L, int Lq, int P, unsigned flags,
                       int has_workspace, char *buf, int buf_len);

/* Measurement helper (bench.py `roofline.secondary`): gathers pseudo-random rows of `table` (row_bytes 128 = fp32 rows as 8 lanes
 * x 16 B, or 64 = bf16 rows as 8 lanes x 8 B; the largest power-of-two row count that fits table_bytes) with the access
 * pattern of the sampling kernels — independent 8-lane row requests, sixteen in flight per lane — and nothing else:
 * *rows_gathered / elapsed time is the row-request rate the vector memory path of this box delivers on a table of that size,
 * the practical ceiling of a gather kernel on cache-resident data.  `sink`: one float the kernel never writes for a finite
 * table.  `blocks` x 256 threads, `iters` x 16 rows per 8-lane group.  No reference counterpart; no product path calls it. */
int msda_probe_row_gather(const void *table, unsigned long long table_bytes, int row_bytes, int blocks, int iters, float *sink,
                          unsigned long long *rows_gathered, msda_stream_t stream);

/* Diagnostic: kernel launches this process has enqueued through the library so far (one count per launcher call that reached
 * the device; a monotonically increasing relaxed atomic — the only process-wide state the library keeps, never read by a
 * kernel path).  Tests use the difference around a call, e.g. "a frozen projection costs no weight-gradient launch". */
unsigned long long msda_launch_count(void);

/* Test hook: force the kernel family for the CALLING THREAD's subsequent calls (-1 = automatic, default; MSDA_PATH_GENERIC).
 * Thread-local, so no caller can change the kernels under another thread's launch; not meant for production callers. */
void msda_force_path(int path);

#ifdef __cplusplus
}
#endif
#endif /* MSDA_H_ */
