// Host-side plumbing shared by the kernel translation units: the thread-local error
// string behind msda_last_error() and the declarations of the per-family launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "msda.h"
#include "msda_common.h"

namespace msda {

// Tuning / A-B knobs (MSDA_SPLIT, MSDA_XCD, MSDA_BWD_MODE, MSDA_BWD_WGS, ... tools/README.md) exist only in DIAGNOSTIC
// builds (-DMSDA_TUNING: `make -C uvhand_amd/csrc tuning`, tools/micro/kbench.cpp).  The shipped library reads no
// environment variable and keeps no process-wide state: these helpers then return the default.
#ifdef MSDA_TUNING
#include <cstdlib>
inline const char *tuning_str(const char *name) { const char *v = getenv(name); return (v && *v) ? v : nullptr; }
#else
inline const char *tuning_str(const char *) { return nullptr; }
#endif
inline int tuning_int(const char *name, int dflt) { const char *v = tuning_str(name); return v ? atoi(v) : dflt; }

// Records `msg` for msda_last_error() on this thread and returns `code`.
int set_error(int code, const char *msg);
// Returns MSDA_OK, or records hipGetLastError() (prefixed by `what`) and returns MSDA_ERR_LAUNCH.
int check_launch(const char *what);

// ---- generic family (msda_generic.hip): any D, float / double; VT = T, or uint16_t (bf16 rows) with T = float --------
template <typename T, typename VT>
int launch_fwd_generic(const VT *value, const int64_t *shapes, const int64_t *level_start,
                       const T *loc, const T *attn, int N, int S, int M, int D, int L, int Lq, int P,
                       VT *out, hipStream_t stream);
template <typename T, typename VT>
int launch_bwd_generic(const VT *grad_out, const VT *value, const int64_t *shapes,
                       const int64_t *level_start, const T *loc, const T *attn, int N, int S, int M,
                       int D, int L, int Lq, int P, T *grad_value, T *grad_loc, T *grad_attn,
                       hipStream_t stream, bool deterministic = false);

// ---- D = 32 fp32 family (msda_d32.hip): the model's shape ------------------------------------
bool d32_supported(int N, int S, int M, int D, int L, int Lq, int P);
int launch_fwd_d32(const float *value, const int64_t *shapes, const int64_t *level_start,
                   const float *loc, const float *attn, int N, int S, int M, int L, int Lq, int P,
                   float *out, hipStream_t stream, void *table = nullptr);
int launch_bwd_d32(const float *grad_out, const float *value, const int64_t *shapes,
                   const int64_t *level_start, const float *loc, const float *attn, int N, int S,
                   int M, int L, int Lq, int P, float *grad_value, float *grad_loc, float *grad_attn,
                   hipStream_t stream, void *workspace = nullptr, size_t ws_bytes = 0, bool deterministic = false,
                   const void *table = nullptr, bool no_dense = false);
// the point table a small problem's forward can leave for its backward (bytes; 0 = the backward's plan reads none); `table`
// arguments of the launchers below: that table (forward: written, backward: read), or null
size_t forward_table_bytes(int N, int S, int M, int D, int L, int Lq, int P, bool prologue);
// clears the 'written' stamp of a table buffer that a forward call could not fill (see msda_d32.hip)
int invalidate_forward_table(void *table, int N, int S, int M, int L, int Lq, int P, hipStream_t stream);
// scratch the D = 32 backward can use to cut long levels into query chunks (0 = none needed); see msda.h
size_t backward_workspace_bytes(int N, int S, int M, int D, int L, int Lq, int P, unsigned flags);

// bf16 storage (uint16_t bits) of value / out / grad_out / grad_value; loc, attn and their gradients fp32.
int launch_fwd_d32_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *level_start,
                        const float *loc, const float *attn, int N, int S, int M, int L, int Lq, int P,
                        uint16_t *out, hipStream_t stream, void *table = nullptr);
int launch_bwd_d32_bf16(const uint16_t *grad_out, const uint16_t *value, const int64_t *shapes,
                        const int64_t *level_start, const float *loc, const float *attn, int N, int S,
                        int M, int L, int Lq, int P, uint16_t *grad_value, float *grad_loc, float *grad_attn,
                        hipStream_t stream, void *workspace = nullptr, size_t ws_bytes = 0, bool deterministic = false,
                        const void *table = nullptr, bool no_dense = false);

// bf16 rows in, fp32 grad_value out: nothing is rounded between the passes of a multi-pass backward (and a
// caller whose value tensor is fp32 needs no conversion of the result)
int launch_bwd_d32_bf16_gv32(const uint16_t *grad_out, const uint16_t *value, const int64_t *shapes,
                             const int64_t *level_start, const float *loc, const float *attn, int N, int S, int M, int L,
                             int Lq, int P, float *grad_value, float *grad_loc, float *grad_attn, hipStream_t stream,
                             void *workspace = nullptr, size_t ws_bytes = 0, bool deterministic = false, const void *table = nullptr,
                             bool no_dense = false);
// number of query chunks ("passes") role B of the D = 32 backward takes for Lq*P sampling points per (b, m, l)
int backward_passes(int Lq, int P);
// text form of the launch plan of a D = 32 geometry (msda_describe_plan); returns the length written
int describe_plan(int row_bytes, int gv_bytes, int N, int S, int M, int L, int Lq, int P, bool prologue, bool has_ws, bool det,
                  char *buf, int len, bool no_dense = false);

// ---- fused prologue (fp32, D = 32 family): softmax over L*P and loc = ref + offset/(W,H) inside the kernels.
// ld_* = floats between consecutive (batch, query) rows of the raw offsets / logits and of their gradients
// (validated by the ABI layer: >= the dense width, offsets' even).
bool prologue_supported(int N, int S, int M, int D, int L, int Lq, int P);
int launch_fwd_prologue(const float *value, const int64_t *shapes, const int64_t *level_start, const float *ref,
                        const float *offsets, const float *logits, int N, int S, int M, int L, int Lq, int P,
                        long long ld_offsets, long long ld_logits, float *out, float *loc_out, float *attn_out,
                        hipStream_t stream, void *table = nullptr);
int launch_bwd_prologue(const float *grad_out, const float *value, const int64_t *shapes, const int64_t *level_start,
                        const float *loc, const float *attn, int N, int S, int M, int L, int Lq, int P, float *grad_value,
                        long long ld_grad_offsets, long long ld_grad_logits, float *grad_offsets, float *grad_logits,
                        float *grad_ref, hipStream_t stream, void *workspace = nullptr, size_t ws_bytes = 0, bool deterministic = false,
                        const void *table = nullptr, bool no_dense = false);

// bf16 rows (value, out, grad_out); offsets / logits / reference points and every gradient fp32 (grad_value included)
int launch_fwd_prologue_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *level_start, const float *ref,
                             const float *offsets, const float *logits, int N, int S, int M, int L, int Lq, int P,
                             long long ld_offsets, long long ld_logits, uint16_t *out, float *loc_out, float *attn_out,
                             hipStream_t stream, void *table = nullptr);
int launch_bwd_prologue_bf16(const uint16_t *grad_out, const uint16_t *value, const int64_t *shapes, const int64_t *level_start,
                             const float *loc, const float *attn, int N, int S, int M, int L, int Lq, int P, float *grad_value,
                             long long ld_grad_offsets, long long ld_grad_logits, float *grad_offsets, float *grad_logits,
                             float *grad_ref, hipStream_t stream, void *workspace = nullptr, size_t ws_bytes = 0,
                             bool deterministic = false, const void *table = nullptr,
                             bool no_dense = false);

// ---- weight / bias gradient of the bracketing nn.Linear layers (msda_linear.hip) -----------------
size_t linear_wgrad_workspace_bytes(int M, int N, int K);
// row_mask (may be null): one byte per row of dY, non-zero = that row counts as zero
int launch_linear_wgrad(const float *dY, const float *X, const uint8_t *row_mask, int M, int N, int K, float *dW, float *db,
                        float *workspace, hipStream_t stream);
// several fp32 weight gradients: first stages back to back, ONE second stage for all (count <= 4; per-problem arrays)
int launch_linear_wgrad_multi(int count, const void *const *dY, const void *const *X, const int *bf16, const uint8_t *const *row_mask,
                              const int *M, const int *N, const int *K, float *const *dW, float *const *db, float *const *workspace,
                              hipStream_t stream);
// bf16 operands (uint16_t bits), fp32 products / accumulation / results
int launch_linear_wgrad_bf16(const uint16_t *dY, const uint16_t *X, const uint8_t *row_mask, int M, int N, int K, float *dW,
                             float *db, float *workspace, hipStream_t stream);
int launch_zero_masked_rows(float *x, const uint8_t *mask, long long rows, int cols, hipStream_t stream);
// up to four fp32 -> bf16 conversions in one launch (msda_linear.hip)
int launch_cast_bf16_multi(int count, const float *const *src, uint16_t *const *dst, const long long *n, hipStream_t stream);

// ---- forward and input gradient of the same layers (msda_gemm.hip); row_mask (may be null): rows written as zeros ----
int launch_linear_forward(const float *x, const float *w, const float *bias, const uint8_t *row_mask, long long rows,
                          int out_features, int in_features, float *y, hipStream_t stream);
int launch_linear_dgrad(const float *grad_out, const float *w, const uint8_t *row_mask, long long rows, int out_features,
                        int in_features, float *grad_in, hipStream_t stream);

// ---- transformer input assembly (msda_flatten.hip): per-level NCHW <-> the flattened [N, S, C] layout ----
struct FlattenPlan {
    int L;
    float *src[kMaxLevels];          // per level [N, C, H_l, W_l] (device), or null
    float *pos[kMaxLevels];
    int hw[kMaxLevels], start[kMaxLevels], first_block[kMaxLevels];
};
int launch_flatten_levels(const FlattenPlan &plan, int N, int C, int S, const float *level_embed, float *src_flat, float *pos_flat,
                          bool unflatten, hipStream_t stream, float *grad_level_embed = nullptr, float *workspace = nullptr);
size_t unflatten_workspace_bytes(const FlattenPlan &plan, int N, int C);

// ---- residual add + LayerNorm of the layers around the op (msda_layernorm.hip) ---------------------
size_t add_layernorm_workspace_bytes(long long rows, int d);
int launch_add_layernorm_fwd(const float *x, const float *res, const float *gamma, const float *beta, long long rows, int d,
                             float eps, float *y, float *mean, float *rstd, hipStream_t stream);
int launch_add_layernorm_bwd(const float *dy, const float *x, const float *res, const float *gamma, const float *mean,
                             const float *rstd, long long rows, int d, float *ds, float *dgamma, float *dbeta, float *workspace,
                             hipStream_t stream);

// ---- FFN of the layers: backward of dropout(relu(h)) in one in-place pass (msda_layernorm.hip) ----
int launch_relu_dropout_bwd(float *grad, const float *act, float scale, long long n, hipStream_t stream);

}  // namespace msda
