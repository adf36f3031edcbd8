// C ABI of libmsda_hip.so (declared in include/msda.h): argument validation, kernel-family
// selection, error reporting.  No torch types, no allocation, no device synchronisation.
#include <cstdio>
#include <cstring>
#include <string>

#include <atomic>

#include "msda_launch.h"

namespace msda {

static thread_local char g_err[512] = "";
static thread_local int g_force_path = -1;          // msda_force_path(): test hook, this thread's calls only

int set_error(int code, const char *msg)
{
    std::snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "unknown error");
    return code;
}

// Start of every entry point that enqueues work: forget this thread's previous message and any sticky
// error an unrelated earlier HIP call left behind, so that check_launch() reports THIS call's launch only.
static void begin_call()
{
    g_err[0] = 0;
    (void)hipGetLastError();
}

// Launches enqueued by this process through the library (every launcher ends in check_launch, once per kernel it queued or
// per pair of dependent kernels): a diagnostic for tests ("a frozen projection costs no weight-gradient launch"), relaxed atomic.
static std::atomic<unsigned long long> g_launch_checks{0};
unsigned long long launch_checks() { return g_launch_checks.load(std::memory_order_relaxed); }

int check_launch(const char *what)
{
    g_launch_checks.fetch_add(1, std::memory_order_relaxed);
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MSDA_OK;
    char buf[400];
    std::snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    return set_error(MSDA_ERR_LAUNCH, buf);
}

static int check_args(const void *const *ptrs, int nptrs, int N, int S, int M, int D, int L, int Lq,
                      int P)
{
    if (N < 0 || S < 0 || M <= 0 || D <= 0 || L <= 0 || Lq < 0 || P <= 0)
        return set_error(MSDA_ERR_ARGUMENT, "msda: sizes must be positive (N, S, Lq may be 0)");
    if (N == 0 || Lq == 0 || S == 0) return MSDA_OK;               // nothing to read
    for (int i = 0; i < nptrs; ++i)
        if (ptrs[i] == nullptr) return set_error(MSDA_ERR_ARGUMENT, "msda: null device pointer");
    return MSDA_OK;
}

// Row strides of the fused-prologue tensors: 0 selects the dense layout; otherwise at least the dense
// width, and for the (x, y) pairs an even stride on an 8-byte aligned base (the kernels move float2).
static int check_row_strides(const char *who, int M, int L, int P, const void *offsets, long long *ld_offsets,
                             long long *ld_logits)
{
    const long long dense = (long long)M * L * P;
    if (*ld_offsets == 0) *ld_offsets = 2 * dense;
    if (*ld_logits == 0) *ld_logits = dense;
    char buf[200];
    if (*ld_offsets < 2 * dense || *ld_logits < dense || (*ld_offsets & 1) || ((uintptr_t)offsets & 7) ||
        *ld_offsets - 2 * dense > 0x7fffffffLL || *ld_logits - dense > 0x7fffffffLL) {
        std::snprintf(buf, sizeof(buf), "%s: row strides (%lld, %lld) must be >= (%lld, %lld), the first even on an "
                      "8-byte aligned base", who, *ld_offsets, *ld_logits, 2 * dense, dense);
        return set_error(MSDA_ERR_ARGUMENT, buf);
    }
    return MSDA_OK;
}

// The D = 32 kernels move rows as 16-byte (fp32) / 8-byte (bf16) vectors and locations as (x, y) pairs.
// Tensors straight from an allocator always qualify; a contiguous VIEW at an odd element offset does not —
// fp32 / bf16 op calls then take the generic kernels (element-wise accesses); the prologue and wgrad entries refuse.
static bool aligned_to(const void *p, size_t bytes) { return ((uintptr_t)p & (bytes - 1)) == 0; }

static int refuse_unaligned(const char *who)
{
    char buf[200];
    std::snprintf(buf, sizeof(buf), "%s: row tensors must be 16-byte (fp32) / 8-byte (bf16) aligned and (x, y) tensors "
                  "8-byte aligned (a contiguous view at an odd element offset is not)", who);
    return set_error(MSDA_ERR_ARGUMENT, buf);
}

static bool use_d32(int N, int S, int M, int D, int L, int Lq, int P)
{
    const int f = g_force_path;
    if (f == MSDA_PATH_GENERIC) return false;
    return d32_supported(N, S, M, D, L, Lq, P);
}

// MSDA_FLAG_FORWARD_TABLE: `workspace` starts with the point table a msda_forward_ws_* call of the same geometry filled
// (include/msda.h).  Used only where the backward's plan reads one and the caller's buffer holds all of it.
static const void *table_of(const void *workspace, size_t ws_bytes, unsigned flags, int N, int S, int M, int D, int L, int Lq, int P,
                            bool prologue)
{
    if (!(flags & MSDA_FLAG_FORWARD_TABLE) || !workspace || !aligned_to(workspace, 16)) return nullptr;
    const size_t need = forward_table_bytes(N, S, M, D, L, Lq, P, prologue);
    return (need > 0 && ws_bytes >= need) ? workspace : nullptr;
}

// The scratch part of a backward call's workspace: all of it, or — with MSDA_FLAG_FORWARD_TABLE on a geometry whose forward
// leaves a table — what follows the table (include/msda.h: the table first, rounded up to 256 bytes, then the scratch).
struct Scratch { void *p; size_t bytes; };
static size_t table_span(int N, int S, int M, int D, int L, int Lq, int P, bool prologue)
{
    return (forward_table_bytes(N, S, M, D, L, Lq, P, prologue) + 255) & ~(size_t)255;
}
static Scratch scratch_of(void *workspace, size_t ws_bytes, unsigned flags, int N, int S, int M, int D, int L, int Lq, int P, bool prologue)
{
    if (!workspace || !(flags & MSDA_FLAG_FORWARD_TABLE)) return Scratch{workspace, workspace ? ws_bytes : 0};
    const size_t span = table_span(N, S, M, D, L, Lq, P, prologue);
    if (ws_bytes <= span) return Scratch{nullptr, 0};
    return Scratch{static_cast<unsigned char *>(workspace) + span, ws_bytes - span};
}

template <typename T>
static int forward_impl(const T *value, const int64_t *shapes, const int64_t *level_start,
                        const T *loc, const T *attn, int N, int S, int M, int D, int L, int Lq, int P,
                        T *out, hipStream_t stream, bool d32, void *table = nullptr)
{
    const void *ptrs[] = {value, shapes, level_start, loc, attn, out};
    if (int rc = check_args(ptrs, 6, N, S, M, D, L, Lq, P)) return rc;
    begin_call();
    if (N == 0 || Lq == 0) return MSDA_OK;                          // empty output
    if (S == 0) {                                                   // no pixels: every tap is outside
        const hipError_t e = hipMemsetAsync(out, 0, sizeof(T) * (size_t)N * Lq * M * D, stream);
        return e == hipSuccess ? MSDA_OK : set_error(MSDA_ERR_LAUNCH, hipGetErrorString(e));
    }
    if constexpr (sizeof(T) == 4) {
        if (d32 && aligned_to(value, 16) && aligned_to(out, 16) && aligned_to(loc, 8))
            return launch_fwd_d32(value, shapes, level_start, loc, attn, N, S, M, L, Lq, P, out, stream, table);
    }
    // (the generic kernels write no table: a buffer the caller handed over gets its stamp cleared)
    if (table) if (int rc = invalidate_forward_table(table, N, S, M, L, Lq, P, stream)) return rc;
    return launch_fwd_generic<T>(value, shapes, level_start, loc, attn, N, S, M, D, L, Lq, P, out, stream);
}

template <typename T>
static int backward_impl(const T *grad_out, const T *value, const int64_t *shapes,
                         const int64_t *level_start, const T *loc, const T *attn, int N, int S, int M,
                         int D, int L, int Lq, int P, T *grad_value, T *grad_loc, T *grad_attn,
                         hipStream_t stream, bool d32, void *workspace = nullptr, size_t ws_bytes = 0, unsigned flags = 0)
{
    const void *ptrs[] = {grad_out, value, shapes, level_start, loc, attn, grad_value, grad_loc, grad_attn};
    if (int rc = check_args(ptrs, 9, N, S, M, D, L, Lq, P)) return rc;
    begin_call();
    if (N == 0) return MSDA_OK;
    if (Lq == 0 || S == 0) {                                        // no contributions at all
        hipError_t e = hipSuccess;
        if (S > 0) e = hipMemsetAsync(grad_value, 0, sizeof(T) * (size_t)N * S * M * D, stream);
        if (e == hipSuccess && Lq > 0) {
            e = hipMemsetAsync(grad_loc, 0, sizeof(T) * (size_t)N * Lq * M * L * P * 2, stream);
            if (e == hipSuccess)
                e = hipMemsetAsync(grad_attn, 0, sizeof(T) * (size_t)N * Lq * M * L * P, stream);
        }
        return e == hipSuccess ? MSDA_OK : set_error(MSDA_ERR_LAUNCH, hipGetErrorString(e));
    }
    if constexpr (sizeof(T) == 4) {
        if (d32 && aligned_to(grad_out, 16) && aligned_to(value, 16) && aligned_to(grad_value, 16) && aligned_to(loc, 8) &&
            aligned_to(grad_loc, 8))
        {
            const Scratch sc = scratch_of(workspace, ws_bytes, flags, N, S, M, D, L, Lq, P, false);
            return launch_bwd_d32(grad_out, value, shapes, level_start, loc, attn, N, S, M, L, Lq, P,
                                  grad_value, grad_loc, grad_attn, stream, sc.p, sc.bytes,
                                  (flags & MSDA_FLAG_DETERMINISTIC) != 0,
                                  table_of(workspace, ws_bytes, flags, N, S, M, D, L, Lq, P, false),
                                  (flags & MSDA_FLAG_EXACT_NONFINITE) != 0);
        }
    }
    return launch_bwd_generic<T>(grad_out, value, shapes, level_start, loc, attn, N, S, M, D, L, Lq, P,
                                 grad_value, grad_loc, grad_attn, stream, (flags & MSDA_FLAG_DETERMINISTIC) != 0);
}

}  // namespace msda

template <typename GT>
static int backward_bf16_impl(const uint16_t *grad_out, const uint16_t *value, const int64_t *spatial_shapes,
                              const int64_t *level_start, const float *sampling_loc, const float *attn_weight, int N,
                              int S, int M, int D, int L, int Lq, int P, GT *grad_value, float *grad_sampling_loc,
                              float *grad_attn_weight, msda_stream_t stream, void *workspace = nullptr, size_t ws_bytes = 0,
                              unsigned flags = 0)
{
    const void *ptrs[] = {grad_out, value, spatial_shapes, level_start, sampling_loc, attn_weight, grad_value,
                          grad_sampling_loc, grad_attn_weight};
    if (int rc = msda::check_args(ptrs, 9, N, S, M, D, L, Lq, P)) return rc;
    msda::begin_call();
    if (N == 0) return MSDA_OK;
    if (Lq == 0 || S == 0) {
        hipError_t e = hipSuccess;
        if (S > 0) e = hipMemsetAsync(grad_value, 0, sizeof(GT) * (size_t)N * S * M * D, (hipStream_t)stream);
        if (e == hipSuccess && Lq > 0) {
            e = hipMemsetAsync(grad_sampling_loc, 0, 4 * (size_t)N * Lq * M * L * P * 2, (hipStream_t)stream);
            if (e == hipSuccess) e = hipMemsetAsync(grad_attn_weight, 0, 4 * (size_t)N * Lq * M * L * P, (hipStream_t)stream);
        }
        return e == hipSuccess ? MSDA_OK : msda::set_error(MSDA_ERR_LAUNCH, hipGetErrorString(e));
    }
    const bool d32 = msda::use_d32(N, S, M, D, L, Lq, P) && msda::aligned_to(grad_out, 8) && msda::aligned_to(value, 8) &&
                     msda::aligned_to(grad_value, sizeof(GT) * 4) && msda::aligned_to(sampling_loc, 8) &&
                     msda::aligned_to(grad_sampling_loc, 8);
    if constexpr (sizeof(GT) == 2) {
        if (!d32)
            return msda::set_error(MSDA_ERR_ARGUMENT, "msda_backward_bf16: bf16 grad_value needs the D=32 kernel family (D == 32, "
                                                      "L <= 16, L*P <= 32, 8-byte aligned rows); msda_backward_bf16_gv32 serves "
                                                      "every other shape");
        const msda::Scratch sc = msda::scratch_of(workspace, ws_bytes, flags, N, S, M, D, L, Lq, P, false);
        return msda::launch_bwd_d32_bf16(grad_out, value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S, M,
                                         L, Lq, P, grad_value, grad_sampling_loc, grad_attn_weight, (hipStream_t)stream,
                                         sc.p, sc.bytes, (flags & MSDA_FLAG_DETERMINISTIC) != 0,
                                         msda::table_of(workspace, ws_bytes, flags, N, S, M, D, L, Lq, P, false),
                                  (flags & MSDA_FLAG_EXACT_NONFINITE) != 0);
    } else {
        if (!d32)                                                   // element-wise accesses: any D, any element offset
            return msda::launch_bwd_generic<float>(grad_out, value, spatial_shapes, level_start, sampling_loc, attn_weight, N,
                                                   S, M, D, L, Lq, P, grad_value, grad_sampling_loc, grad_attn_weight,
                                                   (hipStream_t)stream, (flags & MSDA_FLAG_DETERMINISTIC) != 0);
        const msda::Scratch sc = msda::scratch_of(workspace, ws_bytes, flags, N, S, M, D, L, Lq, P, false);
        return msda::launch_bwd_d32_bf16_gv32(grad_out, value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S,
                                              M, L, Lq, P, grad_value, grad_sampling_loc, grad_attn_weight,
                                              (hipStream_t)stream, sc.p, sc.bytes, (flags & MSDA_FLAG_DETERMINISTIC) != 0,
                                              msda::table_of(workspace, ws_bytes, flags, N, S, M, D, L, Lq, P, false),
                                  (flags & MSDA_FLAG_EXACT_NONFINITE) != 0);
    }
}

extern "C" {

int msda_forward_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start,
                     const float *sampling_loc, const float *attn_weight, int N, int S, int M, int D,
                     int L, int Lq, int P, float *out, msda_stream_t stream)
{
    return msda::forward_impl<float>(value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S,
                                     M, D, L, Lq, P, out, (hipStream_t)stream,
                                     msda::use_d32(N, S, M, D, L, Lq, P));
}

int msda_backward_f32(const float *grad_out, const float *value, const int64_t *spatial_shapes,
                      const int64_t *level_start, const float *sampling_loc, const float *attn_weight,
                      int N, int S, int M, int D, int L, int Lq, int P, float *grad_value,
                      float *grad_sampling_loc, float *grad_attn_weight, msda_stream_t stream)
{
    return msda::backward_impl<float>(grad_out, value, spatial_shapes, level_start, sampling_loc,
                                      attn_weight, N, S, M, D, L, Lq, P, grad_value, grad_sampling_loc,
                                      grad_attn_weight, (hipStream_t)stream,
                                      msda::use_d32(N, S, M, D, L, Lq, P));
}

int msda_forward_f64(const double *value, const int64_t *spatial_shapes, const int64_t *level_start,
                     const double *sampling_loc, const double *attn_weight, int N, int S, int M, int D,
                     int L, int Lq, int P, double *out, msda_stream_t stream)
{
    return msda::forward_impl<double>(value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S,
                                      M, D, L, Lq, P, out, (hipStream_t)stream, false);
}

int msda_backward_f64(const double *grad_out, const double *value, const int64_t *spatial_shapes,
                      const int64_t *level_start, const double *sampling_loc,
                      const double *attn_weight, int N, int S, int M, int D, int L, int Lq, int P,
                      double *grad_value, double *grad_sampling_loc, double *grad_attn_weight,
                      msda_stream_t stream)
{
    return msda::backward_impl<double>(grad_out, value, spatial_shapes, level_start, sampling_loc,
                                       attn_weight, N, S, M, D, L, Lq, P, grad_value, grad_sampling_loc,
                                       grad_attn_weight, (hipStream_t)stream, false);
}

static int forward_bf16_impl(const uint16_t *value, const int64_t *spatial_shapes, const int64_t *level_start,
                             const float *sampling_loc, const float *attn_weight, int N, int S, int M, int D, int L,
                             int Lq, int P, uint16_t *out, msda_stream_t stream, void *table)
{
    const void *ptrs[] = {value, spatial_shapes, level_start, sampling_loc, attn_weight, out};
    if (int rc = msda::check_args(ptrs, 6, N, S, M, D, L, Lq, P)) return rc;
    msda::begin_call();
    if (N == 0 || Lq == 0) return MSDA_OK;
    if (S == 0) {
        const hipError_t e = hipMemsetAsync(out, 0, 2 * (size_t)N * Lq * M * D, (hipStream_t)stream);
        return e == hipSuccess ? MSDA_OK : msda::set_error(MSDA_ERR_LAUNCH, hipGetErrorString(e));
    }
    if (!msda::use_d32(N, S, M, D, L, Lq, P) || !msda::aligned_to(value, 8) || !msda::aligned_to(out, 8) ||
        !msda::aligned_to(sampling_loc, 8)) {
        if (table) if (int rc = msda::invalidate_forward_table(table, N, S, M, L, Lq, P, (hipStream_t)stream)) return rc;
    }
    if (!msda::use_d32(N, S, M, D, L, Lq, P) || !msda::aligned_to(value, 8) || !msda::aligned_to(out, 8) ||
        !msda::aligned_to(sampling_loc, 8))
        return msda::launch_fwd_generic<float>(value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S, M, D, L, Lq,
                                               P, out, (hipStream_t)stream);
    return msda::launch_fwd_d32_bf16(value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S, M, L, Lq, P,
                                     out, (hipStream_t)stream, table);
}

int msda_forward_bf16(const uint16_t *value, const int64_t *spatial_shapes, const int64_t *level_start,
                      const float *sampling_loc, const float *attn_weight, int N, int S, int M, int D, int L,
                      int Lq, int P, uint16_t *out, msda_stream_t stream)
{
    return forward_bf16_impl(value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S, M, D, L, Lq, P, out, stream, nullptr);
}

// ---- forward that leaves the point table for the backward of the same autograd node (include/msda.h) ----
unsigned long long msda_forward_workspace_bytes(int N, int S, int M, int D, int L, int Lq, int P, unsigned flags)
{
    if (N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0) return 0;
    if (msda::g_force_path == MSDA_PATH_GENERIC) return 0;
    return (unsigned long long)msda::forward_table_bytes(N, S, M, D, L, Lq, P, (flags & MSDA_FLAG_PROLOGUE) != 0);
}

static void *fwd_table(void *workspace, unsigned long long workspace_bytes, int N, int S, int M, int D, int L, int Lq, int P, bool prologue)
{
    if (!workspace || !msda::aligned_to(workspace, 16) || N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0) return nullptr;
    if (msda::g_force_path == MSDA_PATH_GENERIC) return nullptr;
    const size_t need = msda::forward_table_bytes(N, S, M, D, L, Lq, P, prologue);
    return (need > 0 && workspace_bytes >= need) ? workspace : nullptr;
}

int msda_forward_ws_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start, const float *sampling_loc,
                        const float *attn_weight, int N, int S, int M, int D, int L, int Lq, int P, float *out, void *workspace,
                        unsigned long long workspace_bytes, msda_stream_t stream)
{
    return msda::forward_impl<float>(value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S, M, D, L, Lq, P, out,
                                     (hipStream_t)stream, msda::use_d32(N, S, M, D, L, Lq, P),
                                     fwd_table(workspace, workspace_bytes, N, S, M, D, L, Lq, P, false));
}

int msda_forward_ws_bf16(const uint16_t *value, const int64_t *spatial_shapes, const int64_t *level_start, const float *sampling_loc,
                         const float *attn_weight, int N, int S, int M, int D, int L, int Lq, int P, uint16_t *out, void *workspace,
                         unsigned long long workspace_bytes, msda_stream_t stream)
{
    return forward_bf16_impl(value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S, M, D, L, Lq, P, out, stream,
                             fwd_table(workspace, workspace_bytes, N, S, M, D, L, Lq, P, false));
}

int msda_backward_bf16(const uint16_t *grad_out, const uint16_t *value, const int64_t *spatial_shapes,
                       const int64_t *level_start, const float *sampling_loc, const float *attn_weight, int N,
                       int S, int M, int D, int L, int Lq, int P, uint16_t *grad_value, float *grad_sampling_loc,
                       float *grad_attn_weight, msda_stream_t stream)
{
    return backward_bf16_impl<uint16_t>(grad_out, value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S, M, D,
                                        L, Lq, P, grad_value, grad_sampling_loc, grad_attn_weight, stream);
}

int msda_backward_bf16_gv32(const uint16_t *grad_out, const uint16_t *value, const int64_t *spatial_shapes,
                            const int64_t *level_start, const float *sampling_loc, const float *attn_weight, int N,
                            int S, int M, int D, int L, int Lq, int P, float *grad_value, float *grad_sampling_loc,
                            float *grad_attn_weight, msda_stream_t stream)
{
    return backward_bf16_impl<float>(grad_out, value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S, M, D, L,
                                     Lq, P, grad_value, grad_sampling_loc, grad_attn_weight, stream);
}

int msda_backward_passes(int Lq, int P) { return (Lq > 0 && P > 0) ? msda::backward_passes(Lq, P) : 0; }

unsigned long long msda_backward_workspace_bytes(int N, int S, int M, int D, int L, int Lq, int P, unsigned flags)
{
    if (N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0) return 0;
    if (msda::g_force_path == MSDA_PATH_GENERIC) return 0;
    // with MSDA_FLAG_FORWARD_TABLE: the forward's table (rounded up to 256 bytes) first, the call's scratch behind it
    const bool prologue = (flags & MSDA_FLAG_PROLOGUE) != 0;
    const size_t table = (flags & MSDA_FLAG_FORWARD_TABLE) ? msda::forward_table_bytes(N, S, M, D, L, Lq, P, prologue) : 0;
    const size_t scratch = msda::backward_workspace_bytes(N, S, M, D, L, Lq, P, flags);
    if (table == 0) return (unsigned long long)scratch;
    return (unsigned long long)(scratch ? msda::table_span(N, S, M, D, L, Lq, P, prologue) + scratch : table);
}

int msda_deterministic_supported(int elem_bytes, int N, int S, int M, int D, int L, int Lq, int P)
{
    if (N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0) return 1;              // (nothing to order)
    if (elem_bytes != 8 && msda::g_force_path != MSDA_PATH_GENERIC && msda::d32_supported(N, S, M, D, L, Lq, P)) return 1;
    return (double)N * (double)S * (double)M * (double)Lq * (double)P <= 68719476736.0 ? 1 : 0;      // msda_generic.hip: 2^36 point tests
}

int msda_backward_ws_f32(const float *grad_out, const float *value, const int64_t *spatial_shapes,
                         const int64_t *level_start, const float *sampling_loc, const float *attn_weight,
                         int N, int S, int M, int D, int L, int Lq, int P, float *grad_value,
                         float *grad_sampling_loc, float *grad_attn_weight, void *workspace,
                         unsigned long long workspace_bytes, unsigned flags, msda_stream_t stream)
{
    return msda::backward_impl<float>(grad_out, value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S, M, D,
                                      L, Lq, P, grad_value, grad_sampling_loc, grad_attn_weight, (hipStream_t)stream,
                                      msda::use_d32(N, S, M, D, L, Lq, P), workspace, (size_t)workspace_bytes, flags);
}

int msda_backward_ws_f64(const double *grad_out, const double *value, const int64_t *spatial_shapes,
                         const int64_t *level_start, const double *sampling_loc, const double *attn_weight,
                         int N, int S, int M, int D, int L, int Lq, int P, double *grad_value,
                         double *grad_sampling_loc, double *grad_attn_weight, void *workspace,
                         unsigned long long workspace_bytes, unsigned flags, msda_stream_t stream)
{
    return msda::backward_impl<double>(grad_out, value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S, M, D,
                                       L, Lq, P, grad_value, grad_sampling_loc, grad_attn_weight, (hipStream_t)stream, false,
                                       workspace, (size_t)workspace_bytes, flags);
}

int msda_backward_ws_bf16(const uint16_t *grad_out, const uint16_t *value, const int64_t *spatial_shapes,
                          const int64_t *level_start, const float *sampling_loc, const float *attn_weight, int N,
                          int S, int M, int D, int L, int Lq, int P, uint16_t *grad_value, float *grad_sampling_loc,
                          float *grad_attn_weight, void *workspace, unsigned long long workspace_bytes,
                          unsigned flags, msda_stream_t stream)
{
    return backward_bf16_impl<uint16_t>(grad_out, value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S, M, D,
                                        L, Lq, P, grad_value, grad_sampling_loc, grad_attn_weight, stream, workspace,
                                        (size_t)workspace_bytes, flags);
}

int msda_backward_ws_bf16_gv32(const uint16_t *grad_out, const uint16_t *value, const int64_t *spatial_shapes,
                               const int64_t *level_start, const float *sampling_loc, const float *attn_weight, int N,
                               int S, int M, int D, int L, int Lq, int P, float *grad_value, float *grad_sampling_loc,
                               float *grad_attn_weight, void *workspace, unsigned long long workspace_bytes,
                               unsigned flags, msda_stream_t stream)
{
    return backward_bf16_impl<float>(grad_out, value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S, M, D, L,
                                     Lq, P, grad_value, grad_sampling_loc, grad_attn_weight, stream, workspace,
                                     (size_t)workspace_bytes, flags);
}



int msda_prologue_supported(int N, int S, int M, int D, int L, int Lq, int P)
{
    if (N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0) return 0;
    if (msda::g_force_path == MSDA_PATH_GENERIC) return 0;
    return msda::prologue_supported(N, S, M, D, L, Lq, P) ? 1 : 0;
}

int msda_forward_prologue_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start,
                              const float *reference_points, const float *sampling_offsets, const float *attn_logits,
                              int N, int S, int M, int D, int L, int Lq, int P, long long ld_offsets, long long ld_logits,
                              float *out, float *sampling_loc_out, float *attn_weight_out, msda_stream_t stream)
{
    return msda_forward_prologue_ws_f32(value, spatial_shapes, level_start, reference_points, sampling_offsets, attn_logits, N, S, M,
                                        D, L, Lq, P, ld_offsets, ld_logits, out, sampling_loc_out, attn_weight_out, nullptr, 0, stream);
}

int msda_forward_prologue_ws_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start,
                                 const float *reference_points, const float *sampling_offsets, const float *attn_logits,
                                 int N, int S, int M, int D, int L, int Lq, int P, long long ld_offsets, long long ld_logits,
                                 float *out, float *sampling_loc_out, float *attn_weight_out, void *workspace,
                                 unsigned long long workspace_bytes, msda_stream_t stream)
{
    const void *ptrs[] = {value, spatial_shapes, level_start, reference_points, sampling_offsets, attn_logits, out,
                          sampling_loc_out, attn_weight_out};
    if (int rc = msda::check_args(ptrs, 9, N, S, M, D, L, Lq, P)) return rc;
    if (!msda_prologue_supported(N, S, M, D, L, Lq, P))
        return msda::set_error(MSDA_ERR_ARGUMENT, "msda_forward_prologue_f32: geometry not supported (msda_prologue_supported)");
    if (int rc = msda::check_row_strides("msda_forward_prologue_f32", M, L, P, sampling_offsets, &ld_offsets, &ld_logits)) return rc;
    if (!msda::aligned_to(value, 16) || !msda::aligned_to(out, 16) || !msda::aligned_to(reference_points, 8) ||
        !msda::aligned_to(sampling_loc_out, 8))
        return msda::refuse_unaligned("msda_forward_prologue_f32");
    msda::begin_call();
    return msda::launch_fwd_prologue(value, spatial_shapes, level_start, reference_points, sampling_offsets, attn_logits, N,
                                     S, M, L, Lq, P, ld_offsets, ld_logits, out, sampling_loc_out, attn_weight_out,
                                     (hipStream_t)stream, fwd_table(workspace, workspace_bytes, N, S, M, D, L, Lq, P, true));
}

int msda_backward_prologue_f32(const float *grad_out, const float *value, const int64_t *spatial_shapes,
                               const int64_t *level_start, const float *sampling_loc, const float *attn_weight, int N,
                               int S, int M, int D, int L, int Lq, int P, long long ld_grad_offsets,
                               long long ld_grad_logits, float *grad_value, float *grad_sampling_offsets,
                               float *grad_attn_logits, float *grad_reference_points, msda_stream_t stream)
{
    return msda_backward_prologue_ws_f32(grad_out, value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S, M, D,
                                         L, Lq, P, ld_grad_offsets, ld_grad_logits, grad_value, grad_sampling_offsets,
                                         grad_attn_logits, grad_reference_points, nullptr, 0, 0, stream);
}

int msda_backward_prologue_ws_f32(const float *grad_out, const float *value, const int64_t *spatial_shapes,
                                  const int64_t *level_start, const float *sampling_loc, const float *attn_weight, int N,
                                  int S, int M, int D, int L, int Lq, int P, long long ld_grad_offsets,
                                  long long ld_grad_logits, float *grad_value, float *grad_sampling_offsets,
                                  float *grad_attn_logits, float *grad_reference_points, void *workspace,
                                  unsigned long long workspace_bytes, unsigned flags, msda_stream_t stream)
{
    const void *ptrs[] = {grad_out, value, spatial_shapes, level_start, sampling_loc, attn_weight, grad_value,
                          grad_sampling_offsets, grad_attn_logits, grad_reference_points};
    if (int rc = msda::check_args(ptrs, 10, N, S, M, D, L, Lq, P)) return rc;
    if (!msda_prologue_supported(N, S, M, D, L, Lq, P))
        return msda::set_error(MSDA_ERR_ARGUMENT, "msda_backward_prologue_f32: geometry not supported (msda_prologue_supported)");
    if (int rc = msda::check_row_strides("msda_backward_prologue_f32", M, L, P, grad_sampling_offsets, &ld_grad_offsets,
                                         &ld_grad_logits)) return rc;
    if (!msda::aligned_to(grad_out, 16) || !msda::aligned_to(value, 16) || !msda::aligned_to(grad_value, 16) ||
        !msda::aligned_to(sampling_loc, 8) || !msda::aligned_to(grad_reference_points, 8))
        return msda::refuse_unaligned("msda_backward_prologue_f32");
    msda::begin_call();
    const msda::Scratch sc = msda::scratch_of(workspace, (size_t)workspace_bytes, flags, N, S, M, D, L, Lq, P, true);
    return msda::launch_bwd_prologue(grad_out, value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S, M, L,
                                     Lq, P, grad_value, ld_grad_offsets, ld_grad_logits, grad_sampling_offsets,
                                     grad_attn_logits, grad_reference_points, (hipStream_t)stream, sc.p,
                                     sc.bytes, (flags & MSDA_FLAG_DETERMINISTIC) != 0,
                                     msda::table_of(workspace, (size_t)workspace_bytes, flags, N, S, M, D, L, Lq, P, true),
                                  (flags & MSDA_FLAG_EXACT_NONFINITE) != 0);
}

int msda_forward_prologue_bf16(const uint16_t *value, const int64_t *spatial_shapes, const int64_t *level_start,
                               const float *reference_points, const float *sampling_offsets, const float *attn_logits,
                               int N, int S, int M, int D, int L, int Lq, int P, long long ld_offsets, long long ld_logits,
                               uint16_t *out, float *sampling_loc_out, float *attn_weight_out, msda_stream_t stream)
{
    return msda_forward_prologue_ws_bf16(value, spatial_shapes, level_start, reference_points, sampling_offsets, attn_logits, N, S, M,
                                         D, L, Lq, P, ld_offsets, ld_logits, out, sampling_loc_out, attn_weight_out, nullptr, 0, stream);
}

int msda_forward_prologue_ws_bf16(const uint16_t *value, const int64_t *spatial_shapes, const int64_t *level_start,
                                  const float *reference_points, const float *sampling_offsets, const float *attn_logits,
                                  int N, int S, int M, int D, int L, int Lq, int P, long long ld_offsets, long long ld_logits,
                                  uint16_t *out, float *sampling_loc_out, float *attn_weight_out, void *workspace,
                                  unsigned long long workspace_bytes, msda_stream_t stream)
{
    const void *ptrs[] = {value, spatial_shapes, level_start, reference_points, sampling_offsets, attn_logits, out,
                          sampling_loc_out, attn_weight_out};
    if (int rc = msda::check_args(ptrs, 9, N, S, M, D, L, Lq, P)) return rc;
    if (!msda_prologue_supported(N, S, M, D, L, Lq, P))
        return msda::set_error(MSDA_ERR_ARGUMENT, "msda_forward_prologue_bf16: geometry not supported (msda_prologue_supported)");
    if (int rc = msda::check_row_strides("msda_forward_prologue_bf16", M, L, P, sampling_offsets, &ld_offsets, &ld_logits)) return rc;
    if (!msda::aligned_to(value, 8) || !msda::aligned_to(out, 8) || !msda::aligned_to(reference_points, 8) ||
        !msda::aligned_to(sampling_loc_out, 8))
        return msda::refuse_unaligned("msda_forward_prologue_bf16");
    msda::begin_call();
    return msda::launch_fwd_prologue_bf16(value, spatial_shapes, level_start, reference_points, sampling_offsets, attn_logits,
                                          N, S, M, L, Lq, P, ld_offsets, ld_logits, out, sampling_loc_out, attn_weight_out,
                                          (hipStream_t)stream, fwd_table(workspace, workspace_bytes, N, S, M, D, L, Lq, P, true));
}

int msda_backward_prologue_bf16_gv32(const uint16_t *grad_out, const uint16_t *value, const int64_t *spatial_shapes,
                                     const int64_t *level_start, const float *sampling_loc, const float *attn_weight, int N,
                                     int S, int M, int D, int L, int Lq, int P, long long ld_grad_offsets,
                                     long long ld_grad_logits, float *grad_value, float *grad_sampling_offsets,
                                     float *grad_attn_logits, float *grad_reference_points, void *workspace,
                                     unsigned long long workspace_bytes, unsigned flags, msda_stream_t stream)
{
    const void *ptrs[] = {grad_out, value, spatial_shapes, level_start, sampling_loc, attn_weight, grad_value,
                          grad_sampling_offsets, grad_attn_logits, grad_reference_points};
    if (int rc = msda::check_args(ptrs, 10, N, S, M, D, L, Lq, P)) return rc;
    if (!msda_prologue_supported(N, S, M, D, L, Lq, P))
        return msda::set_error(MSDA_ERR_ARGUMENT, "msda_backward_prologue_bf16_gv32: geometry not supported (msda_prologue_supported)");
    if (int rc = msda::check_row_strides("msda_backward_prologue_bf16_gv32", M, L, P, grad_sampling_offsets, &ld_grad_offsets,
                                         &ld_grad_logits)) return rc;
    if (!msda::aligned_to(grad_out, 8) || !msda::aligned_to(value, 8) || !msda::aligned_to(grad_value, 16) ||
        !msda::aligned_to(sampling_loc, 8) || !msda::aligned_to(grad_reference_points, 8))
        return msda::refuse_unaligned("msda_backward_prologue_bf16_gv32");
    msda::begin_call();
    const msda::Scratch sc = msda::scratch_of(workspace, (size_t)workspace_bytes, flags, N, S, M, D, L, Lq, P, true);
    return msda::launch_bwd_prologue_bf16(grad_out, value, spatial_shapes, level_start, sampling_loc, attn_weight, N, S, M, L,
                                          Lq, P, grad_value, ld_grad_offsets, ld_grad_logits, grad_sampling_offsets,
                                          grad_attn_logits, grad_reference_points, (hipStream_t)stream, sc.p,
                                          sc.bytes, (flags & MSDA_FLAG_DETERMINISTIC) != 0,
                                          msda::table_of(workspace, (size_t)workspace_bytes, flags, N, S, M, D, L, Lq, P, true),
                                  (flags & MSDA_FLAG_EXACT_NONFINITE) != 0);
}

unsigned long long msda_linear_wgrad_workspace_bytes(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return (unsigned long long)msda::linear_wgrad_workspace_bytes(M, N, K);
}

int msda_linear_wgrad_f32(const float *grad_out, const float *input, int M, int N, int K, float *grad_weight,
                          float *grad_bias, void *workspace, msda_stream_t stream)
{
    return msda_linear_wgrad_masked_f32(grad_out, input, nullptr, M, N, K, grad_weight, grad_bias, workspace, stream);
}

int msda_linear_wgrad_masked_f32(const float *grad_out, const float *input, const uint8_t *row_mask, int M, int N, int K,
                                 float *grad_weight, float *grad_bias, void *workspace, msda_stream_t stream)
{
    if (M < 0 || N <= 0 || K <= 0 || (N & 3) || (K & 3))
        return msda::set_error(MSDA_ERR_ARGUMENT, "msda_linear_wgrad_f32: need N, K > 0 and multiples of 4");
    if (grad_weight == nullptr || (M > 0 && (grad_out == nullptr || input == nullptr)))
        return msda::set_error(MSDA_ERR_ARGUMENT, "msda_linear_wgrad_f32: null device pointer");
    if (!msda::aligned_to(grad_out, 16) || !msda::aligned_to(input, 16) || !msda::aligned_to(workspace, 16))
        return msda::set_error(MSDA_ERR_ARGUMENT, "msda_linear_wgrad_f32: grad_out, input and workspace must be 16-byte aligned");
    msda::begin_call();
    if (M == 0) {
        hipError_t e = hipMemsetAsync(grad_weight, 0, sizeof(float) * (size_t)N * K, (hipStream_t)stream);
        if (e == hipSuccess && grad_bias) e = hipMemsetAsync(grad_bias, 0, sizeof(float) * (size_t)N, (hipStream_t)stream);
        return e == hipSuccess ? MSDA_OK : msda::set_error(MSDA_ERR_LAUNCH, hipGetErrorString(e));
    }
    return msda::launch_linear_wgrad(grad_out, input, row_mask, M, N, K, grad_weight, grad_bias,
                                     static_cast<float *>(workspace), (hipStream_t)stream);
}

int msda_linear_wgrad_multi(int count, const void *const *grad_out, const void *const *input, const int *operands_bf16,
                            const uint8_t *const *row_mask, const int *M, const int *N, const int *K, float *const *grad_weight,
                            float *const *grad_bias, void *const *workspace, msda_stream_t stream)
{
    if (count < 1 || count > 4 || !grad_out || !input || !M || !N || !K || !grad_weight || !workspace)
        return msda::set_error(MSDA_ERR_ARGUMENT, "msda_linear_wgrad_multi: 1..4 problems, non-null arrays");
    float *ws[4];
    for (int p = 0; p < count; ++p) {
        if (M[p] <= 0 || N[p] <= 0 || K[p] <= 0 || (N[p] & 3) || (K[p] & 3))
            return msda::set_error(MSDA_ERR_ARGUMENT, "msda_linear_wgrad_multi: need M, N, K > 0 and N, K multiples of 4");
        if (grad_weight[p] == nullptr || grad_out[p] == nullptr || input[p] == nullptr)
            return msda::set_error(MSDA_ERR_ARGUMENT, "msda_linear_wgrad_multi: null device pointer");
        // (bf16 operands: 8-byte alignment is what msda_linear_wgrad_masked_bf16 asks for)
        const size_t al = operands_bf16 && operands_bf16[p] ? 8 : 16;
        if (!msda::aligned_to(grad_out[p], al) || !msda::aligned_to(input[p], al) || !msda::aligned_to(workspace[p], 16))
            return msda::set_error(MSDA_ERR_ARGUMENT, "msda_linear_wgrad_multi: grad_out, input (16 bytes; bf16: 8) and workspace (16) alignment");
        if (msda::linear_wgrad_workspace_bytes(M[p], N[p], K[p]) > 0 && workspace[p] == nullptr)
            return msda::set_error(MSDA_ERR_ARGUMENT, "msda_linear_wgrad_multi: workspace required");
        ws[p] = static_cast<float *>(workspace[p]);
    }
    msda::begin_call();
    return msda::launch_linear_wgrad_multi(count, grad_out, input, operands_bf16, row_mask, M, N, K, grad_weight, grad_bias, ws,
                                           (hipStream_t)stream);
}

int msda_linear_wgrad_multi_f32(int count, const float *const *grad_out, const float *const *input, const uint8_t *const *row_mask,
                                const int *M, const int *N, const int *K, float *const *grad_weight, float *const *grad_bias,
                                void *const *workspace, msda_stream_t stream)
{
    return msda_linear_wgrad_multi(count, reinterpret_cast<const void *const *>(grad_out), reinterpret_cast<const void *const *>(input),
                                   nullptr, row_mask, M, N, K, grad_weight, grad_bias, workspace, stream);
}

int msda_linear_wgrad_masked_bf16(const uint16_t *grad_out, const uint16_t *input, const uint8_t *row_mask, int M, int N, int K,
                                  float *grad_weight, float *grad_bias, void *workspace, msda_stream_t stream)
{
    if (M < 0 || N <= 0 || K <= 0 || (N & 3) || (K & 3))
        return msda::set_error(MSDA_ERR_ARGUMENT, "msda_linear_wgrad_masked_bf16: need N, K > 0 and multiples of 4");
    if (grad_weight == nullptr || (M > 0 && (grad_out == nullptr || input == nullptr)))
        return msda::set_error(MSDA_ERR_ARGUMENT, "msda_linear_wgrad_masked_bf16: null device pointer");
    if (!msda::aligned_to(grad_out, 8) || !msda::aligned_to(input, 8) || !msda::aligned_to(workspace, 16))
        return msda::set_error(MSDA_ERR_ARGUMENT, "msda_linear_wgrad_masked_bf16: grad_out and input must be 8-byte, workspace "
                                                  "16-byte aligned");
    msda::begin_call();
    if (M == 0) {
        hipError_t e = hipMemsetAsync(grad_weight, 0, sizeof(float) * (size_t)N * K, (hipStream_t)stream);
        if (e == hipSuccess && grad_bias) e = hipMemsetAsync(grad_bias, 0, sizeof(float) * (size_t)N, (hipStream_t)stream);
        return e == hipSuccess ? MSDA_OK : msda::set_error(MSDA_ERR_LAUNCH, hipGetErrorString(e));
    }
    return msda::launch_linear_wgrad_bf16(grad_out, input, row_mask, M, N, K, grad_weight, grad_bias,
                                          static_cast<float *>(workspace), (hipStream_t)stream);
}

int msda_zero_masked_rows_f32(float *x, const uint8_t *row_mask, long long rows, int cols, msda_stream_t stream)
{
    if (rows < 0 || cols <= 0 || (cols & 3) || ((uintptr_t)x & 15))
        return msda::set_error(MSDA_ERR_ARGUMENT, "msda_zero_masked_rows_f32: need rows >= 0, cols > 0 and a multiple of 4, "
                                                  "16-byte aligned rows");
    if (rows > 0 && (x == nullptr || row_mask == nullptr))
        return msda::set_error(MSDA_ERR_ARGUMENT, "msda_zero_masked_rows_f32: null device pointer");
    if (rows > (1LL << 26) - 4) return msda::set_error(MSDA_ERR_ARGUMENT, "msda_zero_masked_rows_f32: too many rows");   // < 2^32 threads
    msda::begin_call();
    return msda::launch_zero_masked_rows(x, row_mask, rows, cols, (hipStream_t)stream);
}

static int check_linear_rows(const char *who, const void *a, const void *w, const void *bias, const void *out, long long rows,
                             int out_features, int in_features)
{
    if (rows < 0 || out_features <= 0 || in_features <= 0 || (out_features & 3) || (in_features & 3))
        return msda::set_error(MSDA_ERR_ARGUMENT, (std::string(who) + ": need rows >= 0 and out_features, in_features > 0 and "
                                                                       "multiples of 4").c_str());
    if (w == nullptr || (rows > 0 && (a == nullptr || out == nullptr)))
        return msda::set_error(MSDA_ERR_ARGUMENT, (std::string(who) + ": null device pointer").c_str());
    if (((uintptr_t)a | (uintptr_t)w | (uintptr_t)out | (uintptr_t)bias) & 15)
        return msda::set_error(MSDA_ERR_ARGUMENT, (std::string(who) + ": operands must be 16-byte aligned").c_str());
    return MSDA_OK;
}

int msda_linear_forward_f32(const float *input, const float *weight, const float *bias, const uint8_t *row_mask, long long rows,
                            int out_features, int in_features, float *output, msda_stream_t stream)
{
    if (int rc = check_linear_rows("msda_linear_forward_f32", input, weight, bias, output, rows, out_features, in_features)) return rc;
    msda::begin_call();
    return msda::launch_linear_forward(input, weight, bias, row_mask, rows, out_features, in_features, output, (hipStream_t)stream);
}

int msda_linear_dgrad_f32(const float *grad_out, const float *weight, const uint8_t *row_mask, long long rows, int out_features,
                          int in_features, float *grad_input, msda_stream_t stream)
{
    if (int rc = check_linear_rows("msda_linear_dgrad_f32", grad_out, weight, nullptr, grad_input, rows, out_features, in_features))
        return rc;
    msda::begin_call();
    return msda::launch_linear_dgrad(grad_out, weight, row_mask, rows, out_features, in_features, grad_input, (hipStream_t)stream);
}

static int check_layernorm_args(const char *who, long long rows, int d, const void *const *ptrs, int n)
{
    char buf[200];
    if (rows < 0 || d <= 0 || (d & 3) || d > 1024) {
        std::snprintf(buf, sizeof(buf), "%s: need rows >= 0 and a row width that is a multiple of 4 in [4, 1024]", who);
        return msda::set_error(MSDA_ERR_ARGUMENT, buf);
    }
    for (int i = 0; i < n; ++i)
        if (rows > 0 && (ptrs[i] == nullptr || ((uintptr_t)ptrs[i] & 15))) {
            std::snprintf(buf, sizeof(buf), "%s: null or not 16-byte aligned device pointer", who);
            return msda::set_error(MSDA_ERR_ARGUMENT, buf);
        }
    return MSDA_OK;
}

unsigned long long msda_add_layernorm_workspace_bytes(long long rows, int d)
{
    return (rows > 0 && d > 0) ? (unsigned long long)msda::add_layernorm_workspace_bytes(rows, d) : 0;
}

int msda_add_layernorm_forward_f32(const float *x, const float *residual, const float *gamma, const float *beta, long long rows,
                                   int d, float eps, float *y, float *mean, float *rstd, msda_stream_t stream)
{
    const void *ptrs[] = {x, gamma, beta, y};
    if (int rc = check_layernorm_args("msda_add_layernorm_forward_f32", rows, d, ptrs, 4)) return rc;
    if (rows > 0 && (mean == nullptr || rstd == nullptr || (residual && ((uintptr_t)residual & 15))))
        return msda::set_error(MSDA_ERR_ARGUMENT, "msda_add_layernorm_forward_f32: null mean / rstd or misaligned residual");
    msda::begin_call();
    if (rows == 0) return MSDA_OK;
    return msda::launch_add_layernorm_fwd(x, residual, gamma, beta, rows, d, eps, y, mean, rstd, (hipStream_t)stream);
}

int msda_add_layernorm_backward_f32(const float *grad_y, const float *x, const float *residual, const float *gamma,
                                    const float *mean, const float *rstd, long long rows, int d, float *grad_sum,
                                    float *grad_gamma, float *grad_beta, void *workspace, msda_stream_t stream)
{
    const void *ptrs[] = {grad_y, x, gamma, grad_sum, workspace};
    if (int rc = check_layernorm_args("msda_add_layernorm_backward_f32", rows, d, ptrs, 5)) return rc;
    if (grad_gamma == nullptr || grad_beta == nullptr || (rows > 0 && (mean == nullptr || rstd == nullptr)) ||
        (residual && ((uintptr_t)residual & 15)))
        return msda::set_error(MSDA_ERR_ARGUMENT, "msda_add_layernorm_backward_f32: null pointer or misaligned residual");
    msda::begin_call();
    if (rows == 0) {
        hipError_t e = hipMemsetAsync(grad_gamma, 0, sizeof(float) * (size_t)d, (hipStream_t)stream);
        if (e == hipSuccess) e = hipMemsetAsync(grad_beta, 0, sizeof(float) * (size_t)d, (hipStream_t)stream);
        return e == hipSuccess ? MSDA_OK : msda::set_error(MSDA_ERR_LAUNCH, hipGetErrorString(e));
    }
    return msda::launch_add_layernorm_bwd(grad_y, x, residual, gamma, mean, rstd, rows, d, grad_sum, grad_gamma, grad_beta,
                                          static_cast<float *>(workspace), (hipStream_t)stream);
}

int msda_cast_bf16_multi_f32(int count, const float *const *src, uint16_t *const *dst, const long long *n, msda_stream_t stream)
{
    if (count < 1 || count > 4 || src == nullptr || dst == nullptr || n == nullptr)
        return msda::set_error(MSDA_ERR_ARGUMENT, "msda_cast_bf16_multi_f32: 1..4 segments");
    for (int k = 0; k < count; ++k)
        if (n[k] < 0 || (n[k] & 1) || (n[k] > 0 && (src[k] == nullptr || dst[k] == nullptr)) || ((uintptr_t)src[k] & 7) || ((uintptr_t)dst[k] & 3))
            return msda::set_error(MSDA_ERR_ARGUMENT, "msda_cast_bf16_multi_f32: element counts must be even, sources 8-byte and destinations 4-byte aligned");
    msda::begin_call();
    return msda::launch_cast_bf16_multi(count, src, dst, n, (hipStream_t)stream);
}

int msda_relu_dropout_backward_f32(float *grad, const float *act, float scale, long long n, msda_stream_t stream)
{
    if (n < 0 || (n & 3) || (n > 0 && (grad == nullptr || act == nullptr)) || (((uintptr_t)grad | (uintptr_t)act) & 15))
        return msda::set_error(MSDA_ERR_ARGUMENT, "msda_relu_dropout_backward_f32: n must be a multiple of 4 and both tensors 16-byte aligned");
    msda::begin_call();
    if (n == 0) return MSDA_OK;
    return msda::launch_relu_dropout_bwd(grad, act, scale, n, (hipStream_t)stream);
}

static int flatten_impl(const char *who, int L, float *const *src_levels, float *const *pos_levels, const float *level_embed,
                        const int *heights, const int *widths, int N, int C, float *src_flatten, float *pos_flatten, bool unflatten,
                        msda_stream_t stream, float *grad_level_embed = nullptr, void *workspace = nullptr,
                        unsigned long long *workspace_bytes_out = nullptr)
{
    char buf[200];
    if (L <= 0 || L > msda::kMaxLevels || N < 0 || C <= 0 || (C & 3) || heights == nullptr || widths == nullptr) {
        std::snprintf(buf, sizeof(buf), "%s: need 1 <= L <= %d, N >= 0, C > 0 and a multiple of 4, host arrays of H and W", who,
                      msda::kMaxLevels);
        return msda::set_error(MSDA_ERR_ARGUMENT, buf);
    }
    msda::FlattenPlan plan;
    plan.L = L;
    long long S = 0;
    for (int l = 0; l < L; ++l) {
        if (heights[l] <= 0 || widths[l] <= 0 || (long long)heights[l] * widths[l] > 0x3fffffffLL)
            return msda::set_error(MSDA_ERR_ARGUMENT, "msda flatten: level sizes must be positive");
        plan.src[l] = src_levels ? src_levels[l] : nullptr;
        plan.pos[l] = pos_levels ? pos_levels[l] : nullptr;
        if (N > 0 && ((src_flatten && plan.src[l] == nullptr) || (pos_flatten && plan.pos[l] == nullptr)))
            return msda::set_error(MSDA_ERR_ARGUMENT, "msda flatten: null level pointer");
        plan.hw[l] = heights[l] * widths[l];
        plan.start[l] = (int)S;
        plan.first_block[l] = 0;
        S += plan.hw[l];
    }
    if (S > 0x3fffffffLL) return msda::set_error(MSDA_ERR_ARGUMENT, "msda flatten: too many pixels");
    if (((uintptr_t)src_flatten & 15) || ((uintptr_t)pos_flatten & 15) || ((uintptr_t)level_embed & 15))
        return msda::set_error(MSDA_ERR_ARGUMENT, "msda flatten: flattened tensors and level_embed must be 16-byte aligned");
    if (workspace_bytes_out) { *workspace_bytes_out = N > 0 ? msda::unflatten_workspace_bytes(plan, N, C) : 0; return MSDA_OK; }
    if (((uintptr_t)grad_level_embed & 15) || ((uintptr_t)workspace & 15))
        return msda::set_error(MSDA_ERR_ARGUMENT, "msda unflatten: grad_level_embed and workspace must be 16-byte aligned");
    msda::begin_call();
    if (N == 0) {
        if (grad_level_embed) {
            const hipError_t e = hipMemsetAsync(grad_level_embed, 0, sizeof(float) * (size_t)L * C, (hipStream_t)stream);
            if (e != hipSuccess) return msda::set_error(MSDA_ERR_LAUNCH, hipGetErrorString(e));
        }
        return MSDA_OK;
    }
    return msda::launch_flatten_levels(plan, N, C, (int)S, level_embed, src_flatten, pos_flatten, unflatten, (hipStream_t)stream,
                                       grad_level_embed, static_cast<float *>(workspace));
}

int msda_flatten_levels_f32(int L, const float *const *src_levels, const float *const *pos_levels, const float *level_embed,
                            const int *heights, const int *widths, int N, int C, float *src_flatten, float *pos_flatten,
                            msda_stream_t stream)
{
    return flatten_impl("msda_flatten_levels_f32", L, const_cast<float *const *>(src_levels), const_cast<float *const *>(pos_levels),
                        level_embed, heights, widths, N, C, src_flatten, pos_flatten, false, stream);
}

int msda_unflatten_levels_f32(int L, float *const *grad_src_levels, float *const *grad_pos_levels, const int *heights,
                              const int *widths, int N, int C, const float *grad_src_flatten, const float *grad_pos_flatten,
                              float *grad_level_embed, void *workspace, msda_stream_t stream)
{
    return flatten_impl("msda_unflatten_levels_f32", L, grad_src_levels, grad_pos_levels, nullptr, heights, widths, N, C,
                        const_cast<float *>(grad_src_flatten), const_cast<float *>(grad_pos_flatten), true, stream,
                        grad_level_embed, workspace);
}

unsigned long long msda_unflatten_workspace_bytes(int L, const int *heights, const int *widths, int N, int C)
{
    unsigned long long n = 0;
    float *dummy[msda::kMaxLevels] = {};
    if (flatten_impl("msda_unflatten_workspace_bytes", L, dummy, dummy, nullptr, heights, widths, N, C, nullptr, nullptr, true, nullptr,
                     nullptr, nullptr, &n) != MSDA_OK)
        return 0;
    return n;
}

const char *msda_last_error(void) { return msda::g_err; }

int msda_version(void) { return MSDA_ABI_VERSION; }

int msda_path_for(int elem_bytes, int M, int D, int L, int P)
{
    // N, S, Lq only matter through the int32-offset limits; probe with small ones.
    return ((elem_bytes == 4 || elem_bytes == 2) && msda::use_d32(1, 1, M, D, L, 1, P)) ? MSDA_PATH_D32 : MSDA_PATH_GENERIC;
}

void msda_force_path(int path) { msda::g_force_path = path; }

unsigned long long msda_launch_count(void) { return msda::launch_checks(); }

int msda_describe_plan(int row_bytes, int grad_value_bytes, int N, int S, int M, int D, int L, int Lq, int P, unsigned flags,
                       int has_workspace, char *buf, int buf_len)
{
    if (!buf || buf_len <= 0) return 0;
    buf[0] = 0;
    if (N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0) return 0;
    if ((row_bytes != 4 && row_bytes != 2) || msda::g_force_path == MSDA_PATH_GENERIC || !msda::use_d32(N, S, M, D, L, Lq, P)) {
        const int k = snprintf(buf, (size_t)buf_len, "generic");
        return k < buf_len ? k : buf_len - 1;
    }
    const bool prologue = (flags & MSDA_FLAG_PROLOGUE) != 0;
    if (prologue && !msda::prologue_supported(N, S, M, D, L, Lq, P)) {
        const int k = snprintf(buf, (size_t)buf_len, "prologue unsupported");
        return k < buf_len ? k : buf_len - 1;
    }
    const int k = msda::describe_plan(row_bytes, row_bytes == 4 ? 4 : grad_value_bytes, N, S, M, L, Lq, P, prologue, has_workspace != 0,
                                      (flags & MSDA_FLAG_DETERMINISTIC) != 0, buf, buf_len, (flags & MSDA_FLAG_EXACT_NONFINITE) != 0);
    return k < buf_len ? k : buf_len - 1;
}

}  // extern "C"
