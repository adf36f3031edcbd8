// _msda_torch — the thin torch extension over the C ABI (include/msda.h) that SURVEY.md §7 step 2 / §8b allow next to
// the ctypes binding: the same two entry points as the reference's pybind module (UVHand models/ops/src/vision.cpp:13-16,
// host checks of models/ops/src/cuda/ms_deform_attn_cuda.cu:28-52, 93-117) plus the autograd Function of
// models/ops/functions/ms_deform_attn_func.py:21-39 as a C++ node, so that an eager forward+backward costs PyTorch's
// engine and two kernel launches instead of Python-side marshalling (bench.py `eager_ms_per_step`).
//
// No kernels here and no torch types beyond this file: every call ends in msda_forward_* / msda_backward_ws_* of
// libmsda_hip.so with raw device pointers, sizes and the current HIP stream.
#include <atomic>
#include <torch/extension.h>

// PyTorch-ROCm presents its HIP devices as DeviceType::CUDA: the guard and the stream are the "masquerading" ones
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include "msda.h"

namespace {

struct Dims { int N, S, M, D, L, Lq, P; };

// Grad mode as the entry point saw it (inside Function::forward it is always off): a forward that will get a backward also
// leaves its per-point table for it (msda_forward_ws_*); under no_grad / inference_mode nothing extra is written.
thread_local bool g_grad_mode_at_entry = false;

void check_inputs(std::initializer_list<std::pair<const char *, const at::Tensor *>> named)
{
    const at::Tensor &value = *named.begin()->second;
    TORCH_CHECK(value.is_cuda(), "Not implemented on the CPU");                          // ms_deform_attn.h:38,60
    for (auto &nt : named) TORCH_CHECK(nt.second->is_contiguous(), nt.first, " tensor has to be contiguous");
    for (auto &nt : named) TORCH_CHECK(nt.second->is_cuda(), nt.first, " must be a CUDA tensor");
    for (auto &nt : named)
        TORCH_CHECK(nt.second->device() == value.device(), nt.first, " must be on the same device as value (", nt.second->device(),
                    " vs ", value.device(), ")");
}

Dims dims(const at::Tensor &value, const at::Tensor &shapes, const at::Tensor &lsi, const at::Tensor &loc, const at::Tensor &attn,
          int64_t im2col_step)
{
    TORCH_CHECK(value.dim() == 4 && loc.dim() == 6 && attn.dim() == 5,
                "ms_deform_attn: expected value[N,S,M,D], sampling_loc[N,Lq,M,L,P,2], attn_weight[N,Lq,M,L,P]");
    Dims d;
    d.N = (int)value.size(0); d.S = (int)value.size(1); d.M = (int)value.size(2); d.D = (int)value.size(3);
    d.L = (int)shapes.size(0); d.Lq = (int)loc.size(1); d.P = (int)loc.size(4);
    TORCH_CHECK(loc.sizes() == at::IntArrayRef({d.N, d.Lq, d.M, d.L, d.P, 2}) && attn.sizes() == at::IntArrayRef({d.N, d.Lq, d.M, d.L, d.P}),
                "ms_deform_attn: sampling_loc ", loc.sizes(), " / attn_weight ", attn.sizes(), " do not match value ", value.sizes(),
                " and ", d.L, " levels");
    TORCH_CHECK(shapes.scalar_type() == at::kLong && lsi.scalar_type() == at::kLong,
                "expected scalar type Long for spatial_shapes / level_start_index");
    TORCH_CHECK(shapes.dim() == 2 && shapes.size(1) == 2 && lsi.dim() == 1 && lsi.size(0) == d.L,
                "ms_deform_attn: spatial_shapes must be [L,2] and level_start_index [L]");
    const int64_t step = std::min<int64_t>(d.N, im2col_step);                            // ms_deform_attn_cuda.cu:50-52
    TORCH_CHECK(d.N == 0 || (step > 0 && d.N % step == 0), "batch(", d.N, ") must divide im2col_step(", step, ")");
    TORCH_CHECK(loc.scalar_type() == attn.scalar_type(), "expected sampling_loc and attn_weight to have the same dtype");
    TORCH_CHECK(value.scalar_type() == loc.scalar_type() && (value.scalar_type() == at::kFloat || value.scalar_type() == at::kDouble),
                "_msda_torch serves float32 / float64 tensors of one dtype (other cases go through uvhand_amd._native)");
    return d;
}

void raise_if(int rc, const char *what)
{
    TORCH_CHECK(rc == 0, what, " failed (code ", rc, "): ", msda_last_error());
}


// models/ops/functions/ms_deform_attn_func.py:31 marks the backward @once_differentiable.  The nodes below end in raw-pointer
// kernels, so a second differentiation through them would return gradients that silently ignore the graph: refuse it where
// the reference's decorator raises (a grad_output that itself carries a graph while grad mode is on, i.e. create_graph=True).
void once_differentiable(const torch::autograd::variable_list &grads, const char *what)
{
    if (!at::GradMode::is_enabled()) return;
    for (const auto &g : grads)
        TORCH_CHECK(!(g.defined() && g.requires_grad()), what,
                    ": trying to differentiate twice a function that was marked with @once_differentiable");
}

// The buffer a forward of this geometry fills with its point table for the backward of the same node (msda_forward_ws_*,
// include/msda.h); undefined where the backward's plan reads none.
// MSDA_FLAG_EXACT_NONFINITE for every backward queued from here (uvhand_amd.set_exact_nonfinite; include/msda.h)
static std::atomic<unsigned> g_extra_flags{0};

at::Tensor forward_table(const at::Tensor &like, const Dims &d, unsigned flags)
{
    unsigned long long n = msda_forward_workspace_bytes(d.N, d.S, d.M, d.D, d.L, d.Lq, d.P, flags);
    if (!n) return at::Tensor();
    // the same buffer goes to the backward with MSDA_FLAG_FORWARD_TABLE: the table first, that call's scratch behind it
    n = std::max(n, msda_backward_workspace_bytes(d.N, d.S, d.M, d.D, d.L, d.Lq, d.P, flags | MSDA_FLAG_FORWARD_TABLE));
    return at::empty({(int64_t)n}, like.options().dtype(at::kByte));
}

at::Tensor forward_t(const at::Tensor &value, const at::Tensor &shapes, const at::Tensor &lsi, const at::Tensor &loc,
                     const at::Tensor &attn, int64_t im2col_step, at::Tensor *table)
{
    check_inputs({{"value", &value}, {"spatial_shapes", &shapes}, {"level_start_index", &lsi}, {"sampling_loc", &loc},
                  {"attn_weight", &attn}});
    const Dims d = dims(value, shapes, lsi, loc, attn, im2col_step);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(value.device());
    auto out = at::empty({d.N, d.Lq, (int64_t)d.M * d.D}, value.options());
    auto stream = (msda_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(value.device().index()).stream();
    int rc;
    if (value.scalar_type() == at::kFloat) {
        if (table) *table = forward_table(value, d, 0);
        const bool t = table && table->defined();
        rc = msda_forward_ws_f32(value.data_ptr<float>(), shapes.data_ptr<int64_t>(), lsi.data_ptr<int64_t>(), loc.data_ptr<float>(),
                                 attn.data_ptr<float>(), d.N, d.S, d.M, d.D, d.L, d.Lq, d.P, out.data_ptr<float>(),
                                 t ? table->data_ptr() : nullptr, t ? (unsigned long long)table->numel() : 0, stream);
    } else
        rc = msda_forward_f64(value.data_ptr<double>(), shapes.data_ptr<int64_t>(), lsi.data_ptr<int64_t>(), loc.data_ptr<double>(),
                              attn.data_ptr<double>(), d.N, d.S, d.M, d.D, d.L, d.Lq, d.P, out.data_ptr<double>(), stream);
    raise_if(rc, "ms_deform_attn_forward");
    return out;
}

at::Tensor forward(const at::Tensor &value, const at::Tensor &shapes, const at::Tensor &lsi, const at::Tensor &loc,
                   const at::Tensor &attn, int64_t im2col_step)
{
    return forward_t(value, shapes, lsi, loc, attn, im2col_step, nullptr);
}

std::vector<at::Tensor> backward_t(const at::Tensor &value, const at::Tensor &shapes, const at::Tensor &lsi, const at::Tensor &loc,
                                   const at::Tensor &attn, const at::Tensor &grad_out_in, int64_t im2col_step, bool deterministic,
                                   const at::Tensor &table)
{
    const at::Tensor grad_out = grad_out_in.contiguous();       // the reference asserts it (ms_deform_attn_cuda.cu:98)
    check_inputs({{"value", &value}, {"spatial_shapes", &shapes}, {"level_start_index", &lsi}, {"sampling_loc", &loc},
                  {"attn_weight", &attn}, {"grad_output", &grad_out}});
    const Dims d = dims(value, shapes, lsi, loc, attn, im2col_step);
    TORCH_CHECK(grad_out.scalar_type() == value.scalar_type() && grad_out.numel() == (int64_t)d.N * d.Lq * d.M * d.D,
                "ms_deform_attn_backward: grad_output must be ", value.scalar_type(), "[", d.N, ",", d.Lq, ",", d.M * d.D, "]");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(value.device());
    auto gv = at::empty_like(value), gl = at::empty_like(loc), ga = at::empty_like(attn);
    auto stream = (msda_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(value.device().index()).stream();
    int rc;
    // no deterministic kernel for this geometry within the library's work bound + warn_only: warn and run the default kernels,
    // as torch does for its own ops without a deterministic form (without warn_only the library's refusal is raised)
    if (deterministic && at::globalContext().deterministicAlgorithmsWarnOnly() &&
        !msda_deterministic_supported(value.scalar_type() == at::kDouble ? 8 : 4, d.N, d.S, d.M, d.D, d.L, d.Lq, d.P)) {
        TORCH_WARN_ONCE("ms_deform_attn_backward: no deterministic grad_value kernel for this geometry within the work bound "
                        "(include/msda.h, MSDA_FLAG_DETERMINISTIC); running the default (atomic) kernel [warn_only]");
        deterministic = false;
    }
    unsigned flags = (deterministic ? MSDA_FLAG_DETERMINISTIC : 0u) | g_extra_flags.load();
    at::Tensor ws;
    // (also without flags: the library says how much scratch a call of this geometry can use, mostly none)
    unsigned long long nbytes = 0;
    if (table.defined()) { ws = table; nbytes = (unsigned long long)table.numel(); flags |= MSDA_FLAG_FORWARD_TABLE; }   // the forward's table, scratch behind it
    else if ((nbytes = msda_backward_workspace_bytes(d.N, d.S, d.M, d.D, d.L, d.Lq, d.P, flags)) != 0)
        ws = at::empty({(int64_t)nbytes}, value.options().dtype(at::kByte));
    if (value.scalar_type() == at::kFloat)
        rc = msda_backward_ws_f32(grad_out.data_ptr<float>(), value.data_ptr<float>(), shapes.data_ptr<int64_t>(),
                                  lsi.data_ptr<int64_t>(), loc.data_ptr<float>(), attn.data_ptr<float>(), d.N, d.S, d.M, d.D, d.L,
                                  d.Lq, d.P, gv.data_ptr<float>(), gl.data_ptr<float>(), ga.data_ptr<float>(),
                                  nbytes ? ws.data_ptr() : nullptr, nbytes, flags, stream);
    else
        rc = msda_backward_ws_f64(grad_out.data_ptr<double>(), value.data_ptr<double>(), shapes.data_ptr<int64_t>(),
                                  lsi.data_ptr<int64_t>(), loc.data_ptr<double>(), attn.data_ptr<double>(), d.N, d.S, d.M, d.D, d.L,
                                  d.Lq, d.P, gv.data_ptr<double>(), gl.data_ptr<double>(), ga.data_ptr<double>(),
                                  nbytes ? ws.data_ptr() : nullptr, nbytes, flags, stream);
    raise_if(rc, "ms_deform_attn_backward");
    return {gv, gl, ga};
}

std::vector<at::Tensor> backward(const at::Tensor &value, const at::Tensor &shapes, const at::Tensor &lsi, const at::Tensor &loc,
                                 const at::Tensor &attn, const at::Tensor &grad_out_in, int64_t im2col_step, bool deterministic)
{
    return backward_t(value, shapes, lsi, loc, attn, grad_out_in, im2col_step, deterministic, at::Tensor());
}

// models/ops/functions/ms_deform_attn_func.py:21-39 as a C++ autograd node (value cast to the compute dtype in both
// directions, the un-cast inputs saved, gradients for value / sampling_locations / attention_weights only).
class MSDAFunction : public torch::autograd::Function<MSDAFunction> {
public:
    static at::Tensor forward(torch::autograd::AutogradContext *ctx, const at::Tensor &value, const at::Tensor &shapes,
                              const at::Tensor &lsi, const at::Tensor &loc, const at::Tensor &attn, int64_t im2col_step,
                              bool deterministic)
    {
        ctx->saved_data["step"] = im2col_step;
        ctx->saved_data["det"] = deterministic;
        at::Tensor table;                                   // the forward's per-point table, if the backward of this geometry reads one
        const bool want = g_grad_mode_at_entry && (value.requires_grad() || loc.requires_grad() || attn.requires_grad());
        at::Tensor out = forward_t(value.to(loc.scalar_type()), shapes, lsi, loc, attn, im2col_step, want ? &table : nullptr);
        ctx->save_for_backward({value, shapes, lsi, loc, attn, table});
        return out;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext *ctx, torch::autograd::variable_list grads)
    {
        once_differentiable(grads, "MSDeformAttnFunction");
        const auto saved = ctx->get_saved_variables();
        const auto g = backward_t(saved[0].to(saved[3].scalar_type()), saved[1], saved[2], saved[3], saved[4], grads[0],
                                  ctx->saved_data["step"].toInt(), (ctx->saved_data["det"].toBool() || at::globalContext().deterministicAlgorithms()),
                                  saved[5]);
        return {g[0], at::Tensor(), at::Tensor(), g[1], g[2], at::Tensor(), at::Tensor()};
    }
};

at::Tensor apply(const at::Tensor &value, const at::Tensor &shapes, const at::Tensor &lsi, const at::Tensor &loc,
                 const at::Tensor &attn, int64_t im2col_step, bool deterministic)
{
    g_grad_mode_at_entry = at::GradMode::is_enabled();
    return MSDAFunction::apply(value, shapes, lsi, loc, attn, im2col_step, deterministic);
}

// ---- bf16 rows (uvhand_amd.functions.MSDeformAttnBF16Function as a C++ node; no reference counterpart) ----
Dims dims_bf16(const at::Tensor &value, const at::Tensor &shapes, const at::Tensor &lsi, const at::Tensor &loc,
               const at::Tensor &attn, int64_t im2col_step)
{
    TORCH_CHECK(value.scalar_type() == at::kBFloat16 && loc.scalar_type() == at::kFloat && attn.scalar_type() == at::kFloat,
                "bf16 rows: expected bfloat16 value and float32 sampling_loc / attn_weight");
    // shape / index checks are dtype-independent: run them on views that satisfy the dtype check of dims()
    TORCH_CHECK(value.dim() == 4 && loc.dim() == 6 && attn.dim() == 5,
                "ms_deform_attn: expected value[N,S,M,D], sampling_loc[N,Lq,M,L,P,2], attn_weight[N,Lq,M,L,P]");
    Dims d;
    d.N = (int)value.size(0); d.S = (int)value.size(1); d.M = (int)value.size(2); d.D = (int)value.size(3);
    d.L = (int)shapes.size(0); d.Lq = (int)loc.size(1); d.P = (int)loc.size(4);
    TORCH_CHECK(loc.sizes() == at::IntArrayRef({d.N, d.Lq, d.M, d.L, d.P, 2}) && attn.sizes() == at::IntArrayRef({d.N, d.Lq, d.M, d.L, d.P}),
                "ms_deform_attn: sampling_loc ", loc.sizes(), " / attn_weight ", attn.sizes(), " do not match value ", value.sizes());
    TORCH_CHECK(shapes.scalar_type() == at::kLong && lsi.scalar_type() == at::kLong,
                "expected scalar type Long for spatial_shapes / level_start_index");
    TORCH_CHECK(shapes.dim() == 2 && shapes.size(1) == 2 && lsi.dim() == 1 && lsi.size(0) == d.L,
                "ms_deform_attn: spatial_shapes must be [L,2] and level_start_index [L]");
    const int64_t step = std::min<int64_t>(d.N, im2col_step);
    TORCH_CHECK(d.N == 0 || (step > 0 && d.N % step == 0), "batch(", d.N, ") must divide im2col_step(", step, ")");
    return d;
}

inline const uint16_t *bf16_ptr(const at::Tensor &t) { return reinterpret_cast<const uint16_t *>(t.data_ptr<at::BFloat16>()); }
inline uint16_t *bf16_mut(at::Tensor &t) { return reinterpret_cast<uint16_t *>(t.data_ptr<at::BFloat16>()); }

class MSDABF16Function : public torch::autograd::Function<MSDABF16Function> {
public:
    static at::Tensor forward(torch::autograd::AutogradContext *ctx, const at::Tensor &value, const at::Tensor &shapes,
                              const at::Tensor &lsi, const at::Tensor &loc, const at::Tensor &attn, int64_t im2col_step,
                              bool deterministic)
    {
        ctx->saved_data["step"] = im2col_step;
        ctx->saved_data["det"] = deterministic;
        const at::Tensor v16 = value.to(at::kBFloat16), l32 = loc.to(at::kFloat), a32 = attn.to(at::kFloat);
        check_inputs({{"value", &v16}, {"spatial_shapes", &shapes}, {"level_start_index", &lsi}, {"sampling_loc", &l32},
                      {"attn_weight", &a32}});
        const Dims d = dims_bf16(v16, shapes, lsi, l32, a32, im2col_step);
        c10::hip::HIPGuardMasqueradingAsCUDA guard(v16.device());
        auto out = at::empty({d.N, d.Lq, (int64_t)d.M * d.D}, v16.options());
        auto stream = (msda_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(v16.device().index()).stream();
        at::Tensor table;
        if (g_grad_mode_at_entry && (value.requires_grad() || loc.requires_grad() || attn.requires_grad())) table = forward_table(v16, d, 0);
        raise_if(msda_forward_ws_bf16(bf16_ptr(v16), shapes.data_ptr<int64_t>(), lsi.data_ptr<int64_t>(), l32.data_ptr<float>(),
                                      a32.data_ptr<float>(), d.N, d.S, d.M, d.D, d.L, d.Lq, d.P, bf16_mut(out),
                                      table.defined() ? table.data_ptr() : nullptr,
                                      table.defined() ? (unsigned long long)table.numel() : 0, stream),
                 "ms_deform_attn_forward (bf16 rows)");
        ctx->save_for_backward({value, shapes, lsi, loc, attn, table});
        return out;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext *ctx, torch::autograd::variable_list grads)
    {
        once_differentiable(grads, "MSDeformAttnBF16Function");
        const auto saved = ctx->get_saved_variables();
        const at::Tensor &value = saved[0], &shapes = saved[1], &lsi = saved[2], &loc = saved[3], &attn = saved[4];
        const at::Tensor v16 = value.to(at::kBFloat16), l32 = loc.to(at::kFloat), a32 = attn.to(at::kFloat);
        const at::Tensor go = grads[0].to(at::kBFloat16).contiguous();
        check_inputs({{"value", &v16}, {"spatial_shapes", &shapes}, {"level_start_index", &lsi}, {"sampling_loc", &l32},
                      {"attn_weight", &a32}, {"grad_output", &go}});
        const Dims d = dims_bf16(v16, shapes, lsi, l32, a32, ctx->saved_data["step"].toInt());
        TORCH_CHECK(go.numel() == (int64_t)d.N * d.Lq * d.M * d.D, "ms_deform_attn_backward: grad_output has the wrong size");
        // fp32 grad_value straight from the kernel when that is what `value` needs, and for multi-pass backwards
        // — and wherever the generic kernels serve the call (they accumulate grad_value with fp32 atomics only)
        const bool gv32 = value.scalar_type() == at::kFloat || msda_backward_passes(d.Lq, d.P) > 1 ||
                          msda_path_for(2, d.M, d.D, d.L, d.P) != MSDA_PATH_D32 ||
                          (((uintptr_t)go.data_ptr() | (uintptr_t)v16.data_ptr() | (uintptr_t)l32.data_ptr()) & 7) != 0;
        bool det = ctx->saved_data["det"].toBool() || at::globalContext().deterministicAlgorithms();
        if (det && at::globalContext().deterministicAlgorithmsWarnOnly() && !msda_deterministic_supported(2, d.N, d.S, d.M, d.D, d.L, d.Lq, d.P)) {
            TORCH_WARN_ONCE("ms_deform_attn_backward (bf16 rows): no deterministic grad_value kernel for this geometry within the work "
                            "bound; running the default (atomic) kernel [warn_only]");
            det = false;
        }
        unsigned flags = (det ? MSDA_FLAG_DETERMINISTIC : 0u) | g_extra_flags.load();
        c10::hip::HIPGuardMasqueradingAsCUDA guard(v16.device());
        auto gv = at::empty_like(v16, v16.options().dtype(gv32 ? at::kFloat : at::kBFloat16));
        auto gl = at::empty_like(l32), ga = at::empty_like(a32);
        at::Tensor ws;
        unsigned long long nbytes = 0;
        if (saved[5].defined()) { ws = saved[5]; nbytes = (unsigned long long)ws.numel(); flags |= MSDA_FLAG_FORWARD_TABLE; }
        else if ((nbytes = msda_backward_workspace_bytes(d.N, d.S, d.M, d.D, d.L, d.Lq, d.P, flags)) != 0)
            ws = at::empty({(int64_t)nbytes}, v16.options().dtype(at::kByte));
        auto stream = (msda_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(v16.device().index()).stream();
        int rc;
        if (gv32)
            rc = msda_backward_ws_bf16_gv32(bf16_ptr(go), bf16_ptr(v16), shapes.data_ptr<int64_t>(), lsi.data_ptr<int64_t>(),
                                            l32.data_ptr<float>(), a32.data_ptr<float>(), d.N, d.S, d.M, d.D, d.L, d.Lq, d.P,
                                            gv.data_ptr<float>(), gl.data_ptr<float>(), ga.data_ptr<float>(),
                                            nbytes ? ws.data_ptr() : nullptr, nbytes, flags, stream);
        else
            rc = msda_backward_ws_bf16(bf16_ptr(go), bf16_ptr(v16), shapes.data_ptr<int64_t>(), lsi.data_ptr<int64_t>(),
                                       l32.data_ptr<float>(), a32.data_ptr<float>(), d.N, d.S, d.M, d.D, d.L, d.Lq, d.P, bf16_mut(gv),
                                       gl.data_ptr<float>(), ga.data_ptr<float>(), nbytes ? ws.data_ptr() : nullptr, nbytes, flags,
                                       stream);
        raise_if(rc, "ms_deform_attn_backward (bf16 rows)");
        return {gv.to(value.scalar_type()), at::Tensor(), at::Tensor(), gl.to(loc.scalar_type()), ga.to(attn.scalar_type()),
                at::Tensor(), at::Tensor()};
    }
};

at::Tensor apply_bf16(const at::Tensor &value, const at::Tensor &shapes, const at::Tensor &lsi, const at::Tensor &loc,
                      const at::Tensor &attn, int64_t im2col_step, bool deterministic)
{
    g_grad_mode_at_entry = at::GradMode::is_enabled();
    return MSDABF16Function::apply(value, shapes, lsi, loc, attn, im2col_step, deterministic);
}

// ---- the whole MSDeformAttn module as ONE autograd node (fp32, fused prologue + merged projection) -------------------------
// What uvhand_amd/modules/ms_deform_attn.py composes from four Python autograd Functions and ctypes calls — value_proj
// (+ padding-mask rows), the merged sampling_offsets / attention_weights projection, the fused-prologue sampling kernels,
// output_proj, and in the backward the input-gradient GEMMs plus the MFMA weight-gradient kernel of every projection
// (models/ops/modules/ms_deform_attn.py:96-139 and its autograd) — queued from C++ with no Python in between: the eager
// step at decoder sizes is bound by host time, not by the GPU (profiles/r02_notes.md section 7).  Same kernels, same
// arithmetic, same results as the Python composition (tests/test_module_gpu.py runs both).
inline bool aligned16(const at::Tensor &t) { return (reinterpret_cast<uintptr_t>(t.data_ptr()) & 15) == 0; }

inline at::Tensor wgrad_into(const at::Tensor &dY2, const at::Tensor &X2, const at::Tensor *mask, at::Tensor &gb, bool want_bias,
                             msda_stream_t stream)
{
    const int M = (int)dY2.size(0), N = (int)dY2.size(1), K = (int)X2.size(1);
    // what msda_linear_wgrad_masked_f32 takes (include/msda.h): N, K multiples of 4, 16-byte aligned dense operands — else the
    // same composition functions/linear_func.py falls back to (n_heads*n_levels*n_points == 2, an unaligned grad_output view)
    if (N % 4 != 0 || K % 4 != 0 || M <= 0 || !dY2.is_contiguous() || !X2.is_contiguous() || !aligned16(dY2) || !aligned16(X2)) {
        const at::Tensor dy = mask ? dY2.masked_fill(mask->view({-1, 1}), 0) : dY2;
        if (want_bias) gb = dy.sum(0);
        return dy.t().mm(X2);
    }
    auto gw = at::empty({N, K}, dY2.options());
    if (want_bias) gb = at::empty({N}, dY2.options());
    const unsigned long long nbytes = msda_linear_wgrad_workspace_bytes(M, N, K);
    at::Tensor ws;
    if (nbytes) ws = at::empty({(int64_t)nbytes}, dY2.options().dtype(at::kByte));
    raise_if(msda_linear_wgrad_masked_f32(dY2.data_ptr<float>(), X2.data_ptr<float>(),
                                          mask ? reinterpret_cast<const uint8_t *>(mask->data_ptr<bool>()) : nullptr, M, N, K,
                                          gw.data_ptr<float>(), want_bias ? gb.data_ptr<float>() : nullptr,
                                          nbytes ? ws.data_ptr() : nullptr, stream),
             "msda_linear_wgrad");
    return gw;
}

// Several weight gradients of one backward in ONE library call (msda_linear_wgrad_multi: first stages back to back, one second
// stage for all; operands fp32 or bf16 per job, results fp32); a job the kernel's preconditions exclude takes wgrad_into's
// fallback on its own (fp32 jobs; the bf16 node's operands are its own dense tensors).
struct WgradJob { at::Tensor dY, X; const at::Tensor *mask; at::Tensor *gw, *gb; };
inline void wgrad_jobs(std::vector<WgradJob> &jobs, msda_stream_t stream)
{
    const void *dy[4], *x[4]; const uint8_t *mk[4]; int bf[4], M[4], N[4], K[4]; float *gw[4], *gb[4]; void *ws[4];
    std::vector<at::Tensor> keep;
    int n = 0;
    for (auto &j : jobs) {
        const int m = (int)j.dY.size(0), nn = (int)j.dY.size(1), k = (int)j.X.size(1);
        const bool half = j.dY.scalar_type() == at::kBFloat16;
        if (!half && (n == 4 || nn % 4 != 0 || k % 4 != 0 || m <= 0 || !j.dY.is_contiguous() || !j.X.is_contiguous() || !aligned16(j.dY) ||
                      !aligned16(j.X))) {
            *j.gw = wgrad_into(j.dY, j.X, j.mask, *j.gb, true, stream);
            continue;
        }
        TORCH_CHECK(n < 4, "wgrad_jobs: at most four weight gradients per call");
        const auto f32 = j.dY.options().dtype(at::kFloat);
        *j.gw = at::empty({nn, k}, f32);
        *j.gb = at::empty({nn}, f32);
        const unsigned long long nbytes = msda_linear_wgrad_workspace_bytes(m, nn, k);
        ws[n] = nullptr;
        if (nbytes) { keep.push_back(at::empty({(int64_t)nbytes}, j.dY.options().dtype(at::kByte))); ws[n] = keep.back().data_ptr(); }
        dy[n] = j.dY.data_ptr(); x[n] = j.X.data_ptr(); bf[n] = half ? 1 : 0;
        mk[n] = j.mask ? reinterpret_cast<const uint8_t *>(j.mask->data_ptr<bool>()) : nullptr;
        M[n] = m; N[n] = nn; K[n] = k; gw[n] = j.gw->data_ptr<float>(); gb[n] = j.gb->data_ptr<float>();
        ++n;
    }
    if (n > 0) raise_if(msda_linear_wgrad_multi(n, dy, x, bf, mk, M, N, K, gw, gb, ws, stream), "msda_linear_wgrad_multi");
}

// nn.Linear forward / input gradient of the module's projections.  While the problem is small the library's own fp32-MFMA
// kernels (msda_linear_forward_f32 / msda_linear_dgrad_f32: one plain launch, ~7 us on the host against ~18-27 us for a GEMM
// through torch, and as fast on the GPU up to ~5000 rows at 256 features — tools/gemm_time.py); the vendor BLAS beyond, where
// its 256 x 144 macro-tiles run 1.3x faster.  rmask (may be null): rows written as zeros.
static bool own_linear(const at::Tensor &a2, const at::Tensor &w)
{
    const int64_t rows = a2.size(0), out_f = w.size(0), in_f = w.size(1);
    return a2.dim() == 2 && w.dim() == 2 && rows * out_f <= 4800LL * 256 && in_f <= 512 && out_f % 4 == 0 && in_f % 4 == 0 && w.is_contiguous() &&
           a2.is_contiguous() && (reinterpret_cast<uintptr_t>(a2.data_ptr()) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(w.data_ptr()) & 15) == 0;
}

static const uint8_t *mask_bytes(const at::Tensor *rmask)
{
    return rmask ? reinterpret_cast<const uint8_t *>(rmask->data_ptr<bool>()) : nullptr;
}

static at::Tensor linear_rows_forward(const at::Tensor &x2, const at::Tensor &w, const at::Tensor &b, const at::Tensor *rmask,
                                      msda_stream_t stream)
{
    const int64_t rows = x2.size(0);
    TORCH_CHECK(x2.dim() == 2 && w.dim() == 2 && x2.size(1) == w.size(1) && b.numel() == w.size(0),
                "MSDeformAttn (C++ node): a projection's input ", x2.sizes(), " does not match its weight ", w.sizes());
    if (own_linear(x2, w) && b.is_contiguous() && (reinterpret_cast<uintptr_t>(b.data_ptr()) & 15) == 0) {
        at::Tensor y = at::empty({rows, w.size(0)}, x2.options());
        raise_if(msda_linear_forward_f32(x2.data_ptr<float>(), w.data_ptr<float>(), b.data_ptr<float>(), mask_bytes(rmask), rows,
                                         (int)w.size(0), (int)w.size(1), y.data_ptr<float>(), stream), "msda_linear_forward");
        return y;
    }
    at::Tensor y = at::addmm(b, x2, w.t());
    if (rmask) raise_if(msda_zero_masked_rows_f32(y.data_ptr<float>(), mask_bytes(rmask), rows, (int)w.size(0), stream),
                        "msda_zero_masked_rows");
    return y;
}

static at::Tensor linear_rows_dgrad(const at::Tensor &g2, const at::Tensor &w, const at::Tensor *rmask, msda_stream_t stream)
{
    const int64_t rows = g2.size(0);
    TORCH_CHECK(g2.dim() == 2 && w.dim() == 2 && g2.size(1) == w.size(0),
                "MSDeformAttn (C++ node): a projection's output gradient ", g2.sizes(), " does not match its weight ", w.sizes());
    if (own_linear(g2, w)) {
        at::Tensor gx = at::empty({rows, w.size(1)}, g2.options());
        raise_if(msda_linear_dgrad_f32(g2.data_ptr<float>(), w.data_ptr<float>(), mask_bytes(rmask), rows, (int)w.size(0),
                                       (int)w.size(1), gx.data_ptr<float>(), stream), "msda_linear_dgrad");
        return gx;
    }
    at::Tensor gx = at::mm(g2, w);
    if (rmask) raise_if(msda_zero_masked_rows_f32(gx.data_ptr<float>(), mask_bytes(rmask), rows, (int)w.size(1), stream),
                        "msda_zero_masked_rows");
    return gx;
}

// Everything the kernels index with: checked here because the node passes raw pointers (the Python composition raised
// 'inconsistent shapes' from uvhand_amd._native; models/ops/modules/ms_deform_attn.py:92-94 asserts the pixel count only).
static void check_module_args(const at::Tensor &query, const at::Tensor &centre, const at::Tensor &input_flatten,
                              const c10::optional<at::Tensor> &mask, const at::Tensor &shapes, const at::Tensor &lsi,
                              const at::Tensor &w_off, const at::Tensor &b_off, const at::Tensor &w_attn, const at::Tensor &b_attn,
                              const at::Tensor &w_val, const at::Tensor &b_val, const at::Tensor &w_out, const at::Tensor &b_out,
                              int64_t M, int64_t L, int64_t P)
{
    TORCH_CHECK(M > 0 && L > 0 && P > 0, "MSDeformAttn (C++ node): n_heads, n_levels, n_points must be positive");
    TORCH_CHECK(query.dim() == 3 && input_flatten.dim() == 3 && query.size(0) == input_flatten.size(0) &&
                query.size(2) == input_flatten.size(2),
                "MSDeformAttn (C++ node): expected query[N,Lq,C] and input_flatten[N,S,C], got ", query.sizes(), " and ",
                input_flatten.sizes());
    const int64_t N = query.size(0), Lq = query.size(1), C = query.size(2), S = input_flatten.size(1), mlp = M * L * P;
    TORCH_CHECK(C % M == 0, "MSDeformAttn (C++ node): d_model ", C, " is not divisible by n_heads ", M);
    TORCH_CHECK(centre.sizes() == at::IntArrayRef({N, Lq, L, 2}),
                "MSDeformAttn (C++ node): reference points must be [N,Lq,n_levels,2] = [", N, ",", Lq, ",", L, ",2], got ", centre.sizes());
    TORCH_CHECK(shapes.sizes() == at::IntArrayRef({L, 2}) && lsi.sizes() == at::IntArrayRef({L}),
                "MSDeformAttn (C++ node): spatial_shapes must be [", L, ",2] and level_start_index [", L, "], got ", shapes.sizes(),
                " and ", lsi.sizes());
    TORCH_CHECK(shapes.is_contiguous() && lsi.is_contiguous(), "spatial_shapes / level_start_index have to be contiguous");
    if (mask.has_value() && mask->defined())
        TORCH_CHECK(mask->scalar_type() == at::kBool && mask->is_cuda() && mask->device() == query.device() && mask->numel() == N * S,
                    "MSDeformAttn (C++ node): padding mask must be a bool CUDA tensor of N*S = ", N * S, " elements");
    TORCH_CHECK(w_off.sizes() == at::IntArrayRef({2 * mlp, C}) && b_off.sizes() == at::IntArrayRef({2 * mlp}),
                "MSDeformAttn (C++ node): sampling_offsets must be Linear(", C, ", ", 2 * mlp, "), got weight ", w_off.sizes());
    TORCH_CHECK(w_attn.sizes() == at::IntArrayRef({mlp, C}) && b_attn.sizes() == at::IntArrayRef({mlp}),
                "MSDeformAttn (C++ node): attention_weights must be Linear(", C, ", ", mlp, "), got weight ", w_attn.sizes());
    TORCH_CHECK(w_val.sizes() == at::IntArrayRef({C, C}) && b_val.sizes() == at::IntArrayRef({C}) &&
                w_out.sizes() == at::IntArrayRef({C, C}) && b_out.sizes() == at::IntArrayRef({C}),
                "MSDeformAttn (C++ node): value_proj / output_proj must be Linear(", C, ", ", C, ")");
    TORCH_CHECK(mlp % 2 == 0, "MSDeformAttn (C++ node): n_heads*n_levels*n_points must be even (the kernels move (x, y) pairs)");
}

class MSDAModuleFunction : public torch::autograd::Function<MSDAModuleFunction> {
public:
    // inputs 0..2 may need gradients (query, centre = reference point per level, input_flatten), 6..13 are the parameters
    static at::Tensor forward(torch::autograd::AutogradContext *ctx, const at::Tensor &query, const at::Tensor &centre,
                              const at::Tensor &input_flatten, const c10::optional<at::Tensor> &mask, const at::Tensor &shapes,
                              const at::Tensor &lsi, const at::Tensor &w_off, const at::Tensor &b_off, const at::Tensor &w_attn,
                              const at::Tensor &b_attn, const at::Tensor &w_val, const at::Tensor &b_val, const at::Tensor &w_out,
                              const at::Tensor &b_out, int64_t n_heads, int64_t n_levels, int64_t n_points, int64_t im2col_step,
                              bool deterministic, const c10::optional<at::Tensor> &wm_cached,
                              const c10::optional<at::Tensor> &bm_cached)
    {
        const int N = (int)query.size(0), Lq = (int)query.size(1), C = (int)query.size(2), S = (int)input_flatten.size(1);
        const int M = (int)n_heads, L = (int)n_levels, P = (int)n_points, D = C / M, mlp = M * L * P;
        TORCH_CHECK(msda_prologue_supported(N, S, M, D, L, Lq, P), "MSDeformAttn (C++ node): geometry outside the fused prologue");
        const int64_t step = std::min<int64_t>(N, im2col_step);
        TORCH_CHECK(N == 0 || (step > 0 && N % step == 0), "batch(", N, ") must divide im2col_step(", step, ")");
        c10::hip::HIPGuardMasqueradingAsCUDA guard(query.device());
        auto stream = (msda_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(query.device().index()).stream();
        const at::Tensor q2 = query.contiguous(), x2 = input_flatten.contiguous();
        at::Tensor c2 = centre.contiguous();
        if (reinterpret_cast<uintptr_t>(c2.data_ptr()) & 7) c2 = c2.clone();       // the kernels read (x, y) pairs as 8 bytes
        at::Tensor rmask;
        if (mask.has_value() && mask->defined()) rmask = mask->reshape({-1}).contiguous();
        // value_proj (+ the padding mask on the masked rows only)
        const at::Tensor value = linear_rows_forward(x2.view({(int64_t)N * S, C}), w_val, b_val, rmask.defined() ? &rmask : nullptr,
                                                     stream);                                 // [N*S, C]
        // sampling_offsets and attention_weights as ONE GEMM (both read the same query); the kernels read its output in place
        // (the caller may hand over the concatenation it keeps while the four parameters are unchanged: two launches less)
        const bool cached = wm_cached.has_value() && wm_cached->defined() && bm_cached.has_value() && bm_cached->defined();
        if (cached)
            TORCH_CHECK(wm_cached->sizes() == at::IntArrayRef({3LL * mlp, (int64_t)C}) && bm_cached->sizes() == at::IntArrayRef({3LL * mlp}) &&
                        wm_cached->scalar_type() == at::kFloat && bm_cached->scalar_type() == at::kFloat && wm_cached->is_contiguous() &&
                        bm_cached->is_contiguous() && wm_cached->device() == query.device() && bm_cached->device() == query.device(),
                        "MSDeformAttn (C++ node): cached merged projection has the wrong shape / dtype / device");
        const at::Tensor wm = cached ? wm_cached->detach() : at::cat({w_off, w_attn}, 0), bm = cached ? bm_cached->detach() : at::cat({b_off, b_attn}, 0);
        const at::Tensor projected = linear_rows_forward(q2.view({(int64_t)N * Lq, C}), wm, bm, nullptr, stream);   // [N*Lq, 3*mlp]
        auto sampled = at::empty({N, Lq, C}, q2.options());
        auto loc = at::empty({N, Lq, M, L, P, 2}, q2.options()), attn = at::empty({N, Lq, M, L, P}, q2.options());
        at::Tensor table;                                   // per-point table for the backward of this node (small problems)
        if (g_grad_mode_at_entry)
            table = forward_table(q2, Dims{N, S, M, D, L, Lq, P}, MSDA_FLAG_PROLOGUE);
        raise_if(msda_forward_prologue_ws_f32(value.data_ptr<float>(), shapes.data_ptr<int64_t>(), lsi.data_ptr<int64_t>(),
                                              c2.data_ptr<float>(), projected.data_ptr<float>(), projected.data_ptr<float>() + 2 * mlp,
                                              N, S, M, D, L, Lq, P, 3LL * mlp, 3LL * mlp, sampled.data_ptr<float>(),
                                              loc.data_ptr<float>(), attn.data_ptr<float>(), table.defined() ? table.data_ptr() : nullptr,
                                              table.defined() ? (unsigned long long)table.numel() : 0, stream),
                 "ms_deform_attn_forward_prologue");
        ctx->save_for_backward({q2, x2, rmask.defined() ? rmask : at::Tensor(), value, loc, attn, sampled, wm, w_val, w_out, shapes, lsi, table});
        ctx->saved_data["dims"] = std::vector<int64_t>{N, S, M, D, L, Lq, P, C};
        ctx->saved_data["det"] = deterministic;
        return linear_rows_forward(sampled.view({(int64_t)N * Lq, C}), w_out, b_out, nullptr, stream).view({N, Lq, C});
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext *ctx, torch::autograd::variable_list grads)
    {
        once_differentiable(grads, "MSDeformAttn (C++ node)");
        const auto sv = ctx->get_saved_variables();
        const at::Tensor &q2 = sv[0], &x2 = sv[1], &rmask = sv[2], &value = sv[3], &loc = sv[4], &attn = sv[5], &sampled = sv[6],
                         &wm = sv[7], &w_val = sv[8], &w_out = sv[9], &shapes = sv[10], &lsi = sv[11];
        // forward arguments 6..13 are the parameters (w_off, b_off, w_attn, b_attn, w_val, b_val, w_out, b_out): a frozen layer
        // (util/settings.py:447-515 lr groups, --not_use_params) gets no weight-gradient launch
        const bool need_m = ctx->needs_input_grad(6) || ctx->needs_input_grad(7) || ctx->needs_input_grad(8) || ctx->needs_input_grad(9);
        const bool need_val = ctx->needs_input_grad(10) || ctx->needs_input_grad(11);
        const bool need_out = ctx->needs_input_grad(12) || ctx->needs_input_grad(13);
        const auto dims = ctx->saved_data["dims"].toIntVector();
        const int N = (int)dims[0], S = (int)dims[1], M = (int)dims[2], D = (int)dims[3], L = (int)dims[4], Lq = (int)dims[5],
                  P = (int)dims[6], C = (int)dims[7], mlp = M * L * P;
        c10::hip::HIPGuardMasqueradingAsCUDA guard(q2.device());
        auto stream = (msda_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(q2.device().index()).stream();
        const at::Tensor go2 = grads[0].contiguous().view({(int64_t)N * Lq, C});
        const at::Tensor *maskp = rmask.defined() ? &rmask : nullptr;
        // output_proj
        at::Tensor gb_out, gb_m, gb_val;
        const at::Tensor g_sampled = linear_rows_dgrad(go2, w_out, nullptr, stream);           // [N*Lq, C]
        at::Tensor gw_out, gw_m, gw_val;
        std::vector<WgradJob> jobs;                          // the (up to three) weight gradients: one call at the end
        if (need_out) jobs.push_back(WgradJob{go2, sampled.view({(int64_t)N * Lq, C}), nullptr, &gw_out, &gb_out});
        // the sampling kernels: gradients of value, of the raw offsets / logits (one tensor, the projection's layout) and of
        // the reference points
        const bool det = ctx->saved_data["det"].toBool() || at::globalContext().deterministicAlgorithms();
        const unsigned flags = MSDA_FLAG_PROLOGUE | (det ? MSDA_FLAG_DETERMINISTIC : 0u);
        unsigned long long nbytes = 0;
        at::Tensor ws;
        unsigned table_flag = 0;
        if (sv[12].defined()) { ws = sv[12]; nbytes = (unsigned long long)ws.numel(); table_flag = MSDA_FLAG_FORWARD_TABLE; }   // table first, scratch behind it
        else if ((nbytes = msda_backward_workspace_bytes(N, S, M, D, L, Lq, P, flags)) != 0)
            ws = at::empty({(int64_t)nbytes}, q2.options().dtype(at::kByte));
        auto gv = at::empty_like(value), gproj = at::empty({(int64_t)N * Lq, 3LL * mlp}, q2.options());
        auto gref = at::empty({N, Lq, L, 2}, q2.options());
        raise_if(msda_backward_prologue_ws_f32(g_sampled.data_ptr<float>(), value.data_ptr<float>(), shapes.data_ptr<int64_t>(),
                                               lsi.data_ptr<int64_t>(), loc.data_ptr<float>(), attn.data_ptr<float>(), N, S, M, D, L,
                                               Lq, P, 3LL * mlp, 3LL * mlp, gv.data_ptr<float>(), gproj.data_ptr<float>(),
                                               gproj.data_ptr<float>() + 2 * mlp, gref.data_ptr<float>(),
                                               nbytes ? ws.data_ptr() : nullptr, nbytes, (det ? MSDA_FLAG_DETERMINISTIC : 0u) | table_flag | g_extra_flags.load(), stream),
                 "ms_deform_attn_backward_prologue");
        // merged projection
        at::Tensor g_query;
        if (ctx->needs_input_grad(0)) g_query = linear_rows_dgrad(gproj, wm, nullptr, stream).view({N, Lq, C});
        if (need_m) jobs.push_back(WgradJob{gproj, q2.view({(int64_t)N * Lq, C}), nullptr, &gw_m, &gb_m});
        // value_proj: masked rows of grad_value count as zero (weight gradient) and get a zero input gradient
        const at::Tensor gv2 = gv.view({(int64_t)N * S, C});
        at::Tensor g_input;
        if (ctx->needs_input_grad(2)) g_input = linear_rows_dgrad(gv2, w_val, maskp, stream).view({N, S, C});
        if (need_val) jobs.push_back(WgradJob{gv2, x2.view({(int64_t)N * S, C}), maskp, &gw_val, &gb_val});
        wgrad_jobs(jobs, stream);
        const at::Tensor none;
        return {g_query, gref, g_input, none, none, none,
                need_m ? gw_m.narrow(0, 0, 2 * mlp) : none, need_m ? gb_m.narrow(0, 0, 2 * mlp) : none,
                need_m ? gw_m.narrow(0, 2 * mlp, mlp) : none, need_m ? gb_m.narrow(0, 2 * mlp, mlp) : none,
                gw_val, gb_val, gw_out, gb_out, none, none, none, none, none, none, none};
    }
};

// The module under torch.autocast(bfloat16) with bf16 rows (MSDeformAttn.bf16_storage) as ONE node — the same steps, in the same
// order, as the Python composition takes there (modules/ms_deform_attn.py; functions/linear_func.py: _BracketLinearAmpFn):
// value_proj and output_proj on bf16 operands (what autocast runs), the query projection, offsets, logits, reference points
// and every gradient in float32, bf16 rows through the sampling kernels, grad_value accumulated in float32, weight gradients
// from bf16 operands on the bf16 MFMA with float32 results.
class MSDAModuleBF16Function : public torch::autograd::Function<MSDAModuleBF16Function> {
public:
    static at::Tensor forward(torch::autograd::AutogradContext *ctx, const at::Tensor &query, const at::Tensor &centre,
                              const at::Tensor &input_flatten, const c10::optional<at::Tensor> &mask, const at::Tensor &shapes,
                              const at::Tensor &lsi, const at::Tensor &w_off, const at::Tensor &b_off, const at::Tensor &w_attn,
                              const at::Tensor &b_attn, const at::Tensor &w_val, const at::Tensor &b_val, const at::Tensor &w_out,
                              const at::Tensor &b_out, int64_t n_heads, int64_t n_levels, int64_t n_points, int64_t im2col_step,
                              bool deterministic, const c10::optional<at::Tensor> &wm_cached,
                              const c10::optional<at::Tensor> &bm_cached)
    {
        const int N = (int)query.size(0), Lq = (int)query.size(1), C = (int)query.size(2), S = (int)input_flatten.size(1);
        const int M = (int)n_heads, L = (int)n_levels, P = (int)n_points, D = C / M, mlp = M * L * P;
        TORCH_CHECK(msda_prologue_supported(N, S, M, D, L, Lq, P), "MSDeformAttn (C++ node): geometry outside the fused prologue");
        const int64_t step = std::min<int64_t>(N, im2col_step);
        TORCH_CHECK(N == 0 || (step > 0 && N % step == 0), "batch(", N, ") must divide im2col_step(", step, ")");
        c10::hip::HIPGuardMasqueradingAsCUDA guard(query.device());
        auto stream = (msda_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(query.device().index()).stream();
        // the casts are spelled out below: autocast must not re-cast the float32 query projection
        c10::impl::ExcludeDispatchKeyGuard no_autocast(c10::DispatchKeySet(c10::DispatchKey::AutocastCUDA));
        const at::Tensor q2 = query.contiguous();
        at::Tensor c2 = centre.contiguous();
        if (reinterpret_cast<uintptr_t>(c2.data_ptr()) & 7) c2 = c2.clone();       // the kernels read (x, y) pairs as 8 bytes
        at::Tensor rmask;
        if (mask.has_value() && mask->defined()) rmask = mask->reshape({-1}).contiguous();
        // value_proj on bf16 operands, then the padding mask
        const at::Tensor xb = input_flatten.to(at::kBFloat16).contiguous();
        // the four parameter casts autocast would run one by one — in ONE launch into one slab (msda_cast_bf16_multi_f32)
        at::Tensor wvb, bvb, wob, bob;
        {
            const int64_t cc = (int64_t)C * C;
            const at::Tensor wv = w_val.contiguous(), bv = b_val.contiguous(), wo = w_out.contiguous(), bo = b_out.contiguous();
            if (C % 2 == 0 && ((reinterpret_cast<uintptr_t>(wv.data_ptr()) | reinterpret_cast<uintptr_t>(bv.data_ptr()) |
                                reinterpret_cast<uintptr_t>(wo.data_ptr()) | reinterpret_cast<uintptr_t>(bo.data_ptr())) & 7) == 0) {
                const at::Tensor slab = at::empty({2 * cc + 2 * C}, xb.options());
                wvb = slab.narrow(0, 0, cc).view({C, C}); wob = slab.narrow(0, cc, cc).view({C, C});
                bvb = slab.narrow(0, 2 * cc, C); bob = slab.narrow(0, 2 * cc + C, C);
                const float *src[4] = {wv.data_ptr<float>(), wo.data_ptr<float>(), bv.data_ptr<float>(), bo.data_ptr<float>()};
                uint16_t *dst[4] = {reinterpret_cast<uint16_t *>(wvb.data_ptr<at::BFloat16>()), reinterpret_cast<uint16_t *>(wob.data_ptr<at::BFloat16>()),
                                    reinterpret_cast<uint16_t *>(bvb.data_ptr<at::BFloat16>()), reinterpret_cast<uint16_t *>(bob.data_ptr<at::BFloat16>())};
                const long long cnt[4] = {cc, cc, C, C};
                raise_if(msda_cast_bf16_multi_f32(4, src, dst, cnt, stream), "msda_cast_bf16_multi");
            } else {
                wvb = w_val.to(at::kBFloat16); bvb = b_val.to(at::kBFloat16); wob = w_out.to(at::kBFloat16); bob = b_out.to(at::kBFloat16);
            }
        }
        at::Tensor value = at::linear(xb, wvb, bvb);                                     // [N, S, C] bf16
        if (rmask.defined()) value = value.masked_fill(rmask.view({N, S, 1}), 0);
        // the query projection stays float32
        // (the caller may hand over the concatenation it keeps while the four parameters are unchanged: two launches less)
        const bool cached = wm_cached.has_value() && wm_cached->defined() && bm_cached.has_value() && bm_cached->defined();
        if (cached)
            TORCH_CHECK(wm_cached->sizes() == at::IntArrayRef({3LL * mlp, (int64_t)C}) && bm_cached->sizes() == at::IntArrayRef({3LL * mlp}) &&
                        wm_cached->scalar_type() == at::kFloat && bm_cached->scalar_type() == at::kFloat && wm_cached->is_contiguous() &&
                        bm_cached->is_contiguous() && wm_cached->device() == query.device() && bm_cached->device() == query.device(),
                        "MSDeformAttn (C++ node): cached merged projection has the wrong shape / dtype / device");
        const at::Tensor wm = cached ? wm_cached->detach() : at::cat({w_off, w_attn}, 0), bm = cached ? bm_cached->detach() : at::cat({b_off, b_attn}, 0);
        const at::Tensor projected = linear_rows_forward(q2.view({(int64_t)N * Lq, C}), wm, bm, nullptr, stream);
        auto sampled = at::empty({N, Lq, C}, xb.options());
        auto loc = at::empty({N, Lq, M, L, P, 2}, q2.options()), attn = at::empty({N, Lq, M, L, P}, q2.options());
        at::Tensor table;
        if (g_grad_mode_at_entry)
            table = forward_table(q2, Dims{N, S, M, D, L, Lq, P}, MSDA_FLAG_PROLOGUE);
        raise_if(msda_forward_prologue_ws_bf16(reinterpret_cast<const uint16_t *>(value.data_ptr<at::BFloat16>()),
                                               shapes.data_ptr<int64_t>(), lsi.data_ptr<int64_t>(), c2.data_ptr<float>(),
                                               projected.data_ptr<float>(), projected.data_ptr<float>() + 2 * mlp, N, S, M, D, L, Lq,
                                               P, 3LL * mlp, 3LL * mlp, reinterpret_cast<uint16_t *>(sampled.data_ptr<at::BFloat16>()),
                                               loc.data_ptr<float>(), attn.data_ptr<float>(), table.defined() ? table.data_ptr() : nullptr,
                                               table.defined() ? (unsigned long long)table.numel() : 0, stream),
                 "ms_deform_attn_forward_prologue (bf16 rows)");
        ctx->save_for_backward({q2, xb, rmask.defined() ? rmask : at::Tensor(), value, loc, attn, sampled, wm, wvb, wob, shapes, lsi, table});
        ctx->saved_data["dims"] = std::vector<int64_t>{N, S, M, D, L, Lq, P, C};
        ctx->saved_data["det"] = deterministic;
        ctx->saved_data["x_float"] = input_flatten.scalar_type() == at::kFloat;
        return at::linear(sampled, wob, bob);                                           // [N, Lq, C] bf16
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext *ctx, torch::autograd::variable_list grads)
    {
        once_differentiable(grads, "MSDeformAttn (C++ node, bf16 rows)");
        const bool need_m = ctx->needs_input_grad(6) || ctx->needs_input_grad(7) || ctx->needs_input_grad(8) || ctx->needs_input_grad(9);
        const bool need_val = ctx->needs_input_grad(10) || ctx->needs_input_grad(11);
        const bool need_out = ctx->needs_input_grad(12) || ctx->needs_input_grad(13);
        const auto sv = ctx->get_saved_variables();
        const at::Tensor &q2 = sv[0], &xb = sv[1], &rmask = sv[2], &value = sv[3], &loc = sv[4], &attn = sv[5], &sampled = sv[6],
                         &wm = sv[7], &wvb = sv[8], &wob = sv[9], &shapes = sv[10], &lsi = sv[11];
        const auto dims = ctx->saved_data["dims"].toIntVector();
        const int N = (int)dims[0], S = (int)dims[1], M = (int)dims[2], D = (int)dims[3], L = (int)dims[4], Lq = (int)dims[5],
                  P = (int)dims[6], C = (int)dims[7], mlp = M * L * P;
        c10::hip::HIPGuardMasqueradingAsCUDA guard(q2.device());
        auto stream = (msda_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(q2.device().index()).stream();
        // output_proj
        const at::Tensor go2 = grads[0].reshape({(int64_t)N * Lq, C}).to(at::kBFloat16).contiguous();
        at::Tensor gb_out, gb_m, gb_val;
        const at::Tensor g_sampled = at::mm(go2, wob);                                  // [N*Lq, C] bf16
        at::Tensor gw_out, gw_m, gw_val;
        // (the three weight gradients are queued and run at the end: ONE second stage for all of them, wgrad_jobs)
        std::vector<WgradJob> jobs;
        if (need_out) jobs.push_back({go2, sampled.view({(int64_t)N * Lq, C}), nullptr, &gw_out, &gb_out});
        // the sampling kernels: bf16 rows in, float32 gradients out
        const bool det = ctx->saved_data["det"].toBool() || at::globalContext().deterministicAlgorithms();
        const unsigned flags = MSDA_FLAG_PROLOGUE | (det ? MSDA_FLAG_DETERMINISTIC : 0u);
        unsigned long long nbytes = 0;
        at::Tensor ws;
        unsigned table_flag = 0;
        if (sv[12].defined()) { ws = sv[12]; nbytes = (unsigned long long)ws.numel(); table_flag = MSDA_FLAG_FORWARD_TABLE; }   // table first, scratch behind it
        else if ((nbytes = msda_backward_workspace_bytes(N, S, M, D, L, Lq, P, flags)) != 0)
            ws = at::empty({(int64_t)nbytes}, q2.options().dtype(at::kByte));
        auto gv = at::empty({N, S, C}, q2.options()), gproj = at::empty({(int64_t)N * Lq, 3LL * mlp}, q2.options());
        auto gref = at::empty({N, Lq, L, 2}, q2.options());
        raise_if(msda_backward_prologue_bf16_gv32(reinterpret_cast<const uint16_t *>(g_sampled.data_ptr<at::BFloat16>()),
                                                  reinterpret_cast<const uint16_t *>(value.data_ptr<at::BFloat16>()),
                                                  shapes.data_ptr<int64_t>(), lsi.data_ptr<int64_t>(), loc.data_ptr<float>(),
                                                  attn.data_ptr<float>(), N, S, M, D, L, Lq, P, 3LL * mlp, 3LL * mlp,
                                                  gv.data_ptr<float>(), gproj.data_ptr<float>(), gproj.data_ptr<float>() + 2 * mlp,
                                                  gref.data_ptr<float>(), nbytes ? ws.data_ptr() : nullptr, nbytes,
                                                  (det ? MSDA_FLAG_DETERMINISTIC : 0u) | table_flag | g_extra_flags.load(), stream),
                 "ms_deform_attn_backward_prologue (bf16 rows)");
        // merged projection (float32)
        at::Tensor g_query;
        if (ctx->needs_input_grad(0)) g_query = linear_rows_dgrad(gproj, wm, nullptr, stream).view({N, Lq, C});
        if (need_m) jobs.push_back({gproj, q2.view({(int64_t)N * Lq, C}), nullptr, &gw_m, &gb_m});
        // value_proj: grad_value goes back to the rows' type, the mask's rows to zero, then the two GEMMs on bf16 operands
        at::Tensor gvb = gv.to(at::kBFloat16);
        if (rmask.defined()) gvb = gvb.masked_fill(rmask.view({N, S, 1}), 0);
        const at::Tensor gvb2 = gvb.view({(int64_t)N * S, C});
        at::Tensor g_input;
        if (ctx->needs_input_grad(2)) {
            g_input = at::mm(gvb2, wvb).view({N, S, C});
            if (ctx->saved_data["x_float"].toBool()) g_input = g_input.to(at::kFloat);
        }
        if (need_val) jobs.push_back({gvb2, xb.view({(int64_t)N * S, C}), nullptr, &gw_val, &gb_val});
        wgrad_jobs(jobs, stream);
        const at::Tensor none;
        return {g_query, gref, g_input, none, none, none,
                need_m ? gw_m.narrow(0, 0, 2 * mlp) : none, need_m ? gb_m.narrow(0, 0, 2 * mlp) : none,
                need_m ? gw_m.narrow(0, 2 * mlp, mlp) : none, need_m ? gb_m.narrow(0, 2 * mlp, mlp) : none,
                gw_val, gb_val, gw_out, gb_out, none, none, none, none, none, none, none};
    }
};

at::Tensor module_forward_bf16(const at::Tensor &query, const at::Tensor &centre, const at::Tensor &input_flatten,
                               const c10::optional<at::Tensor> &mask, const at::Tensor &shapes, const at::Tensor &lsi,
                               const at::Tensor &w_off, const at::Tensor &b_off, const at::Tensor &w_attn, const at::Tensor &b_attn,
                               const at::Tensor &w_val, const at::Tensor &b_val, const at::Tensor &w_out, const at::Tensor &b_out,
                               int64_t n_heads, int64_t n_levels, int64_t n_points, int64_t im2col_step, bool deterministic,
                               const c10::optional<at::Tensor> &wm_cached, const c10::optional<at::Tensor> &bm_cached)
{
    for (const at::Tensor *t : {&query, &centre, &w_off, &b_off, &w_attn, &b_attn, &w_val, &b_val, &w_out, &b_out})
        TORCH_CHECK(t->is_cuda() && t->scalar_type() == at::kFloat && t->device() == query.device(),
                    "MSDeformAttn (C++ node, bf16 rows): float32 CUDA tensors on one device expected");
    TORCH_CHECK(input_flatten.is_cuda() && (input_flatten.scalar_type() == at::kFloat || input_flatten.scalar_type() == at::kBFloat16),
                "MSDeformAttn (C++ node, bf16 rows): float32 or bfloat16 input_flatten expected");
    TORCH_CHECK(shapes.is_cuda() && lsi.is_cuda() && shapes.scalar_type() == at::kLong && lsi.scalar_type() == at::kLong,
                "expected scalar type Long for spatial_shapes / level_start_index (on the device)");
    TORCH_CHECK(input_flatten.device() == query.device(), "MSDeformAttn (C++ node, bf16 rows): input_flatten on another device");
    check_module_args(query, centre, input_flatten, mask, shapes, lsi, w_off, b_off, w_attn, b_attn, w_val, b_val, w_out, b_out,
                      n_heads, n_levels, n_points);
    g_grad_mode_at_entry = at::GradMode::is_enabled();
    return MSDAModuleBF16Function::apply(query, centre, input_flatten, mask, shapes, lsi, w_off, b_off, w_attn, b_attn, w_val,
                                         b_val, w_out, b_out, n_heads, n_levels, n_points, im2col_step, deterministic, wm_cached,
                                         bm_cached);
}

at::Tensor module_forward(const at::Tensor &query, const at::Tensor &centre, const at::Tensor &input_flatten,
                          const c10::optional<at::Tensor> &mask, const at::Tensor &shapes, const at::Tensor &lsi,
                          const at::Tensor &w_off, const at::Tensor &b_off, const at::Tensor &w_attn, const at::Tensor &b_attn,
                          const at::Tensor &w_val, const at::Tensor &b_val, const at::Tensor &w_out, const at::Tensor &b_out,
                          int64_t n_heads, int64_t n_levels, int64_t n_points, int64_t im2col_step, bool deterministic,
                          const c10::optional<at::Tensor> &wm_cached, const c10::optional<at::Tensor> &bm_cached)
{
    for (const at::Tensor *t : {&query, &centre, &input_flatten, &w_off, &b_off, &w_attn, &b_attn, &w_val, &b_val, &w_out, &b_out})
        TORCH_CHECK(t->is_cuda() && t->scalar_type() == at::kFloat && t->device() == query.device(),
                    "MSDeformAttn (C++ node): float32 CUDA tensors on one device expected");
    TORCH_CHECK(shapes.is_cuda() && lsi.is_cuda() && shapes.scalar_type() == at::kLong && lsi.scalar_type() == at::kLong,
                "expected scalar type Long for spatial_shapes / level_start_index (on the device)");
    check_module_args(query, centre, input_flatten, mask, shapes, lsi, w_off, b_off, w_attn, b_attn, w_val, b_val, w_out, b_out,
                      n_heads, n_levels, n_points);
    g_grad_mode_at_entry = at::GradMode::is_enabled();
    return MSDAModuleFunction::apply(query, centre, input_flatten, mask, shapes, lsi, w_off, b_off, w_attn, b_attn, w_val, b_val,
                                     w_out, b_out, n_heads, n_levels, n_points, im2col_step, deterministic, wm_cached, bm_cached);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "torch extension over libmsda_hip.so (C ABI include/msda.h)";
    m.def("ms_deform_attn_forward", &forward, "replaces MSDA.ms_deform_attn_forward (vision.cpp:14)");
    m.def("ms_deform_attn_backward", &backward, "replaces MSDA.ms_deform_attn_backward (vision.cpp:15)");
    m.def("apply", &apply, "MSDeformAttnFunction.apply as a C++ autograd node");
    m.def("apply_bf16", &apply_bf16, "MSDeformAttnBF16Function.apply as a C++ autograd node");
#define MSDA_MODULE_ARGS                                                                                                      \
    py::arg("query"), py::arg("centre"), py::arg("input_flatten"), py::arg("mask"), py::arg("spatial_shapes"),                \
        py::arg("level_start_index"), py::arg("w_off"), py::arg("b_off"), py::arg("w_attn"), py::arg("b_attn"), py::arg("w_val"), \
        py::arg("b_val"), py::arg("w_out"), py::arg("b_out"), py::arg("n_heads"), py::arg("n_levels"), py::arg("n_points"),   \
        py::arg("im2col_step"), py::arg("deterministic"), py::arg("wm_cached") = py::none(), py::arg("bm_cached") = py::none()
    m.def("module_forward_bf16", &module_forward_bf16,
          "MSDeformAttn.forward under autocast(bfloat16) with bf16 rows as one C++ autograd node", MSDA_MODULE_ARGS);
    m.def("module_forward", &module_forward, "MSDeformAttn.forward (fp32, fused prologue + merged projection) as one C++ autograd node",
          MSDA_MODULE_ARGS);
#undef MSDA_MODULE_ARGS
    // the header this file was COMPILED against (not the loaded library's msda_version(): _ext.py compares the two)
    m.def("abi_version", [] { return (int)MSDA_ABI_VERSION; });
    m.def("set_exact_nonfinite", [](bool on) { g_extra_flags.store(on ? MSDA_FLAG_EXACT_NONFINITE : 0u); },
          "MSDA_FLAG_EXACT_NONFINITE on every backward the C++ nodes queue (include/msda.h)");
}
