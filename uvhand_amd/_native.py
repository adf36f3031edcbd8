"""ctypes binding of ``libmsda_hip.so`` (C ABI: ``include/msda.h``).

This is the only route from Python to the HIP kernels, and there is no other
implementation behind it: if the library is missing or a call fails, this module
raises — it never falls back to PyTorch ops or to anything under ``oracle/``.

It stands where the reference's pybind module ``MultiScaleDeformableAttention``
stands (UVHand models/ops/src/vision.cpp:13-16, imported as ``MSDA`` at
models/ops/functions/ms_deform_attn_func.py:18) and reproduces the host-side
checks of models/ops/src/cuda/ms_deform_attn_cuda.cu:28-52, 93-117 and the
CPU-tensor error of models/ops/src/ms_deform_attn.h:38,60.
"""
import ctypes
import os
import warnings

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libmsda_hip.so")

_lib = None

_SYMBOLS = (
    "msda_forward_f32", "msda_backward_f32", "msda_forward_f64", "msda_backward_f64",
    "msda_forward_bf16", "msda_backward_bf16", "msda_backward_bf16_gv32", "msda_backward_passes",
    "msda_backward_workspace_bytes", "msda_deterministic_supported", "msda_backward_ws_f32", "msda_backward_ws_f64", "msda_backward_ws_bf16",
    "msda_backward_ws_bf16_gv32",
    "msda_backward_prologue_ws_f32", "msda_forward_prologue_bf16", "msda_backward_prologue_bf16_gv32",
    "msda_add_layernorm_workspace_bytes", "msda_add_layernorm_forward_f32", "msda_add_layernorm_backward_f32",
    "msda_relu_dropout_backward_f32", "msda_cast_bf16_multi_f32",
    "msda_flatten_levels_f32", "msda_unflatten_levels_f32", "msda_unflatten_workspace_bytes",
    "msda_linear_wgrad_f32", "msda_linear_wgrad_masked_f32", "msda_linear_wgrad_masked_bf16", "msda_linear_wgrad_multi_f32", "msda_linear_wgrad_multi", "msda_attn32_supported", "msda_attn32_forward_f32",
    "msda_attn32_backward_f32", "msda_linear_wgrad_workspace_bytes",
    "msda_zero_masked_rows_f32", "msda_linear_forward_f32", "msda_linear_dgrad_f32",
    "msda_prologue_supported", "msda_forward_prologue_f32", "msda_backward_prologue_f32",
    "msda_last_error", "msda_version", "msda_path_for", "msda_force_path", "msda_describe_plan",
    "msda_forward_workspace_bytes", "msda_forward_ws_f32", "msda_forward_ws_bf16", "msda_forward_prologue_ws_f32",
    "msda_forward_prologue_ws_bf16", "msda_probe_row_gather", "msda_launch_count",
)


def load():
    """Load (once) and return the ctypes handle; raise if the HIP library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "uvhand_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C uvhand_amd/csrc`). There is no non-HIP fallback for this op." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name in _SYMBOLS:
        if not hasattr(lib, name):
            raise RuntimeError("uvhand_amd: %s does not export %s (stale build?)" % (LIB_PATH, name))
    lib.msda_last_error.restype = ctypes.c_char_p
    lib.msda_version.restype = ctypes.c_int
    lib.msda_path_for.restype = ctypes.c_int
    lib.msda_force_path.restype = None
    lib.msda_prologue_supported.restype = ctypes.c_int
    lib.msda_prologue_supported.argtypes = [ctypes.c_int] * 7
    lib.msda_backward_workspace_bytes.restype = ctypes.c_ulonglong
    lib.msda_backward_workspace_bytes.argtypes = [ctypes.c_int] * 7 + [ctypes.c_uint]
    lib.msda_unflatten_workspace_bytes.restype = ctypes.c_ulonglong
    lib.msda_unflatten_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.msda_add_layernorm_workspace_bytes.restype = ctypes.c_ulonglong
    lib.msda_add_layernorm_workspace_bytes.argtypes = [ctypes.c_longlong, ctypes.c_int]
    lib.msda_linear_wgrad_workspace_bytes.restype = ctypes.c_ulonglong
    lib.msda_linear_wgrad_workspace_bytes.argtypes = [ctypes.c_int] * 3
    lib.msda_forward_workspace_bytes.restype = ctypes.c_ulonglong
    lib.msda_forward_workspace_bytes.argtypes = [ctypes.c_int] * 7 + [ctypes.c_uint]
    lib.msda_launch_count.restype = ctypes.c_ulonglong
    lib.msda_describe_plan.restype = ctypes.c_int
    lib.msda_describe_plan.argtypes = [ctypes.c_int] * 9 + [ctypes.c_uint, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
    _lib = lib
    return lib


def _suffix(dtype):
    if dtype == torch.float32:
        return "f32"
    if dtype == torch.float64:
        return "f64"
    if dtype == torch.bfloat16:
        return "bf16"
    raise RuntimeError('"ms_deform_attn" not implemented for \'%s\'' % str(dtype).replace("torch.", ""))


def _check_inputs(named):
    # order and wording follow ms_deform_attn_cuda.cu:28-38 / :93-105
    value = named[0][1]
    if not value.is_cuda:
        raise RuntimeError("Not implemented on the CPU")          # ms_deform_attn.h:38,60
    dev = value.device
    ok = True
    for _, t in named:
        if not (t.is_contiguous() and t.is_cuda and t.device == dev):
            ok = False
            break
    if ok:
        return
    for name, t in named:
        if not t.is_contiguous():
            raise RuntimeError("%s tensor has to be contiguous" % name)
    for name, t in named:
        if not t.is_cuda:
            raise RuntimeError("%s must be a CUDA tensor" % name)
    for name, t in named:
        if t.device != dev:
            raise RuntimeError("%s must be on the same device as value (%s vs %s)" % (name, t.device, dev))


def _dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    if value.dim() != 4 or sampling_loc.dim() != 6 or attn_weight.dim() != 5:
        raise RuntimeError("ms_deform_attn: expected value[N,S,M,D], sampling_loc[N,Lq,M,L,P,2], "
                           "attn_weight[N,Lq,M,L,P]")
    N, S, M, D = value.shape
    L = spatial_shapes.shape[0]
    Lq, P = sampling_loc.shape[1], sampling_loc.shape[4]
    if tuple(sampling_loc.shape) != (N, Lq, M, L, P, 2) or tuple(attn_weight.shape) != (N, Lq, M, L, P):
        raise RuntimeError("ms_deform_attn: sampling_loc %s / attn_weight %s do not match value %s and %d levels"
                           % (tuple(sampling_loc.shape), tuple(attn_weight.shape), tuple(value.shape), L))
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("expected scalar type Long for spatial_shapes / level_start_index")
    if tuple(spatial_shapes.shape) != (L, 2) or tuple(level_start_index.shape) != (L,):
        raise RuntimeError("ms_deform_attn: spatial_shapes must be [L,2] and level_start_index [L]")
    step = min(N, int(im2col_step))                                # ms_deform_attn_cuda.cu:50-52
    if N > 0 and (step <= 0 or N % step != 0):
        raise RuntimeError("batch(%d) must divide im2col_step(%d)" % (N, step))
    return N, S, M, D, L, Lq, P


def _compute_dtypes(value, sampling_loc, attn_weight):
    """value-like tensors' dtype T and location-like tensors' dtype TL (include/msda.h)."""
    if sampling_loc.dtype != attn_weight.dtype:
        raise RuntimeError("expected sampling_loc and attn_weight to have the same dtype, got %s and %s"
                           % (sampling_loc.dtype, attn_weight.dtype))
    suf = _suffix(value.dtype)
    want_tl = torch.float32 if suf == "bf16" else value.dtype
    if sampling_loc.dtype != want_tl:
        raise RuntimeError("expected scalar type %s but found %s"
                           % (str(want_tl).replace("torch.", "").capitalize(),
                              str(sampling_loc.dtype).replace("torch.", "").capitalize()))
    return suf


# Test hook (tests/test_parity_gpu.py): msda_force_path() is THREAD-LOCAL in the library, so that no caller can flip the
# kernel family under another thread's launch.  A setting made through force_path() below is applied around each native
# call on whichever thread performs it (autograd runs backward on its own device thread) and withdrawn right after, so no
# thread keeps a stale override; a process that never forces a path pays one comparison per call.
_forced_path = -1


class _ForcedPathScope:
    __slots__ = ("lib", "active")

    def __init__(self, lib):
        self.lib, self.active = lib, _forced_path != -1

    def __enter__(self):
        if self.active:
            self.lib.msda_force_path(_forced_path)
        return self

    def __exit__(self, *exc):
        if self.active:
            self.lib.msda_force_path(-1)
        return False


def _raise(lib, rc, what):
    msg = lib.msda_last_error()
    raise RuntimeError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))


_VP, _CI = ctypes.c_void_p, ctypes.c_int
_FWD_ARGTYPES = [_VP] * 5 + [_CI] * 7 + [_VP, _VP]
_BWD_ARGTYPES = [_VP] * 6 + [_CI] * 7 + [_VP] * 4
_BWD_WS_ARGTYPES = [_VP] * 6 + [_CI] * 7 + [_VP] * 4 + [ctypes.c_ulonglong, ctypes.c_uint, _VP]
FLAG_DETERMINISTIC = 1                                # MSDA_FLAG_DETERMINISTIC (include/msda.h)
FLAG_PROLOGUE = 2                                     # MSDA_FLAG_PROLOGUE
FLAG_FORWARD_TABLE = 4                                # MSDA_FLAG_FORWARD_TABLE
FLAG_EXACT_NONFINITE = 8                              # MSDA_FLAG_EXACT_NONFINITE

_extra_flags = 0


def set_exact_nonfinite(on=True):
    """Process-wide: every backward of the D = 32 family keeps the coarse levels on the sort + gather kernels instead of the
    dense matrix-core product (MSDA_FLAG_EXACT_NONFINITE, include/msda.h), so that a non-finite grad_output row makes exactly
    the grad_value rows non-finite that the reference's atomicAdd would (ms_deform_im2col_cuda.cuh:125-152) — the dense product
    spreads it over the whole level of that (batch, head).  Costs the dense levels' speed-up on large problems."""
    global _extra_flags
    _extra_flags = FLAG_EXACT_NONFINITE if on else 0
    from . import _ext
    ext = _ext.get()
    if ext is not None and hasattr(ext, "set_exact_nonfinite"):
        ext.set_exact_nonfinite(bool(on))


def deterministic_requested():
    """True when grad_value should be bitwise reproducible: torch.use_deterministic_algorithms(True) (the reference
    sets cudnn.deterministic, main.py:65-66, which does not cover its atomicAdd scatter) or MSDA_DETERMINISTIC=1."""
    return torch.are_deterministic_algorithms_enabled() or os.environ.get("MSDA_DETERMINISTIC", "0") not in ("", "0")

def _deterministic_for(lib, det, elem_bytes, N, S, M, D, L, Lq, P):
    """The deterministic flag of a backward call: `det`, unless the geometry has no deterministic kernel within the library's
    work bound (msda_deterministic_supported: outside the D = 32 family above 2^36 point tests) AND torch runs with
    use_deterministic_algorithms(True, warn_only=True) — then a warning and the default kernels, as torch does for its own
    ops without a deterministic form.  Without warn_only the library's refusal is raised."""
    if det and torch.is_deterministic_algorithms_warn_only_enabled():
        lib.msda_deterministic_supported.argtypes = [ctypes.c_int] * 8
        if not lib.msda_deterministic_supported(elem_bytes, N, S, M, D, L, Lq, P):
            warnings.warn("ms_deform_attn_backward: no deterministic grad_value kernel for this geometry within the work bound "
                          "(include/msda.h, MSDA_FLAG_DETERMINISTIC); running the default (atomic) kernel [warn_only]")
            return False
    return det


_entry_cache = {}


def _entry(lib, name, argtypes):
    """ctypes function with declared argtypes (plain ints go straight through: the call costs a few us
    of host time, which matters when a kernel takes 5)."""
    fn = _entry_cache.get(name)
    if fn is None:
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _CI
        _entry_cache[name] = fn
    return fn


def _raw_stream(device):
    """hipStream_t of PyTorch's current stream on `device` (launch there: ms_deform_attn_cuda.cu:65,135)."""
    try:
        return torch._C._cuda_getCurrentRawStream(device.index)
    except AttributeError:                                   # older/newer torch without the private hook
        return torch.cuda.current_stream(device).cuda_stream


class _DeviceGuard:
    """No reference counterpart (it relies on torch.cuda.set_device(rank), util/misc.py:550): switch to
    the tensors' device only when it is not already current."""
    __slots__ = ("prev",)

    def __init__(self, device):
        cur = torch.cuda.current_device()
        self.prev = cur if cur != device.index else None
        if self.prev is not None:
            torch.cuda.set_device(device.index)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)
        return False


_FWD_WS_ARGTYPES = [_VP] * 5 + [_CI] * 7 + [_VP, _VP, ctypes.c_ulonglong, _VP]


def _forward_table(lib, N, S, M, D, L, Lq, P, device, prologue=False):
    """The buffer a forward of this geometry can fill with its point table for the backward of the same autograd node
    (msda_forward_workspace_bytes, include/msda.h: the point table of small problems, the per-range point lists of large
    ones); None where the backward's plan reads none."""
    flags = FLAG_PROLOGUE if prologue else 0
    nbytes = int(lib.msda_forward_workspace_bytes(N, S, M, D, L, Lq, P, flags))
    if not nbytes:
        return None
    # the same buffer goes to the backward with MSDA_FLAG_FORWARD_TABLE: the table first, that call's scratch behind it
    nbytes = max(nbytes, int(lib.msda_backward_workspace_bytes(N, S, M, D, L, Lq, P, flags | FLAG_FORWARD_TABLE)))
    return torch.empty((nbytes,), dtype=torch.uint8, device=device)


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step, with_table=None):
    """Replaces MSDA.ms_deform_attn_forward (vision.cpp:14). Returns out[N, Lq, M*D].
    with_table=True / False (the autograd Functions): returns (out, table) — with True the forward also leaves its per-point
    table (msda_forward_ws_*; None where the geometry's backward reads none) to be handed to ms_deform_attn_backward."""
    lib = _lib or load()
    _check_inputs((("value", value), ("spatial_shapes", spatial_shapes),
                   ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
                   ("attn_weight", attn_weight)))
    N, S, M, D, L, Lq, P = _dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
    suf = _compute_dtypes(value, sampling_loc, attn_weight)
    table = None
    with _DeviceGuard(value.device), _ForcedPathScope(lib):
        out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
        if with_table and suf != "f64":
            table = _forward_table(lib, N, S, M, D, L, Lq, P, value.device)
        if table is not None:
            rc = _entry(lib, "msda_forward_ws_" + suf, _FWD_WS_ARGTYPES)(
                value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
                attn_weight.data_ptr(), N, S, M, D, L, Lq, P, out.data_ptr(), table.data_ptr(), table.numel(),
                _raw_stream(value.device))
        else:
            rc = _entry(lib, "msda_forward_" + suf, _FWD_ARGTYPES)(
                value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
                attn_weight.data_ptr(), N, S, M, D, L, Lq, P, out.data_ptr(), _raw_stream(value.device))
    if rc != 0:
        _raise(lib, rc, "ms_deform_attn_forward")
    return out if with_table is None else (out, table)


def _backward_workspace(lib, N, S, M, D, L, Lq, P, device, flags=FLAG_DETERMINISTIC):
    """Scratch a backward call with these flags can use (msda_backward_workspace_bytes, include/msda.h): a stream-ordered
    torch buffer, or (None, 0) when the call needs none.  The caller keeps it alive until the launch is queued;
    the caching allocator only hands the block out again to work queued later on the same stream."""
    nbytes = int(lib.msda_backward_workspace_bytes(N, S, M, D, L, Lq, P, flags))
    if nbytes == 0:
        return None, 0
    return torch.empty((nbytes,), dtype=torch.uint8, device=device), nbytes


def backward_passes(Lq, P):
    """Query chunks the D = 32 backward takes for Lq*P sampling points per (batch, head, level); 1 = single pass."""
    return int((_lib or load()).msda_backward_passes(ctypes.c_int(Lq), ctypes.c_int(P)))


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                            im2col_step, fp32_grad_value=False, deterministic=None, table=None):
    """Replaces MSDA.ms_deform_attn_backward (vision.cpp:15).
    Returns (grad_value, grad_sampling_loc, grad_attn_weight).  fp32_grad_value (bf16 rows only): grad_value
    comes back in float32 (msda_backward_bf16_gv32, include/msda.h; the only bf16 backward outside D = 32).  deterministic (None = deterministic_requested()):
    bitwise reproducible grad_value (MSDA_FLAG_DETERMINISTIC; every kernel family and dtype)."""
    lib = _lib or load()
    _check_inputs((("value", value), ("spatial_shapes", spatial_shapes),
                   ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
                   ("attn_weight", attn_weight), ("grad_output", grad_output)))
    N, S, M, D, L, Lq, P = _dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
    suf = _compute_dtypes(value, sampling_loc, attn_weight)
    if grad_output.dtype != value.dtype or grad_output.numel() != N * Lq * M * D:
        raise RuntimeError("ms_deform_attn_backward: grad_output must be %s[%d,%d,%d]"
                           % (value.dtype, N, Lq, M * D))
    if fp32_grad_value and suf != "bf16":
        raise RuntimeError("fp32_grad_value applies to bfloat16 rows only")
    with _DeviceGuard(value.device), _ForcedPathScope(lib):
        grad_value = torch.empty_like(value, dtype=torch.float32) if fp32_grad_value else torch.empty_like(value)
        grad_loc = torch.empty_like(sampling_loc)
        grad_attn = torch.empty_like(attn_weight)
        det = deterministic_requested() if deterministic is None else bool(deterministic)
        det = _deterministic_for(lib, det, 8 if suf == "f64" else 4, N, S, M, D, L, Lq, P)
        # always through the entry with flags and scratch (msda_backward_workspace_bytes says how much a call can use: 0 for
        # most shapes); the deterministic flag reaches every kernel family and dtype
        flags = (FLAG_DETERMINISTIC if det else 0) | _extra_flags
        if table is not None:                                # the forward's table of this very call (with_table=True), scratch behind it
            ws, nbytes, flags = table, table.numel(), flags | FLAG_FORWARD_TABLE
        else:
            ws, nbytes = _backward_workspace(lib, N, S, M, D, L, Lq, P, value.device, flags)
        rc = _entry(lib, "msda_backward_ws_" + suf + ("_gv32" if fp32_grad_value else ""), _BWD_WS_ARGTYPES)(
            grad_output.data_ptr(), value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
            sampling_loc.data_ptr(), attn_weight.data_ptr(), N, S, M, D, L, Lq, P,
            grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr(),
            ws.data_ptr() if ws is not None else None, nbytes, flags, _raw_stream(value.device))
    if rc != 0:
        _raise(lib, rc, "ms_deform_attn_backward")
    return grad_value, grad_loc, grad_attn


_LL = ctypes.c_longlong
_WGRAD_ARGTYPES = [_VP, _VP, _VP, _CI, _CI, _CI, _VP, _VP, _VP, _VP]


def linear_wgrad_supported(grad_out, inp):
    """fp32 (or both bfloat16), contiguous 2-D views [M, N] / [M, K] on one GPU, N and K multiples of 4."""
    return (grad_out.is_cuda and inp.is_cuda and grad_out.dtype == inp.dtype
            and grad_out.dtype in (torch.float32, torch.bfloat16)
            and grad_out.dim() == 2 and inp.dim() == 2 and grad_out.shape[0] == inp.shape[0]
            and grad_out.is_contiguous() and inp.is_contiguous() and grad_out.device == inp.device
            and grad_out.shape[1] % 4 == 0 and inp.shape[1] % 4 == 0 and inp.shape[0] < (1 << 30)
            and grad_out.data_ptr() % 16 == 0 and inp.data_ptr() % 16 == 0)


def _row_mask_ptr(row_mask, rows, device):
    if row_mask is None:
        return None
    if not (row_mask.dtype == torch.bool and row_mask.is_contiguous() and row_mask.numel() == rows
            and row_mask.device == device):
        raise RuntimeError("row_mask must be a contiguous bool tensor with one entry per row, on the same device")
    return row_mask.data_ptr()


def zero_masked_rows_(x, row_mask):
    """In place: x[r, :] = 0 where row_mask[r] (x: contiguous fp32 [rows, cols], cols % 4 == 0) — include/msda.h."""
    lib = _lib or load()
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous() and x.shape[1] % 4 == 0):
        raise RuntimeError("zero_masked_rows_: expected a contiguous fp32 CUDA matrix with cols % 4 == 0")
    with _DeviceGuard(x.device):
        rc = _entry(lib, "msda_zero_masked_rows_f32", [_VP, _VP, _LL, _CI, _VP])(
            x.data_ptr(), _row_mask_ptr(row_mask, x.shape[0], x.device), x.shape[0], x.shape[1], _raw_stream(x.device))
    if rc != 0:
        _raise(lib, rc, "zero_masked_rows_")
    return x


def linear_wgrad(grad_out, inp, want_bias=True, row_mask=None):
    """(grad_weight[N,K], grad_bias[N] or None) = (grad_out^T @ inp, grad_out.sum(0)) — include/msda.h; float32 results
    also for bfloat16 operands.  Rows of grad_out with row_mask[r] True count as zero."""
    lib = _lib or load()
    if not linear_wgrad_supported(grad_out, inp):
        raise RuntimeError("linear_wgrad: expected contiguous fp32 (or both bf16) CUDA matrices [M,N] and [M,K] with N, K % 4 == 0")
    M, N = grad_out.shape
    K = inp.shape[1]
    with _DeviceGuard(inp.device):
        gw = torch.empty((N, K), dtype=torch.float32, device=inp.device)
        gb = torch.empty((N,), dtype=torch.float32, device=inp.device) if want_bias else None
        nbytes = lib.msda_linear_wgrad_workspace_bytes(M, N, K)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=inp.device) if nbytes else None
        rc = _entry(lib, "msda_linear_wgrad_masked_" + ("bf16" if grad_out.dtype == torch.bfloat16 else "f32"), _WGRAD_ARGTYPES)(
            grad_out.data_ptr(), inp.data_ptr(), _row_mask_ptr(row_mask, M, inp.device), M, N, K, gw.data_ptr(),
            gb.data_ptr() if want_bias else None,
            ws.data_ptr() if ws is not None else None, _raw_stream(inp.device))
    if rc != 0:
        _raise(lib, rc, "linear_wgrad")
    return gw, gb


def linear_rows_supported(a, weight, reduce_dim):
    """fp32 contiguous [rows, reduce] matrix and [out, in] weight on one GPU, both feature counts multiples of 4, 16-byte aligned."""
    return (a.is_cuda and weight.is_cuda and a.device == weight.device and a.dtype == torch.float32 and weight.dtype == torch.float32
            and a.dim() == 2 and weight.dim() == 2 and a.is_contiguous() and weight.is_contiguous()
            and a.shape[1] == weight.shape[reduce_dim] and weight.shape[0] % 4 == 0 and weight.shape[1] % 4 == 0
            and a.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0)


def linear_forward(inp, weight, bias=None, row_mask=None):
    """inp[rows, in] @ weight[out, in].T + bias, rows with row_mask[r] True written as zeros — include/msda.h
    (msda_linear_forward_f32): one fp32-MFMA launch, fixed summation order."""
    lib = _lib or load()
    if not linear_rows_supported(inp, weight, 1) or (bias is not None and not (
            bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == weight.shape[0]
            and bias.device == inp.device and bias.data_ptr() % 16 == 0)):
        raise RuntimeError("linear_forward: expected contiguous fp32 CUDA [rows, in] and [out, in] (+ [out]) with in, out % 4 == 0")
    rows, out_f, in_f = inp.shape[0], weight.shape[0], weight.shape[1]
    with _DeviceGuard(inp.device):
        y = torch.empty((rows, out_f), dtype=torch.float32, device=inp.device)
        rc = _entry(lib, "msda_linear_forward_f32", [_VP, _VP, _VP, _VP, _LL, _CI, _CI, _VP, _VP])(
            inp.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
            _row_mask_ptr(row_mask, rows, inp.device), rows, out_f, in_f, y.data_ptr(), _raw_stream(inp.device))
    if rc != 0:
        _raise(lib, rc, "linear_forward")
    return y


def linear_dgrad(grad_out, weight, row_mask=None):
    """grad_out[rows, out] @ weight[out, in], rows with row_mask[r] True written as zeros — include/msda.h (msda_linear_dgrad_f32)."""
    lib = _lib or load()
    if not linear_rows_supported(grad_out, weight, 0):
        raise RuntimeError("linear_dgrad: expected contiguous fp32 CUDA [rows, out] and [out, in] with in, out % 4 == 0")
    rows, out_f, in_f = grad_out.shape[0], weight.shape[0], weight.shape[1]
    with _DeviceGuard(grad_out.device):
        gx = torch.empty((rows, in_f), dtype=torch.float32, device=grad_out.device)
        rc = _entry(lib, "msda_linear_dgrad_f32", [_VP, _VP, _VP, _LL, _CI, _CI, _VP, _VP])(
            grad_out.data_ptr(), weight.data_ptr(), _row_mask_ptr(row_mask, rows, grad_out.device), rows, out_f, in_f,
            gx.data_ptr(), _raw_stream(grad_out.device))
    if rc != 0:
        _raise(lib, rc, "linear_dgrad")
    return gx


def prologue_supported(value, reference_points, sampling_offsets, attn_logits):
    """True when the fused-prologue entry points (include/msda.h) can take these tensors."""
    if not (value.is_cuda and value.dtype == torch.float32 and sampling_offsets.dtype == torch.float32
            and attn_logits.dtype == torch.float32 and reference_points.dtype == torch.float32):
        return False
    if value.dim() != 4 or sampling_offsets.dim() != 6 or reference_points.dim() != 4 or reference_points.shape[-1] != 2:
        return False
    N, S, M, D = value.shape
    Lq, L, P = sampling_offsets.shape[1], sampling_offsets.shape[3], sampling_offsets.shape[4]
    lib = _lib or load()
    with _ForcedPathScope(lib):
        return bool(lib.msda_prologue_supported(N, S, M, D, L, Lq, P))


def _row_stride(t, name):
    """Floats between consecutive (batch, query) rows of `t` [N, Lq, ...]: the trailing dimensions must be
    dense and the batch stride Lq rows — i.e. `t` is a column block of a row-major [N*Lq, ld] matrix."""
    inner = 1
    for size, stride in zip(reversed(t.shape[2:]), reversed(t.stride()[2:])):
        if size != 1 and stride != inner:
            raise RuntimeError("%s tensor has to be contiguous within a query row" % name)
        inner *= size
    if t.shape[1] > 1:
        ld = t.stride(1)
    elif t.shape[0] > 1:
        ld = t.stride(0)                       # one query per batch element: the rows are the batch elements
    else:
        ld = inner
    if ld < inner or (t.shape[0] > 1 and t.stride(0) != t.shape[1] * ld):
        raise RuntimeError("%s tensor has to be contiguous or a column block of a row-major matrix" % name)
    return ld


_LL = ctypes.c_longlong


def prologue_geometry_supported(N, S, M, D, L, Lq, P):
    """msda_prologue_supported (include/msda.h) for fp32 tensors of these sizes."""
    lib = _lib or load()
    with _ForcedPathScope(lib):
        return bool(lib.msda_prologue_supported(N, S, M, D, L, Lq, P))


def _check_prologue_dtypes(spatial_shapes, level_start_index, **floats):
    """The fused-prologue entry points exist for float32 only (include/msda.h); index tensors are int64."""
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("expected scalar type Long for spatial_shapes / level_start_index")
    for name, t in floats.items():
        if t.dtype != torch.float32:
            raise RuntimeError("expected scalar type Float for %s but found %s"
                               % (name, str(t.dtype).replace("torch.", "").capitalize()))


def ms_deform_attn_forward_prologue(value, spatial_shapes, level_start_index, reference_points, sampling_offsets,
                                    attn_logits, im2col_step, with_table=None):
    """Fused-prologue forward (include/msda.h).  Returns (out, sampling_loc, attn_weight); the last two are
    what the reference's Python would have computed and are what the backward consumes.  `sampling_offsets`
    [N,Lq,M,L,P,2] and `attn_logits` [N,Lq,M,L*P] may be column blocks of one wider projection output."""
    lib = _lib or load()
    _check_inputs((("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                   ("reference_points", reference_points)))
    for name, t in (("sampling_offsets", sampling_offsets), ("attn_logits", attn_logits)):
        if not t.is_cuda or t.device != value.device:
            raise RuntimeError("%s must be a CUDA tensor on the device of value" % name)
    if value.dim() != 4 or sampling_offsets.dim() != 6:
        raise RuntimeError("ms_deform_attn_forward_prologue: expected value[N,S,M,D] and sampling_offsets[N,Lq,M,L,P,2]")
    N, S, M, D = value.shape
    Lq, L, P = sampling_offsets.shape[1], sampling_offsets.shape[3], sampling_offsets.shape[4]
    if (tuple(sampling_offsets.shape) != (N, Lq, M, L, P, 2) or tuple(attn_logits.shape) != (N, Lq, M, L * P)
            or tuple(reference_points.shape) != (N, Lq, L, 2) or tuple(spatial_shapes.shape) != (L, 2)
            or tuple(level_start_index.shape) != (L,)):
        raise RuntimeError("ms_deform_attn_forward_prologue: inconsistent shapes")
    bf16 = value.dtype == torch.bfloat16
    _check_prologue_dtypes(spatial_shapes, level_start_index, reference_points=reference_points,
                           sampling_offsets=sampling_offsets, attn_logits=attn_logits, **({} if bf16 else {"value": value}))
    ld_off, ld_log = _row_stride(sampling_offsets, "sampling_offsets"), _row_stride(attn_logits, "attn_logits")
    if reference_points.data_ptr() % 8:                      # a contiguous view at an odd element offset
        reference_points = reference_points.clone()
    step = min(N, int(im2col_step))
    if N > 0 and (step <= 0 or N % step != 0):
        raise RuntimeError("batch(%d) must divide im2col_step(%d)" % (N, step))
    with _DeviceGuard(value.device), _ForcedPathScope(lib):
        out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
        loc = torch.empty((N, Lq, M, L, P, 2), dtype=torch.float32, device=value.device)
        attn = torch.empty((N, Lq, M, L, P), dtype=torch.float32, device=value.device)
        table = _forward_table(lib, N, S, M, D, L, Lq, P, value.device, prologue=True) if with_table else None
        rc = _entry(lib, "msda_forward_prologue_ws_" + ("bf16" if bf16 else "f32"),
                    [_VP] * 6 + [_CI] * 7 + [_LL] * 2 + [_VP] * 4 + [ctypes.c_ulonglong, _VP])(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), reference_points.data_ptr(),
            sampling_offsets.data_ptr(), attn_logits.data_ptr(), N, S, M, D, L, Lq, P, ld_off, ld_log, out.data_ptr(),
            loc.data_ptr(), attn.data_ptr(), table.data_ptr() if table is not None else None,
            table.numel() if table is not None else 0, _raw_stream(value.device))
    if rc != 0:
        _raise(lib, rc, "ms_deform_attn_forward_prologue")
    return (out, loc, attn) if with_table is None else (out, loc, attn, table)


def ms_deform_attn_backward_prologue(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                                     merged=False, deterministic=None, use_workspace=True, table=None):
    """Returns (grad_value (float32, also for bf16 rows), grad_sampling_offsets, grad_attn_logits[N,Lq,M,L*P],
    grad_reference_points[N,Lq,L,2]).
    merged=True: the two raw gradients are the column blocks [0, 2*M*L*P) and [2*M*L*P, 3*M*L*P) of ONE
    [N, Lq, 3*M*L*P] tensor — the gradient of a merged offsets+logits projection — returned as a fifth value."""
    lib = _lib or load()
    _check_inputs((("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                   ("sampling_loc", sampling_loc), ("attn_weight", attn_weight), ("grad_output", grad_output)))
    # same checks as the plain backward (the kernels reinterpret device memory: a wrong dtype is silent garbage)
    N, S, M, D, L, Lq, P = _dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, max(1, value.shape[0]))
    bf16 = value.dtype == torch.bfloat16
    _check_prologue_dtypes(spatial_shapes, level_start_index, sampling_loc=sampling_loc, attn_weight=attn_weight,
                           **({} if bf16 else {"value": value, "grad_output": grad_output}))
    if bf16 and grad_output.dtype != torch.bfloat16:
        raise RuntimeError("expected scalar type BFloat16 for grad_output (bf16 rows)")
    if grad_output.numel() != N * Lq * M * D:
        raise RuntimeError("ms_deform_attn_backward_prologue: grad_output must be float32[%d,%d,%d]" % (N, Lq, M * D))
    mlp = M * L * P
    with _DeviceGuard(value.device), _ForcedPathScope(lib):
        gv = torch.empty_like(value, dtype=torch.float32)          # fp32 also for bf16 rows (msda_backward_prologue_bf16_gv32)
        if merged:
            both = torch.empty((N, Lq, 3 * mlp), dtype=torch.float32, device=value.device)
            goff, glog = both[..., :2 * mlp].view(N, Lq, M, L, P, 2), both[..., 2 * mlp:].view(N, Lq, M, L * P)
            ld_off = ld_log = 3 * mlp
        else:
            both = None
            goff = torch.empty_like(sampling_loc)
            glog = torch.empty((N, Lq, M, L * P), dtype=torch.float32, device=value.device)
            ld_off = ld_log = 0
        gref = torch.empty((N, Lq, L, 2), dtype=torch.float32, device=value.device)
        det = deterministic_requested() if deterministic is None else bool(deterministic)
        # use_workspace=False (tests): the call a caller without scratch makes — the library then runs the kernels that need none
        flags = (FLAG_DETERMINISTIC if det else 0) | _extra_flags
        if table is not None:                                # the forward's table (with_table=True of the forward), scratch behind it
            ws, flags = table, flags | FLAG_FORWARD_TABLE
            nbytes = table.numel() if use_workspace else min(table.numel(), int(lib.msda_forward_workspace_bytes(N, S, M, D, L, Lq, P, FLAG_PROLOGUE)))
        else:
            ws, nbytes = _backward_workspace(lib, N, S, M, D, L, Lq, P, value.device,
                                             FLAG_PROLOGUE | (FLAG_DETERMINISTIC if det else 0)) if use_workspace else (None, 0)
        rc = _entry(lib, "msda_backward_prologue_bf16_gv32" if bf16 else "msda_backward_prologue_ws_f32",
                    [_VP] * 6 + [_CI] * 7 + [_LL] * 2 + [_VP] * 5 + [ctypes.c_ulonglong, ctypes.c_uint, _VP])(
            grad_output.data_ptr(), value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
            sampling_loc.data_ptr(), attn_weight.data_ptr(), N, S, M, D, L, Lq, P, ld_off, ld_log, gv.data_ptr(),
            goff.data_ptr(), glog.data_ptr(), gref.data_ptr(), ws.data_ptr() if ws is not None else None, nbytes,
            flags, _raw_stream(value.device))
    if rc != 0:
        _raise(lib, rc, "ms_deform_attn_backward_prologue")
    return (gv, goff, glog, gref, both) if merged else (gv, goff, glog, gref)


def relu_dropout_backward_(grad, act, scale):
    """grad *= scale * (act > 0), in place (msda_relu_dropout_backward_f32, include/msda.h): the backward of
    dropout(relu(h)) from act = dropout(relu(h)).  fp32 CUDA tensors of the same size, contiguous, 16-byte aligned."""
    lib = _lib or load()
    n = grad.numel()
    if not (grad.is_cuda and act.is_cuda and grad.dtype == act.dtype == torch.float32 and act.numel() == n
            and grad.is_contiguous() and act.is_contiguous()):
        raise RuntimeError("relu_dropout_backward_: expected two contiguous float32 CUDA tensors of one size")
    with _DeviceGuard(grad.device):
        rc = _entry(lib, "msda_relu_dropout_backward_f32", [_VP, _VP, ctypes.c_float, _LL, _VP])(
            grad.data_ptr(), act.data_ptr(), float(scale), n, _raw_stream(grad.device))
    if rc != 0:
        _raise(lib, rc, "relu_dropout_backward_")
    return grad


def _attn_view(t, name, heads):
    """[L, N, heads*32] float32 CUDA view with the last dimension contiguous -> (pointer, batch stride, sequence stride)."""
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 and t.shape[2] == heads * 32
            and t.stride(2) == 1):
        raise RuntimeError("attn32: %s must be a float32 CUDA tensor [L, N, heads*32] with a contiguous last dimension" % name)
    return t.data_ptr(), t.stride(1), t.stride(0)


def attn32_supported(Lq, Lk, head_dim):
    lib = _lib or load()
    return bool(lib.msda_attn32_supported(int(Lq), int(Lk), int(head_dim)))


def attn32_forward(q, k, v, heads, scale, dropout_p=0.0, seed=None):
    """msda_attn32_forward_f32 (include/msda.h): dropout(softmax(q k^T * scale)) v per (batch, head), head_dim 32.
    q [Lq, N, E], k / v [Lk, N, E] (sequence first, as nn.MultiheadAttention projects them; column-block views allowed);
    seed: a one-element int64 CUDA tensor (required when dropout_p > 0).  Returns (out [Lq, N, E], lse [N*heads, Lq])."""
    lib = _lib or load()
    Lq, N, Lk = q.shape[0], q.shape[1], k.shape[0]
    qv, kv, vv = _attn_view(q, "q", heads), _attn_view(k, "k", heads), _attn_view(v, "v", heads)
    if dropout_p > 0 and not (torch.is_tensor(seed) and seed.is_cuda and seed.dtype == torch.int64 and seed.numel() == 1):
        raise RuntimeError("attn32_forward: dropout needs a one-element int64 CUDA seed tensor")
    with _DeviceGuard(q.device):
        out = torch.empty((Lq, N, heads * 32), dtype=torch.float32, device=q.device)
        lse = torch.empty((N * heads, Lq), dtype=torch.float32, device=q.device)
        ov = _attn_view(out, "out", heads)
        rc = _entry(lib, "msda_attn32_forward_f32", [_VP, _LL, _LL] * 3 + [_CI] * 4 + [ctypes.c_float] * 2 + [_VP] + [_VP, _LL, _LL] + [_VP, _VP])(
            *qv, *kv, *vv, N, heads, Lq, Lk, float(scale), float(dropout_p), seed.data_ptr() if dropout_p > 0 else None, *ov,
            lse.data_ptr(), _raw_stream(q.device))
    if rc != 0:
        _raise(lib, rc, "attn32_forward")
    return out, lse


def attn32_backward(q, k, v, out, lse, grad_out, heads, scale, dropout_p=0.0, seed=None, grad_q=None, grad_k=None, grad_v=None):
    """msda_attn32_backward_f32: gradients of q, k, v (written into grad_q / grad_k / grad_v when given — views like the inputs,
    e.g. the two column blocks of one packed [L, N, 2E] tensor — else into new contiguous tensors)."""
    lib = _lib or load()
    Lq, N, Lk = q.shape[0], q.shape[1], k.shape[0]
    with _DeviceGuard(q.device):
        grad_q = torch.empty((Lq, N, heads * 32), dtype=torch.float32, device=q.device) if grad_q is None else grad_q
        grad_k = torch.empty((Lk, N, heads * 32), dtype=torch.float32, device=q.device) if grad_k is None else grad_k
        grad_v = torch.empty((Lk, N, heads * 32), dtype=torch.float32, device=q.device) if grad_v is None else grad_v
        if grad_out.stride(2) != 1:
            grad_out = grad_out.contiguous()
        views = [_attn_view(t, nm, heads) for t, nm in ((q, "q"), (k, "k"), (v, "v"), (out, "out"))]
        gov = _attn_view(grad_out, "grad_out", heads)
        gviews = [_attn_view(t, nm, heads) for t, nm in ((grad_q, "grad_q"), (grad_k, "grad_k"), (grad_v, "grad_v"))]
        rc = _entry(lib, "msda_attn32_backward_f32", [_VP, _LL, _LL] * 4 + [_VP] + [_VP, _LL, _LL] + [_CI] * 4 + [ctypes.c_float] * 2 + [_VP]
                    + [_VP, _LL, _LL] * 3 + [_VP])(
            *views[0], *views[1], *views[2], *views[3], lse.data_ptr(), *gov, N, heads, Lq, Lk, float(scale), float(dropout_p),
            seed.data_ptr() if dropout_p > 0 else None, *gviews[0], *gviews[1], *gviews[2], _raw_stream(q.device))
    if rc != 0:
        _raise(lib, rc, "attn32_backward")
    return grad_q, grad_k, grad_v


def add_layernorm_supported(x, residual, weight, bias):
    """fp32 CUDA rows of a width the kernels take (multiple of 4, <= 1024), contiguous, 16-byte aligned."""
    d = x.shape[-1] if x.dim() else 0
    ts = [x, weight, bias] + ([residual] if residual is not None else [])
    return (x.dim() >= 1 and d % 4 == 0 and 0 < d <= 1024 and weight is not None and bias is not None
            and tuple(weight.shape) == (d,) and tuple(bias.shape) == (d,)
            and (residual is None or residual.shape == x.shape)
            and all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0
                    and t.device == x.device for t in ts))


def add_layernorm_forward(x, residual, weight, bias, eps):
    """(y, mean, rstd) = LayerNorm(x + residual) — msda_add_layernorm_forward_f32 (include/msda.h)."""
    lib = _lib or load()
    d = x.shape[-1]
    rows = x.numel() // d
    with _DeviceGuard(x.device):
        y = torch.empty_like(x)
        mean = torch.empty((rows,), dtype=torch.float32, device=x.device)
        rstd = torch.empty((rows,), dtype=torch.float32, device=x.device)
        rc = _entry(lib, "msda_add_layernorm_forward_f32", [_VP] * 4 + [_LL, _CI, ctypes.c_float] + [_VP] * 4)(
            x.data_ptr(), residual.data_ptr() if residual is not None else None, weight.data_ptr(), bias.data_ptr(), rows, d,
            float(eps), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _raw_stream(x.device))
    if rc != 0:
        _raise(lib, rc, "add_layernorm_forward")
    return y, mean, rstd


def add_layernorm_backward(grad_y, x, residual, weight, mean, rstd):
    """(grad_sum, grad_weight, grad_bias) — msda_add_layernorm_backward_f32; grad_sum is d/dx and d/dresidual."""
    lib = _lib or load()
    d = x.shape[-1]
    rows = x.numel() // d
    with _DeviceGuard(x.device):
        gs = torch.empty_like(x)
        gw = torch.empty((d,), dtype=torch.float32, device=x.device)
        gb = torch.empty((d,), dtype=torch.float32, device=x.device)
        nbytes = max(16, int(lib.msda_add_layernorm_workspace_bytes(rows, d)))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
        rc = _entry(lib, "msda_add_layernorm_backward_f32", [_VP] * 6 + [_LL, _CI] + [_VP] * 5)(
            grad_y.data_ptr(), x.data_ptr(), residual.data_ptr() if residual is not None else None, weight.data_ptr(),
            mean.data_ptr(), rstd.data_ptr(), rows, d, gs.data_ptr(), gw.data_ptr(), gb.data_ptr(), ws.data_ptr(),
            _raw_stream(x.device))
    if rc != 0:
        _raise(lib, rc, "add_layernorm_backward")
    return gs, gw, gb


def flatten_levels_supported(srcs, poss, level_embed):
    """fp32 contiguous NCHW CUDA feature maps of one batch size / channel count (a multiple of 4), at most 16 levels."""
    ts = list(srcs) + list(poss)
    if not ts or len(srcs) != len(poss) or len(srcs) > 16:
        return False
    n, c = ts[0].shape[0], ts[0].shape[1]
    return (c % 4 == 0 and all(t.dim() == 4 and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
                               and t.shape[0] == n and t.shape[1] == c and t.device == ts[0].device for t in ts)
            and all(s.shape == p.shape for s, p in zip(srcs, poss))
            and level_embed.is_cuda and level_embed.dtype == torch.float32 and level_embed.is_contiguous()
            and tuple(level_embed.shape) == (len(srcs), c) and level_embed.data_ptr() % 16 == 0)


def _level_arrays(tensors):
    L = len(tensors)
    ptrs = (ctypes.c_void_p * L)(*[t.data_ptr() for t in tensors])
    hs = (ctypes.c_int * L)(*[t.shape[2] for t in tensors])
    ws = (ctypes.c_int * L)(*[t.shape[3] for t in tensors])
    return ptrs, hs, ws


def flatten_levels(srcs, poss, level_embed):
    """(src_flatten[N,S,C], lvl_pos_embed_flatten[N,S,C]) — msda_flatten_levels_f32 (include/msda.h)."""
    lib = _lib or load()
    L, (N, C) = len(srcs), srcs[0].shape[:2]
    S = sum(t.shape[2] * t.shape[3] for t in srcs)
    sp, hs, ws = _level_arrays(srcs)
    pp, _, _ = _level_arrays(poss)
    with _DeviceGuard(srcs[0].device):
        src_flat = torch.empty((N, S, C), dtype=torch.float32, device=srcs[0].device)
        pos_flat = torch.empty((N, S, C), dtype=torch.float32, device=srcs[0].device)
        rc = _entry(lib, "msda_flatten_levels_f32", [_CI] + [_VP] * 5 + [_CI, _CI] + [_VP] * 3)(
            L, sp, pp, level_embed.data_ptr(), hs, ws, N, C, src_flat.data_ptr(), pos_flat.data_ptr(),
            _raw_stream(srcs[0].device))
    if rc != 0:
        _raise(lib, rc, "flatten_levels")
    return src_flat, pos_flat


def unflatten_levels(grad_src_flat, grad_pos_flat, shapes_nchw, want_level_embed=False):
    """Per-level NCHW gradients from the flattened ones (either may be None) and, on request, the level-embedding
    gradient [L, C] — msda_unflatten_levels_f32."""
    lib = _lib or load()
    ref = grad_src_flat if grad_src_flat is not None else grad_pos_flat
    N, S, C = ref.shape
    L = len(shapes_nchw)
    with _DeviceGuard(ref.device):
        gs = [torch.empty(sh, dtype=torch.float32, device=ref.device) for sh in shapes_nchw] if grad_src_flat is not None else None
        gp = [torch.empty(sh, dtype=torch.float32, device=ref.device) for sh in shapes_nchw] if grad_pos_flat is not None else None
        hs = (ctypes.c_int * L)(*[sh[2] for sh in shapes_nchw])
        ws_ = (ctypes.c_int * L)(*[sh[3] for sh in shapes_nchw])
        sp = _level_arrays(gs)[0] if gs is not None else None
        pp = _level_arrays(gp)[0] if gp is not None else None
        gembed = ws = None
        if want_level_embed and grad_pos_flat is not None:
            gembed = torch.empty((L, C), dtype=torch.float32, device=ref.device)
            nbytes = max(16, int(lib.msda_unflatten_workspace_bytes(L, hs, ws_, N, C)))
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=ref.device)
        rc = _entry(lib, "msda_unflatten_levels_f32", [_CI] + [_VP] * 4 + [_CI, _CI] + [_VP] * 5)(
            L, sp, pp, hs, ws_, N, C, grad_src_flat.data_ptr() if grad_src_flat is not None else None,
            grad_pos_flat.data_ptr() if grad_pos_flat is not None else None,
            gembed.data_ptr() if gembed is not None else None, ws.data_ptr() if ws is not None else None, _raw_stream(ref.device))
    if rc != 0:
        _raise(lib, rc, "unflatten_levels")
    return gs, gp, gembed


PATH_GENERIC, PATH_D32 = 0, 1          # MSDA_PATH_* (include/msda.h)


def path_for(elem_bytes, M, D, L, P):
    with _ForcedPathScope(load()):
        return int(load().msda_path_for(ctypes.c_int(elem_bytes), ctypes.c_int(M), ctypes.c_int(D),
                                        ctypes.c_int(L), ctypes.c_int(P)))


def describe_plan(N, S, M, D, L, Lq, P, row_bytes=4, grad_value_bytes=None, prologue=False, deterministic=False,
                  has_workspace=True, exact_nonfinite=False):
    """msda_describe_plan (include/msda.h): which kernels / launch plan a call of this geometry takes, as text."""
    lib = _lib or load()
    buf = ctypes.create_string_buffer(512)
    flags = ((FLAG_PROLOGUE if prologue else 0) | (FLAG_DETERMINISTIC if deterministic else 0)
             | (FLAG_EXACT_NONFINITE if exact_nonfinite else 0))
    with _ForcedPathScope(lib):
        lib.msda_describe_plan(row_bytes, grad_value_bytes or row_bytes, N, S, M, D, L, Lq, P, flags, 1 if has_workspace else 0,
                               buf, len(buf))
    return buf.value.decode()


def probe_row_gather(table_bytes, device, row_bytes=128, blocks=4096, iters=16, repeats=5):
    """msda_probe_row_gather (include/msda.h): row requests per second the vector memory path of `device` serves on a table
    of `table_bytes` bytes with the sampling kernels' access pattern (measurement helper for bench.py)."""
    lib = _lib or load()
    table = torch.ones((max(int(table_bytes), row_bytes) // 4,), dtype=torch.float32, device=device)
    sink = torch.zeros((1,), dtype=torch.float32, device=device)
    rows = ctypes.c_ulonglong(0)
    fn = _entry(lib, "msda_probe_row_gather", [_VP, ctypes.c_ulonglong, _CI, _CI, _CI, _VP, ctypes.POINTER(ctypes.c_ulonglong), _VP])
    with _DeviceGuard(device):
        stream = torch.cuda.current_stream(device)
        call = lambda: fn(table.data_ptr(), table.numel() * 4, row_bytes, blocks, iters, sink.data_ptr(), ctypes.byref(rows),
                          _raw_stream(device))
        for _ in range(2):
            if call() != 0:
                _raise(lib, 1, "msda_probe_row_gather")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(repeats):
            call()
        e1.record(stream)
        e1.synchronize()
    return rows.value * repeats / (e0.elapsed_time(e1) * 1e-3)


def launch_count():
    """msda_launch_count (include/msda.h): launches enqueued through the library by this process so far."""
    return int((_lib or load()).msda_launch_count())


def force_path(path):
    """Test hook: -1 = automatic selection, PATH_GENERIC = the generic kernels for every call made through this module."""
    global _forced_path
    load()
    _forced_path = int(path)
