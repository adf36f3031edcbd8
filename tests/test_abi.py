"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/msda.h declares, the Python host layer mirrors the reference's interface and error
behaviour, and the module's parameters / init match the reference's.  No GPU, no compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "msda.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msda_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def native():
    import __graft_entry__
    __graft_entry__.build()
    from uvhand_amd import _native
    _native.load()
    return _native


def test_header_declares_the_expected_entry_points():
    names = _declared_functions()
    for must in ("msda_forward_f32", "msda_backward_f32", "msda_forward_f64", "msda_backward_f64",
                 "msda_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol(native):
    lib = ctypes.CDLL(native.LIB_PATH)
    for name in _declared_functions():
        assert hasattr(lib, name), "libmsda_hip.so does not export %s" % name
    lib.msda_version.restype = ctypes.c_int
    assert lib.msda_version() >= 100
    lib.msda_last_error.restype = ctypes.c_char_p
    assert lib.msda_last_error() == b""


def test_library_has_gfx950_code_object(native):
    blob = open(native.LIB_PATH, "rb").read()
    assert b"gfx950" in blob


def test_path_selection_is_host_logic(native):
    assert native.path_for(4, 8, 32, 4, 4) == 1          # the model shape -> tiled D=32 kernels
    assert native.path_for(8, 8, 32, 4, 4) == 0          # fp64 -> generic
    assert native.path_for(4, 2, 30, 2, 2) == 0
    assert native.path_for(4, 8, 32, 4, 16) == 0         # L*P beyond the record table
    native.force_path(0)
    try:
        assert native.path_for(4, 8, 32, 4, 4) == 0
    finally:
        native.force_path(-1)


def _cpu_inputs(dtype=torch.float32, N=2):
    shapes = torch.tensor([[4, 3], [2, 2]], dtype=torch.long)
    lsi = torch.tensor([0, 12], dtype=torch.long)
    value = torch.rand(N, 16, 2, 4, dtype=dtype)
    loc = torch.rand(N, 5, 2, 2, 3, 2, dtype=dtype)
    attn = torch.rand(N, 5, 2, 2, 3, dtype=dtype)
    return value, shapes, lsi, loc, attn


def test_cpu_tensors_raise_like_the_reference(native):
    """ms_deform_attn.h:38 — AT_ERROR("Not implemented on the CPU"); there is no CPU fallback."""
    from uvhand_amd.functions import MSDeformAttnFunction
    value, shapes, lsi, loc, attn = _cpu_inputs()
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDeformAttnFunction.apply(value, shapes, lsi, loc, attn, 64)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        native.ms_deform_attn_backward(value, shapes, lsi, loc, attn, torch.rand(2, 5, 8), 64)


def test_function_signature_matches_reference():
    import inspect
    from uvhand_amd.functions import MSDeformAttnFunction
    sig = list(inspect.signature(MSDeformAttnFunction.forward).parameters)
    assert sig == ["ctx", "value", "value_spatial_shapes", "value_level_start_index",
                   "sampling_locations", "attention_weights", "im2col_step"]
    assert issubclass(MSDeformAttnFunction, torch.autograd.Function)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "uvhand_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "torch_fallback" not in text and "grid_sample" not in text, f


def test_missing_library_fails_loudly(native, monkeypatch):
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(native, "LIB_PATH", os.path.join(ROOT, "uvhand_amd", "no_such_lib.so"))
    with pytest.raises(RuntimeError, match="no non-HIP fallback"):
        native.load()


# ---------------------------------------------------------------------------------------------
# module surface (models/ops/modules/ms_deform_attn.py:30-78)
# ---------------------------------------------------------------------------------------------
def test_module_surface_and_init_match_reference():
    from uvhand_amd.modules import MSDeformAttn
    torch.manual_seed(0)
    mod = MSDeformAttn(d_model=256, n_levels=4, n_heads=8, n_points=4)
    assert (mod.im2col_step, mod.d_model, mod.n_levels, mod.n_heads, mod.n_points) == (64, 256, 4, 8, 4)
    z = load_golden("module_init")            # reference module constructed under manual_seed(0)
    sd = mod.state_dict()
    assert list(sd.keys()) == ["sampling_offsets.weight", "sampling_offsets.bias",
                               "attention_weights.weight", "attention_weights.bias",
                               "value_proj.weight", "value_proj.bias",
                               "output_proj.weight", "output_proj.bias"]
    assert sorted(sd.keys()) == sorted(z.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == z[k].shape, k
        assert np.array_equal(v.numpy(), z[k]), "init of %s differs from the reference's" % k
    # _reset_parameters is called again from outside (models/arctic_transformer.py:80-81)
    with torch.no_grad():
        mod.sampling_offsets.bias.add_(1.0)
    mod._reset_parameters()
    assert np.array_equal(mod.sampling_offsets.bias.detach().numpy(), z["sampling_offsets.bias"])


def test_module_loads_reference_state_dict_strictly():
    from uvhand_amd.modules import MSDeformAttn
    z = load_golden("module_state")
    mod = MSDeformAttn()
    missing, unexpected = mod.load_state_dict({k: torch.from_numpy(v) for k, v in z.items()}, strict=True)
    assert not missing and not unexpected


def test_projection_parameters_share_one_storage():
    """sampling_offsets / attention_weights stay two nn.Linear (names, state_dict keys, Parameter objects) whose weights and
    biases are views of one buffer each — what the one-node path's merged GEMM reads with no concatenation (VERDICT r04 item
    3a).  The sharing survives optimizer steps, load_state_dict and _reset_parameters, and is re-made after .to() / a
    re-seated parameter."""
    from uvhand_amd.modules import MSDeformAttn
    mod = MSDeformAttn(d_model=32, n_levels=2, n_heads=4, n_points=2)
    n_off = mod.sampling_offsets.weight.shape[0]

    def shared():
        wm, bm = mod._merged_projection_weights()
        return (wm is not None and mod.sampling_offsets.weight.data_ptr() == wm.data_ptr()
                and mod.attention_weights.weight.data_ptr() == wm.data_ptr() + 4 * n_off * 32
                and mod.sampling_offsets.bias.data_ptr() == bm.data_ptr()
                and mod.attention_weights.bias.data_ptr() == bm.data_ptr() + 4 * n_off
                and torch.equal(wm, torch.cat([mod.sampling_offsets.weight, mod.attention_weights.weight]))
                and torch.equal(bm, torch.cat([mod.sampling_offsets.bias, mod.attention_weights.bias])))
    assert shared()
    assert sorted(k for k, _ in mod.named_parameters()) == sorted(
        p + s for p in ("sampling_offsets.", "attention_weights.", "value_proj.", "output_proj.") for s in ("weight", "bias"))
    assert not any("merged" in k for k in mod.state_dict())
    params = {n: p for n, p in mod.named_parameters()}
    opt = torch.optim.AdamW(mod.parameters(), lr=0.1)
    sum(p.sum() for p in mod.parameters()).backward()
    opt.step()
    assert shared() and all(params[n] is p for n, p in mod.named_parameters())
    mod.attention_weights.bias.data.copy_(torch.arange(mod.attention_weights.bias.numel(), dtype=torch.float32))   # no version bump
    assert shared() and mod._merged_projection_weights()[1][n_off + 3] == 3
    mod.load_state_dict(MSDeformAttn(d_model=32, n_levels=2, n_heads=4, n_points=2).state_dict(), strict=True)
    assert shared()
    mod._reset_parameters()
    assert shared()
    mod.double()
    assert mod._merged_projection_weights() == (None, None)          # only float32 layers share storage
    mod.float()
    assert shared()
    mod.sampling_offsets.bias = torch.nn.Parameter(torch.zeros(n_off))    # re-seated from outside: shared again on use
    assert shared() and float(mod._merged_projection_weights()[1][:n_off].abs().sum()) == 0.0
    mod.share_projection_storage = False
    assert mod._merged_projection_weights() == (None, None)


def test_module_argument_errors():
    from uvhand_amd.modules import MSDeformAttn
    with pytest.raises(ValueError, match="divisible"):
        MSDeformAttn(d_model=30, n_heads=4)
    mod = MSDeformAttn(d_model=32, n_levels=2, n_heads=4, n_points=2)
    shapes = torch.tensor([[2, 2], [1, 1]], dtype=torch.long)
    lsi = torch.tensor([0, 4], dtype=torch.long)
    q, src = torch.rand(1, 3, 32), torch.rand(1, 5, 32)
    with pytest.raises(ValueError, match="Last dim of reference_points"):
        mod(q, torch.rand(1, 3, 2, 3), src, shapes, lsi)
    with pytest.raises(AssertionError):
        mod(q, torch.rand(1, 3, 2, 2), torch.rand(1, 6, 32), shapes, lsi)
    # the product path needs the GPU: on CPU tensors it raises, it does not fall back
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        mod(q, torch.rand(1, 3, 2, 2), src, shapes, lsi)


def test_torch_extension_loads_and_matches_the_library(native):
    """uvhand_amd/_msda_torch.so (csrc/torch_ext): built by build(), linked against libmsda_hip.so, same ABI version."""
    from uvhand_amd import _ext
    mod = _ext.get()
    assert mod is not None
    assert mod.abi_version() == native.load().msda_version()
    for name in ("ms_deform_attn_forward", "ms_deform_attn_backward", "apply"):
        assert callable(getattr(mod, name))


def test_argument_errors_are_reported_before_anything_is_launched(native):
    """Bad sizes / null pointers come back as MSDA_ERR_ARGUMENT with a message — no GPU needed, nothing is enqueued."""
    lib = ctypes.CDLL(native.LIB_PATH)
    lib.msda_last_error.restype = ctypes.c_char_p
    V, I, LL = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
    # residual add + LayerNorm: row width must be a multiple of 4, at most 1024
    fn = lib.msda_add_layernorm_forward_f32
    fn.argtypes = [V, V, V, V, LL, I, ctypes.c_float, V, V, V, V]
    fn.restype = I
    assert fn(None, None, None, None, 4, 6, 1e-5, None, None, None, None) == 1
    assert b"multiple of 4" in lib.msda_last_error()
    assert fn(None, None, None, None, 4, 2048, 1e-5, None, None, None, None) == 1
    assert fn(None, None, None, None, 4, 256, 1e-5, None, None, None, None) == 1        # null pointers with rows > 0
    # the op with flags / scratch: null tensors with a non-empty problem
    fn = lib.msda_backward_ws_f32
    fn.argtypes = [V] * 6 + [I] * 7 + [V] * 4 + [ctypes.c_ulonglong, ctypes.c_uint, V]
    fn.restype = I
    assert fn(None, None, None, None, None, None, 1, 4, 8, 32, 1, 2, 4, None, None, None, None, 0, 1, None) == 1
    assert b"null device pointer" in lib.msda_last_error()
    assert fn(None, None, None, None, None, None, 1, 4, 0, 32, 1, 2, 4, None, None, None, None, 0, 1, None) == 1   # M = 0
    # pyramid flatten: level count and channel width
    fn = lib.msda_flatten_levels_f32
    fn.argtypes = [I, V, V, V, V, V, I, I, V, V, V]
    fn.restype = I
    hs, ws = (I * 1)(4), (I * 1)(4)
    assert fn(0, None, None, None, hs, ws, 1, 8, None, None, None) == 1
    assert fn(1, None, None, None, hs, ws, 1, 6, None, None, None) == 1
    assert fn(17, None, None, None, hs, ws, 1, 8, None, None, None) == 1
    # sizes only (no pointers, no launch)
    lib.msda_backward_workspace_bytes.restype = ctypes.c_ulonglong
    lib.msda_backward_workspace_bytes.argtypes = [I] * 7 + [ctypes.c_uint]
    assert lib.msda_backward_workspace_bytes(2, 3060, 8, 32, 4, 300, 4, 1) == 0           # deterministic: counters in LDS
    assert lib.msda_backward_workspace_bytes(2, 3060, 8, 32, 4, 3060, 4, 1) == 0
    assert lib.msda_backward_workspace_bytes(2, 3060, 8, 32, 4, 3060, 4, 0) == 0           # default kernels need none
    # MSDA_FLAG_PROLOGUE (2): large problems keep per-head reference-point gradients [N, Lq, M, L, 2] in scratch
    assert lib.msda_backward_workspace_bytes(2, 3060, 8, 32, 4, 3060, 4, 2) == 2 * 3060 * 8 * 4 * 8
    assert lib.msda_backward_workspace_bytes(2, 3060, 8, 32, 4, 300, 4, 2) == 0            # small problem: the fused launch
    # the decoder self-attention core: head_dim 32, at most 320 queries / keys, 0 <= p < 1 with a seed, aligned views
    assert lib.msda_attn32_supported(300, 300, 32) == 1 and lib.msda_attn32_supported(321, 300, 32) == 0
    assert lib.msda_attn32_supported(300, 0, 32) == 0 and lib.msda_attn32_supported(300, 300, 64) == 0
    fn = lib.msda_attn32_forward_f32
    fn.argtypes = [V, LL, LL] * 3 + [I] * 4 + [ctypes.c_float] * 2 + [V] + [V, LL, LL] + [V, V]
    fn.restype = I
    assert fn(*([None, 256, 256] * 3), 2, 8, 400, 400, 0.17, 0.0, None, None, 256, 256, None, None) == 1      # too long
    assert b"320" in lib.msda_last_error()
    assert fn(*([None, 256, 256] * 3), 2, 8, 300, 300, 0.17, 0.1, None, None, 256, 256, None, None) == 1      # dropout without a seed
    assert fn(*([None, 256, 256] * 3), 2, 8, 300, 300, 0.17, 1.0, None, None, 256, 256, None, None) == 1      # p = 1
    assert fn(*([None, 256, 256] * 3), 2, 8, 300, 300, 0.17, 0.0, None, None, 256, 256, None, None) == 1      # null tensors
    assert b"aligned" in lib.msda_last_error()
    lib.msda_unflatten_workspace_bytes.restype = ctypes.c_ulonglong
    lib.msda_unflatten_workspace_bytes.argtypes = [I, V, V, I, I]
    hs4, ws4 = (I * 4)(28, 14, 7, 4), (I * 4)(28, 14, 7, 4)
    assert lib.msda_unflatten_workspace_bytes(4, hs4, ws4, 32, 256) == 32 * (13 + 4 + 1 + 1) * 4 * 64 * 4


def test_deterministic_work_bound_is_host_logic(native):
    """outside the D = 32 family MSDA_FLAG_DETERMINISTIC is a brute force bounded at N*S*M x Lq*P <= 2^36 point
    tests; msda_deterministic_supported says on which side a geometry falls (hosts under warn_only=True ask, warn, and run the
    default kernels instead of raising)."""
    lib = native._lib
    I = ctypes.c_int
    lib.msda_deterministic_supported.restype = I
    lib.msda_deterministic_supported.argtypes = [I] * 8
    assert lib.msda_deterministic_supported(4, 2, 3060, 8, 32, 4, 3060, 4) == 1           # D = 32 family: always
    assert lib.msda_deterministic_supported(4, 32, 1045, 8, 32, 4, 1045, 4) == 1
    # fp64 (generic family): 4 * 65536 * 8 rows x 32768 * 1 points = 2^36 exactly -> allowed; one more query -> refused
    assert lib.msda_deterministic_supported(8, 4, 65536, 8, 32, 1, 32768, 1) == 1
    assert lib.msda_deterministic_supported(8, 4, 65536, 8, 32, 1, 32769, 1) == 0
    assert lib.msda_deterministic_supported(4, 4, 65536, 8, 30, 1, 32769, 1) == 0           # D = 30: generic as well
    assert lib.msda_deterministic_supported(4, 4, 65536, 8, 32, 1, 32769, 1) == 1           # the same sizes at D = 32
    assert lib.msda_deterministic_supported(8, 1, 30, 2, 4, 2, 2, 2) == 1                 # test.py's gradcheck shape
    assert lib.msda_deterministic_supported(8, 0, 30, 2, 4, 2, 2, 2) == 1                 # empty: nothing to order


def test_warn_only_clears_the_flag_for_geometries_without_a_deterministic_kernel(native):
    import warnings
    big = (8, 4, 65536, 8, 32, 1, 32769, 1)
    assert native._deterministic_for(native._lib, True, *big) is True                     # not warn_only: the library refuses later
    prev = torch.are_deterministic_algorithms_enabled(), torch.is_deterministic_algorithms_warn_only_enabled()
    torch.use_deterministic_algorithms(True, warn_only=True)
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert native._deterministic_for(native._lib, True, *big) is False
            assert native._deterministic_for(native._lib, True, 8, 4, 65536, 8, 32, 1, 32768, 1) is True
            assert native._deterministic_for(native._lib, False, *big) is False
        assert len(w) == 1 and "warn_only" in str(w[0].message)
    finally:
        torch.use_deterministic_algorithms(prev[0], warn_only=prev[1])


def test_a_backward_workspace_with_the_forward_table_is_table_then_scratch(native):
    """MSDA_FLAG_FORWARD_TABLE: the forward's buffer first (rounded up to 256 bytes), the call's own scratch behind it."""
    lib = native._lib
    I = ctypes.c_int
    lib.msda_backward_workspace_bytes.restype = ctypes.c_ulonglong
    lib.msda_backward_workspace_bytes.argtypes = [I] * 7 + [ctypes.c_uint]
    lib.msda_forward_workspace_bytes.restype = ctypes.c_ulonglong
    lib.msda_forward_workspace_bytes.argtypes = [I] * 7 + [ctypes.c_uint]
    enc = (2, 3060, 8, 32, 4, 3060, 4)                                                    # cfg-2 encoder: range masks + per-head scratch
    masks = lib.msda_forward_workspace_bytes(*enc, 2)
    heads = lib.msda_backward_workspace_bytes(*enc, 2)
    assert masks == 64 + 2 * 8 * 4 * 3060 * 4 and heads == 2 * 3060 * 8 * 4 * 8
    assert lib.msda_backward_workspace_bytes(*enc, 2 | 4) == ((masks + 255) & ~255) + heads
    assert lib.msda_backward_workspace_bytes(*enc, 4) == lib.msda_forward_workspace_bytes(*enc, 0) == masks   # no scratch: the table alone
    dec = (2, 3060, 8, 32, 4, 300, 4)                                                     # cfg-2 decoder: point table, no scratch
    table = lib.msda_forward_workspace_bytes(*dec, 2)
    assert table > 2 * 8 * 4 * 300 * 4 * 16 and lib.msda_backward_workspace_bytes(*dec, 2 | 4) == table
    assert lib.msda_forward_workspace_bytes(32, 1045, 8, 32, 4, 300, 4, 0) == 0           # cfg-4 decoder: one range per level, no masks
    assert lib.msda_backward_workspace_bytes(32, 1045, 8, 32, 4, 300, 4, 4) == 0


def test_range_masks_are_planned_from_four_ranges_per_level_on(native):
    """plan_masks (uvhand_amd/csrc/msda_d32.hip) is host logic: the forward of a large problem leaves per-point range masks exactly
    where the backward's kept-taps pass cuts the levels into 4..8 ranges — never for single-pass plans, small problems, the
    deterministic flag, or two / three ranges (measured: the forward's bytes cost more than those scans, profiles/r05_notes.md)."""
    lib = native._lib
    lib.msda_forward_workspace_bytes.restype = ctypes.c_ulonglong
    lib.msda_forward_workspace_bytes.argtypes = [ctypes.c_int] * 7 + [ctypes.c_uint]
    cases = [  # (N, shapes, Lq, W in the plan text, masks?)
        (2, [(48, 48), (24, 24), (12, 12), (6, 6)], 3060, "W=6", True),           # cfg-2 encoder
        (4, [(40, 40), (20, 20), (10, 10)], 2100, "W=4", True),
        (4, [(32, 32), (16, 16), (8, 8), (4, 4)], 1600, "W=3", False),
        (32, [(28, 28), (14, 14), (7, 7), (4, 4)], 1045, "W=2", False),           # cfg-4 encoder
        (32, [(28, 28), (14, 14), (7, 7), (4, 4)], 300, "W=1", False),            # cfg-4 decoder: one pass
    ]
    for N, shapes, Lq, w, masks in cases:
        S, L = sum(h * x for h, x in shapes), len(shapes)
        plan = native.describe_plan(N, S, 8, 32, L, Lq, 4)
        assert "bwd=fused_lds(" in plan and w + "," in plan, plan
        assert ("masks" in plan) == masks, plan
        assert lib.msda_forward_workspace_bytes(N, S, 8, 32, L, Lq, 4, 0) == (64 + N * 8 * L * Lq * 4 if masks else 0)
        assert "masks" not in native.describe_plan(N, S, 8, 32, L, Lq, 4, deterministic=True)


def test_int32_row_offsets_bound_the_d32_family():
    """Role B's gathers read a (batch, head) pair's grad_out rows through a buffer descriptor with 32-bit BYTE offsets
    q * M * 128: the tiled family only takes geometries with Lq * M * 128 B < 2^31 (others go to the generic kernels, whose
    indices are 64-bit).  Host logic only: no launch."""
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, "uvhand_amd", "libmsda_hip.so"))
    lib.msda_prologue_supported.restype = ctypes.c_int
    lib.msda_prologue_supported.argtypes = [ctypes.c_int] * 7
    # N = 1, M = 8, L = 1, P = 4: every other limit of the family (items * 32 < 2^31, items*L*P*2 < 2^31) still holds at Lq = 2^21
    assert lib.msda_prologue_supported(1, 64, 8, 32, 1, (1 << 21) - 1, 4) == 1          # 2^31 - 1024 bytes of a pair's rows
    assert lib.msda_prologue_supported(1, 64, 8, 32, 1, 1 << 21, 4) == 0               # exactly 2^31
