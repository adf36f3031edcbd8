#!/usr/bin/env python3
"""Round-2 golden fixtures, made by RUNNING THE REFERENCE in the build container (needs /root/reference):

  module_4d.npz      the dn_dab copy of the module (models/dn_dab_dino_deformable_detr/ops/modules/ms_deform_attn.py,
                     4-d (cx, cy, w, h) reference boxes, :105-108) imported as it stands, its CUDA autograd function
                     replaced by the reference's own pure-PyTorch fallback; forward + every gradient
  callers.npz        the argument construction of models/arctic_transformer.py: the flatten block of
                     DeformableTransformer.forward (:157-177) and the per-layer reference points of
                     DeformableTransformerDecoder.forward (:413-419, 2-d and 42-d), statements taken out of the
                     file with `ast` and executed unchanged (the file itself cannot be imported: util.misc needs
                     torchvision)
  layer_encoder.npz / layer_decoder.npz
                     DeformableTransformerEncoderLayer / DecoderLayer (:261-300, :334-391), class definitions
                     executed unchanged with the reference module as their MSDeformAttn; dropout = 0 so that train
                     and eval mode agree; forward + every parameter / input gradient, and the layers' state_dicts

Nothing of the reference's text is stored: the .npz files hold inputs and expected outputs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_r02.py
"""
import ast
import copy
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
REF_TRANSFORMER = REF + "/models/arctic_transformer.py"

sys.dont_write_bytecode = True
sys.modules.setdefault("MultiScaleDeformableAttention", types.ModuleType("MultiScaleDeformableAttention"))
sys.path.insert(0, REF + "/models")
from ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch as ref_core   # noqa: E402
import ops.modules.ms_deform_attn as ref_mod                                             # noqa: E402


class _FallbackFn:
    """Stands in for the CUDA autograd Function inside the reference modules."""
    @staticmethod
    def apply(value, shapes, lsi, loc, attn, im2col_step):
        return ref_core(value, shapes, loc, attn)


ref_mod.MSDeformAttnFunction = _FallbackFn


def save(name, **arrs):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-20s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def level_start(shapes):
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))


def module_4d():
    sys.path.insert(0, REF + "/models/dn_dab_dino_deformable_detr")
    # its package is also called `ops`: load it under another name
    spec_f = importlib.util.spec_from_file_location(
        "dnops.functions.ms_deform_attn_func", REF + "/models/dn_dab_dino_deformable_detr/ops/functions/ms_deform_attn_func.py")
    pkg = types.ModuleType("dnops"); pkg.__path__ = [REF + "/models/dn_dab_dino_deformable_detr/ops"]
    fpk = types.ModuleType("dnops.functions"); fpk.__path__ = [REF + "/models/dn_dab_dino_deformable_detr/ops/functions"]
    sys.modules["dnops"], sys.modules["dnops.functions"] = pkg, fpk
    fmod = importlib.util.module_from_spec(spec_f)
    sys.modules[spec_f.name] = fmod
    spec_f.loader.exec_module(fmod)
    fpk.MSDeformAttnFunction = fmod.MSDeformAttnFunction
    mpk = types.ModuleType("dnops.modules"); mpk.__path__ = [REF + "/models/dn_dab_dino_deformable_detr/ops/modules"]
    sys.modules["dnops.modules"] = mpk
    spec_m = importlib.util.spec_from_file_location(
        "dnops.modules.ms_deform_attn", REF + "/models/dn_dab_dino_deformable_detr/ops/modules/ms_deform_attn.py")
    mmod = importlib.util.module_from_spec(spec_m)
    sys.modules[spec_m.name] = mmod
    spec_m.loader.exec_module(mmod)
    mmod.MSDeformAttnFunction = _FallbackFn

    state = dict(np.load(os.path.join(HERE, "module_state.npz")))
    mod = mmod.MSDeformAttn(d_model=256, n_levels=4, n_heads=8, n_points=4)
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    g = torch.Generator().manual_seed(11)
    shapes = torch.as_tensor([(8, 8), (4, 4), (2, 2), (1, 1)], dtype=torch.long)
    lsi = level_start(shapes)
    S = int(shapes.prod(1).sum())
    N, Lq = 2, 10
    query = torch.randn(N, Lq, 256, generator=g, requires_grad=True)
    src = torch.randn(N, S, 256, generator=g, requires_grad=True)
    centre = torch.rand(N, Lq, 4, 2, generator=g) * 1.2 - 0.1
    wh = torch.rand(N, Lq, 4, 2, generator=g) * 0.5 + 0.05
    refp = torch.cat([centre, wh], -1).requires_grad_(True)
    mask = torch.zeros(N, S, dtype=torch.bool)
    mask[0, 3:9] = True
    gout = torch.randn(N, Lq, 256, generator=g)
    out = mod(query, refp, src, shapes, lsi, mask)
    out.backward(gout)
    arrs = {"pgrad." + k: p.grad for k, p in mod.named_parameters()}
    save("module_4d", query=query, src=src, refp=refp, mask=mask, shapes=shapes, level_start=lsi, gout=gout, out=out,
         grad_query=query.grad, grad_src=src.grad, grad_refp=refp.grad, **arrs)


def _tree():
    return ast.parse(open(REF_TRANSFORMER).read())


def _method(tree, cls, name):
    for c in [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls]:
        for f in [n for n in c.body if isinstance(n, ast.FunctionDef) and n.name == name]:
            return f
    raise KeyError((cls, name))


def callers():
    tree = _tree()
    # ---- the flatten block of DeformableTransformer.forward: from `src_flatten = []` to `valid_ratios = ...` ----
    fwd = _method(tree, "DeformableTransformer", "forward")

    def targets(st):
        return [t.id for t in getattr(st, "targets", []) if isinstance(t, ast.Name)]
    first = next(i for i, st in enumerate(fwd.body) if "src_flatten" in targets(st))
    last = next(i for i, st in enumerate(fwd.body) if "valid_ratios" in targets(st))
    block = ast.Module(body=fwd.body[first:last + 1], type_ignores=[])
    gvr = _method(tree, "DeformableTransformer", "get_valid_ratio")
    ns_fn = {"torch": torch}
    exec(compile(ast.Module(body=[gvr], type_ignores=[]), REF_TRANSFORMER, "exec"), ns_fn)

    g = torch.Generator().manual_seed(21)
    shapes = [(12, 16), (6, 8), (3, 4), (2, 2)]
    N, C = 3, 8
    srcs = [torch.randn(N, C, h, w, generator=g) for h, w in shapes]
    poss = [torch.randn(N, C, h, w, generator=g) for h, w in shapes]
    masks = []
    for (h, w) in shapes:
        m = torch.zeros(N, h, w, dtype=torch.bool)
        for b in range(N):
            vh = int(torch.randint(max(1, h // 2), h + 1, (1,), generator=g))
            vw = int(torch.randint(max(1, w // 2), w + 1, (1,), generator=g))
            m[b, vh:, :] = True
            m[b, :, vw:] = True
        masks.append(m)
    level_embed = torch.randn(len(shapes), C, generator=g)

    class _Self:
        pass
    slf = _Self()
    slf.level_embed = level_embed
    slf.get_valid_ratio = lambda m: ns_fn["get_valid_ratio"](slf, m)
    ns = {"torch": torch, "self": slf, "srcs": srcs, "masks": masks, "pos_embeds": poss}
    exec(compile(block, REF_TRANSFORMER, "exec"), ns)
    arrs = {"shapes_in": np.asarray(shapes, dtype=np.int64), "level_embed": level_embed,
            "src_flatten": ns["src_flatten"], "mask_flatten": ns["mask_flatten"],
            "lvl_pos_embed_flatten": ns["lvl_pos_embed_flatten"], "spatial_shapes": ns["spatial_shapes"],
            "level_start_index": ns["level_start_index"], "valid_ratios": ns["valid_ratios"]}
    for i in range(len(shapes)):
        arrs["src%d" % i], arrs["pos%d" % i], arrs["mask%d" % i] = srcs[i], poss[i], masks[i]

    # ---- the reference-point statement of DeformableTransformerDecoder.forward's layer loop ----
    dfwd = _method(tree, "DeformableTransformerDecoder", "forward")
    loop = next(st for st in dfwd.body if isinstance(st, ast.For))
    stmt = next(st for st in loop.body if isinstance(st, ast.If))
    valid = ns["valid_ratios"]
    for width in (2, 42):
        rp = torch.rand(N, 7, width, generator=g) * 2 - 1
        ns2 = {"torch": torch, "reference_points": rp, "src_valid_ratios": valid}
        exec(compile(ast.Module(body=[stmt], type_ignores=[]), REF_TRANSFORMER, "exec"), ns2)
        arrs["dec_ref%d" % width] = rp
        arrs["dec_ref%d_input" % width] = ns2["reference_points_input"]
    save("callers", **arrs)


def layers():
    tree = _tree()
    wanted = ("DeformableTransformerEncoderLayer", "DeformableTransformerDecoderLayer", "_get_activation_fn")
    body = [n for n in tree.body if getattr(n, "name", None) in wanted]
    ns = {"torch": torch, "nn": nn, "F": F, "MSDeformAttn": ref_mod.MSDeformAttn, "copy": copy}
    exec(compile(ast.Module(body=body, type_ignores=[]), REF_TRANSFORMER, "exec"), ns)

    g = torch.Generator().manual_seed(31)
    shapes = torch.as_tensor([(8, 8), (4, 4), (2, 2), (1, 1)], dtype=torch.long)
    lsi = level_start(shapes)
    S = int(shapes.prod(1).sum())
    N, Lq, d, heads, ffn = 2, 9, 64, 2, 128          # per-head width 32: the D = 32 kernel family

    def perturb(mod):
        with torch.no_grad():
            for p in mod.parameters():
                p.add_(torch.randn(p.shape, generator=g) * 0.05)

    # encoder layer (:261-300): src, pos, reference_points[N,S,L,2], shapes, level_start, padding_mask
    torch.manual_seed(0)
    enc = ns["DeformableTransformerEncoderLayer"](d, ffn, 0.0, "relu", 4, heads, 4)
    perturb(enc)
    src = torch.randn(N, S, d, generator=g, requires_grad=True)
    pos = torch.randn(N, S, d, generator=g, requires_grad=True)
    ref = torch.rand(N, S, 4, 2, generator=g)
    mask = torch.zeros(N, S, dtype=torch.bool)
    mask[1, -5:] = True
    gout = torch.randn(N, S, d, generator=g)
    out = enc(src, pos, ref, shapes, lsi, mask)
    out.backward(gout)
    arrs = {"state." + k: v.clone() for k, v in enc.state_dict().items()}
    arrs.update({"pgrad." + k: p.grad for k, p in enc.named_parameters()})
    save("layer_encoder", src=src, pos=pos, ref=ref, mask=mask, shapes=shapes, level_start=lsi, gout=gout, out=out,
         grad_src=src.grad, grad_pos=pos.grad, **arrs)

    # decoder layer (:334-391): tgt, query_pos, reference_points[N,Lq,L,2|42], src, shapes, level_start, mask
    torch.manual_seed(1)
    dec = ns["DeformableTransformerDecoderLayer"](d, ffn, 0.0, "relu", 4, heads, 4)
    perturb(dec)
    for width in (2, 42):
        tgt = torch.randn(N, Lq, d, generator=g, requires_grad=True)
        qpos = torch.randn(N, Lq, d, generator=g, requires_grad=True)
        memory = torch.randn(N, S, d, generator=g, requires_grad=True)
        ref = torch.rand(N, Lq, 4, width, generator=g) * 1.4 - 0.2
        gout = torch.randn(N, Lq, d, generator=g)
        dec.zero_grad()
        out = dec(tgt, qpos, ref, memory, shapes, lsi, mask)
        out.backward(gout)
        arrs = {"state." + k: v.clone() for k, v in dec.state_dict().items()}
        arrs.update({"pgrad." + k: p.grad for k, p in dec.named_parameters()})
        save("layer_decoder_%dd" % width, tgt=tgt, qpos=qpos, memory=memory, ref=ref, mask=mask, shapes=shapes,
             level_start=lsi, gout=gout, out=out, grad_tgt=tgt.grad, grad_qpos=qpos.grad, grad_memory=memory.grad, **arrs)


if __name__ == "__main__":
    module_4d()
    callers()
    layers()
