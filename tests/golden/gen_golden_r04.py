#!/usr/bin/env python3
"""Round-4 golden fixtures, made by RUNNING THE REFERENCE in the build container (needs /root/reference):

  module_enc_big.npz   the reference MSDeformAttn module (models/ops/modules/ms_deform_attn.py:80-140, CUDA function
                       replaced by the reference's own pure-PyTorch fallback) at ENCODER geometry above the kernels'
                       LDS-stage threshold: d_model 256 / 8 heads, N = 4, pyramid 28/14/7/4 (S = 1045), Lq = S
                       (N*Lq*M = 33 440 >= 32 768 items), module_state weights, a padding mask.  The inputs are NOT
                       stored: tests/golden/big_inputs.py rebuilds them from numpy's frozen legacy generator (the
                       fixture carries checksums of what the reference saw).  Stored: every 4th query / pixel row of
                       out / grad_query / grad_src, per-row sums of ALL rows (fp64), grad_refp and every parameter gradient.
  stack_2d.npz / stack_42d.npz
                       DeformableTransformerEncoder (2 layers) -> DeformableTransformerDecoder (2 layers,
                       return_intermediate, with the per-layer cls / key / obj_key heads attached so that the reference
                       points are refined and handed on between layers) — models/arctic_transformer.py:302-330 and
                       :394-460, class definitions taken out of the file with `ast` and executed unchanged (the file
                       cannot be imported: util.misc needs torchvision) together with util/misc.py's inverse_sigmoid;
                       d_model 64 / 2 heads / ffn 128, dropout 0, a padding mask with per-sample valid ratios.
                       Inputs, state_dicts, hs, inter_references, memory and every input / parameter gradient.

Nothing of the reference's text is stored: the .npz files hold inputs and expected outputs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_r04.py
"""
import ast
import copy
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
REF_MISC = REF + "/util/misc.py"

sys.dont_write_bytecode = True
sys.modules.setdefault("MultiScaleDeformableAttention", types.ModuleType("MultiScaleDeformableAttention"))
sys.path.insert(0, REF + "/models")
sys.path.insert(0, HERE)
from ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch as ref_core   # noqa: E402
import ops.modules.ms_deform_attn as ref_mod                                             # noqa: E402
import big_inputs                                                                        # noqa: E402


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


def module_enc_big():
    z = big_inputs.module_enc_big_inputs()
    state = dict(np.load(os.path.join(HERE, "module_state.npz")))
    mod = ref_mod.MSDeformAttn(d_model=256, n_levels=4, n_heads=8, n_points=4)
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    query = torch.from_numpy(z["query"]).requires_grad_(True)
    src = torch.from_numpy(z["src"]).requires_grad_(True)
    refp = torch.from_numpy(z["refp"]).requires_grad_(True)
    shapes, lsi, mask = torch.from_numpy(z["shapes"]), torch.from_numpy(z["level_start"]), torch.from_numpy(z["mask"])
    gout = torch.from_numpy(z["gout"])
    out = mod(query, refp, src, shapes, lsi, mask)
    out.backward(gout)
    step = big_inputs.ROW_STEP
    arrs = {"pgrad." + k: p.grad for k, p in mod.named_parameters()}
    for name, t in (("out", out), ("grad_query", query.grad), ("grad_src", src.grad)):
        arrs[name + "_rows"] = t.detach()[:, ::step].contiguous()
        arrs[name + "_rowsum"] = t.detach().double().sum(-1)
    save("module_enc_big", grad_refp=refp.grad, input_checksums=big_inputs.checksums(z), **arrs)


def _tree(path):
    return ast.parse(open(path).read())


def stacks():
    tree = _tree(REF_TRANSFORMER)
    wanted = ("DeformableTransformerEncoderLayer", "DeformableTransformerDecoderLayer", "DeformableTransformerEncoder",
              "DeformableTransformerDecoder", "_get_clones", "_get_activation_fn")
    body = [n for n in tree.body if getattr(n, "name", None) in wanted]
    misc = [n for n in _tree(REF_MISC).body if getattr(n, "name", None) == "inverse_sigmoid"]
    ns = {"torch": torch, "nn": nn, "F": F, "MSDeformAttn": ref_mod.MSDeformAttn, "copy": copy}
    exec(compile(ast.Module(body=misc, type_ignores=[]), REF_MISC, "exec"), ns)
    exec(compile(ast.Module(body=body, type_ignores=[]), REF_TRANSFORMER, "exec"), ns)

    d, heads, ffn, n_layers, n_cls = 64, 2, 128, 2, 16          # per-head width 32: the D = 32 kernel family
    shapes = torch.as_tensor([(8, 8), (4, 4), (2, 2), (1, 1)], dtype=torch.long)
    lsi = level_start(shapes)
    S = int(shapes.prod(1).sum())
    N, Lq = 2, 9

    for width in (2, 42):
        g = torch.Generator().manual_seed(41 + width)

        def perturb(mod):
            with torch.no_grad():
                for p in mod.parameters():
                    p.add_(torch.randn(p.shape, generator=g) * 0.05)

        torch.manual_seed(width)
        enc = ns["DeformableTransformerEncoder"](ns["DeformableTransformerEncoderLayer"](d, ffn, 0.0, "relu", 4, heads, 4), n_layers)
        dec = ns["DeformableTransformerDecoder"](ns["DeformableTransformerDecoderLayer"](d, ffn, 0.0, "relu", 4, heads, 4), n_layers,
                                                 return_intermediate=True)
        # the heads the model attaches from outside (models/actic_detr.py): one per decoder layer
        dec.cls_embed = nn.ModuleList(nn.Linear(d, n_cls) for _ in range(n_layers))
        dec.key_embed = nn.ModuleList(nn.Linear(d, width) for _ in range(n_layers))
        dec.obj_key_embed = nn.ModuleList(nn.Linear(d, width) for _ in range(n_layers))
        perturb(enc)
        perturb(dec)
        with torch.no_grad():                                   # make all three branches (class 0, hand 12 / 13, object) occur
            for lin in dec.cls_embed:
                lin.weight.mul_(8.0)
        # padded samples: the valid extent of sample b, as get_valid_ratio reads it off the masks
        masks, ratios = [], []
        for (h, w) in shapes.tolist():
            m = torch.zeros(N, h, w, dtype=torch.bool)
            m[1, :, w - w // 4:] = True
            m[1, h - h // 4:, :] = True
            masks.append(m)
            vh = (~m[:, :, 0]).sum(1).float() / h
            vw = (~m[:, 0, :]).sum(1).float() / w
            ratios.append(torch.stack([vw, vh], -1))
        mask = torch.cat([m.flatten(1) for m in masks], 1)
        valid = torch.stack(ratios, 1)
        src = torch.randn(N, S, d, generator=g, requires_grad=True)
        pos = torch.randn(N, S, d, generator=g, requires_grad=True)
        tgt = torch.randn(N, Lq, d, generator=g, requires_grad=True)
        qpos = torch.randn(N, Lq, d, generator=g, requires_grad=True)
        refp = torch.rand(N, Lq, width, generator=g) * 1.2 - 0.1
        g_hs = torch.randn(n_layers, N, Lq, d, generator=g)

        memory = enc(src, shapes, lsi, valid, pos, mask)
        hs, inter = dec(tgt, refp, memory, shapes, lsi, valid, qpos, mask)
        (hs * g_hs).sum().backward()
        classes = torch.stack([dec.cls_embed[i](hs[i]).argmax(-1) for i in range(n_layers)])
        arrs = {"enc_state." + k: v.clone() for k, v in enc.state_dict().items()}
        arrs.update({"dec_state." + k: v.clone() for k, v in dec.state_dict().items()})
        arrs.update({"enc_pgrad." + k: p.grad for k, p in enc.named_parameters() if p.grad is not None})
        arrs.update({"dec_pgrad." + k: p.grad for k, p in dec.named_parameters() if p.grad is not None})
        save("stack_%dd" % width, src=src, pos=pos, tgt=tgt, qpos=qpos, refp=refp, mask=mask, valid=valid, shapes=shapes,
             level_start=lsi, g_hs=g_hs, memory=memory, hs=hs, inter=inter, classes=classes, grad_src=src.grad,
             grad_pos=pos.grad, grad_tgt=tgt.grad, grad_qpos=qpos.grad, **arrs)
        print("   classes seen:", sorted(set(classes.flatten().tolist())))


if __name__ == "__main__":
    module_enc_big()
    stacks()
