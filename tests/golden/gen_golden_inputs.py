#!/usr/bin/env python3
"""Golden fixture for uvhand_amd.utils (the op's argument construction), made by RUNNING THE REFERENCE's own
functions.  models/arctic_transformer.py cannot be imported here (its util.misc needs torchvision, which this
image lacks), so the two self-contained functions are taken out of the reference file with `ast` at generation
time and executed as they stand — nothing of the reference is written into this repository; inputs.npz holds
inputs and expected outputs only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_inputs.py
"""
import ast
import os

import numpy as np
import torch

REF = "/root/reference/models/arctic_transformer.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def reference_functions():
    tree = ast.parse(open(REF).read())
    wanted = {("DeformableTransformer", "get_valid_ratio"), ("DeformableTransformerEncoder", "get_reference_points")}
    out = {}
    for cls in [n for n in tree.body if isinstance(n, ast.ClassDef)]:
        for fn in [n for n in cls.body if isinstance(n, ast.FunctionDef)]:
            if (cls.name, fn.name) in wanted:
                fn.decorator_list = []                                   # staticmethod -> plain function
                ns = {"torch": torch}
                exec(compile(ast.Module(body=[fn], type_ignores=[]), REF, "exec"), ns)
                out[fn.name] = ns[fn.name]
    assert len(out) == 2, out
    return out


def main():
    ref = reference_functions()
    g = torch.Generator().manual_seed(7)
    shapes = [(12, 16), (6, 8), (3, 4), (2, 2)]
    N = 3
    masks = []
    for (h, w) in shapes:                                                # padding on the right / bottom, per sample
        m = torch.zeros(N, h, w, dtype=torch.bool)
        for b in range(N):
            vh = int(torch.randint(max(1, h // 2), h + 1, (1,), generator=g))
            vw = int(torch.randint(max(1, w // 2), w + 1, (1,), generator=g))
            m[b, vh:, :] = True
            m[b, :, vw:] = True
        masks.append(m)
    valid = torch.stack([ref["get_valid_ratio"](None, m) for m in masks], 1)             # (self unused)
    enc_ref = ref["get_reference_points"](shapes, valid, device="cpu")
    arrs = {"shapes": np.asarray(shapes, dtype=np.int64), "valid_ratios": valid.numpy(), "enc_reference_points": enc_ref.numpy()}
    for i, m in enumerate(masks):
        arrs["mask%d" % i] = m.numpy()
    np.savez_compressed(os.path.join(HERE, "inputs.npz"), **arrs)
    print("wrote inputs.npz", {k: v.shape for k, v in arrs.items()})


if __name__ == "__main__":
    main()
