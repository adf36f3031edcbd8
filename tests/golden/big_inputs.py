"""Inputs of the module_enc_big fixture, rebuilt wherever they are needed instead of being stored (three tensors of 4.3 MB):
numpy's legacy RandomState stream is frozen by numpy's compatibility policy, so gen_golden_r04.py (which ran the reference on
them) and tests/test_module_gpu.py (which runs the HIP path on them) see the same numbers; `checksums` — stored in the
fixture — guards that assumption.  Test infrastructure, no reference code."""
import numpy as np

ROW_STEP = 4                      # the fixture keeps every 4th row of the large outputs (plus per-row sums of all rows)
SHAPES = [(28, 28), (14, 14), (7, 7), (4, 4)]          # BASELINE configs[3] pyramid, S = 1045
N, C, L = 4, 256, 4               # N*Lq*M = 4 * 1045 * 8 = 33 440 items: above the LDS-stage threshold of 32 768


def module_enc_big_inputs():
    rs = np.random.RandomState(20261004)
    shapes = np.asarray(SHAPES, dtype=np.int64)
    S = int(shapes.prod(1).sum())
    lsi = np.concatenate(([0], np.cumsum(shapes.prod(1))[:-1])).astype(np.int64)
    query = rs.standard_normal((N, S, C)).astype(np.float32)
    src = rs.standard_normal((N, S, C)).astype(np.float32)
    gout = rs.standard_normal((N, S, C)).astype(np.float32)
    # encoder-style reference points: every pixel centre of every level, in every level's frame (all ratios 1), jittered
    ref = []
    for h, w in SHAPES:
        ys, xs = np.meshgrid((np.arange(h) + 0.5) / h, (np.arange(w) + 0.5) / w, indexing="ij")
        ref.append(np.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    ref = np.concatenate(ref, 0)[None, :, None, :].repeat(N, 0).repeat(L, 2)
    refp = (ref + rs.uniform(-0.02, 0.02, ref.shape)).astype(np.float32)
    mask = np.zeros((N, S), dtype=bool)
    mask[1, -9:] = True
    mask[3, 100:140] = True
    return dict(query=query, src=src, gout=gout, refp=refp, mask=mask, shapes=shapes, level_start=lsi)


def checksums(z):
    return np.asarray([float(np.float64(z[k]).sum()) for k in ("query", "src", "gout", "refp")] +
                      [float(z["query"][1, 17, 3]), float(z["refp"][2, 500, 1, 0])], dtype=np.float64)
