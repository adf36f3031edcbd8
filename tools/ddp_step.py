#!/usr/bin/env python3
"""Synthetic data-parallel TRAIN STEP around the op (SURVEY.md §8d config C4, §8b "harness counterpart").

Not part of the product and not the bench line: a harness that exercises the drop-in module the way the
reference's training loop does — engine.py:590-648: forward -> loss -> backward -> clip_grad_norm_(0.1)
-> AdamW step — under DistributedDataParallel on RCCL (main.py:96-98 uses DDP the same way), so that
the op's stream-ordered, sync-free behaviour under DDP's overlapped gradient all-reduce is covered.

    python tools/ddp_step.py                                   # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           tools/ddp_step.py --ballast-mb 900                 # 8 GPUs, Swin-L-sized gradient volume
    MSDA_BENCH_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 ... tools/ddp_step.py   # rehearsal on 1 GPU

The model is the transformer part of the reference with the backbone replaced by synthetic feature
pyramids: `--enc` encoder layers (MSDeformAttn self-attention over all S pixels + FFN,
models/arctic_transformer.py:261-300) and `--dec` decoder layers (MSDeformAttn cross-attention of
`--queries` queries + FFN, :334-391; the decoder's nn.MultiheadAttention self-attention is included).
`--ballast-mb` adds a parameter block whose gradient volume stands for the Swin-L backbone (≈0.9 GB).
Per rank: batch 1 x window `--window` frames folded into the batch (tempo_inference_dataset.py:112-161),
224x224 crops -> levels 28/14/7/4 (S = 1045), i.e. BASELINE cfg-4.
"""
import argparse
import gc
import json
import os
import sys
import time

import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from uvhand_amd import harness                      # noqa: E402
from uvhand_amd.modules import MSDeformAttn         # noqa: E402
from uvhand_amd.utils import encoder_reference_points  # noqa: E402


def _layer_classes(plain):
    """The drop-in layers of this package; plain=True swaps their fused add+LayerNorm and the MFMA
    weight-gradient path of the FFN back to stock PyTorch ops for an A/B run (the attention module stays this package's)."""
    from uvhand_amd.modules import deformable_layers as dl
    if plain:
        dl.add_layer_norm = lambda x, r, norm: norm(x if r is None else x + r)
        dl.bracket_linear = lambda x, layer: layer(x)
    return dl.DeformableTransformerEncoderLayer, dl.DeformableTransformerDecoderLayer


class SyntheticDeformableStack(nn.Module):
    def __init__(self, enc, dec, queries, ballast_mb, d=256, dropout=0.1, plain_layers=False):
        super().__init__()
        EncoderLayer, DecoderLayer = _layer_classes(plain_layers)
        self.enc = nn.ModuleList(EncoderLayer(d, 1024, dropout) for _ in range(enc))
        self.dec = nn.ModuleList(DecoderLayer(d, 1024, dropout) for _ in range(dec))
        self.query_embed = nn.Embedding(queries, 2 * d)
        self.ref_head = nn.Linear(d, 2)
        self.out_head = nn.Linear(d, 64)
        n = int(ballast_mb * 1e6 / 4)
        self.ballast = nn.Parameter(torch.zeros(n)) if n else None

    def forward(self, src, pos, enc_ref, shapes, lsi, host_matcher=False, n_losses=0):
        for layer in self.enc:
            src = layer(src, pos, enc_ref, shapes, lsi)
        n = src.shape[0]
        qpos, tgt = self.query_embed.weight.chunk(2, dim=-1)
        qpos, tgt = qpos.expand(n, -1, -1), tgt.expand(n, -1, -1)
        ref = self.ref_head(qpos).sigmoid()[:, :, None, :].expand(-1, -1, shapes.shape[0], -1)
        for layer in self.dec:
            tgt = layer(tgt, qpos, ref, src, shapes, lsi)
        out = self.out_head(tgt)                            # [n, queries, 64]
        if host_matcher:
            # f4: the reference's Hungarian matcher on the critical path (models/matcher.py:120-123): the cost matrix goes
            # to the host (a device->host sync in the middle of the step), scipy assigns 3 targets (two hands, object) per
            # frame, and the loss is taken on the matched queries only
            from scipy.optimize import linear_sum_assignment
            tgt_key = torch.linspace(-1, 1, 3 * 64, device=out.device).view(3, 64)
            cost = torch.cdist(out.detach().float().flatten(0, 1), tgt_key, p=1).view(n, out.shape[1], 3).cpu().numpy()
            idx = [linear_sum_assignment(c) for c in cost]
            rows = torch.as_tensor([i for i, _ in idx], dtype=torch.int64, device=out.device)           # [n, 3]
            cols = torch.as_tensor([j for _, j in idx], dtype=torch.int64, device=out.device)
            matched = out.float()[torch.arange(n, device=out.device)[:, None], rows]                     # [n, 3, 64]
            per = (matched - tgt_key[cols]).pow(2)
            loss = per.mean() + 1e-3 * out.float().pow(2).mean()
        else:
            per = out.float().pow(2)
            loss = per.mean()
        if self.ballast is not None:
            loss = loss + 0.0 * self.ballast.sum()          # gives the ballast a (zero) gradient to reduce
        if n_losses:
            # the loss dictionary the reference reduces for logging every step (engine.py:617 -> util/misc.py:186-192):
            # ~120 scalar tensors (every loss term x every decoder layer's auxiliary output)
            flat = per.detach().reshape(-1)
            chunk = max(1, flat.numel() // n_losses)
            return loss, {"loss_%03d" % i: flat[i * chunk:(i + 1) * chunk].mean() for i in range(n_losses)}
        return loss


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--window", type=int, default=32, help="frames per rank folded into the batch")
    ap.add_argument("--queries", type=int, default=300)
    ap.add_argument("--enc", type=int, default=6)
    ap.add_argument("--dec", type=int, default=6)
    ap.add_argument("--ballast-mb", type=float, default=0.0)
    ap.add_argument("--levels", default="28,14,7,4")
    ap.add_argument("--dropout", type=float, default=0.1, help="util/settings.py:113")
    ap.add_argument("--vote", action="store_true",
                    help="f4 A/B: the reference's per-step barrier() + all_reduce(num_err) skip vote (engine.py:564-572)")
    ap.add_argument("--find-unused", action="store_true",
                    help="f4 A/B: DistributedDataParallel(find_unused_parameters=True) as in main.py:97")
    ap.add_argument("--reduce-dict", action="store_true",
                    help="f4 A/B: per-step reduce_dict of ~120 stacked loss scalars + .item() for logging "
                         "(engine.py:617-625, util/misc.py:186-192)")
    ap.add_argument("--host-matcher", action="store_true",
                    help="f4 A/B: Hungarian matching on the host inside the step (models/matcher.py:120-123: C.cpu() + scipy)")
    ap.add_argument("--no-cpp-node", action="store_true",
                    help="A/B: the attention modules as the Python composition of their kernels instead of one C++ autograd node")
    ap.add_argument("--no-gc-freeze", action="store_true",
                    help="A/B: leave the interpreter's collector as it is.  By default everything alive after the warm-up is moved "
                         "to the permanent generation (gc.freeze): a full collection walks the ~10^6 objects `import torch` "
                         "leaves behind in 8-9 ms, and with few Python objects per step (the C++ nodes) those pauses land in "
                         "the clip + optimizer phase, where nothing is queued behind them (profiles/r03_notes.md §8)")
    ap.add_argument("--plain-layers", action="store_true",
                    help="A/B: stock add + LayerNorm and stock FFN weight gradients inside the layers")
    ap.add_argument("--graph", action="store_true",
                    help="capture forward + loss + backward of the whole stack into ONE HIP graph (after the warm-up, gradients left "
                         "in static buffers) and replay it per step; clip_grad_norm_ + AdamW stay outside.  One rank only (the "
                         "gradient all-reduce is not captured), none of the host-side f4 flags")
    ap.add_argument("--amp", default="", choices=["", "bf16"],
                    help="bf16: torch.autocast(bfloat16) around the forward + bf16 row storage in the op (BASELINE config 3)")
    args = ap.parse_args()

    rank, local_rank, world = harness.dist_env()
    backend = os.environ.get("MSDA_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and world > 1 and local_rank >= n_dev:
        sys.exit("ddp_step.py: LOCAL_RANK %d but only %d visible GPU(s): RCCL needs one rank per device" % (local_rank, n_dev))
    dev_index = local_rank % max(1, n_dev)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    distributed = harness.init_process_group(backend, device)
    identities = harness.check_one_rank_per_device(backend if distributed else "none", device)     # raises under nccl
    runtime = harness.collective_runtime()
    print("ddp_step: rank %d/%d backend=%s %s pci %s rccl %s visible devices %d" % (
        rank, world, backend if distributed else "none", identities[rank]["device"], identities[rank]["pci_bus_id"],
        runtime["rccl"], runtime["visible_devices"]), file=sys.stderr, flush=True)

    torch.manual_seed(harness.rank_seed(0, 0))                  # identical initial weights on every rank
    model = SyntheticDeformableStack(args.enc, args.dec, args.queries, args.ballast_mb, dropout=args.dropout,
                                     plain_layers=args.plain_layers).to(device)
    if distributed:
        model = nn.parallel.DistributedDataParallel(model, device_ids=[dev_index] if backend == "nccl" else None,
                                                    gradient_as_bucket_view=True, find_unused_parameters=args.find_unused)
    opt = torch.optim.AdamW(model.parameters(), lr=2e-5, weight_decay=1e-4)

    shapes_list = [(int(x), int(x)) for x in args.levels.split(",")]
    shapes = torch.tensor(shapes_list, dtype=torch.long, device=device)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    g = torch.Generator(device="cpu").manual_seed(harness.rank_seed(100, rank))     # each rank: its own frames
    src = torch.randn(args.window, S, 256, generator=g).to(device)
    pos = torch.randn(args.window, S, 256, generator=g).to(device) * 0.1
    valid_ratios = torch.ones(args.window, len(shapes_list), 2, device=device)         # unpadded crops
    enc_ref = encoder_reference_points(shapes_list, valid_ratios, device)               # [window, S, L, 2]

    for mod in model.modules():
        if isinstance(mod, MSDeformAttn):
            mod.bf16_storage = args.amp == "bf16"
            mod.cpp_node = not args.no_cpp_node

    num_err = torch.zeros(1, dtype=torch.int64, device=device)

    def step():
        if args.vote and distributed:                       # engine.py:564-572: every rank agrees the batch is valid
            torch.distributed.barrier()
            torch.distributed.all_reduce(num_err)
            if int(num_err.item()) > 0:                     # (the host read is part of what the vote costs)
                return torch.zeros((), device=device)
        opt.zero_grad(set_to_none=True)
        kw = dict(host_matcher=args.host_matcher, n_losses=120 if args.reduce_dict else 0)
        if args.amp == "bf16":
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = model(src, pos, enc_ref, shapes, lsi, **kw)
        else:
            loss = model(src, pos, enc_ref, shapes, lsi, **kw)
        if args.reduce_dict:
            loss, loss_dict = loss
            with torch.no_grad():                           # util/misc.py:186-192, then engine.py's .item() for the logger
                values = torch.stack([loss_dict[k] for k in sorted(loss_dict)], 0)
                if distributed:
                    torch.distributed.all_reduce(values)
                    values /= world
                step.logged = float(values.sum().item())
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)
        opt.step()
        return loss

    for _ in range(args.warmup):
        loss = step()
    if args.graph:
        if distributed or args.vote or args.reduce_dict or args.host_matcher:
            sys.exit("ddp_step.py --graph: one rank, without --vote / --reduce-dict / --host-matcher (host work cannot be captured)")
        # whole-network capture (forward, loss, backward); the gradients the capture allocates stay where they are and every
        # replay refreshes them, so the optimizer reads them as usual.  Nothing in the modules or the library synchronises,
        # allocates outside the capture pool or reads spatial_shapes back (tests/test_module_gpu.py captures one module).
        # (PyTorch-ROCm 2.10: ending a capture while the autograd graph of an earlier EAGER step is still alive segfaults in
        # capture_end — drop the warm-up's loss first; uvhand_amd/graphs.py does the same for make_graphed_callables)
        loss = None
        gc.collect()
        torch.cuda.synchronize(device)
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            opt.zero_grad(set_to_none=True)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                if args.amp == "bf16":
                    with torch.autocast("cuda", dtype=torch.bfloat16):
                        static_loss = model(src, pos, enc_ref, shapes, lsi)
                else:
                    static_loss = model(src, pos, enc_ref, shapes, lsi)
                static_loss.backward()
        torch.cuda.current_stream(device).wait_stream(side)

        def step():                                         # noqa: F811 (replaces the eager step for the timed loop)
            graph.replay()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)
            opt.step()
            return static_loss
        for _ in range(2):
            loss = step()
    if not args.no_gc_freeze:
        gc.collect()
        gc.freeze()
    harness.barrier(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    harness.barrier(device)
    elapsed = harness.max_over_ranks(time.perf_counter() - t0, device)
    total = harness.sum_over_ranks(args.window * args.steps, device)
    finite = bool(torch.isfinite(loss.detach()).item())
    # after the optimizer steps every rank must hold bit-identical parameters: that is what DDP's bucketed
    # all-reduce around the op's stream-ordered backward guarantees when it really ran (main.py:96-98)
    with torch.no_grad():
        flat = torch.cat([p.detach().reshape(-1).float() for p in model.parameters()])
        digest = [float(flat.double().sum().item()), float(flat.double().abs().sum().item()),
                  int(flat.view(torch.int32).long().sum().item())]
    digests = harness.gather_objects({"rank": rank, "device": "cuda:%d" % dev_index, "pid": os.getpid(),
                                      "pci_bus_id": identities[rank]["pci_bus_id"],
                                      "loss": float(loss.detach().item()), "params": digest})
    in_sync = all(d["params"] == digests[0]["params"] for d in digests)
    if rank == 0:
        print(json.dumps({"harness": "ddp_step", "n_gpus": world, "frames_per_s": total / elapsed,
                          "ms_per_step": 1e3 * elapsed / args.steps, "steps": args.steps,
                          "per_rank": {"window": args.window, "S": S, "queries": args.queries, "enc": args.enc,
                                       "dec": args.dec, "ballast_mb": args.ballast_mb, "amp": args.amp or None,
                                       "dropout": args.dropout, "layers": "plain" if args.plain_layers else "fused",
                                       "attention_modules": "python" if args.no_cpp_node else "cpp_node",
                                       "launch": "hip graph of forward + backward" if args.graph else "eager",
                                       "gc_frozen": not args.no_gc_freeze,
                                       "vote": args.vote, "find_unused_parameters": args.find_unused,
                                       "reduce_dict": args.reduce_dict, "host_matcher": args.host_matcher},
                          "runtime": runtime,
                          "backend": backend if distributed else None, "loss_finite": finite,
                          "params_in_sync": in_sync, "ranks": digests}))
    if distributed:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
