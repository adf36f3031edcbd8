"""Two ranks on the GPU box, each a fresh child process started by torch.distributed.run — BASELINE configs[3]'s
launch path (tools/run_dist_launch.sh:10-29 -> one process per GPU, util/misc.py:519-561 init_process_group,
main.py:96-98 DistributedDataParallel) rehearsed with what a 1-GPU box offers: RCCL when two devices are visible,
otherwise gloo with both ranks sharing the card (MSDA_BENCH_BACKEND=gloo).

  * bench.py --gpus 2: the driver's own N>1 command; n_gpus, per-rank seeds / pids, whole-job value
  * tools/ddp_step.py: DDP's bucketed all-reduce around the op's stream-ordered backward; after three optimizer
    steps every rank must hold bit-identical parameters and a finite loss
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _torchrun(script_args, nproc=2, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    backend = "nccl" if torch.cuda.device_count() >= nproc else "gloo"
    env["MSDA_BENCH_BACKEND"] = backend
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + script_args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, "torchrun failed:\n%s\n%s" % (out.stdout[-2000:], out.stderr[-4000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line from rank 0, got %d" % len(lines)
    return json.loads(lines[0]), backend, out.stderr


def test_bench_two_ranks():
    res, backend, stderr = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                                      "--repeats", "10", "--no-cpu-baseline"])
    assert res["n_gpus"] == 2 and res["steps"] == 20 and res["scaling"] == "weak"
    assert res["config"]["backend"] == backend
    ranks = res["ranks"]
    assert sorted(r["rank"] for r in ranks) == [0, 1]
    assert len({r["pid"] for r in ranks}) == 2                       # two processes
    assert len({r["seed"] for r in ranks}) == 2                      # each rank synthesises its own batch
    if backend == "nccl":
        assert len({r["device"] for r in ranks}) == 2                # one rank per GPU
    # whole-job value = all ranks' samples / the slowest rank's time
    total = sum(r["samples"] for r in ranks)
    assert res["value"] == pytest.approx(total / (res["ms_per_step"] * 1e-3 * res["steps"]), rel=1e-6)
    slowest = max(r["median_block_s"] for r in ranks)
    assert res["value"] <= 1.5 * total / slowest and res["value"] >= 0.5 * total / slowest
    for r in range(2):
        assert ("rank %d/2" % r) in stderr


def test_ddp_train_step_two_ranks_parameters_stay_in_sync():
    res, backend, _ = _torchrun([os.path.join(ROOT, "tools", "ddp_step.py"), "--steps", "3", "--warmup", "1", "--window", "2",
                                 "--enc", "2", "--dec", "2", "--queries", "50", "--ballast-mb", "8"])
    assert res["n_gpus"] == 2 and res["backend"] == backend
    assert res["loss_finite"] and res["params_in_sync"], res["ranks"]
    assert len({r["pid"] for r in res["ranks"]}) == 2
    # every rank saw different frames, so the per-rank losses differ while the parameters do not
    assert res["ranks"][0]["loss"] != res["ranks"][1]["loss"]
    assert res["ranks"][0]["params"] == res["ranks"][1]["params"]
