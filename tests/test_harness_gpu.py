"""GPU smoke test of the synthetic train-step harness (tools/ddp_step.py): forward -> loss -> backward ->
clip_grad_norm_(0.1) -> AdamW step through encoder + decoder layers built on the drop-in module, in the
order of the reference's engine.py:590-648.  One process, one GPU; the 2-rank path is rehearsed with
gloo in tests/test_dist.py (harness helpers) and by hand on the GPU box (README of the tool)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_train_step_harness_single_gpu():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ddp_step.py"), "--steps", "2", "--warmup", "1",
                          "--window", "2", "--enc", "2", "--dec", "2", "--queries", "50"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["loss_finite"] and res["n_gpus"] == 1 and res["frames_per_s"] > 0
