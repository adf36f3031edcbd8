#!/bin/bash
# Runs ON THE GPU BOX: what the reference loop's per-step vote and find_unused_parameters cost the 12-layer train step
# with TWO ranks (RCCL if two devices are visible, else gloo with both ranks on the one card — then an upper bound on
# the collective's share: gloo stages through the host).
set -u
cd $GRAFT_REPO_ROOT
NDEV=$(python3 -c "import torch; print(torch.cuda.device_count())")
BACKEND=nccl; [ "$NDEV" -lt 2 ] && BACKEND=gloo
export MSDA_BENCH_BACKEND=$BACKEND HSA_ENABLE_IPC_MODE_LEGACY=0
for FL in "" "--vote" "--find-unused" "--vote --find-unused"; do
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/ddp_step.py \
      --steps 10 --warmup 3 --window 16 $FL 2>/dev/null | python3 -c "import json,sys; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-24s backend %s: %.2f ms/step, in sync %s' % ('$FL' or '(none)', r['backend'], r['ms_per_step'], r['params_in_sync']))"
done
