#!/bin/bash
# Runs ON THE GPU BOX: what the reference loop's per-step hygiene items cost the 12-layer train step:
#   --vote          barrier() + all_reduce(num_err) + host read          engine.py:564-572
#   --find-unused   DistributedDataParallel(find_unused_parameters=True) main.py:97
#   --reduce-dict   ~120 stacked loss scalars, all_reduce, .item()       engine.py:617-625, util/misc.py:186-192
#   --host-matcher  C.cpu() + scipy linear_sum_assignment in the step    models/matcher.py:120-123
# on ONE rank (the host-synchronisation part needs no second GPU) and with TWO ranks (RCCL if two devices are visible, else
# gloo with both ranks on the one card — then an upper bound on the collectives' share: gloo stages through the host).
set -u
cd $GRAFT_REPO_ROOT
NDEV=$(python3 -c "import torch; print(torch.cuda.device_count())")
BACKEND=nccl; [ "$NDEV" -lt 2 ] && BACKEND=gloo
export HSA_ENABLE_IPC_MODE_LEGACY=0
show() { python3 -c "import json,sys; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-46s ranks %d backend %-5s %.2f ms/step  %.0f frames/s  in sync %s' % ('$1' or '(none)', r['n_gpus'], r['backend'], r['ms_per_step'], r['frames_per_s'], r['params_in_sync']))"; }
python3 tools/ddp_step.py --steps 5 --warmup 3 --window 16 > /dev/null 2>&1      # page the image in: the first run is not a sample
for FL in "" "--reduce-dict" "--host-matcher" "--reduce-dict --host-matcher" ""; do
  python3 tools/ddp_step.py --steps 30 --warmup 5 --window 16 $FL 2>/dev/null | show "$FL"
done
export MSDA_BENCH_BACKEND=$BACKEND
for FL in "" "--vote" "--find-unused" "--reduce-dict" "--host-matcher" "--vote --find-unused --reduce-dict --host-matcher"; do
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/ddp_step.py \
      --steps 30 --warmup 5 --window 16 $FL 2>/dev/null | show "$FL"
done
