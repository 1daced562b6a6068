#!/bin/bash
# bench.py's N > 1 leg with 2, 4 and 8 gloo ranks sharing cuda:0 (code-path check of the whole line: grids 2x1,
# 2x2, 2x4, cpu_baseline on rank 0 under the launcher's OMP_NUM_THREADS=1, barriers; the timing is not meaningful)
set -o pipefail
mkdir -p gpurun_out
for w in ${WORLDS:-2 4 8}; do
  GPX_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w \
    --master-addr 127.0.0.1 --master-port $((29520 + w)) bench.py --gpus $w --steps 1 --warmup 1 --ntrain 16384 --ntest 1024 \
    --cpu-n 4096 > gpurun_out/dist_gloo$w.log 2>&1 || { tail -20 gpurun_out/dist_gloo$w.log; exit 1; }
  grep '^{' gpurun_out/dist_gloo$w.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('world', d['n_gpus'], d['config']['parallelism'][:40], 'potrf TF', d['value'], 'stages', d['stages_ms'], 'lml', d['parity']['lml'], 'res', d['parity']['residual_Ky_alpha_minus_y_rel'], 'grad0', d['parity']['grad'][:2], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'mem', d['resident_bytes_per_rank_max'])"
done
