#!/bin/bash
# world-1 RCCL rehearsal of the block-column path with development overrides from the environment
set -o pipefail
mkdir -p gpurun_out
GPX_FORCE_DIST=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
    --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 1 --warmup 1 --block ${BLOCK:-2048} "$@" \
    > gpurun_out/ab.log 2>&1 || { tail -20 gpurun_out/ab.log; exit 1; }
echo "block ${BLOCK:-2048} grad_block ${GPX_GRAD_BLOCK:-default}: $(tail -1 gpurun_out/ab.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('stages_ms'), d['parity']['grad'][:2])")"
