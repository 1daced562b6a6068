#!/bin/bash
# world-1 RCCL rehearsal of the block-column path at full size: row-piece sizes compared (development)
set -o pipefail
mkdir -p gpurun_out
for cfg in "2048 65536" "2048 16384" "512 65536" "512 8192"; do
  set -- $cfg
  GPX_ROW_CHUNK=$2 GPX_FORCE_DIST=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
    --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 1 --warmup 1 --block $1 \
    > gpurun_out/dist_nccl1_b$1_r$2.log 2>&1 || { tail -20 gpurun_out/dist_nccl1_b$1_r$2.log; exit 1; }
  echo "block $1 row_chunk $2: $(tail -1 gpurun_out/dist_nccl1_b$1_r$2.log | cut -c1-420)"
done
