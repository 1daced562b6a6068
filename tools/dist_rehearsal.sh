#!/bin/bash
# (1) bench.py's N>1 leg with 2 gloo ranks sharing cuda:0 (code-path check, timing not meaningful)
# (2) the same leg on RCCL with world 1 at full size (block-column algorithm on one GPU)
set -o pipefail
mkdir -p gpurun_out
GPX_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --ntrain 16384 --ntest 1024 \
  > gpurun_out/dist_gloo2.log 2>&1 || { tail -20 gpurun_out/dist_gloo2.log; exit 1; }
tail -1 gpurun_out/dist_gloo2.log
for b in ${BLOCKS:-2048}; do
GPX_FORCE_DIST=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
  --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 1 --warmup 1 --block $b \
  > gpurun_out/dist_nccl1_b$b.log 2>&1 || { tail -20 gpurun_out/dist_nccl1_b$b.log; exit 1; }
tail -1 gpurun_out/dist_nccl1_b$b.log
done
