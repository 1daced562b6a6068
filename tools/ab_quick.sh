#!/bin/bash
# dominant launch + block-512 world-1 rehearsal with the current library (development)
set -o pipefail
timeout -k 10 200 python tools/gemm_launch.py 16384 3 2>&1 | grep -i "tflop" | tail -2
GPX_ROW_CHUNK=65536 GPX_FORCE_DIST=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
    --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 1 --warmup 1 --block 512 \
    > gpurun_out/ab.log 2>&1 || { tail -20 gpurun_out/ab.log; exit 1; }
echo "b512: $(tail -1 gpurun_out/ab.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('stages_ms'))")"
