#!/bin/bash
# Round 3, GPU session 2: Gram non-temporal stores and trace prefetch A/B; sharded engine with the trapezoid skip.
set -o pipefail
O=gpurun_out/s2
mkdir -p $O
timeout -k 10 600 python -u -m pytest -x -q --timeout 200 --timeout-method thread -m gpu \
  tests/test_gpu_primitives.py -k "colblocks or gram or grad_trace" tests/test_gpu_regression.py tests/test_gpu_dist.py \
  > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for nt in 0 1; do echo "GPX_GRAM_NT=$nt"; GPX_GRAM_NT=$nt timeout -k 10 120 python tools/gram_bench.py 2>&1 | grep -E "gram (lower|full) n" || exit 1; done
for md in 0 1; do echo "GPX_TRACE_MODE=$md"; GPX_TRACE_MODE=$md timeout -k 10 200 python tools/trace_bench.py 2>&1 | grep -E "grad_trace|gram" || exit 1; done
run() {  # tag, block, extra env...
  tag=$1; b=$2; shift 2
  env "$@" GPX_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
    --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 1 --warmup 1 --block $b --no-cpu-baseline \
    > $O/dist_$tag.log 2>&1 || { tail -20 $O/dist_$tag.log; return 1; }
  grep '^{' $O/dist_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['stages_ms'], d['parity']['residual_Ky_alpha_minus_y_rel'])"
}
run trap_b2048 2048 GPX_DIST_GROUP=1 && run trap_b1024 1024 GPX_DIST_GROUP=1 && run trap_b512 512 GPX_DIST_GROUP=1 && \
run trap_b2048_k4096 2048 GPX_UPDATE_K=4096 && run trap_b512_k4096 512 GPX_UPDATE_K=4096 || exit 1
