#!/bin/bash
# Round 3, GPU session 1: the new tests, then A/B rehearsals of the sharded engine on RCCL world 1.
set -o pipefail
O=gpurun_out/s1
mkdir -p $O
timeout -k 10 600 python -u -m pytest -x -q --timeout 200 --timeout-method thread -m gpu \
  tests/test_gpu_primitives.py -k "colblocks or gram or non_finite or gemm_layouts" tests/test_gpu_ep.py tests/test_zz_no_alloc.py tests/test_gpu_dist.py \
  > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
run() {  # tag, block, extra env...
  tag=$1; b=$2; shift 2
  env "$@" GPX_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
    --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 1 --warmup 1 --block $b --no-cpu-baseline \
    > $O/dist_$tag.log 2>&1 || { tail -20 $O/dist_$tag.log; return 1; }
  grep '^{' $O/dist_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['stages_ms'], d['parity']['residual_Ky_alpha_minus_y_rel'], d['resident_bytes_per_rank_max'])"
}
run new_b2048 2048 GPX_DIST_GROUP=1 && run old_b2048 2048 GPX_DIST_GROUP=0 GPX_DIST_SIDE_POTRF=1 && \
run new_b512 512 GPX_DIST_GROUP=1 && run old_b512 512 GPX_DIST_GROUP=0 GPX_DIST_SIDE_POTRF=1 && \
run serial_nogroup_b512 512 GPX_DIST_GROUP=0 || exit 1
timeout -k 10 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-public-api > $O/bench_other.log 2> $O/bench_other.err || { tail -20 $O/bench_other.err; exit 1; }
python -c "import json; d=json.loads(open('$O/bench_other.log').read().strip().splitlines()[-1]); print(d['value'], d['stages_ms']); print(json.dumps(d['other_configs'], indent=1))"
