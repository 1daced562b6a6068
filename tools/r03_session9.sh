#!/bin/bash
set -o pipefail
O=gpurun_out/s9
mkdir -p $O
timeout -k 10 900 python -u -m pytest -x -q --timeout 300 --timeout-method thread -m gpu tests/test_gpu_primitives.py -k "trmm_right" tests/test_gpu_dist.py > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
run() {  # tag, block, extra env...
  tag=$1; b=$2; shift 2
  env "$@" GPX_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
    --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 1 --warmup 1 --block $b --no-cpu-baseline \
    > $O/dist_$tag.log 2>&1 || { tail -20 $O/dist_$tag.log; return 1; }
  grep '^{' $O/dist_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['stages_ms'], d['parity']['residual_Ky_alpha_minus_y_rel'], d['resident_bytes_per_rank_max'])"
}
run inv_b2048 2048 A=1 && run noinv_b2048 2048 GPX_DIST_PANEL_INV=0 && run inv_b512 512 A=1 && run noinv_b512 512 GPX_DIST_PANEL_INV=0 && run inv_b1024 1024 A=1 || exit 1
timeout -k 10 300 python tools/rank_share.py 65536 512 2 4 > $O/rank_share_2x4.txt 2>&1 || { tail -20 $O/rank_share_2x4.txt; exit 1; }
grep -v amdgpu $O/rank_share_2x4.txt
