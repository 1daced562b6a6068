#!/bin/bash
# Round-3 evidence, part A, B or C (separate gpurun calls keep each under its time limit):
#   A: bench.py under rocprofv3 (kernel stats) + the PMC passes (tools/profile_r03.sh), launch-class tables of
#      potrf / trtri / lauum at n = 65536 and n = 16384
#   B: the other BASELINE configs and the HBM-bound kernels, the plain bench.py line
#   C: the N > 1 leg on RCCL with world size 1 (tiles 2048 / 1024 / 512), its launch classes, one rank's share of 2 x 4
# Summaries land in gpurun_out/prof3/ and are copied to profiles/r03_* afterwards.
set -o pipefail
R=$PWD
O=$R/gpurun_out/prof3
mkdir -p $O
part=${1:-A}
if [ "$part" = A ]; then
  bash tools/profile_r03.sh > $O/profile_r03.log 2>&1 || { tail -20 $O/profile_r03.log; exit 1; }
  tail -3 $O/profile_r03.log
  N=65536 bash tools/potrf_trace.sh > $O/ptrace64k.log 2>&1 || { tail -5 $O/ptrace64k.log; exit 1; }
  grep -v '^"' $O/ptrace64k.log | grep -v KERNEL_DISPATCH > $O/r03_potrf_launch_classes_n65536.txt
  N=16384 bash tools/potrf_trace.sh > $O/ptrace16k.log 2>&1 || { tail -5 $O/ptrace16k.log; exit 1; }
  grep -v '^"' $O/ptrace16k.log | grep -v KERNEL_DISPATCH > $O/r03_potrf_launch_classes_n16384.txt
  rm -f $R/gpurun_out/ptrace/pt_kernel_trace.csv
  head -24 $O/r03_potrf_launch_classes_n16384.txt
elif [ "$part" = B ]; then
  {
    echo "== potrf alone (tools/potrf_small_bench.py)"; timeout -k 10 200 python tools/potrf_small_bench.py 2048 4096 8192 16384 32768 65536 2>&1 | grep PANELS
    echo "== C2 / C3 (tools/configs_bench.py)"; timeout -k 10 200 python tools/configs_bench.py 2>&1 | grep -v amdgpu
    echo "== C5 EP (tools/ep_bench.py, tools/ep_fit_time.py)"; timeout -k 10 200 python tools/ep_bench.py 2>&1 | grep -v amdgpu; timeout -k 10 100 python tools/ep_fit_time.py 2>&1 | grep -v amdgpu
    echo "== single right-hand-side solves (tools/trsv_bench.py; block inverses, then GPX_TRSV_BLOCKINV_MIN=0: the recursion)"; timeout -k 10 100 python tools/trsv_bench.py 2>&1 | grep -v amdgpu
    GPX_TRSV_BLOCKINV_MIN=0 timeout -k 10 100 python tools/trsv_bench.py 2>&1 | grep -E "^trsv"
    timeout -k 10 100 python tools/trsv_bench.py 16384 2>&1 | grep -E "^trsv"
    echo "== kernels of one forward and one backward solve at n = 65536 (tools/trsv_trace.sh)"; bash tools/trsv_trace.sh 2>&1 | grep -v amdgpu
    echo "== Gram, grad-trace (tools/gram_bench.py, tools/trace_bench.py)"; timeout -k 10 100 python tools/gram_bench.py 2>&1 | grep -v amdgpu; timeout -k 10 100 python tools/trace_bench.py 2>&1 | grep -E "gram|grad_trace"
    echo "== trtri / lauum in one buffer (tools/trtri_ab.py)"; timeout -k 10 200 python tools/trtri_ab.py 65536 2>&1 | grep -v amdgpu; timeout -k 10 100 python tools/trtri_ab.py 16384 2>&1 | grep -v amdgpu
    echo "== non-PD early return (tools/nonpd_timing.py, n = 8192)"; timeout -k 10 100 python tools/nonpd_timing.py 2>&1 | grep -v amdgpu | tail -5
  } > $O/r03_other_configs.txt 2>&1 || { tail -20 $O/r03_other_configs.txt; exit 1; }
  tail -30 $O/r03_other_configs.txt
  timeout -k 10 600 python bench.py > $O/r03_bench_n65536.json.log 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
  cut -c1-400 $O/r03_bench_n65536.json.log
else
  : > $O/r03_multi_rank_rehearsal_1gpu.jsonl
  for b in 2048 1024 512; do
    GPX_FORCE_DIST=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
      --master-port 29512 bench.py --gpus 1 --steps 1 --warmup 1 --block $b --no-cpu-baseline > $O/dist_w1_b$b.log 2>&1 || { tail -20 $O/dist_w1_b$b.log; exit 1; }
    grep '^{' $O/dist_w1_b$b.log >> $O/r03_multi_rank_rehearsal_1gpu.jsonl
  done
  cut -c1-330 $O/r03_multi_rank_rehearsal_1gpu.jsonl
  BLOCK=2048 bash tools/dist_trace.sh > $O/dtrace_b2048.log 2>&1 || { tail -20 $O/dtrace_b2048.log; exit 1; }
  cp gpurun_out/dtrace_b2048/summary.txt $O/r03_dist_world1_launch_classes.txt
  BLOCK=512 bash tools/dist_trace.sh > $O/dtrace_b512.log 2>&1 || { tail -20 $O/dtrace_b512.log; exit 1; }
  cp gpurun_out/dtrace_b512/summary.txt $O/r03_dist_world1_launch_classes_b512.txt
  timeout -k 10 300 python tools/rank_share.py 65536 512 2 4 2>&1 | grep -v amdgpu > $O/r03_dist_rank_share_2x4.txt || { tail -20 $O/r03_dist_rank_share_2x4.txt; exit 1; }
  tail -3 $O/r03_dist_rank_share_2x4.txt
  GPX_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --ntrain 16384 --ntest 1024 --no-cpu-baseline \
    > $O/dist_gloo2.log 2>&1 || { tail -20 $O/dist_gloo2.log; exit 1; }
  grep '^{' $O/dist_gloo2.log > $O/r03_multi_rank_gloo_rehearsal_1gpu.jsonl; cut -c1-300 $O/r03_multi_rank_gloo_rehearsal_1gpu.jsonl
fi
