#!/bin/bash
# Round-2 evidence, part A or B (two gpurun calls keep each under its time limit):
#   A: bench.py under rocprofv3 (kernel stats) + the PMC passes (tools/profile_r02.sh), launch-class tables of
#      potrf / trtri / lauum at n = 65536 and n = 16384
#   B: the other BASELINE configs and the HBM-bound kernels, the plain bench.py line, the N > 1 leg on RCCL with
#      world size 1
# Summaries land in gpurun_out/prof2/ and are copied to profiles/r02_* afterwards.
set -o pipefail
R=$PWD
O=$R/gpurun_out/prof2
mkdir -p $O
part=${1:-A}
if [ "$part" = A ]; then
  bash tools/profile_r02.sh > $O/profile_r02.log 2>&1 || { tail -20 $O/profile_r02.log; exit 1; }
  tail -3 $O/profile_r02.log
  N=65536 bash tools/potrf_trace.sh > $O/ptrace64k.log 2>&1 || { tail -5 $O/ptrace64k.log; exit 1; }
  grep -v '^"' $O/ptrace64k.log | grep -v KERNEL_DISPATCH > $O/r02_potrf_launch_classes_n65536.txt
  N=16384 bash tools/potrf_trace.sh > $O/ptrace16k.log 2>&1 || { tail -5 $O/ptrace16k.log; exit 1; }
  grep -v '^"' $O/ptrace16k.log | grep -v KERNEL_DISPATCH > $O/r02_potrf_launch_classes_n16384.txt
  rm -f $R/gpurun_out/ptrace/pt_kernel_trace.csv
  head -20 $O/r02_potrf_launch_classes_n16384.txt
else
  {
    echo "== potrf alone (tools/potrf_small_bench.py)"; timeout -k 10 200 python tools/potrf_small_bench.py 2048 4096 8192 16384 32768 65536 2>&1 | grep PANELS
    echo "== C2 / C3 (tools/configs_bench.py)"; timeout -k 10 200 python tools/configs_bench.py 2>&1 | grep -v amdgpu
    echo "== C5 EP (tools/ep_bench.py, tools/ep_fit_time.py)"; timeout -k 10 200 python tools/ep_bench.py 2>&1 | grep -v amdgpu; timeout -k 10 100 python tools/ep_fit_time.py 2>&1 | grep -v amdgpu
    echo "== single right-hand-side solves (tools/trsv_bench.py)"; timeout -k 10 100 python tools/trsv_bench.py 2>&1 | grep -v amdgpu
    echo "== Gram (tools/gram_bench.py)"; timeout -k 10 100 python tools/gram_bench.py 2>&1 | grep -v amdgpu
    echo "== trtri / lauum in one buffer (tools/trtri_ab.py)"; timeout -k 10 200 python tools/trtri_ab.py 65536 2>&1 | grep -v amdgpu; timeout -k 10 100 python tools/trtri_ab.py 16384 2>&1 | grep -v amdgpu
    echo "== non-PD early return (tools/nonpd_timing.py, n = 8192)"; timeout -k 10 100 python tools/nonpd_timing.py 2>&1 | grep -v amdgpu | tail -5
  } > $O/r02_other_configs.txt 2>&1 || { tail -20 $O/r02_other_configs.txt; exit 1; }
  tail -12 $O/r02_other_configs.txt
  timeout -k 10 400 python bench.py > $O/r02_bench_n65536.json.log 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
  cut -c1-300 $O/r02_bench_n65536.json.log
  GPX_FORCE_DIST=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
    --master-port 29512 bench.py --gpus 1 --steps 1 --warmup 1 > $O/dist_w1.log 2>&1 || { tail -20 $O/dist_w1.log; exit 1; }
  grep '^{' $O/dist_w1.log > $O/r02_multi_rank_rehearsal_1gpu.jsonl
  cut -c1-300 $O/r02_multi_rank_rehearsal_1gpu.jsonl
fi
