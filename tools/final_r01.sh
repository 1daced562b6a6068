#!/bin/bash
# Round-1 evidence run: everything DESIGN.md quotes, in one go (each step under its own limit).
set -o pipefail
mkdir -p gpurun_out
bash tools/profile_r01.sh > gpurun_out/profile_r01.out 2>&1 || { tail -5 gpurun_out/profile_r01.out; exit 1; }
echo "profile done"
bash tools/potrf_trace.sh > gpurun_out/ptrace_join.txt 2>&1 || { tail -5 gpurun_out/ptrace_join.txt; exit 1; }
echo "trace done"
bash tools/potrf_pmc.sh > gpurun_out/potrf_pmc.out 2>&1 || { tail -5 gpurun_out/potrf_pmc.out; exit 1; }
bash tools/gemm_pmc.sh > gpurun_out/gemm_pmc.out 2>&1 || { tail -5 gpurun_out/gemm_pmc.out; exit 1; }
echo "pmc done"
timeout -k 10 400 python tools/configs_bench.py > gpurun_out/configs.log 2>&1 || { tail -5 gpurun_out/configs.log; exit 1; }
timeout -k 10 200 python tools/ep_bench.py 8192 > gpurun_out/ep_bench.log 2>&1 || { tail -5 gpurun_out/ep_bench.log; exit 1; }
timeout -k 10 120 python tools/potrf_small_bench.py > gpurun_out/potrf_small.log 2>&1 || { tail -5 gpurun_out/potrf_small.log; exit 1; }
echo "configs done"
BLOCKS="2048 1024 512" bash tools/dist_rehearsal.sh > gpurun_out/rehearsal.out 2>&1 || { tail -5 gpurun_out/rehearsal.out; exit 1; }
echo "rehearsal done"
