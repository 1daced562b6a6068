#!/bin/bash
# two gloo ranks on cuda:0, traced; each under its own time limit
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 WORLD_SIZE=2
RANK=0 timeout -k 5 90 python -u tools/dist_debug.py "$@" > gpurun_out/dist_debug_r0.log 2>&1 &
p0=$!
RANK=1 timeout -k 5 90 python -u tools/dist_debug.py "$@" > gpurun_out/dist_debug_r1.log 2>&1 &
p1=$!
while kill -0 $p0 2>/dev/null || kill -0 $p1 2>/dev/null; do sleep 5; echo tick; done
wait $p0; echo rc0=$?
wait $p1; echo rc1=$?
