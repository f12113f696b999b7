#!/bin/bash
# usage: dbg_spmd.sh <tag> <mode> <world>
tag=$1; mode=$2; world=$3
port=$((29600 + RANDOM % 300))
for r in $(seq 0 $((world-1))); do
  timeout 120 python tests/_spmd_worker.py $mode $r $world $port gpurun_out/dbg_${tag}_r%d.npy host > gpurun_out/dbg_${tag}_$r.log 2>&1 &
done
wait
for r in $(seq 0 $((world-1))); do echo "== $tag rank $r"; grep -v "Gloo\|socket.cpp\|amdgpu.ids" gpurun_out/dbg_${tag}_$r.log | tail -8; done
