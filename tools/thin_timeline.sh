cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --no-cpu-baseline --no-strict-leg --no-toms-leg --no-host-leg --no-extra-legs --stage-steps 0"
for v in "slab16:--nx 16 --steps 40 --warmup 20" "ring16:--nx 16 --self-ring --steps 40 --warmup 20"; do
  n=${v%%:*}; a=${v#*:}
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$n -- $B $a > gpurun_out/tl_$n.json 2> gpurun_out/tl_$n.err
  python3 tools/step_timeline.py gpurun_out/tl_$n k_move 3 > gpurun_out/tl_$n.txt
  $B $a > gpurun_out/tl_${n}_plain.json 2>/dev/null
done
