#!/bin/bash
# Round profile: kernel trace statistics of the default bench command, then PMC passes (HBM read / write bytes, instruction counts)
# of the dominant kernels -- each pass behind 18 warm-up steps, its last two launches taken: the STEADY state (rounds 1-5 took launches
# 2-3 of a fresh box, when a tenth of the aerosol is still activating and every other wave lists a droplet).  Run on the GPU box:  gpurun -- 'bash tools/profile_round.sh r02f'   (then copy summary.txt, traffic.json,
# bench_under_trace.json and the trace's kernel_stats.csv into profiles/<tag>_*, and run tools/isa_histogram.py on the traffic file)
tag=${1:-r01}
shift
extra="$@"        # further bench.py arguments (e.g. --dbg COND_NO_DEAL, --workload coal-stress)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict-leg --no-toms-leg --no-host-leg --no-extra-legs --stage-steps 0 $extra > $out/bench_under_trace.json 2> $out/trace.log
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_BRANCH" "SQ_INSTS_LDS SQ_IFETCH SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/pmc_$i -- python3 bench.py --steps 2 --warmup 18 --no-cpu-baseline --no-strict-leg --no-toms-leg --no-host-leg --no-extra-legs --no-stage-timers $extra > $out/pmc_$i.log 2>&1
done
python3 tools/summarise_profile.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
