#!/bin/bash
# one C2 step's kernels on the device's clock:  gpurun -- 'bash tools/c2_timeline.sh [cond_solver]'  ->  gpurun_out/c2_timeline.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/c2tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/c2tl -- python3 tools/c2_trace.py 40 "$@" > gpurun_out/c2tl.json 2> gpurun_out/c2tl.err
python3 tools/step_timeline.py gpurun_out/c2tl k_move 3 > gpurun_out/c2_timeline.txt
tail -1 gpurun_out/c2tl.json
