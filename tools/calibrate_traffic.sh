#!/bin/bash
# FETCH_SIZE calibration on known-byte access patterns (tools/traffic_calib.hip):  gpurun -- 'bash tools/calibrate_traffic.sh r03'
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/calib_$tag
mkdir -p $out
[ -x tools/traffic_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/traffic_calib tools/traffic_calib.hip
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- ./tools/traffic_calib > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- ./tools/traffic_calib > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --output-format csv -d $out/pmc_req -- ./tools/traffic_calib > $out/pmc_req.log 2>&1
python3 tools/calibrate_traffic.py $out > $out/calibration.json
cat $out/calibration.json
