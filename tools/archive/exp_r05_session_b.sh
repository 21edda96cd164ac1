#!/bin/bash
# round 5, session B: $1 = 1: the whole GPU suite + kernel trace / counter passes of the headline workload (configs2);
# $1 = 2: configs3 passes, the driver's own invocation under the kernel trace and without, the int8 posterior-mean counters
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "$1" == "1" ]; then
  bash tools/gpu_session.sh tests prof:configs2
else
  python -m weather2alert_amd.build > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
  bash tools/gpu_session.sh prof:configs3
  R=$PWD
  timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_args.log 2>&1; echo "driver args bench exit $?"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt_driver_args -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/prof_kt_driver_args.log 2>&1; echo "prof_kt driver_args exit $?"
  python tools/rocprof_summary.py gpurun_out/prof_kt_driver_args 2>/dev/null | head -5 | cut -c1-250
  find gpurun_out/prof_kt_driver_args -name "*.csv" -size +2M -delete
  bash tools/gpu_session.sh prof:configs2:pm_matrix_i8
fi
