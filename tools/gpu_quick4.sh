#!/bin/bash
export TMPDIR=/tmp
python -m weather2alert_amd.build > /dev/null || exit 1
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "logit or table or oracle_seeded" 2>&1 | tail -3
rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/prof_lt -- python3 bench.py --workload configs3 --steps 17 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python tools/rocprof_summary.py gpurun_out/prof_lt | grep -E "k_logit|k_step" | cut -c1-200
find gpurun_out/prof_lt -name "*.csv" -size +1M -delete
