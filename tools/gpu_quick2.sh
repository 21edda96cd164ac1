#!/bin/bash
export TMPDIR=/tmp
python -m weather2alert_amd.build > /dev/null || exit 1
for eo in iid sorted; do for w in configs3 configs3_gather configs1 configs1_table; do
timeout -k 10 200 python bench.py --workload $w --episode-order $eo --no-cpu-baseline 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$w $eo', 'ms/step %.5f' % d['ms_per_step'], 'kernel us %.2f' % d['roofline']['avg_launch_us'], '%.2f G env-steps/s' % (d['value'] / 1e9))"; done; done
rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/prof_sorted3 -- python3 bench.py --workload configs3 --episode-order sorted --no-cpu-baseline > /dev/null 2>&1
python tools/rocprof_summary.py gpurun_out/prof_sorted3 | head -8 | cut -c1-250
find gpurun_out/prof_sorted3 -name "*.csv" -size +1M -delete
