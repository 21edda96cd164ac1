#!/bin/bash
export TMPDIR=/tmp
python -m weather2alert_amd.build > /dev/null || exit 1
for g in 0 17 51; do for w in configs1 configs1_table; do
timeout -k 10 200 python bench.py --workload $w --graph $g --steps 1530 --no-cpu-baseline 2>&1 | grep -E "^\{|Error|error|assert" | python -c "
import sys, json
t = sys.stdin.read()
try:
    d = json.loads(t); print('$w graph=$g', 'ms/step %.5f' % d['ms_per_step'], 'dev us %.2f' % d['roofline']['avg_launch_us'], '%.2f G env-steps/s' % (d['value'] / 1e9), 'ret', d['mean_final_return'])
except Exception as e: print('FAIL', t[:500])"; done; done
