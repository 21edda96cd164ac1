#!/bin/bash
# step time vs batch size (per-env cost reveals cache-residency effects of the streamed state)
export TMPDIR=/tmp
python -m weather2alert_amd.build > /dev/null || exit 1
for n in 131072 262144 524288 1048576 2097152 4194304; do
timeout -k 10 200 python bench.py --num-envs $n --no-cpu-baseline --no-extras --steps 612 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); n=d['config']['num_envs_per_gpu']; us=d['roofline']['avg_launch_us']; print('n=%8d  %.2f us/step  %.2f ns/env  %.2f G env-steps/s' % (n, us, us*1e3/n, d['value']/1e9))"
done
