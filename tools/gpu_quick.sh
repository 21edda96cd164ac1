#!/bin/bash
# quick GPU checks: host-overhead at small N, 2-rank rehearsal on one GPU (gloo), config benches
export TMPDIR=/tmp
python -m weather2alert_amd.build > /dev/null || exit 1
for n in 64 65536; do timeout -k 10 200 python bench.py --workload configs1 --num-envs $n --no-cpu-baseline --steps 2000 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('n=%d' % d['config']['num_envs_per_gpu'], 'ms/step %.5f' % d['ms_per_step'], 'kernel us %.2f' % d['roofline']['avg_launch_us'], '%.2f G env-steps/s' % (d['value'] / 1e9))"; done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --num-envs 262144 --steps 320 --warmup 10 2>&1 | grep -E "^\{|Error|error" | cut -c1-700
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for eo in iid sorted; do for w in configs2 configs3; do
timeout -k 10 200 python bench.py --workload $w --episode-order $eo --no-cpu-baseline 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$w $eo', 'ms/step %.5f' % d['ms_per_step'], 'kernel us %.2f' % d['roofline']['avg_launch_us'], '%.2f G env-steps/s' % (d['value'] / 1e9))"; done; done
