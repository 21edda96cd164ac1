#!/bin/bash
# A/B at configs[1] (65 536 envs, k_step): lanes per env (LANES = 4 default, 8, 2) -- the kernel is latency-bound and the chip
# underfilled at this size, so more, shorter waves might hide the two memory hops better
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/ab_lanes_configs1.log
: > $out
for flags in "" "-DLANES=8" "-DLANES=2" ""; do
  echo "=== W2A_CXXFLAGS=$flags" | tee -a $out
  W2A_CXXFLAGS="$flags" python -c "from weather2alert_amd import build; build.build_lib(force=True)" || exit 1
  W2A_CXXFLAGS="$flags" timeout -k 10 200 python bench.py --workload configs1 --no-cpu-baseline --no-extras --no-parity --no-calibration --steps 612 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs1 kernel %.2f us  e2e %.2f G/s  kernel %s' % (d['roofline']['avg_launch_us'], d['value']/1e9, d['roofline']['kernel']))" | tee -a $out
done
python -c "from weather2alert_amd import build; build.build_lib(force=True)"
