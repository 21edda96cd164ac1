#!/bin/bash
# A/B compile-flag variants with bench.py workloads: tools/ab_bench.sh "<flags>" ... ; env WL="configs2 configs3" EO="iid sorted"
export TMPDIR=/tmp
WL=${WL:-"configs2 configs3"}; EO=${EO:-"iid sorted"}
for flags in "$@"; do
  echo "=== W2A_CXXFLAGS=$flags"
  W2A_CXXFLAGS="$flags" python -c "from weather2alert_amd import build; build.build_lib(force=True)" || exit 1
  for eo in $EO; do for w in $WL; do
    timeout -k 10 200 python bench.py --workload $w --episode-order $eo --no-cpu-baseline 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$w $eo', 'kernel us %.2f' % d['roofline']['avg_launch_us'], '%.2f G env-steps/s' % (d['value'] / 1e9))"; done; done
done
python -c "from weather2alert_amd import build; build.build_lib(force=True)"
