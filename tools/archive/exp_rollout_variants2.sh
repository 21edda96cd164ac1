#!/bin/bash
# build variants of k_rollout_mfma: tools/exp_rollout_variants2.sh "<flags>;<flags>;..."  (bench_rollout.py iid per build)
IFS=';' read -ra FL <<< "$1"
for flags in "${FL[@]}"; do
  echo "=== W2A_CXXFLAGS=$flags"
  W2A_CXXFLAGS="$flags" python -c "from weather2alert_amd import build; build.build_lib(force=True)" || exit 1
  for i in 1 2; do timeout -k 10 120 python tools/bench_rollout.py 1048576 iid 2>&1 | grep -v amdgpu.ids; done
done
python -c "from weather2alert_amd import build; build.build_lib(force=True)"
