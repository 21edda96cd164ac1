#!/bin/bash
# per-kernel durations of the posterior-mean step for build variants:  tools/exp_pm_trace.sh "flagsA;flagsB"
IFS=';' read -ra FL <<< "$1"
i=0
for flags in "${FL[@]}"; do
  i=$((i+1))
  echo "=== W2A_CXXFLAGS=$flags" | tee -a gpurun_out/pm_trace.log
  W2A_CXXFLAGS="$flags" python -c "from weather2alert_amd import build; build.build_lib(force=True)" || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/pm_trace_$i -- python3 tools/exp_posterior.py > gpurun_out/pm_trace_$i.log 2>&1
  f=$(find gpurun_out/pm_trace_$i -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY' | tee -a gpurun_out/pm_trace.log
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("k_pm","k_posterior","k_step")):
        print("%-50s calls %5s avg %9.2f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"])/1e3))
PY
  find gpurun_out/pm_trace_$i -name "*.csv" -size +2M -delete
done
python -c "from weather2alert_amd import build; build.build_lib(force=True)"
