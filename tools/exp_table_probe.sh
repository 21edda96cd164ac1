#!/bin/bash
# price the logit-slice step patterns against today's coefficient-row gather with the
# arithmetic-free probe, before any kernel code.   usage: tools/exp_table_probe.sh [out.log [variant ...]]
#   -DPROBE_TABLE=1  f32 pair per env from a 6.5 MB day slice + 32 B of run-time coefficients from a compact 2.3 MB table
#   -DPROBE_TABLE=2  one f32 per head from two slices (the effectiveness one read by ~5 % of the envs) + the same table
#   -DPROBE_TABLE=3  ONE 16-B entry per env and head {table part of the logit, the three run-time coefficients} from a
#                    12.7 MB day slice; no other coefficient gather
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/${1:-fabric_probe_table.log}
shift
variants=("$@")
[ ${#variants[@]} -eq 0 ] && variants=("base" "-DPROBE_TABLE=1" "-DPROBE_TABLE=2" "base" "-DPROBE_TABLE=1")
: > $out
base="-DPROBE_F64 -DPROBE_DEP -DPROBE_DAYS -DPROBE_RANDOM_DATA -DPROBE_INPLACE -DPROBE_PACKED"
for extra in "${variants[@]}"; do
  [ "$extra" == "base" ] && extra=""
  echo "== $base $extra" | tee -a $out
  hipcc --offload-arch=gfx950 -O3 $base $extra tools/fabric_probe.hip -o /tmp/fabric_probe || exit 1
  timeout -k 10 120 /tmp/fabric_probe 2>&1 | tee -a $out || exit 1
done
