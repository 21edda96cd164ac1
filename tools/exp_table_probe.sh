#!/bin/bash
# price the logit-slice step pattern (f32 pairs from a 6.5 MB day slice + run-time coefficients from a
# compact 2.3 MB table, lane = env) against today's coefficient-row gather with the arithmetic-free probe, before any kernel code.
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/fabric_probe_table.log
: > $out
base="-DPROBE_F64 -DPROBE_DEP -DPROBE_DAYS -DPROBE_RANDOM_DATA -DPROBE_INPLACE -DPROBE_PACKED"
for extra in "" "-DPROBE_TABLE=1" "-DPROBE_TABLE=2" "" "-DPROBE_TABLE=1"; do
  echo "== $base $extra" | tee -a $out
  hipcc --offload-arch=gfx950 -O3 $base $extra tools/fabric_probe.hip -o /tmp/fabric_probe || exit 1
  timeout -k 10 120 /tmp/fabric_probe 2>&1 | tee -a $out || exit 1
done
