#!/bin/bash
# A/B kernel-geometry variants on the GPU box: rebuild libw2a.so with extra flags, run the locality experiment.
mkdir -p gpurun_out
for flags in "$@"; do
  echo "=== W2A_CXXFLAGS=$flags"
  W2A_CXXFLAGS="$flags" python -c "from weather2alert_amd import build; build.build_lib(force=True)" || exit 1
  timeout -k 10 300 python tools/exp_locality.py --quick 2>&1 | grep -v amdgpu.ids
done
python -c "from weather2alert_amd import build; build.build_lib(force=True)"
