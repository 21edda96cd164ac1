#!/bin/bash
# round 5, session A: the call-sequence fuzz with recorded packed graphs + later replays (statistics of what it visits), the
# rest of the GPU suite, and where an on-device rollout's episode goes after the order / tile-list rework (kernel trace).
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m weather2alert_amd.build > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 500 python tools/sequence_fuzz.py --sequences 1200 --seed 2024 --keep-going 3 > gpurun_out/seqfuzz_r05_seed2024.log 2>&1
echo "seqfuzz exit $?"; tail -4 gpurun_out/seqfuzz_r05_seed2024.log | cut -c1-1500
timeout -k 10 300 python tools/sequence_fuzz.py --sequences 40 --seed 7 --big > gpurun_out/seqfuzz_r05_big.log 2>&1
echo "seqfuzz big exit $?"; tail -3 gpurun_out/seqfuzz_r05_big.log | cut -c1-1500
timeout -k 10 300 python tools/bench_rollout.py 1048576 iid > gpurun_out/bench_rollout.log 2>&1; echo "bench_rollout exit $?"; cat gpurun_out/bench_rollout.log | grep -v amdgpu.ids
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_rollout -- python3 $GRAFT_REPO_ROOT/tools/bench_rollout.py 1048576 iid > $GRAFT_REPO_ROOT/gpurun_out/prof_rollout.log 2>&1; echo "prof rollout exit $?"
cd $GRAFT_REPO_ROOT && python tools/rocprof_summary.py gpurun_out/prof_rollout 2>/dev/null | head -16 > gpurun_out/kernel_trace_rollout_bench.txt; cat gpurun_out/kernel_trace_rollout_bench.txt | cut -c1-220
find gpurun_out/prof_rollout -name "*.csv" -size +2M -delete
