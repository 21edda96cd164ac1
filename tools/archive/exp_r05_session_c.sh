#!/bin/bash
# round 5, session C: the GPU suite on the tree with the packed in-kernel autoreset, the call-sequence fuzz (normal + big
# batches), and bench.py as a hipGraph of 51 recorded steps (in-kernel autoreset) next to the eager default
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu_session.sh tests
timeout -k 10 400 python tools/sequence_fuzz.py --sequences 800 --seed 505 --keep-going 3 > gpurun_out/seqfuzz_r05_seed505.log 2>&1
echo "seqfuzz exit $?"; tail -2 gpurun_out/seqfuzz_r05_seed505.log | cut -c1-1500
timeout -k 10 300 python tools/sequence_fuzz.py --sequences 40 --seed 8 --big > gpurun_out/seqfuzz_r05_big2.log 2>&1
echo "seqfuzz big exit $?"; tail -2 gpurun_out/seqfuzz_r05_big2.log | cut -c1-1500
timeout -k 10 200 python bench.py --graph 51 --steps 1530 --no-cpu-baseline --no-extras > gpurun_out/bench_graph51.log 2>&1; echo "bench graph exit $?"
tail -1 gpurun_out/bench_graph51.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph51:', d['value']/1e9, 'G', d['ms_per_step']*1e3, 'us/step', d['config']['packed_lockstep_state'], d['roofline']['kernel'], d['parity'] and d['parity'].get('ok'), d['parity'])"
# A/B: the packed mirror's words loaded non-temporally (W2A_S64_NT_STATE bit 2)
AB_FLAGS=";-DW2A_S64_NT_STATE=4;" AB_BENCH_ARGS="--no-calibration --no-parity" bash tools/gpu_session.sh ab2
cp gpurun_out/ab2.log gpurun_out/ab_nt_packed_state.log
