#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu_session.sh tests
timeout -k 10 300 python tools/sequence_fuzz.py --sequences 800 --seed 505 --keep-going 3 > gpurun_out/seqfuzz_r05_seed505.log 2>&1
echo "seqfuzz 505 exit $?"; tail -1 gpurun_out/seqfuzz_r05_seed505.log | cut -c1-1500
timeout -k 10 300 python tools/sequence_fuzz.py --sequences 1200 --seed 31 --keep-going 3 > gpurun_out/seqfuzz_r05_seed31.log 2>&1
echo "seqfuzz 31 exit $?"; tail -1 gpurun_out/seqfuzz_r05_seed31.log | cut -c1-1500
