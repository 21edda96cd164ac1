#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, optional rocprof kernel trace + PMC passes.
# usage: tools/gpu_round.sh [prof] [workload ...]     outputs under gpurun_out/
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m weather2alert_amd.build > gpurun_out/build.log 2>&1 || { cat gpurun_out/build.log; exit 1; }
if [ "$SKIPTESTS" != "1" ]; then timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log; fi
tail -16 gpurun_out/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log
timeout -k 10 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench.log
for w in configs1 configs3 configs3_gather configs1_table; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/bench_$w.log 2>&1; tail -1 gpurun_out/bench_$w.log | cut -c1-420
done
if [ "$1" == "prof" ]; then
  R=$PWD
  for w in configs2 configs3; do
    rm -rf gpurun_out/prof_*_$w
    timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt_$w -- python3 bench.py --workload $w --no-cpu-baseline > gpurun_out/prof_kt_$w.log 2>&1; echo "prof_kt $w exit $?"
    timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_fetch_$w -- python3 tools/pmc_probe.py --workload $w > gpurun_out/prof_fetch_$w.log 2>&1; echo "prof_fetch exit $?"
    timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_write_$w -- python3 tools/pmc_probe.py --workload $w > gpurun_out/prof_write_$w.log 2>&1; echo "prof_write exit $?"
    timeout -k 10 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/prof_tcc_$w -- python3 tools/pmc_probe.py --workload $w > gpurun_out/prof_tcc_$w.log 2>&1; echo "prof_tcc exit $?"
    timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/prof_sq_$w -- python3 tools/pmc_probe.py --workload $w > gpurun_out/prof_sq_$w.log 2>&1; echo "prof_sq exit $?"
    timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/prof_mfma_$w -- python3 tools/pmc_probe.py --workload $w > gpurun_out/prof_mfma_$w.log 2>&1; echo "prof_mfma exit $?"
    for d in prof_kt prof_fetch prof_write prof_tcc prof_sq prof_mfma; do
      python tools/rocprof_summary.py gpurun_out/${d}_$w --json gpurun_out/${d}_$w.summary.json > gpurun_out/${d}_$w.summary.txt 2>&1
      find gpurun_out/${d}_$w -name "*.csv" -size +2M -delete
    done
    head -4 gpurun_out/prof_kt_$w.summary.txt | cut -c1-300
  done
fi
