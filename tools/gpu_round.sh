#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof kernel trace + PMC passes.
# Outputs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
python -m weather2alert_amd.build > gpurun_out/build.log 2>&1 || { cat gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log
tail -12 gpurun_out/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log
timeout -k 10 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench.log
timeout -k 10 300 python bench.py --workload configs1 --no-cpu-baseline > gpurun_out/bench_c1.log 2>&1; tail -1 gpurun_out/bench_c1.log
timeout -k 10 300 python bench.py --workload configs3 --no-cpu-baseline > gpurun_out/bench_c3.log 2>&1; tail -1 gpurun_out/bench_c3.log
if [ "$1" == "prof" ]; then
  R=$PWD
  rm -rf gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_tcc gpurun_out/prof_sq
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_kt.log 2>&1; echo "prof_kt exit $?"
  timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_fetch -- python3 tools/pmc_probe.py > gpurun_out/prof_fetch.log 2>&1; echo "prof_fetch exit $?"
  timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_write -- python3 tools/pmc_probe.py > gpurun_out/prof_write.log 2>&1; echo "prof_write exit $?"
  timeout -k 10 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/prof_tcc -- python3 tools/pmc_probe.py > gpurun_out/prof_tcc.log 2>&1; echo "prof_tcc exit $?"
  timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/prof_sq -- python3 tools/pmc_probe.py > gpurun_out/prof_sq.log 2>&1; echo "prof_sq exit $?"
  for d in prof_kt prof_fetch prof_write prof_tcc prof_sq; do
    python tools/rocprof_summary.py gpurun_out/$d --json gpurun_out/$d.summary.json > gpurun_out/$d.summary.txt 2>&1
    find gpurun_out/$d -name "*.csv" -size +2M -delete
  done
  head -5 gpurun_out/prof_kt.summary.txt
fi
