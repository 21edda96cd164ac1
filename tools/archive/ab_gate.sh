for W in configs2 configs3; do for G in "" 1; do
  echo "== $W W2A_NO_GATE_BITS=$G"
  W2A_NO_GATE_BITS=$G python bench.py --workload $W --no-cpu-baseline --no-extras --steps 612 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel %.2f us  e2e %.2f G/s' % (d['roofline']['avg_launch_us'], d['value']/1e9))"
done; done
