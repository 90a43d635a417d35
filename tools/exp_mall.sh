for cfg in "8388608 2000000" "33554432 8000000" "134217728 30000000" "1073741824 200000000"; do
  set -- $cfg
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --table-slots $1 --genome $2 --err 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slots=$1 genome=$2  %.3e kmers/s  kernel %.2f ms distinct %d' % (d['value'], d['roofline']['avg_kernel_ms'], d['config']['distinct_kmers_rank0']))"
done
