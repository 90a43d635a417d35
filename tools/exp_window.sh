for w in 8000000000 12000000000 16000000000; do
  python bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-extras --defer-tuples $w 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('window', $w, json.dumps({'ms_per_step': d['ms_per_step'], 'frac': d['roofline']['frac'], 'kernels': {k: v['total_ms'] for k, v in d['roofline'].get('kernels', {}).items()}}))
"
done
