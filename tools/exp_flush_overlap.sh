cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e9,1), round(d['ms_per_step'],3), {k:round(v['total_ms'],1) for k,v in d['roofline']['kernels'].items()}, d['config']['graph_checksum'])"; }
run base
MCX_FLUSH_OVERLAP=1 run ov_full
MCX_FLUSH_OVERLAP=1 MCX_GRID_SPLIT=512 MCX_GRID_INSERT=256 run ov_512_256
MCX_FLUSH_OVERLAP=1 MCX_GRID_SPLIT=768 MCX_GRID_INSERT=256 run ov_768_256
MCX_FLUSH_OVERLAP=1 MCX_GRID_SPLIT=512 MCX_GRID_INSERT=512 run ov_512_512
MCX_FLUSH_OVERLAP=1 MCX_GRID_SPLIT=1024 MCX_GRID_INSERT=256 run ov_1024_256
