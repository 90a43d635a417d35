#!/bin/bash
# Host-side ingest on the GPU box's CPU: where do the parser threads' seconds go?  (round 5)
set -e
cd "$(dirname "$0")/.."
T=${TMPDIR:-/tmp}/exp_parse; mkdir -p $T
gcc -O2 -std=gnu11 -Imccortex_amd/host -Iinclude tools/exp_parse.c mccortex_amd/host/par_ingest.c mccortex_amd/host/seq_in.c -o $T/harness -lpthread -lz 2>/dev/null
gcc -O2 tools/exp_touch.c -o $T/touch -lpthread
(cd $T && python /root/repo/tools/exp_parse_gen.py 10000000)
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null || true
export MCX_TIMING=1
(cd $T && python /root/repo/tools/exp_parse_gen.py 40000000)
echo "== CLI, parser alone (MCX_PARSE_ONLY=1), then the whole build, 40 M reads"
for i in 1 2; do MCX_PARSE_ONLY=1 mccortex_amd/bin/mccortex31 build -f -k 31 -n 1G -m 30G -t 32 --sort -s x --seq $T/reads.fq $T/out.ctx 2>&1 | grep "timing" | grep -v export | tr '\n' ' '; echo; done
for pb in 33554432 67108864 134217728; do for i in 1 2; do rm -f $T/out.ctx; sleep 2; echo "MCX_PAR_BATCH=$pb"; S=$(date +%s.%N); MCX_PAR_BATCH=$pb MCX_STAGE_TIMING=1 mccortex_amd/bin/mccortex31 build -f -k 31 -n 1G -m 30G -t 32 --sort -s x --seq $T/reads.fq $T/out.ctx 2>&1 | grep "par_ingest\|stage\]\|inputs submitted\|records delivered\|graph written\|table alloc" | tr '\n' ' '; echo " wall $(echo "$(date +%s.%N) - $S" | bc)"; done; done
rm -rf $T
