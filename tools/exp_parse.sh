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
for t in 16 32; do for i in 1 2; do bash -c "time $T/harness $T/reads.fq $t" 2>&1 | tr '\n' ' '; echo; done; done
(cd $T && python /root/repo/tools/exp_parse_gen.py 40000000)
for t in 16 32; do for i in 1 2; do bash -c "time $T/harness $T/reads.fq $t" 2>&1 | tr '\n' ' '; echo; done; done
for m in 0 1; do for i in 1 2; do bash -c "time $T/touch $T/reads.fq 32 $m" 2>&1 | tr '\n' ' '; echo; done; done
echo "== CLI, parser alone (MCX_PARSE_ONLY=1), then the whole build, 40 M reads"
for i in 1 2; do MCX_PARSE_ONLY=1 mccortex_amd/bin/mccortex31 build -f -k 31 -n 1G -m 30G -t 32 --sort -s x --seq $T/reads.fq $T/out.ctx 2>&1 | grep "timing" | grep -v export | tr '\n' ' '; echo; done
for i in 1 2; do rm -f $T/out.ctx; MCX_STAGE_TIMING=1 mccortex_amd/bin/mccortex31 build -f -k 31 -n 1G -m 30G -t 32 --sort -s x --seq $T/reads.fq $T/out.ctx 2>&1 | grep "timing\|stage\]" | tr '\n' ' '; echo; done
rm -rf $T
