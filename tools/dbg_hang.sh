#!/bin/bash
# reproduce / inspect a CLI build that does not finish: thread states of the process after 40 s
cd "$(dirname "$0")/.."
T=${TMPDIR:-/tmp}/dbg_hang; mkdir -p $T
N=${1:-40000000}
(cd $T && python /root/repo/tools/exp_parse_gen.py $N $2)
export MCX_TIMING=1
mccortex_amd/bin/mccortex31 build -f -k 31 -n 1G -m 30G -t 32 --sort -s x --seq $T/reads.fq $T/out.ctx > $T/err.log 2>&1 &
pid=$!
for i in $(seq 1 40); do sleep 1; kill -0 $pid 2>/dev/null || break; done
if kill -0 $pid 2>/dev/null; then
  echo "STILL RUNNING after 40 s: thread states"
  for t in /proc/$pid/task/*; do echo "$(cat $t/comm) state=$(awk '{print $3}' $t/stat) wchan=$(cat $t/wchan 2>/dev/null) utime=$(awk '{print $14}' $t/stat)"; done | sort | uniq -c | sort -rn | head -40
  echo "--- kernel stacks (main thread)"; cat /proc/$pid/stack 2>/dev/null | head
  kill -9 $pid
else
  echo "finished"
fi
echo "--- stderr"; cat $T/err.log | tail -30
rm -rf $T
