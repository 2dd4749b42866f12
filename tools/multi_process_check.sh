#!/bin/bash
# N vp9hip_dec processes side by side on one GPU with --md5: every process's exit code, its MD5 lines against the golden
# list and whether a loop-filter row ever gave up waiting.   tools/multi_process_check.sh <ivf> <N> <entropy threads>
ivf="$1"; n="$2"; thr="$3"
for i in $(seq 1 $n); do
  ( cuda-vp9_amd/vp9hip_dec --md5 -o "img-%wx%h-%4.i420" --loops=${LOOPS:-6} --threads=$thr "$ivf" > /tmp/mpc_$i.md5 2> /tmp/mpc_$i.err; echo "rc=$?" >> /tmp/mpc_$i.err ) &
done
wait
for i in $(seq 1 $n); do echo "proc $i: $(tail -1 /tmp/mpc_$i.err) lines=$(wc -l < /tmp/mpc_$i.md5) bad=$(sort -u /tmp/mpc_$i.md5 | diff - <(sort -u ${ivf%.ivf}.md5) | wc -l) $(grep -a -c "gave up" /tmp/mpc_$i.err)"; done
