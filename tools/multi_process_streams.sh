#!/bin/bash
# N vpxdec_hip_mt processes decoding the same stream side by side on ONE GPU (what a batch of independent streams
# per GPU looks like): per-process warm fps and their sum.   tools/multi_process_streams.sh <ivf> <N> [threads per process]
ivf="$1"; n="${2:-4}"; thr="${3:-8}"
export VP9HIP_PACK_THREADS=4 VP9HIP_SHIM_THREADS=$thr
for i in $(seq 1 $n); do
  ( shim/build/vpxdec_hip_mt --noblit --summary --loops=6 "$ivf" 2>&1 | grep -a -o "([0-9.]* fps)" | tail -3 | tr -d '()fps ' | paste -sd' ' > /tmp/mp_$i.txt ) &
done
wait
for i in $(seq 1 $n); do
  v=$(awk '{s=0; for(i=1;i<=NF;i++) s+=$i; print s/NF}' /tmp/mp_$i.txt); echo "process $i: $v fps (last loops: $(cat /tmp/mp_$i.txt))"
done
cat $(for i in $(seq 1 $n); do echo /tmp/mp_$i.txt; done) | awk -v n=$n '{s=0; for(i=1;i<=NF;i++) s+=$i; t+=s/NF} END {printf "sum of %d processes: %.1f fps\n", n, t}'
