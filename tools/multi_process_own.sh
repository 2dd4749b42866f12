#!/bin/bash
# N vp9hip_dec processes (the decoder built only from this repository) decoding the same stream side by side on ONE
# GPU, every frame fetched to the host: per-process warm fps and their sum.
#   tools/multi_process_own.sh <ivf> <N> [entropy threads per process]
ivf="$1"; n="${2:-4}"; thr="${3:-4}"
export VP9HIP_PACK_THREADS=2
for i in $(seq 1 $n); do
  ( cuda-vp9_amd/vp9hip_dec --noblit --fetch --summary --loops=6 --threads=$thr "$ivf" 2>&1 | grep -a -o "([0-9.]* fps)" | tail -3 | tr -d '()fps ' | paste -sd' ' > /tmp/mpo_$i.txt ) &
done
wait
cat $(for i in $(seq 1 $n); do echo /tmp/mpo_$i.txt; done) | awk -v n=$n '{s=0; for(i=1;i<=NF;i++) s+=$i; t+=s/NF} END {printf "sum of %d processes: %.1f fps\n", n, t}'
