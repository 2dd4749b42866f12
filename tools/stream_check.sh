#!/bin/bash
# tools/stream_check.sh <decoder> <stream.ivf> <golden.md5>: decode with vpxdec's --md5 and diff per frame.
dec="$1"; ivf="$2"; gold="$3"
out=$("$dec" --rawvideo --md5 -o 'img-%wx%h-%4.i420' "$ivf" 2>&1)
rc=$?
echo "$out" | grep -a -E '^[0-9a-f]{32}  img-' > /tmp/stream_check.$$
n=$(wc -l < /tmp/stream_check.$$); g=$(wc -l < "$gold")
bad=$(diff /tmp/stream_check.$$ "$gold" | grep -c '^<')
echo "$(basename $ivf): rc=$rc frames=$n golden=$g mismatching=$bad"
if [ "$rc" != 0 ] || [ "$n" != "$g" ] || [ "$bad" != 0 ]; then echo "$out" | grep -a -v "^frame\|gpu_\|^[0-9a-f]\{32\}" | tail -5; diff /tmp/stream_check.$$ "$gold" | head -4; fi
rm -f /tmp/stream_check.$$
