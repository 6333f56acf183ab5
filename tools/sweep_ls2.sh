#!/bin/bash
# the big size class alone (no other streams): per-chunk times
export IPM_LS_DEBUG=2
for sb in 1048576 16 24; do
for set in "QAP15" "DFL001" "PILOT87" "PILOT87 QAP15 DFL001"; do
  echo "== small_blocks $sb: $set"
  IPM_LS_SMALL_BLOCKS=$sb timeout -k 10 120 python tools/ls_probe.py $set 2> gpurun_out/ls_probe_err.txt | head -4
  grep "lockstep chunk" gpurun_out/ls_probe_err.txt | awk '{print $5, $7, $14}' | tr -d ',' | uniq -c -w 8 | head -40
  grep "^\[lockstep\] batch" gpurun_out/ls_probe_err.txt
done
done
