#!/bin/bash
# Run ON THE GPU BOX: board power (rocm-smi) while every CU runs one kind of instruction stream (tools/probes/power_probe).
R=${GRAFT_REPO_ROOT:-$(pwd)}
for K in ${KINDS:-0 1 2 3 7 4 6 5}; do
  ( timeout -k 5 60 $R/tools/probes/power_probe $K ${SECS:-3.5} > /tmp/pp_$K.out 2>&1 ) &
  P=$!
  sleep ${LEAD:-1.6}
  for i in 1 2 3; do
    rocm-smi --showpower --showclocks 2>&1 | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '
    echo -n "| "
  done
  echo
  wait $P
  cat /tmp/pp_$K.out
done
