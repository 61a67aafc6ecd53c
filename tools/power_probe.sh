#!/bin/bash
# Run ON THE GPU BOX: board power / clocks / temperature (rocm-smi, sampled every 0.2 s) while the hot path loops, idle first.
R=${GRAFT_REPO_ROOT:-$(pwd)}
echo "== idle"; rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -E "Power|sclk|mclk|Max" | head -8
( QB_LANES=${QB_LANES:-2} QB_INPUT=frames python3 $R/tools/quick_bench.py 256 ${STEPS:-40000} noprofile > /tmp/qb.out 2>&1 ) &
QB=$!
sleep 4
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" | tr '\n' ' ' ; echo
  sleep 0.3
done
wait $QB
cat /tmp/qb.out
