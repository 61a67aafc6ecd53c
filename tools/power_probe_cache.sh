R=${GRAFT_REPO_ROOT:-$(pwd)}
for spec in "5 128" "5 16" "9 128" "5 4096"; do
  read K MB <<< "$spec"
  ( timeout -k 5 60 $R/tools/probes/power_probe $K 3.5 $MB > /tmp/pp_$K.out 2>&1 ) &
  P=$!
  sleep 1.8
  for i in 1 2 3; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '; echo -n "| "; done
  echo " buffer $MB MB"
  wait $P; cat /tmp/pp_$K.out
done
