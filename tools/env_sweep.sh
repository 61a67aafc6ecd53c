#!/bin/bash
# Run ON THE GPU BOX: stream_probe over the values of one environment variable (developer experiments).
# usage: env_sweep.sh VAR "v1 v2 ..." "B ctx lanes" ["B ctx lanes" ...]      (value "-" = variable unset)
R=${GRAFT_REPO_ROOT:-$(pwd)}
VAR=$1; VALS=$2; shift 2
python3 -c "
import sys; sys.path.insert(0, '$R')
from cova_amd import weights as W
open('/tmp/covahip_w1234.bin', 'wb').write(W.to_bytes(W.random_init(1234)))"
for v in $VALS; do
  if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
  for spec in "$@"; do
    echo -n "$VAR=$v  "
    read b c l <<< "$spec"
    timeout -k 5 60 $R/tools/stream_probe /tmp/covahip_w1234.bin $b $c 300 $l || exit 1
  done
done
