#!/bin/bash
# Run ON THE GPU BOX: bboxcc at B = 65,536 over objects per frame for library builds, alternating ("cur" = the tree's).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cp $R/cova_amd/libcovahip.so /tmp/keep.so
for v in "$@"; do
  if [ "$v" = cur ]; then cp /tmp/keep.so $R/cova_amd/libcovahip.so; else cp $R/$v $R/cova_amd/libcovahip.so; fi
  echo "== $v"
  SWEEP_KINDS=${CC_KINDS:-blobs,obj6,obj20,obj50,obj100} SWEEP_BATCHES=65536 SWEEP_OUT=/tmp/x.json python3 $R/tools/bboxcc_sweep.py 2>&1 | grep -o "'kind': '[a-z0-9]*'\|'ns_per_frame': [0-9.]*\|'frac_hbm_peak': [0-9.]*\|'cap_pass1': [0-9]*\|'overflow_frac_pass1': [0-9.]*" | tr '\n' ' ' | sed "s/'kind'/\n  'kind'/g"
  echo
done
cp /tmp/keep.so $R/cova_amd/libcovahip.so
