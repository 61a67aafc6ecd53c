R=${GRAFT_REPO_ROOT:-$(pwd)}
cp $R/cova_amd/libcovahip.so /tmp/keep.so
for v in cc2 cc3 cc2 cc3; do
  cp $R/ab_tmp/$v.so $R/cova_amd/libcovahip.so
  echo "== $v"
  SWEEP_KINDS=blobs,obj6,obj20,obj50 SWEEP_BATCHES=65536 SWEEP_OUT=/tmp/x.json python3 $R/tools/bboxcc_sweep.py 2>&1 | grep -o "'kind': '[a-z0-9]*'\|'ns_per_frame': [0-9.]*\|'frac_hbm_peak': [0-9.]*\|'overflow_frac_pass1': [0-9.]*" | tr '\n' ' '
  echo
done
cp /tmp/keep.so $R/cova_amd/libcovahip.so
