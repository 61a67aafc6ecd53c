#!/bin/bash
# Run ON THE GPU BOX: kernel + memory-copy time line of the pinned pipeline (tools/pipe_slots.py) for one or more library builds.
# usage: pipe_trace.sh <outdir> <lib.so|cur> [...]     env: PS_ONLY=lanes,slots (default 3,3)  PS_BLOB=1  PS_STEPS (default 60)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$(realpath -m $1); shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cp $R/cova_amd/libcovahip.so /tmp/ab_cur.so
for v in "$@"; do
    if [ "$v" = cur ]; then cp /tmp/ab_cur.so $R/cova_amd/libcovahip.so; else cp "$R/$v" $R/cova_amd/libcovahip.so; fi
    tag=$(basename $v .so)
    rm -rf /tmp/ptrace_$tag
    PS_ONLY=${PS_ONLY:-3,3} PS_STEPS=${PS_STEPS:-60} GRAFT_REPO_ROOT=$R timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv \
        -d /tmp/ptrace_$tag -- python3 $R/tools/pipe_slots.py > $OUT/trace_$tag.log 2>&1
    for f in $(find /tmp/ptrace_$tag -name "*kernel_trace.csv" -o -name "*memory_copy_trace.csv"); do cp $f $OUT/${tag}_$(basename $f | sed 's/^[0-9]*_//'); done
done
cp /tmp/ab_cur.so $R/cova_amd/libcovahip.so
ls -la $OUT
