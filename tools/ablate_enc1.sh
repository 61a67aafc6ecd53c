#!/bin/bash
# Developer helper (here): builds ab_tmp/abl<N>.so = the library with -DE1_ABL=N (enc1_mfma with one part removed; results
# are wrong, the timing says what the part costs).  Run on the GPU box with tools/ab.sh ab_tmp/abl0.so ab_tmp/abl1.so ...
cd "$(dirname "$0")/../cova_amd/csrc" || exit 1
mkdir -p ../../ab_tmp /tmp/isa
for n in "$@"; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-unused-value -Wno-unused-variable -I../../include -I. -fno-honor-nans \
        -mllvm -pragma-unroll-threshold=1000000 -DE1_ABL=$n -c blobnet_mfma.hip -o /tmp/isa/abl$n.o 2>&1 | grep -E "error" -A5
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../ab_tmp/abl$n.so build/ctx.hip.o build/bboxcc.hip.o build/blobnet.hip.o \
        /tmp/isa/abl$n.o build/pipe.hip.o build/hostlib.cpp.o build/h264_front.cpp.o build/h264_cabac.cpp.o &
done
wait
ls -la ../../ab_tmp
