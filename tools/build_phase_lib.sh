#!/bin/bash
# Developer helper: builds tools/abl/libPHASE.so = the library with -DPHASE_TIMING (per-phase wall clock of enc_mfma's item
# loop; read with tools/phase_timing.py after copying it over cova_amd/libcovahip.so ON THE GPU BOX).
cd "$(dirname "$0")/../cova_amd/csrc" || exit 1
mkdir -p ../../tools/abl /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-unused-value -I../../include -I. -fno-honor-nans \
    -mllvm -pragma-unroll-threshold=1000000 ${PHASE_DEFS:--DPHASE_TIMING} -c blobnet_mfma.hip -o /tmp/isa/phase.o 2>&1 | grep -E "error" -A5
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/abl/${PHASE_OUT:-libPHASE.so} build/ctx.hip.o build/bboxcc.hip.o build/blobnet.hip.o \
    /tmp/isa/phase.o build/pipe.hip.o build/hostlib.cpp.o build/h264_front.cpp.o build/h264_cabac.cpp.o
