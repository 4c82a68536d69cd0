#!/bin/bash
# Builds a variant of the library for same-box A/B runs: tools/ab_build.sh NAME "FLAGS" -> ab/liblf_mkd_NAME.so
# (FLAGS are added to the describe kernel's two objects, e.g. -DLF_KP_ABLATE_PRODUCER; use with LF_MKD_LIB=ab/...)
set -e
cd "$(dirname "$0")/../local-features_amd"
NAME=$1; FLAGS=$2
mkdir -p ../ab obj_ab
C="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wall -Wno-unused-function"
$C $FLAGS -mllvm -amdgpu-sched-strategy=max-ilp -c -o obj_ab/d_$NAME.o csrc/mkd_describe.hip &
$C $FLAGS -DLF_DESCRIBE_KP -Rpass-analysis=kernel-resource-usage -c -o obj_ab/k_$NAME.o csrc/mkd_describe.hip 2> obj_ab/k_$NAME.remarks &
wait
grep -c "ScratchSize \[bytes/lane\]: [1-9]" obj_ab/k_$NAME.remarks || true
OTHERS=$(ls obj/*.o | grep -v "mkd_describe")
$C -shared -o ../ab/liblf_mkd_$NAME.so obj_ab/d_$NAME.o obj_ab/k_$NAME.o $OTHERS
echo built ab/liblf_mkd_$NAME.so
