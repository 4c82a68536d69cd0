#!/bin/bash
# A/B builds of the matcher: tools/ab_match.sh build NAME "FLAGS" ...   (here, on the CPU box: ab/liblf_NAME.so)
#                            tools/ab_match.sh run NAME ...             (on the GPU box: each timed by tools/ab_match.py)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  shift; mkdir -p ab
  make -s -j8 -C local-features_amd
  while [ $# -gt 1 ]; do
    name=$1; flags=$2; shift 2
    (cd local-features_amd && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize $flags -c csrc/mkd_match.hip -o ../ab/match_$name.o &&
     /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o ../ab/liblf_$name.so $(ls obj/*.o | grep -v mkd_match) ../ab/match_$name.o)
    echo "built ab/liblf_$name.so ($flags)"
  done
else
  shift
  for name in "$@"; do
    echo "== $name"
    LF_MKD_LIB=$(realpath ab/liblf_$name.so) timeout -k 10 200 python tools/ab_match.py ${SHAPES:-1048576 1048576} 2>/dev/null
  done
fi
