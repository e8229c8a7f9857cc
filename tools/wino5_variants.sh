#!/bin/bash
# Builds probe variants of libdynavsr_hip.so that differ only in conv2d_wino5.hip's compile-time switches (results of the
# probe builds are WRONG; they bound what a term of the chunk loop costs).  Usage: tools/wino5_variants.sh NOA NOPROD NOMMA ...
set -e
cd "$(dirname "$0")/../dynavsr_amd"
python build.py > /dev/null
for v in "$@"; do
  defs=""
  for d in ${v//+/ }; do defs="$defs -DW5_$d"; done; defs=${defs//ARING6/ARING=6}; defs=${defs//ARING4/ARING=4}
  slp="-fno-slp-vectorize"; case "$v" in *SLP*) slp="";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $slp $defs -c csrc/conv2d_wino5.hip -o /tmp/wino5_$v.o
  objs=$(ls csrc/_obj/*.o | grep -v conv2d_wino5.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libdynavsr_hip_w5_$v.so $objs /tmp/wino5_$v.o
  echo built libdynavsr_hip_w5_$v.so
done
