#!/bin/bash
# Builds probe variants of libdynavsr_hip.so that differ only in mdcn_bwd.hip's compile-time switches (results of the probe
# builds are WRONG; they bound what a term of the sampling phase costs).  Usage: tools/dcnb_variants.sh NOATOM NOCVT ...
set -e
cd "$(dirname "$0")/../dynavsr_amd"
python build.py > /dev/null
for v in "$@"; do
  defs=""
  for d in ${v//+/ }; do defs="$defs -DDCNB_$d"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $defs -c csrc/mdcn_bwd.hip -o /tmp/dcnb_$v.o
  objs=$(ls csrc/_obj/*.o | grep -v mdcn_bwd.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libdynavsr_hip_dcnb_$v.so $objs /tmp/dcnb_$v.o
  echo built libdynavsr_hip_dcnb_$v.so
done
