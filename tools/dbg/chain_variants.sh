#!/bin/bash
# Measurement builds of the chained 1x1 kernel: csrc/chain.hip compiled with -DCHAIN_DBG=<bits> (1 no MFMAs, 2 no residual DMA / mid stores, 4 no
# epilogue arithmetic) and any further -D given in EXTRA, linked with the product's other objects (build/obj, from __graft_entry__.build()) into
# tools/experiments/libchain_<tag>.so; run tools/chainexp.py with MI355SEG_LIB pointing at one.   usage: tools/dbg/chain_variants.sh tag:flags ...
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
cd "$root/rnd_semantic_segmentation_amd/csrc"
objs=$(ls "$root"/build/obj/*.o | grep -v chain.o)
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $flags -c chain.hip -o /tmp/chain_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/experiments/libchain_$tag.so" $objs /tmp/chain_$tag.o -ldl
  echo "built libchain_$tag.so ($flags)"
done
