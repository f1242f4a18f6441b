#!/bin/bash
# A/B of the conv tile variants on single shapes:  tools/sweep_tiles.sh "H W N Cin Cout k" ...
for shape in "$@"; do
  for t in auto 256x256 192x256 128x128 128x64; do
    if [ $t = auto ]; then unset MDE_CONV_TILE; else export MDE_CONV_TILE=$t; fi
    echo -n "$t: "; python tools/conv_microbench.py fwd $shape 30 2>&1 | grep -v amdgpu.ids
  done
done
