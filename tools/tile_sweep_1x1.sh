#!/bin/bash
cd "$(dirname "$0")/.."
for shape in "120 160 32 64 256 1" "60 80 32 128 512 1" "30 40 32 256 1024 1" "30 40 32 1024 256 1" "120 160 32 256 64 1" "15 20 32 512 2048 1" "60 80 32 512 128 1"; do
  for tile in 256x256 192x256 128x128 128x64; do
    echo -n "tile=$tile  "
    MDE_CONV_PP=0 MDE_CONV_TILE=$tile python tools/conv_microbench.py fwd $shape 50 2>/dev/null
  done
done
