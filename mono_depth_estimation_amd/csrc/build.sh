#!/bin/bash
# Build libmde_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libmde_hip.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off"
mkdir -p build
objs=()
pids=()
for src in *.hip; do
    obj=build/${src%.hip}.o
    objs+=("$obj")
    if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ mde_common.h -nt "$obj" ] || [ ../../include/mde_hip.h -nt "$obj" ]; then
        $HIPCC $FLAGS -c "$src" -o "$obj" &
        pids+=($!)
    fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o $OUT "${objs[@]}"
# every kernel referenced by a host stub must exist (hipcc can drop a stub silently)
if nm "$OUT" | grep -q " U .*__device_stub__"; then echo "ERROR: undefined kernel stubs in $OUT" >&2; exit 1; fi
echo "built $(readlink -f $OUT)"

# the same sources with IEEE half as the 16-bit storage type (mde_common.h, MDE_ACT_F16): libmde_hip_f16.so
OUT16=../libmde_hip_f16.so
mkdir -p build_f16
objs=()
pids=()
for src in *.hip; do
    obj=build_f16/${src%.hip}.o
    objs+=("$obj")
    if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ mde_common.h -nt "$obj" ] || [ ../../include/mde_hip.h -nt "$obj" ]; then
        $HIPCC $FLAGS -DMDE_ACT_F16=1 -c "$src" -o "$obj" &
        pids+=($!)
    fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o $OUT16 "${objs[@]}"
if nm "$OUT16" | grep -q " U .*__device_stub__"; then echo "ERROR: undefined kernel stubs in $OUT16" >&2; exit 1; fi
echo "built $(readlink -f $OUT16)"
