#!/bin/bash
# Build libmde_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libmde_hip.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off"
# Files whose kernels hipcc's SLP vectoriser would give packed-fp32 instructions with a lane-crossing op_sel
# (v_pk_add_f32 ... op_sel:[0,1]): on gfx950 the low lane of such an instruction came out wrong whenever a wave of another
# kernel shared the SIMD (check_isa.py).  Without the SLP pass the same arithmetic is scalar or
# packed in natural lane order.  check_isa.py (below) fails the build if any kernel of the library still has one.
NOSLP=" conv_gemm conv_small pointwise losses stdepth_loss vnl_losses "
file_flags() { case "$NOSLP" in *" $1 "*) echo "-fno-slp-vectorize";; esac; }
mkdir -p build
objs=()
pids=()
for src in *.hip; do
    obj=build/${src%.hip}.o
    objs+=("$obj")
    if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ mde_common.h -nt "$obj" ] || [ ../../include/mde_hip.h -nt "$obj" ] || [ build.sh -nt "$obj" ]; then
        $HIPCC $FLAGS $(file_flags "${src%.hip}") -c "$src" -o "$obj" &
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
    if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ mde_common.h -nt "$obj" ] || [ ../../include/mde_hip.h -nt "$obj" ] || [ build.sh -nt "$obj" ]; then
        $HIPCC $FLAGS $(file_flags "${src%.hip}") -DMDE_ACT_F16=1 -c "$src" -o "$obj" &
        pids+=($!)
    fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o $OUT16 "${objs[@]}"
if nm "$OUT16" | grep -q " U .*__device_stub__"; then echo "ERROR: undefined kernel stubs in $OUT16" >&2; exit 1; fi
echo "built $(readlink -f $OUT16)"
# no kernel of either build may hold a packed-fp32 instruction with op_sel (see NOSLP above)
python3 check_isa.py "$OUT" "$OUT16"
