#!/usr/bin/env bash
# Build liblp_hip.so (gfx950) in-tree.  Usage: build.sh [emu]
#   (no arg)  hipcc --offload-arch=gfx950 -> ../liblp_hip.so      (the product library)
#   emu       host clang++ against tests/hipemu -> ../../tests/hipemu/liblp_emu.so  (CPU logic tests only)
set -euo pipefail
cd "$(dirname "$0")"
ROCM=${ROCM_PATH:-/opt/rocm}
SRCS=(api.hip decode.hip heatmap.hip kploss.hip conv.hip bn.hip optim.hip vit.hip attn.hip frames.hip fp32.hip vit_f32.hip)
mode=${1:-hip}
if [ "$mode" = emu ]; then
  out=../../tests/hipemu
  mkdir -p "$out/obj"
  exec 9>"$out/obj/.lock"; flock 9   # concurrent callers (pytest-xdist workers) take turns; the link below is atomic
  objs=()
  for s in "${SRCS[@]}"; do
    [ -f "$s" ] || continue
    o="$out/obj/${s%.hip}.o"
    if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ lp_common.h -nt "$o" ] || { [ "$s" = conv.hip ] && { [ conv_pipe.h -nt "$o" ] || [ conv_res2d.h -nt "$o" ] || [ conv_stem_wgrad.h -nt "$o" ]; }; } || [ ../../include/lp_hip.h -nt "$o" ] || [ "$out/hip/hip_runtime.h" -nt "$o" ]; then
      "$ROCM/lib/llvm/bin/clang++" -x c++ -std=c++17 -O2 -fPIC -Wno-psabi -Wno-unused-value -I"$out" -c "$s" -o "$o" &
    fi
    objs+=("$o")
  done
  wait
  "$ROCM/lib/llvm/bin/clang++" -shared -o "$out/liblp_emu.so.$$" "${objs[@]}" -lpthread
  mv -f "$out/liblp_emu.so.$$" "$out/liblp_emu.so"
  echo "built $out/liblp_emu.so"
else
  # A/B builds: LP_BUILD_TAG=name LP_BUILD_FLAGS="-DX=1" build.sh  ->  ../../build/liblp_hip_name.so (objects in obj_name/); load it with LP_HIP_LIB
  tag=${LP_BUILD_TAG:-}
  od=obj${tag:+_$tag}
  mkdir -p "$od"
  objs=()
  for s in "${SRCS[@]}"; do
    [ -f "$s" ] || continue
    o="$od/${s%.hip}.o"
    if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ lp_common.h -nt "$o" ] || { [ "$s" = conv.hip ] && { [ conv_pipe.h -nt "$o" ] || [ conv_res2d.h -nt "$o" ] || [ conv_stem_wgrad.h -nt "$o" ]; }; } || [ ../../include/lp_hip.h -nt "$o" ]; then
      "$ROCM/bin/hipcc" --offload-arch=gfx950 -O3 -std=c++17 -fPIC ${LP_BUILD_FLAGS:-} -c "$s" -o "$o" &
    fi
    objs+=("$o")
  done
  wait
  out=../liblp_hip.so
  if [ -n "$tag" ]; then mkdir -p ../../build; out=../../build/liblp_hip_$tag.so; fi
  "$ROCM/bin/hipcc" --offload-arch=gfx950 -shared -fPIC -o "$out" "${objs[@]}"
  echo "built $(cd "$(dirname "$out")" && pwd)/$(basename "$out")"
fi
