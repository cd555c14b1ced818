#!/bin/bash
# Builds variants of csrc/attention.hip (one sed edit each) into wan2.1-quantization_amd/lib/variants/ for tools/ab_attn_variants.py.
set -e
PKG=$(cd "$(dirname "$0")/../../wan2.1-quantization_amd" && pwd)
V=$PKG/lib/variants; mkdir -p "$V"; T=$(mktemp -d)
build() {  # name, sed expression
  rm -rf "$T/csrc"; cp -r "$PKG/csrc" "$T/csrc"
  [ -n "$2" ] && sed -i "$2" "$T/csrc/attention.hip"
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -shared -I "$PKG/csrc" -o "$V/lib_$1.so" "$T"/csrc/*.hip
}
build base ""
build prio2 's/__builtin_amdgcn_s_setprio(1);/__builtin_amdgcn_s_setprio(2);/'
build prio3 's/__builtin_amdgcn_s_setprio(1);/__builtin_amdgcn_s_setprio(3);/'
build noprio 's/if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);//'
build lazy20 's/__any(mx > 6.0f)/__any(mx > 20.0f)/'
ls "$V"
