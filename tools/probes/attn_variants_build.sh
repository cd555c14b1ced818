#!/bin/bash
# Builds variants of csrc/attention.hip into wan2.1-quantization_amd/lib/variants/ for tools/ab_attn_variants.py:
#   attn_variants_build.sh name1 "<hipcc -D flags>" name2 "<flags>" ...      (a flag string may be empty)
# Every variant is the whole library (all csrc/*.hip; only attention.hip sees the flags' effect), so any entry point can be A/B'd.
set -e
PKG=$(cd "$(dirname "$0")/../../wan2.1-quantization_amd" && pwd)
V=$PKG/lib/variants; mkdir -p "$V"; O=$PKG/build
python3 "$PKG/build.py" > /dev/null   # objects of the other files
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-result $flags -c "$PKG/csrc/attention.hip" -o "$V/attention_$name.o"
  objs=$(ls "$O"/*.o | grep -v "/attention.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$V/lib_$name.so" $objs "$V/attention_$name.o"
  rm -f "$V/attention_$name.o"
  echo "built lib_$name.so ($flags)"
done
