#!/bin/bash
# An experiment build of ONE kernel file: tools/build_variant.sh NAME FILE.hip -DFLAG=1 [more flags]
# -> bamqc_amd/libbamqc_gpu_NAME.so (every other object from the regular build); use it with BQC_LIB_PATH=... python tools/...
set -e
cd "$(dirname "$0")/../bamqc_amd/csrc"
name=$1; file=$2; shift; shift
base=${file%.*}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result "$@" -c $file -o build/variant_${base}_$name.o
objs=$(ls build/*.o | grep -v "build/$base\.\|variant_\|gpu_inflate_" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbamqc_gpu_$name.so $objs build/variant_${base}_$name.o -lz -lpthread -ldl
echo built ../libbamqc_gpu_$name.so
