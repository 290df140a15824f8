#!/bin/bash
# An experiment build of the inflate kernels: tools/build_inflate_variant.sh NAME -DGW_EXPER=1 [more flags]
# -> bamqc_amd/libbamqc_gpu_NAME.so (every other object from the regular build); use it with BQC_LIB_PATH=... python tools/inflate_scaling.py
set -e
cd "$(dirname "$0")/../bamqc_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result "$@" -c gpu_inflate.hip -o build/gpu_inflate_$name.o
objs=$(ls build/*.o | grep -v "gpu_inflate" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbamqc_gpu_$name.so $objs build/gpu_inflate_$name.o -lz -lpthread -ldl
echo built ../libbamqc_gpu_$name.so
