#!/bin/bash
# developer tool: variants/libnam_hip_<tag>.so = the library with ONE kernel file rebuilt under extra -D knobs
#   bash scripts/build_variant.sh <tag> <kernel_file.hip> "<extra flags>"
set -e
cd "$(dirname "$0")/.."
TAG=$1; SRC=$2; shift; shift
mkdir -p variants/obj
O=variants/obj/${TAG}_$(basename $SRC .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form $@ -c -o $O neuralampmodelercore_amd/csrc/$SRC
OBJS=$(ls neuralampmodelercore_amd/lib/obj/*.o | grep -v "/$(basename $SRC .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o variants/libnam_hip_$TAG.so $OBJS $O
echo built variants/libnam_hip_$TAG.so
