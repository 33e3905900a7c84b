#!/bin/bash
# Build a variant of the library for an in-box A/B:  tools/build_variant.sh <tag> [-DNAME=VALUE ...]
# -> rtl-wmbus_amd/libwmbus_hip_<tag>.so (git-ignored; travels to the GPU box with the snapshot).
# Use with tools/gpu_visit.sh:  tools/gpu_visit.sh TAG "libwmbus_hip_<tag>.so|" "-|"
#
#
set -e
tag=$1; shift
here=$(cd "$(dirname "$0")/.." && pwd)/rtl-wmbus_amd
make -s -C "$here" wm_decoder.o >/dev/null 2>&1 || make -s -C "$here" >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -w "$@" -c -o /tmp/wm_api_$tag.o "$here/csrc/wm_api.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$here/libwmbus_hip_$tag.so" /tmp/wm_api_$tag.o "$here/wm_decoder.o" -lpthread
ls -la "$here/libwmbus_hip_$tag.so"
