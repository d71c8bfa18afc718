#!/bin/bash
# Development aid: build a variant of the pipeline-5 translation unit with extra -D flags and link it against the current rmi_hip.o:
#   tools/scan_variant.sh <tag> [-DRMI_SC_...]   ->  build_ab/librmi_hip_<tag>.so   (use with RMI_HIP_LIB=...)
set -e
tag=$1; shift
mkdir -p build_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c rmi_amd/csrc/rmi_scan.hip -o build_ab/rmi_scan_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/librmi_hip_$tag.so rmi_amd/build/rmi_hip.o build_ab/rmi_scan_$tag.o -ldl
echo build_ab/librmi_hip_$tag.so
