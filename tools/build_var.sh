#!/bin/bash
# usage: tools/build_var.sh NAME [-DMACRO=..]...  -> build_ab/var/NAME.so (an experiment build of the same ABI; RMI_HIP_LIB=... selects it)
NAME=$1; shift
mkdir -p build_ab/var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function -ldl "$@" \
  -o build_ab/var/$NAME.so rmi_amd/csrc/rmi_hip.hip
