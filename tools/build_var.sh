#!/bin/bash
# usage: tools/build_var.sh NAME [-DMACRO=..]...  -> build_ab/var/NAME.so (an experiment build of the same ABI; RMI_HIP_LIB=... selects it)
# rmi_hip.hip is compiled with the macros; pipeline 5's unit is taken from the in-tree build (rmi_amd/build/rmi_scan.o).
NAME=$1; shift
mkdir -p build_ab/var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function "$@" \
  -c rmi_amd/csrc/rmi_hip.hip -o build_ab/var/$NAME.o 2> build_ab/var/$NAME.log || { tail -5 build_ab/var/$NAME.log; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/var/$NAME.so build_ab/var/$NAME.o rmi_amd/build/rmi_scan.o -ldl && rm -f build_ab/var/$NAME.o
