#!/bin/bash
# Build the library as of a git revision into build_ab/old.so (for tools/ab.sh).  usage: tools/build_ref.sh [rev=HEAD]
REV=${1:-HEAD}
D=build_ab/old; rm -rf $D; mkdir -p $D/rmi_amd/csrc $D/include
for f in rmi_hip.hip rmi_kernels.hip.h rmi_stream.hip.h rmi_device.hip.h rmi_root_host.h; do git show $REV:rmi_amd/csrc/$f > $D/rmi_amd/csrc/$f; done
git show $REV:include/rmi_hip.h > $D/include/rmi_hip.h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-function -o build_ab/old.so $D/rmi_amd/csrc/rmi_hip.hip 2>&1 | grep -v "^/" | head -5
ls -la build_ab/old.so
