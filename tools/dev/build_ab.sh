#!/bin/bash
# tools/diag/lib_ab.so = the library built with extra flags ($@), e.g. -DTPIV_COOP=0, for same-box A/B runs
set -e
cd "$(dirname "$0")/../../torchpiv_amd/csrc"
rm -rf ../../build/obj_ab        # make compares time stamps, not flags: a build with other -D flags must start from nothing
make -j8 OBJDIR=../../build/obj_ab OUT=../../tools/diag/lib_ab.so CXXFLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=fast-honor-pragmas -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None -Wno-unused-result $*" 2>&1 | grep -E "error|Error" || true
ls -la ../../tools/diag/lib_ab.so
