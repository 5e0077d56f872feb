#!/bin/bash
# Builds the MEASUREMENT variant of the library (gemm.hip with -DDPC_GEMM_STAMP: in-kernel cycle stamps) into
# tools/_stamp/libdpcgan_stamp.so.  Run here (hipcc cross-compiles); tools/gemm_stamp.py loads it on the GPU box.
set -e
cd "$(dirname "$0")/.."
make -C dp_cgans_amd/csrc -j8 >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DDPC_GEMM_STAMP -Idp_cgans_amd/csrc \
    -c dp_cgans_amd/csrc/gemm.hip -o tools/_stamp/gemm_stamp.o
objs=$(ls dp_cgans_amd/csrc/*.o | grep -v gemm.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 tools/_stamp/gemm_stamp.o $objs -o tools/_stamp/libdpcgan_stamp.so
echo built tools/_stamp/libdpcgan_stamp.so
