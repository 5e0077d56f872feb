#!/bin/bash
# Builds the MEASUREMENT variant of the library (gemm.hip with -DDPC_GEMM_STAMP: in-kernel cycle stamps) into
# tools/_stamp/libdpcgan_stamp.so.  Run here (hipcc cross-compiles); tools/gemm_stamp.py loads it on the GPU box.
# With arguments: one more library per argument N, tools/_stamp/libdpcgan_stamp_eN.so, built with -DDPC_EXP=N (a class
# of instructions left out of the K loop: 1 global loads, 2 LDS stores, 4 fragment reads, 8 the barrier; sums combine).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_stamp
make -C dp_cgans_amd/csrc -j8 >/dev/null
objs=$(ls dp_cgans_amd/csrc/*.o | grep -v gemm.o)
build() {   # $1: extra define, $2: output suffix
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DDPC_GEMM_STAMP $1 -Idp_cgans_amd/csrc \
      -c dp_cgans_amd/csrc/gemm.hip -o tools/_stamp/gemm_stamp$2.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 tools/_stamp/gemm_stamp$2.o $objs -o tools/_stamp/libdpcgan_stamp$2.so
  echo built tools/_stamp/libdpcgan_stamp$2.so
}
build "" "" &
for e in "$@"; do build "-DDPC_EXP=$e" "_e$e" & done
wait
