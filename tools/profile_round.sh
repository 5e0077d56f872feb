#!/bin/bash
# The rocprofv3 runs behind profiles/r02_*: kernel trace + stats of the default bench command, then the counter
# passes (each alone, no trace domains beside --pmc), then tools/kernel_roofline.py.  Run on the GPU box from the
# repo root:  bash tools/profile_round.sh r02
set -e
tag=${1:-r02}
out=gpurun_out/${tag}prof
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$out" && mkdir -p "$out"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o bench -- python3 bench.py --no-cpu-baseline > "$out/bench_under_rocprof.json.log" 2> "$out/trace.err"
# the same step as ONE chain (--serial: no side branches, kernels never overlap): per-kernel durations without co-runners
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_serial" -o bench -- python3 bench.py --serial --no-cpu-baseline > "$out/bench_serial_under_rocprof.json.log" 2> "$out/trace_serial.err"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 180 rocprofv3 --pmc $c --output-format csv -d "$out/pmc_$c" -o pmc -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > "$out/pmc_$c.json.log" 2> "$out/pmc_$c.err"
done
timeout -s KILL 180 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_MFMA" -o pmc -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > "$out/pmc_MFMA.json.log" 2> "$out/pmc_MFMA.err"
timeout -k 10 300 python bench.py > "$out/bench_plain.json.log" 2> "$out/bench_plain.err"
find "$out" -name '*.csv' | head -20
# (then, back in the build container:  bash tools/profile_collect.sh r02)
