#!/bin/bash
# The rocprofv3 runs behind profiles/<tag>_*: kernel trace + stats of a bench command, the same step as one chain
# (--serial), then the counter passes (each alone, no trace domains beside --pmc).  Run on the GPU box from the repo
# root:   bash tools/profile_round.sh r03 [NAME [bench.py arguments of that workload ...]]
#   bash tools/profile_round.sh r03                                   (config 3: the default T1M workload)
#   bash tools/profile_round.sh r03 T10M --table T10M --dims 1024,1024,1024 --steps 60
#   bash tools/profile_round.sh r03 TWIDE --table TWIDE --d-steps 10 --steps 12 --warmup 3
# then, back in the build container:  bash tools/profile_collect.sh r03 [NAME]
set -e
tag=${1:-r03}
name=${2:-T1M}
shift || true
shift || true
out=gpurun_out/${tag}prof_${name}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$out" && mkdir -p "$out"
echo "$@" > "$out/bench_args.txt"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o bench -- python3 bench.py --no-cpu-baseline "$@" > "$out/bench_under_rocprof.json.log" 2> "$out/trace.err"
# the same step as ONE chain (--serial: no side branches, kernels never overlap): per-kernel durations without co-runners
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_serial" -o bench -- python3 bench.py --serial --no-cpu-baseline "$@" > "$out/bench_serial_under_rocprof.json.log" 2> "$out/trace_serial.err"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d "$out/pmc_$c" -o pmc -- python3 bench.py "$@" --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > "$out/pmc_$c.json.log" 2> "$out/pmc_$c.err"
  python3 tools/pmc_reduce.py "$out/pmc_$c/pmc_counter_collection.csv"
done
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_MFMA" -o pmc -- python3 bench.py "$@" --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > "$out/pmc_MFMA.json.log" 2> "$out/pmc_MFMA.err"
python3 tools/pmc_reduce.py "$out/pmc_MFMA/pmc_counter_collection.csv"
# (the CPU baseline is config 3's: one oracle step at the wide configs takes minutes)
nocpu=""; [ "$name" != "T1M" ] && nocpu="--no-cpu-baseline"
timeout -k 10 400 python bench.py "$@" $nocpu > "$out/bench_plain.json.log" 2> "$out/bench_plain.err"
# the full traces stay in gpurun_out/ (tens of MB); what is judged is cut down here so that it fits the 64 MiB that travel back
dsteps=$(python3 -c "import json,sys; print(json.loads([l for l in open('$out/bench_under_rocprof.json.log') if l.startswith('{')][-1])['config']['d_steps'])")
python3 tools/step_timeline.py "$out/trace/bench_kernel_trace.csv" 8 "$dsteps" > "$out/step_timeline.txt"
find "$out" -name '*.csv' | head -20
