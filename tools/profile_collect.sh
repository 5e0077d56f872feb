#!/bin/bash
# Copies the judged summaries of gpurun_out/<tag>prof_<NAME> (tools/profile_round.sh) into profiles/ and builds the
# per-kernel roofline tables.   bash tools/profile_collect.sh r03 [NAME]
set -e
tag=${1:-r03}
name=${2:-T1M}
P=gpurun_out/${tag}prof_${name}
sfx=""; [ "$name" != "T1M" ] && sfx="_${name}"
cp $P/trace/bench_kernel_stats.csv profiles/${tag}_bench_${name}_kernel_stats.csv
cp $P/bench_under_rocprof.json.log profiles/${tag}_bench_${name}_under_rocprof.json.log
cp $P/bench_plain.json.log profiles/${tag}_bench_${name}.json.log
cp $P/pmc_FETCH_SIZE/pmc_counter_collection.csv profiles/${tag}_pmc_FETCH_SIZE_${name}_counter_collection.csv
cp $P/pmc_WRITE_SIZE/pmc_counter_collection.csv profiles/${tag}_pmc_WRITE_SIZE_${name}_counter_collection.csv
cp $P/step_timeline.txt profiles/${tag}_step_timeline_${name}.txt
python tools/kernel_roofline.py --trace $P/trace/bench_kernel_trace.csv --bench $P/bench_under_rocprof.json.log \
  --stats-name ${tag}_bench_${name}_kernel_stats.csv --fetch $P/pmc_FETCH_SIZE/pmc_counter_collection.csv \
  --write $P/pmc_WRITE_SIZE/pmc_counter_collection.csv --mfma $P/pmc_MFMA/pmc_counter_collection.csv \
  --out profiles/${tag}_kernel_roofline${sfx}.json \
  --command "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline $(cat $P/bench_args.txt)"
cp $P/trace_serial/bench_kernel_stats.csv profiles/${tag}_bench_${name}_serial_kernel_stats.csv
cp $P/bench_serial_under_rocprof.json.log profiles/${tag}_bench_${name}_serial_under_rocprof.json.log
python tools/kernel_roofline.py --trace $P/trace_serial/bench_kernel_trace.csv --bench $P/bench_serial_under_rocprof.json.log \
  --stats-name ${tag}_bench_${name}_serial_kernel_stats.csv --out profiles/${tag}_kernel_roofline${sfx}_serial.json \
  --command "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --serial --no-cpu-baseline $(cat $P/bench_args.txt)" > /dev/null
