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
for c in FETCH_SIZE WRITE_SIZE MFMA; do
  [ -f $P/pmc_$c/pmc_counter_summary.csv ] || python3 tools/pmc_reduce.py $P/pmc_$c/pmc_counter_collection.csv --keep
  cp $P/pmc_$c/pmc_counter_summary.csv profiles/${tag}_pmc_${c}_${name}_counter_summary.csv
done
cp $P/step_timeline.txt profiles/${tag}_step_timeline_${name}.txt
python tools/kernel_roofline.py --trace $P/trace/bench_kernel_trace.csv --bench $P/bench_under_rocprof.json.log \
  --stats-name ${tag}_bench_${name}_kernel_stats.csv --fetch $P/pmc_FETCH_SIZE/pmc_counter_summary.csv \
  --write $P/pmc_WRITE_SIZE/pmc_counter_summary.csv --mfma $P/pmc_MFMA/pmc_counter_summary.csv \
  --out profiles/${tag}_kernel_roofline${sfx}.json \
  --command "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline $(cat $P/bench_args.txt)"
cp $P/trace_serial/bench_kernel_stats.csv profiles/${tag}_bench_${name}_serial_kernel_stats.csv
cp $P/bench_serial_under_rocprof.json.log profiles/${tag}_bench_${name}_serial_under_rocprof.json.log
python tools/kernel_roofline.py --trace $P/trace_serial/bench_kernel_trace.csv --bench $P/bench_serial_under_rocprof.json.log \
  --stats-name ${tag}_bench_${name}_serial_kernel_stats.csv --out profiles/${tag}_kernel_roofline${sfx}_serial.json \
  --command "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --serial --no-cpu-baseline $(cat $P/bench_args.txt)" > /dev/null
