#!/bin/bash
# Copies the judged summaries of gpurun_out/<tag>prof (tools/profile_round.sh) into profiles/ and builds the
# per-kernel roofline table.
set -e
tag=${1:-r02}
P=gpurun_out/${tag}prof
cp $P/trace/bench_kernel_stats.csv profiles/${tag}_bench_T1M_kernel_stats.csv
cp $P/bench_under_rocprof.json.log profiles/${tag}_bench_T1M_under_rocprof.json.log
cp $P/bench_plain.json.log profiles/${tag}_bench_T1M.json.log
cp $P/pmc_FETCH_SIZE/pmc_counter_collection.csv profiles/${tag}_pmc_FETCH_SIZE_counter_collection.csv
cp $P/pmc_WRITE_SIZE/pmc_counter_collection.csv profiles/${tag}_pmc_WRITE_SIZE_counter_collection.csv
python tools/step_timeline.py $P/trace/bench_kernel_trace.csv 30 > profiles/${tag}_step_timeline.txt
python tools/kernel_roofline.py --trace $P/trace/bench_kernel_trace.csv --bench $P/bench_under_rocprof.json.log \
  --stats-name ${tag}_bench_T1M_kernel_stats.csv --fetch $P/pmc_FETCH_SIZE/pmc_counter_collection.csv \
  --write $P/pmc_WRITE_SIZE/pmc_counter_collection.csv --mfma $P/pmc_MFMA/pmc_counter_collection.csv \
  --out profiles/${tag}_kernel_roofline.json
cp $P/trace_serial/bench_kernel_stats.csv profiles/${tag}_bench_T1M_serial_kernel_stats.csv
cp $P/bench_serial_under_rocprof.json.log profiles/${tag}_bench_T1M_serial_under_rocprof.json.log
python tools/kernel_roofline.py --trace $P/trace_serial/bench_kernel_trace.csv --bench $P/bench_serial_under_rocprof.json.log \
  --stats-name ${tag}_bench_T1M_serial_kernel_stats.csv --out profiles/${tag}_kernel_roofline_serial.json \
  --command 'rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --serial --no-cpu-baseline' > /dev/null
