#!/bin/bash
# Critical-path probes of the T1M step (StepEngine._probe): one bench run per label with a ~20 us spin node at
# that point of the step graph, beside a run without.  A step that grows by ~20 us has the point on its critical
# path, one that does not has slack there, one that gets faster has a branch that starts too early.
# Run on the GPU box from the repo root:  bash tools/critical_probe.sh [cycles, default 40000]
set -e
cyc=${1:-40000}
out=gpurun_out/probe
mkdir -p $out
rm -f $out/res.log
run() { tag=$1; shift; env "$@" timeout -k 10 120 python3 bench.py --no-roofline --no-cpu-baseline --steps 300 --warmup 10 2>$out/$tag.err | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/$tag /" >> $out/res.log; }
run none DPCGAN_DELAY=
for l in d_inputs g_forward_done real_rows_branch early_lane_start early_lane_end d_forward_done d_weight_grads d_grads_done g_d_forward_done g_dhg_done g_out_dweight_branch g_grads_done; do
  run $l DPCGAN_DELAY=$l:$cyc
done
run none_again DPCGAN_DELAY=
cat $out/res.log
