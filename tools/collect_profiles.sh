#!/bin/bash
# Copy the artefacts of tools/gpu_round_end.sh (merged into gpurun_out/ by gpurun) to profiles/ under this round's names.
R=${1:-r04}; O=gpurun_out
for m in default dccrn_large fullsubnet; do
  s=$([ $m = default ] && echo "" || echo "_$m")
  [ -f $O/fin_pmc_$m.json ] && cp $O/fin_pmc_$m.json profiles/${R}_pmc_traffic$s.json
  [ -f $O/fin_kernel_stats_$m.csv ] && cp $O/fin_kernel_stats_$m.csv profiles/${R}_kernel_stats_$m.csv
done
[ -f $O/fin_timeline_default.txt ] && cp $O/fin_timeline_default.txt profiles/${R}_timeline.txt
[ -f $O/fin_timeline_fullsubnet.txt ] && cp $O/fin_timeline_fullsubnet.txt profiles/${R}_timeline_fullsubnet.txt
for b in default driver B64 dccrn_large fullsubnet pmsqe lms; do
  [ -f $O/fin_bench_$b.log ] && tail -1 $O/fin_bench_$b.log > profiles/${R}_bench_$b.json
done
[ -f $O/bf16_parity.json ] && cp $O/bf16_parity.json profiles/${R}_bf16_parity.json
ls -la profiles/${R}_* | awk '{print $5, $9}'
