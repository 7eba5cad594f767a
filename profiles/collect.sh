#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box:  bash profiles/collect.sh r2
#   kernel stats  : rocprofv3 --kernel-trace --stats      (3 steps of the default workload, 30 M x 150)
#   HBM traffic   : two separate --pmc passes, FETCH_SIZE and WRITE_SIZE (1 step), as MI355X_MICROARCH.md prescribes
#   modes         : kernel stats of bench.py's e2e_host / ebwt_modes legs (k_bfs_*, FASTQ kernels) at 30 M x 150
#   config 1 size : kernel stats + bench line at 1 M x 100
# Summaries land in gpurun_out/prof_$1/ (scratch); profiles/<round>/ holds the copies that are committed.
set -e
R=${1:-r2}
O=gpurun_out/prof_$R
mkdir -p $O
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
B="python3 bench.py --workload 30Mx150 --no-cpu --no-dropin"
rocprofv3 --kernel-trace --stats -d $O -o kt --output-format csv -- $B --no-e2e --steps 3 --warmup 1 > $O/bench_under_rocprof_30Mx150.json 2> $O/kt.log
echo "kernel stats done"
rocprofv3 --pmc FETCH_SIZE -d $O -o pf --output-format csv -- $B --no-e2e --steps 1 --warmup 0 > $O/pf.json 2> $O/pf.log
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE -d $O -o pw --output-format csv -- $B --no-e2e --steps 1 --warmup 0 > $O/pw.json 2> $O/pw.log
echo "write done"
rocprofv3 --kernel-trace --stats -d $O -o modes --output-format csv -- $B --steps 1 --warmup 1 > $O/bench_modes_30Mx150.json 2> $O/modes.log
echo "modes done"
rocprofv3 --kernel-trace --stats -d $O -o kt1M --output-format csv -- python3 bench.py --workload 1Mx100 --no-cpu --no-dropin --no-e2e --steps 20 --warmup 3 > $O/bench_under_rocprof_1Mx100.json 2> $O/kt1M.log
python3 bench.py --workload 1Mx100 --steps 20 --warmup 3 > $O/bench_1Mx100.json 2> $O/bench_1M.log
echo "1M done"
find $O -name "*.csv" | head -30
