#!/bin/bash
# Round profile set (run on the GPU box): bench line, rocprofv3 kernel stats of the same command,
# counter passes.  usage: collect_profiles.sh <tag>
set -u
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py --steps 16 --warmup 2 > $out/bench.json 2> $out/bench.err || { echo bench failed; tail -5 $out/bench.err; exit 1; }
cat $out/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-fused > $out/bench_prof.json 2> $out/bench_prof.err || { echo rocprof failed; tail -5 $out/bench_prof.err; exit 1; }
bash scripts/pmc_passes.sh $out/pmc fetch write sq1 sq2 sq3 tcp1 tatd tcc1 || exit 1
python3 scripts/pmc_summary.py $out/pmc > $out/pmc_summary.txt
# the several-samples-per-call entry: its own trace, and how many kernels were in flight
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/fused -o fused -- python3 scripts/prof_fused.py 16 > $out/fused.txt 2> $out/fused.err || { echo rocprof fused failed; tail -5 $out/fused.err; exit 1; }
python3 scripts/trace_overlap.py $out/fused/fused_kernel_trace.csv >> $out/fused.txt
cat $out/fused.txt
python3 scripts/make_traffic_json.py $out/pmc "k_traverse_p<" $out/pmc_traffic.json
