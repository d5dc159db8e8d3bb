#!/bin/bash
set -u
out=gpurun_out/pmc_bvh
mkdir -p $out
export TMPDIR=/tmp
pass() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" -d $out/pmc_$name -o $name --output-format csv -- python3 scripts/gpu_bvh_profile.py knot 660 2 > $out/$name.log 2>&1; echo "pass $name rc=$?"; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY &&
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU
python3 scripts/pmc_summary.py $out > $out/summary.txt
grep -n "k_subtree(" -A20 $out/summary.txt | head -40
