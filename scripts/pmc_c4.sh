#!/bin/bash
# rocprofv3 counter passes over scripts/bench_c4.py --prof (one run per pass).  usage: pmc_c4.sh <outdir>
set -u
out=${1:-gpurun_out/pmc_c4}
mkdir -p "$out"
export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" -d "$out/pmc_$name" -o "$name" --output-format csv -- python3 scripts/bench_c4.py --prof > "$out/$name.log" 2>&1
  rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
pass fetch FETCH_SIZE &&
pass write WRITE_SIZE &&
pass tcc1 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum &&
pass tatd TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum &&
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY &&
pass sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM &&
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
