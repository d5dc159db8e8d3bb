#!/bin/bash
# rocprofv3 counter passes over scripts/prof_run.py (one run per pass; block limits: 8 SQ, 4 TCC,
# 4 TCP, 2 TA, 2 TD, 2 GRBM).  usage: pmc_passes.sh <outdir> [pass names...]; NOVA_KERNEL selects the kernel
set -u
out=${1:-gpurun_out/pmc}; shift
want=${*:-tcp1 tcp2 tatd tcc1 sq1 sq2 fetch write grbm sq3 sqc}
mkdir -p "$out"
export TMPDIR=/tmp
pass() {
  name=$1; shift
  case " $want " in *" $name "*) ;; *) return 0;; esac
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d "$out/pmc_$name" -o "$name" --output-format csv -- python3 scripts/prof_run.py 2 > "$out/$name.log" 2>&1
  rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum &&
pass tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum &&
pass tatd TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum &&
pass tcc1 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum &&
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY &&
pass sq2 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM &&
pass fetch FETCH_SIZE &&
pass write WRITE_SIZE &&
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT &&
pass sq3 SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY &&
pass sqc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_BUSY_CYCLES SQC_DCACHE_MISSES_DUPLICATE
