#!/bin/bash
# Counter passes over tools/p3m_pmc_probe.py: usage tools/p3m_pmc_passes.sh <outdir> <snapshot> [row lanes] [rows pooled]
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$out"
run() {
  local name=$1; shift
  timeout -s KILL 90 rocprofv3 "$@" --output-format csv -d "$out/$name" -- python3 tools/p3m_pmc_probe.py "${ARGS[@]}" > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$out/$name.log"; return 1; }
  tail -1 "$out/$name.log" | cut -c1-300
}
ARGS=("$@")
run trace --kernel-trace --stats || exit 1
run pmc_valu --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE || exit 1
run pmc_wait --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE || exit 1
run pmc_mem --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE || exit 1
echo "passes complete in $out"
