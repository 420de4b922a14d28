#!/bin/bash
# Profiles `python3 bench.py <args>` with rocprofv3: one kernel-trace pass and separate
# counter passes (rocprofv3 does not split counters over passes by itself).
# usage: tools/profile_bench.sh <outdir> [bench args...]
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$out"
run() {  # name, rocprof args...
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$out/$name" -- python3 bench.py --no-cpu-baseline --no-extras "${BENCH_ARGS[@]}" > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$out/$name.log"; return 1; }
  tail -1 "$out/$name.log" | cut -c1-300
}
BENCH_ARGS=("$@")
run trace --kernel-trace --stats || exit 1
BENCH_ARGS=("$@" --steps 1 --warmup 0)
run pmc_valu --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE || exit 1
run pmc_wait --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE || exit 1
run pmc_fetch --pmc FETCH_SIZE || exit 1
run pmc_write --pmc WRITE_SIZE || exit 1
echo "profile passes complete in $out"
