#!/bin/bash
# The round's profile set on one GPU box: counter passes of the two bench.py workloads and the
# kernel trace of the 128^3 whole-step run.  usage: tools/profile_round.sh <tag, e.g. r02>
set -u
tag=$1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
tools/profile_bench.sh gpurun_out/prof_${tag}_direct || exit 1
python3 tools/summarize_prof.py gpurun_out/prof_${tag}_direct gpurun_out/${tag}_direct128_sym k_pairs_sym || exit 1
tools/profile_bench.sh gpurun_out/prof_${tag}_p3m --mode p3m || exit 1
python3 tools/summarize_prof.py gpurun_out/prof_${tag}_p3m gpurun_out/${tag}_p3m256 k_p3m || exit 1
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_run128 -- \
    nbody_amd/host/nbody_host_tests run tools/configs/p3m_hydro_128.ini 100 > gpurun_out/${tag}_run128.log 2>&1 || { tail -5 gpurun_out/${tag}_run128.log; exit 1; }
python3 tools/summarize_run_prof.py gpurun_out/${tag}_run128.log "gpurun_out/prof_${tag}_run128/**/*_kernel_stats.csv" gpurun_out/${tag}_run128_p3m_hydro.md || exit 1
cp gpurun_out/${tag}_direct128_sym.json gpurun_out/latest_direct128.json   # -> profiles/latest_*.json: what bench.py
cp gpurun_out/${tag}_p3m256.json gpurun_out/latest_p3m.json                # quotes while the kernel sources match
rm -rf gpurun_out/prof_${tag}_direct gpurun_out/prof_${tag}_p3m gpurun_out/prof_${tag}_run128   # raw CSVs: large
echo "profiles of ${tag} written to gpurun_out/"
