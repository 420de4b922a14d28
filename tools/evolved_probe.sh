#!/bin/bash
# Evolves the 128^3 P3M + gas configuration for a number of cycles, then times the P3M kernel on the
# evolved particle set (P3M_PROBE_QUICK=1: whole-set timings only; NBPP_LIBS="a.so b.so": one
# probe per library build; INI=<file>: another configuration).  usage: tools/evolved_probe.sh <cycles>
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
sed 's/debug_info_every_cycles = .*/debug_info_every_cycles = 4000/' "${INI:-tools/configs/p3m_hydro_128.ini}" > /tmp/evolve.ini
printf "save_hdf5_every_delta_a = 10\n" >> /tmp/evolve.ini
nbody_amd/host/nbody_host_tests run /tmp/evolve.ini "$1" /tmp/evolved | grep "^cycle\|^timing"
for so in ${NBPP_LIBS:-libnbpp.so}; do
    echo "== $so"
    NBPP_LIB=$PWD/nbody_amd/lib/$so python tools/p3m_clustered_probe.py --snapshot /tmp/evolved_0001.bin
done
