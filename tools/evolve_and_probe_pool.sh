#!/bin/bash
# Evolves the 128^3 P3M + gas configuration for <cycles> cycles (12000: a = 0.51), then runs
# tools/p3m_pool_probe.py on the lattice, the blobs and the evolved state.
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
sed 's/debug_info_every_cycles = .*/debug_info_every_cycles = 4000/' tools/configs/p3m_hydro_128.ini > /tmp/evolve.ini
printf "save_hdf5_every_delta_a = 10\n" >> /tmp/evolve.ini
nbody_amd/host/nbody_host_tests run /tmp/evolve.ini "$1" /tmp/evolved | grep "^cycle\|^timing"
python tools/p3m_pool_probe.py --snapshot /tmp/evolved_0001.bin --json gpurun_out/p3m_pool_probe.json
