import csv, sys, glob
log, statsglob, out = sys.argv[1], sys.argv[2], sys.argv[3]
rows = list(csv.DictReader(open(glob.glob(statsglob, recursive=True)[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(out, "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- nbody_amd/host/nbody_host_tests run tools/configs/p3m_hydro_128.ini 100\n\n")
    f.write("Full step (mesh force + P3M short range + HLL gas) of 128^3 particles on a 128^3 mesh, 100 KDK cycles resident in HBM.\n\n```\n")
    for line in open(log):
        if line.startswith(("initial", "final", "timing")):
            f.write(line)
    f.write("```\n\n| kernel | calls | avg ms | total ms | % |\n|---|---|---|---|---|\n")
    for r in rows[:30]:
        f.write(f"| `{r['Name'][:80]}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.4f} | {float(r['TotalDurationNs'])/1e6:.3f} | {100*float(r['TotalDurationNs'])/tot:.1f} |\n")
    f.write(f"\nall kernels: {tot/1e6:.1f} ms\n")
