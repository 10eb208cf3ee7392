"""Per-launch counter averages of one scripts/pmc_family.sh run, as a small JSON under profiles/.

    python scripts/summarize_family.py gpurun_out/pmcfam_<tag> profiles/<round>/pmc_per_dispatch_<name>.json

FETCH_SIZE and WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 64-byte requests as 32 bytes, hence
the factor 2 on the fetch side. The file carries the hash of the kernel sources it was measured on.
"""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for path in sorted(glob.glob(os.path.join(src, "*", "pmc_counter_collection.csv"))):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            if "pbs_kernel" not in k:
                continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            meta[k] = {"grid": int(row["Grid_Size"]), "workgroup": int(row["Workgroup_Size"]), "vgpr": int(row["VGPR_Count"]),
                       "agpr": int(row["Accum_VGPR_Count"]), "scratch": int(row["Scratch_Size"])}
out = {"source_hash": bench.kernel_source_hash()}
for k, counters in acc.items():
    avg = {c: sum(v) / len(v) for c, v in counters.items()}
    e = {"launch": meta[k], "per_launch": avg}
    if "FETCH_SIZE" in avg:
        e["hbm_fetch_bytes"] = avg["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in avg:
        e["hbm_write_bytes"] = avg["WRITE_SIZE"] * 1024
    wc = avg.get("SQ_WAVE_CYCLES")
    if wc:
        e["fractions_of_wave_cycles"] = {"valu_active": avg.get("SQ_ACTIVE_INST_VALU", 0.0) / wc,
                                         "lds_active": avg.get("SQ_ACTIVE_INST_LDS", 0.0) / wc,
                                         "waiting_on_counter": avg.get("SQ_WAIT_INST_ANY", 0.0) / wc}
    if "TCC_HIT_sum" in avg and "TCC_MISS_sum" in avg:
        e["l2_hit_rate"] = avg["TCC_HIT_sum"] / (avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"])
    out[k] = e
with open(dst, "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(json.dumps({k: v.get("fractions_of_wave_cycles") for k, v in out.items() if isinstance(v, dict)}, indent=1))
