"""Condenses the rocprofv3 outputs of scripts/profile.sh into the small summaries kept under profiles/.

    python scripts/summarize_pmc.py gpurun_out/prof_<tag> profiles/<round> <suffix>

Writes kernel_stats_<suffix>.csv (our kernels' rows of the --stats table) and
pmc_per_dispatch_<suffix>.json (per-launch averages of every collected counter for the bootstrap and
keyswitch kernels, plus the derived HBM traffic: FETCH_SIZE and WRITE_SIZE are reported in KB, and
on gfx950 FETCH_SIZE counts 64-byte requests as 32 bytes, hence the factor 2 on the fetch side).
"""
from __future__ import annotations

import csv
import glob
import json
import os
import sys
from collections import defaultdict

OURS = ("w16::pbs_kernel", "w4::pbs_kernel", "wave8_pbs_kernel", "wave_pbs_kernel", "block_pbs_kernel", "generic_pbs_kernel", "keyswitch_kernel", "ks_mfma_kernel",
        "ks_digits_kernel")


def short(name: str) -> str | None:
    for k in OURS:
        if k in name:
            return k
    return None


def main() -> None:
    src, dst, suffix = sys.argv[1], sys.argv[2], sys.argv[3]
    os.makedirs(dst, exist_ok=True)
    stats = os.path.join(src, "trace", "trace_kernel_stats.csv")
    with open(stats, newline="") as f, open(os.path.join(dst, f"kernel_stats_{suffix}.csv"), "w", newline="") as g:
        rd = csv.reader(f)
        wr = csv.writer(g)
        wr.writerow(next(rd))
        for row in rd:
            if "camd::" in row[0]:
                wr.writerow(row)

    acc: dict[str, dict[str, list[float]]] = defaultdict(lambda: defaultdict(list))
    meta: dict[str, dict] = {}
    for path in sorted(glob.glob(os.path.join(src, "pmc_*", "pmc_counter_collection.csv"))):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if k is None:
                    continue
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                meta[k] = {"grid": int(row["Grid_Size"]), "workgroup": int(row["Workgroup_Size"]),
                           "lds_bytes": int(row["LDS_Block_Size"]), "scratch": int(row["Scratch_Size"]),
                           "vgpr": int(row["VGPR_Count"]), "agpr": int(row["Accum_VGPR_Count"]),
                           "sgpr": int(row["SGPR_Count"])}
    out = {}
    for k, counters in acc.items():
        avg = {c: sum(v) / len(v) for c, v in counters.items()}
        entry = {"launch": meta[k], "launches_averaged": {c: len(v) for c, v in counters.items()},
                 "per_launch": avg}
        if "FETCH_SIZE" in avg:
            entry["hbm_fetch_bytes"] = avg["FETCH_SIZE"] * 1024 * 2  # KB units, gfx950 x2 correction
        if "WRITE_SIZE" in avg:
            entry["hbm_write_bytes"] = avg["WRITE_SIZE"] * 1024
        if "SQ_BUSY_CYCLES" in avg and "SQ_ACTIVE_INST_VALU" in avg and "SQ_WAVE_CYCLES" in avg:
            wc = avg["SQ_WAVE_CYCLES"]
            entry["fractions_of_wave_cycles"] = {
                "valu_active": avg["SQ_ACTIVE_INST_VALU"] / wc,
                "lds_active": avg.get("SQ_ACTIVE_INST_LDS", 0.0) / wc,
                "waiting_on_counter": avg.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                "any_instruction_active": avg.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
            }
        if "SQ_INSTS_VALU_FMA_F64" in avg:
            # wave-instruction counts: a wave-instruction is 64 lanes; an FMA is two flop
            fma, add, mul = avg["SQ_INSTS_VALU_FMA_F64"], avg.get("SQ_INSTS_VALU_ADD_F64", 0.0), avg.get("SQ_INSTS_VALU_MUL_F64", 0.0)
            entry["counted_fp64_flop"] = 64.0 * (2.0 * fma + add + mul)
            entry["counted_fp64_instructions"] = fma + add + mul + avg.get("SQ_INSTS_VALU_TRANS_F64", 0.0)
        if "TCC_HIT_sum" in avg and "TCC_MISS_sum" in avg:
            entry["l2_hit_rate"] = avg["TCC_HIT_sum"] / (avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"])
        out[k] = entry
    # ties the summary to the kernel sources it was measured on (bench.py reports the traffic
    # figure only while this hash matches the tree)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    out["source_hash"] = bench.kernel_source_hash()
    with open(os.path.join(dst, f"pmc_per_dispatch_{suffix}.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
