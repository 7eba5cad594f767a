"""Reduce rocprofv3 outputs to bytes per eBWT row per launch.

  python profiles/reduce_pmc.py <kernel_stats.csv> <pmc_FETCH.csv> <pmc_WRITE.csv> <rows> <workload> > traffic_per_row.json

kernel_stats.csv : `rocprofv3 --kernel-trace --stats` kernel_stats file (Name, Calls, AverageNs ...)
pmc_*.csv        : counter_collection files of separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (KB per dispatch)
FETCH_SIZE is reported raw; bench.py doubles it for the coalesced streaming kernels (MI355X_MICROARCH.md, HBM section).
"""
import csv, json, sys, collections


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def per_kernel(path, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    return {k: tot[k] / cnt[k] * 1024.0 for k in tot}            # bytes per dispatch


def main():
    stats, fetch, write, rows, workload = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), sys.argv[5]
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    out = {"workload": workload, "rows": rows,
           "note": "FETCH_SIZE is raw (KB per dispatch x 1024 / rows); MI355X_MICROARCH.md: double it for wide coalesced "
                   "streaming reads, raw for random sector reads", "kernels": {}}
    for r in csv.DictReader(open(stats)):
        k = short(r["Name"])
        out["kernels"][k] = {"fetch_B_per_row_raw": round(f.get(k, 0.0) / rows, 3), "write_B_per_row": round(w.get(k, 0.0) / rows, 3),
                             "avg_ms_rocprof": round(float(r["AverageNs"]) / 1e6, 3), "calls": int(r["Calls"])}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
