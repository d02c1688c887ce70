"""FETCH_SIZE / WRITE_SIZE passes (rocprofv3 --pmc, one counter per run) -> profiles/pmc_traffic.json, the per-launch
HBM bytes bench.py reports as roofline.traffic.  Units and corrections per MI355X_MICROARCH.md "HBM": the counters
are in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of 16-B-per-lane loads at 64 B, so reads are doubled
(the traversal kernels read nodes / triangles / rays as 16-B-per-lane loads); WRITE_SIZE is exact.

  python tools/pmc_traffic.py <fetch pmc_results.db> <write pmc_results.db> <tag>"""
import json
import sqlite3
import sys


def per_launch(db, counter, pattern):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    name = "kernel_name" if "kernel_name" in cols else "name"
    row = c.execute(f"select count(distinct dispatch_id), sum(value) from counters_collection where counter_name = ? and {name} like ?",
                    (counter, pattern)).fetchone()
    return row[1] / max(row[0], 1), row[0]


if __name__ == "__main__":
    fetch_db, write_db, tag = sys.argv[1:4]
    out = {"source": f"profiles/{tag}_pmc_fetch_size.csv, profiles/{tag}_pmc_write_size.csv (tools/gpu_measure.sh {tag})"}
    for key, pat in (("k_trace_closest", "%k_trace<0%"), ("k_trace_any", "%k_trace<1%"), ("k_shade", "k_shade%")):
        f, n = per_launch(fetch_db, "FETCH_SIZE", pat)
        w, _ = per_launch(write_db, "WRITE_SIZE", pat)
        out[key + "_fetch_kib_raw"] = f
        out[key + "_write_kib_raw"] = w
        out[key + "_launches"] = n
        out[key + "_bytes_per_launch"] = 2.0 * f * 1024.0 + w * 1024.0
    json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
    print(json.dumps(out, indent=1))
