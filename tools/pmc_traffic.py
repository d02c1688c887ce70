"""FETCH_SIZE / WRITE_SIZE passes (rocprofv3 --pmc, one counter per run) -> profiles/pmc_traffic.json, the per-launch
HBM bytes bench.py reports as roofline.traffic -- a builder-side measurement of an earlier run of the same kernels, labelled
as such in the bench line (`traffic_source`) and dropped there when the kernel sources have changed since.

Units and corrections: the counters are in KiB.  MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide coalesced
streaming reads (x2).  The traversal kernels read scattered 64-B node and 48-B triangle records, so the factor is taken from
this repo's own calibration of that shape: tools/calib_records.hip under --pmc FETCH_SIZE (profiles/r02_calib_fetch_size.csv)
reads 8.59 GB of random 64-B records as 8.83 GB (factor 1.0; the streaming kernel of the same program reads x2, as the
guide says).  Round 1 doubled these reads and so overstated k_trace's traffic by about 2x.  WRITE_SIZE is exact.

  python tools/pmc_traffic.py <fetch pmc_results.db> <write pmc_results.db> <tag> [samples per step of the profiled run] [calibration csv]"""
import csv
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ["mi355pt.hip", "pt_device.h", "pt_shade.h", "pt_scene.h", "scene_flatten.hip", "bvh_build.cpp", "bvh_build.h"]


def source_hash():
    """Identifies the kernel build the counters were taken from (bench.py compares it with the sources it runs)."""
    h = hashlib.sha1()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "pathtracer_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def per_launch(db, counter, pattern):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    name = "kernel_name" if "kernel_name" in cols else "name"
    row = c.execute(f"select count(distinct dispatch_id), sum(value) from counters_collection where counter_name = ? and {name} like ?",
                    (counter, pattern)).fetchone()
    return (row[1] or 0.0) / max(row[0], 1), row[0]


def calibration(path):
    """FETCH_SIZE factor of the scattered-record shape and of the streaming shape from the calibration program's CSV."""
    requested_kib = 256 * 16 * 8 * 64 * 64 * 64 / 1024.0        # waves x lanes x records per lane x 64 B (tools/calib_records.hip)
    fac = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter"] == "FETCH_SIZE" and r["Kernel"].startswith("k_"):
                fac[r["Kernel"].split("(")[0]] = requested_kib / float(r["MeanPerDispatch"])
    return fac


if __name__ == "__main__":
    fetch_db, write_db, tag = sys.argv[1:4]
    spp = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    cal_path = sys.argv[5] if len(sys.argv) > 5 else os.path.join(ROOT, "profiles", "r02_calib_fetch_size.csv")
    fac = calibration(cal_path)
    scattered = fac["k_lane4"]
    out = {"source": f"profiles/{tag}_pmc_fetch_size.csv, profiles/{tag}_pmc_write_size.csv (builder PMC pass {tag}, tools/gpu_measure.sh)",
           "tag": tag, "kernel_source_hash": source_hash(), "spp_per_step": spp,
           "fetch_size_factor_scattered_records": scattered, "fetch_size_factor_streaming": fac.get("k_stream"),
           "calibration": os.path.relpath(cal_path, ROOT)}
    for key, pat in (("k_trace_closest", "%k_trace<0%"), ("k_trace_any", "%k_trace<1%"), ("k_shade", "%k_shade%")):
        f, n = per_launch(fetch_db, "FETCH_SIZE", pat)
        w, _ = per_launch(write_db, "WRITE_SIZE", pat)
        out[key + "_fetch_kib_raw"] = f
        out[key + "_write_kib_raw"] = w
        out[key + "_launches"] = n
        out[key + "_bytes_per_launch"] = scattered * f * 1024.0 + w * 1024.0
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
