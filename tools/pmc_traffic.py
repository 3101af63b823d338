#!/usr/bin/env python
"""pmc_traffic.py -- HBM bytes per launch of the mask kernel from rocprofv3 PMC passes, calibrated.

usage: pmc_traffic.py <out_dir> <workload> [<workload> ...]

Reads (all written by tools/gpu_round.sh under <out_dir>):
    calib_fetch/ , calib_write/     rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/pmc_calib (known byte counts)
    <wl>_fetch/ , <wl>_write/       the same two passes over `python bench.py --workload <wl>`
and writes <out_dir>/pmc_traffic.json:
    {"calibration": {...}, "<wl>:<kernel>": {"fetch_raw", "write_raw", "fetch_bytes", "write_bytes",
                                            "hbm_bytes_per_launch", ...}}
Correction (MI355X_MICROARCH.md, HBM section): the counters are only trusted after dividing a known byte count
by the counter value in the same access pattern; reads use the flat 16 B/lane factor (the guide's x2 for wide
coalesced reads shows up here as factor ~2 x 1024 when the counter is in KiB), writes use the factor of the
mask-shaped 128-byte-segment store pattern.
"""
import collections
import csv
import glob
import json
import os
import sys

CALIB_BYTES = {
    "calib_write_flat16": 512 << 20,
    "calib_read_flat16": 512 << 20,
    "calib_write_tile128": 125000 * 10 * 128,
}


def per_kernel(dirname):
    """-> {kernel short name: {counter: mean over dispatches of (sum over rows of one dispatch)}}"""
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            for pre in ("void ", "ksched::"):
                if name.startswith(pre):
                    name = name[len(pre):]
            name = name.split("<")[0]
            acc[name][r["Counter_Name"]][r.get("Dispatch_Id", "0")] += float(r["Counter_Value"])
    return {k: {c: sum(v.values()) / len(v) for c, v in d.items()} | {"_dispatches": len(next(iter(d.values())))}
            for k, d in acc.items()}


def main():
    out_dir = sys.argv[1]
    wls = sys.argv[2:]
    res = {}
    cf = per_kernel(os.path.join(out_dir, "calib_fetch"))
    cw = per_kernel(os.path.join(out_dir, "calib_write"))
    cal = {}
    try:
        cal["fetch_flat16_bytes_per_count"] = CALIB_BYTES["calib_read_flat16"] / cf["calib_read_flat16"]["FETCH_SIZE"]
        cal["write_flat16_bytes_per_count"] = CALIB_BYTES["calib_write_flat16"] / cw["calib_write_flat16"]["WRITE_SIZE"]
        cal["write_tile128_bytes_per_count"] = CALIB_BYTES["calib_write_tile128"] / cw["calib_write_tile128"]["WRITE_SIZE"]
        # cross terms: what a pure writer fetches / a pure reader writes (should be ~0)
        cal["fetch_count_of_write_flat16"] = cf.get("calib_write_flat16", {}).get("FETCH_SIZE")
        cal["write_count_of_read_flat16"] = cw.get("calib_read_flat16", {}).get("WRITE_SIZE")
    except KeyError as e:
        cal["error"] = f"calibration kernels missing: {e}"
    cal["raw"] = {"fetch_pass": cf, "write_pass": cw}
    res["calibration"] = cal
    for wl in wls:
        f = per_kernel(os.path.join(out_dir, f"{wl}_fetch"))
        w = per_kernel(os.path.join(out_dir, f"{wl}_write"))
        for k in sorted(set(f) | set(w)):
            if not k.startswith("k_eval"):
                continue
            short = k.replace("k_eval_", "")
            fr = f.get(k, {}).get("FETCH_SIZE")
            wr = w.get(k, {}).get("WRITE_SIZE")
            rec = {"kernel": k, "fetch_raw": fr, "write_raw": wr,
                   "dispatches": [f.get(k, {}).get("_dispatches"), w.get(k, {}).get("_dispatches")]}
            if "error" not in cal and fr is not None and wr is not None:
                rec["fetch_bytes"] = fr * cal["fetch_flat16_bytes_per_count"]
                rec["write_bytes"] = wr * cal["write_tile128_bytes_per_count"]
                rec["hbm_bytes_per_launch"] = rec["fetch_bytes"] + rec["write_bytes"]
            res[f"{wl}:{short}"] = rec
    res["_session"] = "session " + os.path.basename(os.path.normpath(out_dir))
    json.dump(res, open(os.path.join(out_dir, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k not in ("calibration", "_session")}, indent=1))
    print("calibration:", {k: v for k, v in cal.items() if k != "raw"})


if __name__ == "__main__":
    main()
