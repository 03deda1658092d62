#!/usr/bin/env python
"""Turn the FETCH_SIZE / WRITE_SIZE rocprofv3 passes (tools/profile_all.sh) into profiles/<round>/traffic.json.
Units and corrections (MI355X_MICROARCH.md, HBM section; re-calibrated here on a known byte count):
  * both counters are in KiB;
  * WRITE_SIZE is exact for this code's store patterns (edge_dete_kernel<.,.,true> writes 32*3840*2160 B = 259200 KiB and
    the counter reads 259200.0);
  * FETCH_SIZE reports exactly 1/2 of the bytes of a coalesced streaming read on gfx950: the pure-read calibration kernel
    edge_dete_kernel<.,.,false> must fetch 32*3840*2160 * 66/64 B = 267300 KiB and the counter reads 133587 (x2.00).
So hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch."""
import csv, glob, json, os, sys
from collections import defaultdict


# device kernel -> the stage name bench.py / compvhip_plan_get_timing use
STAGE_NAMES = {"sht_vote_tiles_kernel": "sht_vote_kernel", "sht_reduce_tiles_kernel": "sht_reduce_kernel", "sht_compact_tiles_kernel": "sht_compact_kernel"}


def mean_counter(d, counter):
    agg = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                k = row["Kernel_Name"].replace("void ", "").split("(")[0]
                agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    return {k: v[0] / v[1] for k, v in agg.items()}


def main(prof_dir, out):
    fetch = mean_counter(os.path.join(prof_dir, "pmc_FETCH_SIZE"), "FETCH_SIZE")
    write = mean_counter(os.path.join(prof_dir, "pmc_WRITE_SIZE"), "WRITE_SIZE")
    res = {"workload": {"W": 3840, "H": 2160, "frames": 32}, "units": "bytes per launch", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("compvhip"):
            continue
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        name = k.replace("compvhip::", "")
        if not name.startswith("edge_dete_kernel"):   # the two edge_dete instantiations are the read / write calibration kernels
            name = name.split("<")[0]
        name = STAGE_NAMES.get(name, name)
        res["kernels"][name] = {
            "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
            "hbm_read_bytes": int(2 * f * 1024), "hbm_write_bytes": int(w * 1024), "hbm_bytes": int((2 * f + w) * 1024)}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res["kernels"], indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
