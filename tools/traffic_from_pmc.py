#!/usr/bin/env python
"""Turn the FETCH_SIZE / WRITE_SIZE rocprofv3 passes (tools/profile_all.sh) into profiles/<round>/traffic.json.
Units and corrections (MI355X_MICROARCH.md, HBM section; re-calibrated here on a known byte count):
  * both counters are in KiB;
  * WRITE_SIZE is exact for this code's store patterns (edge_dete_kernel<.,.,true> writes 32*3840*2160 B = 259200 KiB and
    the counter reads 259200.0);
  * FETCH_SIZE reports exactly 1/2 of the bytes of a coalesced streaming read on gfx950: the pure-read calibration kernel
    edge_dete_kernel<.,.,false> must fetch 32*3840*2160 * 66/64 B = 267300 KiB and the counter reads 133587 (x2.00).
So hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch."""
import csv, glob, hashlib, json, os, subprocess, sys, tempfile
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


# device kernel -> the stage name bench.py / compvhip_plan_get_timing use
STAGE_NAMES = {"sht_vote_tiles_kernel": "sht_vote_kernel", "sht_reduce_tiles_kernel": "sht_reduce_kernel", "sht_compact_tiles_kernel": "sht_compact_kernel",
               "canny_swar_tile_kernel": "canny_tile_kernel"}


def mean_counter(d, counter):
    agg = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                k = row["Kernel_Name"].replace("void ", "").split("(")[0]
                agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    return {k: v[0] / v[1] for k, v in agg.items()}


def valu_mix(src, kernel_substr):
    """Static VALU mix of a kernel (tools/isa_cost.py classes over hipcc's assembly of `src`): average issue cycles per wave64 VALU
    instruction with the microbenchmark costs (2.3 fast / 4.3 everything else).  Whole-kernel static counts: the row loops dominate them."""
    import isa_cost
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-inline-asm", "--cuda-device-only",
                               "-S", os.path.join(ROOT, "compv_amd", "csrc", src), "-o", asm], stderr=subprocess.DEVNULL)
        fast = slow = 0
        cur = None
        for raw in open(asm):
            line = raw.split(";")[0].rstrip()
            if line and not line[0].isspace() and line.endswith(":") and line.startswith("_Z"):
                cur = line[:-1]
            if not cur or kernel_substr not in cur or not line.strip() or line.lstrip().startswith("."):
                continue
            k, _ = isa_cost.classify(line)
            fast += k == "vfast"
            slow += k == "vslow"
    n = fast + slow
    return {"static_valu_fast": fast, "static_valu_other": slow, "valu_cycles_per_instruction": round((fast * isa_cost.C_FAST + slow * isa_cost.C_SLOW) / max(n, 1), 3)}


def main(prof_dir, out):
    fetch = mean_counter(os.path.join(prof_dir, "pmc_FETCH_SIZE"), "FETCH_SIZE")
    write = mean_counter(os.path.join(prof_dir, "pmc_WRITE_SIZE"), "WRITE_SIZE")
    valu = mean_counter(os.path.join(prof_dir, "pmc_sq1"), "SQ_INSTS_VALU")
    so = hashlib.sha256(open(os.path.join(ROOT, "compv_amd", "lib", "libcompv_hip.so"), "rb").read()).hexdigest()
    sys.path.insert(0, ROOT)
    import bench
    res = {"workload": {"W": 3840, "H": 2160, "frames": 32}, "units": "bytes per launch", "so_sha256": so, "src_sha256": bench.src_sha256(), "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("compvhip"):
            continue
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        name = k.replace("compvhip::", "")
        if not name.startswith("edge_dete_kernel"):   # the two edge_dete instantiations are the read / write calibration kernels
            name = name.split("<")[0]
        name = STAGE_NAMES.get(name, name)
        extra = {}
        if k in valu:
            extra["SQ_INSTS_VALU"] = int(valu[k])
        if name == "canny_tile_kernel":
            extra.update(valu_mix("canny_swar_kernels.hip", "canny_swar_tile_kernel"))
        if name == "sht_vote_kernel":
            extra.update(valu_mix("sht_tiles_kernels.hip", "sht_vote_tiles_kernel"))
        res["kernels"][name] = {**extra,
            "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
            "hbm_read_bytes": int(2 * f * 1024), "hbm_write_bytes": int(w * 1024), "hbm_bytes": int((2 * f + w) * 1024)}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res["kernels"], indent=1))
    # the single-frame KHT call (tools/kht_bench.py under the same two passes): HBM bytes of its kernels per call -> kht_traffic.json (bench.py: kht.roofline.traffic)
    kf = mean_counter(os.path.join(prof_dir, "pmc_kht_FETCH_SIZE"), "FETCH_SIZE")
    kw = mean_counter(os.path.join(prof_dir, "pmc_kht_WRITE_SIZE"), "WRITE_SIZE")
    kk = {}
    for k in sorted(set(kf) | set(kw)):
        if "kht_" not in k and "bytes_to_bits" not in k:
            continue
        f, w = kf.get(k, 0.0), kw.get(k, 0.0)
        kk[k.replace("compvhip::", "").split("<")[0]] = {"hbm_read_bytes": int(2 * f * 1024), "hbm_write_bytes": int(w * 1024), "hbm_bytes": int((2 * f + w) * 1024)}
    if kk:
        kres = {"workload": "tools/kht_bench.py: compvhip_houghkht_u8 on one 3840x2160 Canny(59,119) edge map, threshold 100", "units": "bytes per call (one launch of each kernel)",
                "so_sha256": so, "kernels": kk, "hbm_bytes_per_call": sum(v["hbm_bytes"] for v in kk.values())}
        json.dump(kres, open(os.path.join(os.path.dirname(out), "kht_traffic.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
