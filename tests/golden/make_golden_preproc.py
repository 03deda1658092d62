#!/usr/bin/env python
"""Generate tests/golden/golden_preproc.json from the COMPILED REFERENCE (oracle/_ref, built by oracle/build_ref.sh from
/root/reference): MD5 of CompVImage::convertGrayscale outputs for every packed format and CompVImage::thresholdOtsu values,
on inputs any box can regenerate (numpy default_rng / the SURVEY 8d synthetic frame).  Run in the build container only."""
import hashlib, json, os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from oracle_bindings import RefShim, synth_frame  # noqa: E402

FMT_NAMES = ["RGBA32", "ARGB32", "BGRA32", "RGB24", "BGR24", "RGB565LE", "RGB565BE", "BGR565LE", "BGR565BE", "YUYV422", "UYVY422", "Y"]
FMT_BYTES = [4, 4, 4, 3, 3, 2, 2, 2, 2, 2, 2, 1]


def packed_input(fmt, W, H, S, seed):
    return np.random.default_rng(seed).integers(0, 256, size=(H, S * FMT_BYTES[fmt]), dtype=np.uint8)


def otsu_input(kind, W, H, seed):
    rng = np.random.default_rng(seed)
    if kind == "synth":
        return synth_frame(W, H, 12345 + seed)
    if kind == "noise":
        return rng.integers(0, 256, size=(H, W), dtype=np.uint8)
    if kind == "bimodal":
        return np.where(rng.random((H, W)) < 0.3, rng.integers(150, 220, (H, W)), rng.integers(10, 90, (H, W))).astype(np.uint8)
    if kind == "flat":
        return np.full((H, W), 77, np.uint8)
    raise ValueError(kind)


def main():
    ref = RefShim(threads=1)
    out = {"grayscale": [], "otsu": []}
    for fmt in range(len(FMT_NAMES)):
        for (W, H, S) in ((64, 8, 64), (130, 17, 160), (642, 31, 704), (1920, 24, 1920)):
            seed = 5000 + 10 * fmt + (W % 7)
            data = packed_input(fmt, W, H, S, seed)
            g = ref.grayscale(data, fmt, W, FMT_BYTES[fmt])
            out["grayscale"].append({"fmt": fmt, "name": FMT_NAMES[fmt], "W": W, "H": H, "S": S, "seed": seed,
                                     "md5": hashlib.md5(np.ascontiguousarray(g).tobytes()).hexdigest()})
    for kind in ("synth", "noise", "bimodal", "flat"):
        for (W, H, seed) in ((20, 20, 1), (333, 77, 2), (641, 480, 3), (1282, 720, 4), (1920, 1080, 5), (3840, 2160, 6)):
            img = otsu_input(kind, W, H, seed)
            out["otsu"].append({"kind": kind, "W": W, "H": H, "seed": seed, "threshold": int(ref.otsu(img))})
    with open(os.path.join(HERE, "golden_preproc.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", len(out["grayscale"]), "grayscale and", len(out["otsu"]), "otsu vectors")


if __name__ == "__main__":
    main()
