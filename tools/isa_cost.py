#!/usr/bin/env python
"""Static issue-cost estimate of a gfx950 kernel from hipcc's assembly (-S --cuda-device-only).

The Canny and Hough kernels of this repository are bound by VALU issue, so the basic blocks' instruction mix predicts their
time well before a GPU run.  Costs per wave64 instruction (SIMD cycles at the nominal 2.4 GHz clock) come from
tools/microbench/valu_rate_bench2/3 (results quoted in DESIGN.md section 4.1): the "fast" class (v_add/sub_u32, v_and/or/xor,
v_lshrrev/ashrrev, v_mov, v_not, 16-bit VOP2 arithmetic, v_add_f32, v_add/max_f16) issues in ~2.3 cycles when every operand is a VGPR or
a literal, everything else in ~4.3.

    python tools/isa_cost.py file.s [kernel-name-substring]   -> per basic block: VALU fast/slow, SALU, LDS, VMEM, est. cycles
"""
import re
import sys

FAST = {
    "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32",
    "v_not_b32", "v_add_u16", "v_sub_u16", "v_subrev_u16", "v_max_u16", "v_min_u16", "v_max_i16", "v_min_i16", "v_mul_lo_u16",
    "v_lshlrev_b16", "v_lshrrev_b16", "v_ashrrev_i16", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_add_f16", "v_sub_f16", "v_max_f16", "v_min_f16",
}
C_FAST, C_SLOW = 2.3, 4.3


def classify(line):
    m = re.match(r"\s*([a-z_0-9]+)", line)
    if not m:
        return None, None
    op = m.group(1)
    if op.startswith("v_"):
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
        ops = line.split(None, 1)[1] if len(line.split(None, 1)) > 1 else ""
        srcs = ops.split(",")[1:]
        sgpr_src = any(re.match(r"\s*(s\d+|s\[|vcc|exec|m0)", s) for s in srcs)
        if base in FAST and not sgpr_src and not op.endswith(("_sdwa", "_dpp", "_e64")) and "dpp" not in line and "sdwa" not in line.lower():
            return "vfast", op
        return "vslow", op
    if op.startswith("s_"):
        if op.startswith(("s_load", "s_buffer_load")):
            return "smem", op
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_sleep")):
            return "swait", op
        if op.startswith("s_cbranch") or op == "s_branch":
            return "branch", op
        return "salu", op
    if op.startswith("ds_"):
        return "lds", op
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem", op
    return "other", op


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    cur_kernel = None
    blocks = []  # (kernel, label, counts dict, slow-op histogram)
    cur = None
    for raw in open(path):
        line = raw.split(";")[0].rstrip()
        if not line.strip():
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", line)
        if m:
            lab = m.group(1)
            if lab.startswith("_Z") or (not lab.startswith(".") and cur_kernel is None):
                cur_kernel = lab
            cur = {"kernel": cur_kernel, "label": lab, "n": {}, "hist": {}}
            blocks.append(cur)
            continue
        if line.lstrip().startswith("."):
            if ".end_amdhsa_kernel" in line or line.strip().startswith(".size"):
                pass
            continue
        if cur is None:
            continue
        k, op = classify(line)
        if k is None:
            continue
        cur["n"][k] = cur["n"].get(k, 0) + 1
        if k in ("vslow", "vfast", "lds", "vmem", "salu"):
            cur["hist"][op] = cur["hist"].get(op, 0) + 1
    print("%-34s %6s %6s %6s %5s %5s %5s %8s" % ("block", "vfast", "vslow", "salu", "lds", "vmem", "br", "valu_cyc"))
    for b in blocks:
        if want and want not in (b["kernel"] or ""):
            continue
        n = b["n"]
        tot = sum(n.values())
        if tot < 8:
            continue
        cyc = n.get("vfast", 0) * C_FAST + n.get("vslow", 0) * C_SLOW
        print("%-34s %6d %6d %6d %5d %5d %5d %8.0f" % (b["label"][-34:], n.get("vfast", 0), n.get("vslow", 0), n.get("salu", 0), n.get("lds", 0),
                                                        n.get("vmem", 0), n.get("branch", 0), cyc))
        if "-v" in sys.argv:
            top = sorted(b["hist"].items(), key=lambda kv: -kv[1])[:14]
            print("      " + "  ".join("%s:%d" % kv for kv in top))


if __name__ == "__main__":
    main()
