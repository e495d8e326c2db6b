#!/usr/bin/env python3
"""Classify the VALU instructions of the product kernels by their measured issue cost on gfx950.

usage: tools/isa_valu_classes.py [kernel-substring ...]      (compiles csrc/*.hip with -S, static counts)
Classes come from profiles/r3_valu_issue.txt (tools/ubench/valu_issue.hip):
  2 cycles / wave64: v_add/sub/subrev_u32, v_and/or/xor/not_b32, v_mov_b32, v_lshrrev_b32, v_ashrrev_i32, the unpacked 16-bit
                     VOP2 ops (add/sub/min/max/mul_lo/shifts), v_add/sub/mul/min/max_f16, v_add/sub/subrev/mul/fma/fmac_f32
                     -- ONLY with VGPR / constant operands, no SDWA, no DPP: one SGPR operand makes the same opcode 4 cycles
  8 cycles:          transcendentals, v_mad_u16 / v_fma_f16 (unpacked), v_min3/max3/med3_{f16,i16,u16}
  4 cycles:          everything else (VOP3 three-operand ops, v_perm, v_alignb*, packed 16-bit and packed f32 ops, v_dot*,
                     v_sad*, v_lshlrev_b32, v_min/max_{u32,i32,f32}, v_cmp*, v_cvt*, all SDWA / DPP forms, f64 add/mul/fma)
The output bounds a kernel's VALU-pipe time: cycles = sum over classes of count x cost (static mix; the PMC counters cannot
separate the classes: SQ_ACTIVE_INST_VALU ticks once per 2- and per 4-cycle instruction, profiles/r3_pmc_valu_calibration.txt).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "orb_slam3_fast_amd", "csrc")
DUAL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32",
        "v_lshrrev_b32", "v_ashrrev_i32", "v_add_u16", "v_sub_u16", "v_max_u16", "v_min_u16", "v_max_i16", "v_min_i16",
        "v_mul_lo_u16", "v_lshlrev_b16", "v_lshrrev_b16", "v_add_f16", "v_sub_f16", "v_mul_f16", "v_max_f16", "v_min_f16",
        "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32"}
OCT = re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_|^v_(mad_u16|mad_i16|fma_f16|mad_f16)$|^v_(min3|max3|med3)_(f16|i16|u16)$")
SGPR = re.compile(r"(?<![a-z0-9_])(s\d+|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec|exec_lo|exec_hi|m0|scc)(?![a-z0-9_])")


def classify(line):
    parts = line.split(None, 1)
    op = parts[0]
    if not op.startswith("v_") or op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_mfma", "v_accvgpr")):
        return None
    operands = parts[1].split(";")[0] if len(parts) > 1 else ""
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if OCT.search(base):
        return 8
    if op.endswith(("_sdwa", "_dpp")) or base not in DUAL:
        return 4
    srcs = operands.split(",")[1:]    # destination first
    if any(SGPR.search(x) for x in srcs):
        return 4
    return 2


def main(pats):
    out_json = None
    if "--json" in pats:
        i = pats.index("--json")
        out_json = pats[i + 1]
        pats = pats[:i] + pats[i + 2:]
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for f in sorted(os.listdir(CSRC)):
            if not f.endswith(".hip") or "_api" in f:
                continue
            out = os.path.join(td, f + ".s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                                   "--cuda-device-only", os.path.join(CSRC, f), "-o", out], stderr=subprocess.DEVNULL)
            cur = None
            for line in open(out):
                m = re.match(r"^(_Z\w+):", line)
                if m:
                    cur = m.group(1)
                    continue
                if line.startswith(".Lfunc_end"):
                    cur = None
                if cur is None or not line.startswith("\tv_"):
                    continue
                c = classify(line.strip())
                if c:
                    res.setdefault(cur, {2: 0, 4: 0, 8: 0})[c] += 1
    print("%-70s %6s %6s %6s  %s" % ("kernel (mangled)", "2-cyc", "4-cyc", "8-cyc", "static mean cycles / VALU instruction"))
    for k, d in sorted(res.items()):
        if pats and not any(p in k for p in pats):
            continue
        n = sum(d.values())
        print("%-70s %6d %6d %6d  %.2f" % (k[:70], d[2], d[4], d[8], (2 * d[2] + 4 * d[4] + 8 * d[8]) / max(1, n)))
    if out_json:
        import json
        short = {}
        for k, d in sorted(res.items()):
            m = re.match(r"_ZN4orbx\d+(k_[a-z_0-9]+?)(I|E)", k)
            if not m:
                continue
            n = sum(d.values())
            e = {"static_2cycle": d[2], "static_4cycle": d[4], "static_8cycle": d[8],
                 "mean_cycles_per_valu_inst": round((2 * d[2] + 4 * d[4] + 8 * d[8]) / max(1, n), 3)}
            # several instantiations of one kernel: keep the one with the most instructions seen last (they differ by < 2 %)
            if m.group(1) not in short or "ILb0ELi48" in k or "ILi80" in k:
                short[m.group(1)] = e
        short["_note"] = ("static VALU instruction mix of the product kernels by measured issue class (profiles/r3_valu_issue.txt): "
                          "mean cycles per wave64 VALU instruction; bench.py's roofline.valu multiplies SQ_INSTS_VALU by it")
        json.dump(short, open(out_json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:])
