"""VALU-issue roofline of a GPIS kernel from its ISA, the measured per-instruction issue costs and the dynamic
instruction-class counters.

    issue_cycles(launch) = sum over the kernel's loops g of  w_g x sum_{instruction i of g} cost(i)
    frac                 = issue_cycles / (kernel duration x 1024 SIMDs x 2.4 GHz)

cost(i)  SIMD issue cycles per wave64 instruction, MEASURED per mnemonic (tools/valu_issue_bench →
         profiles/r02_valu_issue_cycles.json: 2.0 for full-rate ops, ~3.7 for f64 / packed f32 / integer multiply /
         3-operand integer ops / conversions / compares, ~7 for transcendentals and v_readlane).
w_g      how often loop g's body ran in the launch.  No counter reports it directly; it is FITTED (non-negative least
         squares) so that the loops' static instruction counts reproduce the twelve DYNAMIC counters of the launch:
         SQ_INSTS_VALU and its eleven class counters (ADD/MUL/FMA/TRANS x F32/F64, CVT, INT32, INT64).  Which mnemonic
         each class counter counts was calibrated by running the microbenchmark's one-mnemonic kernels under the same
         counters ("pmc_class_of_kernel" in the cycles file).  The fit's residual is reported.
Bounds:  lo = 2 cycles for every instruction not in a homogeneous 4-/7-/14-cycle class (the floor the counters prove);
         hi = SQ_ACTIVE_INST_VALU x 4 (the hardware's own busy counter; it has quad-cycle granularity, so a 2-cycle
         instruction counts as 4: an upper bound).

usage: python tools/issue_model.py --isa gpis_hip.s --kernel 24k_guided_sample_distanceILb1E --counters counters.json --ms 252.3
"""
import argparse
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD = 256 * 4
PEAK_CLOCK_HZ = 2.4e9            # MI355X_MICROARCH.md "Max clock"


def parse_kernel_isa(path, kernel_substr):
    """→ {loop header label: [VALU mnemonics]} for the first kernel whose mangled name contains kernel_substr; code outside
    any loop is group "top"."""
    label_re = re.compile(r"^(\.LBB\d+_\d+|_Z\w+):")
    groups, cur, pending, inside = {}, "top", None, False
    for line in open(path):
        if not inside:
            m = label_re.match(line)
            if m and m.group(1).startswith("_Z") and kernel_substr in m.group(1):
                inside = True
            continue
        if line.lstrip().startswith(".end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            break
        m = label_re.match(line)
        if m:
            pending = m.group(1)
            h = re.search(r"in Loop: Header=(\S+) Depth=(\d+)", line)
            if "Loop Header" in line:
                cur = pending
            elif h:
                cur = ".L" + h.group(1)
            continue
        if re.search(r";\s+(?:=>)?\s*This (?:Inner )?Loop Header: Depth=\d+", line):
            cur = pending
            continue
        h = re.search(r";\s+in Loop: Header=(\S+) Depth=\d+", line)
        if h:
            cur = ".L" + h.group(1)
            continue
        t = line.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        mn = re.sub(r"_e(32|64)$|_dpp$|_sdwa$", "", t.split()[0])
        if mn.startswith("v_"):
            groups.setdefault(cur, []).append(mn)
    return groups


def pmc_class(m):
    """PMC class (SQ_INSTS_VALU_*) a VALU mnemonic is counted in — calibrated by running tools/valu_issue_bench's
    one-mnemonic kernels under the counters (profiles/r02_valu_issue_cycles.json "pmc_class_of_kernel"):
    integer ARITHMETIC, compares and bfe count as INT32; logic, shifts, alignbit, bfi, perm, moves, float compares,
    cndmask, min/max, readlane are counted by no class counter (OTHER = SQ_INSTS_VALU minus the classes)."""
    if not m.startswith("v_"):
        return None
    if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_f32", m):
        return "TRANS_F32"
    if re.match(r"v_(rcp|rsq|sqrt)_f64", m):
        return "TRANS_F64"
    if re.match(r"v_(fma|fmac|mad|mac|pk_fma)_f32", m):
        return "FMA_F32"
    if re.match(r"v_(fma|fmac)_f64", m):
        return "FMA_F64"
    if re.match(r"v_(add|sub|subrev|pk_add)_f32", m):
        return "ADD_F32"
    if re.match(r"v_(mul|pk_mul|mul_legacy)_f32", m):
        return "MUL_F32"
    if re.match(r"v_add_f64", m):
        return "ADD_F64"
    if re.match(r"v_mul_f64", m):
        return "MUL_F64"
    if re.match(r"v_cvt_", m):
        return "CVT"
    if re.match(r"v_(mad_u64_u32|mad_i64_i32|lshl_add_u64)", m):
        return "INT64"
    if re.match(r"v_(add|sub|subrev|addc|subb|subbrev)_(co_)?(ci_)?[ui]32|v_add3_u32|v_mul_(lo|hi)_[ui]32|v_mul_[ui]32_[ui]24|v_mad_[ui]32_[ui]24|"
                r"v_lshl_add_u32|v_add_lshl_u32|v_xad_u32|v_bfe_[ui]32|v_mbcnt|v_cmp\w*_[ui](16|32|64)$|v_(min|max|med3)_[ui]32|v_sad_", m):
        return "INT32"
    return "OTHER"


def cost_table(micro_path, column=None):
    """mnemonic → SIMD issue cycles per wave64 instruction (profiles/r02_valu_issue_cycles.json, v_add_f32 = 2.0)"""
    t = json.load(open(micro_path))["cycles"]
    return {k: v for k, v in t.items() if "(" not in k}, t


FAMILY = [  # unmeasured mnemonic → a measured one of the same encoding / pipe
    (r"v_subrev_f32|v_sub_f32", "v_sub_f32"), (r"v_min_f32|v_max_f32|v_med3_f32", "v_max_f32"),
    (r"v_fmac_f32|v_mac_f32|v_mad_f32", "v_fmac_f32"), (r"v_fma_f32", "v_fma_f32"),
    (r"v_pk_fma_f32", "v_pk_fma_f32"), (r"v_pk_mul_f32", "v_pk_mul_f32"), (r"v_pk_add_f32", "v_pk_add_f32"),
    (r"v_(fma|fmac)_f64", "v_fma_f64"), (r"v_mul_f64", "v_mul_f64"), (r"v_add_f64", "v_add_f64"), (r"v_ldexp_f64|v_frexp|v_trig|v_fract_f64|v_floor_f64|v_rndne_f64", "v_ldexp_f64"),
    (r"v_cvt_f64_|v_cvt_f32_f64|v_cvt_.*_f64", "v_cvt_f64_f32"), (r"v_cvt_", "v_cvt_f32_u32"),
    (r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_f32", "v_exp_f32"), (r"v_(rcp|rsq|sqrt)_f64", "v_rsq_f64"),
    (r"v_mad_u64_u32|v_mad_i64_i32", "v_mad_u64_u32"), (r"v_lshl_add_u64|v_lshlrev_b64|v_lshrrev_b64|v_ashrrev_i64", "v_lshl_add_u64"), (r"v_mov_b64", "v_mov_b64"),
    (r"v_mul_lo_u32|v_mul_lo_i32", "v_mul_lo_u32"), (r"v_mul_hi_u32|v_mul_hi_i32", "v_mul_hi_u32"), (r"v_mul_u32_u24|v_mul_i32_i24|v_mad_u32_u24|v_mad_i32_i24", "v_mul_u32_u24"),
    (r"v_add3_u32|v_xad_u32|v_add_lshl_u32", "v_add3_u32"), (r"v_lshl_add_u32", "v_lshl_add_u32"), (r"v_lshl_or_b32", "v_lshl_or_b32"), (r"v_and_or_b32", "v_and_or_b32"),
    (r"v_or3_b32", "v_or3_b32"), (r"v_bfe_[ui]32", "v_bfe_u32"), (r"v_bfi_b32", "v_bfi_b32"), (r"v_perm_b32", "v_perm_b32"), (r"v_bitop3_b32", "v_bitop3_b32"),
    (r"v_alignbit_b32|v_alignbyte_b32", "v_alignbit_b32"),
    (r"v_addc_co_u32|v_subb_co_u32", "v_addc_co_u32"), (r"v_add_co_u32|v_sub_co_u32|v_subrev_co_u32", "v_add_co_u32"),
    (r"v_cmp_.*_f64|v_cmp_class_f64", "v_cmp_gt_f32"), (r"v_cmp_.*_[ui]32|v_cmp_.*_[ui]64", "v_cmp_lt_u32"), (r"v_cmp", "v_cmp_lt_f32"),
    (r"v_lshlrev_b32", "v_lshlrev_b32"), (r"v_lshrrev_b32|v_ashrrev_i32", "v_lshrrev_b32"), (r"v_lshlrev_b64|v_lshrrev_b64|v_ashrrev_i64", "v_lshl_add_u64"),
    (r"v_div_scale|v_div_fmas|v_div_fixup", "v_fma_f64"), (r"v_min_|v_max_|v_med3_", "v_max_f32"),
    (r"v_cndmask_b32", "v_cndmask_b32"), (r"v_readlane_b32|v_writelane_b32", "v_readlane_b32"), (r"v_readfirstlane_b32", "v_readfirstlane_b32"),
    (r"v_mbcnt", "v_mbcnt_lo_u32_b32"), (r"v_floor_f32|v_rndne_f32|v_fract_f32|v_trunc_f32|v_ceil_f32", "v_floor_f32"),
    (r"v_(sub|subrev|add)_u32|v_(and|or|xor|not)_b32|v_(lshlrev|lshrrev|ashrrev)_[bi]32|v_mov_b32|v_accvgpr|v_swap|v_nop", "v_add_u32"),
    (r"v_mul_f32|v_mul_legacy_f32|v_add_f32", "v_mul_f32"),
]


def cost_of(m, costs):
    if m in costs:
        return costs[m]
    for pat, rep in FAMILY:
        if re.match(pat, m) and rep in costs:
            return costs[rep]
    return None


CLASSES = ["ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "CVT", "INT32", "INT64"]
HOMOGENEOUS = {"TRANS_F32": 7.0, "ADD_F64": 3.65, "MUL_F64": 3.69, "FMA_F64": 3.72, "TRANS_F64": 13.93, "CVT": 3.58, "INT64": 3.83}


def model(groups, costs, counters, ms, clock_hz=PEAK_CLOCK_HZ):
    """→ issue cycles of one launch (fit + bounds) and the roofline fraction at duration `ms`"""
    import numpy as np
    from scipy.optimize import nnls
    G = sorted(groups)
    A = np.zeros((len(CLASSES) + 1, len(G)))
    cost_g = np.zeros(len(G))
    unknown = defaultdict(int)
    for j, g in enumerate(G):
        for mn in groups[g]:
            k = pmc_class(mn)
            if k in CLASSES:
                A[CLASSES.index(k), j] += 1
            A[-1, j] += 1
            c = cost_of(mn, costs)
            if c is None:
                unknown[mn] += 1
                c = 3.7
            cost_g[j] += c
    total = counters["SQ_INSTS_VALU"]
    b = np.array([counters.get("SQ_INSTS_VALU_" + k, 0.0) for k in CLASSES] + [total])
    scale = 1.0 / np.maximum(b, 1e-6 * total)
    scale[-1] *= 4.0                     # the total instruction count is the best-known of the 12 numbers
    w, resid = nnls(A * scale[:, None], b * scale)
    pred_total = float((A @ w)[-1])
    if pred_total > 0:
        w = w * (total / pred_total)     # the fitted mix, scaled to exactly the counted number of instructions
    fit = float(cost_g @ w)
    pred = A @ w
    # floor: every instruction costs at least 2 cycles; the homogeneous classes are known exactly
    known = sum(counters.get("SQ_INSTS_VALU_" + k, 0.0) * c for k, c in HOMOGENEOUS.items())
    n_known = sum(counters.get("SQ_INSTS_VALU_" + k, 0.0) for k in HOMOGENEOUS)
    lo = known + 2.0 * (total - n_known)
    hi = counters.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 or None
    if hi:
        fit = min(fit, hi)               # the counter-based ceiling binds when the per-mnemonic costs over-price a mix
    avail = ms * 1e-3 * clock_hz * N_SIMD
    top = sorted(range(len(G)), key=lambda j: -w[j] * cost_g[j])[:6]
    return {"issue_cycles": {"lo": lo, "model": fit, "hi": hi}, "simd_cycles_available": avail,
            "frac": {"lo": lo / avail, "model": fit / avail, "hi": (hi / avail) if hi else None},
            "mean_cycles_per_valu_instruction": fit / max(total, 1.0),
            "fit": {"loops": len(G), "loops_used": int((w > 0).sum()), "relative_residual": float(resid),
                    "counters_reproduced": {k: [float(p), float(t)] for k, p, t in zip(CLASSES + ["SQ_INSTS_VALU"], pred, b)},
                    "heaviest_loops": [{"loop": G[j], "valu_instructions": int(A[-1, j]), "executions": float(w[j]), "share_of_issue_cycles": float(w[j] * cost_g[j] / max(fit, 1.0)),
                                        "cycles_per_instruction": float(cost_g[j] / max(A[-1, j], 1))} for j in top]},
            "unmeasured_mnemonics": dict(sorted(unknown.items(), key=lambda x: -x[1])[:12])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--isa", required=True)
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--counters", required=True)
    ap.add_argument("--ms", type=float, required=True)
    ap.add_argument("--micro", default=os.path.join(ROOT, "profiles", "r02_valu_issue_cycles.json"))
    args = ap.parse_args()
    costs, _ = cost_table(args.micro)
    blocks = parse_kernel_isa(args.isa, args.kernel)
    if not blocks:
        sys.exit("kernel %r not found in %s" % (args.kernel, args.isa))
    counters = json.load(open(args.counters))
    print(json.dumps(model(blocks, costs, counters, args.ms), indent=1))


if __name__ == "__main__":
    main()
