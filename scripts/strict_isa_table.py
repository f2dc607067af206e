#!/usr/bin/env python3
"""Instruction classes of the pieces of the STRICT 2-D Euler PLM + HLLE row step (scripts/probes/strict_pieces.hip compiled for gfx950), and
their sum per cell-row: one recover_primitive, two slopes, two pairs of face states, two HLLE problems, one update (+ RK average).
The `baseline` kernel (loads, stores, index arithmetic of the probe itself) is subtracted from every piece. Cold fallbacks of the shared-
denominator division (behind s_cbranch_exec*) are counted apart. usage: python scripts/strict_isa_table.py > profiles/r03/strict_hlle_instruction_classes.md"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "p.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only", "-DMH_PROBE_NO_DIV_FALLBACK",
                           "-I", os.path.join(ROOT, "mara3_amd", "csrc"), os.path.join(ROOT, "scripts", "probes", "strict_pieces.hip"), "-o", out])
    txt = open(out).read()

CLASSES = [("fp64 add / sub", r"v_add_f64"), ("fp64 mul", r"v_mul_f64"), ("fp64 fma (division / sqrt refinement only)", r"v_fma_f64|v_fmac_f64"),
           ("v_div_scale_f64", r"v_div_scale_f64"), ("v_div_fmas_f64", r"v_div_fmas_f64"), ("v_div_fixup_f64", r"v_div_fixup_f64"),
           ("v_rcp_f64 (quarter rate)", r"v_rcp_f64"), ("v_rsq_f64 / v_sqrt_f64 (quarter rate)", r"v_rsq_f64|v_sqrt_f64"), ("v_min / v_max f64", r"v_min_f64|v_max_f64"),
           ("v_ldexp / v_frexp (sqrt scaling)", r"v_ldexp_f64|v_frexp"), ("compares", r"v_cmp"), ("selects (v_cndmask_b32)", r"v_cndmask"), ("sign logic (v_bfi, v_and, v_or, v_xor)", r"v_bfi|v_and_b32|v_or_b32|v_xor_b32"),
           ("moves", r"v_mov_b32|v_mov_b64|v_accvgpr"), ("integer / address (probe residue)", r"v_lshl|v_add_u32|v_add_co|v_addc|v_mad_u|v_mul_lo|v_mul_hi|v_ashr|v_lshr|v_sub_u32|v_add3|v_mad_i|v_mbcnt"), ("other VALU", r"v_")]

def classify(lines):
    c = collections.Counter()
    for l in lines:
        m = re.match(r"\s+(v_[a-z0-9_]+)", l)
        if not m:
            continue
        for name, pat in CLASSES:
            if re.match(pat, m.group(1)):
                c[name] += 1
                break
    return c

def pieces():
    res = {}
    for m in re.finditer(r"\n(piece_\w+):[^\n]*\n(.*?)\.Lfunc_end", txt, flags=re.S):
        name, body = m.group(1), m.group(2)
        lines = body.split("\n")
        hot, cold = lines, []          # MH_PROBE_NO_DIV_FALLBACK: the probe holds the hot path only (the guard's compares stay)
        res[name] = (classify(hot), classify(cold))
    return res

P = pieces()
base = P["piece_baseline"][0]
def net(name):
    c = collections.Counter(P[name][0])
    c.subtract(base)
    return collections.Counter({k: max(v, 0) for k, v in c.items()})
per_row = {"recover_primitive (1)": (net("piece_c2p"), 1), "plm_gradient, 5 variables (2 axes)": (net("piece_plm"), 2), "face states P +- G/2 (2 axes)": (net("piece_faces"), 2),
           "riemann_hlle axis 0 (1)": (net("piece_hlle0"), 1), "riemann_hlle axis 1 (1)": (net("piece_hlle1"), 1), "update + RK average (1)": (net("piece_update"), 1)}
names = [n for n, _ in CLASSES]
print("# STRICT 2-D Euler, PLM + HLLE: instruction classes per cell-row (hot path; gfx950, hipcc -O3 -ffp-contract=off)\n")
print("Pieces compiled alone (`scripts/probes/strict_pieces.hip`), probe overhead subtracted; `x n` = times per cell-row.\n")
print("| class | " + " | ".join(per_row) + " | per cell-row |")
print("|---|" + "---:|" * (len(per_row) + 1))
total = collections.Counter()
for n in names:
    row, s = [], 0
    for piece, (c, mult) in per_row.items():
        row.append("%d x %d" % (c[n], mult) if c[n] else "")
        s += c[n] * mult
    total[n] = s
    if s:
        print("| %s | %s | %d |" % (n, " | ".join(row), s))
print("| **all VALU** | %s | **%d** |" % (" | ".join(str(sum(c.values()) * m) for c, m in per_row.values()), sum(total.values())))
print("\nThe shared-denominator division's fallback (`x / den` in full, behind a branch never taken on ordinary data: a numerator whose exponent makes "
      "v_div_scale rescale the denominator) is compiled out of the probe; the comparisons that guard it are counted.")
