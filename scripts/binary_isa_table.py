#!/usr/bin/env python3
"""Instruction classes per cell-row of binary_stage_kernel<BinFast, COMBINE, false> (BASELINE config 3; mara3_amd/csrc/binary_kernel.hpp), gfx950:
  (1) the pieces of the row step compiled alone (scripts/probes/binary_pieces.hip, probe overhead subtracted) and their sum per cell-row;
  (2) the kernel's OWN row loop as the compiler emits it (binary_fast.hip -> assembly: the loop of three row steps, divided by three).
usage: python scripts/binary_isa_table.py > profiles/r05/binary_instruction_classes.md"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "mara3_amd", "csrc")]
EXTRA = sys.argv[1:]          # e.g. -DMH_BIN_... to tabulate a variant


def asm(src):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "p.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + EXTRA + [src, "-o", out], stderr=subprocess.DEVNULL)
        return open(out).read()


CLASSES = [("fp64 fma", r"v_fma_f64|v_fmac_f64"), ("fp64 mul", r"v_mul_f64"), ("fp64 add / sub", r"v_add_f64"),
           ("v_rcp_f64 (quarter rate)", r"v_rcp_f64"), ("v_rsq_f64 (quarter rate)", r"v_rsq_f64"), ("other transcendental / exp pieces (v_exp, v_ldexp, v_frexp, v_rndne, v_cvt)", r"v_exp|v_ldexp|v_frexp|v_rndne|v_cvt|v_trunc|v_fract|v_floor"),
           ("v_min / v_max f64", r"v_min_f64|v_max_f64"),
           ("compares", r"v_cmp"), ("selects (v_cndmask_b32)", r"v_cndmask"), ("sign / bit logic (v_bfi, v_and, v_or, v_xor)", r"v_bfi|v_and_b32|v_or_b32|v_xor_b32|v_and_or|v_not"),
           ("DPP moves (v_mov_b32_dpp)", r"v_mov_b32_dpp"), ("lane reads / writes (v_readlane, v_writelane, v_readfirstlane)", r"v_readlane|v_writelane|v_readfirstlane"),
           ("plain moves", r"v_mov_b32|v_mov_b64|v_accvgpr|v_pk_mov"),
           ("integer / address", r"v_lshl|v_add_u32|v_add_co|v_addc|v_mad_u|v_mul_lo|v_mul_hi|v_ashr|v_lshr|v_sub_u32|v_add3|v_mad_i|v_mbcnt|v_sub_co|v_subb|v_add_nc|v_sub_nc|v_lshlrev|v_ashrrev|v_lshrrev|v_bfe|v_min_i|v_max_i|v_min_u|v_max_u|v_sub_i|v_add_i|v_subrev"),
           ("other VALU", r"v_")]
NAMES = [n for n, _ in CLASSES]


def classify(lines):
    c = collections.Counter()
    for l in lines:
        m = re.match(r"\s+(v_[a-z0-9_]+)", l)
        if not m:
            continue
        op = m.group(1)
        if re.search(r"\b(wave_sh[lr]|row_sh[lr]|quad_perm|row_bcast|wave_ro[lr])", l):
            c["DPP moves (v_mov_b32_dpp)"] += 1
            continue
        for name, pat in CLASSES:
            if re.match(pat, op):
                c[name] += 1
                break
    return c


def other(lines):
    """non-VALU instructions of a block: scalar, LDS, vector memory, waits"""
    c = collections.Counter()
    for l in lines:
        m = re.match(r"\s+((?:s|ds|buffer|global|flat|scratch)_[a-z0-9_]+)", l)
        if m:
            op = m.group(1)
            key = ("s_waitcnt" if op.startswith("s_waitcnt") else "s_load / s_buffer_load" if re.match(r"s_(buffer_)?load", op) else "SALU" if op.startswith("s_") else
                   "LDS" if op.startswith("ds_") else "scratch (spill)" if op.startswith("scratch_") else "vector memory loads" if "load" in op else "vector memory stores")
            c[key] += 1
    return c


txt = asm(os.path.join(ROOT, "scripts", "probes", "binary_pieces.hip"))
P = {}
for m in re.finditer(r"\n(piece_\w+):[^\n]*\n(.*?)\.Lfunc_end", txt, flags=re.S):
    P[m.group(1)] = classify(m.group(2).split("\n"))
base = P["piece_baseline"]


def net(name, minus=()):
    c = collections.Counter(P[name])
    c.subtract(base)
    for other_piece in minus:
        c.subtract(net(other_piece))
    c["integer / address"] = 0          # the probe's own addressing of its operands
    return collections.Counter({k: max(v, 0) for k, v in c.items()})


def print_table(per_row, head):
    print("| class | " + " | ".join(per_row) + " | per cell-row |")
    print("|---|" + "---:|" * (len(per_row) + 1))
    total = collections.Counter()
    for n in NAMES:
        row, s = [], 0
        for piece, (c, mult) in per_row.items():
            row.append("%d x %d" % (c[n], mult) if c[n] else "")
            s += c[n] * mult
        total[n] = s
        if s:
            print("| %s | %s | %d |" % (n, " | ".join(row), s))
    print("| **all VALU** | %s | **%d** |" % (" | ".join(str(sum(c.values()) * m) for c, m in per_row.values()), sum(total.values())))
    print()
    return total


print("# `binary_stage_kernel<BinFast, COMBINE, false>` (BASELINE config 3, `advance_u`): instruction classes per cell-row (gfx950, hipcc -O3 -ffp-contract=off)\n")
print("Reference: `src/subprog_binary_scheme.cpp` - `intercell_flux_u` :268-293 with `cs2_at_position` :160-175 (two softened potentials per face), `nu_at_position` :177-193,")
print("`viscous_flux` :220-262, `iso2d::riemann_hlle` (`physics_iso2d.hpp:488-506`), `source_terms_u` :345-411, `block_update_u` :568-587, totals :390-408.\n")
print("## 1. The pieces of one cell-row, compiled alone (`scripts/probes/binary_pieces.hip`, probe overhead subtracted; `x n` = times per cell-row)\n")
print("Every cell-row evaluates ONE axis-0 face (carried to the next row) and ONE axis-1 face (handed to the left neighbour by DPP): no face is evaluated twice.")
print("C3's defaults: alpha viscosity (no tanh cut-off), non-axisymmetric sound speed (two inverse roots per position), both sinks out of range for most waves (one `exp` per body where a lane of the wave is within 38 sink radii).\n")
for combine in (False, True):
    su = "piece_sources_update_combine" if combine else "piece_sources_update"
    per_row = collections.OrderedDict([
        ("recover_primitive (1)", (net("piece_c2p"), 1)),
        ("limited slopes per length, 3 variables (2 axes)", (net("piece_plm"), 2)),
        ("DPP moves of 3 doubles (6: P left / right, Gy, Gx, P for the y face, Fy back)", (net("piece_dpp3"), 6)),
        ("face: cs2 at position (2 faces)", (net("piece_cs2"), 2)),
        ("face: nu at position, given cs2 (2 faces)", (net("piece_cs2_nu", minus=("piece_cs2",)), 2)),
        ("face: HLLE, given cs2 (axis 0)", (net("piece_hlle0"), 1)),
        ("face: HLLE (axis 1)", (net("piece_hlle1"), 1)),
        ("face: states + viscous stress, and what nu and HLLE share (axis 0)", (net("piece_face0", minus=("piece_cs2_nu", "piece_hlle0")), 1)),
        ("face: states + viscous stress (axis 1)", (net("piece_face1", minus=("piece_cs2_nu", "piece_hlle1")), 1)),
        ("gravity of both bodies (1)", (net("piece_gravity2"), 1)),
        ("sink ranges of both bodies, wave out of range (1)", (net("piece_sink2_far"), 1)),
        ("buffer, floor, totals, update%s (1)" % (" + RK average" if combine else ""), (net(su, minus=("piece_gravity2", "piece_sink2_far")), 1))])
    print("### %s RK2 stage (`COMBINE = %s`)\n" % ("second" if combine else "first", "true" if combine else "false"))
    tot = print_table(per_row, "")
    flops = 2 * tot["fp64 fma"] + tot["fp64 mul"] + tot["fp64 add / sub"] + tot["v_rcp_f64 (quarter rate)"] + tot["v_rsq_f64 (quarter rate)"]
    s = sum(tot.values())
    print("**%d VALU instructions per cell-row = %d fma + %d mul + %d add + %d rcp + %d rsq + %d min / max + %d DPP + %d other; %d flop, %.2f flop per issue slot.**\n"
          % (s, tot["fp64 fma"], tot["fp64 mul"], tot["fp64 add / sub"], tot["v_rcp_f64 (quarter rate)"], tot["v_rsq_f64 (quarter rate)"], tot["v_min / v_max f64"], tot["DPP moves (v_mov_b32_dpp)"],
             s - tot["fp64 fma"] - tot["fp64 mul"] - tot["fp64 add / sub"] - tot["v_rcp_f64 (quarter rate)"] - tot["v_rsq_f64 (quarter rate)"] - tot["v_min / v_max f64"] - tot["DPP moves (v_mov_b32_dpp)"],
             flops, flops / s))
near = sum(net("piece_sink2_near").values()) - sum(net("piece_sink2_far").values())
print("A wave with a lane within 38 sink radii of a body also evaluates `exp` (device libm): **+%d** VALU instructions per body and cell-row (counted statically: both bodies %d).\n"
      % (near // 2, near))

# ---- (2) the kernel's own loop
print("## 2. The kernel's own row loop (`binary_fast.hip` compiled to assembly: the loop of three row steps, per row)\n")
ktxt = asm(os.path.join(ROOT, "mara3_amd", "csrc", "binary_fast.hip"))
print("| kernel | VGPRs | SGPRs | spilled SGPRs / VGPRs | scratch bytes | loop, hot path: VALU per row | cold blocks per row (tanh, exp, status) | of them fma / mul / add | rcp + rsq | min / max | DPP | v_readlane + v_writelane | moves | integer | other VALU | SALU per row | s_load per row | vector loads / stores per row | s_waitcnt per row |")
print("|---|---:|---:|---|---:|---:|---:|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---|---:|")
for m in re.finditer(r"\n(_ZN2mh19binary_stage_kernelINS_8BinFastTILb([01])EEELb([01])ELb0EEEvNS_17BinaryStageParamsE):[^\n]*\n(.*?)\.Lfunc_end", ktxt, flags=re.S):
    body = m.group(4)
    disk, comb = m.group(2) == "1", m.group(3) == "1"
    meta = ktxt[m.end():m.end() + 6000]
    g = lambda pat: (re.search(pat, meta) or [None, "?"])[1]
    vg, sg = g(r"; NumVgprs: (\d+)"), g(r"; NumSgprs: (\d+)")
    amd = re.search(r"\.name:\s+%s\n(?:.*\n)*?\s+\.sgpr_spill_count:\s+(\d+)(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)" % re.escape(m.group(1)), ktxt)
    ss, vs = (amd.group(1), amd.group(2)) if amd else ("?", "?")
    scratch = g(r"; ScratchSize: (\d+)")
    # the row loop: from the header of the innermost loop with the most VALU instructions to its back edge; inside it, basic blocks that hold device
    # libm code (exp of a sink in range, tanh of the viscosity cut-off: v_rndne_f64) or a status report (atomics) are COLD in C3 and counted apart
    lines = body.split("\n")
    labels = {}
    for i, l in enumerate(lines):
        mm = re.match(r"(\.LBB\d+_\d+):", l)
        if mm:
            labels[mm.group(1)] = i
    best = None
    for i, l in enumerate(lines):
        mm = re.match(r"\s+s_c?branch\w* (\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            seg = lines[labels[mm.group(1)]:i + 1]
            n = sum(classify(seg).values())
            if best is None or n > best[0]:
                best = (n, seg)
    blocks = [[]]
    for l in best[1]:
        if re.match(r"\.LBB\d+_\d+:", l):
            blocks.append([])
        blocks[-1].append(l)
        if re.match(r"\s+s_c?branch", l):
            blocks.append([])
    seg, cold = [], []
    for b in blocks:
        (cold if re.search(r"v_rndne_f64|atomic", "\n".join(b)) else seg).extend(b)
    n, ncold = sum(classify(seg).values()), sum(classify(cold).values())
    c, o = classify(seg), other(seg)
    per = lambda x: "%.1f" % (x / 3.0)
    print("| `<%s, %s, false>` | %s | %s | %s / %s | %s | **%s** | %s | %s / %s / %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s / %s | %s |"
          % ("BinFastT<DISK = true>" if disk else "BinFastT<false> (generic)", "true" if comb else "false", vg, sg, ss, vs, scratch, per(n), per(ncold), per(c["fp64 fma"]), per(c["fp64 mul"]), per(c["fp64 add / sub"]),
             per(c["v_rcp_f64 (quarter rate)"] + c["v_rsq_f64 (quarter rate)"]), per(c["v_min / v_max f64"]), per(c["DPP moves (v_mov_b32_dpp)"]),
             per(c["lane reads / writes (v_readlane, v_writelane, v_readfirstlane)"]), per(c["plain moves"]), per(c["integer / address"]),
             per(c["other VALU"] + c["compares"] + c["selects (v_cndmask_b32)"] + c["sign / bit logic (v_bfi, v_and, v_or, v_xor)"] + c["other transcendental / exp pieces (v_exp, v_ldexp, v_frexp, v_rndne, v_cvt)"]),
             per(o["SALU"]), per(o["s_load / s_buffer_load"]), per(o["vector memory loads"]), per(o["vector memory stores"]), per(o["s_waitcnt"])))
print()
