#!/usr/bin/env python3
"""Static instruction table of a gfx950 kernel by loop (VERDICT r04 task 1a).

    hipcc ... --cuda-device-only -S -o event.s sigtk_amd/csrc/event_kernels.hip     (tools/isa_phases.py --build does it)
    python tools/isa_phases.py event.s _ZN3sgk7k_eventILi3EsEEvNS_6EvArgsE

Finds the natural loops of the function (backward branches), and prints for every loop and for the straight-line code
around them the instruction counts by issue class:

    valu_fast  v_add/sub/mul/fma_f32, v_mov, logic, int add, right shifts        (2.3 cycles per wave64 instruction,
    valu_slow  everything f64, conversions, v_cmp, v_cndmask, min/max, packed,    tools/valu_rate.hip)
               left shifts / bfe / perm / mul_lo / add3                          (4.45 cycles)
    valu_trans v_rcp/rsq/sqrt/exp/log                                            (8.5 / 16 cycles)
    dpp        VALU instructions with a dpp / row_ / wave_ modifier, v_readlane, v_writelane, ds_bpermute, ds_swizzle
    salu, lds (ds_*), vmem (global_/buffer_/flat_), scratch, branch, waitcnt

Loop bodies are reported EXCLUSIVE of their inner loops, so the rows add up to the function.  The dynamic count of a
phase is (static count of its row) x (trips), the trips being a property of the input: the caller supplies them with
--trips name=count after naming the loops with --name line=name (see profiles/r05_event_instruction_table.md).
"""
import argparse
import os
import re
import subprocess
import sys

SLOW = re.compile(r"^v_(.*_f64|cvt_|cmp|cmpx|cndmask|max|min|med3|pk_|lshlrev|bfe|bfi|perm|mul_lo|mul_hi|add3|lshl_|"
                  r"and_or|or3|xad|mad_|sad_|ldexp|frexp|fract|trunc|ceil|floor|rndne|alignbit|alignbyte|mbcnt|bcnt|ffb)")
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)")
CROSS = re.compile(r"(row_|wave_|quad_perm|row_bcast|dpp)")


def classify(op, rest):
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("ds_"):
        return "dpp" if op.startswith(("ds_bpermute", "ds_permute", "ds_swizzle")) else "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("s_"):
        if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call", "s_endpgm")):
            return "branch"
        if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier", "s_setprio")):
            return "wait"
        return "salu"
    if op.startswith("v_"):
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")) or CROSS.search(rest):
            return "dpp"
        if TRANS.match(op):
            return "valu_trans"
        if SLOW.match(op):
            return "valu_slow"
        return "valu_fast"
    return "other"


CLASSES = ["valu_fast", "valu_slow", "valu_trans", "dpp", "salu", "lds", "vmem", "scratch", "branch", "wait", "other"]


def function_lines(path, name):
    out, on = [], False
    for ln in open(path):
        if ln.startswith(name + ":"):
            on = True
            continue
        if on:
            if ln.lstrip().startswith((".end_amdhsa_kernel", ".Lfunc_end")):
                break
            out.append(ln.rstrip("\n"))
    if not out:
        sys.exit("function %s not found in %s" % (name, path))
    return out


def analyse(lines):
    insts = []   # (line index in `lines`, op, rest)
    labels = {}
    for i, ln in enumerate(lines):
        t = ln.strip()
        if not t or t.startswith((";", "//", ".")) and not t.endswith(":"):
            continue
        if t.endswith(":") and not t.startswith(("s_", "v_")):
            labels[t[:-1]] = len(insts)
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        parts = t.split(None, 1)
        insts.append((i, parts[0], parts[1] if len(parts) > 1 else ""))
    loops = []
    for k, (_, op, rest) in enumerate(insts):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = rest.strip()
            if tgt in labels and labels[tgt] <= k:
                loops.append((labels[tgt], k, tgt))
    # merge loops with the same header (several back edges)
    by_head = {}
    for a, b, t in loops:
        by_head[a] = (a, max(b, by_head.get(a, (a, b, t))[1]), t)
    loops = sorted(by_head.values(), key=lambda x: (x[0], -x[1]))
    return insts, loops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm", nargs="?")
    ap.add_argument("function", nargs="?")
    ap.add_argument("--build", metavar="HIP_SOURCE", help="compile this source to <asm> first (device only, the library's flags)")
    ap.add_argument("--define", action="append", default=[], help="-D for --build")
    ap.add_argument("--min", type=int, default=40, help="loops with fewer instructions are folded into their parent")
    ap.add_argument("--name", action="append", default=[], help="HEADER_INSTRUCTION_INDEX=name for a loop")
    ap.add_argument("--list", action="store_true", help="list the kernels / functions of the file with their resources")
    a = ap.parse_args()
    if a.build:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
               "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-unused-function", "-Wno-bitwise-instead-of-logical",
               "-Wno-c++20-extensions", "-Wno-pass-failed", "-I", os.path.join(root, "include"), "--cuda-device-only", "-S",
               "-o", a.asm, a.build] + ["-D" + d for d in a.define]
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    if a.list:
        txt = open(a.asm).read()
        for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n){0,12}?", txt):
            blk = txt[m.start():m.start() + 900]
            g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [None, "?"])[1]
            print("%-60s vgpr %s  sgpr %s  vgpr_spill %s  sgpr_spill %s  scratch %s B  lds %s B" %
                  (m.group(1), g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
                   g("private_segment_fixed_size"), g("group_segment_fixed_size")))
        return
    lines = function_lines(a.asm, a.function)
    insts, loops = analyse(lines)
    names = dict((int(x.split("=")[0]), x.split("=")[1]) for x in a.name)
    # fold small loops
    big = [l for l in loops if l[1] - l[0] + 1 >= a.min]
    # exclusive ranges: assign every instruction to the innermost big loop containing it
    owner = [-1] * len(insts)
    for li, (s, e, _) in enumerate(big):       # sorted outer-first, so inner loops overwrite
        for k in range(s, e + 1):
            owner[k] = li
    rows = {}
    order = []
    # straight-line segments between loops at top level get their own rows
    seg = 0
    prev = None
    for k, (_, op, rest) in enumerate(insts):
        o = owner[k]
        if o == -1:
            if prev != -1:
                seg += 1
            key = ("straight", seg)
        else:
            key = ("loop", o)
        prev = o
        if key not in rows:
            rows[key] = dict.fromkeys(CLASSES, 0)
            rows[key]["first"] = k
            order.append(key)
        rows[key][classify(op, rest)] += 1
        rows[key]["last"] = k
    print("# %s: %d instructions, %d loops (%d of >= %d instructions)" % (a.function, len(insts), len(loops), len(big), a.min))
    hdr = "%-34s %7s %7s" % ("block [first..last instr]", "total", "bytes~") + "".join("%11s" % c for c in CLASSES)
    print(hdr)
    for key in order:
        r = rows[key]
        tot = sum(r[c] for c in CLASSES)
        if key[0] == "loop":
            s, e, t = big[key[1]]
            depth = sum(1 for (s2, e2, _) in big if s2 <= s and e2 >= e) - 1
            nm = names.get(s, "loop@%d" % s)
            label = "%s%s [%d..%d]" % ("  " * depth, nm, s, e)
        else:
            label = "straight [%d..%d]" % (r["first"], r["last"])
        print("%-34s %7d %7s" % (label[:34], tot, "") + "".join("%11d" % r[c] for c in CLASSES))
    tot = dict((c, sum(rows[k][c] for k in order)) for c in CLASSES)
    print("%-34s %7d %7s" % ("function", sum(tot.values()), "") + "".join("%11d" % tot[c] for c in CLASSES))


if __name__ == "__main__":
    main()
