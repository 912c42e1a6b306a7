#!/usr/bin/env python3
"""Developer tool: instruction accounting of a kernel's ISA (hipcc -S output) by loop — for every innermost loop (label .. backward
branch to it) the number of instructions by issue class (matrix, other vector, scalar, LDS, vector memory, waits / nops), so that a
change to a kernel can be judged by what it does to the instruction streams the SIMDs issue (the port this repository's kernels run
out of: profiles/r05/valu_rate_microbench.txt).
    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S -o k.s kernel.hip ; python tools/isa_regions.py k.s [--min 40] [--kernel SUBSTR]
    python tools/isa_regions.py k.s --range 5150 5470     (one line range)"""
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier", "s_setprio", "s_sethalt")):
        return "wait"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_memtime", "s_memrealtime", "s_dcache")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def parse(path):
    lines = open(path).read().splitlines()
    items = []  # (lineno, kind, text) kind: label | inst
    for i, l in enumerate(lines, 1):
        t = l.strip()
        if not t or t.startswith((";", "//", ".")) and not re.match(r"^\.LBB\d+_\d+:", t):
            continue
        m = re.match(r"^(\.LBB\d+_\d+|[A-Za-z_][\w$.]*):", t)
        if m:
            items.append((i, "label", m.group(1)))
            continue
        if re.match(r"^\d+:$", t):  # local asm labels
            items.append((i, "label", "asm" + t[:-1] + "@" + str(i)))
            continue
        op = t.split()[0]
        if re.match(r"^[a-z_0-9]+$", op):
            items.append((i, "inst", t))
    return items


def count(items, a, b):
    c = {}
    for i, k, t in items:
        if k == "inst" and a <= i <= b:
            cl = classify(t.split()[0])
            c[cl] = c.get(cl, 0) + 1
    return c


def markers(path, kernel_substr):
    """--markers: nam_wn_reg_kernel built with -DNAM_WR_MARKERS (kernel_wn_reg.hip: '; nam_op program P op I type T stages N' in
    front of every op of a compiled-in program): instructions by class per op, in program order, for one kernel function."""
    lines = open(path).read().splitlines()
    names = {0: "ARRAY_BEGIN", 1: "LAYER", 2: "ARRAY_END", 3: "SET_COND", 4: "OUTPUT", 5: "RUN", 6: "ARRAY_END_K", 7: "POST_HEAD", -1: "(per-buffer remainder)"}
    cur_fn, rows, cur = None, [], None
    for i, l in enumerate(lines, 1):
        t = l.strip()
        m = re.match(r"^([A-Za-z_][\w$.]*):", t)
        if m and not t.startswith(".L"):
            cur_fn = m.group(1)
            cur = None
            continue
        if cur_fn is None or kernel_substr not in cur_fn:
            continue
        m = re.match(r"^; nam_op program (-?\d+) op (-?\d+) type (-?\d+) stages (\d+)", t)
        if m:
            cur = {"program": int(m.group(1)), "op": int(m.group(2)), "type": names.get(int(m.group(3)), m.group(3)), "c": {}}
            rows.append(cur)
            continue
        if cur is None or not t or t.startswith((";", "//", ".")) or re.match(r"^\S+:$", t):
            continue
        if t.startswith("s_endpgm"):
            cur = None
            continue
        op = t.split()[0]
        if re.match(r"^[a-z_0-9]+$", op):
            cl = classify(op)
            cur["c"][cl] = cur["c"].get(cl, 0) + 1
    print(f"{'program':>7s} {'op':>3s} {'type':22s} {'valu':>6s} {'salu':>6s} {'lds':>6s} {'vmem':>6s} {'wait':>6s} {'smem':>5s}")
    tot = {}
    for r in rows:
        c = r["c"]
        for k, v in c.items():
            tot[k] = tot.get(k, 0) + v
        print(f"{r['program']:7d} {r['op']:3d} {r['type']:22s} {c.get('valu', 0):6d} {c.get('salu', 0):6d} {c.get('lds', 0):6d} {c.get('vmem', 0):6d} {c.get('wait', 0):6d} {c.get('smem', 0):5d}")
    print(f"{'':7s} {'':3s} {'total':22s} {tot.get('valu', 0):6d} {tot.get('salu', 0):6d} {tot.get('lds', 0):6d} {tot.get('vmem', 0):6d} {tot.get('wait', 0):6d} {tot.get('smem', 0):5d}")


def main():
    args = sys.argv[1:]
    path = args[0]
    if "--markers" in args:
        return markers(path, args[args.index("--markers") + 1])
    items = parse(path)
    if "--range" in args:
        k = args.index("--range")
        a, b = int(args[k + 1]), int(args[k + 2])
        print(a, b, count(items, a, b))
        return
    mn = int(args[args.index("--min") + 1]) if "--min" in args else 40
    label_line = {t: i for i, k, t in items if k == "label"}
    loops = []
    for i, k, t in items:
        if k != "inst":
            continue
        m = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)|^s_branch\s+(\.LBB\d+_\d+)", t)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in label_line and label_line[tgt] < i:
                loops.append((label_line[tgt], i, tgt))
    # innermost first: drop loops that contain another loop entirely? keep all, mark nesting
    loops.sort()
    print(f"{'loop':14s} {'lines':>15s} {'mfma':>6s} {'valu':>6s} {'salu':>6s} {'lds':>6s} {'vmem':>6s} {'wait':>6s} {'smem':>5s}  inner")
    for a, b, tgt in loops:
        c = count(items, a, b)
        tot = sum(c.values())
        if tot < mn:
            continue
        inner = sum(1 for a2, b2, _ in loops if a < a2 and b2 < b)
        print(f"{tgt:14s} {a:7d}-{b:<7d} {c.get('mfma', 0):6d} {c.get('valu', 0):6d} {c.get('salu', 0):6d} {c.get('lds', 0):6d} {c.get('vmem', 0):6d} {c.get('wait', 0):6d} {c.get('smem', 0):5d}  {inner}")


if __name__ == "__main__":
    main()
