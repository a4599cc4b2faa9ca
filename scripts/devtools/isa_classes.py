#!/usr/bin/env python3
"""Instruction classes of a kernel's basic blocks, from hipcc's assembly (round-5 review, item 4: "a disassembly count of the pixel
loop by class").

  hipcc <FLAGS> --cuda-device-only -S -o maze.s xenoverse_amd/csrc/maze.hip
  python scripts/devtools/isa_classes.py maze.s '_Z19maze_raycast_kernelILb0ELb1ELi0ELb0EEv8MazeArgsPhPf' [min_instructions]

Prints, for every basic block of at least `min_instructions` instructions (default 60), the number of instructions per class;
a block that ends in (or falls into) a backward branch is marked `loop`.  Classes:
  f64 arith (v_fma/add/mul/max/min/..._f64), f64 other (cvt to/from f64, floor, rcp, cmp_f64 ...), f32, int/bit VALU, select
  (v_cndmask), mov, LDS (ds_*), vector memory (global_/buffer_/flat_), scalar (s_*, incl. waitcnt), other"""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith("s_"):
        return "scalar"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_cndmask"):
        return "select"
    if op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane", "v_permlane", "v_swap")):
        return "mov"
    if "f64" in op:
        if op.startswith(("v_fma_f64", "v_add_f64", "v_mul_f64", "v_max_f64", "v_min_f64", "v_fmac_f64", "v_pk_")):
            return "f64 arith"
        return "f64 other"
    if "f32" in op or "f16" in op:
        return "f32"
    if op.startswith("v_"):
        return "int/bit"
    return "other"


def main():
    path, kern = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    lines = open(path).read().splitlines()
    start = next(i for i, ln in enumerate(lines) if ln.startswith(kern + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = []
    label_line = {}
    for i in range(start + 1, end):
        ln = lines[i].split(";")[0].rstrip()
        if not ln.strip():
            continue
        m = re.match(r"^(\.L[A-Za-z0-9_]+):", ln)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            label_line[cur] = i
            continue
        if ln.strip().startswith("."):
            continue
        tok = ln.split()
        blocks[cur].append((i, tok[0], tok[1:] if len(tok) > 1 else []))
    total = Counter()
    print("kernel %s: %d basic blocks, %d instructions" % (kern, len(blocks), sum(len(b) for b in blocks.values())))
    names = list(blocks)
    for bi, name in enumerate(names):
        ins = blocks[name]
        c = Counter(classify(op) for _, op, _ in ins)
        total.update(c)
        back = False
        for i, op, args in ins:
            if op.startswith(("s_cbranch", "s_branch")) and args:
                tgt = args[-1]
                if tgt in label_line and label_line[tgt] <= label_line.get(name, start):
                    back = True
        if len(ins) >= min_n:
            valu = sum(v for k, v in c.items() if k not in ("scalar", "LDS", "vmem", "other"))
            print("%-12s %5d instr%s  VALU %4d | %s" % (name, len(ins), " loop" if back else "     ", valu,
                                                        ", ".join("%s %d" % kv for kv in sorted(c.items(), key=lambda kv: -kv[1]))))
    print("whole kernel:", ", ".join("%s %d" % kv for kv in sorted(total.items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    main()
