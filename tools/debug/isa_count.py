"""Count instructions of a kernel in a hipcc -S listing, per basic block: vector / scalar / LDS / memory / matrix.

    python tools/debug/isa_count.py file.s <kernel-name-substring> [--ops]
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    show_ops = "--ops" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(name), l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], ["entry", collections.Counter(), collections.Counter()]
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            blocks.append(cur)
            cur = [m.group(1), collections.Counter(), collections.Counter()]
            continue
        t = l.strip()
        if not t or t.startswith((";", ".")):
            continue
        op = t.split()[0]
        cur[1][classify(op)] += 1
        cur[2][op] += 1
    blocks.append(cur)
    tot = collections.Counter()
    for nm, c, ops in blocks:
        n = sum(c.values())
        tot.update(c)
        if n >= 20:
            print(f"{nm:14s} n={n:5d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
            if show_ops:
                print("    " + " ".join(f"{k}:{v}" for k, v in ops.most_common(40)))
    print("total", dict(tot))


main()
