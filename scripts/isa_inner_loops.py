"""Static check of the generated ISA: row loops that pay a memory round trip per iteration.

`vmcnt` retires in order, so inside a loop that both loads and stores, a wait for a load that is placed AFTER a store of
the same iteration (or at the top of the next) also waits for that store's round trip -- and a load that is consumed in
the iteration that issues it puts its own latency on the row.  For every innermost loop of more than 60 lines this prints
the loads, stores and vmcnt waits, and flags ("SERIAL") the loops in which some vmcnt wait follows a store or a load of
the same iteration that is younger than the data it waits for cannot be told apart -- i.e. any wait that is not ahead of
every store of the body.  A loop whose only wait sits before its first store (the explicit wait of the envelope kernel's
decoding pass) or that has no loads at all (Backward: stores only) is fine.

usage: isa_inner_loops.py file.s [name-filter] [--flagged]"""
import re
import sys


def loops(path, flt=""):
    """(function, header label, lines, loads, stores, vmcnt waits, serial) of every innermost loop.  The body of a loop is
    its header block plus every block the compiler labels `in Loop: Header=<that block>`."""
    lines = open(path).read().split("\n")
    # split into (function, label, comment, block lines)
    blocks = []
    func = None
    cur = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            func = m.group(1)
            cur = None
            continue
        m = re.match(r"^\.L(BB\d+_\d+):(.*)$", l)
        if m and func:
            cur = {"func": func, "label": m.group(1), "comment": m.group(2), "lines": []}
            blocks.append(cur)
            continue
        if cur is not None:
            if l.lstrip().startswith(";") and not cur["lines"]:
                cur["comment"] += l                       # the loop comments continue on the lines after the label
            else:
                cur["lines"].append(l)
    out = []
    for h in blocks:
        if "Inner Loop Header" not in h["comment"] or flt not in h["func"]:
            continue
        body = list(h["lines"])
        for b in blocks:
            if b["func"] == h["func"] and re.search(r"in Loop: Header=" + h["label"] + r"\b", b["comment"]):
                body += b["lines"]
        if len(body) <= 60:
            continue
        loads = [n for n, b in enumerate(body) if re.search(r"\b(global|buffer|flat)_load", b)]
        stores = [n for n, b in enumerate(body) if re.search(r"\b(global|buffer|flat)_store", b)]
        waits = [n for n, b in enumerate(body) if re.search(r"s_waitcnt.*vmcnt\(\d+\)", b)]
        fine = not loads or not waits or (stores and all(w < stores[0] for w in waits))
        out.append((h["func"], ".L" + h["label"], len(body), len(loads), len(stores), len(waits), not fine))
    return out


if __name__ == "__main__":
    flagged_only = "--flagged" in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    for func, lab, n, nl, ns, nw, serial in loops(args[0], args[1] if len(args) > 1 else ""):
        if flagged_only and not serial:
            continue
        print(f"{func[:72]:72s} {lab:12s} {n:5d} lines, loads {nl:3d}, stores {ns:3d}, vmcnt waits {nw}{'  SERIAL' if serial else ''}")
