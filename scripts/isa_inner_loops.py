"""Static check of the generated ISA: innermost loops that both load from global memory and wait with vmcnt(0) -- a
row loop of that kind pays a memory round trip per iteration (and, when it stores too, the stores' as well: the
counter retires in order).  usage: isa_inner_loops.py file.s [name-filter]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
func = None
i = 0
while i < len(lines):
    l = lines[i]
    m = re.match(r"^(_Z\w+):", l)
    if m:
        func = m.group(1)
    m = re.match(r"^(\.LBB\d+_\d+):.*Inner Loop Header", l)
    if m and func and flt in func:
        lab = m.group(1)
        j = i + 1
        while j < len(lines) and not re.search(r"s_cbranch\w+ " + re.escape(lab) + r"\b|s_branch " + re.escape(lab) + r"\b", lines[j]):
            if re.match(r"^_Z\w+:", lines[j]): break
            j += 1
        body = lines[i:j]
        nload = sum("global_load" in b or "buffer_load" in b for b in body)
        nstore = sum("global_store" in b or "buffer_store" in b for b in body)
        nwait0 = sum(bool(re.search(r"vmcnt\(0\)", b)) for b in body)
        if nload and nwait0 and len(body) > 60:
            print(f"{func[:70]:70s} {lab:12s} {len(body):5d} lines, loads {nload:3d}, stores {nstore:3d}, vmcnt(0) x{nwait0}")
    i += 1
