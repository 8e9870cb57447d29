"""Static check of the hand-issued LDS loads in the MSV kernels' ISA (hipcc -S output).

The kernels issue ds_read_b64 from inline asm and count completions with s_waitcnt lgkmcnt(N).  The compiler does not
know the loads are asynchronous, so this walks the ISA and reports any instruction that reads or writes a VGPR whose
ds_read is still outstanding (including a later ds_read using such a register as its address: missing early-clobber).
Usage: check_lds_asm.py file.s   -> prints violations, exit status 1 if any.
"""
import re, sys
pend = []   # list of (lo,hi) in issue order
func = None
nbad = 0
def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1): out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.append(int(m.group(3)))
    return out
for ln, line in enumerate(open(sys.argv[1]), 1):
    s = line.strip()
    if s.startswith("_ZN3p7x") and s.endswith(":") is False and ":" in s and s[0] == "_":
        func = s.split(":")[0]; pend = []
    if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"): continue
    m = re.match(r"ds_read_b64 (v\[\d+:\d+\]), (v\d+)", s)
    if m:
        for r in regs(m.group(2)):
            if any(lo <= r <= hi for lo, hi in pend):
                print("ADDR-PENDING", func, ln, s); nbad += 1
        d = regs(m.group(1)); pend.append((d[0], d[-1])); continue
    m = re.match(r"s_waitcnt.*lgkmcnt\((\d+)\)", s)
    if m:
        n = int(m.group(1)); pend = pend[len(pend) - n:] if n else []
        continue
    if s.startswith("s_waitcnt"):
        continue
    used = regs(s)
    for r in used:
        if any(lo <= r <= hi for lo, hi in pend):
            print("TOUCH-PENDING", func, ln, s, pend); nbad += 1; break
print("violations", nbad)
sys.exit(1 if nbad else 0)
