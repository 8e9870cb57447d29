"""Offline explorer for the sampled-null2 deviation.

Input: per-hit ensemble dumps (region, residues, every sampled trace with its domains and their per-domain null2
vectors) and /tmp/ens_hits.json with our scores and HMMER's printed E-values; both are written by a temporarily
instrumented build (the dump code is not kept in the product; see the git history of this file's commit message).
`evaluate(make_n2sc, post=...)` recomputes the per-residue null2 scores of every clustered region under an alternative
accumulation rule, propagates the change into sequence and domain scores, and compares them with the scores implied by
HMMER's E-values (precision 0.01-0.04 bit).  Round-1 findings are summarised in DESIGN.md."""
import json, math, sys
import numpy as np
OMEGA = 1.0 / 256
hits = json.load(open("/tmp/ens_hits.json"))

def load(f):
    regs = []
    cur = None
    for line in open(f):
        p = line.split()
        if p[0] == "REGION":
            cur = dict(i=int(p[1]), j=int(p[2]), L=int(p[3]), traces=[]); regs.append(cur)
        elif p[0] == "SEQ":
            cur["seq"] = np.array(p[1:], dtype=int)
        elif p[0] == "T":
            st = [tuple(int(v) for v in x.split(":")) for x in p[4:]]
            cur["traces"].append(dict(ndom=int(p[2]), states=st, doms=[]))
        elif p[0] == "D":
            cur["traces"][-1]["doms"].append(dict(sqfrom=int(p[1]), sqto=int(p[2]), hmmfrom=int(p[3]), hmmto=int(p[4]), null2=np.array(p[5:], dtype=float)))
        elif p[0] == "N2SC":
            cur["n2sc"] = np.array(p[1:], dtype=float)
    return regs

def n2sc_rule(reg, rule):
    Lr = reg["j"] - reg["i"] + 1
    acc = np.zeros(Lr + 2)
    seq = reg["seq"]
    for tr in reg["traces"]:
        rule(reg, tr, acc, seq, Lr)
    return np.log(acc[1:Lr + 1] / len(reg["traces"]))

def rule_ours(reg, tr, acc, seq, Lr):
    pos = 1
    for d in tr["doms"]:
        while pos <= d["sqfrom"]: acc[pos] += 1.0; pos += 1
        while pos <= d["sqto"]: acc[pos] += d["null2"][seq[pos - 1]]; pos += 1
    while pos <= Lr: acc[pos] += 1.0; pos += 1

def bias_from_S(S): return math.log2(1.0 + OMEGA * math.exp(S))
def S_from_bias(b): return math.log((2.0 ** b - 1.0) / OMEGA) if b > 1e-9 else -50.0

def evaluate(make_n2sc, verbose=True, post=None):
    """make_n2sc(reg, hitrec) -> n2sc array over region; post(hitrec, reg, n2) may modify. Returns total badness."""
    tot = 0.0
    for key, h in hits.items():
        regs = load(h["file"])
        dS_seq = 0.0
        dom_d = {tuple(d["env"]): 0.0 for d in h["doms"]}
        for reg in regs:
            ours = reg["n2sc"]
            new = make_n2sc(reg, h)
            if post: new = post(h, reg, new)
            dS_seq += float(new.sum() - ours.sum())
            for d in h["doms"]:
                a, b = d["env"]
                lo, hi = max(a, reg["i"]), min(b, reg["j"])
                if lo <= hi:
                    dom_d[(a, b)] += float(new[lo - reg["i"]:hi - reg["i"] + 1].sum() - ours[lo - reg["i"]:hi - reg["i"] + 1].sum())
        S0 = S_from_bias(h["bias"])
        nb = bias_from_S(S0 + dS_seq)
        nscore = h["pre"] - nb
        # golden score inferred from the E-value
        Eg = float(h["g_E"])
        gs = h["score"] - math.log(Eg / h["evalue"]) / h["flambda"]
        mant = float(f"{Eg:.1e}".split("e")[0]); prec = 0.05 / mant / h["flambda"]
        z = (gs - nscore) / max(prec, 0.005)
        line = f"{h['name'][-12:]:12s} seq: golden-new {gs - nscore:+.3f}+-{prec:.3f}"
        tot += min(z * z, 100)
        for d in h["doms"]:
            if d["g"] is None: continue
            Sd0 = d["corr"]
            nbd = bias_from_S(Sd0 + dom_d[tuple(d["env"])])
            nds = d["score"] + d["bias"] - nbd
            Eg = float(d["g"][1])
            if Eg <= 0: continue
            gds = d["score"] - math.log(Eg / d["ievalue"]) / h["flambda"]
            mant = float(f"{Eg:.1e}".split("e")[0]); precd = 0.05 / mant / h["flambda"]
            zd = (gds - nds) / max(precd, 0.005)
            tot += min(zd * zd, 100)
            line += f" | {d['env'][0]}-{d['env'][1]} {gds - nds:+.3f}+-{precd:.3f}"
        if verbose: print(line)
    if verbose: print("chi2-like:", round(tot, 1))
    return tot

if __name__ == "__main__":
    evaluate(lambda reg, h: n2sc_rule(reg, rule_ours))
