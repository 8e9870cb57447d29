"""numpy emulation of the algorithm of p7x_vitpk.hip (test infrastructure): T lanes x 2 halves = 2T stripes of P nodes,
node k = s*P + q + 1, stripe shift at register 0, lazy-F closure to convergence.  tests/test_oracle_golden.py checks it
against the oracle's xC; it fixed the layout and the closure of the kernel before any device time was spent."""
import numpy as np
import oracle_lib

NEG = -32768
def sat(a): return np.clip(a, -32768, 32767)

def unstripe(op):
    p = op.p; M, Q = p.M, p.Q8
    twv, rwv = op.arr("twv"), op.arr("rwv")
    tw = np.full((8, M + 2), NEG, np.int64); rw = np.full((p.Kp, M + 2), NEG, np.int64)
    for k in range(1, M + 1):
        q, z = (k - 1) % Q, (k - 1) // Q
        for t in range(7): tw[t][k] = twv[7 * q + t][z]
        tw[7][k] = twv[7 * Q + q][z]
        rw[:, k] = rwv[:, q * 8 + z]
    return tw, rw

def vit_striped(op, seq, T, P, stats=None):
    p = op.p; M = p.M; L = len(seq)
    oracle_lib.lib().p7o_reconfig_length(op.ptr, L)
    tw, rw = unstripe(op)
    S = 2 * T
    assert S * P >= M
    node = np.array([[s * P + q + 1 for s in range(S)] for q in range(P)])        # [P][S]
    ok = node <= M
    def tab(t): return np.where(ok, tw[t][np.minimum(node, M + 1)], NEG)
    BM, MM, IM, DM, MD, MI, II, DD = [tab(t) for t in range(8)]
    base, xwe, xwm, ddb = int(p.base_w), int(p.xw[0][0]), int(p.xw[1][0]), int(p.ddbound_w)   # xw[E][MOVE], xw[N][MOVE]
    Mr = np.full((P, S), NEG, np.int64); Ir = Mr.copy(); Dr = Mr.copy()
    xN, xJ, xC = base, NEG, NEG
    xB = xN + xwm
    def shift(v):                       # value of the previous stripe, -inf into stripe 0
        o = np.empty_like(v); o[0] = NEG; o[1:] = v[:-1]; return o
    for i in range(L):
        x = int(seq[i])
        em = np.where(ok, rw[x][np.minimum(node, M + 1)], NEG)
        Mn = np.empty_like(Mr); In = np.empty_like(Mr); Dn = np.full_like(Mr, NEG)
        mp, ip, dp = shift(Mr[P - 1]), shift(Ir[P - 1]), shift(Dr[P - 1])
        dcv = None
        for q in range(P):
            sv = sat(xB + BM[q])
            sv = np.maximum(sv, sat(mp + MM[q])); sv = np.maximum(sv, sat(ip + IM[q])); sv = np.maximum(sv, sat(dp + DM[q]))
            sv = sat(sv + em[q])
            Mn[q] = sv
            dcv = sat(sv + MD[q])
            if q + 1 < P: Dn[q + 1] = dcv
            In[q] = np.maximum(sat(Mr[q] + MI[q]), sat(Ir[q] + II[q]))
            mp, ip, dp = Mr[q], Ir[q], Dr[q]
        Dn[0] = shift(dcv)
        xE = int(Mn.max())
        dmax = int(max(Dn.max(), dcv.max()))        # every M->D candidate, as dmaxv in the kernel
        if xE >= 32767: return 32767
        xC = max(xC, xE + xwe); xJ = max(xJ, xE + xwe); xB = max(xJ + xwm, xN + xwm)
        if dmax + ddb > xB:             # lazy F: closure to convergence
            npass = 0
            while True:
                npass += 1
                for q in range(1, P):
                    Dn[q] = np.maximum(Dn[q], sat(Dn[q - 1] + DD[q - 1]))
                c = shift(sat(Dn[P - 1] + DD[P - 1]))
                if not (c > Dn[0]).any(): break
                Dn[0] = np.maximum(Dn[0], c)
            if stats is not None: stats.append(npass)
        Mr, Ir, Dr = Mn, In, Dn
    return xC

