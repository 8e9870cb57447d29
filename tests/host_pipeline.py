"""CPU harness for the product's HOST half (domain definition + hit list), test infrastructure only.

The device half of the pipeline (filters, Forward/Backward parsers) is replaced here by the oracle: its cascade
selects the Forward survivors and its parsers provide the special-state rows, which are then handed to the
product's exported host entry point ``p7x_postprocess_targets``.  This lets ``-m "not gpu"`` runs check
p7_domaindef / p7_tophits logic against the golden tables without a GPU.
"""
import ctypes as C

import numpy as np

from pyhmmer_amd import _lib, plan7


def host_search(oracle, hmm, block, pipeline=None, F=(0.02, 1e-3, 1e-5)):
    pipeline = pipeline or plan7.Pipeline(hmm.alphabet)
    bg = pipeline.background
    op = oracle.OracleProfile(hmm, bg, 400)
    pk = block.packed()
    recs, ctr = op.cascade_block(pk, F1=F[0], F2=F[1], F3=F[2], do_bias=pipeline.bias_filter)
    surv = [t for t in range(len(block)) if recs[t].stage == 4]
    fwdsc = np.array([recs[t].fwdsc for t in surv], dtype=np.float32)
    fx, bx, off = [], [], []
    pos = 0
    for t in surv:
        st, bsc, f, b = op.bck(block[t].sequence)
        fx.append(f.reshape(-1)); bx.append(b.reshape(-1)); off.append(pos)
        pos += f.size
    fxa = np.concatenate(fx).astype(np.float32) if fx else np.zeros(1, np.float32)
    bxa = np.concatenate(bx).astype(np.float32) if bx else np.zeros(1, np.float32)
    offa = np.array(off if off else [0], dtype=np.int64)
    surva = np.array(surv if surv else [0], dtype=np.int32)
    counts = (C.c_uint64 * 4)(ctr.n_past_msv, ctr.n_past_bias, ctr.n_past_vit, ctr.n_past_fwd)
    om = plan7.OptimizedProfile(hmm, bg, 400)
    n = len(block)
    names = (C.c_char_p * max(n, 1))(*[s.name.encode() for s in block])
    accs = (C.c_char_p * max(n, 1))(*[(s.accession or "").encode() for s in block])
    descs = (C.c_char_p * max(n, 1))(*[(s.description or "").encode() for s in block])
    cfg = pipeline._cfg()
    out = C.c_void_p()
    st = _lib.lib().p7x_postprocess_targets(C.byref(cfg), om._handle, pk.dsq.ctypes.data, pk.offsets.ctypes.data,
                                            pk.lengths.ctypes.data, n, surva.ctypes.data, len(surv), fwdsc.ctypes.data,
                                            fxa.ctypes.data, bxa.ctypes.data, offa.ctypes.data, counts, names, accs, descs,
                                            C.byref(out))
    if st != 0:
        raise RuntimeError(f"p7x_postprocess_targets failed: {st} {_lib.last_error()}")
    hits = plan7.TopHits(hmm, out)
    hits._keep = (om, names, accs, descs)
    return hits
