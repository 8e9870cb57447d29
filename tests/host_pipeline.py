"""CPU harness for the product's HOST half (domain definition + hit list), test infrastructure only.

The device half of the pipeline (filters, Forward/Backward parsers) is replaced here by the oracle: its cascade
selects the Forward survivors and its parsers provide the special-state rows, which are then handed to the
product's exported host entry point ``p7x_postprocess_targets``.  This lets ``-m "not gpu"`` runs check
p7_domaindef / p7_tophits logic against the golden tables without a GPU.
"""
import ctypes as C

import numpy as np

from pyhmmer_amd import _lib, plan7


def host_search(oracle, hmm, block, pipeline=None, F=(0.02, 1e-3, 1e-5), perturb_fwd=None):
    """The oracle's filter cascade (thresholds F: what the first stage applies) followed by the product's host stage
    (thresholds of <pipeline>).  perturb_fwd: {target: Forward score handed to the host stage instead of the oracle's}."""
    pipeline = pipeline or plan7.Pipeline(hmm.alphabet)
    bg = pipeline.background
    op = oracle.OracleProfile(hmm, bg, 400)
    pk = block.packed()
    recs, ctr = op.cascade_block(pk, F1=F[0], F2=F[1], F3=F[2], do_bias=pipeline.bias_filter)
    surv = [t for t in range(len(block)) if recs[t].stage == 4]
    fwdsc = np.array([(perturb_fwd or {}).get(t, recs[t].fwdsc) for t in surv], dtype=np.float32)
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
    if getattr(pipeline, "_mode", 0) == plan7._P7X_SCAN_MODELS:
        # a scan's per-model results carry the last filter every target passed (the device path records it); here it is
        # the oracle's, attached through the test seam
        stages = np.array([recs[t].stage for t in range(n)], dtype=np.uint8)
        assert _lib.lib().p7x_debug_tophits_set_stages(out, stages.ctypes.data, n) == 0
    return hits


DNA_COMP = np.array([3, 2, 1, 0, 4, 6, 5, 8, 7, 9, 10, 14, 13, 12, 11, 15, 16, 17], dtype=np.uint8)


def blocks_of(L, W, Cov):
    """(start i, residues n) of the blocks LongTargetsPipeline._search_loop_longtargets cuts a target into (plan7.pyx:7604-7610)."""
    out = []
    i = 0
    while i < L:
        c = 0 if i == 0 else min(Cov, L - i)
        w = min(W, L - i - c)
        n = c + w
        if n <= 0:
            break
        out.append((i, n))
        i += W - Cov
    return out


def host_nhmmer(oracle, hmm, sequences, pipeline=None, nparts=1):
    """CPU harness of the long-target path: the oracle's sequential SSV scan seeds the windows of every (target, block,
    strand); the product's host tail (p7x_longtarget_from_seeds) does the rest.  nparts > 1: the units are dealt over
    that many parts as a multi-device search deals them (cfg.lt_part / lt_nparts: consecutive units per part, each part
    sees only the seeds of its own units) and the parts are finished together by p7x_tophits_merge_longtargets."""
    pipeline = pipeline or plan7.LongTargetsPipeline(hmm.alphabet)
    bg = pipeline.background
    op = oracle.OracleProfile(hmm, bg, 400)
    om = plan7.OptimizedProfile(hmm, bg, 400)
    max_length = pipeline.window_length or hmm.max_length
    dsq, offsets, lengths, names, accs, descs = plan7.LongTargetsPipeline._pack(sequences)
    units = []                                    # (target, block start, strand, seeds) in the order of the reference's loop
    for t, s in enumerate(sequences):
        seq = np.asarray(s.sequence, dtype=np.uint8)
        for (i, n) in blocks_of(len(seq), pipeline.block_length, max_length):
            for strand in (0, 1):
                if (strand == 0 and pipeline.strand == "crick") or (strand == 1 and pipeline.strand == "watson"):
                    continue
                blk = seq[i:i + n] if strand == 0 else DNA_COMP[seq[i:i + n][::-1]]
                units.append((t, i, strand, list(oracle.ssv_longtarget(op, blk, max_length, pipeline.F1))))
    handles, nseeds = [], 0
    for part in range(nparts):
        cfg = pipeline._cfg()
        cfg.lt_part, cfg.lt_nparts = part, nparts
        st_t, st_b, st_s, seeds = [], [], [], []
        for u, (t, i, strand, sds) in enumerate(units):
            if nparts > 1 and (u * nparts) // len(units) != part:
                continue
            for sd in sds:
                st_t.append(t); st_b.append(i); st_s.append(strand); seeds.append(sd)
        ns = len(seeds)
        nseeds += ns
        a_t = np.array(st_t or [0], dtype=np.int64); a_b = np.array(st_b or [0], dtype=np.int64); a_s = np.array(st_s or [0], dtype=np.int32)
        a_seeds = np.array(seeds if seeds else [[0, 0, 0]], dtype=np.int64).reshape(-1, 3)
        out = C.c_void_p()
        st = _lib.lib().p7x_longtarget_from_seeds(C.byref(cfg), om._handle, dsq.ctypes.data, offsets.ctypes.data, lengths.ctypes.data,
                                                  len(sequences), names, accs, descs, a_t.ctypes.data, a_b.ctypes.data, a_s.ctypes.data,
                                                  np.ascontiguousarray(a_seeds).ctypes.data, ns, C.byref(out))
        if st != 0:
            raise RuntimeError(f"p7x_longtarget_from_seeds failed: {st} {_lib.last_error()}")
        handles.append(out)
    if nparts > 1:
        arr = (C.c_void_p * nparts)(*[h.value for h in handles])
        out = C.c_void_p()
        st = _lib.lib().p7x_tophits_merge_longtargets(arr, nparts, C.byref(out))
        if st != 0:
            raise RuntimeError(f"p7x_tophits_merge_longtargets failed: {st} {_lib.last_error()}")
    hits = plan7.TopHits(hmm, out)
    hits._keep = (om, names, accs, descs, dsq)
    hits._nseeds = nseeds
    hits._nunits = len(units)
    return hits
