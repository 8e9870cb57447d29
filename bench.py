#!/usr/bin/env python
"""bench.py -- hmmsearch throughput of the MI355X-native p7_Pipeline path.

The line's workload is the one BASELINE.json's metric is quoted on ("hmmsearch, Pfam-A vs proteome": configs[3], SURVEY.md 8d
"config 4"): a Pfam-shaped library of 20,000 calibrated profiles (bench_workloads.py; Pfam-A itself is not available offline)
against 500,000 Swiss-Prot-shaped synthetic targets, through hmmer.hmmsearch with the library's defaults: the whole pipeline
(MSV -> bias -> Viterbi -> Forward -> Backward on the device, domain definition + hit lists on the host), targets sharded by
residues over the GPUs, per-query TopHits merged on rank 0 inside the timed region.  A "step" is --pfam-profiles-per-step
(1,000) consecutive library profiles searched against the resident target database; with the driver's --steps 20 the timed
region is the whole library once.  Profiles' device images and the targets are resident in HBM before the clock starts.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W            # one rank per GPU, strong scaling (the same 500,000 targets sharded)

Rank 0 prints ONE JSON line.  `value` is whole-job GCUPS: sum over the profiles searched of M x residues of all targets,
divided by the max-over-ranks wall time (all-pairs M*L denominator, as the HMMER literature does).  The other BASELINE configs
are fields of the same line: `config1` (one profile x 1,000,000 targets: throughput of a stream of queries, and the latency of
ONE query on an idle device), `scan` (hmmscan orientation, 4,000-protein block), `nhmmer` (250 Mbp chromosome).
`--workload config1 | scan | nhmmer` prints that workload as the line instead.
"""
import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

import numpy as np

# before torch initialises HIP (see pyhmmer_amd/__init__.py).  A rehearsal with many ranks on ONE device (P7X_BENCH_SHARE_DEVICE)
# must not ask for 8 hardware queues per process: 8 ranks x 8 queues crashed the runtime on the shared device.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2" if (os.environ.get("P7X_BENCH_SHARE_DEVICE") == "1" and int(os.environ.get("WORLD_SIZE", "1")) > 4) else "8")

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))     # oracle_lib, for the cpu_baseline legs only (after the timed regions)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_LANE_OPS_PER_S = 256 * 4 * 16 * 2.4e9    # 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz: packed-16 VOP3P ops take 4 cycles per
                                              # wave64 (PMC: SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU quad-cycles, profiles/r01_bench_pmc.md)
MSV_OPS_PER_CELL = 0.75          # fast MSV kernel since round 5: two v_pk_add_f16 clamp + ONE v_pk_maximum3_f16 per two registers,
                                 # i.e. three packed instructions per four cells (p7x_msv.hip; 1.0 with the int16 flavour of rounds 1-4)
VALU_SUSTAINED_GCUPS = 49500.0   # the kernel's three packed ops per four cells and nothing else, held for 20-100 ms with every SIMD
                                 # busy in the MSV launch shape (scripts/ubench/sustained, profiles/r05_sustained_ubench.md): the
                                 # packed-op roof in wall-clock terms (2.27 GHz-equivalent; 35,600 with the int16 flavour's four ops)
MSV_SUSTAINED_MIX_GCUPS = 41000.0     # ... with the kernel's LDS reads (one conflict-free ds_read_b64 per three packed ops): same file
F32_LANE_OPS_PER_S = 256 * 4 * 32 * 2.4e9     # two-source f32 VOP2 ops issue in 2 cycles per wave64 (profiles/r03_pk_issue_fma.md)
# Static instruction counts per DP row of the kernels behind the headline's tail, from the ISA of the instantiations that
# serve M = 262 (KR: 5 nodes per lane; scripts/isa_inner_loops.py): wave64 instructions per row of one
# target (packed Viterbi: per row of a wavefront's 8 targets).
TAIL_KERNELS_M262 = {
    "vitpk_kernel<8, 17>": {"stage": "viterbi", "instr_per_row": 483.0, "targets_per_wave": 8, "class": "packed int16 (16 lanes/clk/SIMD)"},
    "fwd_kernel<5>": {"stage": "forward", "instr_per_row": 330.0, "targets_per_wave": 1, "class": "f32 (32 lanes/clk/SIMD at best)"},
    "fwd_kernel<5> (rows kept) + bck_kernel<5> + regions_kernel": {"stage": "fwd_rows", "instr_per_row": 330.0 + 260.0, "targets_per_wave": 1,
                                                                   "class": "f32 (32 lanes/clk/SIMD at best)"},
}


def tail_kernels(hmm, args, sc, solo, cells_rank):
    """roofline.kernels: the stages of one query ALONE on the device (stand-alone HIP-event times of the library's own streams,
    measured after the timed region) against the issue roof of their instruction class: cells actually run x wave64
    instructions per cell x 64 lanes, over lanes the SIMDs can issue per second."""
    if not solo:
        return None
    L = float(args.seqlen)
    out = [{"kernel": "p7x::msv_fast_kernel<R, 1, half>", "stage": "msv", "standalone_ms": round(solo["msv_kernel"], 4),
            "cells": int(cells_rank), "lane_ops_per_cell": MSV_OPS_PER_CELL, "class": "packed half (16 lanes/clk/SIMD)",
            "gcups": round(cells_rank / (solo["msv_kernel"] * 1e-3) / 1e9, 1),
            "frac": round(cells_rank / (solo["msv_kernel"] * 1e-3) * MSV_OPS_PER_CELL / VALU_LANE_OPS_PER_S, 4)}]
    if hmm.M != 262:
        return out           # the instruction counts below are those of the M = 262 instantiations
    counts = {"viterbi": sc["bias"], "forward": sc["vit"], "fwd_rows": sc["fwd"]}
    for name, k in TAIL_KERNELS_M262.items():
        ms = solo.get(k["stage"], 0.0)
        if ms <= 0.0:
            continue
        cells = float(counts[k["stage"]]) * L * hmm.M
        lane_ops_per_cell = k["instr_per_row"] * 64.0 / (k["targets_per_wave"] * hmm.M)
        peak = VALU_LANE_OPS_PER_S if k["class"].startswith("packed") else F32_LANE_OPS_PER_S
        out.append({"kernel": "p7x::" + name, "stage": k["stage"], "standalone_ms": round(ms, 4), "cells": int(cells),
                    "lane_ops_per_cell": round(lane_ops_per_cell, 1), "class": k["class"],
                    "gcups": round(cells / (ms * 1e-3) / 1e9, 1), "frac": round(cells / (ms * 1e-3) * lane_ops_per_cell / peak, 4)})
    # the envelope stage is timed on the host (first launch to last result: two envelope rounds and the ensemble kernels)
    if solo.get("envelopes", 0.0) > 0.0:
        cells = float(sc["fwd"]) * L * hmm.M          # an envelope is at most its target
        per_cell = (330.0 + 260.0 + 1124.0) * 64.0 / hmm.M
        out.append({"kernel": "p7x::env_kernel<5, false, false> (+ ens_forward / ens_walk, second envelope round)", "stage": "envelopes",
                    "standalone_ms": round(solo["envelopes"], 4), "cells_upper_bound": int(cells), "lane_ops_per_cell": round(per_cell, 1),
                    "class": "f32 (32 lanes/clk/SIMD at best)", "gcups": round(cells / (solo["envelopes"] * 1e-3) / 1e9, 1),
                    "frac": round(cells / (solo["envelopes"] * 1e-3) * per_cell / F32_LANE_OPS_PER_S, 4),
                    "note": "wall time of the stage on the host, not a kernel duration: a latency chain of small launches"})
    return out


def pipeline_valu_budget(hmm, args, sc, B, ms_per_batch):
    """roofline.pipeline: the WHOLE cascade of one device batch against VALU issue, from static instruction counts (the ISA of
    the M = 262 instantiations) x the units every stage actually processed: wave64 instructions per batch, the cycles they
    need at their class's best issue rate (packed 16-bit: 4 cycles, f32: 2), and that lower bound in ms on all 1,024 SIMDs
    at the nominal 2.4 GHz next to the measured time per batch.  MSV's count is the PMC's (SQ_INSTS_VALU per cell,
    profiles/r05_bench_pmc.md); an envelope is counted as long as its target (an upper bound)."""
    if hmm.M != 262:
        return None
    L = float(args.seqlen)
    rows = lambda n: float(n) * L * B
    instr = {
        "msv (packed)": 6.837e9 / 5.502e11 * float(args.nseq) * L * hmm.M * B,        # 0.795 lane-ops per cell / 64 lanes
        "viterbi (packed)": rows(sc["bias"]) / 8.0 * 483.0,
        "forward parser (f32)": rows(sc["vit"]) * 330.0,
        "forward rows + backward (f32)": rows(sc["fwd"]) * (330.0 + 260.0),
        "envelopes (f32)": rows(sc["fwd"]) * (330.0 + 260.0 + 1124.0),
    }
    cycles = sum(v * (4.0 if "packed" in k else 2.0) for k, v in instr.items())
    bound_ms = cycles / (256 * 4 * 2.4e9) * 1e3
    return {"wave_instructions_per_batch": {k: float(f"{v:.4g}") for k, v in instr.items()}, "total": float(f"{sum(instr.values()):.4g}"),
            "issue_cycles_lower_bound": float(f"{cycles:.4g}"), "issue_bound_ms_per_batch": round(bound_ms, 2),
            "measured_ms_per_batch": round(ms_per_batch, 2), "frac": round(bound_ms / ms_per_batch, 4),
            "note": "the pipeline as a whole is VALU-issue bound: every kernel of the cascade is arithmetic on registers, and the tail "
                    "kernels run in what the MSV launch of the next batch leaves; bias filter, decisions and ensembles (~5 %) not counted"}


def valu_roofline(cells, seconds, ops_per_cell, what):
    """A whole workload's cell rate against the packed-op issue roof of its scan kernel (upper bound: the later stages are
    not in the denominator)."""
    rate = cells / max(seconds, 1e-12)
    peak = VALU_LANE_OPS_PER_S / ops_per_cell
    return {"bound": "valu", "kernel": what, "achieved": round(rate / 1e9, 1), "peak": round(peak / 1e9, 1), "unit": "GCUPS",
            "frac": round(rate / peak, 4), "ops_per_cell": ops_per_cell,
            "note": "whole-job DP cells per second against the packed-16 VALU issue roof (256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz) of the "
                    "scan kernel alone; HBM is not the binding roof on this path (tables in LDS, ~1/M byte per cell)"}


def emit_from_model(hmm, rng, tabs):
    """Sample one full pass through the core model (match / insert / delete path, node 1 -> M)."""
    cmat, cins, ct = tabs
    out = []
    state, k, M = 0, 1, hmm.M          # 0 = M, 1 = I, 2 = D
    while k <= M:
        u = rng.random()
        if state == 0:
            out.append(int(np.searchsorted(cmat[k], rng.random())))
            if k == M:
                break
            nxt = 0 if u < ct[k, 0] else (1 if u < ct[k, 1] else 2)       # MM | MI | MD
            if nxt != 1:
                k += 1
            state = nxt
        elif state == 1:
            out.append(int(np.searchsorted(cins[k], rng.random())))
            if u < ct[k, 2]:                                               # IM | II
                state, k = 0, k + 1
        else:
            if k == M:
                break
            state = 0 if u < ct[k, 3] else 2                               # DM | DD
            k += 1
    return np.minimum(np.array(out, dtype=np.uint8), hmm.alphabet.K - 1)


def pipe_opts(depth, feeders, finishers):
    """Keyword arguments for hmmer.hmmsearch: only what a command line flag set; everything else is the library's default."""
    o = {}
    if depth is not None:
        o["pipeline_depth"] = depth
    if feeders is not None:
        o["feeders"] = feeders
    if finishers is not None:
        o["finishers"] = finishers
    return o


def pipe_effective(depth, feeders, finishers, many=False):
    """What the search runs with (the library's defaults where no flag was given), for the report.  many: batches of several
    different profiles (feeders = 0, the default, lets the first batch decide: hmmer.hmmsearch)."""
    import inspect
    from pyhmmer_amd import hmmer
    d = {k: v.default for k, v in inspect.signature(hmmer.hmmsearch).parameters.items() if k in ("pipeline_depth", "feeders", "finishers")}
    d.update(pipe_opts(depth, feeders, finishers))
    if not d["feeders"]:
        d["feeders"] = 3 if many else 2
    if not d["finishers"]:
        d["finishers"] = max(d["feeders"], d["pipeline_depth"])
    d["library_defaults"] = not pipe_opts(depth, feeders, finishers)
    return d


def make_workload(hmm, nseq, L, seed, planted_frac=0.001):
    """Flat arrays in the C-ABI's input format: 255 x1..xL 255 x1..xL 255 ..."""
    from pyhmmer_amd import plan7
    rng = np.random.default_rng(seed)
    K = hmm.alphabet.K
    bg = plan7.Background(hmm.alphabet).residue_frequencies.astype(np.float64)
    cum = np.cumsum(bg / bg.sum())
    cum[-1] = 1.0
    dsq = np.full((nseq, L + 1), 255, dtype=np.uint8)
    # inverse-CDF sampling through a 16-bit lookup table (background frequencies resolved to 1/65536)
    lut = np.minimum(np.searchsorted(cum, (np.arange(65536) + 0.5) / 65536.0, side="right"), K - 1).astype(np.uint8)
    for lo in range(0, nseq, 100_000):
        hi = min(nseq, lo + 100_000)
        dsq[lo:hi, :L] = lut[rng.integers(0, 65536, size=(hi - lo, L), dtype=np.uint16)]
    t = hmm.transition_probabilities.astype(np.float64)
    mat, ins = hmm.match_emissions.astype(np.float64), hmm.insert_emissions.astype(np.float64)
    cmat = np.cumsum(mat / np.maximum(mat.sum(axis=1, keepdims=True), 1e-30), axis=1)
    cins = np.cumsum(ins / np.maximum(ins.sum(axis=1, keepdims=True), 1e-30), axis=1)
    ct = np.zeros((hmm.M + 1, 4))
    s3 = np.maximum(t[:, 0:3].sum(axis=1), 1e-30)
    ct[:, 0], ct[:, 1] = t[:, 0] / s3, (t[:, 0] + t[:, 1]) / s3
    ct[:, 2] = t[:, 3] / np.maximum(t[:, 3] + t[:, 4], 1e-30)
    ct[:, 3] = t[:, 5] / np.maximum(t[:, 5] + t[:, 6], 1e-30)
    nplant = int(round(nseq * planted_frac))
    planted = rng.choice(nseq, size=nplant, replace=False) if nplant else np.zeros(0, dtype=np.int64)
    for tgt in planted:
        dom = emit_from_model(hmm, rng, (cmat, cins, ct))[: L - 20]
        start = int(rng.integers(0, L - len(dom) + 1))
        dsq[tgt, start:start + len(dom)] = dom
    flat = np.concatenate([np.array([255], dtype=np.uint8), dsq.reshape(-1)])
    offsets = 1 + np.arange(nseq, dtype=np.int64) * (L + 1)
    lengths = np.full(nseq, L, dtype=np.int32)
    return flat, offsets, lengths, np.sort(planted)


def msv_traffic_bytes(workload_key, algorithmic_bytes=None):
    """HBM bytes per launch of the dominant kernel from the PMC passes of the same command (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate runs, corrected as MI355X_MICROARCH.md prescribes), as summarised by
    scripts/rocprof_pmc_summary.py into profiles/msv_traffic.json.  None when no measurement of this workload is committed."""
    try:
        rec = json.load(open(ROOT / "profiles" / "msv_traffic.json"))
    except (OSError, ValueError):
        return None
    e = rec.get(workload_key, {})
    if "traffic_over_algorithmic" in e and algorithmic_bytes is not None:      # many-profile workload: measured over a run's launches as a ratio
        return int(e["traffic_over_algorithmic"] * algorithmic_bytes)
    return e.get("traffic_bytes_per_launch")


def host_cpus():
    """CPUs this job may use: affinity mask, then the cgroup quota (the GPU boxes are containers on many-core hosts)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def pyhmmer_probe():
    """"probe `import pyhmmer` on the GPU box at run time".  Only /root/repo travels there, so the import is
    expected to fail; the line's cpu_baseline then says so and times the in-repo restatement (kind "port")."""
    try:
        import pyhmmer  # noqa: F401
        return True
    except Exception:       # noqa: BLE001
        return False


def cpu_baseline(hmm, bg, flat, offsets, lengths, n, L_hint):
    """The whole search on the host cores, C threads: filter cascade + parsers through oracle/ (the SSE2 restatement
    of impl_sse; ctypes releases the GIL, one oracle profile per thread because the length model is configured per
    target), then domain definition + hit list for the Forward survivors through the product's host twin
    (p7x_postprocess_targets, the CPU test seam).  Reported next to `value`; never part of it."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    import oracle_lib
    from pyhmmer_amd import _lib, plan7
    cores = host_cpus()
    ops = [oracle_lib.OracleProfile(hmm, bg, L_hint) for _ in range(cores)]
    bounds = np.linspace(0, n, cores + 1).astype(np.int64)
    recs = (oracle_lib.Record * n)()

    class _Pk:            # the oracle's block interface (PackedBlock duck type)
        pass

    def work(c):
        lo, hi = int(bounds[c]), int(bounds[c + 1])
        ctr = oracle_lib.Counters()
        if hi > lo:
            sub = (oracle_lib.Record * (hi - lo)).from_buffer(recs, lo * C.sizeof(oracle_lib.Record))
            oracle_lib.lib().p7o_cascade_block(ops[c].ptr, flat.ctypes.data, offsets[lo:hi].ctypes.data, lengths[lo:hi].ctypes.data,
                                               hi - lo, 0.02, 1e-3, 1e-5, 1, sub, C.byref(ctr))
        return ctr

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        ctrs = list(ex.map(work, range(cores)))
    t_filters = time.perf_counter() - t0
    stage = np.frombuffer(recs, dtype=np.dtype([("f", "f4", 5), ("P", "f8", 4), ("i", "i4", 4)], align=True))["i"][:, 2]   # Record.stage
    surv = np.nonzero(stage == 4)[0].astype(np.int32)
    fwdsc = np.array([recs[int(t)].fwdsc for t in surv], dtype=np.float32)

    def rows(t):
        _, _, f, b = ops[0].bck(flat[offsets[t]: offsets[t] + lengths[t]])
        return f.reshape(-1), b.reshape(-1)

    t0 = time.perf_counter()
    fx, bx, off, pos = [], [], [], 0
    for t in surv:          # parser rows of the survivors (Backward), then the host twin of domain definition
        f, b = rows(int(t))
        fx.append(f); bx.append(b); off.append(pos); pos += f.size
    fxa = np.concatenate(fx).astype(np.float32) if fx else np.zeros(1, np.float32)
    bxa = np.concatenate(bx).astype(np.float32) if bx else np.zeros(1, np.float32)
    offa = np.array(off if off else [0], dtype=np.int64)
    counts = (C.c_uint64 * 4)(sum(c.n_past_msv for c in ctrs), sum(c.n_past_bias for c in ctrs),
                              sum(c.n_past_vit for c in ctrs), sum(c.n_past_fwd for c in ctrs))
    pli = plan7.Pipeline(hmm.alphabet, host_threads=cores)
    om = plan7.OptimizedProfile(hmm, bg, L_hint)
    cfg = pli._cfg()
    outp = C.c_void_p()
    st = _lib.lib().p7x_postprocess_targets(C.byref(cfg), om._handle, flat.ctypes.data, offsets.ctypes.data, lengths.ctypes.data, n,
                                            surv.ctypes.data if len(surv) else offa.ctypes.data, len(surv), fwdsc.ctypes.data,
                                            fxa.ctypes.data, bxa.ctypes.data, offa.ctypes.data, counts, None, None, None, C.byref(outp))
    t_dd = time.perf_counter() - t0
    nhits = len(plan7.TopHits(hmm, outp)) if st == 0 else -1
    dt = t_filters + t_dd
    # The same survivors once more through the oracle's OWN domain definition and sequence scoring (oracle/p7_oracle_dd.c:
    # no product code past the filters), not timed: how many targets it reports at the pipeline's thresholds (E <= 10 over
    # these n targets).  The host twin's count above and this one agree unless a target sits on the threshold.
    oracle_hits = None
    try:
        oracle_hits = 0
        for t in surv:
            envs, cnt, sq = oracle_lib.domains(ops[0], flat[offsets[t]: offsets[t] + lengths[t]], seed=42, want_sequence=True)
            if sq["ndom"] > 0 and math.exp(sq["lnP"]) * n <= 10.0:
                oracle_hits += 1
    except Exception:                                   # the check must never cost the bench its line
        oracle_hits = None
    return {
        "value": round(float(hmm.M) * float(lengths[:n].sum()) / dt / 1e9, 3), "unit": "GCUPS", "cores": cores, "kind": "port",
        "sample": f"the first {n} targets of the same workload, whole search: oracle/ filter cascade + parsers (SSE2 restatement of "
                  f"impl_sse) on {cores} threads {t_filters:.2f} s, Backward rows + domain definition / hit list (product host twin) "
                  f"{t_dd:.2f} s; pyhmmer itself is {'importable here but not used (kind stays port)' if pyhmmer_probe() else 'not installed on this box (import pyhmmer fails: only the repo travels)'}",
        "pyhmmer_importable": pyhmmer_probe(),
        "past_msv": int(counts[0]), "past_fwd": int(counts[3]), "hits": nhits, "hits_by_the_oracles_own_domain_definition": oracle_hits,
    }


def shard_bounds(lengths, world):
    """Residue-balanced contiguous target shards (hmmer.make_chunks / _hmmsearch.py:153-171 on the packed arrays)."""
    csum = np.cumsum(lengths.astype(np.int64))
    total = int(csum[-1])
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(csum, total * r / world, side="left")) + 1)
    cuts.append(len(lengths))
    return cuts


def cpu_baseline_many(hmms, bg, flat, offsets, lengths, profile_idx, ntargets):
    """cpu_baseline() (the whole search through oracle/ + the product's host twin, all usable cores) for a SAMPLE of a
    many-profile workload: the profiles `profile_idx` against the first `ntargets` targets, one after the other."""
    t = cells = 0.0
    hits = 0
    cores = 0
    for e in profile_idx:
        r = cpu_baseline(hmms[e], bg, flat, offsets, lengths, ntargets, 400)
        c = float(hmms[e].M) * float(lengths[:ntargets].sum())
        t += c / (r["value"] * 1e9); cells += c; hits += max(0, r["hits"]); cores = r["cores"]
    return {"value": round(cells / t / 1e9, 3), "unit": "GCUPS", "cores": cores, "kind": "port",
            "sample": f"{len(profile_idx)} profiles (every {profile_idx[1] - profile_idx[0] if len(profile_idx) > 1 else 1}th of the {len(hmms)} searched) "
                      f"x the first {ntargets} targets of the same workload, whole search per profile (oracle/ filter cascade + parsers, "
                      f"product host twin for domain definition), {t:.2f} s of wall time", "hits": hits}


def build_library(args, local_rank):
    import bench_workloads as bw
    from pyhmmer_amd import plan7
    t0 = time.perf_counter()
    hmms, lib_lengths, templates = bw.make_library(args.pfam_library, device=local_rank, count=args.pfam_profiles)
    bg = plan7.Background(hmms[0].alphabet)
    oms = [plan7.OptimizedProfile(h, bg, 400) for h in hmms]
    return {"hmms": hmms, "lengths": lib_lengths, "templates": templates, "bg": bg, "oms": oms, "seconds": time.perf_counter() - t0}


def run_pfam(args, rank, world, local_rank, dist, red_dev, torch, host_threads, lib):
    """the synthetic 20k-profile library against `--pfam-targets` Swiss-Prot-shaped targets, STRONG
    scaling: the targets are sharded by residues over the ranks, every rank searches every profile of a step against its shard,
    rank 0 gathers the per-rank hit lists and merges them per query (one native call, p7x_tophits_merge_many) -- all inside
    the timed region.  Profiles (device images) and targets are resident in HBM before the clock starts, like a pressed
    database and a loaded proteome.  Step s searches profiles [s * pps, (s + 1) * pps) of the library (cyclically); the
    `--warmup` steps are the ones before step 0, the timed region is ONE hmmer.hmmsearch call over the `--steps` steps'
    profiles, bracketed by barrier + synchronize."""
    import bench_workloads as bw
    from pyhmmer_amd import hmmer, plan7
    hmms, lib_lengths, templates, bg, oms = lib["hmms"], lib["lengths"], lib["templates"], lib["bg"], lib["oms"]
    t0 = time.perf_counter()
    frac = min(0.5, 12.5 * len(hmms) / args.pfam_targets)
    flat, offsets, lengths, nplanted = bw.make_targets(args.pfam_targets, len(hmms), templates, lib_lengths, planted_frac=frac)
    cuts = shard_bounds(lengths, world)
    lo, hi = cuts[rank], cuts[rank + 1]
    db = plan7.SequenceDatabase.from_packed(hmms[0].alphabet, flat, offsets[lo:hi], lengths[lo:hi], device=local_rank)
    t_tgt = time.perf_counter() - t0

    fopts = {k: v for k, v in (("F1", args.F1), ("F2", args.F2), ("F3", args.F3)) if v is not None}     # diagnostics: where the time goes

    def search(qs):
        return list(hmmer.hmmsearch(qs, db, cpus=host_threads, batch=args.pfam_batch, **pipe_opts(args.pfam_depth, args.feeders, args.pfam_finishers), **fopts))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    pps = max(1, min(args.pfam_profiles_per_step, len(oms)))
    step_profiles = lambda first, n: [(first * pps + i) % len(oms) for i in range(n * pps)]
    timed_idx = step_profiles(0, args.steps)
    t0 = time.perf_counter()
    search(oms)                     # untimed: device images of every profile become resident; pools, clocks and workers settle
    t_resident = time.perf_counter() - t0
    if args.warmup > 0:
        search([oms[e] for e in step_profiles(-args.warmup, args.warmup)])
    barrier()
    t0 = time.perf_counter()
    hits = search([oms[e] for e in timed_idx])
    t_search = time.perf_counter() - t0
    merged = hits
    t_ser = t_gather = t_merge = 0.0
    if dist is not None:
        t1 = time.perf_counter()
        mine = [h.to_bytes() for h in hits]
        t_ser = time.perf_counter() - t1
        t1 = time.perf_counter()
        blobs = [None] * world if rank == 0 else None
        dist.gather_object(mine, blobs, dst=0)
        t_gather = time.perf_counter() - t1
        if rank == 0:
            t1 = time.perf_counter()
            merged = plan7.TopHits.merge_many(blobs, threads=host_threads)
            t_merge = time.perf_counter() - t1
    barrier()
    elapsed = time.perf_counter() - t0
    t_max = elapsed
    if dist is not None:
        b = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(b, op=dist.ReduceOp.MAX)
        t_max = float(b.item())
    if rank != 0:
        return None
    nodes = float(sum(hmms[e].M for e in timed_idx))
    nprof = len(timed_idx)
    residues = float(lengths.sum())
    nhits = sum(len(h) for h in merged)
    nrep = sum(len(h.reported) for h in merged)
    sc = {k: sum(h.stage_counts[k] for h in hits) for k in ("msv", "bias", "vit", "fwd")}
    out = {
        "workload": f"configs[3]: hmmsearch of a Pfam-shaped library ({len(hmms)} calibrated profiles: 14 fixture models resampled to "
                    f"M ~ lognormal(median 120) in [20, 2000], calibrated on the device; Pfam-A itself is not available offline) vs "
                    f"{args.pfam_targets} Swiss-Prot-shaped synthetic targets (L ~ lognormal(5.65, 0.65) in [30, 5000], {nplanted} with a "
                    f"planted domain), sharded by residues over {world} GPU(s), hmmer.hmmsearch defaults, per-query TopHits gathered and "
                    f"merged on rank 0 (p7x_tophits_merge_many) inside the timed region; a step = {pps} consecutive library profiles "
                    f"x all targets, {args.steps} steps = {nprof} profiles",
        "value": round(nodes * residues / t_max / 1e9, 2), "unit": "GCUPS", "scaling": "strong",
        "profiles": nprof, "profiles_per_step": pps, "library": len(hmms), "targets": int(args.pfam_targets),
        "mean_M": round(nodes / nprof, 1), "mean_L": round(residues / len(lengths), 1),
        "seconds": round(t_max, 4), "ms_per_profile": round(1e3 * t_max / nprof, 4), "profiles_per_s": round(nprof / t_max, 1),
        "seqs_per_s": round(float(nprof) * len(lengths) / t_max, 1),
        "untimed_residency_pass_seconds": round(t_resident, 2),
        "search_seconds_rank0": round(t_search, 4),
        "merge_seconds_rank0": {"serialise": round(t_ser, 4), "gather": round(t_gather, 4), "merge_many": round(t_merge, 4)},
        "batch": args.pfam_batch, **pipe_effective(args.pfam_depth, args.feeders, args.pfam_finishers, many=True),
        "hits": nhits, "reported": nrep, "stage_counts_rank0": sc,
        "guards_rank0": {k: sum(h.guard_counts[k] for h in hits) for k in ("f3_dropped", "oa_redone", "ens_device", "ens_redone")},
        # per-BATCH times (every query of a batch reports its batch's): device stages by HIP events of the first class
        "batch_ms_mean_rank0": {k: round(sum(h.timings_ms[k] for h in hits) / len(hits), 3) for k in hits[0].timings_ms},
        "setup_seconds": {"library": round(lib["seconds"], 2), "targets": round(t_tgt, 2)},
    }
    # ---- roofline of the dominant kernel: the fast MSV launch (one launch per tier of register tiles and batch, p7x_msv.hip
    # msv_tier_kernel<T>; a batch's LARGEST launch is timed by HIP events on the stream it runs on and reported by every
    # query of the batch with the lanes (queries) and nodes it covered).  Algorithmic bytes per launch: every
    # lane streams this rank's residues (L + 2 bytes per comparison incl. the sentinels) and writes 16 B per comparison,
    # plus its MSV table once (29 x 16 x Q16 bytes).
    shard_res, shard_n = float(lengths[lo:hi].sum()), float(hi - lo)
    w = np.array([1.0 / max(1.0, h.timings_ms["batch_queries"]) for h in hits])
    t_l = np.array([h.timings_ms["msv_kernel"] for h in hits])
    lanes = np.array([h.timings_ms["msv_launch_lanes"] for h in hits])
    lnodes = np.array([h.timings_ms["msv_launch_nodes"] for h in hits])
    nlaunch = float(w.sum())
    sum_ms = float((w * t_l).sum())
    sum_bytes = float((w * (lanes * (shard_res + 2.0 * shard_n + 16.0 * shard_n) + lnodes / 16.0 * 29 * 16)).sum())
    sum_cells = float((w * lnodes).sum()) * shard_res
    ach = sum_bytes / max(sum_ms * 1e-3, 1e-12) / 1e9
    kcups = sum_cells / max(sum_ms * 1e-3, 1e-12)
    out["roofline"] = {
        "kernel": "p7x::msv_tier_kernel<T> (the fast lane-per-target MSV kernel of all lanes of a batch that share a tier of register tiles, p7x_msv.hip)",
        "bound": "hbm", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 6),
        "traffic": msv_traffic_bytes(f"pfam:{len(hmms)}x{args.pfam_targets}", sum_bytes / max(nlaunch, 1e-9)),
        "kernel_ms": round(sum_ms / max(nlaunch, 1e-9), 4), "launches_timed": round(nlaunch, 1),
        "algorithmic_bytes": int(sum_bytes / max(nlaunch, 1e-9)), "queries_per_launch": round(float((w * lanes).sum()) / max(nlaunch, 1e-9), 2),
        "note": "average over the timed region's batches of the batch's largest fast-MSV launch (HIP events on its stream); the MSV "
                "working set (emission tables) lives in LDS, only residues stream from HBM (~1/M byte per cell): the binding roof "
                "is VALU issue, reported in `valu`",
        "valu": {"msv_gcups": round(kcups / 1e9, 1), "ops_per_cell": MSV_OPS_PER_CELL, "peak_gcups": round(VALU_LANE_OPS_PER_S / MSV_OPS_PER_CELL / 1e9, 1),
                 "frac": round(kcups * MSV_OPS_PER_CELL / VALU_LANE_OPS_PER_S, 4),
                 "whole_job": valu_roofline(nodes * residues, t_max, MSV_OPS_PER_CELL, "whole job against the scan kernel's issue roof")},
    }
    del db
    if world == 1 and not args.no_cpu_baseline:
        step = max(1, len(hmms) // max(1, args.pfam_cpu_profiles))
        sample = list(range(0, len(hmms), step))[:args.pfam_cpu_profiles]
        nt = min(args.pfam_cpu_targets, len(lengths))
        out["cpu_baseline"] = cpu_baseline_many(hmms, bg, flat, offsets, lengths, sample, nt)
        # the same sample through the device path (outside every timed region): the two legs' hit counts side by side
        sdb = plan7.SequenceDatabase.from_packed(hmms[0].alphabet, flat, offsets[:nt], lengths[:nt], device=local_rank)
        out["cpu_baseline"]["gpu_hits_same_sample"] = sum(len(h) for h in hmmer.hmmsearch([oms[e] for e in sample], sdb, cpus=host_threads))
        del sdb
    return out


def run_scan(args, rank, world, local_rank, dist, red_dev, torch, host_threads, lib, which="synthetic"):
    """BASELINE configs[2], the hmmscan orientation: the same profile library (device images resident, like a pressed
    database loaded into an OptimizedProfileBlock) against a block of query proteins through hmmer.hmmscan (one hit list per
    query SEQUENCE, Z = number of profiles).  which = "synthetic": BASELINE's 4k-protein block (SURVEY.md 8d config 3:
    4,000 targets, L ~ lognormal(5.65, 0.65) in [30, 5000], seed 43, half of them with a planted domain of a library
    entry); "fixture": the 2,100-protein fixture proteome (a real proteome's length distribution; the figure of rounds
    1-5).  N > 1: the profiles are dealt over the ranks (shard the profiles when the targets are few), every
    rank scans the whole block with its share."""
    import bench_workloads as bw
    from pyhmmer_amd import easel, hmmer, plan7
    hmms, bg, oms = lib["hmms"], lib["bg"], lib["oms"]
    if which == "fixture":
        with easel.SequenceFile(ROOT / "tests" / "golden" / "seqs" / "938293.PRJEB85.HG003687.faa", digital=True, alphabet=hmms[0].alphabet) as sf:
            proteome = sf.read_block()
        what = f"the {len(proteome)}-protein fixture proteome"
    else:
        f4, o4, l4, np4 = bw.make_targets(args.scan_targets, len(hmms), lib["templates"], lib["lengths"], seed=43, planted_frac=0.5)
        abc = hmms[0].alphabet
        proteome = easel.DigitalSequenceBlock(abc, [easel.DigitalSequence(abc, name=f"syn{t:05d}", sequence=f4[o4[t]:o4[t] + l4[t]]) for t in range(len(l4))])
        what = f"BASELINE's synthetic {len(proteome)}-protein block (L ~ lognormal(5.65, 0.65) in [30, 5000], seed 43, {np4} with a planted domain)"
    mine = oms[rank::world]
    block = plan7.OptimizedProfileBlock(hmms[0].alphabet, mine)

    def scan():
        return list(hmmer.hmmscan(proteome, block, cpus=host_threads, devices=[local_rank]))

    scan()                                  # images of this rank's profiles resident on this device, pools warm
    # a pass is a third of a second and varies by +-15 % from pass to pass (which hardware queues the host stage's kernels
    # share with the class chains is a matter of timing: DESIGN.md 8): three timed passes, the MEDIAN is reported, all three listed
    passes = []
    for _ in range(3):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = scan()
        torch.cuda.synchronize()
        passes.append(time.perf_counter() - t0)
    dt = sorted(passes)[1]
    t_max = dt
    if dist is not None:
        b = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(b, op=dist.ReduceOp.MAX)
        t_max = float(b.item())
    if rank != 0:
        return None
    nodes = float(sum(h.M for h in hmms))
    residues = float(proteome.total_length())
    out = {
        "workload": f"configs[2]: hmmscan orientation, {len(hmms)} library profiles (device images resident) x {what} "
                    f"({int(residues)} residues), hmmer.hmmscan defaults, profiles dealt over {world} GPU(s)",
        "value": round(nodes * residues / t_max / 1e9, 2), "unit": "GCUPS", "scaling": "strong",
        "profiles": len(hmms), "query_sequences": len(proteome), "seconds": round(t_max, 4),
        "ms_per_profile": round(1e3 * t_max / len(hmms), 5), "query_sequences_per_s": round(len(proteome) / t_max, 1),
        "passes_seconds_rank0": [round(x, 4) for x in passes], "seconds_is": "the median of three timed passes (max over ranks)",
        "hits_rank0": sum(len(r) for r in res),
    }
    out["roofline"] = valu_roofline(nodes * residues, t_max, MSV_OPS_PER_CELL, "p7x::msv_fast_kernel<R, K, half> (latency bound on this block: DESIGN.md 8)")
    if world == 1 and not args.no_cpu_baseline:
        pk = proteome.packed()
        step = max(1, len(hmms) // max(1, args.pfam_cpu_profiles))
        out["cpu_baseline"] = cpu_baseline_many(hmms, bg, pk.dsq, pk.offsets, pk.lengths, list(range(0, len(hmms), step))[:args.pfam_cpu_profiles], pk.n)
    return out


def run_nhmmer(args, rank, world, local_rank, dist, red_dev, torch):
    """BASELINE configs[4]: nhmmer long-target, one DNA profile (fixture bmyD, M = 1203) against a synthetic chromosome
    per GPU (weak scaling: every rank searches a chromosome of its own, both strands), the whole search timed:
    SSV scan + window filters + Viterbi scan + Forward on the device, seed bookkeeping / Backward / domain definition
    on the host."""
    import bench_workloads as bw
    from pyhmmer_amd import easel, plan7
    with plan7.HMMFile(ROOT / "tests" / "golden" / "hmms" / "bmyD.hmm") as hf:
        hmm = next(iter(hf))
    L = int(args.nhmmer_mbp * 1e6)
    seq = bw.make_chromosome(hmm, L, planted=50, seed=45 + rank)
    block = easel.DigitalSequenceBlock(hmm.alphabet, [easel.DigitalSequence(hmm.alphabet, name=f"chrSyn{rank}", sequence=seq)])
    from pyhmmer_amd import hmmer
    pli = plan7.LongTargetsPipeline(hmm.alphabet, device=local_rank, host_envelopes=args.nhmmer_envelopes)
    pli.search_hmm(hmm, block)                                   # warm-up: tables, workspaces
    t0 = time.perf_counter()
    alone = pli.search_hmm(hmm, block)                           # one search with nothing else in flight: its latency
    torch.cuda.synchronize()
    alone_s = time.perf_counter() - t0
    n = max(1, args.nhmmer_searches)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    scan_ms = 0.0
    # a stream of queries through the public entry point (the reference's own benchmark is 100 genes against one genome,
    # BASELINE.md 1): hmmer.nhmmer keeps several searches in flight (its default), the scan of one under the tails of the others
    for hits in hmmer.nhmmer([hmm] * n, block, devices=[local_rank], host_envelopes=args.nhmmer_envelopes):
        scan_ms += hits.timings_ms["msv_kernel"]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cells = 2.0 * L * hmm.M
    tot = cells
    if dist is not None:
        b = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(b, op=dist.ReduceOp.MAX)
        dt = float(b.item())
        c = torch.tensor([cells], dtype=torch.float64, device=red_dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        tot = float(c.item())
    if rank != 0:
        return None
    sc = hits.stage_counts
    ssv_ops = 0.75         # ssvlong_quad_kernel, PAIR flavour: a packed add per two cells and row, a packed max on every second row
    return {
        "roofline": {**valu_roofline(tot * n, dt, ssv_ops, "p7x::ssvlong_quad_kernel<R, PAIR> (whole search: scan + window stages + host tail)"),
                     "scan_kernel": {"kernel_ms": round(scan_ms / n, 3), "gcups": round(cells / (scan_ms / n * 1e-3) / 1e9, 1),
                                     "frac": round(cells / (scan_ms / n * 1e-3) * ssv_ops / VALU_LANE_OPS_PER_S, 4)}},
        "workload": f"configs[4]: nhmmer, profile {hmm.name} (M={hmm.M}) vs one synthetic {args.nhmmer_mbp:g} Mbp chromosome per GPU "
                    "(i.i.d. ACGT + 50 planted mutated consensus stretches), both strands, block_length 262144",
        "value": round(tot * n / dt / 1e9, 1), "unit": "GCUPS", "searches": n, "s_per_search": round(dt / n, 4),
        "s_one_search_alone": round(alone_s, 4), "hits_alone": len(alone),
        "mbp_per_s": round(2.0 * L * world * n / dt / 1e6, 1),
        "ssv_scan_kernel_ms": round(scan_ms / n, 3), "ssv_scan_gcups": round(cells / (scan_ms / n * 1e-3) / 1e9, 1),
        "windows": {"past_ssv": sc["msv"], "past_bias": sc["bias"], "past_vit": sc["vit"], "past_fwd": sc["fwd"]},
        "hits": len(hits), "reported": len(hits.reported),
        "ms": {"scan_and_seeds": round(hits.timings_ms["msv"], 1), "window_batches_device": round(hits.timings_ms["bias"], 1),
               "host_tail": round(hits.timings_ms["host_domaindef"], 1)},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nseq", type=int, default=1_000_000, help="targets per GPU")
    ap.add_argument("--seqlen", type=int, default=300)
    ap.add_argument("--hmm", default="KR")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline-depth", type=int, default=None, help="A/B: batches of queries in flight (default: the library's own, hmmer.hmmsearch)")
    ap.add_argument("--feeders", type=int, default=None, help="A/B: host threads issuing device stages (default: the library's own)")
    ap.add_argument("--finishers", type=int, default=None, help="A/B: host-stage threads (default: the library's own, one per batch in flight)")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000, help="targets timed through the CPU oracle (rank 0, N=1)")
    ap.add_argument("--queries-per-step", type=int, default=32,
                    help="a step is this many consecutive queries, each a complete search of the resident target block: the "
                         "pipeline's fill and drain (about two query times) then weigh as little in a 20-step run as in a long one")
    ap.add_argument("--batch", type=int, default=0, help="headline workload: queries per device batch (0: the library's own choice)")
    ap.add_argument("--oa-guard", type=float, default=None, help="A/B: the optimal-accuracy near-tie guard (default: the library's; 0 switches it off)")
    ap.add_argument("--ens-guard", type=float, default=None, help="A/B: the near-threshold guard of the sampled tracebacks (default: the library's; 0 switches it off)")
    ap.add_argument("--debug-option", action="append", default=[], metavar="NAME=VALUE",
                    help="diagnostics: set a knob of the library's test seam (p7x_debug_set_option), e.g. trace_finish=1")
    ap.add_argument("--host-ensembles", action="store_true", help="A/B: the stochastic traceback ensembles on the host workers instead of the device")
    ap.add_argument("--spinup-max", type=int, default=15, help="at most this many untimed 20-query windows before the warm-up")
    ap.add_argument("--inproc-devices", type=int, default=0,
                    help="headline workload through the API's own multi-device path: ONE process, hmmer.hmmsearch over this many devices "
                         "(a block of --nseq targets resident on each; P7X_BENCH_SHARE_DEVICE=1: all on this rank's device, a rehearsal) -- "
                         "next to the process-per-GPU path of --gpus N")
    ap.add_argument("--workload", choices=("all", "config1", "pfam", "scan", "nhmmer"), default="all",
                    help="all (default): the line is the pfam workload (BASELINE's metric config, configs[3]) with --steps / --warmup, "
                         "and config1 / scan / nhmmer are fields; pfam / scan / nhmmer: only that workload; config1: the line is "
                         "configs[1] (one profile x 1M targets per GPU, a step = --queries-per-step searches) with --steps / --warmup")
    for f in ("F1", "F2", "F3"):
        ap.add_argument(f"--{f}", type=float, default=None, help="diagnostics (pfam workload): filter threshold, e.g. --F1 1e-12 = the MSV stage alone")
    ap.add_argument("--pfam-profiles-per-step", type=int, default=1000, help="library profiles of one step of the pfam workload")
    ap.add_argument("--config1-steps", type=int, default=12, help="steps of the configs[1] FIELD when it is not the line (32 queries each; round 6: 12 -- over 4 steps the last batches' host stage is a fifth of the region)")
    ap.add_argument("--scan-targets", type=int, default=4000, help="query proteins of the scan workload's synthetic block (BASELINE: 4k)")
    ap.add_argument("--nhmmer-mbp", type=float, default=250.0, help="chromosome length per GPU")
    ap.add_argument("--nhmmer-searches", type=int, default=24, help="queries of the timed stream (hmmer.nhmmer; the reference's own benchmark is 100 genes against one genome)")
    ap.add_argument("--nhmmer-envelopes", type=int, default=0, help="A/B: 0 the library decides where envelopes are rescored, 1 host workers, 2 envelope kernel")
    ap.add_argument("--pfam-profiles", type=int, default=20000, help="library entries searched (the first ones of the 20k-entry library; default: all)")
    ap.add_argument("--pfam-cpu-profiles", type=int, default=40, help="cpu_baseline of the many-profile workloads: this many profiles, evenly spaced")
    ap.add_argument("--pfam-cpu-targets", type=int, default=100_000, help="... against the first this many targets")
    ap.add_argument("--pfam-library", type=int, default=20000)
    ap.add_argument("--pfam-targets", type=int, default=500_000, help="targets in total (sharded over the GPUs)")
    ap.add_argument("--pfam-batch", type=int, default=0, help="queries per device batch (0: the library's choice)")
    ap.add_argument("--pfam-depth", type=int, default=None, help="A/B, many-profile workload (default: the library's own)")
    ap.add_argument("--pfam-finishers", type=int, default=None, help="A/B, many-profile workload (default: the library's own)")
    args = ap.parse_args()
    line_steps, line_warmup = args.steps, args.warmup      # the driver's K and W belong to the line's workload
    c1_cpu = not args.no_cpu_baseline
    if args.workload != "config1":
        # configs[1] is a field here: a short stream (its steady state shows within a few dozen queries) and no CPU leg of its own
        args.steps, args.warmup, args.spinup_max, c1_cpu = max(1, args.config1_steps), 1, min(args.spinup_max, 4), False
        if args.workload in ("pfam", "scan", "nhmmer"):    # development switch: configs[1] shrinks to a token run
            args.steps, args.warmup, args.spinup_max = 1, 0, 1

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    # P7X_BENCH_SHARE_DEVICE=1: rehearsal of the N > 1 path on a box with one GPU (all ranks on device 0, gloo for the
    # few scalars that are exchanged); never used for reported numbers
    rehearsal = os.environ.get("P7X_BENCH_SHARE_DEVICE") == "1"
    if rehearsal:
        local_rank = 0
    red_dev = "cpu" if rehearsal else "cuda"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path (only the reported cpu_baseline runs on the host)")

    from pyhmmer_amd import _lib, plan7
    lib = _lib.lib()
    for item in args.debug_option:          # after torch: the library must bind to the HIP runtime torch has loaded
        name, _, value = item.partition("=")
        _lib.set_debug_option(name, int(value or 1))
    with plan7.HMMFile(ROOT / "tests" / "golden" / "hmms" / f"{args.hmm}.hmm") as hf:      # fixture data, not test code
        hmm = next(iter(hf))
    bg = plan7.Background(hmm.alphabet)
    om = plan7.OptimizedProfile(hmm, bg, args.seqlen)

    t0 = time.perf_counter()
    flat, offsets, lengths, planted = make_workload(hmm, args.nseq, args.seqlen, seed=42 + rank)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    db = plan7.SequenceDatabase.from_packed(hmm.alphabet, flat, offsets, lengths, device=local_rank)
    torch.cuda.synchronize()
    t_pack = time.perf_counter() - t0
    inproc = max(0, args.inproc_devices)
    if inproc > 1:                    # one process, N devices: every device gets a block of its own (weak scaling, like --gpus N)
        from pyhmmer_amd import hmmer as _hm
        ndev = lib.p7x_device_count()
        ids = [local_rank] * inproc if rehearsal else list(range(inproc))
        if not rehearsal and inproc > ndev:
            raise SystemExit(f"--inproc-devices {inproc} but {ndev} device(s) visible")
        parts = [db]
        extra_res = 0
        for k in range(1, inproc):
            f2, o2, l2, _ = make_workload(hmm, args.nseq, args.seqlen, seed=42 + rank + 1000 * k)
            parts.append(plan7.SequenceDatabase.from_packed(hmm.alphabet, f2, o2, l2, device=ids[k]))
            extra_res += int(l2.sum())
        db = _hm.ShardedDatabase.from_databases(parts)

    from pyhmmer_amd import hmmer
    # the ranks of one node share its CPUs: split them instead of letting every rank start a full-size worker pool
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    host_threads = max(2, host_cpus() // max(1, local_world))

    qps = max(1, args.queries_per_step)
    pli_opts = {} if args.oa_guard is None else {"oa_guard": args.oa_guard}
    if args.ens_guard is not None:
        pli_opts["ens_guard"] = args.ens_guard
    if args.host_ensembles:
        pli_opts["host_ensembles"] = True
    lanes_per_launch = args.batch or hmmer._auto_batch(db if inproc > 1 else hmmer.ShardedDatabase.from_database(db), hmm.M)

    def run(nsteps):
        """nsteps x queries_per_step searches of the same workload through the public entry point.  hmmsearch overlaps
        the device stage of later queries with the host stage of earlier ones (pipeline_depth), exactly as it does for
        distinct queries."""
        last, acc = None, {}
        for h in hmmer.hmmsearch((om for _ in range(nsteps * qps)), db, cpus=host_threads, batch=args.batch, **pipe_opts(args.pipeline_depth, args.feeders, args.finishers), **pli_opts):
            last = h
            for k, v in h.timings_ms.items():
                acc[k] = acc.get(k, 0.0) + v
        return last, acc

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed spin-up (independent of --warmup): a fresh process needs a few dozen queries before the host worker pool,
    # the pooled device workspaces and the clocks settle (first windows measure 10-25 % low).  Windows of 40 queries are
    # run until three consecutive ones agree to 2 % (at most `--spinup-max` windows); all ranks run the same number.
    spin = []
    for _ in range(args.spinup_max):
        t0 = time.perf_counter()
        run(max(1, 40 // qps))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            b = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            dist.all_reduce(b, op=dist.ReduceOp.MAX)
            dt = float(b.item())
        spin.append(dt)
        if len(spin) >= 3 and max(spin[-3:]) <= 1.02 * min(spin[-3:]):
            break
    hits = None
    if args.warmup > 0:
        hits, _ = run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    hits, stage = run(args.steps)
    pstats = hmmer.pipeline_stats()           # this rank's query pipeline over the timed region (before the barrier: its own clock)
    barrier()
    elapsed = time.perf_counter() - t0
    nqueries = args.steps * qps
    stage = {k: v / nqueries for k, v in stage.items()}          # per query (a batch's times are reported by each of its queries)
    # Outside the timed region: the same kernels without a second search in flight.  With several feeders the device
    # stages of two queries share the device, which raises throughput and stretches every single launch; the
    # stand-alone duration is what the kernel itself achieves.
    solo = {}
    idle_ms = []
    if pipe_effective(args.pipeline_depth, args.feeders, args.finishers)["pipeline_depth"] > 0 and rank == 0:
        n_solo = 5
        for _ in range(n_solo):          # ONE query at a time, nothing else in flight: the literal "single profile vs 1M sequences"
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            h = next(iter(hmmer.hmmsearch([om], db, pipeline_depth=0, cpus=host_threads, batch=1, **pli_opts)))
            idle_ms.append(1e3 * (time.perf_counter() - t1))
            for k, v in h.timings_ms.items():
                solo[k] = solo.get(k, 0.0) + v / n_solo

    residues = int(lengths.sum())
    cells_rank = float(hmm.M) * residues
    t_max, cells_total, seqs_total = elapsed, cells_rank, float(args.nseq)
    if inproc > 1:
        cells_total, seqs_total = float(hmm.M) * (residues + extra_res), float(args.nseq) * inproc
    # What makes a scaling run diagnosable: per rank, how its threads spent the timed region.  A rank whose
    # feeders mostly waited for a free slot is bound by its host stages (too few CPUs or too little depth), one whose feeders
    # mostly waited for the device is device bound (the intended state); rank 0 adds the time it took to merge the ranks' hits.
    wall = max(pstats.get("wall", 0.0), 1e-9)
    nfeed_eff = max(1, pstats.get("feeders", 1))
    rank_diag = {"rank": rank, "host_threads": host_threads, "elapsed_s": round(elapsed, 4),
                 "feeder_device_wait_frac": round(pstats.get("feeder_device_wait", 0.0) / (wall * nfeed_eff), 4),
                 "feeder_slot_wait_frac": round(pstats.get("feeder_slot_wait", 0.0) / (wall * nfeed_eff), 4),
                 "feeder_enqueue_frac": round(pstats.get("feeder_enqueue", 0.0) / (wall * nfeed_eff), 4),
                 "host_stage_s_per_batch": round(pstats.get("finish", 0.0) / max(1, pstats.get("batches", 1)), 4),
                 "consumer_finish_wait_frac": round(pstats.get("consumer_finish_wait", 0.0) / wall, 4),
                 "host_stages_in_flight": pstats.get("finishers"), "batches": pstats.get("batches")}
    merge_s = 0.0
    diags = [rank_diag]
    if dist is not None:
        buf = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(buf, op=dist.ReduceOp.MAX)
        t_max = float(buf.item())
        tot = torch.tensor([cells_rank, float(args.nseq)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        cells_total, seqs_total = float(tot[0].item()), float(tot[1].item())
        # per-GPU TopHits merged on the host of rank 0 (no data-path collective: this only moves the results)
        blobs = [None] * world
        dist.all_gather_object(blobs, hits.to_bytes())
        diags = [None] * world
        dist.all_gather_object(diags, rank_diag)
        if rank == 0:
            tm = time.perf_counter()
            merged = plan7.TopHits.from_bytes(blobs[0])
            for b in blobs[1:]:
                merged = merged.merge(plan7.TopHits.from_bytes(b))
            merge_s = time.perf_counter() - tm
            hits_total, reported_total = len(merged), len(merged.reported)
    if dist is None:
        hits_total, reported_total = len(hits), len(hits.reported)

    pfam = scan = None
    c1_steps, c1_warmup = args.steps, args.warmup
    if args.workload in ("all", "pfam", "scan"):
        del db                       # the configs[1] target block leaves HBM first
        lib = build_library(args, local_rank)
        if args.workload in ("all", "pfam"):
            args.steps, args.warmup = line_steps, line_warmup
            pfam = run_pfam(args, rank, world, local_rank, dist, red_dev, torch, host_threads, lib)
            args.steps, args.warmup = c1_steps, c1_warmup
        if args.workload in ("all", "scan"):
            scan = run_scan(args, rank, world, local_rank, dist, red_dev, torch, host_threads, lib, "synthetic")
            fx = run_scan(args, rank, world, local_rank, dist, red_dev, torch, host_threads, lib, "fixture")
            if scan is not None:
                scan["fixture_proteome"] = {k: fx[k] for k in ("workload", "value", "seconds", "query_sequences", "passes_seconds_rank0", "hits_rank0", "roofline")}
        del lib
    nh = None
    if args.workload in ("all", "nhmmer"):
        if args.workload == "nhmmer":
            del db
        nh = run_nhmmer(args, rank, world, local_rank, dist, red_dev, torch)

    if rank == 0:
        ms_per_step = 1e3 * t_max / args.steps
        gcups = cells_total * nqueries / t_max / 1e9
        sc = hits.stage_counts
        # ---- roofline of the dominant kernel (MSV), from HIP events recorded on the library's stream
        # one MSV launch serves the B queries of a device batch (each reads the residue tiles): per launch B x the
        # per-comparison bytes of SURVEY.md 8(d); every query of a batch reports its batch's launch duration
        B = lanes_per_launch
        msv_ms = stage["msv_kernel"]
        table_bytes = 29 * 16 * max(2, (hmm.M - 1) // 16 + 1)
        alg_bytes = B * (float(residues + 2 * args.nseq) + 16.0 * args.nseq + table_bytes)
        achieved_gbs = alg_bytes / (msv_ms * 1e-3) / 1e9
        msv_cups = B * cells_rank / (msv_ms * 1e-3)
        om_R = (hmm.M + 1) // 2 + 1
        om_R = ((om_R + 3) // 4) * 4 if om_R <= 160 else ((om_R + 15) // 16) * 16
        out = {
            "metric": "GCUPS (DP cells/s) + seqs/s for hmmsearch, Pfam-A vs proteome, 1/2/4/8 GPUs",
            "value": round(gcups, 2), "unit": "GCUPS",
            "n_gpus": world * max(1, inproc), "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/i16 filters (MSV, Viterbi), f32 Forward/Backward",
            "data": "synthetic",
            "seqs_per_s": round(seqs_total * nqueries / t_max, 1),
            "ms_per_query": round(1e3 * t_max / nqueries, 3),
            "config": {
                "workload": f"configs[1]: single profile {hmm.name} (M={hmm.M}, fixture {args.hmm}.hmm) vs {args.nseq} synthetic "
                            f"{args.seqlen}-aa targets per GPU, i.i.d. background + 0.1% planted positives, full pipeline "
                            "MSV->bias->Viterbi->Forward->Backward on device + domain definition/TopHits on host",
                "targets_per_gpu": args.nseq, "target_len": args.seqlen, "M": hmm.M, "parallelism": (f"ONE process, hmmer.hmmsearch over {inproc} devices (a resident block each), per-query merge" if inproc > 1
                                                                        else f"targets sharded over {world} GPU(s), host merge"),
                "timed_region": "hmmer.hmmsearch over steps x queries_per_step queries (the same profile each time), every query runs "
                                "the complete search; device stage of later queries overlaps the host stage of earlier ones "
                                "(pipeline_depth=%d batches); targets resident in HBM (pack+upload once: %.2fs, generation %.2fs, "
                                "not timed)" % (pipe_effective(args.pipeline_depth, args.feeders, args.finishers)["pipeline_depth"], t_pack, t_gen),
                "queries_per_step": qps, "queries_per_device_batch": lanes_per_launch,
                **pipe_effective(args.pipeline_depth, args.feeders, args.finishers), "host_threads_per_rank": host_threads,
                "spinup_windows_s": [round(x, 4) for x in spin],
                "latency_ms_per_query": round(stage.get("total", 0.0), 3),
                # ONE hmmsearch of the one profile against the resident 1M-target block, nothing else in flight (wall clock of the
                # call, median of five), and where that time goes (the call's own stage clocks, ms)
                "latency_ms_one_query_idle_device": round(sorted(idle_ms)[len(idle_ms) // 2], 3) if idle_ms else None,
                "idle_device_gcups": round(cells_rank / (sorted(idle_ms)[len(idle_ms) // 2] * 1e-3) / 1e9, 1) if idle_ms else None,
                "idle_device_stage_ms": ({k: round(solo[k], 3) for k in ("msv_kernel", "msv", "bias", "viterbi", "forward", "fwd_rows", "stage1",
                                                                        "envelopes", "ensemble_wait", "envelope_wait", "host_stage_busy", "stage2", "total") if k in solo}
                                         if solo else None),
            },
            "ranks": {"per_rank": diags, "merge_seconds_rank0": round(merge_s, 5),
                      "note": "fractions of the timed region, per feeder thread; slot wait = every batch slot was taken and the feeders "
                              "waited for a host stage (host bound), device wait = waiting for the cascade (device bound)"},
            "stages": {
                "n_targets": args.nseq, "past_msv": sc["msv"], "past_bias": sc["bias"], "past_vit": sc["vit"], "past_fwd": sc["fwd"],
                "hits": hits_total, "reported": reported_total, "planted": int(len(planted)),
                "device_ms": {k: round(v, 4) for k, v in stage.items()},
                # later-stage work, reported separately from the headline: cells actually run per stage
                "stage_gcups": {
                    "msv": round(B * cells_rank / (stage["msv_kernel"] * 1e-3) / 1e9, 1),
                    "viterbi": round(B * sc["bias"] * args.seqlen * hmm.M / (stage["viterbi"] * 1e-3) / 1e9, 1),
                    "forward": round(B * sc["vit"] * args.seqlen * hmm.M / (stage["forward"] * 1e-3) / 1e9, 1),
                },
            },
            "roofline": {
                "kernel": "p7x::msv_fast_kernel<R> (lane-per-target MSV, p7x_msv.hip; R = %d row registers)" % om_R,
                "bound": "hbm", "achieved": round(achieved_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved_gbs / HBM_PEAK_GBS, 6),
                # HBM bytes per launch from the PMC passes of the same command (profiles/msv_traffic.json names the runs);
                # None when this workload has no committed measurement.
                "traffic": msv_traffic_bytes(f"{args.hmm}:{args.nseq}x{args.seqlen}:b{B}"),
                "note": "the MSV working set (emission tables) lives in LDS; only residues stream from HBM (~1/M byte per cell), "
                        "so the binding roof is VALU issue, reported below",
                "valu": {"msv_gcups": round(msv_cups / 1e9, 1), "ops_per_cell": MSV_OPS_PER_CELL,
                         "peak_gcups": round(VALU_LANE_OPS_PER_S / MSV_OPS_PER_CELL / 1e9, 1),
                         "frac": round(msv_cups * MSV_OPS_PER_CELL / VALU_LANE_OPS_PER_S, 4),
                         # against what the chip sustains for the packed ops alone / with the kernel's LDS reads
                         "sustained_peak_gcups": round(VALU_SUSTAINED_GCUPS, 1), "frac_sustained": round(msv_cups / 1e9 / VALU_SUSTAINED_GCUPS, 4),
                         "sustained_mix_gcups": MSV_SUSTAINED_MIX_GCUPS,
                         # the same launch when no other search shares the device (measured after the timed region)
                         "standalone": ({"kernel_ms": round(solo["msv_kernel"], 4),
                                         "msv_gcups": round(cells_rank / (solo["msv_kernel"] * 1e-3) / 1e9, 1),
                                         "frac": round(cells_rank / (solo["msv_kernel"] * 1e-3) * MSV_OPS_PER_CELL / VALU_LANE_OPS_PER_S, 4),
                                         "frac_sustained": round(cells_rank / (solo["msv_kernel"] * 1e-3) / 1e9 / VALU_SUSTAINED_GCUPS, 4)}
                                        if solo else None)},
                "kernel_ms": round(msv_ms, 4), "algorithmic_bytes": int(alg_bytes), "queries_per_launch": B,
                "kernels": tail_kernels(hmm, args, sc, solo, cells_rank),
                "pipeline": pipeline_valu_budget(hmm, args, sc, B, 1e3 * t_max / nqueries * B),
            },
        }
        if pfam is not None:
            out["pfam"] = pfam
        if scan is not None:
            out["scan"] = scan
        if nh is not None:
            out["nhmmer"] = nh
        if world == 1 and c1_cpu:
            out["cpu_baseline"] = cpu_baseline(hmm, bg, flat, offsets, lengths, min(args.cpu_sample, args.nseq), args.seqlen)
        if args.workload in ("all", "pfam") and pfam is not None:
            # the line is BASELINE's metric config (configs[3]); configs[1], the scan orientation and nhmmer are fields
            first = {"metric": out["metric"], "value": pfam["value"], "unit": "GCUPS", "n_gpus": world,
                     "steps": line_steps, "warmup": line_warmup, "ms_per_step": round(1e3 * pfam["seconds"] / max(1, line_steps), 3),
                     "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": out["dtype"], "data": "synthetic",
                     "seqs_per_s": pfam["seqs_per_s"],
                     "config": {"workload": pfam["workload"],
                                "step": f"{pfam['profiles_per_step']} library profiles x all {pfam['targets']} targets (complete searches, hit lists merged)",
                                "profiles_per_step": pfam["profiles_per_step"], "targets": pfam["targets"], "library": pfam["library"],
                                "parallelism": f"targets sharded by residues over {world} GPU(s), one process each, host merge of the per-GPU TopHits",
                                "timed_region": "one hmmer.hmmsearch call over the steps' profiles + gather + merge, barrier and synchronize on both "
                                                "sides; device images of the profiles and the targets resident in HBM (an untimed pass over the whole "
                                                "library first); library defaults", **{k: pfam[k] for k in ("batch", "pipeline_depth", "feeders", "finishers", "library_defaults")}},
                     "roofline": pfam.pop("roofline"), "cpu_baseline": pfam.pop("cpu_baseline", None),
                     "pfam": pfam}
            if args.workload == "all":
                first["config1"] = out
            if scan is not None:
                first["scan"] = out.pop("scan", scan)
            if nh is not None:
                first["nhmmer"] = out.pop("nhmmer", nh)
            out.pop("pfam", None)
            out = first
        elif args.workload in ("scan", "nhmmer"):
            sub = scan if args.workload == "scan" else nh
            # --workload scan / nhmmer: that workload IS the line (value, config, roofline, cpu_baseline)
            secs = sub.get("seconds", sub.get("s_per_search", 0.0) * sub.get("searches", 1))
            out = {"metric": out["metric"], "value": sub["value"], "unit": "GCUPS", "n_gpus": world,
                   "steps": sub.get("searches", 1), "warmup": 1, "ms_per_step": round(1e3 * secs / max(1, sub.get("searches", 1)), 3),
                   "higher_is_better": True, "scaling": sub.get("scaling", "weak"), "vs_baseline": None,
                   "dtype": out["dtype"] if args.workload != "nhmmer" else "i16 SSV scan and window filters, f32 Forward/Backward",
                   "data": "synthetic",
                   "config": {"workload": sub["workload"], "step": "one pass over the whole workload after one untimed pass"
                              if args.workload != "nhmmer" else "one complete search of a stream of searches",
                              "parallelism": f"{world} GPU(s), one process each"},
                   "roofline": sub.get("roofline"), "cpu_baseline": sub.get("cpu_baseline"), args.workload: sub}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
