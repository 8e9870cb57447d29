"""ctypes binding of ``libp7x.so`` (the C-ABI declared in ``include/p7x.h``).

The library is built in-tree by :func:`build` (``hipcc --offload-arch=gfx950``); there is no
pure-Python or CPU fallback: if the shared object is missing, importing any compute entry
point raises ``ImportError`` with the build command.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess
from pathlib import Path

_PKG = Path(__file__).resolve().parent
_ROOT = _PKG.parent
CSRC = _PKG / "csrc"
LIB_PATH = _PKG / "libp7x.so"

SOURCES = ["p7x_profile.cpp", "p7x_device.hip", "p7x_devimage.hip", "p7x_msv.hip", "p7x_vitfwd.hip", "p7x_vitpk.hip", "p7x_fwdpk.hip", "p7x_envelope.hip", "p7x_ensemble.hip", "p7x_ssvlong.hip", "p7x_longtarget.hip",
           "p7x_envscore.hip", "p7x_pipeline.hip", "p7x_domaindef.cpp", "p7x_tophits.cpp"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC=/path/to/hipcc)")


ISA_CHECKED = ("p7x_msv.hip", "p7x_envelope.hip")


def isa_path(name: str) -> Path:
    """Where build() leaves the gfx950 assembly of a translation unit of ISA_CHECKED."""
    return CSRC / "build" / (name + ".isa.s")


def fresh_isa(name: str):
    """The assembly build() wrote for <name>, if it is newer than the unit and every header; None otherwise."""
    isa, src = isa_path(name), CSRC / name
    deps = [src] + list(CSRC.glob("*.hpp")) + list(CSRC.glob("*.h")) + [_ROOT / "include" / "p7x.h"]
    if isa.exists() and isa.stat().st_mtime >= max(p.stat().st_mtime for p in deps if p.exists()):
        return isa
    return None


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every translation unit for gfx950 and link ``libp7x.so`` in-tree."""
    srcs = [CSRC / s for s in SOURCES if (CSRC / s).exists()]
    deps = srcs + list(CSRC.glob("*.hpp")) + list(CSRC.glob("*.h")) + [_ROOT / "include" / "p7x.h"]
    if LIB_PATH.exists() and not force:
        newest = max(p.stat().st_mtime for p in deps)
        if LIB_PATH.stat().st_mtime >= newest:
            return LIB_PATH
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)
    hipcc = _hipcc()
    common = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
              "--offload-arch=gfx950", "-I", str(_ROOT / "include"), "-Wno-unused-result"]
    common += os.environ.get("P7X_CXXFLAGS", "").split()        # build-time experiments (e.g. -DP7X_ENV_UNROLL_MAX=16)
    objs = []
    procs = []
    for s in srcs:
        o = objdir / (s.name + ".o")
        objs.append(o)
        if (not force) and o.exists() and o.stat().st_mtime >= max(p.stat().st_mtime for p in deps if p.suffix in (".hpp", ".h") or p == s):
            continue
        cmd = [hipcc, *common, "-c", str(s), "-o", str(o)]     # hipcc compiles .cpp units as HIP too (host code only)
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    # the generated ISA of the two translation units whose hand-scheduled parts the CPU tests check statically
    # (tests/test_host.py: LDS hazards of the MSV kernels, the envelope kernel's row loops): written beside the objects so
    # that the tests read what was built instead of compiling the units again (three minutes of the CPU suite)
    for name in ISA_CHECKED:
        src = CSRC / name
        isa = isa_path(name)
        if src.exists() and (force or not isa.exists() or isa.stat().st_mtime < max(p.stat().st_mtime for p in deps if p.suffix in (".hpp", ".h") or p == src)):
            cmd = [hipcc, *[c for c in common if c != "-fPIC"], "-S", "--cuda-device-only", str(src), "-o", str(isa)]
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s.name}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB_PATH), *map(str, objs), "-lpthread"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB_PATH


# --------------------------------------------------------------------------- ctypes mirrors of p7x.h

class HmmView(C.Structure):
    _fields_ = [
        ("M", C.c_int32), ("abc_type", C.c_int32),
        ("t", C.POINTER(C.c_float)), ("mat", C.POINTER(C.c_float)), ("ins", C.POINTER(C.c_float)),
        ("compo", C.POINTER(C.c_float)),
        ("evparam", C.c_float * 6), ("cutoff", C.c_float * 6),
        ("max_length", C.c_int32),
        ("name", C.c_char_p), ("acc", C.c_char_p), ("desc", C.c_char_p),
        ("consensus", C.c_char_p), ("rf", C.c_char_p), ("mm", C.c_char_p), ("cs", C.c_char_p),
    ]


class OprofileInfo(C.Structure):
    _fields_ = [
        ("M", C.c_int32), ("K", C.c_int32), ("Kp", C.c_int32), ("abc_type", C.c_int32),
        ("L", C.c_int32), ("max_length", C.c_int32), ("mode", C.c_int32),
        ("Q16", C.c_int32), ("Q8", C.c_int32), ("Q4", C.c_int32),
        ("tbm_b", C.c_uint8), ("tec_b", C.c_uint8), ("tjb_b", C.c_uint8), ("base_b", C.c_uint8), ("bias_b", C.c_uint8),
        ("scale_b", C.c_float),
        ("xw", (C.c_int16 * 2) * 4),
        ("scale_w", C.c_float), ("base_w", C.c_int16), ("ddbound_w", C.c_int16), ("ncj_roundoff", C.c_float),
        ("xf", (C.c_float * 2) * 4),
        ("evparam", C.c_float * 6), ("cutoff", C.c_float * 6), ("compo", C.c_float * 20),
        ("nj", C.c_float),
    ]


class PipelineCfg(C.Structure):
    _fields_ = [
        ("by_E", C.c_int32), ("E", C.c_double), ("T", C.c_double),
        ("dom_by_E", C.c_int32), ("domE", C.c_double), ("domT", C.c_double), ("use_bit_cutoffs", C.c_int32),
        ("inc_by_E", C.c_int32), ("incE", C.c_double), ("incT", C.c_double),
        ("incdom_by_E", C.c_int32), ("incdomE", C.c_double), ("incdomT", C.c_double),
        ("Z", C.c_double), ("domZ", C.c_double), ("Z_setby", C.c_int32), ("domZ_setby", C.c_int32),
        ("F1", C.c_double), ("F2", C.c_double), ("F3", C.c_double),
        ("do_max", C.c_int32), ("do_biasfilter", C.c_int32), ("do_null2", C.c_int32),
        ("seed", C.c_uint32), ("mode", C.c_int32), ("host_threads", C.c_int32), ("host_envelopes", C.c_int32), ("host_regions", C.c_int32),
        ("long_targets", C.c_int32), ("strands", C.c_int32), ("B1", C.c_int32), ("B2", C.c_int32), ("B3", C.c_int32),
        ("block_length", C.c_int32), ("window_length", C.c_int32), ("evalue_window_length", C.c_int32), ("lt_part", C.c_int32), ("lt_nparts", C.c_int32), ("oa_guard", C.c_float),
        ("f3_guard", C.c_float),
        ("lt_resident_key", C.c_uint64),
        ("host_ensembles", C.c_int32),
        ("ens_guard", C.c_float),
    ]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "nmodels", "nseqs", "nres", "nnodes", "n_past_msv", "n_past_bias", "n_past_vit", "n_past_fwd",
        "n_output", "pos_past_msv", "pos_past_bias", "pos_past_vit", "pos_past_fwd", "pos_output")]


class DomainRec(C.Structure):
    _fields_ = [
        ("ienv", C.c_int64), ("jenv", C.c_int64), ("iali", C.c_int64), ("jali", C.c_int64),
        ("iorf", C.c_int64), ("jorf", C.c_int64),
        ("envsc", C.c_float), ("domcorrection", C.c_float), ("dombias", C.c_float), ("oasc", C.c_float),
        ("bitscore", C.c_float), ("lnP", C.c_double),
        ("is_reported", C.c_int32), ("is_included", C.c_int32),
        ("N", C.c_int32), ("hmmfrom", C.c_int32), ("hmmto", C.c_int32), ("M", C.c_int32),
        ("sqfrom", C.c_int64), ("sqto", C.c_int64), ("L", C.c_int64),
        ("model", C.c_char_p), ("mline", C.c_char_p), ("aseq", C.c_char_p), ("ppline", C.c_char_p),
        ("rfline", C.c_char_p), ("mmline", C.c_char_p), ("csline", C.c_char_p),
        ("hmmname", C.c_char_p), ("hmmacc", C.c_char_p), ("hmmdesc", C.c_char_p),
        ("sqname", C.c_char_p), ("sqacc", C.c_char_p), ("sqdesc", C.c_char_p),
    ]


class HitRec(C.Structure):
    _fields_ = [
        ("name", C.c_char_p), ("acc", C.c_char_p), ("desc", C.c_char_p),
        ("seqidx", C.c_int64), ("window_length", C.c_int32), ("sortkey", C.c_double),
        ("score", C.c_float), ("pre_score", C.c_float), ("sum_score", C.c_float),
        ("lnP", C.c_double), ("pre_lnP", C.c_double), ("sum_lnP", C.c_double),
        ("nexpected", C.c_float),
        ("nregions", C.c_int32), ("nclustered", C.c_int32), ("noverlaps", C.c_int32),
        ("nenvelopes", C.c_int32), ("ndom", C.c_int32),
        ("flags", C.c_uint32), ("nreported", C.c_int32), ("nincluded", C.c_int32), ("best_domain", C.c_int32),
    ]


# every symbol include/p7x.h declares: (name, restype, argtypes)
_VP = C.c_void_p
_SIGNATURES = {
    "p7x_abi_version": (C.c_int, []),
    "p7x_tophits_merge_longtargets": (C.c_int, [C.POINTER(_VP), C.c_size_t, C.POINTER(_VP)]),
    "p7x_tophits_merge_many": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t, C.c_int, C.POINTER(_VP)]),
    "p7x_tophits_get_guard_counts": (C.c_int, [_VP, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "p7x_tophits_get_ensemble_counts": (C.c_int, [_VP, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "p7x_hmm_max_length": (C.c_int, [C.POINTER(HmmView), C.c_double, C.POINTER(C.c_int32)]),
    "p7x_expf_neg": (None, [_VP, _VP, C.c_size_t]),
    "p7x_oprofile_create": (C.c_int, [C.POINTER(HmmView), _VP, C.c_int32, C.POINTER(_VP)]),
    "p7x_oprofile_destroy": (None, [_VP]),
    "p7x_oprofile_get_info": (C.c_int, [_VP, C.POINTER(OprofileInfo)]),
    "p7x_oprofile_striped": (C.c_int64, [_VP, C.c_int, _VP, C.c_size_t]),
    "p7x_device_count": (C.c_int, []),
    "p7x_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "p7x_seqdb_create": (C.c_int, [C.c_int, C.c_int32, _VP, _VP, _VP, C.c_size_t, C.POINTER(_VP)]),
    "p7x_seqdb_destroy": (None, [_VP]),
    "p7x_seqdb_ntargets": (C.c_int64, [_VP]),
    "p7x_seqdb_nresidues": (C.c_int64, [_VP]),
    "p7x_msv_filter": (C.c_int, [_VP, C.c_int, _VP, C.c_int32, C.POINTER(C.c_float)]),
    "p7x_vit_filter": (C.c_int, [_VP, C.c_int, _VP, C.c_int32, C.POINTER(C.c_float)]),
    "p7x_fwd_parser": (C.c_int, [_VP, C.c_int, _VP, C.c_int32, C.POINTER(C.c_float)]),
    "p7x_bck_parser": (C.c_int, [_VP, C.c_int, _VP, C.c_int32, C.POINTER(C.c_float)]),
    "p7x_filters_batch": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "p7x_pipeline_cfg_default": (None, [C.POINTER(PipelineCfg)]),
    "p7x_search_block": (C.c_int, [C.POINTER(PipelineCfg), _VP, _VP, _VP, _VP, _VP, _VP, C.POINTER(_VP)]),
    "p7x_postprocess_targets": (C.c_int, [C.POINTER(PipelineCfg), _VP, _VP, _VP, _VP, C.c_size_t, _VP, C.c_size_t, _VP, _VP, _VP,
                                          _VP, _VP, _VP, _VP, _VP, C.POINTER(_VP)]),
    "p7x_tophits_destroy": (None, [_VP]),
    "p7x_tophits_destroy_many": (None, [_VP, C.c_size_t]),
    "p7x_tophits_clone": (_VP, [_VP]),
    "p7x_tophits_nhits": (C.c_int64, [_VP]),
    "p7x_tophits_get_counters": (C.c_int, [_VP, C.POINTER(Counters)]),
    "p7x_tophits_get_cfg": (C.c_int, [_VP, C.POINTER(PipelineCfg)]),
    "p7x_tophits_get_hit": (C.c_int, [_VP, C.c_int64, C.POINTER(HitRec)]),
    "p7x_tophits_get_domain": (C.c_int, [_VP, C.c_int64, C.c_int32, C.POINTER(DomainRec)]),
    "p7x_tophits_merge": (C.c_int, [_VP, _VP]),
    "p7x_tophits_serialize": (C.c_int64, [_VP, _VP, C.c_size_t]),
    "p7x_tophits_deserialize": (_VP, [_VP, C.c_size_t]),
    "p7x_tophits_sort_by_key": (C.c_int, [_VP]),
    "p7x_tophits_threshold": (C.c_int, [_VP]),
    "p7x_tophits_sort_by_seqidx": (C.c_int, [_VP]),
    "p7x_tophits_is_sorted": (C.c_int, [_VP, C.c_int]),
    "p7x_tophits_set_hit_flags": (C.c_int, [_VP, C.c_int64, C.c_uint32]),
    "p7x_tophits_set_hit_text": (C.c_int, [_VP, C.c_int64, C.c_int, C.c_char_p]),
    "p7x_tophits_get_timings": (C.c_int, [_VP, C.POINTER(C.c_double), C.c_int]),
    "p7x_search_block_begin": (C.c_int, [C.POINTER(PipelineCfg), _VP, _VP, _VP, C.POINTER(_VP)]),
    "p7x_search_block_enqueue": (C.c_int, [C.POINTER(PipelineCfg), _VP, _VP, _VP, C.POINTER(_VP)]),
    "p7x_search_block_wait": (C.c_int, [_VP]),
    "p7x_search_block_finish": (C.c_int, [_VP, _VP, _VP, _VP, C.POINTER(_VP)]),
    "p7x_pending_destroy": (None, [_VP]),
    "p7x_search_batch_enqueue": (C.c_int, [C.POINTER(PipelineCfg), C.POINTER(_VP), C.c_size_t, _VP, _VP, C.POINTER(_VP)]),
    "p7x_search_batch_finish": (C.c_int, [_VP, _VP, _VP, _VP, C.POINTER(_VP)]),
    "p7x_pending_nqueries": (C.c_size_t, [_VP]),
    "p7x_debug_log_of_float": (C.c_int, [C.c_int, _VP, _VP, C.c_size_t]),
    "p7x_debug_choice": (C.c_int, [_VP, C.c_int, C.c_uint32, _VP, _VP]),
    "p7x_debug_order_spread": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32, C.c_int32, C.c_int, _VP]),
    "p7x_debug_ssv_tables": (C.c_int64, [_VP, C.c_int, _VP, _VP, _VP, C.c_size_t]),
    "p7x_debug_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "p7x_longtargets_release_resident": (C.c_int, [C.c_int, C.c_uint64]),
    "p7x_debug_ensemble": (C.c_int, [_VP, _VP, C.c_int64, C.c_int32, C.c_int32, C.c_uint32, C.c_int, _VP, _VP, C.c_int32, _VP, _VP]),
    "p7x_search_batch_raw": (C.c_int, [C.POINTER(PipelineCfg), C.POINTER(_VP), C.c_size_t, _VP, _VP, _VP, _VP, _VP]),
    "p7x_search_longtargets": (C.c_int, [C.POINTER(PipelineCfg), _VP, C.c_int, _VP, _VP, _VP, C.c_size_t, _VP, _VP, _VP, C.POINTER(_VP)]),
    "p7x_ssv_longtarget_seeds": (C.c_int64, [C.POINTER(PipelineCfg), _VP, C.c_int, _VP, C.c_int64, C.c_int, _VP, C.c_size_t]),
    "p7x_forward_parser_exact": (C.c_int, [_VP, _VP, C.c_int32, C.POINTER(C.c_float)]),
    "p7x_longtarget_from_seeds": (C.c_int, [C.POINTER(PipelineCfg), _VP, _VP, _VP, _VP, C.c_size_t, _VP, _VP, _VP, _VP, _VP, _VP, _VP,
                                            C.c_size_t, C.POINTER(_VP)]),
    "p7x_oprofile_write_pressed": (C.c_int, [_VP, C.POINTER(C.c_int64), _VP, C.c_size_t, C.POINTER(C.c_size_t), _VP, C.c_size_t,
                                             C.POINTER(C.c_size_t)]),
    "p7x_oprofile_read_pressed": (C.c_int, [_VP, C.c_size_t, _VP, C.c_size_t, _VP, C.POINTER(_VP), C.POINTER(C.c_size_t),
                                            C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]),
    "p7x_oprofile_get_string": (C.c_int, [_VP, C.c_int, C.c_char_p, C.c_size_t]),
    "p7x_scan_collect": (C.c_int, [C.POINTER(_VP), C.c_size_t, C.POINTER(PipelineCfg), C.c_size_t, _VP, _VP, _VP, _VP, C.POINTER(_VP)]),
    "p7x_scan_accum_create": (C.c_int, [C.POINTER(PipelineCfg), C.c_size_t, _VP, _VP, _VP, _VP, C.POINTER(_VP)]),
    "p7x_scan_accum_add": (C.c_int, [_VP, C.POINTER(_VP), C.c_size_t]),
    "p7x_scan_accum_add_indexed": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(C.c_int64), C.c_size_t]),
    "p7x_debug_tophits_set_stages": (C.c_int, [_VP, _VP, C.c_size_t]),
    "p7x_scan_accum_finish": (C.c_int, [_VP, C.POINTER(_VP)]),
    "p7x_scan_accum_destroy": (None, [_VP]),
    "p7x_fasta_parse": (C.c_int, [_VP, C.c_size_t, _VP, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                  _VP, _VP, _VP, _VP, _VP, _VP, C.POINTER(C.c_size_t)]),
    "p7x_last_error": (C.c_char_p, []),
}

_lib = None


def lib() -> C.CDLL:
    """Load ``libp7x.so`` (building is explicit: ``python -m pyhmmer_amd.build`` / ``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc); there is no CPU fallback.")
        l = C.CDLL(str(LIB_PATH))
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)      # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        if l.p7x_abi_version() != 8:
            raise ImportError("libp7x ABI version mismatch; rebuild")
        _lib = l
    return _lib


def declared_symbols():
    return sorted(_SIGNATURES)


def last_error() -> str:
    e = lib().p7x_last_error()
    return e.decode() if e else ""


def set_debug_option(name: str, value: int) -> None:
    """Test / diagnostic seam (``p7x_debug_set_option``): kernel-family choices for parity tests, traces.  -1 unsets."""
    if lib().p7x_debug_set_option(name.encode(), int(value)) != 0:
        raise ValueError(last_error())
