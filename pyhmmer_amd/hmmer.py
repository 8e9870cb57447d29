"""``hmmsearch`` with the reference's calling convention (``src/pyhmmer/hmmer/_hmmsearch.py:294-436``),
running every query on one or several MI355X.

The reference parallelises either over queries (worker threads, one ``Pipeline`` each,
``hmmer/_base.py:416-489``) or over targets ("reverse" dispatcher: residue-balanced target chunks, partial
``TopHits`` combined with ``TopHits.merge``, ``_hmmsearch.py:115-289``).  On GPUs the second model is the
natural one: the target block is sharded by residues across devices (``_make_chunks``), each device keeps its
shard resident in HBM, every query visits every shard, and the per-device ``TopHits`` are merged and
re-thresholded on the host.  No collective is involved.
"""
from __future__ import annotations

import os
import sys
import time
import threading
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Iterable, Iterator, List, Optional, Sequence, Union

from . import _lib
from .easel import Alphabet, DigitalSequenceBlock, SequenceFile
from .plan7 import HMM, LongTargetsPipeline, OptimizedProfile, Pipeline, Profile, SequenceDatabase, TopHits

__all__ = ["hmmsearch", "hmmscan", "nhmmer", "hmmpress", "make_chunks", "ShardedDatabase", "ReplicatedDatabase"]


def make_chunks(block: DigitalSequenceBlock, n: int) -> List[DigitalSequenceBlock]:
    """Split ``block`` into ``n`` contiguous chunks holding about the same number of residues
    (reference ``_ReverseSEARCHDispatcher._make_chunks``, ``_hmmsearch.py:153-171``)."""
    if n <= 1:
        return [block]
    total = sum(len(s) for s in block)
    target = total / n
    chunks: List[DigitalSequenceBlock] = []
    cur: list = []
    acc = 0
    for s in block:
        cur.append(s)
        acc += len(s)
        if acc >= target and len(chunks) < n - 1:
            chunks.append(DigitalSequenceBlock(block.alphabet, cur))
            cur, acc = [], 0
    chunks.append(DigitalSequenceBlock(block.alphabet, cur))
    while len(chunks) < n:
        chunks.append(DigitalSequenceBlock(block.alphabet, []))
    return chunks


def default_devices() -> List[int]:
    """Every HIP device the process can see: what a search uses when the caller names none -- the reference's default is
    the whole machine as well (``cpus=0``: every physical core, ``_hmmsearch.py:384``).  ``HIP_VISIBLE_DEVICES`` /
    ``ROCR_VISIBLE_DEVICES`` narrow it, as ``cpus=`` or ``taskset`` narrow the reference's; a process that drives one GPU
    of a node (one rank per GPU under ``torch.distributed``) passes ``devices=[local_rank]`` or a resident
    ``SequenceDatabase``."""
    return list(range(max(1, int(_lib.lib().p7x_device_count()))))


class ShardedDatabase:
    """Target block sharded by residues over several devices; each shard is a :class:`SequenceDatabase`."""

    def __init__(self, block: DigitalSequenceBlock, devices: Sequence[int]):
        self.block = block
        self.devices = list(devices)
        self.chunks = make_chunks(block, len(self.devices))
        self.shards = [SequenceDatabase(c, device=d) for c, d in zip(self.chunks, self.devices)]

    @classmethod
    def from_database(cls, database: SequenceDatabase) -> "ShardedDatabase":
        """One resident shard that was packed by the caller (``SequenceDatabase.from_packed``)."""
        self = cls.__new__(cls)
        self.block = database.block
        self.devices = [database.device]
        self.chunks = [database.block]
        self.shards = [database]
        return self

    @classmethod
    def from_databases(cls, databases: Sequence[SequenceDatabase]) -> "ShardedDatabase":
        """Shards that the caller packed and made resident itself, one per device (``SequenceDatabase.from_packed``): the
        targets of shard k follow those of shard k - 1 in the merged hit lists' accounting, as chunks of one block would."""
        self = cls.__new__(cls)
        self.block = None
        self.devices = [d.device for d in databases]
        self.chunks = [d.block for d in databases]
        self.shards = list(databases)
        return self

    # A "pending" search is a list with one entry per shard: the (handle, profiles, database, labels) tuple of
    # Pipeline._search_enqueue_batch.  Stage 1 is asynchronous on every device (own stream per search), so one host
    # thread queues a batch of queries on all shards before it waits for any of them: the shards run side by side
    # without per-query threads, and the query pipeline of hmmsearch keeps its depth with any number of shards.
    def enqueue(self, pipelines: Sequence[Pipeline], queries) -> list:
        """Queue stage 1 (device filters + parsers) of a batch of queries on every shard, without waiting."""
        pendings: list = []
        try:
            for i, shard in enumerate(self.shards):
                pendings.append(pipelines[i]._search_enqueue_batch(queries, shard))
        except BaseException:
            self.abandon(pendings)
            raise
        return pendings

    def wait(self, pendings: list) -> None:
        for pend in pendings:
            Pipeline._search_wait(pend)

    @staticmethod
    def abandon(pendings) -> None:
        """Release the device side of searches that will not be finished (a failure on another shard, an abandoned
        iterator): every handle is destroyed exactly once."""
        for pend in pendings or ():
            _lib.lib().p7x_pending_destroy(pend[0])

    def finish(self, pendings: list, raw: bool = False) -> List[TopHits]:
        """Stage 2 (domain definition, hit lists) on every shard, then the per-query merge (``TopHits.merge``:
        concatenate, sum the counters and ``Z``, re-threshold, re-sort -- reference ``_hmmsearch.py:259-263``).
        ``raw`` (one shard only: the scan orientation): the results as ``plan7.HitHandles`` instead of ``TopHits`` objects."""
        per_shard: list = []
        todo = list(pendings)
        if raw:
            if len(todo) != 1:
                self.abandon(todo)
                raise ValueError("raw results come from one shard")
            return Pipeline._search_finish_batch(todo[0], raw=True)
        if len(todo) > 1:
            # the host stage of a shard is mostly a wait for its own device (envelope kernel, ensembles: tens of milliseconds):
            # the shards' stages run side by side, one thread each -- one after the other they would add up to a host stage
            # of N x that per batch and starve N devices (the C call releases the GIL)
            pool = getattr(self, "_finish_pool", None)
            if pool is None:
                pool = self._finish_pool = ThreadPoolExecutor(max_workers=len(self.shards), thread_name_prefix="p7x-shard-finish")
            futs = [pool.submit(Pipeline._search_finish_batch, pend) for pend in todo]      # every handle is consumed, also on failure
            err = None
            for f in futs:
                try:
                    per_shard.append(f.result())
                except BaseException as e:          # noqa: BLE001 - the other shards finish (they own device buffers), then it surfaces
                    err = err or e
            if err is not None:
                raise err
            return [hits[0].merge(*hits[1:]) for hits in zip(*per_shard)]
        try:
            while todo:
                pend = todo.pop(0)                      # finish consumes the handle, also when it fails
                per_shard.append(Pipeline._search_finish_batch(pend))
        except BaseException:
            self.abandon(todo)
            raise
        return per_shard[0]

    def search(self, pipelines: Sequence[Pipeline], queries) -> List[TopHits]:
        pendings = self.enqueue(pipelines, queries)
        try:
            self.wait(pendings)
        except BaseException:
            self.abandon(pendings)
            raise
        return self.finish(pendings)


class ReplicatedDatabase(ShardedDatabase):
    """The whole block resident on every device; a batch of queries runs on ONE of them, dealt round robin.  This is the
    multi-device mode of the scan orientation (few sequences, many profiles: SURVEY.md 8e "for config 3 shard profiles
    instead"): nothing has to be merged, every profile's result comes from a single device."""

    def __init__(self, block: DigitalSequenceBlock, devices: Sequence[int]):
        self.block = block
        self.devices = list(devices)
        self.chunks = [block] * len(self.devices)
        self.shards = [SequenceDatabase(block, device=d) for d in self.devices]
        self._next = 0
        self._deal = threading.Lock()

    def enqueue(self, pipelines: Sequence[Pipeline], queries) -> list:
        with self._deal:
            i = self._next
            self._next = (i + 1) % len(self.shards)
        return [pipelines[i]._search_enqueue_batch(queries, self.shards[i])]

    def finish(self, pendings: list, raw: bool = False) -> List[TopHits]:
        return Pipeline._search_finish_batch(pendings[0], raw=raw)


def hmmpress(hmms: Iterable, output) -> int:
    """Press HMMs into ``<output>.h3m`` (the core models, binary), ``<output>.h3f`` and ``<output>.h3p`` (the optimized
    profiles: MSV filter part and the rest), as the reference does (``hmmer/_hmmpress.py:29-66``: every model is
    configured for L=400, converted and written, each profile record carrying the offsets of its three parts).  Returns
    the number of models.  The ``.h3i`` SSI index of upstream's ``hmmpress`` is not written: nothing on the search path
    reads it (``HMMPressedFile`` walks the profile files in order)."""
    from .plan7 import Background
    path = os.fspath(output)
    n = 0
    bgs: dict = {}
    with open(path + ".h3m", "wb") as fm, open(path + ".h3f", "wb") as ff, open(path + ".h3p", "wb") as fp:
        for hmm in hmms:
            bg = bgs.setdefault(hmm.alphabet.type_code, Background(hmm.alphabet))
            at = (fm.tell(), ff.tell(), fp.tell())
            hmm.write(fm, binary=True)
            OptimizedProfile(hmm, bg, 400).write(ff, fp, offsets=at)
            n += 1
    return n


def hmmsearch(queries: Union[HMM, Profile, OptimizedProfile, Iterable], sequences, *, cpus: int = 0,
              callback: Optional[Callable] = None, devices: Optional[Sequence[int]] = None,
              pipeline_depth: int = 8, feeders: int = 0, finishers: int = 0, batch: int = 0,
              backend: Optional[str] = None, parallel: Optional[str] = None, builder=None, timeout: Optional[float] = None,
              chunk_bytes: Optional[int] = None, **options) -> Iterator[TopHits]:
    """Search HMMs against a sequence database; yields one ``TopHits`` per query, in query order.

    A ``SequenceFile`` of targets is walked in chunks, as the reference walks it sequence by sequence
    (``plan7.pyx:6244-6252``, ``_search_loop_file`` ``:6456``): ``chunk_bytes`` of FASTA text (default 1 GiB; a file that
    fits is one chunk) are parsed, made resident and searched by every query of a span of up to 2,048 queries, then the
    next chunk replaces them; the per-chunk hit lists of a query are merged as the reference merges target chunks
    (``TopHits.merge``: counters and ``Z`` summed, thresholds re-applied).  Neither host memory nor HBM ever hold more
    than one chunk, so the database may be larger than both.

    ``batch`` queries share one set of device launches (``p7x_search_batch_enqueue``: every kernel of the cascade
    serves all of them); 0 picks a size from the amount of work one query is (a query that fills the device for
    milliseconds runs alone, Pfam-sized models against a proteome run 64 at a time).  The reference's ``backend``,
    ``parallel`` ("queries" / "targets"), ``builder`` and ``timeout`` keywords (``_hmmsearch.py:294-436``) are accepted:
    the first two choose between host worker models that have no counterpart here (the device path always shards
    targets), ``builder`` only applies to sequence / MSA queries, which this path does not build HMMs from, and a
    search cannot time out waiting for workers.

    ``devices`` lists the HIP devices to shard the targets over; by default every device the process sees
    (:func:`default_devices`: the reference's default is the whole machine too, ``cpus=0``, ``_hmmsearch.py:384``).  ``cpus`` is accepted for
    signature compatibility and sets the number of host threads used for domain definition.  All other keyword
    arguments are forwarded to :class:`~pyhmmer_amd.plan7.Pipeline` (reference ``_hmmsearch.py:294-436``).

    Consecutive queries are overlapped the way the reference overlaps them on worker threads
    (``hmmer/_base.py:416-489``): ``feeders`` threads run the device stage (filters and parsers) of up to
    ``pipeline_depth`` batches of queries ahead while ``finishers`` threads (default: one per batch in flight, i.e.
    ``pipeline_depth``) run the host stage of the batches before them; results come back in query order.  Since round 4
    the host stage is mostly a wait for the device -- the envelope kernel, the stochastic traceback ensembles of the
    multi-domain regions and a second envelope round for their clustered envelopes all run there, a chain of small
    launches that takes 60-100 ms while the filter kernels of the following batches saturate the device -- so enough
    batches must be in flight to cover that latency: the headline stream (one 262-node profile, a million targets)
    measured 14.8-15.4 TCUPS at depth 4, 16.0 at 6 and 17.6 at 8; the many-profile stream does not depend on it
    (34.8 s with 2, 4 or 6 host stages in flight).  One default serves both.  ``pipeline_depth=0`` runs the two stages of
    every query back to back.  ``feeders=0`` (the default) lets the first batch decide: batches of several different profiles
    (a profile library: their launches are spread over the register tiles' tiers and eight streams, and leave gaps) get
    three cascades in flight, batches of one profile (a single query, or a stream of the same one: one launch that fills
    the device, and a heavy host stage) two -- measured on one MI355X, round 6: the 20,000-profile library against 500,000
    targets 23.5 TCUPS with two feeders, 25.6 with three (and +6 ... +11 % on 1/2, 1/4 and 1/8 of the targets, what a rank
    of an N-GPU run holds; four feeders: no better, and unstable on small shards); one 262-node profile against a million
    targets 18.8 with two, 16.5 with three (``profiles/r06_feeders.txt``).
    ``sequences`` may also be a :class:`~pyhmmer_amd.plan7.SequenceDatabase` already resident on one device.
    """
    if backend not in (None, "threading", "multiprocessing"):
        raise ValueError(f"invalid value for `backend`: {backend!r}")           # _base.py: the reference's own check
    if parallel not in (None, "queries", "targets"):
        raise ValueError(f"invalid value for `parallel`: {parallel!r}")
    if isinstance(queries, (HMM, Profile, OptimizedProfile)):
        queries = (queries,)
    if builder is not None or timeout is not None:
        import warnings
        warnings.warn("hmmsearch: `builder` only applies to sequence / MSA queries (not built on this path) and `timeout` "
                      "to worker processes (there are none); both are ignored", RuntimeWarning, stacklevel=2)
    ndev = _lib.lib().p7x_device_count()
    if isinstance(sequences, SequenceFile):
        if sequences.name is None:
            raise ValueError("expected named `SequenceFile` for targets")               # _hmmsearch.py:392-393
        if not sequences.digital:
            raise ValueError("target sequences file is not in digital mode")
        if ndev < 1:
            from .errors import DeviceUnavailable
            raise DeviceUnavailable("hmmsearch: no HIP device is usable and there is no CPU fallback")
        yield from _search_file(queries, sequences, chunk_bytes or (1 << 30), list(devices) if devices else default_devices(), cpus, callback,
                                pipeline_depth, feeders, batch, options, finishers=finishers)
        return
    if not isinstance(sequences, (DigitalSequenceBlock, SequenceDatabase, ShardedDatabase)):
        raise TypeError(f"Expected DigitalSequenceBlock or SequenceFile, found {type(sequences).__name__}")
    alphabet: Alphabet = sequences.shards[0].alphabet if isinstance(sequences, ShardedDatabase) else sequences.alphabet
    if ndev < 1:
        from .errors import DeviceUnavailable
        raise DeviceUnavailable("hmmsearch: no HIP device is usable and there is no CPU fallback")
    if isinstance(sequences, SequenceDatabase):
        db = ShardedDatabase.from_database(sequences)
        devs = db.devices
    elif isinstance(sequences, ShardedDatabase):
        db = sequences                              # shards resident on their devices already
        devs = db.devices
    else:
        devs = list(devices) if devices else default_devices()
        db = ShardedDatabase(sequences, devs)
    pipelines = [Pipeline(alphabet, device=d, host_threads=cpus, **options) for d in devs]
    total = None
    try:
        total = len(queries)          # type: ignore[arg-type]
    except TypeError:
        pass
    for q, hits in _run_queries(db, pipelines, queries, pipeline_depth, feeders, finishers=finishers, batch=batch):
        if callback is not None:
            callback(q, total)
        yield hits


_FILE_SPAN = 2048          # queries that share one pass over a target file


def _search_file(queries: Iterable, file: SequenceFile, chunk_bytes: int, devs: List[int], cpus: int, callback, pipeline_depth: int,
                 feeders: int, batch: int, options: dict, finishers: int = 0) -> Iterator[TopHits]:
    """hmmsearch against a target FILE: chunk after chunk resident, every query of a span over every chunk, the chunks'
    hit lists of a query merged (see hmmsearch)."""
    alphabet: Alphabet = file.alphabet
    total = None
    try:
        total = len(queries)          # type: ignore[arg-type]
    except TypeError:
        pass
    it = iter(queries)
    while True:
        span: list = []
        for q in it:
            span.append(q)
            if len(span) >= _FILE_SPAN:
                break
        if not span:
            return
        file.rewind()
        parts: List[List[TopHits]] = [[] for _ in span]
        nchunks = 0
        while True:
            block = file.read_chunk(chunk_bytes)
            if len(block) == 0 and nchunks > 0:
                break
            nchunks += 1
            db = ShardedDatabase(block, devs)           # an empty file: one empty chunk, so that every query still reports
            pipelines = [Pipeline(alphabet, device=d, host_threads=cpus, **options) for d in devs]
            for i, (_, hits) in enumerate(_run_queries(db, pipelines, span, pipeline_depth, feeders, finishers=finishers, batch=batch)):
                parts[i].append(hits)
            del db, pipelines                            # the chunk leaves HBM before the next one is read
            if len(block) == 0:
                break
        for q, hs in zip(span, parts):
            hits = hs[0] if len(hs) == 1 else hs[0].merge(*hs[1:])
            if callback is not None:
                callback(q, total)
            yield hits


_BATCH_CELLS = float(os.environ.get("P7X_BATCH_CELLS", 6e11))        # (profile, target) cells per device batch when the caller leaves the batch size open: ~25 ms of MSV (the variable: sweeps)
_BATCH_CELLS_MAX = 4 * _BATCH_CELLS
_BATCH_MIN = int(os.environ.get("P7X_BATCH_MIN", 64))     # a library search (>= 1,000 queries) against a large block: at least this many profiles per batch
                         # while the batch stays within _BATCH_CELLS_MAX: 20-profile batches of the line's workload leave ~3 % on the
                         # table (profiles/r06_feeders.txt: 25.6 TCUPS at one budget, 25.9 at two, 26.35 at four)
_BATCH_MAX = 256         # queries per batch against a large block (the cell budget usually cuts far below this)
_BATCH_LIMIT = 4096      # p7x_search_batch_enqueue takes at most this many profiles
_BATCH_SLOTS = 1 << 23   # (profile, target) score slots of one batch's workspace (~30 bytes each)


def _batch_cap(db: "ShardedDatabase") -> int:
    """Most queries a batch may hold.  Against a small block (a proteome, hmmscan's query sequences) the kernels of a
    batch are latency bound -- a few dozen launches, each as long as its longest target takes one wavefront -- so the more
    profiles share a launch set the better: round 3 measured 3.4 TCUPS with batches of 256, 4.5 with 1,024, 5.2 with 2,048
    and 5.6 with 4,096 profiles on the 20,000-profile x 2,100-sequence scan (profiles/r03_scan_sweep.txt).  The cap falls
    with the number of targets, because the batch's workspace is (profiles x targets) slots."""
    known = getattr(db, "shard_targets", None)
    ntargets = max(1, int(known)) if known is not None else max(1, max(int(_lib.lib().p7x_seqdb_ntargets(sh._handle)) for sh in db.shards))
    return int(min(_BATCH_LIMIT, max(_BATCH_MAX, _BATCH_SLOTS // ntargets)))


def _shard_residues(db: "ShardedDatabase") -> int:
    """Residues of the largest shard (what one device sees of a query)."""
    known = getattr(db, "shard_residues", None)
    if known is not None:
        return max(1, int(known))
    return max(1, max(int(_lib.lib().p7x_seqdb_nresidues(sh._handle)) for sh in db.shards))


def _auto_batch(db: "ShardedDatabase", M_hint: int = 150) -> int:
    """Queries of length ``M_hint`` per device batch: enough (profile, target) cells per launch set to amortise its
    fixed cost (a few dozen launches, one event wait, one pass of the host stage), few enough that the host stage of one
    batch does not outlast the device stage of the next.  Measured on the MI355X: 4-8 for M = 262 against 3e8 residues,
    ~32 for the Pfam-shaped library (median M 120) against 1.75e8."""
    cells = float(_shard_residues(db)) * max(1, M_hint)          # one query, one shard
    return int(max(1, min(_batch_cap(db), _BATCH_CELLS // cells)))


def _batches(queries: Iterable, size: int) -> Iterator[list]:
    cur: list = []
    for q in queries:
        cur.append(q)
        if len(cur) >= size:
            yield cur
            cur = []
    if cur:
        yield cur


def _query_length(q) -> int:
    try:
        return int(q.M)
    except Exception:
        return 0


def _run_queries(db: "ShardedDatabase", pipelines: Sequence[Pipeline], queries: Iterable, pipeline_depth: int,
                 feeders: int, window: int = 1, finishers: int = 0, batch: int = 1, reorder: int = 32, by_batch: bool = False,
                 raw: bool = False, fold: Optional[Callable] = None) -> Iterator:
    """``fold(input indices, results)``, if given, runs on the thread that finished a batch, straight after its host stage, and
    what it returns takes the results' place (the scan orientation folds a batch into the per-sequence lists there: six folds of
    4 ms on the consumer's thread, behind one another at the end of a pass, were 20 ms of 300).

    Yield ``(query, TopHits)`` for every query, in input order -- or, with ``by_batch``, ``(input indices, queries, [TopHits])``
    for every batch as soon as it is finished (the scan orientation folds a batch's results into per-sequence lists while
    the next batches are still on the device; their order is restored from the indices).  Queries travel in batches of ``batch`` (one set of
    device launches each); the two stages of consecutive batches overlap.  ``window`` > 1: every feeder queues the
    device stage of that many batches before it waits for the oldest.

    Inside a span of ``reorder`` batches the queries are sorted by model length before they are cut into batches:
    profiles of similar length share every kernel instantiation, so a batch is a few launches with many profiles each
    instead of one launch per profile.  Results are held back until every earlier query of the input has been yielded."""
    auto = batch <= 0
    res = _shard_residues(db) if auto else 0
    cap = _batch_cap(db) if auto else batch
    if auto:
        batch = cap                       # upper bound; the cut below follows the cell budget
    span = batch * max(1, reorder) if batch > 1 else 1
    hint = 0
    if auto:
        # a query source of unknown length (a generator, a file being parsed) is read a shorter way ahead: the first
        # result of a span waits for the whole span to be read
        import operator
        hint = operator.length_hint(queries)
        span = min(max(hint, 8 * _BATCH_MAX), 1 << 16) if hint > 0 else 2 * _BATCH_MAX      # a sized source is sorted as a whole (up to 65,536)
    # a LIBRARY of queries (a sized source of a thousand or more): a batch holds at least _BATCH_MIN profiles while that stays within
    # four cell budgets -- see _BATCH_MIN
    nmin = _BATCH_MIN if (auto and not by_batch and hint >= 1000) else 0
    order: list = []                      # input index of every query handed to the device, in hand-over order
    batch_members: list = []              # ... batch by batch (batch number -> input indices)
    inputs: dict = {}                     # the queries that have not been yielded yet, by input index
    it = iter(queries)
    src_error: list = []

    def sorted_batches():
        base = 0
        while True:
            chunk = []
            try:
                for q in it:
                    chunk.append(q)
                    if len(chunk) >= span:
                        break
            except BaseException as e:        # the caller's iterable failed: deliver what it produced, then the error
                src_error.append(e)
            if not chunk:
                return
            inputs.update((base + i, q) for i, q in enumerate(chunk))
            # by_batch (the scan orientation, whose results are re-ordered by index anyway): the longest models first, so that
            # the batches with the most class chains and the longest host stage run while the others are still to come and the
            # pass ends on the cheapest batch's host stage, not the dearest's
            lens = [_query_length(q) for q in chunk]          # once: a library of 20,000 profiles is sorted and cut in 5 ms, not 35
            idx = sorted(range(len(chunk)), key=lens.__getitem__, reverse=by_batch) if span > 1 else list(range(len(chunk)))
            lo = 0
            while lo < len(idx):
                if auto:                      # as many queries as the cell budget holds (short models: many, long ones: few)
                    hi, cells = lo, 0.0
                    while hi < len(idx) and hi - lo < cap:
                        cells += float(res) * max(1, lens[idx[hi]] or 150)
                        if hi > lo and cells > _BATCH_CELLS and (hi - lo >= nmin or cells > _BATCH_CELLS_MAX):
                            break
                        hi += 1
                else:
                    hi = min(len(idx), lo + batch)
                part = idx[lo:hi]
                lo = hi
                order.extend(base + i for i in part)
                batch_members.append([base + i for i in part])
                yield [chunk[i] for i in part]
            base += len(chunk)
            if src_error:
                return

    done: dict = {}
    nxt = 0
    pos = 0                               # batches' members consumed from `order`
    failure = None
    post = (lambda b, res: fold(batch_members[b], res)) if fold is not None else None
    runner = _run_batches(db, pipelines, sorted_batches(), pipeline_depth, feeders, window, finishers, raw=raw, post=post)
    for qs, hits, err in runner:
        members = order[pos:pos + len(qs)]
        pos += len(qs)
        if err is not None:
            failure = (members, err)
            break
        if by_batch:
            for i in members:
                inputs.pop(i, None)
            yield members, qs, hits
            continue
        for i, h in zip(members, hits):
            done[i] = h
        while nxt in done:
            yield inputs.pop(nxt), done.pop(nxt)
            nxt += 1
    runner.close()                        # feeders stop, queued device work is released
    if failure is not None and by_batch:
        raise failure[1]                  # the results so far were handed out batch by batch; nothing to finish in order
    if failure is not None:
        # A member of a batch failed (missing cutoffs, a device error ...).  The reference would have yielded the results
        # of every query before it first (_base.py:305-318): finish the input order one query at a time up to the failing
        # batch's last member; the error then surfaces at its own position.
        members, err = failure
        last = max(members)
        while nxt <= last:
            one = done.pop(nxt) if nxt in done else db.search(pipelines, [inputs[nxt]])[0]
            yield inputs.pop(nxt), one
            nxt += 1
        raise err           # not reproducible query by query: report it after the batch
    while nxt in done:
        yield inputs.pop(nxt), done.pop(nxt)
        nxt += 1
    if src_error:
        raise src_error[0]


_T0 = time.perf_counter()


PIPE_TRACE = False       # diagnostics: set to True for a timeline of the batches on stderr (the library reads no environment)
_LAST_STATS: dict = {}


def pipeline_stats() -> dict:
    """Where the threads of the last (finished) query pipeline of this process spent their time, in seconds summed over the
    threads of a kind: ``feeder_enqueue`` / ``feeder_device_wait`` (issuing a batch's device stage, waiting for it),
    ``feeder_slot_wait`` (all ``pipeline_depth`` batches were in flight: the feeders waited for a host stage to finish --
    the search is host-stage bound when this dominates), ``finish`` (host stages, most of it waits for the envelope and
    ensemble kernels), ``consumer_finish_wait`` (the consumer waited for the oldest host stage), ``wall``; plus the thread
    counts.  What a scaling run needs to tell a device-bound rank from a host-bound one."""
    return dict(_LAST_STATS)


def _pipe_trace(what, idx, n=None):
    # PIPE_TRACE: timeline of the batches (ms since import, thread, event, batch index) on stderr
    sys.stderr.write(f"[pipe] {1e3 * (time.perf_counter() - _T0):9.2f} {threading.current_thread().name[-10:]:>10} {what:9} {idx}{'' if n is None else f' ({n})'}\n")


def _traced_finish(db, pendings, trace, idx, stats=None, lock=None, raw=False, post=None):
    trace("finish", idx)
    t0 = time.perf_counter()
    try:
        res = db.finish(pendings, raw=True) if raw else db.finish(pendings)
        return post(idx, res) if post is not None else res
    finally:
        trace("finished", idx)
        if stats is not None:
            with lock:
                stats["finish"] += time.perf_counter() - t0


def _run_batches(db: "ShardedDatabase", pipelines: Sequence[Pipeline], queries: Iterable, pipeline_depth: int,
                 feeders: int, window: int = 1, finishers: int = 0, raw: bool = False, post: Optional[Callable] = None) -> Iterator:
    """``queries`` yields lists of queries; yields ``(list, [TopHits], None)`` in order, or ``(list, None, error)`` for
    the first batch that failed (nothing follows it)."""
    if pipeline_depth <= 0:
        for nb, q in enumerate(queries):
            try:
                if raw:
                    pendings = db.enqueue(pipelines, q)
                    try:
                        db.wait(pendings)
                    except BaseException:
                        db.abandon(pendings)
                        raise
                    res = db.finish(pendings, raw=True)
                else:
                    res = db.search(pipelines, q)
                if post is not None:
                    res = post(nb, res)
            except BaseException as e:
                yield q, None, e
                return
            yield q, res, None
        return

    # two-stage software pipeline over the batches.  Feeder threads (each with its own device stream) run the
    # device stage ahead of the host stage; results are handed over in order, at most pipeline_depth of them
    # staged or in flight at any time.
    if feeders <= 0:                  # the first batch decides (see hmmsearch): several different profiles -> three cascades in flight
        import itertools
        queries = iter(queries)
        head = list(itertools.islice(queries, 1))
        feeders = 3 if head and len({id(q) for q in head[0]}) > 1 else 2
        queries = itertools.chain(head, queries)
    nfeed = max(1, min(feeders, pipeline_depth))
    trace = _pipe_trace if PIPE_TRACE else (lambda *a: None)
    stats = {"feeder_enqueue": 0.0, "feeder_device_wait": 0.0, "feeder_slot_wait": 0.0, "finish": 0.0, "consumer_finish_wait": 0.0,
             "batches": 0}
    t_start = time.perf_counter()
    lock = threading.Lock()
    ready = threading.Condition(lock)
    slots = threading.Semaphore(pipeline_depth)
    stop = threading.Event()
    qiter = enumerate(queries)
    staged: dict = {}                 # index -> (batch, pendings, error)
    state = {"issued": 0, "exhausted": False, "live": nfeed}

    def feeder():
        queued: "deque" = deque()         # (idx, batch, pendings, error): device work queued by this thread, not yet waited for

        def hand_over_oldest():
            idx, q, pendings, err = queued.popleft()
            if err is None:
                try:
                    trace("wait", idx)
                    t0 = time.perf_counter()
                    db.wait(pendings)
                    with lock:
                        stats["feeder_device_wait"] += time.perf_counter() - t0
                    trace("waited", idx)
                except BaseException as e:          # forwarded to the caller like _base.py:305-318
                    err = e
                    db.abandon(pendings)
                    pendings = None
            with lock:
                staged[idx] = (q, pendings, err)
                ready.notify_all()

        try:
            while not stop.is_set():
                if queued:
                    if not slots.acquire(blocking=False):   # the depth budget is spent: make room by finishing our oldest
                        hand_over_oldest()
                        continue
                else:
                    t0 = time.perf_counter()
                    got = slots.acquire(timeout=0.1)
                    with lock:
                        stats["feeder_slot_wait"] += time.perf_counter() - t0
                    if not got:
                        continue
                with lock:
                    if state["exhausted"] or stop.is_set():
                        slots.release()
                        return
                    try:
                        idx, q = next(qiter)
                    except StopIteration:
                        state["exhausted"] = True
                        slots.release()
                        return
                    except BaseException as e:      # the caller's iterable failed: report it in order
                        state["exhausted"] = True
                        staged[state["issued"]] = (None, None, e)
                        state["issued"] += 1
                        ready.notify_all()
                        return
                    state["issued"] = idx + 1
                try:
                    trace("enqueue", idx, len(q))
                    t0 = time.perf_counter()
                    queued.append((idx, q, db.enqueue(pipelines, q), None))
                    with lock:
                        stats["feeder_enqueue"] += time.perf_counter() - t0
                        stats["batches"] += 1
                    trace("enqueued", idx)
                except BaseException as e:
                    queued.append((idx, q, None, e))
                if len(queued) >= window:
                    hand_over_oldest()
        finally:
            while queued:
                if stop.is_set():                   # abandoned: release what was queued
                    idx, q, pendings, err = queued.popleft()
                    db.abandon(pendings)
                else:
                    hand_over_oldest()
            with lock:
                state["live"] -= 1
                ready.notify_all()

    threads = [threading.Thread(target=feeder, name=f"p7x-hmmsearch-feeder-{i}", daemon=True) for i in range(nfeed)]
    for t in threads:
        t.start()
    nxt = 0
    # the host stage of several batches may be in flight as well (each waits for its own envelope kernel): with one
    # feeder it runs in the caller's thread, with more a small pool finishes batches concurrently, results in order
    nfin = finishers if finishers > 0 else max(nfeed, pipeline_depth)
    pool = ThreadPoolExecutor(max_workers=nfin, thread_name_prefix="p7x-hmmsearch-finish") if nfin > 1 else None
    inflight: "deque" = deque()

    def result_of(q, fut):
        try:
            return q, fut.result(), None
        except BaseException as e:
            return q, None, e

    try:
        while True:
            with lock:
                while nxt not in staged and not (state["live"] == 0 and nxt >= state["issued"]) and \
                        not (inflight and inflight[0][1].done()):
                    ready.wait(timeout=0.002 if inflight else None)
                item = staged.pop(nxt) if nxt in staged else None
                drained = item is None and state["live"] == 0 and nxt >= state["issued"]
            if item is not None:
                q, pendings, err = item
                nxt += 1
                slots.release()
                if err is not None:            # surfaces at the failing batch's position: first the results before it
                    while inflight:
                        q0, fut = inflight.popleft()
                        item0 = result_of(q0, fut)
                        yield item0
                        if item0[2] is not None:
                            return
                    if q is None:
                        raise err              # the caller's iterable failed
                    yield q, None, err
                    return
                if pool is None:
                    try:
                        res = _traced_finish(db, pendings, trace, nxt - 1, stats, lock, raw, post)
                    except BaseException as e:
                        yield q, None, e
                        return
                    yield q, res, None
                    continue
                inflight.append((q, pool.submit(_traced_finish, db, pendings, trace, nxt - 1, stats, lock, raw, post)))
            while inflight and (inflight[0][1].done() or len(inflight) >= nfin or drained):
                q, fut = inflight.popleft()
                t0 = time.perf_counter()
                item0 = result_of(q, fut)
                stats["consumer_finish_wait"] += time.perf_counter() - t0
                yield item0
                if item0[2] is not None:
                    return
            if drained and not inflight:
                break
    finally:
        stop.set()
        if pool is not None:
            for _, fut in inflight:
                try:
                    fut.result()            # results own device-side buffers: let them finish and be released
                except BaseException:
                    pass
            pool.shutdown(wait=True)
        with lock:
            state["exhausted"] = True
        for t in threads:
            slots.release()
        for t in threads:
            t.join()
        stats.update(wall=time.perf_counter() - t_start, feeders=nfeed, finishers=nfin, pipeline_depth=pipeline_depth)
        _LAST_STATS.clear()
        _LAST_STATS.update(stats)
        for _, pendings, _ in staged.values():
            db.abandon(pendings)


def hmmscan(queries, profiles, *, cpus: int = 0, callback: Optional[Callable] = None, devices: Optional[Sequence[int]] = None,
            pipeline_depth: int = 4, feeders: int = 3, window: int = 1, finishers: int = 0, batch: int = 0,
            backend: Optional[str] = None, query_block_sequences: int = 16384, query_block_residues: int = 1 << 23,
            **options) -> Iterator[TopHits]:
    """Scan query sequences against a profile database; yields one ``TopHits`` per query sequence, in query order, whose
    hits are the profiles (reference ``hmmer/_hmmscan.py:90-231``, ``Pipeline.scan_seq`` ``plan7.pyx:6534-6622``).

    ``queries``: a ``DigitalSequence``, an iterable of them, a ``DigitalSequenceBlock`` or a digital ``SequenceFile``;
    ``profiles``: an iterable of ``HMM`` / ``Profile`` / ``OptimizedProfile`` (``HMMFile``, ``HMMPressedFile`` ...).

    The device-friendly orientation is profile-major: the query sequences are packed into one resident block, every
    profile makes one pass over all of them (the same two-stage pipeline as ``hmmsearch``), and the per-profile results
    are transposed into per-sequence hit lists (``p7x_scan_collect``): reportability with the running number of models,
    E-values with ``Z`` = number of profiles, per-sequence accounting.  A query block is small next to a search
    database, so one profile's kernels are a few wavefronts running for the length of the longest query: several profiles
    are kept in flight on separate device streams to fill the device: each of the ``feeders`` threads queues the
    device stage of ``window`` batches of ``batch`` profiles before it waits for the oldest one, at most
    ``pipeline_depth`` in total.  ``batch=0`` (default) takes as many profiles per batch as the workspace allows for this
    query block (``_batch_cap``: 4,096 for a proteome-sized block): the kernels of a batch are latency bound, and the
    20,000-profile x 2,100-sequence scan went from 3.4 TCUPS with batches of 256 to 5.6 with 4,096, three feeders and a
    depth of four (``scripts/scan_sweep.py``, ``profiles/r03_scan_sweep.txt``).  The device images of a batch are laid
    out by the host workers and go up in one copy.
    """
    from .easel import DigitalSequence
    from .plan7 import _P7X_SCAN_MODELS
    import ctypes as C
    if _lib.lib().p7x_device_count() < 1:
        from .errors import DeviceUnavailable
        raise DeviceUnavailable("hmmscan: no HIP device is usable and there is no CPU fallback")
    if isinstance(queries, DigitalSequence):
        queries = (queries,)
    if isinstance(queries, SequenceFile) and not queries.digital:
        raise ValueError("query sequences file is not in digital mode")
    # The queries are taken a block at a time (the reference takes them one at a time, plan7.pyx:6680-6737): a block is
    # scanned against the whole profile database and its results are handed out before the next block is read, so a query
    # file of any size streams through, and the first results arrive after one pass over the profiles, not after all of them.
    devs = list(devices) if devices else default_devices()
    seen = 0
    # A one-shot iterable of profiles (a generator streaming a library) is walked once per query block: it is held in
    # memory only if a second block of queries really arrives -- with one block (the usual case: a proteome is a fraction of
    # a block) the profiles stream through batch by batch, as they do in the reference.
    blocks = (b for b in _query_blocks(queries, query_block_sequences, query_block_residues) if len(b) > 0)
    one_shot = not hasattr(profiles, "rewind") and not isinstance(profiles, (list, tuple))
    ahead = []
    if one_shot:
        for b in blocks:
            ahead.append(b)
            if len(ahead) == 2:
                profiles = list(profiles)
                break
    import itertools as _it
    for block in _it.chain(ahead, blocks):
        if seen and hasattr(profiles, "rewind"):
            profiles.rewind()
        seen += 1
        alphabet: Alphabet = block.alphabet
        stamps = [("start", time.perf_counter())] if PIPE_TRACE else None
        # the query block is small by construction: every device holds all of it and takes its share of the profiles
        db = ReplicatedDatabase(block, devs) if len(devs) > 1 else ShardedDatabase(block, devs)
        if stamps: stamps.append(("block resident", time.perf_counter()))
        pipelines = [Pipeline(alphabet, device=d, host_threads=cpus, **options) for d in devs]
        for pl in pipelines:
            pl._mode = _P7X_SCAN_MODELS
        n = len(block)
        shard = db.shards[0]
        lengths = (C.c_int32 * n)(*[len(s) for s in block])
        cfg = pipelines[0]._cfg()
        acc = C.c_void_p()
        st = _lib.lib().p7x_scan_accum_create(C.byref(cfg), n, shard._names, shard._accs, shard._descs, lengths, C.byref(acc))
        if st != 0:
            from .errors import status_to_exception
            raise status_to_exception(st, "p7x_scan_accum_create", _lib.last_error())
        try:
            # per-model results are folded into the per-sequence lists batch by batch, as the batches are finished, and
            # released: the scan holds a batch of them, not the library's worth, and only the last batch's fold is left
            # when the device is done.  Batches hold the profiles in order of length, not of the database: every result
            # goes in with its profile's number (the running Z of the reference's loop, plan7.pyx:6680-6737), and the
            # accumulator restores the database's order at the end.
            def fold(members, per_model):          # on the thread that finished the batch (the accumulator takes results from any thread)
                numbers = (C.c_int64 * len(members))(*members)
                st2 = _lib.lib().p7x_scan_accum_add_indexed(acc, per_model.array, numbers, len(per_model))       # plan7.HitHandles
                if st2 != 0:
                    from .errors import status_to_exception
                    raise status_to_exception(st2, "p7x_scan_accum_add_indexed", _lib.last_error())
                return len(per_model)               # the handles go with per_model, here

            for members, _, folded in _run_queries(db, pipelines, profiles, pipeline_depth, feeders, window, finishers, batch=batch,
                                                   by_batch=True, raw=True, fold=fold):
                if stamps: stamps.append((f"batch of {folded} folded", time.perf_counter()))
            if stamps: stamps.append(("pipeline closed", time.perf_counter()))
            out = (C.c_void_p * n)()
            st = _lib.lib().p7x_scan_accum_finish(acc, out)          # consumes the accumulator
            acc = C.c_void_p()
            if st != 0:
                from .errors import status_to_exception
                raise status_to_exception(st, "p7x_scan_accum_finish", _lib.last_error())
        finally:
            if acc:
                _lib.lib().p7x_scan_accum_destroy(acc)
        results = [TopHits(q, C.c_void_p(out[i])) for i, q in enumerate(block)]
        del db, pipelines                                            # the block leaves HBM before the next one is read
        if stamps:
            stamps.append(("results", time.perf_counter()))
            print("[scan] " + ", ".join(f"{what} +{1e3 * (t - stamps[0][1]):.1f}" for what, t in stamps[1:]) + " ms", file=sys.stderr, flush=True)
        for q, hits in zip(block, results):
            if callback is not None:
                callback(q, n)
            yield hits


def _query_blocks(queries, max_sequences: int, max_residues: int):
    """Blocks of query sequences for hmmscan: from a block as it is, from a file or an iterable a bounded number at a time."""
    from .easel import DigitalSequence
    if isinstance(queries, DigitalSequenceBlock):
        yield queries
        return
    if isinstance(queries, SequenceFile):
        while True:
            if max_sequences >= 16384:             # the native chunked parser (FASTA), about max_residues bytes of text at a time
                block = queries.read_chunk(max_residues)
            else:
                block = queries.read_block(sequences=max_sequences, residues=max_residues)
            if len(block) == 0:
                return
            yield block
    cur, nres = [], 0
    for s in queries:
        cur.append(s)
        nres += len(s)
        if len(cur) >= max_sequences or nres >= max_residues:
            yield DigitalSequenceBlock(cur[0].alphabet, cur)
            cur, nres = [], 0
    if cur:
        yield DigitalSequenceBlock(cur[0].alphabet, cur)


def nhmmer(queries, sequences, *, cpus: int = 0, callback: Optional[Callable] = None, devices: Optional[Sequence[int]] = None,
           backend: Optional[str] = None, builder=None, timeout: Optional[float] = None, searches_in_flight: int = 8,
           **options) -> Iterator[TopHits]:
    """Search nucleotide HMMs against long nucleotide targets; yields one ``TopHits`` per query, in query order
    (reference ``hmmer/_nhmmer.py:24-56``: one ``LongTargetsPipeline.search_hmm`` per query, the queries spread over
    worker threads, ``hmmer/_base.py:416-489``).

    A search is a device phase (the SSV scan of both strands: the whole device for tens of milliseconds) followed by a
    tail that is mostly host work (seed bookkeeping, window merging, domain definition of the surviving windows, with a
    few small device batches in between).  ``searches_in_flight`` consecutive queries run at the same time on their own
    threads, so that the scan of the next query fills the device while the tails of the previous ones run (the scans
    themselves take turns at the device); results are handed back in query order.  1 runs the queries one after the other.
    Measured on the 250 Mbp benchmark (round 5, after the window stages stopped calling hipMalloc / hipFree, which wait for
    the whole device): 0.140 s per search alone, 0.088-0.095 with two in flight, 0.078-0.084 with three, 0.074 with four
    and no less with six.  Round 6 (the host tail sums in upstream's order and is longer; the scan on binary16 cells): 0.092 /
    0.065 / 0.059 / 0.057 s per search with 2 / 4 / 6 / 8 in flight over a stream of twelve (`scripts/nh_inflight.py`): the
    default is eight.

    ``queries``: ``HMM`` / ``Profile`` / ``OptimizedProfile`` objects (one or an iterable).  Sequence and alignment
    queries of the reference go through the HMM builder first, which is outside this path: build the HMM and pass it.
    ``sequences``: a ``DigitalSequenceBlock`` or a digital ``SequenceFile`` (FASTA or GenBank).  Keyword arguments are
    those of :class:`~pyhmmer_amd.plan7.LongTargetsPipeline`."""
    if isinstance(queries, (HMM, Profile, OptimizedProfile)):
        queries = (queries,)
    if isinstance(sequences, SequenceFile):
        if not sequences.digital:
            raise ValueError("target sequences file is not in digital mode")
        sequences = sequences.read_block()
    if not isinstance(sequences, DigitalSequenceBlock):
        raise TypeError(f"Expected DigitalSequenceBlock or SequenceFile, found {type(sequences).__name__}")
    if _lib.lib().p7x_device_count() < 1:
        from .errors import DeviceUnavailable
        raise DeviceUnavailable("nhmmer: no HIP device is usable and there is no CPU fallback")
    if devices is not None and len(devices) == 0:
        raise ValueError("devices must name at least one device")
    if devices is None:
        devices = default_devices()                 # the units of a search are dealt over every device the process sees
    pipeline = LongTargetsPipeline(sequences.alphabet, device=devices[0], host_threads=cpus, **options)
    total = None
    try:
        total = len(queries)          # type: ignore[arg-type]
    except TypeError:
        pass
    def prepare(q):
        # what a search reads from and writes to the query object (the profile it scans with, hmm.max_length's replacement) is
        # settled here, on the caller's thread and in query order: the searches themselves overlap, and the same HMM object may
        # be in flight twice
        if not isinstance(q, (HMM, Profile, OptimizedProfile)):
            raise TypeError(f"Unsupported query type for `nhmmer`: {type(q).__name__} (build an HMM from it first)")
        if q.alphabet != pipeline.alphabet:
            from .errors import AlphabetMismatch
            raise AlphabetMismatch(pipeline.alphabet, q.alphabet)
        if isinstance(q, (Profile, OptimizedProfile)) and pipeline.window_length is None and (getattr(q, "max_length", None) or -1) <= 0:
            raise TypeError("Cannot use `Profile` or `OptimizedProfile` query without `max_length` set")     # plan7.pyx:7354
        return pipeline._prepare_query(q, sequences)

    def search(q, prepared=None):
        if prepared is None:
            prepared = prepare(q)
        return pipeline.search_hmm(q, sequences, devices=devices, _prepared=prepared)      # the units of one search dealt over the devices

    if searches_in_flight <= 1:
        for q in queries:
            hits = search(q)
            if callback is not None:
                callback(q, total)
            yield hits
        return
    if isinstance(sequences, DigitalSequenceBlock):
        sequences.packed()                      # the flat image (and its residency token) once, before the threads race for it
    inflight: "deque" = deque()
    with ThreadPoolExecutor(max_workers=searches_in_flight, thread_name_prefix="p7x-nhmmer") as pool:
        try:
            for q in queries:
                try:
                    fut = pool.submit(search, q, prepare(q))
                except Exception as e:              # noqa: BLE001 - surfaces at this query's position, after the results before it
                    from concurrent.futures import Future
                    fut = Future()
                    fut.set_exception(e)
                inflight.append((q, fut))
                while len(inflight) >= searches_in_flight:
                    q0, fut = inflight.popleft()
                    hits = fut.result()
                    if callback is not None:
                        callback(q0, total)
                    yield hits
            while inflight:
                q0, fut = inflight.popleft()
                hits = fut.result()
                if callback is not None:
                    callback(q0, total)
                yield hits
        finally:
            for _, fut in inflight:             # abandoned or failed: let what was started finish (it owns device buffers)
                try:
                    fut.result()
                except BaseException:           # noqa: BLE001
                    pass
