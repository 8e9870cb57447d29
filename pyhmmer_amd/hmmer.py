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

import threading
from typing import Callable, Iterable, Iterator, List, Optional, Sequence, Union

from . import _lib
from .easel import Alphabet, DigitalSequenceBlock, SequenceFile
from .plan7 import HMM, OptimizedProfile, Pipeline, Profile, SequenceDatabase, TopHits

__all__ = ["hmmsearch", "make_chunks", "ShardedDatabase"]


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


class ShardedDatabase:
    """Target block sharded by residues over several devices; each shard is a :class:`SequenceDatabase`."""

    def __init__(self, block: DigitalSequenceBlock, devices: Sequence[int]):
        self.block = block
        self.devices = list(devices)
        self.chunks = make_chunks(block, len(self.devices))
        self.shards = [SequenceDatabase(c, device=d) for c, d in zip(self.chunks, self.devices)]

    def search(self, pipelines: Sequence[Pipeline], query) -> TopHits:
        results: List[Optional[TopHits]] = [None] * len(self.shards)
        errors: List[BaseException] = []

        def work(i: int):
            try:
                results[i] = pipelines[i].search_hmm(query, self.shards[i])
            except BaseException as e:      # forwarded to the caller like _base.py:305-318
                errors.append(e)

        if len(self.shards) == 1:
            work(0)
        else:
            threads = [threading.Thread(target=work, args=(i,)) for i in range(len(self.shards))]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        if errors:
            raise errors[0]
        hits = results[0]
        return hits if len(results) == 1 else hits.merge(*results[1:])


def hmmsearch(queries: Union[HMM, Profile, OptimizedProfile, Iterable], sequences, *, cpus: int = 0,
              callback: Optional[Callable] = None, devices: Optional[Sequence[int]] = None,
              **options) -> Iterator[TopHits]:
    """Search HMMs against a sequence database; yields one ``TopHits`` per query, in query order.

    ``devices`` lists the HIP devices to shard the targets over (default: device 0).  ``cpus`` is accepted for
    signature compatibility and sets the number of host threads used for domain definition.  All other keyword
    arguments are forwarded to :class:`~pyhmmer_amd.plan7.Pipeline` (reference ``_hmmsearch.py:294-436``).
    """
    if isinstance(queries, (HMM, Profile, OptimizedProfile)):
        queries = (queries,)
    if isinstance(sequences, SequenceFile):
        if not sequences.digital:
            raise ValueError("target sequences file is not in digital mode")
        sequences = sequences.read_block()
    if not isinstance(sequences, DigitalSequenceBlock):
        raise TypeError(f"Expected DigitalSequenceBlock or SequenceFile, found {type(sequences).__name__}")
    alphabet: Alphabet = sequences.alphabet
    devs = list(devices) if devices else [0]
    ndev = _lib.lib().p7x_device_count()
    if ndev < 1:
        from .errors import DeviceUnavailable
        raise DeviceUnavailable("hmmsearch: no HIP device is usable and there is no CPU fallback")
    db = ShardedDatabase(sequences, devs)
    pipelines = [Pipeline(alphabet, device=d, host_threads=cpus, **options) for d in devs]
    total = None
    try:
        total = len(queries)          # type: ignore[arg-type]
    except TypeError:
        pass
    for q in queries:
        hits = db.search(pipelines, q)
        if callback is not None:
            callback(q, total)
        yield hits
