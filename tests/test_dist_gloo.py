"""The N>1 path on CPU: two `gloo` ranks each search their residue-balanced shard of the targets, ship their
serialised TopHits to rank 0 (all_gather_object -- results only, no data-path collective), rank 0 merges and
re-thresholds.  The merged list must equal the single-process search field by field
(reference tests/test_plan7/test_tophits.py:191-224; bench.py uses the same gather + merge for --gpus N).

No GPU here: each rank's device half is stood in for by the oracle (tests/host_pipeline.py); the sharding, the
serialisation, the gather and p7x_tophits_merge are the product's."""
import os
import socket
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import pickle
    import torch.distributed as dist
    import host_pipeline
    import oracle_lib
    from conftest import GOLDEN, load_hmms
    from pyhmmer_amd import easel, hmmer, plan7
    dist.init_process_group("gloo", rank=rank, world_size=world)
    with easel.SequenceFile(GOLDEN / "seqs" / "938293.PRJEB85.HG003687.faa", digital=True, alphabet=easel.Alphabet.amino()) as sf:
        block = sf.read_block()
    hmm = load_hmms("PF02826")[0]
    shard = hmmer.make_chunks(block, world)[rank]
    hits = host_pipeline.host_search(oracle_lib, hmm, shard)
    blobs = [None] * world
    dist.all_gather_object(blobs, hits.to_bytes())
    if rank == 0:
        merged = plan7.TopHits.from_bytes(blobs[0])
        for b in blobs[1:]:
            merged = merged.merge(plan7.TopHits.from_bytes(b))
        whole = host_pipeline.host_search(oracle_lib, hmm, block)
        def dump(th):
            return [(h.name, h.score, h.pre_score, h.sum_score, h.evalue, h.reported, h.included,
                     [(d.env_from, d.env_to, d.score, d.c_evalue, d.i_evalue, d.reported, d.included,
                       d.alignment.target_sequence) for d in h.domains]) for h in th]
        res = dict(merged=dump(merged), whole=dump(whole), Z=(merged.Z, whole.Z), domZ=(merged.domZ, whole.domZ),
                   counts=(merged.stage_counts, whole.stage_counts),
                   residues=(merged.searched_residues, whole.searched_residues), shard_sizes=[len(shard)])
        with open(out_path, "wb") as f:
            pickle.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_gather_merge_equals_whole(tmp_path):
    import pickle
    import torch.multiprocessing as mp
    out = tmp_path / "res.pkl"
    mp.spawn(_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    res = pickle.load(open(out, "rb"))
    assert res["Z"] == (2100.0, 2100.0) and res["domZ"][0] == res["domZ"][1]
    assert res["counts"][0] == res["counts"][1] and res["residues"][0] == res["residues"][1] == 682583
    assert len(res["merged"]) == len(res["whole"]) == 22
    assert res["merged"] == res["whole"]


def _worker8(rank, world, port, out_path):
    """world_size 8 with an uneven split: one rank has no targets at all, one a single sequence; two queries per rank, one
    of them with model-specific (gathering) cutoffs, whose per-hit flags a merge must leave alone (plan7.pyx:9257-9273)."""
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import pickle
    import torch.distributed as dist
    import host_pipeline
    import oracle_lib
    from conftest import GOLDEN, load_hmms
    from pyhmmer_amd import easel, plan7
    dist.init_process_group("gloo", rank=rank, world_size=world)
    with easel.SequenceFile(GOLDEN / "seqs" / "938293.PRJEB85.HG003687.faa", digital=True, alphabet=easel.Alphabet.amino()) as sf:
        block = sf.read_block()
    cuts = [0, 400, 800, 800, 1300, 1301, 1700, 1900, 2100]          # rank 2: empty shard, rank 4: one sequence
    shard = block[cuts[rank]:cuts[rank + 1]]
    hmm = load_hmms("PF02826")[0]
    abc = hmm.alphabet
    plis = [lambda: plan7.Pipeline(abc), lambda: plan7.Pipeline(abc, bit_cutoffs="gathering"), lambda: plan7.Pipeline(abc, Z=5000.0, domZ=33.0)]
    mine = [host_pipeline.host_search(oracle_lib, hmm, shard, mk()).to_bytes() for mk in plis]
    blobs = [None] * world
    dist.all_gather_object(blobs, mine)
    if rank == 0:
        merged = plan7.TopHits.merge_many(blobs, threads=4)
        # the same through pairwise TopHits.merge
        pair = []
        for q in range(len(plis)):
            th = plan7.TopHits.from_bytes(blobs[0][q])
            pair.append(th.merge(*[plan7.TopHits.from_bytes(blobs[r][q]) for r in range(1, world)]))
        whole = [host_pipeline.host_search(oracle_lib, hmm, block, mk()) for mk in plis]
        def dump(th):
            return dict(hits=[(h.name, h.score, h.pre_score, h.sum_score, h.evalue, h.reported, h.included,
                               [(d.env_from, d.env_to, d.score, d.c_evalue, d.i_evalue, d.reported, d.included,
                                 d.alignment.target_sequence) for d in h.domains]) for h in th],
                        Z=th.Z, domZ=th.domZ, counts=th.stage_counts, residues=th.searched_residues, nseq=th.searched_sequences,
                        nrep=len(th.reported), ninc=len(th.included))
        res = dict(merged=[dump(t) for t in merged], pair=[dump(t) for t in pair], whole=[dump(t) for t in whole],
                   sizes=[cuts[r + 1] - cuts[r] for r in range(world)])
        with open(out_path, "wb") as f:
            pickle.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_with_an_empty_and_a_one_sequence_shard(tmp_path):
    import pickle
    import torch.multiprocessing as mp
    out = tmp_path / "res8.pkl"
    mp.spawn(_worker8, args=(8, _free_port(), str(out)), nprocs=8, join=True)
    res = pickle.load(open(out, "rb"))
    assert res["sizes"][2] == 0 and res["sizes"][4] == 1 and sum(res["sizes"]) == 2100
    for q, (m, p, w) in enumerate(zip(res["merged"], res["pair"], res["whole"])):
        assert m == p, q                                  # one native call == pairwise merges
        assert m == w, q                                  # == the unsharded search, field by field
        assert m["nseq"] == 2100 and m["residues"] == 682583
    assert res["whole"][0]["Z"] == 2100.0 and res["whole"][2]["Z"] == 5000.0 and res["whole"][2]["domZ"] == 33.0
    assert len(res["whole"][0]["hits"]) == 22 and 0 < res["whole"][1]["nrep"] <= 22


def test_merge_many_rejects_mismatched_shards(models, oracle, proteome):
    import host_pipeline
    from pyhmmer_amd import plan7
    a = host_pipeline.host_search(oracle, models["PF02826"][0], proteome[:300]).to_bytes()
    b = host_pipeline.host_search(oracle, models["Thioesterase"][0], proteome[300:600]).to_bytes()
    with pytest.raises(ValueError):
        plan7.TopHits.merge_many([[a], [b]])              # different queries
    with pytest.raises(ValueError):
        plan7.TopHits.merge_many([[a], [b"junk"]])
    with pytest.raises(ValueError):
        plan7.TopHits.merge_many([[None], [b""]])         # no shard reported anything
    one = plan7.TopHits.merge_many([[a], [None]])[0]      # an idle shard is skipped
    assert len(one) == len(plan7.TopHits.from_bytes(a))


def test_tophits_bytes_roundtrip(models, oracle, proteome):
    import pickle
    import host_pipeline
    from pyhmmer_amd import plan7
    hmm = models["PF02826"][0]
    hits = host_pipeline.host_search(oracle, hmm, proteome[:700])
    again = plan7.TopHits.from_bytes(hits.to_bytes())
    assert [(h.name, h.score, h.evalue, len(h.domains)) for h in again] == [(h.name, h.score, h.evalue, len(h.domains)) for h in hits]
    assert again.Z == hits.Z and again.stage_counts == hits.stage_counts
    viap = pickle.loads(pickle.dumps(hits))
    assert [h.name for h in viap] == [h.name for h in hits]
    with pytest.raises(ValueError):
        plan7.TopHits.from_bytes(b"garbage")


def test_tophits_bytes_keep_order_and_flags(models, oracle, proteome):
    """Pickling keeps what the user did to the list (reference plan7.pyx:8394-8572): the seqidx sort order and the
    reported / included / dropped flags are not recomputed on load; a corrupt hit count is an error, not an abort."""
    import struct
    import host_pipeline
    from pyhmmer_amd import plan7
    hmm = models["PF02826"][0]
    hits = host_pipeline.host_search(oracle, hmm, proteome[:700])
    assert len(hits) >= 3
    hits.sort(by="seqidx")
    hits[0].dropped = True
    hits[1].included = False
    blob = hits.to_bytes()
    again = plan7.TopHits.from_bytes(blob)
    assert again.is_sorted(by="seqidx") and [h.name for h in again] == [h.name for h in hits]
    assert again[0].dropped and not again[0].included and not again[1].included
    assert [(h.reported, h.included) for h in again] == [(h.reported, h.included) for h in hits]
    assert len(again.reported) == len(hits.reported) and len(again.included) == len(hits.included)
    # the 64-bit hit count sits right before the first hit's name: find it and overwrite it with an absurd value
    name = hits[0].name if isinstance(hits[0].name, bytes) else hits[0].name.encode()
    key = struct.pack("<Q", len(hits))
    at = blob.rfind(key, 0, blob.find(name))
    assert at > 0
    with pytest.raises(ValueError):
        plan7.TopHits.from_bytes(blob[:at] + struct.pack("<Q", 1 << 60) + blob[at + 8:])
    with pytest.raises(ValueError):
        plan7.TopHits.from_bytes(blob[:len(blob) // 2])


def test_block_mutation_invalidates_the_packed_copy():
    from pyhmmer_amd import easel
    abc = easel.Alphabet.amino()
    mk = lambda name, text: easel.TextSequence(name=name, sequence=text).digitize(abc)
    block = easel.DigitalSequenceBlock(abc, [mk("a", "ACDEFG"), mk("b", "HIKLMN")])
    v0, pk0 = block._version, block.packed()
    block[1] = mk("c", "PQRSTV")                       # same length, different residues
    assert block._version != v0
    pk1 = block.packed()
    assert pk1 is not pk0 and bytes(pk1.dsq) != bytes(pk0.dsq)
    block.insert(0, mk("d", "WY"))
    assert block.packed().n == 3 and block.pop().name in (b"c", "c")
    assert block.packed().n == 2
