"""Where a scan's per-profile time goes: one thread, W profiles queued ahead; host time of enqueue / wait / finish."""
import sys, time, collections
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import load_hmms, GOLDEN
from pyhmmer_amd import easel, plan7, hmmer
with easel.SequenceFile(GOLDEN / "seqs" / "938293.PRJEB85.HG003687.faa", digital=True, alphabet=easel.Alphabet.amino()) as sf:
    block = sf.read_block()
models = []
for name in ("PF02826", "Thioesterase", "RREFam", "KR", "LuxC"):
    models += load_hmms(name)
bg = plan7.Background(models[0].alphabet)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
db = plan7.SequenceDatabase(block)
pli = plan7.Pipeline(block.alphabet)
pli._mode = plan7._P7X_SCAN_MODELS
for rnd in range(2):
    oms = [plan7.OptimizedProfile(h, bg, 400) for _ in range(10) for h in models]
    te = tw = tf = 0.0
    q = collections.deque()
    t0 = time.perf_counter()
    res = []
    for om in oms:
        a = time.perf_counter(); q.append(pli._search_enqueue(om, db)); te += time.perf_counter() - a
        if len(q) >= W:
            p = q.popleft()
            a = time.perf_counter(); pli._search_wait(p); tw += time.perf_counter() - a
            a = time.perf_counter(); res.append(pli._search_finish(p)); tf += time.perf_counter() - a
    while q:
        p = q.popleft()
        a = time.perf_counter(); pli._search_wait(p); tw += time.perf_counter() - a
        a = time.perf_counter(); res.append(pli._search_finish(p)); tf += time.perf_counter() - a
    dt = time.perf_counter() - t0
    n = len(oms)
    print(f"round {rnd} W={W}: {1e3*dt/n:.3f} ms/profile: enqueue {1e3*te/n:.3f} wait {1e3*tw/n:.3f} finish {1e3*tf/n:.3f}")
