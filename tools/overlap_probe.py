"""Does the front kernel (VALU-bound) of one half of a batch hide under the streaming kernel (memory-bound) of the other?
Two handles on one device, two streams; the second half's launch is held back by a spin kernel of about one front kernel."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from epik_amd import synth
from epik_amd.placer import Placer

leaves = int(os.environ.get("LEAVES", 1250))
n = 1_000_000
tree = synth.make_tree(leaves, seed=42)
db = synth.make_db(tree.num_nodes, kmer_size=10, seed=43)
data, offs = synth.make_reads(n, 150, seed=44)
dev = torch.device("cuda", 0)
d_seqs = torch.from_numpy(data).to(dev)
d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
h = n // 2
d_offs2 = (d_offs[h:] - d_offs[h]).contiguous()
d_seqs2 = d_seqs[int(offs[h]):].contiguous()
keep = 7
rows = torch.zeros(n * keep * 2, dtype=torch.float64, device=dev)
nrows = torch.zeros(n, dtype=torch.int32, device=dev)
A, B = Placer.from_synth(db), Placer.from_synth(db)
A.choose_counts(150); B.choose_counts(150)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

def whole():
    A.place_device(d_seqs.data_ptr(), d_offs.data_ptr(), n, rows.data_ptr(), nrows.data_ptr(), 0, s1.cuda_stream)

def halves(delay_cycles):
    A.place_device(d_seqs.data_ptr(), d_offs.data_ptr(), h, rows.data_ptr(), nrows.data_ptr(), 0, s1.cuda_stream)
    if delay_cycles:
        with torch.cuda.stream(s2):
            torch.cuda._sleep(int(delay_cycles))
    B.place_device(d_seqs2.data_ptr(), d_offs2.data_ptr(), n - h, rows.data_ptr() + 16 * keep * h, nrows.data_ptr() + 4 * h, 0, s2.cuda_stream)

def timed(fn, reps=7):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
    return min(out), float(np.median(out))

whole(); halves(0); torch.cuda.synchronize()
ref = nrows.clone()
print("N =", tree.num_nodes)
print("one launch of 1 M reads          min %.3f median %.3f ms" % timed(whole))
print("two halves, two streams, at once min %.3f median %.3f ms" % timed(lambda: halves(0)))
for us in (200, 400, 600, 800, 1000, 1400):
    cyc = us * 100  # (the spin kernel counts a 100 MHz clock)
    print("second half %4d us late         min %.3f median %.3f ms" % ((us,) + timed(lambda: halves(cyc))))
assert torch.equal(ref, nrows)
