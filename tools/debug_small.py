import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import mixed_reads
from epik_amd import synth
from epik_amd.placer import Placer
from oracle.oracle import Oracle
tree = synth.make_tree(8, seed=1)
db = synth.make_db(tree.num_nodes, kmer_size=4, p_present=0.7, seed=5, lognormal=(1.0, 1.0))
rng = np.random.default_rng(0)
reads = mixed_reads(rng, 40, db.kmer_size, max_len=300)
data, offs = synth.pack_reads(reads)
ref = Oracle.from_synth(db).place(data, offs)
with Placer.from_synth(db) as pl:
    got = pl.place_packed(data, offs)
for i in range(len(reads)):
    ok = got[1][i] == ref[1][i] and np.array_equal(got[0][i]["branch"][:ref[1][i]], ref[0][i]["branch"][:ref[1][i]])
    print(i, len(reads[i]), "ok" if ok else "BAD", "gpu", got[1][i], [(int(r["branch"]), float(r["score"]), float(r["lwr"])) for r in got[0][i][:max(1,got[1][i])][:3]],
          "ref", ref[1][i], [(int(r["branch"]), float(r["score"]), float(r["lwr"])) for r in ref[0][i][:ref[1][i]][:3]])
