"""Worker of tests/test_dist_cpu.py: run under torch.distributed.run with the gloo
backend.  Each rank places its shard of a seeded read batch -- on CPU the oracle stands
in for the kernel (test infrastructure), with EPIK_AMD_DIST_GPU=1 the HIP placer on
device 0 -- and rank 0 checks the gathered rows against a single-process run."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from epik_amd import dist as edist, synth  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def main():
    rank, _, world = edist.env_rank_world()
    dist = edist.init_process_group("gloo")
    assert dist is not None and dist.get_world_size() == world
    tree = synth.make_tree(8, seed=1)
    db = synth.make_db(tree.num_nodes, kmer_size=4, p_present=0.7, seed=5, lognormal=(1.0, 1.0))
    data, offs = synth.make_reads(1001, 37, seed=3)      # not divisible by the world size
    oracle = Oracle.from_synth(db)
    if os.environ.get("EPIK_AMD_DIST_GPU") == "1":
        from epik_amd.placer import Placer
        placer = Placer.from_synth(db, device=0)
        place_fn = placer.place_packed
    else:
        place_fn = lambda s, o: oracle.place(s, o, num_threads=1)  # noqa: E731
    got = edist.place_sharded(place_fn, data, offs, dist, gather_to=0)
    b, e = edist.shard_bounds(1001, rank, world)
    slowest = edist.max_over_ranks(float(rank + 1), dist)
    assert slowest == float(world)
    dist.barrier()
    if rank == 0:
        ref = oracle.place(data, offs, num_threads=1)
        assert got[1].shape == ref[1].shape and np.array_equal(got[1], ref[1])
        valid = np.arange(ref[0].shape[1])[None, :] < ref[1][:, None]
        assert np.array_equal(got[0]["branch"][valid], ref[0]["branch"][valid])
        assert np.array_equal(got[0]["score"][valid].view(np.uint32), ref[0]["score"][valid].view(np.uint32))
        assert np.abs(got[0]["lwr"][valid] - ref[0]["lwr"][valid]).max() <= 1e-5
        print(f"dist ok: world={world} shard0=[{b},{e}) rows={int(valid.sum())}", flush=True)
    else:
        assert got is None
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
