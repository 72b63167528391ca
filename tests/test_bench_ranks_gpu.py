"""bench.py's multi-rank paths on the one device of a GPU box: the collectives on the real backend (RCCL) with a process
group of one rank, and the launch the driver uses for N = 2 rehearsed with two ranks over gloo."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["reads", "kmer-shard"])
def test_the_collectives_run_on_rccl(gpu_available, mode):
    """The multi-GPU paths of bench.py with a process group of ONE rank on the real backend (nccl = RCCL): the barrier and
    the maximum over ranks of the reads-sharded mode; the gather of the part sizes, the all-to-all of the partial lists
    with split sizes and of their index as byte views, on the communication stream, of the k-mer-space shard.  (Two ranks
    need two devices: RCCL refuses two ranks on one.  What this pins is that every collective is called with dtypes,
    shapes and streams RCCL accepts -- the exchange itself is rank 0 talking to rank 0.)"""
    assert gpu_available
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, EPIK_AMD_DIST_WORLD1="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    for var in ("EPIK_AMD_KERNEL", "EPIK_AMD_LAYOUT", "EPIK_AMD_TEAM_FRONT", "EPIK_AMD_TEAM_POOL", "EPIK_AMD_MAX_BLOCKS"):
        env.pop(var, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--no-extras",
           "--cpu-baseline-seconds", "0", "--reads-per-step", "8192", "--mode", mode]
    if mode == "kmer-shard":
        cmd += ["--leaves", "5000"]  # the team kernels: partial lists
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["config"]["collectives"] == "nccl" and line["n_gpus"] == 1 and line["value"] > 0
    assert line["config"]["world_size"] == 1 and line["config"]["ranks"][0]["pci_bus_id"]
    if mode == "kmer-shard":
        assert line["config"]["kmer_shard"]["partials"] == "lists"


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["torchrun", "self"])
@pytest.mark.parametrize("mode,world", [("reads", 2), ("kmer-shard", 2), ("reads", 4), ("kmer-shard", 4)])
def test_bench_with_several_ranks(gpu_available, mode, world, launcher):
    """bench.py as the driver launches it for N = 2 and 4 -- torch.distributed.run, one process per rank, barrier +
    maximum over ranks, one JSON line from rank 0 -- rehearsed on the one device (EPIK_AMD_BENCH_REHEARSAL=1: all ranks on
    device 0, gloo in place of RCCL, which wants a device per rank).  `self`: the command is `python bench.py --gpus N`
    with NO launcher in it and none in the environment, the shape of the driver's N = 1 command -- bench.py starts
    torch.distributed.run itself as a child process and passes its one line and exit code through."""
    assert gpu_available
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, EPIK_AMD_BENCH_REHEARSAL="1", OMP_NUM_THREADS="1")
    for var in ("EPIK_AMD_KERNEL", "EPIK_AMD_LAYOUT", "EPIK_AMD_TEAM_FRONT", "EPIK_AMD_TEAM_POOL", "EPIK_AMD_MAX_BLOCKS",
                "EPIK_AMD_DIST_WORLD1", "RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(var, None)
    cmd = [sys.executable]
    if launcher == "torchrun":
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += [os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
            "--reads-per-step", "16384", "--mode", mode]
    if mode == "kmer-shard":
        cmd += ["--leaves", "5000"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [x for x in out.stdout.strip().splitlines() if x.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["value"] > 0 and line["config"]["collectives"] == "gloo"
    assert line["scaling"] == ("weak" if mode == "reads" else "strong")
    assert "cpu_baseline" not in line or line["cpu_baseline"] is None  # N = 1 only
    # the line says by itself how many ranks took part, on which devices, and what each of them measured
    config = line["config"]
    assert config["world_size"] == world and [r["rank"] for r in config["ranks"]] == list(range(world))
    for r in config["ranks"]:
        assert r["device"] == 0 and r["ms_per_step"] > 0 and r["host"]
        assert r["pci_bus_id"] is None or ":" in r["pci_bus_id"]
        if mode == "kmer-shard":
            assert r["sent_bytes_per_step"] > 0
    assert line["build"]["hipcc"] and line["build"]["kernel_source_hash"]
