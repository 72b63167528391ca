"""`python bench.py --gpus N` with no launcher around it (the shape of the driver's N = 1 command) starts
torch.distributed.run itself -- as a CHILD process, before anything has touched HIP -- and passes the child's exit code
through.  Without a GPU the ranks stop at "needs a HIP device": what this pins on the CPU is the launch itself."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_without_a_launcher_starts_one_as_a_child_process():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""  # the same outcome on a GPU box: no rank may find a device
    env["CUDA_VISIBLE_DEVICES"] = ""
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--reads-per-step", "1000"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert "as a child process" in run.stderr
    # both ranks were started (each says why it stops) and the launcher's failure comes back as this process's own
    assert run.stderr.count("needs a HIP device") >= 1 and run.returncode != 0
    assert "must be launched with" not in run.stderr + run.stdout


def test_the_self_launch_never_replaces_the_process():
    """A process that has initialised the GPU must not exec (that takes the host down on this pool); bench.py may not
    call os.exec* at all."""
    import re
    text = open(os.path.join(ROOT, "bench.py")).read()
    assert not re.search(r"os\.exec|os\.spawn.*P_OVERLAY|execv", text)


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=120, env=env, cwd=ROOT)
    assert run.returncode != 0 and "does not match --gpus" in run.stderr
