"""`python bench.py --gpus N` must run N ranks: the parent starts them as a child process before it
touches torch or a GPU, and refuses a world size that differs from --gpus."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env_extra=None, drop=()):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + argv, capture_output=True, text=True, env=env,
                          timeout=300)


def test_gpus_2_spawns_two_ranks():
    p = _run(["--gpus", "2", "--launch-check"], drop=("WORLD_SIZE", "RANK", "LOCAL_RANK"))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                      # rank 0 prints exactly one line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["requested"] == 2
    assert len(set(rec["rank_pids"])) == 2                # two distinct rank processes


def test_world_size_must_equal_gpus():
    p = _run(["--gpus", "4", "--launch-check"], env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert p.returncode == 2
    assert "WORLD_SIZE is 1" in p.stderr and p.stdout.strip() == ""


def test_launcher_builds_a_child_command_and_never_imports_torch():
    code = ("import sys; sys.argv=['bench.py']; import bench; "
            "cmd = bench.rank_command(8, ['--gpus','8','--steps','5'], port=1234); "
            "assert 'torch' not in sys.modules, 'the launcher imported torch'; "
            "assert cmd[1:4] == ['-m','torch.distributed.run','--nnodes=1'], cmd; "
            "assert cmd[cmd.index('--nproc-per-node')+1] == '8'; "
            "assert cmd[cmd.index('--master-addr')+1] == '127.0.0.1'; "
            "assert cmd[-4:] == ['--gpus','8','--steps','5']; print('ok')")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT,
                       timeout=60)
    assert p.returncode == 0 and p.stdout.strip() == "ok", p.stderr
