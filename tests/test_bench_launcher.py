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


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_ranks_started_by_torch_distributed_run_get_the_ipc_default():
    """The driver starts the ranks with torch.distributed.run itself, without this script's
    launcher: the rank must still run with HSA_ENABLE_IPC_MODE_LEGACY=0 (set before torch is
    imported) and a bounded BLAS pool."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HSA_ENABLE_IPC_MODE_LEGACY",
                        "OMP_NUM_THREADS")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), BENCH,
           "--gpus", "2", "--launch-check"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert rec["n_gpus"] == 2
    assert rec["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert int(rec["env"]["OMP_NUM_THREADS"]) >= 1


def test_failed_comm_records_give_a_nonzero_status_and_keep_the_line():
    for how, extra in (("raise", []), ("hang", ["--comm-timeout", "2"])):
        p = _run(["--gpus", "2", "--launch-check", "--inject-comm-failure", how] + extra,
                 drop=("WORLD_SIZE", "RANK", "LOCAL_RANK"))
        assert p.returncode != 0, (how, p.stdout, p.stderr[-1500:])
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, (how, p.stdout)
        assert "error" in json.loads(lines[0])["comm"], how
