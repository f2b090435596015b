"""CPU tier: bench.py's launch contract.  `python bench.py --gpus N` outside torchrun must re-execute itself as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`, the N
ranks must rendezvous, time EXACTLY K steps between barriers, reduce the MAX over ranks and rank 0 must print ONE JSON line
with n_gpus = N.  The step here is bench.py's `launcher-selftest` mode (a no-op on gloo: no GPU in this tier)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launch_command_is_the_contract_command():
    import bench
    cmd = bench.launch_command(["--gpus", "4", "--steps", "7"], 4, 29555)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7"]
    # N = 1, or already under torchrun: no re-launch
    assert not bench.maybe_self_launch(bench.parse_args(["--gpus", "1"]), ["--gpus", "1"])
    os.environ["WORLD_SIZE"] = "2"
    try:
        assert not bench.maybe_self_launch(bench.parse_args(["--gpus", "2"]), ["--gpus", "2"])
    finally:
        del os.environ["WORLD_SIZE"]


def test_gpus_2_self_launches_two_ranks_and_prints_one_line():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "launcher-selftest", "--steps", "5", "--warmup", "2"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and line["config"]["world_size"] == 2 and line["steps"] == 5 and line["warmup"] == 2
    assert line["metric"] == "launcher-selftest" and "NOT a measurement" in line["data"]
    # a mismatch between --gpus and the torchrun world is an error, not a silent n_gpus: 1
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "launcher-selftest"], env=env2,
                        capture_output=True, text=True, timeout=120)
    assert r2.returncode != 0 and "WORLD_SIZE=1" in (r2.stderr + r2.stdout)


def test_go_probe_reports_the_toolchain_state():
    import bench
    p = bench.go_probe()
    assert set(p) == {"go", "gnark_cpu"}
    if p["go"] is None:
        assert "no Go toolchain" in p["gnark_cpu"]
    assert os.path.exists(os.path.join(ROOT, "bench", "gnark_cpu", "main.go")) and os.path.exists(os.path.join(ROOT, "tools", "gnark_dump", "main.go"))
