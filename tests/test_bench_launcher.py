"""bench.py's own launcher (CPU, gloo): `--gpus N` without a launcher must start N ranks itself and relay rank 0's
line; a box with fewer GPUs than ranks, or a WORLD_SIZE that contradicts --gpus, must end non-zero — never a silent
1-GPU measurement reported as N (reference launcher: deepspeed --num_gpus=4, scripts/search_sparse.sh:14)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e,
                          timeout=timeout)


def test_launcher_starts_n_ranks_and_relays_one_json_line():
    p = _run(["--gpus", "2", "--launcher-selftest"])
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out == {"launcher_selftest": True, "n_gpus": 2, "max_rank_seen": 1}


def test_more_ranks_than_gpus_is_refused():
    import torch

    n_dev = torch.cuda.device_count()
    p = _run(["--gpus", str(n_dev + 2), "--steps", "1", "--warmup", "0", "--no-c4", "--no-c5", "--no-c3", "--no-cpu"])
    assert p.returncode == 2
    assert b"HIP device(s) are visible" in p.stderr and p.stdout.strip() == b""


def test_world_size_mismatch_is_refused():
    p = _run(["--gpus", "4", "--launcher-selftest"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode == 2 and b"WORLD_SIZE=1" in p.stderr
