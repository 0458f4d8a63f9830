"""bench.py --gpus N must really start N ranks when the driver calls it without a launcher (no GPU needed: the
ranks stop after the rendezvous, ALIGNN_BENCH_RENDEZVOUS_ONLY=1)."""

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env):
    env = dict(os.environ, ALIGNN_BENCH_BACKEND="gloo", ALIGNN_BENCH_RENDEZVOUS_ONLY="1", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=300)


def test_gpus_2_without_launcher_spawns_two_ranks():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2


def test_gpus_flag_must_match_the_launcher():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", ALIGNN_BENCH_RENDEZVOUS_ONLY="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
