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


def test_gpus_8_rendezvous_counts_every_rank():
    """The driver's 8-GPU launch shape (one process per GPU): all eight ranks join the group on the loopback address and
    are counted (no GPU: they stop after the rendezvous)."""
    r = _run(["--gpus", "8", "--steps", "1", "--warmup", "0"], {})
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 8 and out["ranks_seen"] == 8


def test_pmc_traffic_json_is_derived_from_the_committed_profiles():
    """bench.py reports HBM traffic from profiles/pmc_traffic.json; that file must equal what tools/pmc_constants.py derives
    from the newest committed rNN_pmc_fetch_size.txt / rNN_pmc_write_size.txt, and bench.py must carry no byte constants of
    its own for it."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_constants

    committed = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert committed == pmc_constants.build()
    for src in committed["source"]:
        assert os.path.exists(os.path.join(ROOT, src))
    text = open(os.path.join(ROOT, "bench.py")).read()
    assert "PMC_MIB_BY_VARIANT" not in text and "PMC_TRAFFIC_F16X3" not in text
    v = committed["variants"]
    # (the variants a training STEP launches at T rows; the bare projection only exists in bench.py's stand-alone
    # micro-timing, which the PMC passes skip with --no-micro)
    # EVERY variant a headline step times must have a constant, or `roofline.traffic` of the driver's line is null (VERDICT
    # r04 weak 3a: the `addend` variant was timed without one); the GPU half of this check - the labels a real step
    # produces - is tests/test_gpu_cmodel.py::test_every_timed_projection_variant_has_a_pmc_constant
    assert set(pmc_constants.STEP_VARIANTS) <= set(v), (pmc_constants.STEP_VARIANTS, sorted(v))
    import re

    m = re.search(r"rows_moved = \{([^}]*)\}", text)
    bench_labels = set(re.findall(r'"(\w+)":', m.group(1)))
    assert bench_labels == set(pmc_constants.VARIANT_KERNELS) == set(pmc_constants.ROWS_MOVED), bench_labels
    for name, e in v.items():  # no wasted traffic: within 1.15x of the algorithmic rows of each variant
        rows = pmc_constants.ROWS_MOVED[name]
        assert 0.95 < e["bytes_per_launch"] / (rows * committed["triplets"] * 1024) < 1.15, (name, e)
