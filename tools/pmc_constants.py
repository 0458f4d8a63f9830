"""profiles/rNN_pmc_fetch_size.txt + rNN_pmc_write_size.txt  ->  profiles/pmc_traffic.json

bench.py cannot sample PMC counters from inside its own process, so the HBM traffic it reports (``roofline.traffic``,
``step_roofline.pmc_GB_per_step``) is carried over from rocprofv3 ``--pmc FETCH_SIZE`` / ``--pmc WRITE_SIZE`` passes over
bench.py itself (separate passes, ``tools/rocpd_pmc.py <db> 0`` summaries committed under profiles/).  This script is the
ONLY place those numbers are turned into what bench.py reads; tests/test_host_logic.py re-runs it and asserts that the
committed JSON equals the committed profile, so the constants cannot go stale silently.

    python tools/pmc_constants.py            # regenerate profiles/pmc_traffic.json from the newest round's profiles
    python tools/pmc_constants.py --check    # exit 1 if the committed JSON differs

gfx950 correction (MI355X_MICROARCH.md): FETCH_SIZE counts half of the bytes of 16 B/lane streaming reads -> x2.
"""
from __future__ import annotations

import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILES = os.path.join(ROOT, "profiles")
OUT = os.path.join(PROFILES, "pmc_traffic.json")

# epilogue variant of the T-row f16x3 NT projection -> kernel names that implement it (persistent and one-tile forms)
VARIANT_KERNELS = {
    "plain": ["gemm_nt_f16p_kernel<false>"],
    # input gradient + residual addend WITHOUT the BatchNorm-backward sums (since round 4: the first line-graph convolution's
    # edge input gradient - the fused angle embedding keeps no pre-activation to reduce against)
    "addend": ["gemm_nt_x6_kernel<true, true, 2, false>", "gemm_nt_f16p_kernel<true>"],
    "gather": ["gemm_nt_f16p_gather_kernel<true>", "gemm_nt_f16p_gather_kernel<false>", "gemm_nt_x6_gather_kernel<2, true>"],
    "stats": ["gemm_nt_f16p_stats_kernel", "gemm_nt_x6_stats_kernel<2>"],
    "bnred": ["gemm_nt_f16p_bnred_kernel<false>", "gemm_nt_x6_bnred_kernel<false, 2>"],
    "bnred_addend": ["gemm_nt_f16p_bnred_kernel<true>", "gemm_nt_x6_bnred_kernel<true, 2>"],
    # csrc/gemm_dw.hip <HAS_ADD, BNRED>: the input gradient and the weight gradient of a T-row projection in one pass over g_m
    "dw": ["gemm_dw_kernel<false, false>"],
    "dw_addend": ["gemm_dw_kernel<true, false>"],
    "dw_bnred": ["gemm_dw_kernel<false, true>"],
    "dw_bnred_addend": ["gemm_dw_kernel<true, true>"],
}
# what ONE training step of the headline workload launches at T rows (bench.py `roofline.in_step` labels): every one of them
# must have a PMC constant, or bench.py's `roofline.traffic` is null (tests/test_bench_launch.py, tests/test_gpu_cmodel.py)
STEP_VARIANTS = ("gather", "dw_bnred", "dw_bnred_addend", "dw_addend")
# algorithmic H-wide fp32 rows moved per output row, by variant (bench.py uses the same table; gather: + 2 E/T for its tables)
ROWS_MOVED = {"plain": 2, "addend": 3, "gather": 2.15, "stats": 2, "bnred": 3, "bnred_addend": 4,
              "dw": 3, "dw_addend": 4, "dw_bnred": 4, "dw_bnred_addend": 5}
GATHER_LAUNCHES_PER_STEP = 4  # one per line-graph convolution of the default 4-layer model: fixes steps-per-profile
T_ROW_MIN_WRITE_MIB = 600.0  # a T = 676 200 x 256 fp32 output is 660 MiB


def parse(path):
    rows = []
    for line in open(path):
        m = re.match(r"\s*(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*) grid=(\d+)\s*$", line)
        if m:
            rows.append({"counter": m.group(1), "calls": int(m.group(2)), "avg_MiB": float(m.group(3)),
                         "max_MiB": float(m.group(4)), "avg_us": float(m.group(5)), "kernel": m.group(6).strip(),
                         "grid": int(m.group(7))})
    return rows


def _short(name):
    m = re.search(r"::(\w+(?:<[^>]*>)?)\(", name)
    return m.group(1) if m else name


def newest_round():
    rounds = sorted(int(re.search(r"r(\d+)_pmc_fetch_size", p).group(1)) for p in glob.glob(os.path.join(PROFILES, "r*_pmc_fetch_size.txt"))
                    if re.search(r"r(\d+)_pmc_fetch_size\.txt$", p) and os.path.exists(p.replace("fetch", "write")))
    if not rounds:
        raise SystemExit("no profiles/rNN_pmc_fetch_size.txt + rNN_pmc_write_size.txt pair")
    return rounds[-1]


# other workloads of BASELINE.json benched beside the headline (bench.py `other_configs`): name -> (triplets, model); their PMC
# passes are profiles/rNN_pmc_<name>_fetch_size.txt / _write_size.txt (tools/gpu/r6_pmc.sh with BARGS / SUFFIX)
WORKLOADS = {"cfg3_ff": (561792, "alignn_ff"), "cfg4_mol": (909126, "alignn")}


def build_one(ff, wf, triplets):
    fetch, write = parse(ff), parse(wf)
    min_write = T_ROW_MIN_WRITE_MIB * triplets / 676200.0

    def find(rows, names):
        # T-row launches of the kernel: the row of that kernel with the largest per-launch maximum
        best = None
        for r in rows:
            if any(n in r["kernel"] for n in names) and (best is None or r["max_MiB"] > best["max_MiB"]):
                best = r
        return best

    variants = {}
    for v, names in VARIANT_KERNELS.items():
        f, w = find(fetch, names), find(write, names)
        if f is None or w is None or w["max_MiB"] < min_write:
            continue
        # the T-row launches of a kernel that also runs at E rows are its largest ones -> per-launch maximum; for a kernel
        # only launched at T rows max ~= avg
        variants[v] = {"fetch_MiB_raw": f["max_MiB"], "write_MiB": w["max_MiB"], "kernel": _short(f["kernel"]),
                       "bytes_per_launch": round((2 * f["max_MiB"] + w["max_MiB"]) * 1048576)}
    g = find(fetch, VARIANT_KERNELS["gather"])
    steps = g["calls"] // GATHER_LAUNCHES_PER_STEP if g else None
    total = None
    if steps:
        tot = sum(2 * r["avg_MiB"] * r["calls"] for r in fetch) + sum(r["avg_MiB"] * r["calls"] for r in write)
        total = tot * 1048576 / steps
    return {"source": [os.path.relpath(ff, ROOT), os.path.relpath(wf, ROOT)], "triplets": triplets, "steps_profiled": steps,
            "variants": variants, "pmc_bytes_per_step": None if total is None else round(total)}


def build(rnd=None):
    rnd = newest_round() if rnd is None else rnd
    ff = os.path.join(PROFILES, f"r{rnd:02d}_pmc_fetch_size.txt")
    wf = os.path.join(PROFILES, f"r{rnd:02d}_pmc_write_size.txt")
    out = build_one(ff, wf, 676200)
    out.update({
        "round": rnd,
        "fetch_correction": "FETCH_SIZE x2 (gfx950, 16 B/lane streaming reads; MI355X_MICROARCH.md)",
        "pmc_bytes_per_step_covers": "every launch listed in the two summaries (those whose largest launch moved >= the "
                                     "summary's min_MiB cut-off; see the header of the .txt files)",
    })
    workloads = {}
    for name, (trip, model) in WORKLOADS.items():
        f2 = os.path.join(PROFILES, f"r{rnd:02d}_pmc_{name}_fetch_size.txt")
        w2 = os.path.join(PROFILES, f"r{rnd:02d}_pmc_{name}_write_size.txt")
        if os.path.exists(f2) and os.path.exists(w2):
            workloads[name] = dict(build_one(f2, w2, trip), model=model)
    out["workloads"] = workloads
    return out


def main():
    new = build()
    if "--check" in sys.argv:
        old = json.load(open(OUT))
        if old != new:
            print("profiles/pmc_traffic.json is stale against", new["source"])
            sys.exit(1)
        print("pmc_traffic.json matches", new["source"])
        return
    with open(OUT, "w") as f:
        json.dump(new, f, indent=1, sort_keys=True)
        f.write("\n")
    print(json.dumps(new, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
