"""Scan the gfx950 ISA of every source of libalignn_hip.so for the packed-fp32 instruction form that tools/pk_f32_repro2.hip shows
to return wrong values on MI355X:

    v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 whose op_sel selects the HIGH half of src1 for the LOW result  (op_sel:[_,1,...])

Beside MFMA waves of another workgroup on the same SIMD the swizzled operand reads as +0.0 in lanes 48-63.  Every other packed
form (no op_sel, op_sel on src0 / src2, op_sel_hi, neg) measured clean.  The library is built with the target feature
``packed-fp32-ops`` switched off, so the expected count is ZERO for every file; ``--flags`` scans another build for comparison.

usage: python tools/isa_pk_scan.py [--default-flags]      (prints one line per source; exit code 1 if any faulting form is found)"""
from __future__ import annotations

import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alignn_amd import build as B  # noqa: E402

PK = re.compile(r"^\s*(v_pk_(?:fma|mul|add)_f32)\b(.*)$")
OPSEL = re.compile(r"op_sel:\[([01]),([01])(?:,([01]))?\]")
KERNEL = re.compile(r"^([A-Za-z_][\w$.]*):\s*(?:;.*)?$")


def scan_source(src: str, extra: list[str]) -> dict:
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + B.FLAGS + extra + ["-S", "--cuda-device-only", os.path.join(B.CSRC, src), "-o", "-"]
    out = subprocess.run(cmd, check=True, capture_output=True, text=True).stdout
    total = bad = 0
    where: dict[str, int] = {}
    cur = "?"
    for line in out.splitlines():
        m = KERNEL.match(line)
        if m and not m.group(1).startswith((".", "$")):
            cur = m.group(1)
        m = PK.match(line)
        if not m:
            continue
        total += 1
        o = OPSEL.search(m.group(2))
        if o and o.group(2) == "1":
            bad += 1
            where[cur] = where.get(cur, 0) + 1
    return {"src": src, "packed": total, "faulting": bad, "where": where}


def main(argv):
    default_flags = "--default-flags" in argv
    def extra(src):
        if default_flags:
            return []
        return B.DEVICE_FLAGS + B.EXTRA_FLAGS.get(src, [])
    with ThreadPoolExecutor(max_workers=8) as pool:
        res = list(pool.map(lambda s: scan_source(s, extra(s)), B.SOURCES))
    print(f"# v_pk_{{fma,mul,add}}_f32 per source ({'hipcc defaults' if default_flags else 'library flags'}): all packed / with op_sel src1 = high half")
    n_bad = 0
    for r in res:
        n_bad += r["faulting"]
        print(f"{r['src']:16s} {r['packed']:6d} {r['faulting']:6d}")
        for k, v in sorted(r["where"].items(), key=lambda kv: -kv[1])[:6]:
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k
            print(f"        {v:4d}  {name[:150]}")
    print(f"# total faulting-form instructions: {n_bad}")
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
