"""Instruction mix per kernel (and per basic block with --blocks NAME) of a hipcc -S listing:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only file.hip -o /tmp/x.s ; python tools/isa_mix.py /tmp/x.s"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--blocks" else None


def kind(x):
    for pre, k in (("v_mfma", "mfma"), ("global_load", "gload"), ("global_store", "gstore"), ("ds_", "ds"), ("scratch_", "scratch"),
                   ("s_waitcnt", "wait"), ("s_barrier", "barrier"), ("v_exp", "trans"), ("v_rcp", "trans"), ("v_cvt", "cvt"),
                   ("v_cndmask", "cndmask"), ("v_cmp", "cmp"), ("s_load", "sload"), ("v_", "valu"), ("s_", "salu")):
        if x.startswith(pre):
            return k
    return "other"


parts = re.split(r"\n(_Z[^\n:]*):[^\n]*\n", s)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split(".Lfunc_end")[0]
    if want is None:
        c = collections.Counter(kind(l.split()[0]) for l in body.split("\n") if l.startswith("\t") and l[1] not in ".;")
        print(name[:70], sum(c.values()), dict(c))
    elif want in name:
        blocks = re.split(r"\n(\.LBB[0-9_]+):[^\n]*\n", body)
        for j in range(1, len(blocks), 2):
            c = collections.Counter(kind(l.split()[0]) for l in blocks[j + 1].split("\n") if l.startswith("\t") and l[1] not in ".;")
            if sum(c.values()) > 40:
                print(blocks[j], sum(c.values()), dict(c))
