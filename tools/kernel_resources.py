"""Registers / spills / occupancy of every kernel in one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py alignn_amd/csrc/gemm_x6.hip [substring] [-- extra hipcc flags]"""
import re, subprocess, sys
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
src = args[0]; sub = args[1] if len(args) > 1 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: \s*(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    if sub in name:
        print(f"{name:58s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>4} spill {r.get('VGPRs Spill','?'):>3} "
              f"scratch {r.get('ScratchSize [bytes/lane]','?'):>4} occ {r.get('Occupancy [waves/SIMD]','?')}")
