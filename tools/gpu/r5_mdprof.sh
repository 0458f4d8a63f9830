#!/bin/bash
# kernel trace of the 200-atom MD evaluation (energies, forces, stresses in one C call)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/p_md
cat > /tmp/md_eval.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig, neighbors
from alignn_amd.synthetic import make_crystal
dev="cuda"; n=200
lat, frac, _ = make_crystal(n, 1234)
feats = torch.randn(n, 92, device=dev)
lat_d, frac_d = torch.from_numpy(lat).to(dev), torch.from_numpy(frac).to(dev)
torch.manual_seed(0)
model = ALIGNNAtomWise(ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=4, gcn_layers=4, hidden_features=256, atom_input_features=92, calculate_gradient=True, stresswise_weight=0.05)).to(dev).eval()
batch = neighbors.crystal_batch([lat_d], [frac_d], atom_features=[feats])
for _ in range(3): model(batch)
torch.cuda.synchronize()
import ctypes
for _ in range(5):
    torch.cuda.synchronize()
    model(batch)
torch.cuda.synchronize()
PY
timeout 600 rocprofv3 --kernel-trace -d /tmp/p_md -o r -- python /tmp/md_eval.py > gpurun_out/prof_md.err 2>&1
db=$(find /tmp/p_md -name "*.db" | head -1)
python tools/rocpd_stats.py $db > gpurun_out/prof_md_kernel_stats.txt
python tools/rocpd_sequence.py $db 2 > gpurun_out/prof_md_sequence.txt 2>/dev/null || true
head -60 gpurun_out/prof_md_kernel_stats.txt | cut -c1-150
