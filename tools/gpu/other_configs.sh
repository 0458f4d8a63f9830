#!/bin/bash
# bench lines of the non-headline BASELINE configs + the 2-rank smoke (ranks share the one GPU, gloo collectives)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
b() { name=$1; shift; timeout 600 python bench.py "$@" > gpurun_out/cfg_$name.json 2> gpurun_out/cfg_$name.err; echo "$name rc=$? $(python -c "import json;d=json.load(open('gpurun_out/cfg_$name.json'));print(d['ms_per_step'],d['value'],d.get('eager_launches'))" 2>&1 | tail -1)"; }
b cfg1_b8 --batch 8 --no-cpu-baseline --streamed-steps 0
b cfg5_mol256 --kind molecule --batch 256 --cpu-graphs 64 --streamed-steps 0
b cfg2_b256 --batch 256 --no-cpu-baseline --streamed-steps 0 --steps 5
b cfg4_ff --model alignn_ff --batch 16 --atoms 200 --cpu-graphs 4 --streamed-steps 0 --steps 5 --warmup 2
b cfg4_energy_only --model alignn_atomwise --batch 16 --atoms 200 --no-cpu-baseline --streamed-steps 0 --steps 5 --warmup 2
ALIGNN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --streamed-steps 0 > gpurun_out/cfg_dp2_gloo_one_gpu.json 2> gpurun_out/cfg_dp2.err; echo "dp2 rc=$?"; tail -2 gpurun_out/cfg_dp2.err; python -c "import json;d=json.load(open('gpurun_out/cfg_dp2_gloo_one_gpu.json'));print('dp2', d['n_gpus'], d['ms_per_step'], d['value'])"
