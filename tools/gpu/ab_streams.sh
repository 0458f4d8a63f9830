run() { env "$@" python bench.py --no-cpu-baseline --streamed-steps 0 --eager-steps 0 --steps 30 $BARGS 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('$*', 'replay', d['ms_per_step'])"; }
for rep in 1 2; do
run A=1
run ALIGNN_AMD_SIDE_STREAM=0
run ALIGNN_AMD_LANES=0
run ALIGNN_AMD_LANES=0 ALIGNN_AMD_SIDE_STREAM=0
done
