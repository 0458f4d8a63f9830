run() { env "$@" python bench.py --no-cpu-baseline --streamed-steps 0 --eager-steps 20 --steps 5 $BARGS 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); e=d['eager_launches']; print('$*', 'eager', e['ms_per_step'], 'enqueue', e['host_enqueue_ms_per_step'], 'replay', d['ms_per_step'])"; }
for rep in 1 2; do
run A=1
run ALIGNN_AMD_LANES=1
run ALIGNN_AMD_LANES=1 ALIGNN_AMD_FORK=1
done
