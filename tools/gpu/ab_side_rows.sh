for b in 8 16 32 64; do for v in 0 32768 1000000; do ALIGNN_AMD_SIDE_MIN_ROWS=$v python bench.py --batch $b --no-cpu-baseline --streamed-steps 0 --eager-steps 12 --steps 5 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); e=d['eager_launches']; print('B=$b SIDE_MIN_ROWS=$v eager', e['ms_per_step'], 'enqueue', e['host_enqueue_ms_per_step'], 'replay', d['ms_per_step'])"; done; done
