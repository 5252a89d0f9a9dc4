for e in 0 1 2; do for tw in 256 64; do echo "EXP=$e TW=$tw"; SSP_WARP_EXP=$e SSP_WARP_TW=$tw timeout -k 10 200 python bench.py --no-traffic --no-cpu-baseline --steps 10 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['ms_per_step'], [(k['kernel'], round(k['avg_us'],1)) for k in j['kernels'] if k['kernel'].startswith('warp')])
" || exit 1; done; done
