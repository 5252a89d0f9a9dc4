for e in 0 1 2 3 4; do echo "DBG=$e"; SSP_BLEND_DBG=$e timeout -k 10 200 python bench.py --no-traffic --no-cpu-baseline --steps 10 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['ms_per_step'], [(k['kernel'], round(k['avg_us'],1)) for k in j['kernels'] if k['kernel'].startswith('blend')])
" || exit 1; done
