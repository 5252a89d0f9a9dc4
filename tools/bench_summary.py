import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['ms_per_step'], j['value']); [print("  ", k['kernel'], k['launches_per_step'], round(k['avg_us'], 1), round(k['achieved_GBps'])) for k in j['kernels']]
    else: print(l, end='')
