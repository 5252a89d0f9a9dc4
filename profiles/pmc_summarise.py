import csv, collections, sys, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sorted(glob.glob(sys.argv[1] + '/s[0-9]')):
    for r in csv.DictReader(open(d + '/p_counter_collection.csv')):
        k = r['Kernel_Name'].split('(')[0][:40]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for r in csv.DictReader(open(d + '/p_kernel_trace.csv')):
        k = r['Kernel_Name'].split('(')[0][:40]
        dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in agg:
    if not any(s in k for s in ('warp_sep', 'pyr_down', 'blend_quad', 'blend_level', 'blend_oct', 'border0', 'apron', 'resize_area')): continue
    d = agg[k]; m = {c: sum(v)/len(v) for c, v in d.items()}
    us = sum(dur[k])/len(dur[k])
    print(f"{k}  avg {us:.1f} us")
    for c in sorted(m): print(f"    {c:34s} {m[c]:16.0f}")
    if 'SQ_INSTS_VALU' in m and m.get('SQ_WAVES'):
        print(f"    -> VALU/wave {m['SQ_INSTS_VALU']/m['SQ_WAVES']:.0f}, VMEM_RD/wave {m['SQ_INSTS_VMEM_RD']/m['SQ_WAVES']:.1f}, cyc/VALU {m['SQ_ACTIVE_INST_VALU']*4/m['SQ_INSTS_VALU']:.2f}, VALU busy share {m['SQ_ACTIVE_INST_VALU']*4/1024/(us*2.1e3):.2f} (at 2.1GHz)")
    if 'FETCH_SIZE' in m: print(f"    -> HBM read {m['FETCH_SIZE']*1024*2/1e6:.1f} MB (FETCH_SIZE x2 gfx950 correction), write {m.get('WRITE_SIZE',0)*1024/1e6:.1f} MB")
