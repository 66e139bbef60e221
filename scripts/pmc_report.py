import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in agg.items():
    if not k.startswith('pol::') and not k.startswith('void pol::'): continue
    if d.get('SQ_WAVE_CYCLES', 0) < 1e8: continue
    print('%-28s waves=%.3g valu/wave=%.0f lane_util=%.3f valu_busy=%.3f wait_any=%.3f wait_inst=%.3f' % (
        k[-28:], d['SQ_WAVES'], d['SQ_INSTS_VALU'] / d['SQ_WAVES'], d['SQ_THREAD_CYCLES_VALU'] / (64 * d['SQ_ACTIVE_INST_VALU']),
        d['SQ_ACTIVE_INST_VALU'] / d['SQ_WAVE_CYCLES'], d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES'], d['SQ_WAIT_INST_ANY'] / d['SQ_WAVE_CYCLES']))
