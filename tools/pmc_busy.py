#!/usr/bin/env python
"""Matrix-pipe occupancy per kernel from one rocprofv3 PMC pass:

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d DIR -o p --output-format csv -- <cmd>
    python tools/pmc_busy.py DIR [regex] > profiles/rNN_pmc_mfma_busy.txt

Per kernel: mean duration (End - Start of the dispatch), dispatches, GRBM_GUI_ACTIVE / 8 XCDs = shader cycles of the dispatch,
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles), clock = cycles / duration."""
import collections
import csv
import glob
import os
import re
import sys

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else 'conv_|mfma_spin'
rows = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        if not re.search(pat, name):
            continue
        rows[name][r['Counter_Name']].append(float(r['Counter_Value']))
        if r.get('Start_Timestamp') and r.get('End_Timestamp'):
            dur[name][r.get('Dispatch_Id', len(dur[name]))] = (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-6
if not any(dur.values()):
    for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name']
            if re.search(pat, name):
                dur[name][r.get('Dispatch_Id', len(dur[name]))] = (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-6
out = []
for name, c in rows.items():
    n = len(c.get('GRBM_GUI_ACTIVE', [])) or 1
    cyc = sum(c.get('GRBM_GUI_ACTIVE', [0])) / n / 8.0
    busy = sum(c.get('SQ_VALU_MFMA_BUSY_CYCLES', [0])) / n
    ms = sum(dur[name].values()) / max(1, len(dur[name])) if dur[name] else 0.0
    short = re.sub(r'void tpz::|tpz::|\(.*', '', name)
    out.append((ms * n, ms, n, cyc, busy / (1024.0 * cyc) if cyc else 0.0, cyc / (ms * 1e6) if ms else 0.0, short[:118]))
print(f'{"ms":>9} {"n":>4} {"cycles/XCD":>12} {"MFMA busy":>10} {"GHz":>6}  kernel')
for tot, ms, n, cyc, b, ghz, short in sorted(out, reverse=True):
    print(f'{ms:9.3f} {n:4d} {cyc:12.3e} {b:10.3f} {ghz:6.2f}  {short}')
