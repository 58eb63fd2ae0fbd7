#!/usr/bin/env python
"""Per-kernel mean of one PMC counter from a rocprofv3 --pmc output directory (counter_collection.csv):
    python tools/pmc_kernels.py DIR COUNTER [regex]"""
import collections
import csv
import glob
import os
import re
import sys

d, counter = sys.argv[1], sys.argv[2]
pat = sys.argv[3] if len(sys.argv) > 3 else 'conv_split'
vals = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter and re.search(pat, r['Kernel_Name']):
            vals[re.sub(r'void tpz::|tpz::|\(.*', '', r['Kernel_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(vals.items(), key=lambda kv: -sum(kv[1])):
    print(f'{counter} mean {sum(v) / len(v):16.1f}  n {len(v):3d}  {k[:110]}')
