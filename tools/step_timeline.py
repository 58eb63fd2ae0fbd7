#!/usr/bin/env python
"""One bench step out of a rocprofv3 kernel trace (rocpd SQLite): the dispatches between two consecutive launches of the
dominant kernel (the fused-head conv = one per micrograph), grouped by kernel -- time, launches -- with the GPU-busy union, the
idle time between dispatches and the overlap the patch lanes produce (sum of kernel times - busy time).

    python tools/step_timeline.py gpurun_out/prof/x_results.db [which_step_from_the_end = 2]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'^void ', '', name)
    m = re.search(r'conv_split_kernel<tpz::SplitCfg<([^>]*)>\s*,\s*(\d+)\s*,\s*(\d+)\s*,\s*(\d+)\s*>', name)
    if m:
        return f'conv_split<{m.group(1).replace(" ", "")}|EPI={m.group(2)}|MODE={m.group(4)}>'
    return name[:90]


def main(path, back=2):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith('rocpd_kernel_dispatch'))
    ks = next(t for t in tabs if t.startswith('rocpd_info_kernel_symbol'))
    cols = [r[1] for r in c.execute(f'pragma table_info({ks})')]
    namecol = 'display_name' if 'display_name' in cols else ('kernel_name' if 'kernel_name' in cols else 'name')
    rows = c.execute(f'select d.start, d.end, s.{namecol} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start').fetchall()
    heads = [i for i, r in enumerate(rows) if 'SplitCfg<5, 4, 128' in r[2] and ', 3, 0' in r[2]]
    if len(heads) < back + 1:
        print('not enough steps in the trace')
        return
    i0, i1 = heads[-back - 1] + 1, heads[-back] + 1          # (after head k .. through head k + 1)
    step = rows[i0:i1]
    t0, t1 = rows[i0 - 1][1], step[-1][1]
    busy, cur = 0, t0
    for s, e, _ in sorted(step):
        if e > cur:
            busy += e - max(s, cur)
            cur = e
    tot = sum(e - s for s, e, _ in step)
    print(f'step of {len(step)} dispatches: wall {1e-6 * (t1 - t0):.3f} ms, GPU busy (union) {1e-6 * busy:.3f} ms, idle {1e-6 * (t1 - t0 - busy):.3f} ms, '
          f'sum of kernel times {1e-6 * tot:.3f} ms (overlap between the patch lanes {1e-6 * (tot - busy):.3f} ms)')
    acc = {}
    for s, e, n in step:
        a = acc.setdefault(short(n), [0, 0])
        a[0] += e - s
        a[1] += 1
    for n, (t, k) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:40]:
        print(f'{1e-6 * t:9.3f} ms  {k:5d}  {n}')
    # the longest idle gaps and what follows them
    gaps, cur = [], t0
    for s, e, n in sorted(step):
        if s > cur:
            gaps.append((s - cur, short(n)))
        cur = max(cur, e)
    gaps.sort(reverse=True)
    print('longest gaps (us, next kernel):', [(round(g / 1e3, 1), n[:50]) for g, n in gaps[:8]])
    print(f'gaps: {len(gaps)} totalling {1e-6 * sum(g for g, _ in gaps):.3f} ms')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
