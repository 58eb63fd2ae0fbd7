#!/usr/bin/env python
"""GPU busy time vs wall time of a rocprofv3 kernel trace (rocpd SQLite): the union of the kernel intervals, the idle
gaps between them (histogram), and the kernels that follow the longest gaps.

    python tools/trace_gaps.py gpurun_out/prof/x_results.db [t0_fraction t1_fraction]
"""
import sqlite3
import sys


def main(path, f0=0.0, f1=1.0):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith('rocpd_kernel_dispatch'))
    ks = next(t for t in tabs if t.startswith('rocpd_info_kernel_symbol'))
    cols = [r[1] for r in c.execute(f'pragma table_info({ks})')]
    namecol = 'display_name' if 'display_name' in cols else ('kernel_name' if 'kernel_name' in cols else 'name')
    rows = c.execute(f'select d.start, d.end, s.{namecol} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start').fetchall()
    t_lo, t_hi = rows[0][0], max(r[1] for r in rows)
    a, b = t_lo + f0 * (t_hi - t_lo), t_lo + f1 * (t_hi - t_lo)
    rows = [r for r in rows if r[0] >= a and r[1] <= b]
    wall = rows[-1][1] - rows[0][0] if rows else 0
    busy, cur_end, gaps, prev = 0, rows[0][0], [], ''
    for s, e, n in rows:
        if s > cur_end:
            gaps.append((s - cur_end, n, prev, (s - rows[0][0]) / 1e6))
            busy += e - s
            cur_end = e
        elif e > cur_end:
            busy += e - cur_end
            cur_end = e
        prev = n
    print(f'{len(rows)} dispatches, wall {wall / 1e6:.3f} ms, GPU busy (union) {busy / 1e6:.3f} ms, idle {100 * (1 - busy / wall):.1f} %')
    edges = [2, 5, 10, 20, 50, 100, 1000, 1e9]
    hist = [0] * len(edges)
    tot = [0.0] * len(edges)
    for g, *_ in gaps:
        us = g / 1e3
        for i, e in enumerate(edges):
            if us < e:
                hist[i] += 1
                tot[i] += us
                break
    lo = 0
    for e, h, t in zip(edges, hist, tot):
        print(f'  gaps {lo:>6g} .. {e:<6g} us: {h:6d}  total {t / 1e3:8.3f} ms')
        lo = e
    print('  longest gaps (us, at ms, previous kernel -> next kernel):')
    for g, n, pv, at in sorted(gaps, reverse=True)[:40]:
        print(f'    {g / 1e3:9.1f}  {at:9.2f}  {pv[:60]} -> {n[:60]}')


if __name__ == '__main__':
    main(sys.argv[1], *(float(x) for x in sys.argv[2:4]))
