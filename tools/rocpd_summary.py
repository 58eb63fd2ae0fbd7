#!/usr/bin/env python
"""Per-kernel summary (calls, total/avg/min/max duration) of a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes DIR/NAME_results.db on ROCm 7.2).

    python tools/rocpd_summary.py gpurun_out/prof/r01_results.db > profiles/r01_kernel_stats.txt
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    m = re.search(r'conv_mfma_kernel<tpz::ConvCfg<([^>]*)>\s*,\s*(\d+)\s*,\s*(\d+)\s*>', name)
    if m:
        a = [x.strip() for x in m.group(1).split(',')]
        keys = ['K', 'D', 'MT', 'TD', 'TH', 'TW', 'KG', 'RPS', 'CIN1', 'DIMS']
        cfg = ','.join(f'{k}={v}' for k, v in zip(keys, a)).replace('CIN1=false', 'CIN1=0').replace('CIN1=true', 'CIN1=1')
        return f'conv_mfma_kernel<{cfg},EPI={m.group(2)}>'
    name = re.sub(r'^void ', '', name)
    return name if len(name) < 110 else name[:107] + '...'


def main(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith('rocpd_kernel_dispatch'))
    ks = next(t for t in tabs if t.startswith('rocpd_info_kernel_symbol'))
    cols = [r[1] for r in c.execute(f'pragma table_info({ks})')]
    namecol = 'display_name' if 'display_name' in cols else ('kernel_name' if 'kernel_name' in cols else 'name')
    rows = c.execute(f'select s.{namecol}, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) '
                     f'from {kd} d join {ks} s on d.kernel_id = s.id group by s.{namecol} order by 3 desc').fetchall()
    total = sum(r[2] for r in rows)
    print(f'# {path}: {sum(r[1] for r in rows)} kernel dispatches, {total / 1e6:.3f} ms total GPU kernel time')
    print(f'{"calls":>7} {"total_ms":>11} {"avg_ms":>10} {"min_ms":>10} {"max_ms":>10} {"pct":>6}  kernel')
    for name, n, tot, mn, mx in rows:
        print(f'{n:7d} {tot / 1e6:11.3f} {tot / n / 1e6:10.4f} {mn / 1e6:10.4f} {mx / 1e6:10.4f} {100.0 * tot / total:6.2f}  {short(name)}')


if __name__ == '__main__':
    main(sys.argv[1])
