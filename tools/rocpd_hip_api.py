"""Slowest / most expensive HIP runtime API calls of a `rocprofv3 --hip-runtime-trace` rocpd database:
    python tools/rocpd_hip_api.py <results.db> [skip_fraction]"""
import sqlite3
import sys


def main(path, skip=0.0):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute('pragma table_info(regions)')]
    name_col = 'name' if 'name' in cols else cols[1]
    rows = db.execute(f'select {name_col}, start, end from regions order by start').fetchall()
    t0, t1 = rows[0][1], rows[-1][2]
    cut = t0 + skip * (t1 - t0)
    agg = {}
    worst = []
    for name, s, e in rows:
        if s < cut:
            continue
        a = agg.setdefault(name, [0, 0, 0])
        a[0] += 1
        a[1] += e - s
        a[2] = max(a[2], e - s)
        if e - s > 500_000:
            worst.append((e - s, name, s - cut))
    print('api,calls,total_ms,avg_us,max_us')
    for name, (n, ns, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:15]:
        print(f'{name},{n},{ns / 1e6:.3f},{ns / n / 1e3:.1f},{mx / 1e3:.1f}')
    print('calls longer than 0.5 ms:')
    for d, name, at in sorted(worst, key=lambda w: w[2])[:40]:
        print(f'  t={at / 1e6:9.3f} ms  {d / 1e6:7.3f} ms  {name}')


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.0)
