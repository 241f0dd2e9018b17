"""Per-kernel summary of a rocprofv3 rocpd (SQLite) kernel trace:
    python tools/rocpd_stats.py <results.db> [skip_fraction] [--by-grid]
Prints a CSV like `rocprofv3 --stats` (name, calls, total ms, avg us, % of GPU time); `skip_fraction` drops the leading
part of the trace (warm-up steps); `--by-grid` keeps launches of one kernel with different grid sizes apart (the
attention kernels serve three problem shapes in the decoder step)."""
import sqlite3
import sys


def main(path, skip=0.0, by_grid=False):
    db = sqlite3.connect(path)
    grid = ", d.grid_size_x" if by_grid else ", 0"
    rows = db.execute(f'select k.kernel_name, d.start, d.end{grid} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol k '
                      'on d.kernel_id = k.id order by d.start').fetchall()
    t0, t1 = rows[0][1], rows[-1][2]
    cut = t0 + skip * (t1 - t0)
    agg = {}
    for name, s, e, gx in rows:
        if s < cut:
            continue
        short = name.split('(')[0]
        for pre in ('void ', 'vq::'):
            short = short.replace(pre, '')
        a = agg.setdefault(short[:90] + (f' grid={gx}' if by_grid else ''), [0, 0])
        a[0] += 1
        a[1] += e - s
    total = sum(v[1] for v in agg.values())
    print('kernel,calls,total_ms,avg_us,pct')
    for name, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'"{name}",{n},{ns / 1e6:.3f},{ns / n / 1e3:.1f},{100.0 * ns / total:.2f}')
    print(f'"TOTAL kernel time",{sum(v[0] for v in agg.values())},{total / 1e6:.3f},,100.0')
    print(f'"trace span (ms)",,{(t1 - cut) / 1e6:.3f},,')


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if a != '--by-grid']
    main(args[0], float(args[1]) if len(args) > 1 else 0.0, by_grid='--by-grid' in sys.argv)
