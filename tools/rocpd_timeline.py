"""GPU occupancy of the tail of a rocprofv3 rocpd (SQLite) kernel trace:
    python tools/rocpd_timeline.py <results.db> [tail_fraction=0.5]
Prints, for the last `tail_fraction` of the trace: number of launches, sum of kernel durations, the UNION of the busy
intervals (time with at least one kernel running), the idle time inside the span and a histogram of the idle gaps between
consecutive kernels -- i.e. whether a step is bounded by kernel time or by launch / dependency latency."""
import sqlite3
import sys


def main(path, tail=0.5):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute('pragma table_info(rocpd_kernel_dispatch)')]
    qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
    q = f', d.{qcol}' if qcol else ', 0'
    rows = db.execute(f'select d.start, d.end{q} from rocpd_kernel_dispatch d order by d.start').fetchall()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    cut = t1 - tail * (t1 - t0)
    rows = [r for r in rows if r[0] >= cut]
    span = max(r[1] for r in rows) - rows[0][0]
    busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    gaps = []
    for s, e, _ in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    ksum = sum(e - s for s, e, _ in rows)
    print(f'launches {len(rows)}  span {span / 1e6:.3f} ms  sum of kernel durations {ksum / 1e6:.3f} ms  '
          f'busy (union) {busy / 1e6:.3f} ms = {100.0 * busy / span:.1f}%  idle {(span - busy) / 1e6:.3f} ms')
    per_q = {}
    for s, e, qid in rows:
        per_q[qid] = per_q.get(qid, 0) + e - s
    print('kernel time per queue/stream (ms):', {k: round(v / 1e6, 3) for k, v in sorted(per_q.items(), key=lambda kv: -kv[1])})
    edges = [0, 2e3, 5e3, 10e3, 20e3, 50e3, 1e5, 1e6, 1e12]
    for lo, hi in zip(edges[:-1], edges[1:]):
        g = [x for x in gaps if lo <= x < hi]
        if g:
            print(f'  idle gaps {lo / 1e3:7.0f}..{hi / 1e3:<9.0f} us: {len(g):6d}  total {sum(g) / 1e6:8.3f} ms')


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
