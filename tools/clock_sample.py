"""Samples the shader clock (pp_dpm_sclk) and socket power (hwmon power1_average / power1_input) from sysfs every 20 ms
while a command runs; prints the distribution of the samples taken while the GPU was busy (power above 500 W).
    python tools/clock_sample.py <command...>"""
import glob
import statistics
import subprocess
import sys
import threading
import time


def read(path):
    try:
        return open(path).read()
    except Exception:
        return ''


def main():
    cards = []                       # every GPU of the node is in sysfs; the busy one is found by its power
    for d in glob.glob('/sys/class/drm/card*/device'):
        if not read(d + '/pp_dpm_sclk'):
            continue
        pfile = None
        for h in glob.glob(d + '/hwmon/hwmon*'):
            for n in ('power1_average', 'power1_input'):
                if pfile is None and read(f'{h}/{n}').strip():
                    pfile = f'{h}/{n}'
        cards.append((d, pfile))
    per_card = {d: [] for d, _ in cards}
    stop = [False]

    def loop():
        while not stop[0]:
            for d, pfile in cards:
                sclk = None
                for line in read(d + '/pp_dpm_sclk').splitlines():
                    if '*' in line and not line.startswith('S'):
                        sclk = int(''.join(c for c in line.split(':')[1] if c.isdigit()))
                p = read(pfile).strip() if pfile else ''
                per_card[d].append((time.time(), sclk, int(p) / 1e6 if p.isdigit() else None))
            time.sleep(0.02)

    t = threading.Thread(target=loop)
    t.start()
    rc = subprocess.call(sys.argv[1:])
    stop[0] = True
    t.join()
    samples = max(per_card.values(), key=lambda v: max([x[2] or 0 for x in v] or [0]))
    busy = [(s, p) for _, s, p in samples if s and p and p > 500]
    print(f'[clock_sample] {len(samples)} samples, {len(busy)} busy (> 500 W)')
    if busy:
        cl, pw = [b[0] for b in busy], [b[1] for b in busy]
        qs = statistics.quantiles(cl, n=10) if len(cl) >= 10 else cl
        print(f'[clock_sample] sclk MHz: median {statistics.median(cl):.0f}  min {min(cl)}  max {max(cl)}  deciles {[round(q) for q in qs]}')
        print(f'[clock_sample] power W: median {statistics.median(pw):.0f}  max {max(pw):.0f}')
    return rc


if __name__ == '__main__':
    sys.exit(main())
