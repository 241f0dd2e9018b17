"""HBM bytes per GEMM launch from two rocprofv3 PMC passes of bench.py (MI355X_MICROARCH.md, HBM / rocprofv3 section):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d <dir>/fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d <dir>/write -- python bench.py ... (same command)
    python tools/pmc_hbm_traffic.py <dir> [steps incl. warm-up = 3] > profiles/rNN_gemm_hbm_traffic.json

FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 64 B requests as 32 B, hence the factor 2 (calibrated on
the QKV projection: WRITE_SIZE == M*N*4 exactly).  Kernels are grouped as gemm_nt (every NT kernel) / gemm_tn."""
import collections
import csv
import glob
import json
import sys


def collect(d, counter):
    tot = collections.defaultdict(float)
    n = collections.Counter()
    for f in glob.glob(f'{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            k = r['Kernel_Name']
            grp = 'gemm_nt' if 'gemm_nt' in k else 'gemm_tn' if 'gemm_tn' in k else None
            if grp:
                tot[grp] += float(r['Counter_Value'])
                n[grp] += 1
    return tot, n


def main():
    base = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3          # bench steps + warm-up steps of the profiled command
    fetch, nf = collect(base + '/fetch', 'FETCH_SIZE')
    write, nw = collect(base + '/write', 'WRITE_SIZE')
    out = {}
    for grp in ('gemm_nt', 'gemm_tn'):
        launches = nf[grp]
        assert launches and launches == nw[grp], (grp, nf[grp], nw[grp])
        f_kb, w_kb = fetch[grp] / launches, write[grp] / launches
        per_launch = (2 * f_kb + w_kb) * 1024
        out[grp] = dict(launches_profiled=launches, steps_profiled=steps, fetch_size_kb_raw=f_kb, write_size_kb_raw=w_kb,
                        hbm_bytes_per_launch=per_launch, hbm_bytes_per_step=per_launch * launches / steps)
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
