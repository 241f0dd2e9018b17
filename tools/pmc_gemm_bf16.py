"""Memory-side counter triage of ONE bf16 NT GEMM shape (what paces the operand delivery?): several rocprofv3 --kernel-trace --pmc
passes (counters in their own runs, no other trace domain) over tools/bench_gemm_bf16.py restricted to one shape / form / kernel
variant; per kernel: average duration and per-launch counter averages, plus derived rates.
    python tools/pmc_gemm_bf16.py NxK "form" variant [rows]        e.g.  512x2048 "none -> bf16" 1"""
import collections, csv, glob, os, shutil, subprocess, sys
REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
GROUPS = [['SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_VALU_MFMA_BUSY_CYCLES',
           'SQ_INST_CYCLES_VMEM_RD', 'SQ_ACTIVE_INST_LDS', 'GRBM_GUI_ACTIVE'],
          ['TCP_PENDING_STALL_CYCLES_sum', 'TCP_TCC_READ_REQ_sum', 'TCP_TCC_READ_REQ_LATENCY_sum', 'TCP_TCR_TCP_STALL_CYCLES_sum'],
          ['TCC_HIT_sum', 'TCC_MISS_sum', 'TCC_REQ_sum', 'TCC_TAG_STALL_sum'],
          ['TCP_TCC_WRITE_REQ_sum', 'TCP_TCC_WRITE_REQ_LATENCY_sum', 'TCP_TOTAL_CACHE_ACCESSES_sum', 'TCP_TCP_TA_DATA_STALL_CYCLES_sum'],
          ['SQ_INSTS_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_WAIT_INST_LDS', 'SQ_LDS_ADDR_CONFLICT', 'SQ_LDS_DATA_FIFO_FULL']]


def main():
    shape, form, variant = sys.argv[1], sys.argv[2], sys.argv[3]
    rows = sys.argv[4] if len(sys.argv) > 4 else '1114112'
    env = dict(os.environ, TMPDIR='/tmp', VQCPC_BF16_SHAPES=shape, VQCPC_BF16_FORMS=form, VQCPC_BF16_VARIANTS=variant, VQCPC_BF16_NO_TN='1')
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    dur = collections.defaultdict(list)
    for g in GROUPS:
        d = '/tmp/pmc_gb'
        shutil.rmtree(d, ignore_errors=True)
        cmd = ['rocprofv3', '--kernel-trace', '--pmc'] + g + ['-f', 'csv', '-d', d, '--', sys.executable,
               os.path.join(REPO, 'tools', 'bench_gemm_bf16.py'), rows]
        import signal
        proc = subprocess.Popen(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            _, err = proc.communicate(timeout=100)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
            proc.wait()
            print('group timed out (100 s):', g, flush=True)
            continue
        r = type('R', (), {'stderr': err})()
        got = False
        for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
            for row in csv.DictReader(open(f)):
                k = row['Kernel_Name']
                if 'gemm_nt_bf16' not in k:
                    continue
                got = True
                tot[k][row['Counter_Name']] += float(row['Counter_Value']); n[k][row['Counter_Name']] += 1
        for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
            for row in csv.DictReader(open(f)):
                if 'gemm_nt_bf16' in row['Kernel_Name']:
                    dur[row['Kernel_Name']].append(float(row['End_Timestamp']) - float(row['Start_Timestamp']))
        if not got:
            print('group failed:', g, (r.stderr or '')[-300:].replace('\n', ' | '))
    for k in tot:
        a = {c: tot[k][c] / n[k][c] for c in tot[k]}
        us = sum(dur[k]) / max(len(dur[k]), 1) / 1e3
        print(f'== {k[:90]}  avg {us:.1f} us under counters, {shape} {form} variant {variant}')
        for c in sorted(a):
            print(f'   {c:38s} {a[c]:16.0f}')
        cyc = a.get('GRBM_GUI_ACTIVE', 0) / 8.0
        if cyc:
            print(f'   -> clock {cyc / us / 1e3:.2f} GHz; MFMA pipes busy {a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024):.3f}')
        if a.get('TCP_TCC_READ_REQ_sum'):
            print(f'   -> L1->L2 read latency {a["TCP_TCC_READ_REQ_LATENCY_sum"] / a["TCP_TCC_READ_REQ_sum"]:.0f} cycles over '
                  f'{a["TCP_TCC_READ_REQ_sum"]:.3g} requests')
        if a.get('TCP_TCC_WRITE_REQ_sum'):
            print(f'   -> L1->L2 write latency {a.get("TCP_TCC_WRITE_REQ_LATENCY_sum", 0) / a["TCP_TCC_WRITE_REQ_sum"]:.0f} cycles over '
                  f'{a["TCP_TCC_WRITE_REQ_sum"]:.3g} requests')
        if a.get('TCC_REQ_sum'):
            print(f'   -> L2 hit rate {a.get("TCC_HIT_sum", 0) / max(a.get("TCC_HIT_sum", 0) + a.get("TCC_MISS_sum", 0), 1):.3f}')


if __name__ == '__main__':
    main()
