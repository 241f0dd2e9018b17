"""Counters of the f16x3 NT kernel with B split in the kernel (fp32 operand) against B from pre-split planes (P4), same shapes:
instructions, matrix-pipe occupancy and effective clock per launch.  One rocprofv3 pass per variant (counters + --kernel-trace only).
    python tools/pmc_p4.py [out.txt]          (needs a GPU)"""
import collections, csv, glob, os, shutil, subprocess, sys

REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
M, SIMDS = 557056, 1024
COUNTERS = ['GRBM_GUI_ACTIVE', 'SQ_WAVE_CYCLES', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU', 'SQ_WAIT_INST_ANY',
            'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_WAVES']


def run(N, K, p4, out):
    d = '/tmp/pmc_p4'
    shutil.rmtree(d, ignore_errors=True)
    cmd = ['rocprofv3', '--kernel-trace', '--pmc'] + COUNTERS + ['-f', 'csv', '-d', d, '--', sys.executable,
                                                                 os.path.join(REPO, 'tools', 'one_gemm_g3.py'), 'nt', str(N), str(K)] + (['p4'] if p4 else [])
    r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), capture_output=True, text=True, timeout=300)
    tot, n, dur = collections.defaultdict(float), collections.Counter(), []
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if 'gemm_nt_g3_kernel' in row['Kernel_Name']:
                tot[row['Counter_Name']] += float(row['Counter_Value']); n[row['Counter_Name']] += 1
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if 'gemm_nt_g3_kernel' in row['Kernel_Name']:
                dur.append(float(row['End_Timestamp']) - float(row['Start_Timestamp']))
    label = f'{M} x {N} x {K}, B {"from P4 planes" if p4 else "fp32, split in the kernel"}'
    if not dur or not tot:
        out.write(f'== {label}: no data (rc {r.returncode})\n{r.stderr[-800:]}\n'); return
    t = sum(dur) / len(dur) * 1e-9
    avg = {k: tot[k] / n[k] for k in tot}
    mfma = 3.0 * M * N * K / (32 * 32 * 16)
    out.write(f'== {label}: {t * 1e6:.1f} us per launch in this profiled pass ({2.0 * M * N * K / t / 1e12:.0f} TFLOP/s)\n')
    for k in COUNTERS:
        if k in avg: out.write(f'   {k:28s} {avg[k]:16.0f}\n')
    g = avg.get('GRBM_GUI_ACTIVE')
    if g:
        for div in (1, 8):
            clk = g / div / t / 1e9
            if clk <= 2.6:
                out.write(f'   effective clock {clk:.3f} GHz; matrix pipe busy {100 * avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (g / div * SIMDS):.1f} %; '
                          f'VALU instructions per MFMA {avg.get("SQ_INSTS_VALU", 0) / (mfma / 64 * 1.0) if False else avg.get("SQ_INSTS_VALU", 0) / mfma:.2f}\n')
                break
    out.flush()


def main():
    out = open(sys.argv[1], 'w') if len(sys.argv) > 1 else sys.stdout
    out.write(__doc__.split('\n\n')[0] + '\n\n')
    for N, K in ((1024, 256), (256, 1024), (256, 256)):
        for p4 in (False, True):
            run(N, K, p4, out)


if __name__ == '__main__':
    main()
