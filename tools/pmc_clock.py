"""Effective shader clock and MFMA-pipe occupancy of the bf16x6 GEMM kernels, by the microarchitecture guide's method:

    effective clock = GRBM_GUI_ACTIVE / kernel wall time          (MI355X_MICROARCH.md, "DVFS give-back")
    MFMA pipe busy  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs)

next to round 2's own derivation (4 x SQ_WAVE_CYCLES / resident waves / wall time), so the two can be compared.

    python tools/pmc_clock.py [out.txt]      (runs rocprofv3 --kernel-trace --pmc ... over tools/one_gemm*.py; needs a GPU)

One rocprofv3 pass per shape: GRBM_GUI_ACTIVE is a GRBM counter (2 slots, independent of the 8 SQ slots), so it is collected
in the SAME pass as the SQ counters it is compared with.  Counters only + --kernel-trace (no other trace domain).
"""
import collections, csv, glob, os, re, shutil, subprocess, sys

REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
M = 557056
SIMDS = 1024
COUNTERS = ['GRBM_GUI_ACTIVE', 'GRBM_COUNT', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_WAVES', 'SQ_BUSY_CU_CYCLES',
            'SQ_INSTS_VALU_MFMA_MOPS_BF16']


def available(counters):
    """keep the counters rocprofv3 lists on this box (an unknown name fails the whole pass)"""
    try:
        r = subprocess.run(['rocprofv3', '-L'], capture_output=True, text=True, timeout=120, cwd='/tmp')
        txt = r.stdout + r.stderr
        keep = [c for c in counters if re.search(r'\b' + c + r'\b', txt)]
        return keep if 'GRBM_GUI_ACTIVE' in keep or keep else counters
    except Exception:
        return counters


def run(script, args, env_extra, match, waves, N, K, label, out):
    d = '/tmp/pmc_clock'
    shutil.rmtree(d, ignore_errors=True)
    env = dict(os.environ, TMPDIR='/tmp', **env_extra)
    cmd = ['rocprofv3', '--kernel-trace', '--pmc'] + COUNTERS + ['-f', 'csv', '-d', d, '--', sys.executable,
                                                                 os.path.join(REPO, 'tools', script)] + [str(a) for a in args]
    try:
        r = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True, timeout=300)
    except subprocess.TimeoutExpired:
        out.write(f'== {label}: rocprofv3 timed out\n'); return
    tot = collections.defaultdict(float); n = collections.Counter(); dur = []
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if any(m in row['Kernel_Name'] for m in match):
                tot[row['Counter_Name']] += float(row['Counter_Value']); n[row['Counter_Name']] += 1
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if any(m in row['Kernel_Name'] for m in match):
                dur.append(float(row['End_Timestamp']) - float(row['Start_Timestamp']))
    if not dur or not tot:
        out.write(f'== {label}: no data (rc {r.returncode})\n{r.stderr[-1500:]}\n'); return
    t = sum(dur) / len(dur) * 1e-9
    avg = {k: tot[k] / n[k] for k in tot}
    out.write(f'== {label}   ({len(dur)} launches, {t * 1e6:.1f} us each in this profiled pass, '
              f'{2.0 * M * N * K / t / 1e12:.1f} TFLOP/s)\n')
    for k in COUNTERS:
        if k in avg: out.write(f'   {k:30s} {avg[k]:18.0f} per launch\n')
    mfma = 6.0 * M * N * K / (32 * 32 * 16)                       # v_mfma_f32_32x32x16_bf16 issued per launch
    if 'GRBM_GUI_ACTIVE' in avg:
        g = avg['GRBM_GUI_ACTIVE']
        # rocprofv3 may report the counter per XCD-summed (8 GRBMs) or once; a clock above 2.4 GHz is impossible
        for div in (1, 8):
            clk = g / div / t / 1e9
            if clk <= 2.6:
                out.write(f'   GRBM: effective clock = GRBM_GUI_ACTIVE / {div} / wall = {clk:.3f} GHz')
                if 'SQ_VALU_MFMA_BUSY_CYCLES' in avg:
                    busy = avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (g / div * SIMDS)
                    out.write(f';  MFMA pipe busy = MFMA_BUSY / (GUI_ACTIVE x {SIMDS}) = {100 * busy:.1f} %')
                out.write(f';  32 x #MFMA / (GUI_ACTIVE x {SIMDS}) = {100 * 32 * mfma / (g / div * SIMDS):.1f} %\n')
                break
    if 'SQ_WAVE_CYCLES' in avg:
        cyc = 4.0 * avg['SQ_WAVE_CYCLES'] / waves
        out.write(f'   SQ (round-2 method): 4 x SQ_WAVE_CYCLES / {waves} resident waves = {cyc:.0f} cycles per wave -> {cyc / t / 1e9:.3f} GHz; '
                  f'32 x #MFMA per SIMD / cycles = {100 * 32 * mfma / SIMDS / cyc:.1f} %\n')
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in avg:
        out.write(f'   SQ_VALU_MFMA_BUSY_CYCLES / (32 x #MFMA) = {avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (32 * mfma):.3f}\n')
    out.flush()


def main():
    out = open(sys.argv[1], 'w') if len(sys.argv) > 1 else sys.stdout
    out.write(__doc__.split('\n\n')[0] + '\n\n')
    global COUNTERS
    COUNTERS = available(COUNTERS)
    out.write('counters collected: ' + ' '.join(COUNTERS) + '\n\n')
    for N, K in ((768, 256), (256, 256), (1024, 256), (256, 768), (256, 1024)):
        run('one_gemm.py', [N, K], {'VQCPC_ONE_GEMM_MODE': '1'}, ['gemm_nt_x6_'], 2048, N, K, f'NT gemm_nt_x6_pp_kernel {M} x {N} x {K} (bias epilogue)', out)
    run('one_gemm.py', [256, 1024], {'VQCPC_ONE_GEMM_MODE': '1', 'VQCPC_PP_ABL': '3'}, ['gemm_nt_x6_'], 2048, 256, 1024,
        f'NT schedule-only ablation (ABL 3) {M} x 256 x 1024', out)
    for N, K in ((1024, 256), (256, 256), (768, 256)):
        run('one_gemm_tn.py', [N, K], {'VQCPC_TN_MODE': '1'}, ['gemm_tn_x6_p', 'gemm_tn_x6_256'], 2048, N, K,
            f'TN gemm_tn_x6_pp_kernel (weight gradient) {M} x {N} x {K}', out)
    if out is not sys.stdout: out.close()


if __name__ == '__main__':
    main()
