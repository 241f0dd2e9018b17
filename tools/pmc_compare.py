"""SQ counter groups of the bf16x6 NT (K = 1024) and TN (557056 x 1024 x 256, both LDS images) GEMM kernels side by side, per
launch and per wave and K step (16 rows / columns of the contraction), to see what the TN memory phase does more of:
    python tools/pmc_compare.py [out.txt]   (needs a GPU; rocprofv3 --kernel-trace --pmc <8 counters> per pass)."""
import collections, csv, glob, os, shutil, subprocess, sys
REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
M = 557056
GROUPS = [['SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_INSTS_VALU', 'SQ_INSTS_LDS', 'SQ_INSTS_SALU', 'SQ_INSTS_VMEM', 'SQ_INSTS_SMEM', 'GRBM_GUI_ACTIVE'],
          ['SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_MISC'],
          ['SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_LDS_ADDR_CONFLICT', 'SQ_LDS_UNALIGNED_STALL', 'SQ_LDS_DATA_FIFO_FULL', 'SQ_LDS_CMD_FIFO_FULL', 'SQ_INST_CYCLES_VMEM_RD', 'SQ_VALU_MFMA_BUSY_CYCLES']]


def run(script, args, env_extra, match, counters):
    d = '/tmp/pmc_cmp'
    shutil.rmtree(d, ignore_errors=True)
    env = dict(os.environ, TMPDIR='/tmp', **env_extra)
    cmd = ['rocprofv3', '--kernel-trace', '--pmc'] + counters + ['-f', 'csv', '-d', d, '--', sys.executable, os.path.join(REPO, 'tools', script)] + [str(a) for a in args]
    subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True, timeout=300)
    tot = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if any(m in row['Kernel_Name'] for m in match):
                tot[row['Counter_Name']] += float(row['Counter_Value']); n[row['Counter_Name']] += 1
    dur = []
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if any(m in row['Kernel_Name'] for m in match):
                dur.append((float(row['End_Timestamp']) - float(row['Start_Timestamp'])) / 1e3)
    res = {k: tot[k] / n[k] for k in tot}
    if dur:
        res['kernel_us(profiled pass)'] = sum(dur) / len(dur)
    return res


def main():
    out = open(sys.argv[1], 'w') if len(sys.argv) > 1 else sys.stdout
    cases = [('NT 557056 x 256 x 1024', 'one_gemm.py', [256, 1024], {'VQCPC_ONE_GEMM_MODE': '1'}, ['gemm_nt_x6_'], 1024 // 16 * (M // 256) * 1 / 256.0),
             ('TN row-pair image (pp)', 'one_gemm_tn.py', [1024, 256], {'VQCPC_TN_MODE': '1', 'VQCPC_TN_PQ': '0'}, ['gemm_tn_x6_p'], M / 16.0 * 4 / 256.0),
             ('TN quad-row image (pq)', 'one_gemm_tn.py', [1024, 256], {'VQCPC_TN_MODE': '1', 'VQCPC_TN_PQ': '1'}, ['gemm_tn_x6_p'], M / 16.0 * 4 / 256.0)]
    if os.environ.get('PMC_COMPARE') == 'nt':       # the output-tile epilogue: K = 256 (16 steps per tile) against K = 1024 (64)
        cases = [('NT 557056 x 256 x 1024', 'one_gemm.py', [256, 1024], {'VQCPC_ONE_GEMM_MODE': '1'}, ['gemm_nt_x6_'], 544.0),
                 ('NT 557056 x 1024 x 256', 'one_gemm.py', [1024, 256], {'VQCPC_ONE_GEMM_MODE': '1'}, ['gemm_nt_x6_'], 544.0),
                 ('NT 1024 x 256, no stores (ABL 32)', 'one_gemm.py', [1024, 256], {'VQCPC_ONE_GEMM_MODE': '1', 'VQCPC_PP_ABL': '32'}, ['gemm_nt_x6_'], 544.0)]
    if os.environ.get('PMC_COMPARE') == 'grad':     # six-product against three-product (gradient arithmetic) kernels, round 5
        g3 = {'VQCPC_ONE_GEMM_MODE': '1', 'VQCPC_ONE_GEMM_GRAD': '3'}
        g6 = {'VQCPC_ONE_GEMM_MODE': '1', 'VQCPC_ONE_GEMM_NOBIAS': '1'}
        cases = [('NT 256x1024 six', 'one_gemm.py', [256, 1024], g6, ['gemm_nt_x6_'], 544.0),
                 ('NT 256x1024 three', 'one_gemm.py', [256, 1024], g3, ['gemm_nt_x6_'], 544.0),
                 ('NT 1024x256 six', 'one_gemm.py', [1024, 256], g6, ['gemm_nt_x6_'], 544.0),
                 ('NT 1024x256 three', 'one_gemm.py', [1024, 256], g3, ['gemm_nt_x6_'], 544.0),
                 ('TN 1024x256 six', 'one_gemm_tn.py', [1024, 256], {'VQCPC_TN_MODE': '1'}, ['gemm_tn_x6_p'], M / 16.0 * 4 / 256.0),
                 ('TN 1024x256 three', 'one_gemm_tn.py', [1024, 256], {'VQCPC_TN_MODE': '1', 'VQCPC_ONE_GEMM_GRAD': '3'}, ['gemm_tn_x6_p'], M / 16.0 * 4 / 256.0)]
    if os.environ.get('PMC_COMPARE') == 'g3':       # the software-pipelined fp16 three-product kernels (csrc/gemm_grad.hip), round 5
        g3 = {'VQCPC_ONE_GEMM_MODE': '1', 'VQCPC_ONE_GEMM_GRAD': '3'}
        cases = [('NT 256x1024 bf16x3 pp', 'one_gemm.py', [256, 1024], g3, ['gemm_nt_x6_'], 544.0),
                 ('NT 256x1024 f16x3 g3', 'one_gemm_g3.py', ['nt', 256, 1024], {}, ['gemm_nt_g3'], 544.0),
                 ('NT 1024x256 f16x3 g3', 'one_gemm_g3.py', ['nt', 1024, 256], {}, ['gemm_nt_g3'], 544.0),
                 ('TN 1024x256 f16x3 g3', 'one_gemm_g3.py', ['tn', 1024, 256], {}, ['gemm_tn_g3'], M / 16.0 * 4 / 256.0)]
    res = {}
    for label, script, args, env, match, steps_per_wg in cases:
        r = {}
        for g in GROUPS:
            r.update(run(script, args, env, match, g))
        res[label] = (r, steps_per_wg)
    names = [c for g in GROUPS for c in g] + ['kernel_us(profiled pass)']
    out.write(f'{"counter (per launch)":32s}' + ''.join(f'{l:>28s}' for l in res) + '   | per wave and K step (2048 waves):' + '\n')
    for c in names:
        line = f'{c:32s}'
        per = ''
        for l, (r, steps) in res.items():
            v = r.get(c)
            line += f'{v:28.0f}' if v is not None else f'{"n/a":>28s}'
            if v is not None:
                per += f'{v / 2048.0 / steps:12.2f}'
        out.write(line + '   | ' + per + '\n')
    out.write('K steps per workgroup: ' + ', '.join(f'{l}: {s:.0f}' for l, (r, s) in res.items()) + '\n')


if __name__ == '__main__':
    main()
