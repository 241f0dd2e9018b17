"""SQ counter triage of the NON-GEMM kernels of a step: is a kernel waiting (memory latency) or issuing (instruction-bound)?
Two rocprofv3 --kernel-trace --pmc passes over an eager `bench.py --steps 2 --warmup 1 --no-graph`; per kernel family: average
duration, wave-cycle shares.     python tools/pmc_kernels.py [config] [out.txt]          (needs a GPU)
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves: shares of SQ_WAVE_CYCLES are comparable."""
import collections, csv, glob, os, re, shutil, subprocess, sys
REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
GROUPS = [['SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU',
           'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_LDS'],
          ['SQ_INSTS_VALU', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_LDS', 'SQ_INSTS_SALU', 'SQ_WAVES', 'SQ_INST_CYCLES_VMEM_RD',
           'SQ_WAIT_INST_LDS']]
FAMILIES = ['relattn16_bwd', 'relattn16_fwd', 'relattn_sub16_bwd', 'relattn_sub16_fwd', 'relattn_bwd_kernel', 'relattn_fwd_kernel',
            'relattn_sub_bwd', 'relattn_sub_fwd', 'block_table_segsum', 'embed_pos_bwd', 'embed_pos_fwd', 'add_ln_bwd', 'add_ln_fwd',
            'vq_fwd', 'vq_bwd', 'gru_step_fwd', 'gru_step_bwd', 'relattn_x_fwd', 'relattn_x_bwd_dq', 'relattn_x_bwd_dkv',
            'relattn_x_bwd_de', 'softmax_ce', 'upscale_bwd', 'upscale_fwd', 'embedding_bwd', 'adam_dev', 'sumsq_stage1',
            'reduce_many', 'reduce_splits', 'splitk_epilogue', 'gemm_nt_skinny', 'transpose_many', 'dropout_selu', 'nce_',
            'count_distinct', 'accumulate8', 'scale_rows', 'same_seq']


def family(name):
    for f in FAMILIES:
        if f in name:
            return f
    return None


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'C1'
    out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    dur = collections.defaultdict(list)
    for g in GROUPS:
        d = '/tmp/pmc_k'
        shutil.rmtree(d, ignore_errors=True)
        cmd = ['rocprofv3', '--kernel-trace', '--pmc'] + g + ['-f', 'csv', '-d', d, '--', sys.executable, os.path.join(REPO, 'bench.py'),
               '--config', cfg, '--steps', '2', '--warmup', '1', '--no-graph', '--no-cpu-baseline', '--no-kernel-timing', '--no-live-pmc',
               '--no-extras'] + (['--gemm-mode', 'bf16'] if cfg == 'C4' else [])
        subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), capture_output=True, text=True, timeout=900)
        for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
            for row in csv.DictReader(open(f)):
                fam = family(row['Kernel_Name'])
                if fam:
                    tot[fam][row['Counter_Name']] += float(row['Counter_Value']); n[fam][row['Counter_Name']] += 1
        for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
            for row in csv.DictReader(open(f)):
                fam = family(row['Kernel_Name'])
                if fam:
                    dur[fam].append(float(row['End_Timestamp']) - float(row['Start_Timestamp']))
    out.write(f'{"kernel":22s} {"us":>8s} {"waves":>8s} | shares of SQ_WAVE_CYCLES: {"wait_any":>9s} {"wait_inst":>9s} {"act_any":>8s} '
              f'{"valu":>6s} {"vmem":>6s} {"lds":>6s} | per wave: {"valu":>7s} {"vm_rd":>6s} {"vm_wr":>6s} {"lds":>6s} {"salu":>6s}\n')
    for fam in FAMILIES:
        if fam not in tot:
            continue
        a = {k: tot[fam][k] / n[fam][k] for k in tot[fam]}
        wc = a.get('SQ_WAVE_CYCLES', 0) or 1
        w = a.get('SQ_WAVES', 0) or 1
        us = sum(dur[fam]) / max(len(dur[fam]), 1) / 1e3
        sh = lambda k: 100.0 * a.get(k, 0) / wc
        out.write(f'{fam:22s} {us:8.1f} {w:8.0f} | {"":26s} {sh("SQ_WAIT_ANY"):8.1f}% {sh("SQ_WAIT_INST_ANY"):8.1f}% {sh("SQ_ACTIVE_INST_ANY"):7.1f}% '
                  f'{sh("SQ_ACTIVE_INST_VALU"):5.1f}% {sh("SQ_ACTIVE_INST_VMEM"):5.1f}% {sh("SQ_ACTIVE_INST_LDS"):5.1f}% | {"":9s} '
                  f'{a.get("SQ_INSTS_VALU", 0) / w:7.0f} {a.get("SQ_INSTS_VMEM_RD", 0) / w:6.0f} {a.get("SQ_INSTS_VMEM_WR", 0) / w:6.0f} '
                  f'{a.get("SQ_INSTS_LDS", 0) / w:6.0f} {a.get("SQ_INSTS_SALU", 0) / w:6.0f}\n')


if __name__ == '__main__':
    main()
