"""profiles/rNN_roofline_table.md from a kernel-stats CSV of the C1 step (tools/rocpd_stats.py output):
    python tools/roofline_table.py profiles/r06_c1_step_kernel_stats.csv > profiles/r06_roofline_table.md
Work per step = algorithmic FLOPs / bytes of the C1 configuration (each operand once), peaks from
MI355X_MICROARCH.md (HBM 8 TB/s, bf16x6 algorithmic MFMA ceiling 2500 / 6 = 416.7 TFLOP/s, f16x3 2500 / 3 = 833.3).
Round 6: TWO ceilings for the GEMM families -- t_mfma = FLOPs / MFMA peak of the arithmetic, t_hbm = algorithmic bytes (A, B, C and
every epilogue operand once, summed over the step's launches by bench.py: roofline.hbm.floor_ms_per_step x 8 TB/s) / 8 TB/s; the
binding ceiling is the larger, "of binding" = max(t_mfma, t_hbm) / measured (bench.py reports the same per launch family)."""
import csv, sys

ROWS_LN = 2 * (557056 + 139264 + 139264 + 34816)           # LayerNorm rows per step (8 instances), d = 256
GEMM_BYTES = {'gemm_nt_g3': 31.4e9, 'gemm_tn_g3': 13.9e9}     # algorithmic HBM bytes per step of the f16x3 NT / TN launches (bench.py, r06)
FAMILIES = [   # (label, match substrings, algorithmic work per step, unit, peak, note)
    ('NT GEMMs on three fp16 MFMAs per product (`gemm_nt_g3_kernel` + `gemm_nt_g3_tail_kernel`, round 5: every 256-tile launch of the step, forward and input gradients; ragged launches as whole rounds + tail rows)', ('gemm_nt_g3',), 2.577e12, 'TFLOP/s', 833.3e12, 'peak = 2500 / 3'),
    ('NT GEMMs on six bf16 MFMAs per product (what is left there: skinny N = 32 products, sub-128-tile launches)', ('gemm_nt', 'splitk'), 0.0104e12, 'TFLOP/s', 416.7e12, 'peak = 2500 / 6; launch-bound sizes'),
    ('TN GEMMs on three fp16 MFMAs per product (`gemm_tn_g3_kernel`)', ('gemm_tn_g3',), 1.28e12, 'TFLOP/s', 833.3e12, 'weight / bias gradients of the 256-tile shapes'),
    ('TN GEMMs on six products (`gemm_tn_x6*`: the small products of the step, grouped launches)', ('gemm_tn',), 0.02e12, 'TFLOP/s', 416.7e12, ''),
    ('`add_ln_bwd` (reads dy, s; writes d_s, d_r; mask regenerated)', ('add_ln_bwd',), 4.0 * ROWS_LN * 1024, 'TB/s', 8e12, '4 streams of rows x 1 KB'),
    ('`add_ln_fwd` (reads the residual sum s, writes y)', ('add_ln_fwd',), 2.0 * ROWS_LN * 1024, 'TB/s', 8e12, '2 streams'),
    ('`relattn16_bwd`', ('relattn16_bwd',), 2.57e9, 'TB/s', 8e12, ''),
    ('`relattn_sub16_bwd`', ('relattn_sub16_bwd',), 2.65e9, 'TB/s', 8e12, ''),
    ('`relattn16_fwd`', ('relattn16_fwd',), 0.86e9, 'TB/s', 8e12, 'q k v gathered from the L2-resident block table'),
    ('`relattn_sub16_fwd`', ('relattn_sub16_fwd',), 1.6e9, 'TB/s', 8e12, ''),
    ('L = 4 attention (`relattn_fwd/bwd`, `relattn_sub_fwd/bwd`)', ('relattn_bwd_kernel', 'relattn_fwd_kernel', 'relattn_sub_bwd_kernel', 'relattn_sub_fwd_kernel'), None, '', None, ''),
    ('`block_table_segsum`', ('segsum',), 1.71e9, 'TB/s', 8e12, ''),
    ('`embed_pos_fwd/bwd` (+ scatter)', ('embed_pos',), None, '', None, ''),
    ('`reduce_splits*` (deterministic partial sums)', ('reduce_splits',), None, '', None, 'launch / L2-bound'),
    ('GRU step kernels (`gru_step_fwd/bwd`, one launch per time step) + `gru_cell_bwd`', ('gru_',), None, '', None, 'latency'),
    ('torch glue (adds, fills, cat, copies)', ('at6native', 'rocclr', 'elementwise_kernel_with_index'), None, '', None, 'launch-bound'),
    ('`vq_fwd` / `vq_bwd`', ('vq_fwd', 'vq_bwd'), None, '', None, ''),
    ('`nce_fwd` / `nce_bwd` / `nce_dw_mfma`', ('nce_',), None, '', None, ''),
]


def main(path):
    rows = [r for r in csv.reader(open(path))][1:]
    steps = sum(int(r[1]) for r in rows if 'adam_dev' in r[0] or 'adam_kernel' in r[0])     # replayed + eager warm-up steps in the trace
    used = set()
    out = []
    for label, match, work, unit, peak, note in FAMILIES:
        n = ms = 0
        for i, r in enumerate(rows):
            if i in used or r[0].startswith('TOTAL') or r[0].startswith('trace'):
                continue
            if any(m in r[0] for m in match):
                used.add(i); n += int(r[1]); ms += float(r[2])
        if not n:
            continue
        per = ms / steps
        if work:
            rate = work / (per * 1e-3)
            ach = f'{work / 1e12:.2f} TFLOP -> {rate / 1e12:.0f} TFLOP/s' if unit == 'TFLOP/s' else f'{work / 1e9:.2f} GB -> {rate / 1e12:.1f} TB/s'
            frac = f'{rate / peak:.2f}'
            gb = next((v for k, v in GEMM_BYTES.items() if k in match), None)
            if gb:       # two ceilings: the binding one is the larger floor
                t_m, t_h = work / peak * 1e3, gb / 8e12 * 1e3
                note = (f'{note}; MFMA floor {t_m:.2f} ms, HBM floor {t_h:.2f} ms ({gb / 1e9:.1f} GB algorithmic, {gb / (per * 1e-3) / 1e12:.2f} TB/s achieved) '
                        f'-> bound: {"hbm" if t_h >= t_m else "mfma"}, **{max(t_m, t_h) / per:.2f} of the binding ceiling**')
        else:
            ach, frac = note or '-', '-'
        out.append((label, n / steps, per, ach, frac, note if work else ''))
    rest_n = sum(int(r[1]) for i, r in enumerate(rows) if i not in used and not r[0].startswith(('TOTAL', 'trace')))
    rest_ms = sum(float(r[2]) for i, r in enumerate(rows) if i not in used and not r[0].startswith(('TOTAL', 'trace')))
    out.append(('everything else (upscaler activation, Adam / clip, transposes, dropout masks, codeword counts, token check, ...)', rest_n / steps, rest_ms / steps, 'latency', '-', ''))
    tot_n = sum(o[1] for o in out); tot_ms = sum(o[2] for o in out)
    print(f'# Per-kernel time per step at C1 (1 x MI355X, B = 256, training defaults: bf16x6 mode with the f16x3 kernels for the whole-round products, graph replay) -- from `{path}` ({steps} steps)\n')
    print('| kernel family | launches / step | ms / step | achieved | of peak | note |')
    print('|---|---|---|---|---|---|')
    for label, n, per, ach, frac, note in out:
        print(f'| {label} | {n:.0f} | {per:.3f} | {ach} | {frac} | {note} |')
    gemm_ms = sum(o[2] for o in out if 'GEMMs' in o[0])
    print(f'\nTotal: {tot_n:.0f} launches and {tot_ms:.2f} ms of kernel time per step (GEMMs {gemm_ms:.2f} ms = {100 * gemm_ms / tot_ms:.0f} %).')


if __name__ == '__main__':
    main(sys.argv[1])
