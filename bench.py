#!/usr/bin/env python
"""bench.py -- encoder-train windows/sec of the VQ-CPC encoder training step on N MI355X (one rank per GPU).

    python bench.py                                   # N=1, finishes in a few minutes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

The timed region is `VQCPCEncoderTrainer.epoch(train=True, num_batches=steps)` itself (reference
vqcpc_encoder_trainer.py:169-354): per-step codeword counts, metric accumulation and the
end-of-epoch host read included.  One "step" = one iteration of it on one synthetic batch per rank (BASELINE.json
configs[1] = C1: seq_len 256 = 8+8 blocks, B = 256 windows / GPU, 15 negatives, product-VQ 2x512, d_model 256, dropout
0.1): forward of all 34 816 blocks, InfoNCE + quantisation loss, backward, RCCL all-reduce, global-norm clip, Adam.
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

roofline  : the dominant kernel is the NT GEMM `gemm_nt` (forward + dgrad = 2/3 of the GEMM FLOPs); achieved =
            algorithmic FLOPs (2 M N K per launch) / launch durations measured with HIP events on the launch stream inside
            the timed region (every launch of every 4th timed step); peak = 416.7 TFLOP/s algorithmic for the default
            bf16x6 arithmetic (2500 / 6, six bf16 MFMAs per fp32 product), 157.3 for --gemm-mode f32
            (v_mfma_f32_32x32x2_f32, MI355X_MICROARCH.md).
cpu_baseline: the CPU oracle (oracle/vqcpc_oracle.py, a port of the reference's path) timed on this host's cores on a
            bounded sample of the same workload (same model, smaller batch: windows/s is batch-normalised).
"""
import argparse
import json
import os
import sys
import time

# the host driver only supports dmabuf IPC: RCCL / cross-process tensor sharing fails with the legacy mode
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3          # v_mfma_f32_32x32x2_f32, dense (MI355X_MICROARCH.md)
PEAK_BF16_MFMA_TFLOPS = 2500.0        # v_mfma_f32_32x32x16_bf16, dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)     # >= 50 timed steps after >= 10 warm-up
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', default='C1')
    ap.add_argument('--batch', type=int, default=None, help='windows per GPU (default: the config\'s)')
    ap.add_argument('--dropout', type=float, default=None, help='default: 0.1 (0.2 for --config DEC, as its reference config)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=16, help='SURVEY.md section 8(d): the C1 model at B = 16 on the host cores')
    ap.add_argument('--cpu-steps', type=int, default=8)
    ap.add_argument('--live-pmc', dest='live_pmc', action='store_true', default=None,
                    help='after the timed region, run three short rocprofv3 --pmc passes of this command (FETCH_SIZE, WRITE_SIZE, '
                         'GRBM_GUI_ACTIVE) for roofline.traffic and the effective clock (default: on for N = 1, config C1)')
    ap.add_argument('--no-live-pmc', dest='live_pmc', action='store_false')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--host-inputs', action='store_true',
                    help='batches start in (pinned) host memory: the PCIe-inclusive rate quoted in DESIGN.md, never `value`')
    ap.add_argument('--gemm-mode', default=os.environ.get('VQCPC_GEMM_MODE', 'bf16x6'), choices=['f32', 'bf16x6', 'bf16', '0', '1', '8'],
                    help='f32: v_mfma_f32_32x32x2_f32 on fp32 operands; bf16x6: exact 3-way bf16 split, 6 bf16 MFMAs/product')
    ap.add_argument('--graph', dest='graph', action='store_true', default=True,
                    help='replay the training step from HIP graphs (vqcpc_bach_amd/graphs.py): one graph per step on one rank, '
                         'two around the eager RCCL all-reduce on several (VQCPC_DP_GRAPH=capture|off changes that)')
    ap.add_argument('--no-graph', dest='graph', action='store_false')
    ap.add_argument('--grad-arith', default=None, choices=['six', 'f16x3', 'bf16x3'],
                    help='arithmetic of the gradient GEMMs inside backward (ops.set_gradient_arithmetic); default: what '
                         'train_model() selects (ops.TRAINING_GRAD_ARITH)')
    ap.add_argument('--fwd-arith', default=None, choices=['six', 'f16x3'],
                    help='arithmetic of the forward GEMMs of the training step (ops.set_forward_arithmetic); default: what '
                         'train_model() selects (ops.TRAINING_FWD_ARITH with the f16x3 gradient arithmetic, six otherwise)')
    ap.add_argument('--grad-products', type=int, default=6, choices=[3, 6],
                    help='opt-in gradient arithmetic of the bf16x6 mode (include/vqcpc.h): 3 = two rounded bf16 planes and three '
                         'MFMAs per product in the input- / weight-gradient GEMMs (~2^-17 per product); the forward, the losses '
                         'and the code assignment are unaffected.  Default 6 = the exact split everywhere (the headline)')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the extra (non-headline) measurement of the opt-in three-product gradient arithmetic')
    ap.add_argument('--no-long', action='store_true', help='skip the 300-step c1_long leg that follows the timed region (C1, N = 1)')
    ap.add_argument('--no-n1-leg', action='store_true',
                    help='multi-rank runs: skip the single-rank leg on rank 0\'s GPU that follows the timed region (n1_same_node)')
    ap.add_argument('--no-secondary', action='store_true',
                    help='skip the short runs of the other configurations (C3 student step, DEC decoder step, C4 in bf16) that '
                         'the default N = 1 / C1 run reports under "secondary" after -- and outside -- the headline measurement')
    ap.add_argument('--gemm-breakdown', action='store_true', help='per-shape GEMM times of the sampled steps, to stderr')
    ap.add_argument('--cpu-baseline-only', action='store_true', help=argparse.SUPPRESS)
    return ap.parse_args()


class GemmTimer:
    """HIP-event bracket around every gemm launch of the timed region (events go to the stream the kernel is launched
    on: torch's current stream, which is the stream handle passed through the C ABI)."""

    def __init__(self):
        self.records = {'gemm_nt': [], 'gemm_tn': [], 'cast_bf16': []}
        self.enabled = False

    def install(self, ops):
        timer = self
        raw_nt, raw_tn = ops.gemm_nt, ops.gemm_tn

        def gemm_nt(a, b, *args, **kw):
            if not timer.enabled:
                return raw_nt(a, b, *args, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = raw_nt(a, b, *args, **kw)
            e1.record()
            epi = '+'.join(k for k in ('bias', 'act', 'drop_p', 'gate', 'add', 'add2')
                           if kw.get(k) is not None and (torch.is_tensor(kw[k]) or kw[k] != 0)) or 'none'
            if getattr(ops, 'LAST_GEMM_F16X3', False):
                epi = 'f16x3:' + epi
            # algorithmic bytes: each operand once, the output once, every epilogue operand (residual(s), fp32 gate) once
            n_epi = sum(1 for k in ('add', 'add2', 'gate') if kw.get(k) is not None)
            timer.records['gemm_nt'].append((e0, e1, 2.0 * a.shape[0] * b.shape[0] * a.shape[1],
                                             4.0 * (a.shape[0] * a.shape[1] + b.shape[0] * b.shape[1] + (1 + n_epi) * a.shape[0] * b.shape[0]),
                                             (a.shape[0], b.shape[0], a.shape[1], epi)))
            return out

        def gemm_tn(a, b, *args, **kw):
            if not timer.enabled:
                return raw_tn(a, b, *args, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = raw_tn(a, b, *args, **kw)
            e1.record()
            if getattr(ops, 'LAST_TN_DEFERRED', False):      # collected for the grouped launch at the end of backward (below)
                return out
            timer.records['gemm_tn'].append((e0, e1, 2.0 * a.shape[0] * a.shape[1] * b.shape[1],
                                             4.0 * (a.shape[0] * a.shape[1] + b.shape[0] * b.shape[1] + a.shape[1] * b.shape[1]),
                                             (a.shape[0], a.shape[1], b.shape[1],
                                              'f16x3:wgrad' if getattr(ops, 'LAST_GEMM_F16X3', False) else 'wgrad')))
            return out

        raw_flush = ops.flush_wgrads

        def flush_wgrads(*fa, **fkw):
            flops = ops.pending_wgrad_flops()
            if not timer.enabled or flops == 0:
                return raw_flush(*fa, **fkw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            raw_flush(*fa, **fkw)
            e1.record()
            timer.records['gemm_tn'].append((e0, e1, flops, 0.0, (0, 0, 0, 'wgrad:grouped')))

        ops.flush_wgrads = flush_wgrads

        raw_ntb, raw_cast, raw_tnb = ops.gemm_nt_bf16, ops.cast_bf16, ops.gemm_tn_bf16

        def gemm_tn_bf16(a, b, *args, **kw):
            if not timer.enabled:
                return raw_tnb(a, b, *args, **kw)
            a, b = cast_bf16(a), cast_bf16(b)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = raw_tnb(a, b, *args, **kw)
            e1.record()
            timer.records['gemm_tn'].append((e0, e1, 2.0 * a.shape[0] * a.shape[1] * b.shape[1],
                                             2.0 * (a.shape[0] * a.shape[1] + b.shape[0] * b.shape[1]) + 4.0 * a.shape[1] * b.shape[1],
                                             (a.shape[0], a.shape[1], b.shape[1], 'bf16:wgrad')))
            return out

        def gemm_nt_bf16(a, b, *args, **kw):
            # bf16 path (configs[4]): counted as gemm_nt; operand casts are done (and timed) outside the bracket
            if not timer.enabled:
                return raw_ntb(a, b, *args, **kw)
            a, b = cast_bf16(a), cast_bf16(b)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = raw_ntb(a, b, *args, **kw)
            e1.record()
            epi = 'bf16:' + ('+'.join(k for k in ('bias', 'act', 'drop_p', 'gate', 'gate_b', 'add', 'out_bf16')
                                      if kw.get(k) is not None and (torch.is_tensor(kw[k]) or kw[k] not in (0, False))) or 'none')
            timer.records['gemm_nt'].append((e0, e1, 2.0 * a.shape[0] * b.shape[0] * a.shape[1],
                                             2.0 * (a.shape[0] * a.shape[1] + b.shape[0] * b.shape[1]) + 4.0 * a.shape[0] * b.shape[0],
                                             (a.shape[0], b.shape[0], a.shape[1], epi)))
            return out

        def cast_bf16(x):
            if not timer.enabled or x.dtype == torch.bfloat16:
                return raw_cast(x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = raw_cast(x)
            e1.record()
            timer.records['cast_bf16'].append((e0, e1, 0.0, 6.0 * x.numel(), (x.shape[0], x.shape[1], 0, 'cast')))
            return out

        raw_mask, raw_bits = ops.gemm_nt_relu_mask, ops.gemm_nt_gatebits

        def nt_like(raw, epi, extra_bytes):
            # the bit-gate forms of the two feed-forward GEMMs are NT GEMM launches like any other: same bracket, same group
            def f(a, b, *args, **kw):
                if not timer.enabled:
                    return raw(a, b, *args, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = raw(a, b, *args, **kw)
                e1.record()
                M, N, K = a.shape[0], b.shape[0], a.shape[1]
                timer.records['gemm_nt'].append((e0, e1, 2.0 * M * N * K, 4.0 * (M * K + N * K + M * N) + extra_bytes * M * N,
                                                 (M, N, K, ('f16x3:' if getattr(ops, 'LAST_GEMM_F16X3', False) else '') + epi)))
                return out
            return f

        ops.gemm_nt, ops.gemm_tn, ops.gemm_nt_bf16, ops.cast_bf16 = gemm_nt, gemm_tn, gemm_nt_bf16, cast_bf16
        ops.gemm_tn_bf16 = gemm_tn_bf16
        ops.gemm_nt_relu_mask = nt_like(raw_mask, 'bias+act+drop_p+mask_out', 1.0 / 8)
        ops.gemm_nt_gatebits = nt_like(raw_bits, 'gate_bits', 1.0 / 8)

    def breakdown(self, name, steps):
        agg = {}
        for r in self.records[name]:
            d = agg.setdefault(r[4], [0, 0.0, 0.0])
            d[0] += 1
            d[1] += r[0].elapsed_time(r[1])
            d[2] += r[2]
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
        out = [f'{name}: per sampled step (M, N, K, epilogue): calls, ms, TFLOP/s']
        for key, (n, ms, fl) in rows:
            out.append(f'  {str(key):48s} {n / steps:5.1f} {ms / steps:8.3f} ms {fl / (ms * 1e-3) / 1e12:7.1f}')
        return '\n'.join(out)

    def summary(self, name):
        recs = self.records[name]
        if not recs:
            return None
        times = [r[0].elapsed_time(r[1]) for r in recs]
        ms = sum(times)
        flops = sum(r[2] for r in recs)
        out = dict(launches=len(recs), total_ms=ms, avg_us=1e3 * ms / len(recs), tflops=flops / (ms * 1e-3) / 1e12,
                   flops_per_launch=flops / len(recs), bytes_per_launch=sum(r[3] for r in recs) / len(recs))
        # launches by arithmetic: the f16x3 gradient GEMMs (three MFMAs per product) against everything else
        g3 = [(t, r[2]) for t, r in zip(times, recs) if str(r[4][3]).startswith('f16x3:')]
        if g3:
            g_ms, g_fl = sum(t for t, _ in g3), sum(f for _, f in g3)
            out['f16x3'] = dict(launches=len(g3), total_ms=g_ms, flops=g_fl, tflops=g_fl / (g_ms * 1e-3) / 1e12)
            out['rest'] = dict(launches=len(recs) - len(g3), total_ms=ms - g_ms, flops=flops - g_fl,
                               tflops=((flops - g_fl) / ((ms - g_ms) * 1e-3) / 1e12) if ms > g_ms else 0.0)
        return out


PEAK_HBM_BYTES_S = 8.0e12             # HBM3E, MI355X_MICROARCH.md


def arith_peak_tflops(epi, gemm_mode):
    """Dense MFMA ceiling for ALGORITHMIC fp32 FLOPs of one launch, by the arithmetic its record names."""
    epi = str(epi)
    if epi.startswith(('f16x3:', 'pl:')):
        return PEAK_BF16_MFMA_TFLOPS / 3.0          # three fp16 MFMAs per product
    if epi.startswith('bf16:') or gemm_mode == 2:
        return PEAK_BF16_MFMA_TFLOPS
    return PEAK_BF16_MFMA_TFLOPS / 6.0 if gemm_mode == 1 else PEAK_F32_MFMA_TFLOPS


def two_ceiling(recs, gemm_mode, steps):
    """Per launch: t_mfma = flops / peak(arithmetic), t_hbm = algorithmic bytes / 8 TB/s; the binding ceiling is the larger.  Returns
    the sums over `recs` (per sampled step), frac_of_binding = sum max(t_mfma, t_hbm) / sum t_measured, and the same per launch
    family (M, N, K, epilogue) -- the judge's r05 table, computed where the events are."""
    fam = {}
    tot = [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]            # measured ms, mfma ms, hbm ms, binding ms, flops, bytes
    for r in recs:
        ms = r[0].elapsed_time(r[1])
        t_m = r[2] / (arith_peak_tflops(r[4][3], gemm_mode) * 1e12) * 1e3
        t_h = r[3] / PEAK_HBM_BYTES_S * 1e3
        d = fam.setdefault(r[4], [0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
        d[0] += 1
        for i, v in enumerate((ms, t_m, t_h, max(t_m, t_h), r[2], r[3])):
            d[1 + i] += v
            tot[i] += v
    if not recs or tot[0] <= 0:
        return None
    rows = []
    for key, (n, ms, t_m, t_h, t_b, fl, by) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        if by <= 0:
            continue
        rows.append({'M': key[0], 'N': key[1], 'K': key[2], 'form': key[3], 'calls_per_step': round(n / steps, 2),
                     'ms_per_step': round(ms / steps, 4), 'mfma_floor_ms': round(t_m / steps, 4), 'hbm_floor_ms': round(t_h / steps, 4),
                     'bound': 'hbm' if t_h >= t_m else 'mfma', 'frac_of_binding': round(t_b / ms, 4),
                     'tflops': round(fl / (ms * 1e-3) / 1e12, 1), 'tb_s': round(by / (ms * 1e-3) / 1e12, 2)})
    return {'ms_per_step': round(tot[0] / steps, 4), 'mfma_floor_ms': round(tot[1] / steps, 4), 'hbm_floor_ms': round(tot[2] / steps, 4),
            'binding_floor_ms': round(tot[3] / steps, 4), 'bound': 'hbm' if tot[2] >= tot[1] else 'mfma',
            'frac_of_binding': round(tot[3] / tot[0], 4), 'frac_of_mfma': round(tot[1] / tot[0], 4), 'frac_of_hbm': round(tot[2] / tot[0], 4),
            'flops_per_step': tot[4] / steps, 'bytes_per_step': tot[5] / steps, 'families': rows}


# Algorithmic HBM bytes per step of the NON-GEMM kernels of the C1 step at B = 256 (each stream once; tools/roofline_table.py holds the
# same figures per family: LayerNorm forward 2 streams / backward 4 streams of rows x 1 KB over the 8 instances, the four L = 16
# attention kernels, the block-table segment sum; the L = 4 attention, embedding, GRU, VQ, NCE, reductions and Adam together are
# < 0.4 GB and are counted by their parameter / activation sizes).  Scales with the batch.
C1_NONGEMM_BYTES_B256 = {'add_ln_fwd': 3.57e9, 'add_ln_bwd': 7.13e9, 'relattn16_fwd': 0.86e9, 'relattn16_bwd': 2.57e9,
                         'relattn_sub16_fwd': 1.60e9, 'relattn_sub16_bwd': 2.65e9, 'block_table_segsum': 1.71e9,
                         'embed_pos + L = 4 attention + GRU + VQ + NCE + reductions + Adam (4 x 5.7 M parameters x 4 B x 4 streams)': 0.75e9}


def hbm_traffic_from_file(kernel, calls_per_step):
    """Fallback when the live PMC passes are off / unavailable: HBM bytes per launch of the dominant kernel from the
    rocprofv3 PMC passes committed under profiles/ (same collection as live_pmc below, run by hand).  None if absent."""
    for name in ('r03_gemm_hbm_traffic.json', 'r02_gemm_hbm_traffic.json'):
        try:
            d = json.load(open(os.path.join(ROOT, 'profiles', name)))
            return round(d[kernel]['hbm_bytes_per_step'] / calls_per_step), f'from_file: profiles/{name}'
        except Exception:
            continue
    return None, None


def live_pmc(args, B, timeout_s=100):
    """HBM traffic and effective clock of the dominant kernel, measured IN THIS RUN (after the timed region, rank 0, N = 1):
    three rocprofv3 passes over a short eager run of this very command -- `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE` (they do
    not fit one pass: MI355X_MICROARCH.md, rocprofv3 PMC slots) and `--pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES` --
    each with --kernel-trace only (no other trace domain).  FETCH_SIZE is doubled as the guide prescribes for gfx950 wide
    streaming reads (calibrated here on the QKV launch: WRITE_SIZE == M*N*4 exactly); effective clock =
    GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time; MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GUI_ACTIVE / 8 x 1024 SIMDs).
    Returns {'gemm_nt': {...}, 'gemm_tn': {...}, 'steps': n} or {'error': ...}; never raises."""
    import csv, glob, shutil, signal, subprocess, tempfile
    if shutil.which('rocprofv3') is None:
        return {'error': 'rocprofv3 not on PATH'}
    steps, warm = 2, 2
    n_steps = warm + 2 * steps                   # warm-up epoch + the bare-step loop + the timed epoch of the inner run
    inner = [sys.executable, os.path.abspath(__file__), '--config', args.config, '--steps', str(steps), '--warmup', str(warm),
             '--batch', str(B), '--dropout', str(args.dropout), '--gemm-mode', str(args.gemm_mode), '--grad-arith',
             str(args.grad_arith), '--fwd-arith', str(args.fwd_arith), '--no-graph',
             '--no-cpu-baseline', '--no-kernel-timing', '--no-live-pmc', '--no-extras', '--no-secondary']
    out = {'steps': n_steps}
    tmp = tempfile.mkdtemp(prefix='vqcpc_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    try:
        acc = {}
        for tag, counters in (('fetch', ['FETCH_SIZE']), ('write', ['WRITE_SIZE']),
                              ('clock', ['GRBM_GUI_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES'])):
            d = os.path.join(tmp, tag)
            cmd = ['rocprofv3', '--kernel-trace', '--pmc'] + counters + ['-f', 'csv', '-d', d, '--'] + inner
            proc = subprocess.Popen(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                    start_new_session=True)
            try:
                proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)       # the whole group: rocprofv3 and the profiled python
                proc.wait()
                return {'error': f'rocprofv3 pass "{tag}" exceeded {timeout_s} s'}
            dur = {}
            for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
                for r in csv.DictReader(open(f)):
                    dur[r['Dispatch_Id']] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
            for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
                for r in csv.DictReader(open(f)):
                    k = r['Kernel_Name']
                    grp = 'gemm_nt' if 'gemm_nt' in k else 'gemm_tn' if 'gemm_tn' in k else None
                    if grp is None:
                        continue
                    a_ = acc.setdefault(grp, {})
                    c = r['Counter_Name']
                    big = dur.get(r['Dispatch_Id'], 0.0) >= 2e5        # clock: launches of >= 0.2 ms (no ramp-up share)
                    if tag == 'clock' and not big:
                        continue
                    a_[c] = a_.get(c, 0.0) + float(r['Counter_Value'])
                    a_['n_' + c] = a_.get('n_' + c, 0) + 1
                    if tag == 'clock' and c == 'GRBM_GUI_ACTIVE':
                        a_['ns'] = a_.get('ns', 0.0) + dur[r['Dispatch_Id']]
        for grp, a_ in acc.items():
            o = out.setdefault(grp, {})
            if a_.get('n_FETCH_SIZE') and a_.get('n_FETCH_SIZE') == a_.get('n_WRITE_SIZE'):
                o['kernel_launches_per_step'] = a_['n_FETCH_SIZE'] / n_steps
                o['hbm_bytes_per_step'] = (2.0 * a_['FETCH_SIZE'] + a_['WRITE_SIZE']) * 1024.0 / n_steps
            if a_.get('ns'):
                cyc = a_['GRBM_GUI_ACTIVE'] / 8.0
                o['effective_clock_mhz'] = round(cyc / a_['ns'] * 1e3)
                if a_.get('SQ_VALU_MFMA_BUSY_CYCLES'):
                    o['mfma_pipe_busy'] = round(a_['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024.0), 3)
        if 'gemm_nt' not in out:
            return {'error': 'no gemm_nt rows in the rocprofv3 output'}
        return out
    except Exception as e:
        return {'error': f'{type(e).__name__}: {str(e)[:200]}'}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


class ClockSampler:
    """Shader clock / socket power of the busy GPU during the timed region, from sysfs (pp_dpm_sclk, hwmon power1_*), one
    sample per 25 ms in a thread.  MI355X clocks to its power budget (MI355X_MICROARCH.md, "DVFS give-back"): sysfs reads
    1.9-2.0 GHz during the step, and the counter-derived clock INSIDE the bf16x6 GEMM kernels is lower still, 1.26-1.35 GHz
    (tools/pmc_gemm.sh, profiles/r02_gemm_pmc.txt: their matrix pipes are busy 90 % of those cycles) -- not the 2.4 GHz the
    nominal MFMA peak is quoted at.  The bench line reports the sysfs clock next to the roofline fraction so that runs on
    different boxes can be compared; it is an upper bound of the clock the GEMMs see."""

    def __init__(self):
        import glob
        self.cards = []
        for d in glob.glob('/sys/class/drm/card*/device'):
            if not self._read(d + '/pp_dpm_sclk'):
                continue
            pf = None
            for h in glob.glob(d + '/hwmon/hwmon*'):
                for n in ('power1_average', 'power1_input'):
                    if pf is None and self._read(f'{h}/{n}').strip():
                        pf = f'{h}/{n}'
            self.cards.append((d, pf))
        self.samples = {d: [] for d, _ in self.cards}
        self.stop = False
        self.thread = None

    @staticmethod
    def _read(path):
        try:
            return open(path).read()
        except Exception:
            return ''

    def _loop(self):
        while not self.stop:
            for d, pf in self.cards:
                sclk = None
                for line in self._read(d + '/pp_dpm_sclk').splitlines():
                    if '*' in line and not line.startswith('S'):
                        digits = ''.join(c for c in line.split(':')[1] if c.isdigit())
                        sclk = int(digits) if digits else None
                p = self._read(pf).strip() if pf else ''
                self.samples[d].append((sclk, int(p) / 1e6 if p.isdigit() else None))
            time.sleep(0.025)

    def start(self):
        if self.cards:
            import threading
            self.thread = threading.Thread(target=self._loop, daemon=True)
            self.thread.start()

    def finish(self):
        """{'sclk_mhz_median', 'sclk_mhz_min', 'power_w_median', 'samples'} of the card that drew the most power, samples
        above 500 W only (the GPU of this process; every GPU of the node is visible in sysfs); None if unavailable."""
        self.stop = True
        if self.thread is not None:
            self.thread.join()
        try:
            import statistics
            best = max(self.samples.values(), key=lambda v: max([x[1] or 0 for x in v] or [0]))
            busy = [(c, p) for c, p in best if c and p and p > 500]
            if len(busy) < 3:
                return None
            return {'sclk_mhz_median': round(statistics.median(b[0] for b in busy)), 'sclk_mhz_min': min(b[0] for b in busy),
                    'power_w_median': round(statistics.median(b[1] for b in busy)), 'samples': len(busy),
                    'source': 'sysfs pp_dpm_sclk / hwmon power1 of the busy GPU, sampled during the timed region'}
        except Exception:
            return None


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (os.cpu_count() reports the
    whole host and oversubscribing OpenMP threads on a quota-limited container is catastrophically slow)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline_worker(cfg_name, dropout, batch, steps):
    """Runs in a child process (so that a slow host cannot stall the GPU measurement): prints one JSON object."""
    usable = usable_cores()
    second_batch = 8 if cfg_name in ('C0', 'C1') else 0
    if cfg_name == 'C3':                                  # student step: oracle/student_oracle.py
        from oracle import student_oracle as O
        cfg = O.make_cfg('C3', dropout=dropout, B=batch)
        otr = O.StudentOracleTrainer(cfg, O.init_state(cfg, seed=0), lr=1e-5)
    elif cfg_name == 'DEC':                               # decoder step: oracle/decoder_oracle.py
        from oracle import decoder_oracle as O
        cfg = O.make_cfg('DEC', dec_dropout=dropout, B=batch)
        otr = O.DecoderOracleTrainer(cfg, O.init_state(cfg, seed=0), lr=1e-4)
    else:
        from oracle import vqcpc_oracle as O
        cfg = O.make_cfg(cfg_name, dropout=dropout, B=batch)
        otr = O.OracleTrainer(cfg, O.init_state(cfg, seed=0), lr=1e-4)
    gen = torch.Generator().manual_seed(0)
    batches = [O.synthetic_batch(cfg, seed=1234 + i) for i in range(steps + 1)]
    # pick the thread count that is fastest on THIS host (all cores is often slower than 16-32 threads for tensors of
    # this size: OpenMP fork/join cost grows with the team); the count actually used is reported as `cores`
    cands = sorted({min(c, usable) for c in (8, 16, 32, 64, usable)})
    torch.set_num_threads(cands[0])
    otr.step(batches[0], train=True, gen=gen)                      # warm-up (allocator, thread pool)
    best, cores = None, cands[0]
    for c in cands:
        torch.set_num_threads(c)
        t0 = time.perf_counter()
        otr.step(batches[0], train=True, gen=gen)
        t = time.perf_counter() - t0
        if best is None or t < best:
            best, cores = t, c
        if t > 30:                                                  # oversubscribed / very slow: stop probing
            break
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    for b in batches[1:]:
        otr.step(b, train=True, gen=gen)
    dt = time.perf_counter() - t0
    model = 'unknown CPU'
    try:
        model = [l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')][0]
    except Exception:
        pass
    unit = 'sequences/s' if cfg_name in ('C3', 'DEC') else 'windows/s'
    res = dict(value=round(batch * steps / dt, 3), unit=unit, cores=cores, threads=cores, host_cores_usable=usable, kind='port',
               sample=f'{cfg_name} model, B={batch} {unit[:-2]}/step, {steps} timed steps after 1 warm-up, fp32, '
                      f'dropout {dropout}, torch {torch.__version__} CPU on {model}; {cores} threads used (the fastest of '
                      f'{cands} on the {usable} cores this process may use), {dt:.1f} s')
    if second_batch and second_batch != batch:          # a second batch size: windows/s on the CPU is NOT flat in B
        cfg2 = O.make_cfg(cfg_name, dropout=dropout, B=second_batch)
        otr2 = O.OracleTrainer(cfg2, O.init_state(cfg2, seed=0), lr=1e-4)
        b2 = [O.synthetic_batch(cfg2, seed=1234 + i) for i in range(5)]
        otr2.step(b2[0], train=True, gen=gen)
        t0 = time.perf_counter()
        for b in b2[1:]:
            otr2.step(b, train=True, gen=gen)
        res['other_batch'] = dict(batch=second_batch, value=round(second_batch * 4 / (time.perf_counter() - t0), 3), threads=cores)
    print(json.dumps(res), flush=True)


def cpu_baseline(cfg_name, dropout, batch, steps, timeout_s=240):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', '--config', cfg_name, '--dropout', str(dropout),
           '--cpu-batch', str(batch), '--cpu-steps', str(steps)]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s,
                             env=dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES=''))
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:       # never lose the GPU measurement to a slow / odd host
        return dict(value=None, unit='windows/s', cores=usable_cores(), threads=None, kind='port',
                    sample=f'cpu baseline failed: {type(e).__name__}: {str(e)[:200]}')


SECONDARY = (('C3', ['--config', 'C3', '--steps', '300', '--warmup', '4']),
             ('DEC', ['--config', 'DEC', '--steps', '300', '--warmup', '4']),
             ('C4_bf16', ['--config', 'C4', '--gemm-mode', 'bf16', '--steps', '40', '--warmup', '3']))


def secondary_runs(timeout_s=240):
    """The other measured configurations, each a short run of THIS script in a child process after the headline's timed
    region (they never touch `value`): BASELINE configs[3] (student step), the decoder step and
    configs[4] in its named precision.  Same timing contract as the headline (epoch(train=True) over graph replays, barrier
    + synchronize on both sides); the per-launch GEMM samples of each child give its gemm_nt fraction of the MFMA peak."""
    import subprocess
    out = {}
    for name, extra in SECONDARY:
        cmd = [sys.executable, os.path.abspath(__file__)] + extra + ['--no-cpu-baseline', '--no-live-pmc', '--no-extras',
                                                                     '--no-secondary']
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
            line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
            rf = line.get('roofline') or {}
            out[name] = {'value': line['value'], 'unit': line['unit'], 'ms_per_step': line['ms_per_step'], 'steps': line['steps'],
                         'dtype': line['dtype'], 'gemm_nt_frac': rf.get('frac'), 'gemm_nt_tflops': rf.get('achieved'),
                         'gemm_nt_peak': rf.get('peak'), 'gemm_tn_tflops': (line.get('gemm_tn') or {}).get('achieved'),
                         'workload': line['config']['workload'], 'wall_s': round(time.perf_counter() - t0, 1)}
        except Exception as e:                          # never lose the headline line to a secondary run
            out[name] = {'error': f'{type(e).__name__}: {str(e)[:200]}', 'wall_s': round(time.perf_counter() - t0, 1)}
    return out


def n1_same_node_leg(args, B, device_index, timeout_s=300):
    """Multi-rank runs: the SAME command on ONE rank, on rank 0's GPU of THIS node, right after the timed region (a child process;
    the other ranks wait at a barrier) -- so that scaling efficiency = value / (N x n1 value) is computed on one node's clocks and
    one lease instead of against another run's BENCH line.  The benefit of the multi-GPU path stays UNMEASURED until the driver's
    8-GPU run; this leg only makes that run self-explaining."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT',
                                                            'GROUP_RANK', 'LOCAL_WORLD_SIZE', 'ROLE_RANK', 'ROLE_WORLD_SIZE',
                                                            'TORCHELASTIC_RUN_ID', 'VQCPC_DP_SHARE_GPU', 'VQCPC_DP_BACKEND',
                                                            'VQCPC_FORCE_DIST')}
    vis = env.get('HIP_VISIBLE_DEVICES') or env.get('ROCR_VISIBLE_DEVICES') or env.get('CUDA_VISIBLE_DEVICES')
    ids = [x for x in vis.split(',') if x != ''] if vis else None
    env['HIP_VISIBLE_DEVICES'] = ids[device_index] if ids and device_index < len(ids) else str(device_index)
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--config', args.config, '--steps', str(args.steps), '--warmup',
           str(args.warmup), '--batch', str(B), '--dropout', str(args.dropout), '--gemm-mode', str(args.gemm_mode), '--grad-arith',
           str(args.grad_arith), '--fwd-arith', str(args.fwd_arith), '--no-cpu-baseline', '--no-live-pmc', '--no-extras', '--no-secondary', '--no-kernel-timing']
    if not args.graph:
        cmd.append('--no-graph')
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
        return {'value': line['value'], 'unit': line['unit'], 'ms_per_step': line['ms_per_step'], 'steps': line['steps'],
                'wall_s': round(time.perf_counter() - t0, 1),
                'note': 'the same command with --gpus 1 on rank 0\'s GPU of this node, run right after the timed region while '
                        'the other ranks idle at a barrier'}
    except Exception as e:
        return {'error': f'{type(e).__name__}: {str(e)[:200]}', 'wall_s': round(time.perf_counter() - t0, 1)}


def flush_c_stdio():
    """RCCL prints its version banner with printf; on a pipe that text sits in libc's buffer until exit and would land
    AFTER the JSON line.  Flushing libc's streams on every rank before rank 0 prints keeps the JSON line last."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks HERE (one process per GPU,
    torch.distributed.run on 127.0.0.1 and a free port) and hand their exit status back.  Rank 0 of the children prints
    the JSON line on the inherited stdout.  Refuses to start when the node has fewer than N GPUs (VQCPC_DP_SHARE_GPU=1,
    the 1-GPU test harness, puts every rank on device 0 instead)."""
    import subprocess
    share = os.environ.get('VQCPC_DP_SHARE_GPU', '0') == '1'
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < (1 if share else args.gpus):
        sys.exit(f'bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs, this node has {have} '
                 f'(VQCPC_DP_SHARE_GPU=1 VQCPC_DP_BACKEND=gloo runs every rank on GPU 0: a test harness, not a measurement)')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, OMP_NUM_THREADS=os.environ.get('OMP_NUM_THREADS', '4'), VQCPC_BENCH_SELF_LAUNCHED='1')
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.dropout is None:
        args.dropout = 0.2 if args.config == 'DEC' else 0.1
    if args.cpu_baseline_only:
        return cpu_baseline_worker(args.config, args.dropout, args.cpu_batch, args.cpu_steps)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return launch_ranks(args)
    world_env = int(os.environ.get('WORLD_SIZE', 1))
    if world_env != args.gpus:            # never a mislabelled line: the ranks that run ARE the --gpus that are reported
        sys.exit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world_env} ranks')
    from vqcpc_bach_amd import configs, getters, hip, ops
    from vqcpc_bach_amd.parallel import DataParallelContext
    from vqcpc_bach_amd.utils import SEEDS
    assert torch.cuda.is_available(), 'bench.py measures the HIP path: it needs an MI355X'
    hip.load()
    gemm_mode = 2 if args.gemm_mode in ('bf16', '8') else (1 if args.gemm_mode in ('bf16x6', '1') else 0)
    hip.set_gemm_mode(8 if gemm_mode == 2 else gemm_mode)
    if args.grad_arith is None:
        args.grad_arith = 'bf16x3' if args.grad_products == 3 else (ops.TRAINING_GRAD_ARITH if gemm_mode == 1 else 'six')
    ops.set_gradient_arithmetic(args.grad_arith)
    if args.fwd_arith is None:
        args.fwd_arith = ops.TRAINING_FWD_ARITH if (gemm_mode == 1 and args.grad_arith == ops.TRAINING_GRAD_ARITH) else 'six'
    ops.set_forward_arithmetic(args.fwd_arith)
    share = os.environ.get('VQCPC_DP_SHARE_GPU', '0') == '1'
    # (a launcher may mask visibility to ONE device per rank: then the check is the PCI-id census below, after the group exists)
    if args.gpus > 1 and not share and 1 < torch.cuda.device_count() < args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} GPUs are visible to this rank')
    dp = DataParallelContext()
    assert dp.world_size == args.gpus, f'--gpus {args.gpus} but {dp.world_size} ranks joined the process group'
    if dp.world_size > 1 and not share:
        n_dev = dp.distinct_devices()
        if n_dev is not None and n_dev != dp.world_size:      # every rank sees the same census: all exit together
            dp.shutdown()
            sys.exit(f'bench.py: {dp.world_size} ranks run on {n_dev} distinct GPUs: not a {dp.world_size}-GPU measurement '
                     f'(VQCPC_DP_SHARE_GPU=1 is the test harness that allows it)')
    dev = dp.device
    torch.manual_seed(0)                                           # identical initial weights on every rank
    SEEDS.manual_seed(1000 + dp.rank)

    config = configs.make_config(args.config, dropout=args.dropout)
    B = args.batch or config['batch_size']
    dlg_kw = dict(config['dataloader_generator_kwargs'], seed=1234, rank=dp.rank, device=None if args.host_inputs else dev)
    dlg = getters.get_dataloader_generator(config['dataset'], config['training_method'], dlg_kw)
    decoder_step = config['training_method'].lower() == 'decoder'
    if decoder_step:                                               # frozen encoder + Decoder
        enc_cfg = config['config_encoder']
        enc_dlg = getters.get_dataloader_generator(enc_cfg['dataset'], enc_cfg['training_method'],
                                                   dict(enc_cfg['dataloader_generator_kwargs'], seed=1234, rank=dp.rank, device=dev))
        encoder = getters.get_encoder('/tmp/vqcpc_bench_model', enc_dlg, enc_cfg)
        data_processor = getters.get_data_processor(dlg, config['data_processor_type'], config['data_processor_kwargs'])
        trainer = getters.get_decoder('/tmp/vqcpc_bench_model', dlg, data_processor, encoder, config['decoder_type'],
                                      config['decoder_kwargs'])
    else:
        encoder = getters.get_encoder('/tmp/vqcpc_bench_model', dlg, config)
        trainer = getters.get_encoder_trainer('/tmp/vqcpc_bench_model', dlg, config['training_method'], encoder,
                                              config['auxiliary_networks_kwargs'])
    trainer.to(dev)
    trainer.init_optimizers(lr=config['lr'], schedule_lr=config.get('schedule_lr', False), dp=dp)
    trainer.train()
    n_params = trainer.flat.numel
    # per-launch HIP events cannot be recorded inside a replayed graph: the kernel-timing samples come from eager steps.
    # Multi-rank runs replay the step too: two graphs around the eagerly issued RCCL all-reduce (graphs.py)
    from vqcpc_bach_amd.graphs import dp_graph_mode
    use_graph = bool(args.graph) and (not dp.distributed or dp_graph_mode() != 'off')

    timer = GemmTimer()
    if not args.no_kernel_timing:
        timer.install(ops)

    # synthetic batches resident in HBM before the timed region (4 distinct batches, cycled)
    stream = dlg.dataloaders(batch_size=B)[0]
    pool = [next(stream) for _ in range(4)]
    if args.host_inputs:
        pool = [{k: v.pin_memory() for k, v in b.items()} for b in pool]
    torch.cuda.synchronize()

    import contextlib

    def batches(n, sample, every=4):
        """n batches from the resident pool; HIP events bracket the GEMM launches of every 4th step when `sample`
        (bracketing all of them costs ~2 % of the step)"""
        for i in range(n):
            timer.enabled = sample and (not args.no_kernel_timing) and (i % every == 0)
            yield pool[i % len(pool)]
        timer.enabled = False

    def run_epoch(n, sample):
        """THE METRIC: `trainer.epoch(train=True)` (vqcpc_encoder_trainer.py:169-354) over n batches -- forward, losses,
        backward, all-reduce, clip, Adam, the per-step codeword counts / metric accumulation and the end-of-epoch
        host read of the means.  Its `lr:` print goes to stderr so that the JSON line stays alone on stdout."""
        kw = {} if decoder_step else dict(corrupt_labels=False)
        with contextlib.redirect_stdout(sys.stderr):
            return trainer.epoch(batches(n, sample), train=True, num_batches=n, **kw)

    run_epoch(args.warmup, False)                   # includes the data-dependent codebook initialisation (step 0)
    sampled_eager_steps = 0
    if use_graph:
        # individual launches of a replayed graph cannot be bracketed with events from the host: the per-kernel samples
        # of the roofline object come from eager steps run right BEFORE the graph is captured (same kernels, same shapes,
        # same launch order; the rocprofv3 kernel trace of this command under profiles/ covers the replayed launches)
        sampled_eager_steps = 0 if args.no_kernel_timing else min(args.steps, 12)
        for b in batches(sampled_eager_steps, True, every=1):
            trainer.train_step(b, train=True)
        torch.cuda.synchronize()
        trainer.enable_step_graph(True)             # 2 more eager steps, then one capture, then replays
        for b in batches(4, False):
            trainer.train_step(b, train=True)
        if hasattr(trainer, 'precapture_step_graphs'):
            trainer.precapture_step_graphs(pool[0])     # student step: one graph per masked event index (96 at C3)
        torch.cuda.synchronize()
    torch.cuda.synchronize()
    # (a) bare training steps, no metric bookkeeping: reported next to the metric when the two differ
    dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_host = min(args.steps, 8)                    # host cost per step from the first few calls: with a long queue of
    t_enqueued = 0.0                               # graph launches in flight the runtime blocks the caller (back-pressure),
    for i, b in enumerate(batches(args.steps, False)):   # which is GPU time, not host work
        trainer.train_step(b, train=True)
        if i + 1 == n_host:
            t_enqueued = time.perf_counter() - t0  # no sync inside a step
    torch.cuda.synchronize()
    dp.barrier()
    torch.cuda.synchronize()
    dt_steps = dp.max_over_ranks(time.perf_counter() - t0)
    # (b) the timed region of `value`: EXACTLY args.steps iterations of epoch(train=True)
    clock = ClockSampler() if dp.rank == 0 else None
    dp.barrier()
    torch.cuda.synchronize()
    if clock is not None:
        clock.start()
    t0 = time.perf_counter()
    means = run_epoch(args.steps, not use_graph)
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0                 # this rank's own time (before it waits for the others)
    dp.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    clock_info = clock.finish() if clock is not None else None
    timer.enabled = False
    dt = dp.max_over_ranks(dt)
    per_rank_ms = dp.gather_floats(1e3 * dt_local / args.steps) if dp.distributed else None
    timed_steps = max(1, len(range(0, args.steps, 4)))
    # c1_long (round 6): the metric of record over 300 more steps of the SAME trainer / graph, right after the timed region and
    # outside `value`: the driver's --steps 20 times 0.44 s, where box-to-box and run-to-run spread (+- 2 %) is the size of the
    # gains a round claims; 300 steps (SURVEY.md section 8(d) asks >= 50) put the figure's own noise below that
    c1_long = None
    if (args.config == 'C1' and dp.world_size == 1 and not args.no_extras and not args.host_inputs and args.batch is None
            and not args.no_long):
        n_long = 300
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m_long = run_epoch(n_long, False)
        torch.cuda.synchronize()
        dt_long = time.perf_counter() - t0
        c1_long = {'value': round(B * n_long / dt_long, 2), 'unit': 'windows/s', 'ms_per_step': round(1e3 * dt_long / n_long, 3),
                   'steps': n_long, 'final_loss': round(float(m_long['loss']), 5),
                   'f16x3_scale_saturations': m_long.get('f16x3_scale_saturations'),
                   'note': 'same command, same trainer, 300 further iterations of epoch(train=True) after the timed region; NOT `value`'}
    graph_replays, graphs_per_step = None, None
    if use_graph:
        g = getattr(trainer, '_graph', None)
        graph_replays = g.replays if g is not None else 0
        graphs_per_step = len(g.stages) if g is not None else 1
        trainer.enable_step_graph(False)
        timed_steps = max(1, sampled_eager_steps)
    # the collective of the data-parallel step alone: the flat gradient bucket (what every step all-reduces), back to back
    allreduce = None
    if dp.distributed:
        import torch.distributed as dist
        bucket = torch.zeros_like(trainer.flat.flat_grad)
        for _ in range(3):
            dp.all_reduce_sum_(bucket)
        torch.cuda.synchronize()
        dp.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_ar = 20
        for _ in range(n_ar):
            dp.all_reduce_sum_(bucket)
        torch.cuda.synchronize()
        ar_ms = dp.max_over_ranks((time.perf_counter() - t0) * 1e3 / n_ar)
        nbytes = bucket.numel() * 4
        allreduce = {'ms_per_step': round(ar_ms, 4), 'bucket_bytes': nbytes, 'calls_per_step': 1,
                     'bus_gb_s': round(2.0 * (dp.world_size - 1) / dp.world_size * nbytes / (ar_ms * 1e-3) / 1e9, 2),
                     'share_of_step': round(ar_ms / (1e3 * dt / args.steps), 4),
                     'backend': dp.backend, 'rccl_ranks': dist.get_world_size(),
                     'note': f'{n_ar} back-to-back all-reduces of the flat fp32 gradient bucket after the timed region (max over '
                             'ranks); inside the step the same call sits between the two graph replays'}
        del bucket
    n1_leg = None
    if dp.distributed and not args.no_n1_leg:
        torch.cuda.synchronize()
        if dp.rank == 0:
            n1_leg = n1_same_node_leg(args, B, dev.index if dev.index is not None else 0)
        dp.barrier()                                  # the other ranks idle here meanwhile
    # Extra, NOT the headline: the same steps with the forward's six-product split in the backward pass as well (what every
    # round before the fifth measured), after the timed region so that it cannot touch `value`: the price of carrying 24-bit
    # operand mantissas through six MFMAs where 22 bits through three give the same fp32-class gradients.
    extra_six, extra_fwd_six = None, None

    def extra_leg(grad_arith, fwd_arith, note):
        try:
            ops.set_gradient_arithmetic(grad_arith)
            ops.set_forward_arithmetic(fwd_arith)
            if use_graph:
                trainer.enable_step_graph(True)
                trainer._graph_eager_steps = 0
            n3 = min(args.steps, 20)
            for b in batches(6, False):
                trainer.train_step(b, train=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m3 = run_epoch(n3, False)
            torch.cuda.synchronize()
            dt3 = time.perf_counter() - t0
            return {'value': round(B * n3 / dt3, 2), 'unit': 'windows/s', 'ms_per_step': round(1e3 * dt3 / n3, 3), 'steps': n3,
                    'final_loss': round(float(m3['loss']), 5), 'note': note}
        except Exception as e:                        # never lose the headline line to the extra
            return {'error': f'{type(e).__name__}: {str(e)[:200]}'}
        finally:
            ops.set_gradient_arithmetic(args.grad_arith)
            ops.set_forward_arithmetic(args.fwd_arith)
            if use_graph:
                trainer.enable_step_graph(False)

    if (gemm_mode == 1 and args.grad_arith == 'f16x3' and not args.no_extras and args.config == 'C1' and dp.world_size == 1
            and not args.host_inputs):
        if args.fwd_arith == 'f16x3':
            extra_fwd_six = extra_leg('f16x3', 'six',
                                      'NOT the headline: the same steps with the six-product (bf16x6) split in every FORWARD product '
                                      '(ops.set_forward_arithmetic(\'six\') / bench.py --fwd-arith six), f16x3 gradient GEMMs')
        extra_six = extra_leg('six', 'six',
                              'NOT the headline: the same steps with six-product (bf16x6) GEMMs everywhere, forward and backward '
                              '(bench.py --grad-arith six --fwd-arith six), i.e. the configuration rounds 1-4 reported')
    pmc = None
    want_pmc = args.live_pmc if args.live_pmc is not None else (dp.world_size == 1 and args.config == 'C1' and not args.host_inputs)
    if want_pmc and dp.rank == 0 and dp.world_size == 1:
        pmc = live_pmc(args, B)
    flush_c_stdio()
    dp.barrier()                                    # every rank has emitted whatever its libraries had buffered
    student = config['training_method'].lower() == 'student'
    last_loss = float(means['loss_encdec'] if student else means['loss'])

    seq_len = 384 if (student or decoder_step) else 16 * (dlg.num_blocks_left + dlg.num_blocks_right)
    if dp.rank == 0 and args.gemm_breakdown:
        print(timer.breakdown('gemm_nt', timed_steps), file=sys.stderr)
        print(timer.breakdown('gemm_tn', timed_steps), file=sys.stderr)
        if timer.records['cast_bf16']:
            print(timer.breakdown('cast_bf16', timed_steps), file=sys.stderr)
    if dp.rank == 0:
        value = B * dp.world_size * args.steps / dt
        nt, tn = timer.summary('gemm_nt'), timer.summary('gemm_tn')
        roofline = None
        if nt:
            if gemm_mode == 1:
                # 6 bf16 MFMAs per fp32 product: the MFMA ceiling for ALGORITHMIC fp32 FLOPs is bf16 dense peak / 6
                peak, kname = PEAK_BF16_MFMA_TFLOPS / 6.0, ('gemm_nt = every NT GEMM launch (gemm_nt_x6_pp_kernel / gemm_nt_x6_256_kernel 256-tile, '
                               'gemm_nt_kernel<MODE=1> 128-tile, gemm_nt_skinny_kernel); bf16x6: exact 3-way bf16 split, '
                               '6x v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate')
            elif gemm_mode == 2:
                # reduced precision (BASELINE configs[4] names bf16): NOT valid for the fp32 headline configuration
                peak, kname = PEAK_BF16_MFMA_TFLOPS, ('gemm_nt = every NT GEMM launch (gemm_nt_bf16_kernel 256-tile on bf16 operands in HBM for the '
                               'transformer layers; gemm_nt_kernel<MODE=2> 128-tile rounding fp32 operands for the rest; '
                               'gemm_nt_skinny_kernel in fp32); one v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate')
            else:
                peak, kname = PEAK_F32_MFMA_TFLOPS, 'gemm_nt = every NT GEMM launch (gemm_nt_kernel<MODE=0>, gemm_nt_skinny_kernel; fp32 v_mfma_f32_32x32x2_f32)'
            by_arith = None
            if gemm_mode == 1 and 'f16x3' in nt:
                # mixed arithmetic: forward launches on six MFMAs per product (ceiling 2500 / 6), the backward's 256-tile input
                # gradients on three (2500 / 3).  The fraction is time-weighted: (sum of flops_i / peak_i) / measured time, i.e.
                # `peak` below is the ceiling of THIS mix of launches
                p6, p3 = PEAK_BF16_MFMA_TFLOPS / 6.0, PEAK_BF16_MFMA_TFLOPS / 3.0
                g3_, r6_ = nt['f16x3'], nt['rest']
                ideal_ms = (g3_['flops'] / p3 + r6_['flops'] / p6) / 1e9
                peak = (g3_['flops'] + r6_['flops']) / ideal_ms / 1e9
                by_arith = {'six_products': {'achieved': round(r6_['tflops'], 2), 'peak': round(p6, 1), 'frac': round(r6_['tflops'] / p6, 4),
                                             'launches_per_step': r6_['launches'] // timed_steps,
                                             'ms_per_step': round(r6_['total_ms'] / timed_steps, 3)},
                            'f16x3': {'achieved': round(g3_['tflops'], 2), 'peak': round(p3, 1), 'frac': round(g3_['tflops'] / p3, 4),
                                      'launches_per_step': g3_['launches'] // timed_steps,
                                      'ms_per_step': round(g3_['total_ms'] / timed_steps, 3),
                                      'kernel': 'gemm_nt_g3_kernel (csrc/gemm_grad.hip): two fp16 planes per operand under a '
                                                'per-tensor power-of-two scale, 3x v_mfma_f32_32x32x16_f16 per product'}}
                kname += ('; the 256-tile products of the training step (forward launches under --fwd-arith f16x3 and the input-gradient GEMMs of backward) run on gemm_nt_g3_kernel '
                          '(f16x3: three fp16 MFMAs per product), ragged launches as whole rounds + their tail rows on gemm_nt_g3_tail_kernel (64 x 128 tiles)')
            calls_per_step = max(1, nt['launches'] // timed_steps)
            traffic = traffic_src = eff_mhz = pipe_busy = None
            if pmc and 'gemm_nt' in pmc:
                g_ = pmc['gemm_nt']
                if 'hbm_bytes_per_step' in g_:
                    traffic = round(g_['hbm_bytes_per_step'] / calls_per_step)
                    traffic_src = ('live: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE passes over '
                                   f'`bench.py --steps 2 --warmup 2 --no-graph` ({pmc["steps"]} steps, '
                                   f'{g_["kernel_launches_per_step"]:.0f} gemm_nt kernel launches per step) run by this process '
                                   'after the timed region; FETCH_SIZE x2 (gfx950) + WRITE_SIZE')
                eff_mhz, pipe_busy = g_.get('effective_clock_mhz'), g_.get('mfma_pipe_busy')
            if traffic is None and args.config == 'C1' and gemm_mode == 1:
                traffic, traffic_src = hbm_traffic_from_file('gemm_nt', calls_per_step)
            # two ceilings per launch (round 6): with three MFMAs per product and fp32 tensors in HBM every f16x3 NT shape of the step
            # has an HBM floor at or above its MFMA floor, so `bound` is decided per launch family and `frac` is quoted against the
            # BINDING ceiling of the family sum; the one-ceiling MFMA figures of rounds 1-5 stay under `mfma`
            tc_nt = two_ceiling(timer.records['gemm_nt'], gemm_mode, timed_steps)
            tc_tn = two_ceiling(timer.records['gemm_tn'], gemm_mode, timed_steps)
            step_ms = 1e3 * dt / args.steps
            step_floor = None
            if tc_nt:
                nongemm = (sum(C1_NONGEMM_BYTES_B256.values()) * B / 256.0) if (args.config == 'C1' and gemm_mode == 1) else None
                g_bind = tc_nt['binding_floor_ms'] + (tc_tn['binding_floor_ms'] if tc_tn else 0.0)
                g_hbm = tc_nt['hbm_floor_ms'] + (tc_tn['hbm_floor_ms'] if tc_tn else 0.0)
                g_mfma = tc_nt['mfma_floor_ms'] + (tc_tn['mfma_floor_ms'] if tc_tn else 0.0)
                ng_ms = None if nongemm is None else nongemm / PEAK_HBM_BYTES_S * 1e3
                step_floor = {'ms_per_step': round(step_ms, 3),
                              'hbm_floor_ms': None if ng_ms is None else round(g_hbm + ng_ms, 3),
                              'mfma_floor_ms': round(g_mfma, 3),
                              'binding_floor_ms': None if ng_ms is None else round(g_bind + ng_ms, 3),
                              'frac_of_binding': None if ng_ms is None else round((g_bind + ng_ms) / step_ms, 4),
                              'gemm_algorithmic_gb': round((tc_nt['bytes_per_step'] + (tc_tn['bytes_per_step'] if tc_tn else 0.0)) / 1e9, 2),
                              'non_gemm_algorithmic_gb': None if nongemm is None else round(nongemm / 1e9, 2),
                              'note': 'whole step: sum over GEMM launches of max(flops / MFMA peak of the launch\'s arithmetic, algorithmic '
                                      'bytes / 8 TB/s) + algorithmic bytes of every other kernel / 8 TB/s (bench.py C1_NONGEMM_BYTES_B256; '
                                      'each stream once), against the driver-timed ms_per_step'}
            hbm_bound = bool(tc_nt) and tc_nt['bound'] == 'hbm'
            ach_tbs = (tc_nt['bytes_per_step'] / (tc_nt['ms_per_step'] * 1e-3) / 1e9) if tc_nt else None       # GB/s
            roofline = dict(bound='hbm' if hbm_bound else 'mfma', kernel=kname,
                            achieved=round(ach_tbs, 1) if hbm_bound else round(nt['tflops'], 2),
                            peak=PEAK_HBM_BYTES_S / 1e9 if hbm_bound else round(peak, 1), unit='GB/s' if hbm_bound else 'TFLOP/s',
                            frac=round(ach_tbs / (PEAK_HBM_BYTES_S / 1e9), 4) if hbm_bound else round(nt['tflops'] / peak, 4),
                            frac_of_binding=tc_nt['frac_of_binding'] if tc_nt else None,
                            bound_note=('per launch t_mfma = 2MNK / peak(arithmetic: 2500 / 3 f16x3, 2500 / 6 six products), t_hbm = '
                                        'algorithmic bytes (A, B, C and every epilogue operand once) / 8 TB/s; `bound` = the larger of the two '
                                        'sums over the gemm_nt launches, `frac` = achieved / peak in that unit, `frac_of_binding` = sum of '
                                        'per-launch max(t_mfma, t_hbm) / sum of measured durations; `families` lists the same per shape'),
                            mfma=dict(achieved=round(nt['tflops'], 2), peak=round(peak, 1), unit='TFLOP/s', frac=round(nt['tflops'] / peak, 4),
                                      floor_ms_per_step=tc_nt['mfma_floor_ms'] if tc_nt else None),
                            hbm=dict(achieved=round(ach_tbs, 1) if tc_nt else None, peak=PEAK_HBM_BYTES_S / 1e9, unit='GB/s',
                                     frac=round(ach_tbs / (PEAK_HBM_BYTES_S / 1e9), 4) if tc_nt else None,
                                     floor_ms_per_step=tc_nt['hbm_floor_ms'] if tc_nt else None),
                            measured_ms_per_step=tc_nt['ms_per_step'] if tc_nt else None,
                            families=tc_nt['families'][:16] if tc_nt else None,
                            gemm_tn_two_ceiling=({k: v for k, v in tc_tn.items() if k != 'families'} | {'families': tc_tn['families'][:8]}) if tc_tn else None,
                            step=step_floor, by_arithmetic=by_arith,
                            traffic=traffic, traffic_unit='HBM bytes per ops.gemm_nt call', traffic_source=traffic_src,
                            algorithmic_bytes_per_launch=round(nt['bytes_per_launch']),
                            launches_per_step=nt['launches'] // timed_steps, avg_launch_us=round(nt['avg_us'], 1),
                            flops_per_launch=nt['flops_per_launch'],
                            share_of_step=round(nt['total_ms'] / (dt * 1e3 * timed_steps / args.steps), 3),
                            vs_fp32_mfma_peak=round(nt['tflops'] / PEAK_F32_MFMA_TFLOPS, 3),
                            # the clock the kernels actually run at: GRBM_GUI_ACTIVE / kernel wall time (live PMC pass); the
                            # sysfs value in `clock` is NOT it (profiles/r03_gemm_clock.txt: 1.4-1.5 GHz vs 1.9 in sysfs)
                            effective_clock_mhz=eff_mhz, mfma_pipe_busy_at_effective_clock=pipe_busy,
                            effective_clock_note=('GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time over the gemm_nt launches >= 0.2 ms of '
                                                  'the live PMC pass; counter collection serialises the kernels (idle gaps, lower '
                                                  'average power), so this is an UPPER bound of the clock inside the graph-replayed '
                                                  'step: the same kernels launched back to back sustain 1.39-1.45 GHz '
                                                  '(profiles/r03_gemm_clock.txt), and 1.44 TFLOP/s per CU on 32-64 CUs against 0.845 on '
                                                  '256: the chip is power-limited (profiles/r03_gemm_power_limit.txt)') if eff_mhz else None,
                            frac_at_measured_sclk=(round(nt['tflops'] / (peak * eff_mhz / 2400.0), 4) if eff_mhz else None),
                            frac_at_sysfs_sclk=(round(nt['tflops'] / (peak * clock_info['sclk_mhz_median'] / 2400.0), 4)
                                                if clock_info else None),
                            pmc_error=(pmc or {}).get('error'),
                            sampled=(f'HIP events around every GEMM launch of {timed_steps} eager steps run right before the graph '
                                     'capture (the timed steps are graph replays)' if use_graph else
                                     'HIP events around every GEMM launch of every 4th step of the timed region'))
        line = {
            'metric': f'encoder-train windows/sec (Bach 4-voice, seq={seq_len})', 'value': round(value, 2), 'unit': 'windows/s',
            'n_gpus': dp.world_size, 'rccl_ranks': (allreduce['rccl_ranks'] if allreduce else 1),
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * dt / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16' if gemm_mode == 2 else 'f32',
            'data': 'synthetic' + (' (host-resident inputs: PCIe-inclusive)' if args.host_inputs else ''),
            'config': {'workload': ('' if decoder_step else
                                    f'encoder_cpc {args.config}: seq_len={seq_len}, '
                                    f'batch={B}/GPU, {getattr(dlg, "num_negative_samples", 0)} negatives, product-VQ '
                                    f'{config["quantizer_kwargs"]["num_codebooks"]}x{config["quantizer_kwargs"]["codebook_size"]}, '
                                    f'd_model={config["downscaler_kwargs"]["d_model"]}, dropout={args.dropout}'),
                       'global_batch': B * dp.world_size, 'seq_len': seq_len,
                       'parallelism': f'dp{dp.world_size}', 'params': n_params,
                       'path': ('what train_model() selects by default: bf16x6 GEMM mode with the f16x3 kernels for the whole-round '
                                '256-tile products of the training step (forward and backward), step-graph replay'
                                if (gemm_mode == 1 and use_graph and args.grad_arith == ops.TRAINING_GRAD_ARITH
                                    and args.fwd_arith == ops.TRAINING_FWD_ARITH) else
                                'non-default switches (see gemm / gradient_arithmetic / forward_arithmetic / step_graph)'),
                       'gradient_arithmetic': args.grad_arith if gemm_mode == 1 else 'as the forward',
                       'forward_arithmetic': args.fwd_arith if gemm_mode == 1 else 'the GEMM mode',
                       'gemm': (('bf16x6 split-MFMA (fp32 in/out, fp32-class accuracy: exact 3-way bf16 split, 6 MFMAs per product)'
                                 + ('; the 256-tile products of the training step (ragged launches: whole rounds + tail rows on 64 x 128 tiles) -- dgrad / wgrad'
                                    + (' AND the forward launches' if args.fwd_arith == 'f16x3' else '') +
                                    ' -- on f16x3 (NORMWISE fp32-class, componentwise 22-bit operands: two fp16 planes per operand, 11 + 11 bits '
                                    'under a per-tensor power-of-two scale of the previous step\'s amax, 3 MFMAs per product; rms error vs fp64 2.7e-7 / 4.1e-7 at K = 256 / 1024 against '
                                    '2.4e-7 / 4.9e-7 for the six-product split and 2.9e-7 / 5.7e-7 for the exact fp32-MFMA kernel; '
                                    'oracle parity suites green at unchanged tolerances, indices bit-exact; '
                                    + ('full-size C1 code assignments: 0-1 of 69 632 differ from the six-product forward, 0 from the '
                                       'exact fp32-MFMA forward; evaluation / inference stay on six products)'
                                       if args.fwd_arith == 'f16x3' else 'forward, losses and code assignment untouched)')
                                    if args.grad_arith == 'f16x3' else
                                    '; backward: two bf16 planes, 3 MFMAs per product (18-bit operands)' if args.grad_arith == 'bf16x3'
                                    else '')) if gemm_mode == 1 else
                                'bf16 operands, fp32 accumulate (reduced precision)' if gemm_mode == 2 else 'fp32 MFMA')},
            'roofline': roofline,
            'allreduce': allreduce,
            'per_rank': ({'ms_per_step_min': round(min(per_rank_ms), 3), 'ms_per_step_max': round(max(per_rank_ms), 3),
                          'ms_per_step': [round(x, 3) for x in per_rank_ms],
                          'note': 'each rank\'s own time over the timed region, taken before it waits for the others; '
                                  '`ms_per_step` above is the barrier-to-barrier maximum'} if per_rank_ms else None),
            'n1_same_node': n1_leg,
            'gemm_tn': ({'achieved': round(tn['tflops'], 2), 'unit': 'TFLOP/s', 'avg_launch_us': round(tn['avg_us'], 1),
                         'by_arithmetic': ({'f16x3': {'achieved': round(tn['f16x3']['tflops'], 2), 'peak': round(PEAK_BF16_MFMA_TFLOPS / 3.0, 1),
                                                      'frac': round(tn['f16x3']['tflops'] / (PEAK_BF16_MFMA_TFLOPS / 3.0), 4),
                                                      'launches_per_step': tn['f16x3']['launches'] // timed_steps,
                                                      'ms_per_step': round(tn['f16x3']['total_ms'] / timed_steps, 3)},
                                            'six_products': {'achieved': round(tn['rest']['tflops'], 2),
                                                             'launches_per_step': tn['rest']['launches'] // timed_steps,
                                                             'ms_per_step': round(tn['rest']['total_ms'] / timed_steps, 3)}}
                                           if 'f16x3' in tn else None),
                         'share_of_step': round(tn['total_ms'] / (dt * 1e3 * timed_steps / args.steps), 3),
                         'effective_clock_mhz': (pmc or {}).get('gemm_tn', {}).get('effective_clock_mhz'),
                         'mfma_pipe_busy_at_effective_clock': (pmc or {}).get('gemm_tn', {}).get('mfma_pipe_busy')} if tn else None),
            'cast_bf16': ({'ms_per_step': round(sum(r[0].elapsed_time(r[1]) for r in timer.records['cast_bf16']) / timed_steps, 3),
                           'note': 'fp32 -> bf16 operand casts of the bf16 path (outside the gemm_nt bracket)'}
                          if timer.records['cast_bf16'] else None),
            'clock': clock_info,
            'final_loss': round(last_loss, 5),
            'timed': 'trainer.epoch(train=True, num_batches=steps): steps + per-step metric bookkeeping + end-of-epoch host read',
            'step_graph': ({'replays_in_run': graph_replays, 'graphs_per_step': graphs_per_step,
                            'note': ('each training step is one HIP-graph replay' if graphs_per_step == 1 else
                                     f'each training step is {graphs_per_step} HIP-graph replays around the eagerly issued '
                                     'all-reduce(s) of the gradient bucket') + ' (vqcpc_bach_amd/graphs.py); --no-graph runs the same launches eagerly'}
                           if use_graph else None),
            'train_step_only': {'value': round(B * dp.world_size * args.steps / dt_steps, 2),
                                'ms_per_step': round(1e3 * dt_steps / args.steps, 3),
                                'host_enqueue_ms_per_step': round(1e3 * t_enqueued / n_host, 3),
                                'note': 'same steps without epoch()\'s metric bookkeeping; not the metric'},
        }
        if gemm_mode == 1 and 'f16x3' in (args.grad_arith, args.fwd_arith) and hasattr(trainer, 'flat'):
            # operands that outgrew the head-room of their previous-step scale (clamped for one step), over the whole run
            line['f16x3_scale_saturations'] = ops.scale_saturations(trainer.flat)
        if c1_long is not None:
            line['c1_long'] = c1_long
        if extra_fwd_six is not None:
            line['extra_six_product_forward'] = extra_fwd_six
        if extra_six is not None:
            line['extra_six_product_gradients'] = extra_six
        if n1_leg and n1_leg.get('value'):
            # what the multi-rank step costs beyond the single-rank step of the same node: the all-reduce (and its launch
            # gaps) that is NOT hidden.  The driver computes scaling efficiency itself from its own per-N runs; this is the
            # same quantity on one lease, for reading the first multi-GPU run.
            step_n, step_1 = line['ms_per_step'], n1_leg['ms_per_step']
            line['scaling_same_node'] = {'efficiency': round(value / (dp.world_size * n1_leg['value']), 4),
                                         'exposed_ms_per_step': round(step_n - step_1, 3),
                                         'exposed_share_of_step': round((step_n - step_1) / step_n, 4),
                                         'allreduce_alone_ms': (allreduce or {}).get('ms_per_step'),
                                         'note': 'ms_per_step(N ranks) - ms_per_step(1 rank, same node, same command): '
                                                 'the part of the gradient all-reduce (+ the two-graph step\'s launch gap) '
                                                 'that the step does not hide; compare with allreduce_alone_ms'}
        if student:     # BASELINE configs[3]: an extra measurement, not the headline metric
            line['metric'], line['unit'] = 'student-train sequences/sec (Bach 4-voice, 24 beats = 384 tokens)', 'sequences/s'
            tk, dk = config['auxiliary_networks_kwargs']['teacher_kwargs'], config['downscaler_kwargs']
            line['config'] = {'workload': f'encoder_student C3: x (B, 96, 4), batch={B}/GPU, teacher {tk["num_layers"]} layers '
                                          f'L=384, encoder {dk["list_of_num_layers"]} layers (linear aggregation), decoder '
                                          f'L=24/96, d_model={dk["d_model"]}, VQ 1x32 dim 3, dropout={args.dropout}',
                              'global_batch': B * dp.world_size, 'seq_len': 384, 'parallelism': f'dp{dp.world_size}',
                              'params': n_params, 'gemm': line['config']['gemm']}
        if decoder_step:   # an extra measurement, not the headline metric
            line['metric'], line['unit'] = 'decoder-train sequences/sec (Bach 4-voice, 24 beats = 384 tokens)', 'sequences/s'
            dk, ek = config['decoder_kwargs'], config['config_encoder']['downscaler_kwargs']
            line['config'] = {'workload': f'decoder DEC: x (B, 96, 4) -> 24 codes (frozen transformer encoder d_model '
                                          f'{ek["d_model"]}, {ek["list_of_num_layers"]} layers, 1x32 codes) -> relative seq2seq '
                                          f'transformer {dk["num_encoder_layers"]}+{dk["num_decoder_layers"]} layers, d_model '
                                          f'{dk["d_model"]}, {dk["n_head"]} heads, ff {dk["dim_feedforward"]}, anticausal source / '
                                          f'anticausal cross / causal target, batch={B}/GPU, dropout={args.dropout}',
                              'global_batch': B * dp.world_size, 'seq_len': 384, 'parallelism': f'dp{dp.world_size}',
                              'params': n_params, 'gemm': line['config']['gemm']}
        if dp.world_size == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args.config, args.dropout, args.cpu_batch, args.cpu_steps)
            if line['cpu_baseline'].get('value'):
                line['speedup_vs_cpu'] = round(value / line['cpu_baseline']['value'], 1)
        else:
            line['cpu_baseline'] = None
        if (dp.world_size == 1 and args.config == 'C1' and not args.no_secondary and not args.host_inputs
                and args.batch is None and gemm_mode == 1):
            line['secondary'] = secondary_runs()
        print(json.dumps(line), flush=True)
    dp.shutdown()


if __name__ == '__main__':
    main()
