"""Copy the output of scripts/measure_round.sh (gpurun_out/<tag>_*) into profiles/ with generated '#' headers (what was run, the un-profiled
bench line of the same box, how to read the counters) and refresh profiles/pmc_summary.json (HBM-side bytes per launch of the option-LSTM
kernels, fp32-MFMA and split9, tied to the kernel sources by their digest).      python scripts/install_profiles.py r05"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TAG = sys.argv[1] if len(sys.argv) > 1 else 'r05'
G, P = os.path.join(ROOT, 'gpurun_out') + '/', os.path.join(ROOT, 'profiles') + '/'


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def body(name):
    return [l for l in open(G + name).read().split('\n') if not l.startswith('# ') and 'amdgpu.ids' not in l]


for f in ('bench_default.json', 'bench_50steps.json', 'bench_config4.json'):
    shutil.copy(G + '%s_%s' % (TAG, f), P + '%s_%s' % (TAG, f))
d, d50 = last_json(G + TAG + '_bench_default.json'), last_json(G + TAG + '_bench_50steps.json')
r, alt = d['roofline'], d.get('alt') or {}
fam = r['families'][r['kernel']]
hdr = [
    "# rocprofv3 --kernel-trace --marker-trace -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-alt   (MI355X, round %s state,"
    % TAG[1:].lstrip('0'),
    "# scripts/measure_round.sh: native host = model-level C ABI, GPU_MAX_HW_QUEUES=1, 7 training steps traced; summarised from the rocpd database by",
    "# `scripts/rocpd_stats.py <db> --steps-only`, i.e. up to the last optimiser launch -- bench.py's `roofline.alone` leg is excluded).  The option recurrence runs",
    "# the exact-split kernels (csrc/split_core.h: gemm_split_kernel<9, Epi> = the step kernels, gemm_split_tn_kernel<9> = dWh), everything else v_mfma_f32_32x32x2_f32",
    "# (gemm_f32<WMxWKxNTxKWxDBxMINWxLDSMINxBF16|A, B, Epi>, gemm_f32_glds = LDS-DMA pipeline kernels, gemm_f32_grouped = encoder ticks).",
    "# Same box, un-profiled: `python bench.py` = %.3f ms/step (%.0f QA-rounds/s), 50 steps: %.3f ms/step; alt (fp32-MFMA recurrence, same run): %s ms/step."
    % (d['ms_per_step'], d['value'], d50['ms_per_step'], alt.get('ms_per_step')),
    "# Dominant kernel = %s timestep (split9): HIP events %.1f us per launch in the step -> %.0f TFLOP/s of executed bf16 MFMA work = %.3f of the 2.5 PFLOP/s"
    % (r['kernel'], fam['avg_launch_ms'] * 1e3, r['achieved'], r['frac']),
    "# dense peak (%.3f of the 1.86 PFLOP/s the pipe sustains on non-zero operands); alone %.1f us = %.3f." % (
        r.get('frac_of_sustained_peak', 0), r['alone']['avg_launch_ms'] * 1e3, r['alone']['frac']),
]
open(P + TAG + '_kernel_stats_bench.txt', 'w').write('\n'.join(hdr + body(TAG + '_kernel_stats_bench.txt')))
for f in ('_stream_timeline_bench.txt', '_hbm_kernels.txt', '_roctx_ranges.txt'):
    if os.path.exists(G + TAG + f):
        open(P + TAG + f, 'w').write('\n'.join(l for l in open(G + TAG + f).read().split('\n') if 'amdgpu.ids' not in l))

txt = open(G + TAG + '_pmc_option_lstm_kernels.txt').read()
S3 = 'SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES'
KERN = {('fp32', 'fwd'): r'gemm_f32_glds_kernel<[^\n]*false, EpiLstmFwdT<0', ('fp32', 'bwd'): r'gemm_f32_glds_kernel<[^\n]*false, EpiLstmBwd<2,',
        ('fp32', 'dWh'): r'gemm_f32_glds_kernel<[^\n]*true, EpiAtomic<4>', ('split9', 'fwd'): r'gemm_split_kernel<9, EpiLstmFwdT<0',
        ('split9', 'bwd'): r'gemm_split_kernel<9, EpiLstmBwd<4,', ('split9', 'dWh'): r'gemm_split_tn_kernel<9>'}
ALG = {'fwd': 496, 'bwd': 660, 'dWh': 3900}     # algorithmic MB per launch


def grab(mode, section, kern, ctr):
    sec = txt[txt.index('## %s %s' % (mode, section)):]
    nxt = sec.find('\n## ', 3)
    sec = sec[:nxt] if nxt > 0 else sec
    m = re.search(kern + r'[^\n]*\n(?:    .*\n)*?    ' + ctr + r'\s+avg ([0-9.e+]+)', sec)
    return float(m.group(1))


summ = {"_comment": "HBM-side bytes per launch from rocprofv3 PMC passes (profiles/%s_pmc_option_lstm_kernels.txt): (2*FETCH_SIZE + WRITE_SIZE) KB, FETCH doubled "
                    "per MI355X_MICROARCH.md (gfx950 counts wide coalesced reads at half).  opt_lstm_fwd / _bwd = ONE timestep launch of the fp32-MFMA kernels "
                    "(19 per direction per step), opt_lstm_dWh = the single weight-gradient launch; 'split9:opt_lstm_fwd' / '_bwd' / '_dWh' = the exact-split kernels.  "
                    "bench.py copies hbm_bytes_per_launch of the dominant kernel into roofline.traffic while csrc_sha256 matches the build (a committed "
                    "measurement, not a live counter)." % TAG}
lines = []
for (mode, k), kern in KERN.items():
    try:
        fe, wr = grab(mode, 'FETCH_SIZE', kern, 'FETCH_SIZE') / 1e3, grab(mode, 'WRITE_SIZE', kern, 'WRITE_SIZE') / 1e3
        g, mb = grab(mode, S3, kern, 'GRBM_GUI_ACTIVE'), grab(mode, S3, kern, 'SQ_VALU_MFMA_BUSY_CYCLES')
    except Exception as exc:
        lines.append("# %-6s %-4s not found in the PMC output (%s)" % (mode, k, type(exc).__name__))
        continue
    tot = 2 * fe + wr
    lines.append("# %-6s %-4s FETCH %.1f MB raw (x2 = %.0f MB) + WRITE %.0f MB -> %.0f MB HBM-side per launch vs %d MB algorithmic (%.2fx); GRBM %.4g "
                 "cycles; MFMA busy %.1f %%" % (mode, k, fe, 2 * fe, wr, tot, ALG[k], tot / ALG[k], g, 100 * mb / (g * 32)))
    summ[('' if mode == 'fp32' else mode + ':') + 'opt_lstm_' + k] = dict(
        hbm_bytes_per_launch=int(round(tot * 1e6)), fetch_raw_bytes=int(round(fe * 1e6)), write_bytes=int(round(wr * 1e6)),
        algorithmic_bytes=ALG[k] * 1e6, mfma_busy=round(mb / (g * 32), 3), grbm_cycles=g)
from bench import csrc_digest  # noqa: E402
summ['csrc_sha256'] = csrc_digest()     # the kernel sources the PMC passes were taken on: bench.py reports roofline.traffic only while they match
json.dump(summ, open(P + 'pmc_summary.json', 'w'), indent=1)
head = ["# separate passes per counter group and per arithmetic: rocprofv3 --pmc <counters> -- python scripts/pmc_target.py <fp32|split9> (the option-LSTM kernel",
        "# families alone at the headline shape; MI355X, round %s build, scripts/measure_round.sh).  Only this library's step / contraction kernels are listed." % TAG[1:].lstrip('0')]
tail = ["## how to read (per launch; SQ_* values are per shader-engine averages: x32 for the chip; 1024 SIMDs; FETCH_SIZE / WRITE_SIZE are KB, FETCH x2 on gfx950 per "
        "MI355X_MICROARCH.md; GRBM cycles / launch time = the clock the kernel ran at)"] + lines
open(P + TAG + '_pmc_option_lstm_kernels.txt', 'w').write('\n'.join(head + [l for l in txt.split('\n') if l and 'amdgpu.ids' not in l] + tail) + '\n')
print('\n'.join(lines))
print('bench: %.3f ms/step, %.0f QA-rounds/s, frac %.3f (alone %.3f); alt %s ms/step' % (d['ms_per_step'], d['value'], r['frac'], r['alone']['frac'], alt.get('ms_per_step')))
