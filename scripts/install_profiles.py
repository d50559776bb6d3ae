"""Copy the output of scripts/measure_round.sh (gpurun_out/<tag>_*) into profiles/, keeping the hand-written '#' headers of the text
files and refreshing the numbers they quote (bench lines, dominant-kernel line, the PMC 'how to read' block, profiles/pmc_summary.json)."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else 'r03'
G, P = os.path.join(ROOT, 'gpurun_out') + '/', os.path.join(ROOT, 'profiles') + '/'


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def hdr(name):
    return [l for l in open(P + name).read().split('\n') if l.startswith('#')]


def body(name):
    return [l for l in open(G + name).read().split('\n') if not l.startswith('#')]


for f in ('bench_default.json', 'bench_50steps.json', 'bench_config4.json'):
    shutil.copy(G + '%s_%s' % (TAG, f), P + '%s_%s' % (TAG, f))
d, d50 = last_json(G + TAG + '_bench_default.json'), last_json(G + TAG + '_bench_50steps.json')
r = d['roofline']
bw = r['families']['opt_lstm_bwd']
h = hdr(TAG + '_kernel_stats_bench.txt')
for i, l in enumerate(h):
    if l.startswith('# Same box, un-profiled'):
        h[i] = ("# Same box, un-profiled: `python bench.py` = %.3f ms/step (%.0f QA-rounds/s), 50 steps: %.3f ms/step; dominant kernel = "
                "option-LSTM backward timestep:" % (d['ms_per_step'], d['value'], d50['ms_per_step']))
        h[i + 1] = ("# HIP events %.1f us per launch in the step -> %.1f TFLOP/s executed = %.3f of the 157.3 TFLOP/s fp32 MFMA peak; alone "
                    "%.1f us = %.1f TFLOP/s = %.3f." % (bw['avg_launch_ms'] * 1e3, r['achieved'], r['frac'], r['alone']['avg_launch_ms'] * 1e3,
                                                        r['alone']['achieved'], r['alone']['frac']))
open(P + TAG + '_kernel_stats_bench.txt', 'w').write('\n'.join(h + body(TAG + '_kernel_stats_bench.txt')))
for f in (TAG + '_stream_timeline_bench.txt', TAG + '_hbm_kernels.txt'):
    open(P + f, 'w').write('\n'.join(hdr(f) + body(f)))

txt = open(G + TAG + '_pmc_option_lstm_kernels.txt').read()
S3 = 'SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES'
K = {'fwd': 'false, EpiLstmFwdT<0', 'bwd': 'false, EpiLstmBwd<2,', 'dWh': 'true, EpiAtomic<4>'}
ALG = {'fwd': 496, 'bwd': 660, 'dWh': 3900}


def grab(section, kern, ctr):
    sec = txt[txt.index('## ' + section):]
    nxt = sec.find('\n## ', 3)
    sec = sec[:nxt] if nxt > 0 else sec
    m = re.search(r'gemm_f32_glds_kernel<[^\n]*' + re.escape(kern) + r'[^\n]*\n(?:    .*\n)*?    ' + ctr + r'\s+avg ([0-9.e+]+)', sec)
    return float(m.group(1))


summ, lines = json.load(open(P + 'pmc_summary.json')), []
for k in ('fwd', 'bwd', 'dWh'):
    fe, wr = grab('FETCH_SIZE', K[k], 'FETCH_SIZE') / 1e3, grab('WRITE_SIZE', K[k], 'WRITE_SIZE') / 1e3
    g, mb = grab(S3, K[k], 'GRBM_GUI_ACTIVE'), grab(S3, K[k], 'SQ_VALU_MFMA_BUSY_CYCLES')
    tot = 2 * fe + wr
    lines.append("# %-4s FETCH %.1f MB raw (x2 = %.0f MB) + WRITE %.0f MB -> %.0f MB HBM-side per launch vs %d MB algorithmic (%.2fx); GRBM %.4g "
                 "cycles; MFMA busy %.1f %%" % (k, fe, 2 * fe, wr, tot, ALG[k], tot / ALG[k], g, 100 * mb / (g * 32)))
    e = summ['opt_lstm_' + k]
    e.update(hbm_bytes_per_launch=int(round(tot * 1e6)), fetch_raw_bytes=int(round(fe * 1e6)), write_bytes=int(round(wr * 1e6)),
             mfma_busy=round(mb / (g * 32), 3), grbm_cycles=g)
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_digest
summ['csrc_sha256'] = csrc_digest()     # the kernel sources the PMC pass was taken on: bench.py reports roofline.traffic only while they match
json.dump(summ, open(P + 'pmc_summary.json', 'w'), indent=1)
old = open(P + TAG + '_pmc_option_lstm_kernels.txt').read().split('\n')
out, keep = [], False
for l in [l for l in txt.split('\n') if not l.startswith('# ')]:
    if l.startswith('## '):
        out.append(l)
        keep = False
        continue
    if not l.startswith('    '):
        keep = l.startswith('gemm_f32')
    if keep:
        out.append(l)
open(P + TAG + '_pmc_option_lstm_kernels.txt', 'w').write('\n'.join(old[:2] + out + [l for l in old if l.startswith('## how to read')] + lines) + '\n')
print('\n'.join(lines))
print('bench: %.3f ms/step, %.0f QA-rounds/s, frac %.3f, alone %.3f' % (d['ms_per_step'], d['value'], r['frac'], r['alone']['frac']))
