"""bench.py's bookkeeping without a GPU: the roofline blocks (what `frac` divides by, per arithmetic), the stored-traffic lookup (tied to the
kernel sources by their digest) and the GPU_MAX_HW_QUEUES probe that multi-GPU runs use (children mocked).  The timed path itself needs a
GPU (bench.py asserts it) and is exercised by scripts/measure_round.sh."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def bench(monkeypatch):
    monkeypatch.setenv('VD_BENCH_PROBE', '1')          # importing bench.py must not probe or touch the environment's queue setting
    sys.modules.pop('bench', None)
    monkeypatch.setattr(sys, 'argv', ['bench.py'])
    import bench as b
    yield b
    sys.modules.pop('bench', None)


def _fams(ms=(7.2, 9.0, 6.2)):
    N, O, H, To, E = 200, 100, 512, 20, 300
    NO = N * O
    ex = 2.0 * NO * H * 4 * H * (To - 1)
    out = {}
    for tag, t, launches in zip(('opt_lstm_fwd', 'opt_lstm_bwd', 'opt_lstm_dWh'), ms, (To - 1, To - 1, 1)):
        out[tag] = dict(ms_total_per_step=t, kernel_launches_per_step=launches, avg_launch_ms=t / launches, gflop_executed_per_launch=ex / launches / 1e9,
                        tflops_nominal=ex / t / 1e9, tflops_executed=ex / t / 1e9)
    return out


def test_split_roofline_prices_nine_bf16_products_against_the_bf16_peak(bench):
    f = _fams()
    r = bench.split_roofline(f, 'opt_lstm_bwd', 'split9', traffic=811e6)
    a32 = f['opt_lstm_bwd']['tflops_executed']
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] == 2500.0 and r['kernel'] == 'opt_lstm_bwd'
    assert abs(r['achieved'] - 9 * a32) < 0.01 and abs(r['frac'] - 9 * a32 / 2500.0) < 1e-4
    assert r['sustained_peak'] == bench.BF16_MFMA_SUSTAINED_TFLOPS and r['frac_of_sustained_peak'] > r['frac']
    assert abs(r['fp32_equivalent_tflops'] - a32) < 0.01 and r['traffic'] == 811e6
    assert bench.split_roofline(f, 'opt_lstm_fwd', 'split6')['achieved'] == pytest.approx(6 * f['opt_lstm_fwd']['tflops_executed'], abs=0.01)


def test_fp32_roofline_is_the_block_of_rounds_1_to_4(bench):
    f = _fams((8.1, 9.3, 6.6))
    r = bench.fp32_roofline(f, 'opt_lstm_bwd', 8100.0, traffic=918e6, traffic_build='abc')
    assert r['peak'] == 157.3 and abs(r['frac'] - f['opt_lstm_bwd']['tflops_executed'] / 157.3) < 1e-4
    assert r['traffic'] == 918e6 and r['traffic_build']['pmc_taken_on_csrc'] == 'abc' and set(r['families']) == set(f)


def test_stored_traffic_is_reported_only_for_the_build_it_was_measured_on(bench, tmp_path, monkeypatch):
    summ = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_summary.json')))
    assert {'opt_lstm_fwd', 'opt_lstm_bwd', 'opt_lstm_dWh', 'split9:opt_lstm_fwd', 'split9:opt_lstm_bwd', 'split9:opt_lstm_dWh'} <= set(summ)
    lookup, build = bench.stored_traffic()
    assert build == summ['csrc_sha256']
    if build == bench.csrc_digest():
        assert lookup('split9:opt_lstm_bwd') == summ['split9:opt_lstm_bwd']['hbm_bytes_per_launch'] and lookup('nope') is None
    monkeypatch.setattr(bench, 'csrc_digest', lambda: 'different-sources')
    lookup, _ = bench.stored_traffic()
    assert lookup('opt_lstm_bwd') is None


def test_hw_queue_probe_picks_the_faster_setting_and_survives_failures(bench, monkeypatch):
    calls = []

    class R(object):
        def __init__(self, rc, out, err=''):
            self.returncode, self.stdout, self.stderr = rc, out, err

    def fake_run(cmd, env=None, **kw):
        calls.append(env)
        assert env['VD_BENCH_PROBE'] == '1' and 'TORCHELASTIC_USE_AGENT_STORE' not in env and env['MASTER_PORT'] != os.environ.get('MASTER_PORT')
        one = env.get('GPU_MAX_HW_QUEUES') == '1'
        return R(0, 'noise\nPROBE_MS %s\n' % ('23.1' if one else '24.0'))
    monkeypatch.setenv('RANK', '3')
    monkeypatch.setenv('MASTER_PORT', '29533')
    monkeypatch.setenv('TORCHELASTIC_USE_AGENT_STORE', 'True')
    monkeypatch.setattr(subprocess, 'run', fake_run)
    choice, report = bench._probe_hw_queues(8)
    assert choice == '1' and "'1': 23.1" in report and len(calls) == 2
    assert calls[0]['MASTER_PORT'] != calls[1]['MASTER_PORT'] and 'GPU_MAX_HW_QUEUES' not in calls[1]
    monkeypatch.setattr(subprocess, 'run', lambda cmd, env=None, **kw: R(0, 'PROBE_MS %s\n' % ('25.0' if env.get('GPU_MAX_HW_QUEUES') == '1' else '24.0')))
    assert bench._probe_hw_queues(8)[0] == 'default'
    monkeypatch.setattr(subprocess, 'run', lambda cmd, env=None, **kw: R(1, '', 'boom'))
    choice, report = bench._probe_hw_queues(8)
    assert choice is None and 'failed' in report                          # -> the caller keeps HIP's default

    def raising(cmd, env=None, **kw):
        raise subprocess.TimeoutExpired(cmd, 180)
    monkeypatch.setattr(subprocess, 'run', raising)
    assert bench._probe_hw_queues(8)[0] is None


def test_bench_configs_name_the_baseline_workloads(bench):
    assert bench.config_params(3)['encoder'] == 'mn-att-ques-im-hist' and bench.config_params(3)['decoder'] == 'disc'
    assert bench.config_params(1)['decoder'] == 'gen' and bench.config_params(4)['lstmPrecision'] == 'bf16'
    assert bench.config_params(2)['lstmPrecision'] == 'split9' and bench.config_params(3).get('lstmPrecision', 'fp32') == 'fp32'   # the headline sets it from --recurrence
