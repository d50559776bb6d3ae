"""bench.py's bookkeeping without a GPU: the roofline blocks (what `frac` divides by, per arithmetic), the stored-traffic lookup (tied to the
kernel sources by their digest), the hardware-queue policy (no probe jobs: fixed per world size, overridable) and the startup deadline.  The timed path itself needs a
GPU (bench.py asserts it) and is exercised by scripts/measure_round.sh."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def bench(monkeypatch):
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', 'as-found')     # an explicit setting wins: importing bench.py leaves the environment alone
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


def _import_bench(monkeypatch, argv, **env):
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', 'x')
    monkeypatch.delenv('GPU_MAX_HW_QUEUES')            # (set-then-delete: monkeypatch restores "absent" afterwards, whatever the import does)
    for k in ('VD_BENCH_HW_QUEUES', 'WORLD_SIZE', 'RANK'):
        monkeypatch.setenv(k, 'x')
        monkeypatch.delenv(k)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sys.modules.pop('bench', None)
    monkeypatch.setattr(sys, 'argv', ['bench.py'] + argv)
    import bench as b
    sys.modules.pop('bench', None)
    return b


def test_hw_queue_policy_needs_no_probe_jobs(monkeypatch):
    """world 1: the measured single-queue setting; world > 1: HIP's default (never measured with RCCL on the queue) -- decided from argv /
    the environment alone, identically on every rank, with nothing launched before the timed job"""
    b = _import_bench(monkeypatch, [])
    assert os.environ['GPU_MAX_HW_QUEUES'] == '1' and 'single' in b.QUEUE_CHOICE
    assert not hasattr(b, '_probe_hw_queues')
    b = _import_bench(monkeypatch, ['--gpus', '8'], RANK='3', WORLD_SIZE='8')
    assert 'GPU_MAX_HW_QUEUES' not in os.environ and 'HIP default' in b.QUEUE_CHOICE
    b = _import_bench(monkeypatch, ['--gpus', '8'], RANK='3', WORLD_SIZE='8', VD_BENCH_HW_QUEUES='2')
    assert os.environ['GPU_MAX_HW_QUEUES'] == '2' and b.QUEUE_CHOICE == 'VD_BENCH_HW_QUEUES=2'
    b = _import_bench(monkeypatch, ['--host', 'python'])
    assert 'GPU_MAX_HW_QUEUES' not in os.environ
    src = open(os.path.join(ROOT, 'bench.py')).read()
    assert 'subprocess' not in src                      # no child jobs of any kind


def test_startup_deadline_exits_instead_of_hanging(tmp_path):
    """the cap on everything before the timed region: a process stuck in its start-up phase leaves with code 3 and says where it was"""
    code = ("import sys, time; sys.argv=['bench.py']; sys.path.insert(0, %r); import bench\n"
            "d = bench.StartupDeadline(0.3); d.phase = 'rendezvous / communicator'; time.sleep(5); print('NOT REACHED')" % ROOT)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3 and 'rendezvous / communicator' in r.stderr and 'NOT REACHED' not in r.stdout
    code = ("import sys, time; sys.argv=['bench.py']; sys.path.insert(0, %r); import bench\n"
            "d = bench.StartupDeadline(0.3); s = d.disarm(); time.sleep(0.6); print('OK', s < 0.3)" % ROOT)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and 'OK True' in r.stdout


def test_bench_configs_name_the_baseline_workloads(bench):
    assert bench.config_params(3)['encoder'] == 'mn-att-ques-im-hist' and bench.config_params(3)['decoder'] == 'disc'
    assert bench.config_params(1)['decoder'] == 'gen' and bench.config_params(4)['lstmPrecision'] == 'bf16'
    assert bench.config_params(2)['lstmPrecision'] == 'split9' and bench.config_params(3)['lstmPrecision'] == 'split9'   # the library default; bench.py sets it again from --recurrence
