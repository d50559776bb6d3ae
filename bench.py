#!/usr/bin/env python
"""bench.py -- QA-rounds/s of one full training step (zero-grad, fresh batch, forward, backward, [RCCL grad
all-reduce], clamp +-5, Adam) of mn-att-ques-im-hist + disc on synthetic VisDial-v1.0-shaped batches
(BASELINE.json configs[3]: batch 20 dialogs x 10 rounds x 100 options, 14x14x512 pool5 map) per GPU.

  python bench.py --gpus N --steps K --warmup W

N > 1: when RANK is not in the environment the script re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` (one rank per GPU over
RCCL); launched by the driver through torch.distributed.run it uses the environment it is given.

Prints ONE JSON line on rank 0.  Weak scaling: every rank trains its own 20 dialogs; gradients are summed over
RCCL/xGMI, averaged, clamped and applied identically on every rank.  Every timed step is the
reference's `Model:trainIteration` (model.lua:66-106): it draws a NEW batch (the host prepares and uploads the next
batch on a copy stream while the device executes the current step), so input preparation is inside the timed
region.  The oracle is used ONLY for the `cpu_baseline` leg (oracle/cpu_step.cpp, the C++17/OpenMP fp32 restatement
on the host cores), never for the measured path.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _host_arg(argv):
    for i, a in enumerate(argv):
        if a == '--host' and i + 1 < len(argv):
            return argv[i + 1]
        if a.startswith('--host='):
            return a.split('=', 1)[1]
    return os.environ.get('VD_BENCH_HOST', 'native')


def _world_arg(argv):
    w = int(os.environ.get('WORLD_SIZE', '1') or 1)
    for i, a in enumerate(argv):
        if a == '--gpus' and i + 1 < len(argv):
            w = max(w, int(argv[i + 1]))
        elif a.startswith('--gpus='):
            w = max(w, int(a.split('=', 1)[1]))
    return w


# The native host issues the whole step from one thread onto five library-owned HIP streams.  HIP multiplexes streams
# onto GPU_MAX_HW_QUEUES hardware queues (default 4); with all of them on ONE queue the cross-stream event waits
# resolve inside the command processor and the step is 3-4 % faster (profiles/r02_hw_queues.txt).
# Must be in the environment before HIP initialises, i.e. before torch touches the device; an explicit setting wins.
# (The Python operator-level host is slower that way -- 29.0 vs 26.8 ms -- so it keeps the default.)
# With peers (world > 1) RCCL's kernels share the hardware queues with the compute streams and the single-queue setting was never
# measured on a multi-GPU node: HIP's default is kept there (VD_BENCH_HW_QUEUES=<n> overrides, identically on every rank because it is
# read from the environment torch.distributed.run hands to all of them).  Rounds 4-5 chose by running probe JOBS per rank before the
# real run (own communicators, own ports); that was the most fragile code in the file and is gone: nothing runs before the timed job
# any more except its own warm-up, under the wall-clock cap below.
QUEUE_CHOICE = None
if _host_arg(sys.argv[1:]) == 'native' and 'GPU_MAX_HW_QUEUES' not in os.environ:
    if os.environ.get('VD_BENCH_HW_QUEUES'):
        os.environ['GPU_MAX_HW_QUEUES'] = os.environ['VD_BENCH_HW_QUEUES']
        QUEUE_CHOICE = 'VD_BENCH_HW_QUEUES=%s' % os.environ['VD_BENCH_HW_QUEUES']
    elif _world_arg(sys.argv[1:]) == 1:
        os.environ['GPU_MAX_HW_QUEUES'] = '1'
        QUEUE_CHOICE = 'single hardware queue (world 1, measured: profiles/r02_hw_queues.txt)'
    else:
        QUEUE_CHOICE = "HIP default (world > 1: never measured with RCCL's kernels on the queue)"


class StartupDeadline(object):
    """Hard wall-clock cap on everything between `import torch` and the timed region (rendezvous, RCCL communicator, model build, warm-up:
    normally < 30 s): a rank that is still not timing after `seconds` prints what it was doing and exits with code 3 -- a multi-GPU run can
    fail, it cannot hang.  (VD_BENCH_STARTUP_CAP_S / --startup-cap override.)"""

    def __init__(self, seconds):
        import threading
        self.phase, self.seconds, self.t0 = 'import', seconds, time.time()
        self._done = threading.Event()
        t = threading.Thread(target=self._watch, daemon=True)
        t.start()

    def _watch(self):
        if not self._done.wait(self.seconds):
            print('[rank %s] bench.py: startup exceeded %.0f s in phase "%s" -- giving up (exit 3)'
                  % (os.environ.get('RANK', '0'), self.seconds, self.phase), file=sys.stderr, flush=True)
            os._exit(3)

    def disarm(self):
        self._done.set()
        return time.time() - self.t0


FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2500.0         # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense
BF16_MFMA_SUSTAINED_TFLOPS = 1862.0    # MEASURED on non-zero operands (scripts/probes/mfma_bf16_peak.hip, profiles/r05_mfma_sustained.txt): the clock
                                       # follows the power budget (2 482 on zeros, 1 862 on random data; v_mfma_f32_32x32x2_f32 holds 154.6 of 157.3)
STEP_GFLOP_PER_ROUND = 21.934          # nominal dense math of the reference graph per QA round


def config_params(config, rank=0, batch=20):
    """BASELINE.json configs[config] at its quoted size (batch 20, V = 11 322, H = 512, E = 300):
      0 = lf-ques + gen, batch 8, no image features (the reference's CPU-runnable plumbing case, run on the GPU here);
      1 = lf-ques-im-hist + gen, VGG-16 fc7 (4096-d);            2 = hre-ques-im-hist + disc, fc7, 100 options (option recurrence on the exact
      split, like the headline);
      3 = mn-att-ques-im-hist + disc, 14x14x512 pool5 (HEADLINE); 4 = the same with ResNet-200 7x7x2048 features and bf16 operands
      in the option recurrence (informative, never the default)"""
    from visdial_amd.opts import default_params
    if config == 0:
        batch = 8                      # `th train.lua -encoder lf-ques -decoder gen -gpuid -1`, batch 8 -- here on the HIP path (there is no CPU product path)
    kw = {0: dict(encoder='lf-ques', decoder='gen'),
          1: dict(encoder='lf-ques-im-hist', decoder='gen', imgFeatureSize=4096),
          2: dict(encoder='hre-ques-im-hist', decoder='disc', imgFeatureSize=4096, lstmPrecision='split9'),
          3: dict(encoder='mn-att-ques-im-hist', decoder='disc', imgFeatureSize=512, imgSpatialSize=14),
          4: dict(encoder='mn-att-ques-im-hist', decoder='disc', imgFeatureSize=2048, imgSpatialSize=7, lstmPrecision='bf16')}[config]
    gpuid = 0 if os.environ.get('VD_BENCH_SHARE_GPU') == '1' else int(os.environ.get('LOCAL_RANK', 0))
    return default_params(batchSize=batch, vocabSize=11322, gpuid=gpuid, rank=rank,
                          maxHistoryLenPerRound=40, **kw)


def headline_params(rank=0, batch=20, config=3):
    return config_params(config, rank=rank, batch=batch)


def dominant_kernel_alone(p, N, iters=3, recurrence='fp32'):
    """The dominant kernel family (option-LSTM backward, To-1 timestep launches) run by itself on the same shapes, HIP
    events on its stream: the rate the kernel reaches when it does not share the matrix pipe with the encoder."""
    import torch
    from visdial_amd import ops
    NO, H, To = N * p['numOptions'], p['rnnHiddenSize'], p['maxAnsLen']
    g = torch.Generator(device='cuda').manual_seed(0)
    Wh = torch.randn(H, 4 * H, device='cuda', generator=g) * 0.04
    gates = torch.rand(To, NO, 4 * H, device='cuda', generator=g)
    c = torch.randn(To, NO, H, device='cuda', generator=g) * 0.1
    dcw = torch.empty(NO, H, device='cuda')
    dh_last = torch.randn(NO, H, device='cuda', generator=g) * 0.01
    flags = ops.PRECISION_FLAGS[recurrence]
    run = lambda: ops.lstm_backward(Wh, gates, c, dcw, To, NO, H, dh_last=dh_last, flags=flags)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters / (To - 1)
    tf = 2.0 * NO * H * 4 * H / ms / 1e9
    del gates, c
    if recurrence == 'fp32':
        return {"avg_launch_ms": round(ms, 4), "achieved": round(tf, 2), "frac": round(tf / FP32_MFMA_PEAK_TFLOPS, 4)}
    nprod = int(recurrence[-1])
    return {"avg_launch_ms": round(ms, 4), "achieved": round(nprod * tf, 2), "frac": round(nprod * tf / BF16_MFMA_PEAK_TFLOPS, 4),
            "fp32_equivalent_tflops": round(tf, 2)}


def csrc_digest():
    """SHA-256 over the kernel sources: ties a stored PMC measurement (profiles/pmc_summary.json) to the build it was taken on"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, 'visdial_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(ROOT, 'visdial_amd', 'csrc', '*.h'))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def option_families(p, N, prof, steps_for_prof):
    """per-family figures of the option LSTM (decoder disc) from HIP-event times: {tag: {...}}"""
    NO, E, H, To = N * p['numOptions'], p['embedSize'], p['rnnHiddenSize'], p['maxAnsLen']
    fams = {}
    for tag, nominal, executed, klaunch in (
            ('opt_lstm_fwd', 2.0 * NO * (E + H) * 4 * H * To, 2.0 * NO * H * 4 * H * (To - 1), To - 1),
            ('opt_lstm_bwd', 2.0 * NO * (E + H) * 4 * H * To, 2.0 * NO * H * 4 * H * (To - 1), To - 1),
            ('opt_lstm_dWh', 2.0 * NO * (To - 1) * H * 4 * H, 2.0 * NO * (To - 1) * H * 4 * H, 1)):
        if tag in prof:
            ms, n = prof[tag]                       # n family invocations (one per step) took ms in total
            per_family = ms / n
            fams[tag] = dict(ms_total_per_step=ms / steps_for_prof, kernel_launches_per_step=klaunch * n / steps_for_prof,
                             avg_launch_ms=per_family / klaunch, gflop_executed_per_launch=executed / klaunch / 1e9,
                             tflops_nominal=nominal / per_family / 1e9, tflops_executed=executed / per_family / 1e9)
    return fams


def bf16_option_roofline(fams, dom, rows=20000, H=512, To=20):
    """bf16 operands make the matrix work 16x cheaper: the recurrence is priced by its bytes.  ALGORITHMIC bytes per timestep launch over
    the COMPACT state of round 4: forward = bf16 table rows + bf16 gates written + bf16 h read and written + fp32 c
    read and written; backward = gates[t] read + da[t+1] read + da[t] written (bf16, in place) + c[t], c[t-1] read + dc read and written
    (fp32); dWh = the bf16 h and da streams once."""
    g16, h16, c32 = rows * 4 * H * 2, rows * H * 2, rows * H * 4
    per_launch = {'opt_lstm_fwd': 2 * g16 + 2 * h16 + 2 * c32, 'opt_lstm_bwd': 3 * g16 + 4 * c32, 'opt_lstm_dWh': (To - 1) * (g16 + h16)}[dom]
    launches = 1 if dom == 'opt_lstm_dWh' else To
    gbs = per_launch * launches / (fams[dom]['ms_total_per_step'] * 1e-3) / 1e9      # bytes of the whole family / its event time
    return {"bound": "hbm", "kernel": dom, "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
            "frac": round(gbs / 8000.0, 4), "traffic": None, "avg_launch_ms": round(fams[dom]['avg_launch_ms'], 4),
            "algorithmic_MB_per_launch": round(per_launch / 1e6, 1),
            "note": "bf16 operands / fp32 accumulation in the option recurrence over the compact bf16 state: priced by HBM bytes "
                    "(algorithmic bytes of the family / its HIP-event time); MFMA side: %.0f TFLOP/s of the 2 500 TFLOP/s dense bf16 "
                    "peak.  Neither bound is near: the cell update's VALU work and the CU's load path set the time (profiles/r04_experiments.txt section 2)"
                    % fams[dom]['tflops_executed'],
            "families": {k: {kk: round(vv, 4) for kk, vv in v.items()} for k, v in fams.items()}}


def stored_traffic():
    """-> (lookup(key) -> HBM-side bytes per launch or None, digest of the build the PMC passes were taken on).  profiles/pmc_summary.json is
    a STORED rocprofv3 measurement (2 x FETCH_SIZE + WRITE_SIZE per launch); it is reported only while the kernel sources are the ones it
    was taken on.  Keys: opt_lstm_fwd / _bwd / _dWh = the fp32-MFMA kernels, 'split9:opt_lstm_fwd' / '_bwd' = the split kernels."""
    pmc = os.path.join(ROOT, 'profiles', 'pmc_summary.json')
    try:
        summ = json.load(open(pmc))
    except Exception:
        return (lambda key: None), None
    build = summ.get('csrc_sha256')
    ok = build == csrc_digest()
    return (lambda key: (summ.get(key) or {}).get('hbm_bytes_per_launch') if ok else None), build


def split_roofline(fams, step_fam, recurrence, traffic=None):
    """the dominant timestep kernel of a split pass: executed bf16-MFMA FLOPs (nprod x the fp32 product's) against the 2.5 PFLOP/s dense peak"""
    nprod = int(recurrence[-1])
    a32 = fams[step_fam]['tflops_executed']
    return {"bound": "mfma", "kernel": step_fam, "achieved": round(nprod * a32, 2), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(nprod * a32 / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
            "sustained_peak": BF16_MFMA_SUSTAINED_TFLOPS, "frac_of_sustained_peak": round(nprod * a32 / BF16_MFMA_SUSTAINED_TFLOPS, 4),
            "fp32_equivalent_tflops": round(a32, 2), "fp32_equivalent_frac_of_fp32_mfma_peak": round(a32 / FP32_MFMA_PEAK_TFLOPS, 4),
            "avg_launch_ms": round(fams[step_fam]['avg_launch_ms'], 4),
            "note": "every fp32 recurrent product h*Wh / da*Wh^T issued as %d bf16 MFMAs (v_mfma_f32_32x32x16_bf16, fp32 accumulate) on the exact "
                    "hi/mid/lo split of BOTH operands (all 24 significand bits); achieved = executed bf16-MFMA FLOPs (%d x the fp32 product's) / "
                    "HIP-event launch time inside the overlapped step, priced against the 2.5 PFLOP/s dense bf16 peak; sustained_peak = what the "
                    "matrix pipe holds on non-zero operands under the power budget (measured, profiles/r05_mfma_sustained.txt) -- the bound "
                    "this kernel actually meets (profiles/r05_experiments.txt section 1); the dWh contraction h^T * da runs on the same split "
                    "(both operands split in registers: csrc/split_core.h gemm_split_tn_kernel)"
                    % (nprod, nprod),
            "families": {k: {kk: round(vv, 4) for kk, vv in v.items()} for k, v in fams.items()}}


def fp32_roofline(fams, dom, value, traffic=None, traffic_build=None):
    a = fams[dom]['tflops_executed']
    return {"bound": "mfma", "kernel": dom, "achieved": round(a, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(a / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
            "traffic_build": {"pmc_taken_on_csrc": traffic_build, "this_build_csrc": csrc_digest(),
                              "note": "traffic is null when the kernel sources changed since the PMC pass"},
            "achieved_nominal": round(fams[dom]['tflops_nominal'], 2),
            "avg_launch_ms": round(fams[dom]['avg_launch_ms'], 4),
            "note": "achieved = EXECUTED FLOPs of the dominant kernel per launch (the recurrent h*Wh / da*Wh^T "
                    "products actually issued on the matrix pipe) / its HIP-event launch time measured inside "
                    "the overlapped step; achieved_nominal also counts the x*Wx product of the reference graph "
                    "that this build replaces by an exact table gather (SURVEY 8d); traffic = HBM-side bytes per launch of the "
                    "same kernel on the same shapes from the committed rocprofv3 PMC passes (profiles/pmc_summary.json: 2 x "
                    "FETCH_SIZE + WRITE_SIZE), a stored measurement, not a live counter",
            "step_tflops_nominal": round(STEP_GFLOP_PER_ROUND * value / 1e3, 2),
            "families": {k: {kk: round(vv, 4) for kk, vv in v.items()} for k, v in fams.items()}}


def alt_leg(args, N):
    """The SAME timed protocol (W warm-up + K steps, a fresh batch every step) with the option recurrence on v_mfma_f32_32x32x2_f32 -- the
    arithmetic of the headline of rounds 1-4, kept beside the split9 headline in the same JSON line.  Single GPU only."""
    from visdial_amd.dataloader import SyntheticDataloader
    from visdial_amd.native import NativeModel
    p = headline_params(batch=args.batch, config=3)
    p['lstmPrecision'] = 'fp32'
    model = NativeModel(p)
    dl = SyntheticDataloader(p, seed=1234, fast=True)
    for _ in range(args.warmup):
        model.trainIteration(dl)
    model.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = model.trainIteration(dl)
    model.synchronize()
    elapsed = time.perf_counter() - t0
    f = model.family_ms()
    fams = option_families(p, N, {'opt_lstm_fwd': (f[0], 1), 'opt_lstm_bwd': (f[1], 1), 'opt_lstm_dWh': (f[2], 1)}, 1)
    dom = max(fams, key=lambda k: fams[k]['ms_total_per_step'])
    value = N * args.steps / elapsed
    traffic_of, traffic_build = stored_traffic()
    roof = fp32_roofline(fams, dom, value, traffic_of(dom), traffic_build)
    try:
        roof["alone"] = dominant_kernel_alone(p, N)
    except Exception as exc:
        roof["alone"] = {"error": str(exc)[:120]}
    model.close()
    return {"dtype": "f32 (v_mfma_f32_32x32x2_f32 in the option recurrence as everywhere else)", "recurrence": "fp32",
            "value": round(value, 2), "unit": "QA-rounds/s", "ms_per_step": round(elapsed / args.steps * 1e3, 3), "steps": args.steps,
            "warmup": args.warmup, "loss": round(float(loss), 5), "roofline": roof}


def other_config(cfg, steps=10, warmup=3):
    """one of the non-headline single-GPU configurations (BASELINE.json configs[1], [2], [4]): ms/step of the same pipelined
    trainIteration + the roofline of ITS dominant kernel family from the library's HIP events in the last step"""
    import numpy as np
    from visdial_amd.dataloader import SyntheticDataloader
    from visdial_amd.native import NativeModel
    p = config_params(cfg)
    model = NativeModel(p)
    dl = SyntheticDataloader(p, seed=4321, fast=True)
    served = []
    inner = dl.getTrainBatch

    def recording(params, **kw):
        b = inner(params, **kw)
        served.append(b)
        del served[:-3]
        return b
    dl.getTrainBatch = recording
    for _ in range(warmup):
        model.trainIteration(dl)
    model.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = model.trainIteration(dl)
    model.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    f = model.family_ms()
    N, H = p['batchSize'] * p['maxQuesCount'], p['rnnHiddenSize']
    out = {"workload": "BASELINE.json configs[%d]: %s + %s, batch %d, %s" % (
               cfg, p['encoder'], p['decoder'], p['batchSize'],
               {0: "synthetic tokens, no image features (the reference runs this one on CPU: -gpuid -1)", 1: "VGG-16 fc7 4096-d features",
                2: "fc7 4096-d features, 100 options", 4: "ResNet-200 7x7x2048 features, 100 options"}[cfg]),
           "dtype": {0: "f32", 1: "f32", 2: "f32 operands and results; option recurrence on the exact 3-way bf16 split (9 products, f32 accumulate), f32 MFMA elsewhere",
                     4: "bf16 operands / f32 accumulate (option recurrence only), f32 elsewhere"}[cfg],
           "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 3), "qa_rounds_per_s": round(N / ms * 1e3, 1),
           "loss": round(float(loss), 5)}
    if p['decoder'] == 'disc':
        fams = option_families(p, N, {'opt_lstm_fwd': (f[0], 1), 'opt_lstm_bwd': (f[1], 1), 'opt_lstm_dWh': (f[2], 1)}, 1)
        dom = max(fams, key=lambda k: fams[k]['ms_total_per_step'])
        if cfg == 4:
            out["roofline"] = bf16_option_roofline(fams, dom, rows=N * p['numOptions'], H=H, To=p['maxAnsLen'])
        elif p.get('lstmPrecision') == 'split9':
            out["roofline"] = split_roofline(fams, dom, 'split9')
        else:
            a = fams[dom]['tflops_executed']
            out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": round(a, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(a / FP32_MFMA_PEAK_TFLOPS, 4), "avg_launch_ms": round(fams[dom]['avg_launch_ms'], 4),
                               "families": {k: {kk: round(vv, 4) for kk, vv in v.items()} for k, v in fams.items()}}
        out["option_rows_executed_of_total"] = list(model.option_rows())
    elif 'hist' not in served[-1]:
        # configs[0]: 80 rows per step, ~230 launches -- launch- and latency-bound plumbing.  The only sizeable product is the vocabulary
        # projection with its two gradients; priced against the WHOLE step's time (a lower bound on its rate: no family events here)
        b = served[-2] if len(served) >= 2 else served[-1]
        gflop = 3 * 2.0 * np.asarray(b['answer_in']).size * p['vocabSize'] * H / 1e9
        out["roofline"] = {"bound": "mfma", "kernel": "vocab (projection + 2 gradients) over the whole step's time", "achieved": round(gflop / ms, 2),
                           "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(gflop / ms / FP32_MFMA_PEAK_TFLOPS, 4),
                           "gflop_executed": round(gflop, 2),
                           "note": "latency-bound: 80 rows per step; the figure is the vocabulary family's FLOPs / the step time, not a kernel rate"}
    else:
        # gen over a Sequential encoder: the history branch (two-layer length-sorted wavefront, Th + 1 ticks per direction) is the
        # step.  Executed FLOPs of the batch the LAST step trained on: per tick and active row h*Wh1, h1*Wx2, h2*Wh2 (forward; twice
        # that backward: da*Wh^T products + ... counted as the reference does: bwd = the same three products transposed).
        b = served[-2] if len(served) >= 2 else served[-1]          # the last step trained on the batch BEFORE the prefetched one
        hist = np.asarray(b['hist']).reshape(-1, b['hist'].shape[-1])
        lens = (hist != 0).sum(1)
        Th = hist.shape[1]
        nact = np.array([(lens >= Th - t).sum() for t in range(Th)])           # right-aligned: row active from step Th - len on
        flop = 2.0 * nact.sum() * 3 * H * 4 * H
        fam = {"hist_fwd": {"ms": round(f[0], 4), "ticks": Th + 1, "gflop_executed": round(flop / 1e9, 2), "tflops": round(flop / max(f[0], 1e-6) / 1e9, 2)},
               "hist_bwd": {"ms": round(f[1], 4), "ticks": Th + 1, "gflop_executed": round(flop / 1e9, 2), "tflops": round(flop / max(f[1], 1e-6) / 1e9, 2)},
               "vocab": {"ms": round(f[2], 4), "gflop_executed": round(3 * 2.0 * np.asarray(b['answer_in']).size * p['vocabSize'] * H / 1e9, 2)}}
        fam["vocab"]["tflops"] = round(fam["vocab"]["gflop_executed"] / max(f[2], 1e-6), 2)
        dom = max(fam, key=lambda k: fam[k]['ms'])
        out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": fam[dom]['tflops'], "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(fam[dom]['tflops'] / FP32_MFMA_PEAK_TFLOPS, 4),
                           "avg_launch_ms": round(fam[dom]['ms'] / fam[dom].get('ticks', 1), 4),
                           "note": "a chain of Th + 1 DEPENDENT launches over <= 200 rows each (N = B x 10 rounds): latency-bound, not a "
                                   "throughput kernel; executed FLOPs = active (t, row) pairs x (h*Wh1 + h1*Wx2 + h2*Wh2)",
                           "families": fam, "history_steps": int(Th), "mean_active_rows": round(float(nact.mean()), 1)}
    model.close()
    return out


def cpu_baseline(seconds_budget=30.0, batch=20):
    """oracle/cpu_step.cpp -- the repo's C++17/OpenMP fp32 restatement of the same training step, organised like the
    reference's CPU path (per-timestep GEMMs, 10x image replication, no table hoist) -- timed on all host cores on
    the IDENTICAL synthetic batch shape (B = 20 dialogs, seed 1234), dropout on, update included."""
    import numpy as np
    from oracle import cpu_step, visdial_oracle as vo          # checker / baseline only
    from visdial_amd.dataloader import SyntheticDataloader
    p = headline_params(batch=batch)
    threads = cpu_step.fit_threads_to_quota()       # the boxes run the container under a CPU quota (see cpu_step.cpu_quota)
    quota = cpu_step.cpu_quota()
    dl = SyntheticDataloader(p, seed=1234, fast=True)
    b = dl.getTrainBatch(p)
    P = vo.init_params(p['encoder'], p['decoder'], p, seed=1234, dtype=np.float32)
    cs = cpu_step.CpuStep(p, vo.param_spec(p['encoder'], p['decoder'], p), P)
    B, R, Tq = b['ques_fwd'].shape
    Th = b['hist'].shape[2]
    N, H, E, S2, K = B * R, p['rnnHiddenSize'], p['embedSize'], p['imgSpatialSize'] ** 2, p['commonEmbeddingSize']
    rng = np.random.RandomState(0)
    shp = dict(q_emb=(Tq, N, E), h_emb=(Th, N, E), hatt=(N, H), img_tr=(N, S2, H), iqc=(N, S2, K), u=(N, H))
    drop = {k: (rng.rand(*s) > 0.5).astype(np.uint8) for k, s in shp.items()}
    t0 = time.time()
    cs.step(b, drop, update=True)                                                      # warm-up
    first = time.time() - t0
    n = int(max(0, min(5, (seconds_budget - first) // max(first, 1e-3))))
    times = []
    for _ in range(n):
        t0 = time.time()
        cs.step(b, drop, update=True)
        times.append(time.time() - t0)
    dt = float(np.median(times)) if times else first
    cores = int(round(quota)) if quota else (os.cpu_count() or threads)   # what the container may actually use
    return {"value": round(N / dt, 3), "unit": "QA-rounds/s", "cores": cores, "threads": threads, "kind": "port",
            "sample": "oracle/cpu_step.cpp: C++17/OpenMP fp32 restatement of the step (per-timestep GEMMs on the %s "
                      "kernel, 10x image replication, clamp+adam), identical shape B=%d dialogs x 10 rounds x 100 "
                      "options, %s, %.2f s/step; %d OpenMP threads, os.cpu_count=%d, container CPU quota=%s; restated "
                      "CPU baseline, not Torch7"
                      % (cpu_step.gemm_kernel(), batch,
                         ("median of %d timed steps after 1 warm-up" % n) if times else "the single warm-up step only",
                         dt, threads, os.cpu_count() or 0, ("%.0f CPUs" % quota) if quota else "none")}


def comm_self_check(model, collective, lib_comm, local):
    """what this rank's gradient exchange looked like in the warm-up steps"""
    import torch
    from visdial_amd.parallel import library_comm_available, library_comm_stats
    out = {'device': torch.cuda.get_device_name(local), 'collective': collective}
    if lib_comm:
        out['rccl_version_code'] = library_comm_available()[1]
        out.update(library_comm_stats())
        out['bucket1_MB'] = round(out['bucket1_floats'] * 4 / 1e6, 2)
        out['bucket2_MB'] = round(out['bucket2_floats'] * 4 / 1e6, 2)
    else:
        out['rccl_version_code'] = '.'.join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, 'nccl') else None
        sl = getattr(model, '_enc_slice', None)
        n = int(model.wrapperdW.numel()) if hasattr(model, 'wrapperdW') else None
        if sl and n:
            out.update(bucket1_floats=sl[1] - sl[0], bucket2_floats=n - (sl[1] - sl[0]), overlapped=True)
    return out


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` with N > 1 and no rendezvous environment: become N ranks."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=20, help='dialogs per GPU (headline: 20)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the configs[0], [1], [2], [4] legs after the headline')
    ap.add_argument('--same-batch', action='store_true', help='reuse one resident batch (round-1 behaviour; A/B only)')
    ap.add_argument('--no-streams', action='store_true', help='whole step on one HIP stream (A/B only)')
    ap.add_argument('--config', type=int, choices=[3, 4], default=3,
                    help='BASELINE.json configs index: 3 = headline (fp32, 14x14x512); 4 = 7x7x2048 features + bf16 option recurrence')
    ap.add_argument('--recurrence', choices=['fp32', 'split9', 'split6'], default='split9',
                    help='arithmetic of the option recurrence at --config 3: split9 (DEFAULT, the headline AND the default of the library / CLIs) = every fp32 operand as the exact sum '
                         'of three bf16 values, all 9 bf16 MFMA products, fp32 accumulate -- fp32-grade results (errors at the fp32 MFMA\'s own level, 0.5-1.4x per tensor: '
                         'tests/test_ops_gpu.py::test_split_error_table, tests/test_full_size_golden.py); fp32 = v_mfma_f32_32x32x2_f32 (reported '
                         'beside the headline as `alt`); split6 = 6 products (data only, never a headline)')
    ap.add_argument('--no-alt', action='store_true', help='skip the `alt` leg (the same steps with the fp32-MFMA recurrence) after the headline')
    ap.add_argument('--collective', choices=['library', 'torch'], default='library',
                    help='native host, N > 1: library = RCCL behind the C ABI (default); torch = host-side torch.distributed')
    ap.add_argument('--host', choices=['python', 'native'], default=os.environ.get('VD_BENCH_HOST', 'native'),
                    help='native (default) = the model-level ABI (csrc/runtime.hip: the orchestration a Lua host gets); '
                         'python = visdial_amd.Model composing the operator-level ABI')
    ap.add_argument('--startup-cap', type=float, default=float(os.environ.get('VD_BENCH_STARTUP_CAP_S', 600)),
                    help='seconds allowed between process start and the timed region (rendezvous + communicator + warm-up); exit 3 beyond')
    args = ap.parse_args()

    if args.gpus > 1 and 'RANK' not in os.environ:
        respawn_under_torchrun(args)
    import numpy as np
    import torch                  # (not under the cap: the first import on a fresh box pages the image in for 1-2 minutes)
    import torch.distributed as dist

    deadline = StartupDeadline(args.startup_cap)

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    # VD_BENCH_SHARE_GPU=1: DRY RUN of the multi-rank path on a one-GPU box -- every rank on cuda:0, gloo process group, gradients staged
    # through host memory (RCCL refuses duplicate devices).  Exercises sharding, barriers, the max-over-ranks clock and the JSON line;
    # its number means nothing.
    share_gpu = os.environ.get('VD_BENCH_SHARE_GPU') == '1'
    if share_gpu:
        local = 0
    deadline.phase = 'rendezvous / communicator'
    if world > 1:
        print('[rank %d] hardware queues: %s' % (rank, QUEUE_CHOICE), file=sys.stderr, flush=True)
    assert torch.cuda.is_available(), "bench.py measures the HIP path; it needs a GPU"
    torch.cuda.set_device(local)
    group = None
    collective = None
    lib_comm = False
    if 'RANK' in os.environ:      # launched by torch.distributed.run (any world size, incl. 1)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if share_gpu:
            dist.init_process_group(backend='gloo')
            group = dist.group.WORLD
            collective = 'gloo, gradients staged through host memory (VD_BENCH_SHARE_GPU dry run: not a measurement)'
        elif args.host == 'native' and args.collective == 'library':
            # The gradient all-reduce is the LIBRARY's: RCCL communicator + communication stream behind the C ABI
            # (csrc/comm.hip, vd_model_allreduce_grads).  torch.distributed (gloo) is only the courier of the 128-byte
            # rendezvous token, the barriers and the max-over-ranks of the wall time.
            dist.init_process_group(backend='gloo')
            group = dist.group.WORLD
            from visdial_amd.parallel import join_library_comm
            # agreed phases (RCCL loadable? -> token -> init): every rank joins or none does, nobody is left inside a collective
            lib_comm, join_report = join_library_comm(group, device=local)
            for line in join_report:
                print('[rank %d] library communicator: %s' % (rank, line), file=sys.stderr, flush=True)
            collective = 'library RCCL (vd_model_allreduce_grads; 2 buckets, library comm stream)'
            if not lib_comm:
                group = dist.new_group(backend='nccl', device_id=torch.device('cuda', local))
                collective = 'torch.distributed nccl (fallback: library communicator failed)'
        else:
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local))   # nccl == RCCL on ROCm
            group = dist.group.WORLD
            collective = 'torch.distributed nccl (host-side, 2 buckets)'
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)

    from visdial_amd import ops
    from visdial_amd.dataloader import SyntheticDataloader
    from visdial_amd.model import Model

    deadline.phase = 'model build + warm-up'
    p = headline_params(rank=rank, batch=args.batch, config=args.config)
    if args.config != 3:
        args.recurrence = 'fp32'                # (--recurrence is the arithmetic of the fp32-grade headline; configs[4] is the bf16 pass)
    else:
        p['lstmPrecision'] = args.recurrence    # (split9 is also the library / CLI default: opts.py)
    if args.no_streams:
        p['useStreams'] = 0
    if args.host == 'native':
        from visdial_amd.native import NativeModel
        model = NativeModel(p, dist_group=None if lib_comm else group, library_comm=lib_comm)
    else:
        model = Model(p, dist_group=group)
    dl = SyntheticDataloader(p, seed=1234 + rank, fast=True)
    N = p['batchSize'] * p['maxQuesCount']
    if args.same_batch:
        fixed = dl.getTrainBatch(p)
        dl.getTrainBatch = lambda params, **kw: fixed

    for _ in range(args.warmup):
        loss = model.trainIteration(dl)
    torch.cuda.synchronize()
    if group is not None:
        # per-rank self-check, so that a first multi-GPU run is diagnosable from its log: which collective, which RCCL, what the
        # two buckets hold and whether the encoder bucket really went out early (under the decoder's backward)
        print('[rank %d] %s' % (rank, json.dumps(comm_self_check(model, collective, lib_comm, local))), file=sys.stderr, flush=True)
        dist.barrier(group=group)
    torch.cuda.synchronize()
    startup_s = deadline.disarm()
    ops.PROFILE = {}
    per_step = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        loss = model.trainIteration(dl)        # returns after this step's loss has left the device
        per_step.append(time.perf_counter() - ts)
    torch.cuda.synchronize()
    getattr(model, 'synchronize', lambda: None)()     # the native host's streams are not torch's
    if group is not None:
        dist.barrier(group=group)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = ops.prof_summary()
    ops.PROFILE = None
    if args.host == 'native':       # HIP events recorded by the library around the three families in the LAST step
        f = model.family_ms()
        prof = {'opt_lstm_fwd': (f[0], 1), 'opt_lstm_bwd': (f[1], 1), 'opt_lstm_dWh': (f[2], 1)}
        args_steps_for_prof = 1
    else:
        args_steps_for_prof = args.steps
    if group is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if dist.get_backend(group) == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * N * args.steps / elapsed
        # dominant kernel family = the option-LSTM recurrence (fused recurrent GEMM + cell update): To-1 kernel launches
        # per direction and step (the first step has h0 = 0 and no recurrent product), one launch for dWh.  The HIP
        # events bracket the whole family on its stream; figures are per KERNEL LAUNCH so they can be compared with
        # the rocprofv3 average duration of the same kernel (profiles/r02_kernel_stats_bench.txt).
        fams = option_families(p, N, prof, args_steps_for_prof)
        dom = max(fams, key=lambda k: fams[k]['ms_total_per_step']) if fams else None
        # HBM-side bytes per launch of the dominant kernel: a STORED rocprofv3 PMC measurement, reported only while the kernel
        # sources are the ones it was taken on (profiles/pmc_summary.json carries their digest)
        traffic_of, traffic_build = stored_traffic()
        traffic = traffic_of(dom)
        roof = None
        if dom and args.config == 4:
            roof = bf16_option_roofline(fams, dom, rows=N * p['numOptions'], H=p['rnnHiddenSize'], To=p['maxAnsLen'])
        elif dom and args.recurrence != 'fp32':
            # the dWh contraction runs on the split only for split9 at shapes gemm_split_tn_kernel takes (csrc/gemm_ops.hip: M % 256 == 0,
            # N % 128 == 0, K >= 8192 rows); otherwise it is an fp32-MFMA launch and must not be priced as nprod bf16 products
            dwh_on_split = args.recurrence == 'split9' and p['rnnHiddenSize'] % 256 == 0 and (p['maxAnsLen'] - 1) * N * p['numOptions'] >= 8192
            if dom == 'opt_lstm_dWh' and not dwh_on_split:
                dom = max(('opt_lstm_fwd', 'opt_lstm_bwd'), key=lambda k: fams[k]['ms_total_per_step'])
            roof = split_roofline(fams, dom, args.recurrence, traffic_of(args.recurrence + ':' + dom))
        elif dom:
            roof = fp32_roofline(fams, dom, value, traffic, traffic_build)
        if roof and roof.get("bound") == "mfma" and world == 1 and args.config == 3:
            try:
                roof["alone"] = dominant_kernel_alone(p, N, recurrence=args.recurrence)   # same kernel, same shapes, nothing else on the chip
            except Exception as exc:                            # never let the extra figure break the bench line
                roof["alone"] = {"error": str(exc)[:120]}
        ps = np.array(per_step) * 1e3
        out = {
            "metric": "QA-rounds/sec training mn-att-ques-im-hist+disc (batch 20x10x100)",
            "value": round(value, 2), "unit": "QA-rounds/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "ms_per_step_median": round(float(np.median(ps)), 3), "ms_per_step_p10_p90": [round(float(np.percentile(ps, 10)), 3),
                                                                                            round(float(np.percentile(ps, 90)), 3)],
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32" if args.recurrence == 'fp32' else
                      "f32 operands and results; the option recurrence multiplies the EXACT 3-way bf16 split of both operands (%s bf16 MFMA "
                      "products per fp32 product, f32 accumulate: fp32-grade, errors at the f32 MFMA's own level) -- its two step kernels and its "
                      "weight-gradient contraction; f32 MFMA (v_mfma_f32_32x32x2_f32) everywhere else; the all-f32-MFMA line of the same run is "
                      "under `alt`" % args.recurrence[-1]) if args.config == 3
            else "bf16 operands / f32 accumulate (option recurrence only), f32 elsewhere",
            "data": "synthetic" + (" (one resident batch reused)" if args.same_batch else
                                   " (a fresh batch every step: host generation + length sort + H2D upload inside the timed region, overlapped)"),
            "config": {"workload": "mn-att-ques-im-hist + disc, B=%d dialogs/GPU x 10 rounds x 100 options, "
                                   "%s, V=11322, E=300, H=512 (BASELINE.json configs[%d])"
                                   % (args.batch, "14x14x512 pool5 map" if args.config == 3 else "7x7x2048 ResNet-200 map", args.config),
                       "global_batch_dialogs": world * args.batch, "parallelism": "dp%d" % world,
                       "dropout": "on (device generator)", "loss": round(float(loss), 5), "host": args.host,
                       "GPU_MAX_HW_QUEUES": os.environ.get('GPU_MAX_HW_QUEUES', 'default'),
                       "GPU_MAX_HW_QUEUES_choice": QUEUE_CHOICE, "startup_s": round(startup_s, 1),
                       "collective": collective,
                       "option_rows_executed_of_total": (list(model.option_rows()) if args.host == 'native' else None)},
            "roofline": roof,
        }
        if world == 1 and args.config == 3 and args.host == 'native':
            model.close()
        if world == 1 and args.config == 3 and args.recurrence != 'fp32' and args.host == 'native' and not args.no_alt:
            try:
                out["alt"] = alt_leg(args, N)
            except Exception as exc:
                out["alt"] = {"recurrence": "fp32", "error": str(exc)[:200]}
        if world == 1 and args.config == 3 and args.host == 'native' and not args.no_other_configs:
            # the other single-GPU configurations of BASELINE.json, driver-visible (the headline `value` above is unaffected:
            # they run after its timed region, on their own models)
            out["other_configs"] = []
            for cfg in (0, 1, 2, 4):
                try:
                    out["other_configs"].append(other_config(cfg))
                except Exception as exc:
                    out["other_configs"].append({"workload": "BASELINE.json configs[%d]" % cfg, "error": str(exc)[:200]})
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if lib_comm:
        from visdial_amd.parallel import destroy_library_comm
        if getattr(model, 'h', None):            # (a world-1 run under torch.distributed.run has closed its model before the extra legs)
            model.synchronize()
        destroy_library_comm()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
