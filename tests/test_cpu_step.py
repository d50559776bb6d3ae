"""oracle/cpu_step.cpp (the C++17/OpenMP fp32 restatement, third restatement of the headline step and the CPU
baseline of bench.py) against the numpy oracle: the two were written independently from the reference's Lua files
(per-timestep GEMMs + 10x image replication vs hoisted/vectorised numpy), so agreement pins both."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import small_params
from oracle import cpu_step
from oracle import visdial_oracle as vo
from visdial_amd.dataloader import SyntheticDataloader
from visdial_amd.opts import derive

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("M,N,K,ta,tb", [(37, 50, 19, False, False), (130, 70, 300, False, True), (64, 257, 1000, True, False),
                                         (1, 5, 3, False, False), (300, 2048, 513, True, False)])
def test_gemm_kernel_matches_numpy(M, N, K, ta, tb):
    rng = np.random.RandomState(M + N + K)
    A = rng.randn(*((K, M) if ta else (M, K))).astype(np.float32)
    B = rng.randn(*((N, K) if tb else (K, N))).astype(np.float32)
    ref = (A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)
    assert relerr(cpu_step.gemm(A, B, ta, tb), ref) < 1e-6


@pytest.mark.parametrize("isa", ["scalar", "avx2"])
def test_gemm_other_kernels_in_subprocess(isa):
    code = ("import numpy as np, sys; sys.path.insert(0, %r); from oracle import cpu_step as c; "
            "r = np.random.RandomState(0); A = r.randn(70, 90).astype('f4'); B = r.randn(90, 45).astype('f4'); "
            "e = np.abs(c.gemm(A, B) - A.astype('f8') @ B.astype('f8')).max(); print(c.gemm_kernel()); assert e < 1e-4, e") % ROOT
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VDCPU_ISA=isa), capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def _case(batchSize, dropout_on, seed):
    p = derive(small_params(encoder='mn-att-ques-im-hist', decoder='disc', batchSize=batchSize))
    batch = SyntheticDataloader(p, seed=seed).getTrainBatch(p)
    P = vo.init_params(p['encoder'], p['decoder'], p, seed=seed + 1, dtype=np.float32)
    drop = None
    if dropout_on:
        B, R, Tq = batch['ques_fwd'].shape
        N, H, E = B * R, p['rnnHiddenSize'], p['embedSize']
        Th, S2, K = batch['hist'].shape[2], p['imgSpatialSize'] ** 2, p['commonEmbeddingSize']
        rng = np.random.RandomState(seed + 2)
        shp = dict(q_emb=(Tq, N, E), h_emb=(Th, N, E), hatt=(N, H), img_tr=(N, S2, H), iqc=(N, S2, K), u=(N, H))
        drop = {k: (rng.rand(*s) > 0.5).astype(np.float64) for k, s in shp.items()}
    return p, batch, P, drop


@pytest.mark.parametrize("dropout_on", [False, True])
def test_cpu_step_matches_numpy_oracle(dropout_on):
    p, batch, P, drop = _case(2, dropout_on, 5)
    spec = vo.param_spec(p['encoder'], p['decoder'], p)
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    ref = vo.forward_backward(p['encoder'], p['decoder'], P64, p, batch, drop)
    cs = cpu_step.CpuStep(p, spec, P)
    loss, scores = cs.step(batch, drop, want_scores=True)
    assert abs(loss - ref['loss']) < 1e-5
    assert relerr(scores, ref['scores']) < 1e-5
    G = cs.named(cs.G)
    for k in ref['grads']:
        err = relerr(G[k], ref['grads'][k])
        assert err < 2e-5 or np.abs(G[k] - ref['grads'][k]).max() < 1e-7, (k, err)


def test_cpu_step_update_matches_numpy_oracle():
    """two full trainIteration()s (clamp + adam with the eps-on-uncorrected-sqrt(v) quirk)"""
    p, batch, P, drop = _case(1, True, 9)
    spec = vo.param_spec(p['encoder'], p['decoder'], p)
    cs = cpu_step.CpuStep(p, spec, P)
    Pn = {k: v.astype(np.float64) for k, v in P.items()}
    st = {}
    for _ in range(2):
        cs.step(batch, drop, update=True, lr=1e-3)
        Pn, _ = vo.train_iteration(p['encoder'], p['decoder'], Pn, p, batch, drop, st, 1e-3)
    Wn = cs.named(cs.W)
    for k in Pn:
        if k == 'embed':
            continue   # row 0 is re-zeroed lazily (next forward) in both; compare the rest
        assert np.abs(Wn[k] - Pn[k]).max() < 2e-5, k
    assert np.abs(Wn['embed'][1:] - Pn['embed'][1:]).max() < 2e-5
