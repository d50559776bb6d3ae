"""CPU-only checks (-m "not gpu"): the C-ABI library loads and exports every symbol the header declares,
host logic (flags, batch layout, metrics), and the data-parallel reduction with gloo, world_size 2."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, small_params
from oracle import visdial_oracle as vo


def test_library_exports_every_declared_symbol():
    """no compute calls here (no GPU): load the .so, compare its symbols with include/visdial_hip.h"""
    from visdial_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    header = open(os.path.join(ROOT, 'include', 'visdial_hip.h')).read()
    declared = set(re.findall(r'\b(vd_[a-z0-9_]+)\s*\(', header))
    assert len(declared) >= 40
    for name in sorted(declared):
        assert hasattr(lib, name), "library does not export %s" % name
    assert declared == set(_lib.PROTOTYPES), (declared ^ set(_lib.PROTOTYPES))
    assert lib.vd_abi_version() == 2 == _lib.ABI_VERSION
    assert isinstance(lib.vd_last_error(), bytes)


def test_product_path_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'visdial_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src.replace('the oracle', '').replace('# oracle', ''), (dirpath, f)


def test_model_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from visdial_amd.model import Model
    from visdial_amd.opts import derive
    with pytest.raises(RuntimeError):
        Model(derive(small_params()))


def test_opts_defaults_and_derived_flags():
    from visdial_amd import opts
    o = opts.default_params()
    assert o['batchSize'] == 40 and o['learningRate'] == 1e-3 and o['lrDecayRate'] == 0.9997592083      # opts.lua:33-38
    assert o['encoder'] == 'lf-ques-hist' and o['decoder'] == 'gen' and o['rnnHiddenSize'] == 512
    assert o['useHistory'] and not o['useIm'] and o['concatHistory']                                    # opts.lua:54-59
    a = opts.default_params(encoder='mn-att-ques-im-hist')
    assert a['useIm'] and a['useHistory'] and not a['concatHistory']
    assert a['imgNorm'] == 0 and a['inputImg'] == 'data/data_img_pool5.h5'                              # opts.lua:61-67
    p = opts.parse(['-encoder', 'hre-ques-im-hist', '-decoder', 'disc', '-batchSize', '20'])
    assert p['encoder'] == 'hre-ques-im-hist' and p['batchSize'] == 20 and 'hre-ques-im-hist-disc' in p['savePath']


def test_plugin_registry_knows_every_reference_file():
    from visdial_amd import encoders, decoders
    ref = ['hre-ques-hist', 'hre-ques-im-hist', 'hrea-ques-im-hist', 'lf-att-ques-im-hist', 'lf-ques-hist',
           'lf-ques-im-hist', 'lf-ques-im', 'lf-ques', 'mn-att-ques-im-hist', 'mn-ques-hist', 'mn-ques-im-hist']
    assert sorted(encoders.NAMES) == sorted(ref) and decoders.NAMES == ['disc', 'gen']
    for n in ('lf-ques', 'lf-ques-im-hist', 'hre-ques-im-hist', 'mn-att-ques-im-hist'):
        m = encoders.load(n)
        assert hasattr(m, 'model') and hasattr(m, 'declare')
    with pytest.raises(ValueError):
        encoders.load('no-such-encoder')
    for n in ('disc', 'gen'):
        m = decoders.load(n)
        assert all(hasattr(m, f) for f in ('model', 'forwardConnect', 'backwardConnect'))
    assert hasattr(decoders.load('gen'), 'decoderConnect') and not hasattr(decoders.load('disc'), 'decoderConnect')


def test_synthetic_batch_layout_matches_dataloader_contract():
    from visdial_amd.dataloader import SyntheticDataloader
    from visdial_amd.opts import derive
    p = derive(small_params(batchSize=3, maxQuesCount=5, numOptions=9))
    b = SyntheticDataloader(p, seed=0).getTrainBatch(p)
    q, h, o = b['ques_fwd'], b['hist'], b['options']
    assert q.shape[:2] == (3, 5) and h.shape[:2] == (3, 5) and o.shape[:2] == (15, 9)
    for seq in q.reshape(-1, q.shape[2]):                       # right-aligned: zeros only on the left
        nz = np.nonzero(seq)[0]
        assert len(nz) and (seq[nz[0]:] != 0).all()
    assert (q.reshape(-1, q.shape[2])[:, 0] != 0).any()         # trimmed to the batch maximum
    for seq in o.reshape(-1, o.shape[2]):                       # options left-aligned, trailing zeros
        nz = np.nonzero(seq)[0]
        assert len(nz) and (seq[:nz[-1] + 1] != 0).all()
    assert b['answer_ind'].min() >= 1 and b['answer_ind'].max() <= 9            # 1-based (prepro.py:169)
    assert b['img_feat'].shape == (3, 3, 3, 16) and (b['img_feat'] >= 0).all()
    g = derive(small_params(encoder='lf-ques-im-hist', decoder='gen', imgNorm=1))
    bg = SyntheticDataloader(g, seed=0).getTrainBatch(g)
    ai, ao = bg['answer_in'], bg['answer_out']
    assert ai.shape == ao.shape and (ai[:, :, 0] == g['vocabSize'] - 1).all()   # <START> first
    assert ((ai != 0) == (ao != 0)).all()                                         # same pad pattern
    np.testing.assert_allclose(np.linalg.norm(bg['img_feat'], axis=1), 1.0, rtol=1e-5)   # imgNorm (dataloader.lua:64-68)


def test_process_ranks_matches_oracle():
    from visdial_amd import utils
    rng = np.random.RandomState(0)
    ranks = rng.randint(1, 101, size=(50, 10)).astype(np.float64)
    a, b = utils.processRanks(ranks, verbose=False), vo.process_ranks(ranks)
    for k in b:
        assert a[k] == b[k]


def test_seqsort_metadata_and_non_pad_row_lists():
    """host-side length sort of a right-aligned [T x N] token matrix (nn.SeqSort; the native runtime builds the same
    arrays in upload_tokens): permutations are inverse of each other, step t owns the prefix [0, nact[t]), and the row
    lists the masked weight gradients contract (vd_gemm_tn_rows_acc) are exactly the non-pad (t, row) pairs"""
    import torch
    from visdial_amd.nn import SeqSort
    rng = np.random.RandomState(0)
    T, N = 7, 13
    lens = rng.randint(0, T + 1, size=N)
    lens[3], lens[5] = 0, T                                   # an empty row and a full-length row
    tok = np.zeros((T, N), np.int32)
    for n, l in enumerate(lens):
        tok[T - l:, n] = rng.randint(1, 50, size=l)
    s = SeqSort(tok, torch.device('cpu'))
    perm, inv = s.perm.numpy(), s.inv.numpy()
    assert np.array_equal(perm[inv], np.arange(N)) and np.all(np.diff(lens[perm]) <= 0)
    ts = s.tok_sorted.numpy()
    assert np.array_equal(ts, tok[:, perm])
    for t in range(T):
        assert np.all(ts[t, :s.nact[t]] != 0) and np.all(ts[t, s.nact[t]:] == 0)
    assert np.array_equal(ts.reshape(-1)[s.inv_idx.numpy()].reshape(T, N), tok)      # sorted -> original layout
    assert np.array_equal(tok.reshape(-1)[s.fwd_idx.numpy()].reshape(T, N), ts)
    act, act1, prev1 = (a.numpy() for a in s.rows)
    assert np.array_equal(act, np.flatnonzero(ts.reshape(-1) != 0))                  # all non-pad pairs, in step order
    assert np.array_equal(act1, act[act >= N]) and np.array_equal(prev1, act1 - N)
    assert act.size == lens.sum()


def test_shard_dialogs_partition():
    from visdial_amd.parallel import shard_dialogs
    for n, w in ((20, 8), (160, 8), (7, 3), (5, 1)):
        spans = [shard_dialogs(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


DP_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import numpy as np, torch, torch.distributed as dist
from conftest import small_params
from oracle import visdial_oracle as vo
from visdial_amd.dataloader import SyntheticDataloader
from visdial_amd.opts import derive
from visdial_amd.parallel import reduce_gradients, shard_dialogs
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%(port)d', rank=int(sys.argv[1]), world_size=2)
rank = dist.get_rank()
p = derive(small_params(batchSize=4))
full = SyntheticDataloader(p, seed=123).getTrainBatch(p)          # the same global batch on both ranks
lo, hi = shard_dialogs(4, rank, 2)
R, O = p['maxQuesCount'], p['numOptions']
mine = {'ques_fwd': full['ques_fwd'][lo:hi], 'hist': full['hist'][lo:hi], 'img_feat': full['img_feat'][lo:hi],
        'options': full['options'][lo * R:hi * R], 'answer_ind': full['answer_ind'][lo * R:hi * R]}
P = vo.init_params(p['encoder'], p['decoder'], p, seed=1)
spec = vo.param_spec(p['encoder'], p['decoder'], p)
r = vo.forward_backward(p['encoder'], p['decoder'], P, p, mine, None)
flat = torch.from_numpy(vo.flatten(r['grads'], spec).copy())
gscale, _ = reduce_gradients(flat)                                  # gloo all-reduce (RCCL on the GPU box)
w_dp, _ = vo.clamp_adam(vo.flatten(P, spec), flat.numpy() * gscale, {}, 1e-3)
if rank == 0:
    big = vo.forward_backward(p['encoder'], p['decoder'], P, p, full, None)
    w_big, _ = vo.clamp_adam(vo.flatten(P, spec), vo.flatten(big['grads'], spec), {}, 1e-3)
    np.testing.assert_allclose(flat.numpy() * gscale, vo.flatten(big['grads'], spec), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(w_dp, w_big, rtol=0, atol=1e-12)
    print("DP_OK")
dist.destroy_process_group()
'''


def test_data_parallel_step_equals_big_batch_step_gloo(tmp_path):
    """2 ranks x 2 dialogs (gloo) == 1 rank x 4 dialogs: averaged gradient and post-Adam weights."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER % dict(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "DP_OK" in outs[0]


COMM_WORKER = r'''
import ctypes as C, os, sys
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from visdial_amd import _lib
from visdial_amd.parallel import join_library_comm, library_comm_available, library_comm_stats, library_comm_world
rank = int(sys.argv[1])
if rank == 1 and sys.argv[2] == 'broken':
    os.environ['VD_RCCL_LIB'] = '/nonexistent/librccl.so.1'     # ... and no other candidate either:
    import visdial_amd.parallel as par
    par.library_comm_available = lambda: (False, 0, 'cannot load librccl.so.1 (test)')
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%(port)d', rank=rank, world_size=2)
lib = _lib.load()
# argument / ordering errors, the same on both ranks and never a collective:
tok = C.create_string_buffer(128)
assert lib.vd_comm_init(2, 2, tok) != 0 and b'bad rank 2 / world 2' in lib.vd_last_error()          # rank outside the world
assert lib.vd_comm_init(-1, 2, tok) != 0 and lib.vd_comm_init(0, 0, tok) != 0 and lib.vd_comm_init(0, 2, None) != 0
assert lib.vd_model_allreduce_grads(None) != 0                                                       # reduce before init
assert library_comm_world() == 0 and library_comm_stats()['calls'] == 0
assert lib.vd_comm_destroy() == 0                                                                    # destroy before init: no-op
# the agreed three-phase join: here (no GPU) phase 3 fails on every rank -- or phase 1 on the 'broken' rank -- and BOTH ranks
# come back with the same answer instead of one of them hanging in a collective
ok, report = join_library_comm(dist.group.WORLD)
assert ok is False and library_comm_world() == 0, report
if sys.argv[2] == 'broken':
    assert any('no rank joins' in r for r in report), report                   # stopped after phase 1, before any token / init
    assert not any('vd_comm_init' in r for r in report), report
else:
    assert any('vd_comm_init failed' in r for r in report) and any('abandoned on every rank' in r for r in report), report
assert lib.vd_comm_destroy() == 0                                             # nothing half-built is left behind ...
assert lib.vd_comm_init(3, 2, tok) != 0 and b'bad rank' in lib.vd_last_error()   # ... and the entry points still answer
dist.barrier()
print("COMM_OK", rank, report)
dist.destroy_process_group()
'''


@pytest.mark.parametrize("mode", ['healthy', 'broken'])
def test_library_comm_world2_argument_and_ordering_errors_gloo(tmp_path, mode):
    """csrc/comm.hip at world 2 without hardware: argument and ordering errors of vd_comm_* / vd_model_allreduce_grads, and
    parallel.join_library_comm's agreed phases (RCCL loadable? -> token -> init) -- both ranks always leave with the SAME
    outcome and a clean state, also when one rank cannot load RCCL at all ('broken')."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "comm_worker.py"
    script.write_text(COMM_WORKER % dict(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), mode], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("COMM_OK" in o for o in outs), outs


def test_library_comm_surface_without_a_gpu():
    """csrc/comm.hip: the communicator entry points exist, report "no communicator" (world 0) before vd_comm_init, refuse
    to reduce without one -- and the library carries no link-time dependency on RCCL (it is dlopen'ed on first use, so a
    single-GPU Lua host never loads it)."""
    import ctypes as C
    import subprocess
    from visdial_amd import _lib
    from visdial_amd.parallel import library_comm_world
    assert library_comm_world() == 0
    lib = _lib.load()
    assert lib.vd_model_allreduce_grads(None) != 0 and b'null model' in lib.vd_last_error()
    assert lib.vd_comm_init(3, 2, C.create_string_buffer(128)) != 0          # rank outside the world
    assert lib.vd_comm_destroy() == 0                                        # nothing to destroy is not an error
    needed = subprocess.run(['readelf', '-d', _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert 'rccl' not in needed.lower() and 'torch' not in needed.lower()


def test_graft_entry_build_runs():
    """the driver's "does it build" check: make is a no-op when the library is current, the checker builds, every declared symbol loads and
    the ABI version matches the binding's (a hard-coded version here once outlived a bump)"""
    import __graft_entry__
    __graft_entry__.build()
