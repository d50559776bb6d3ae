"""ctypes loader for oracle/cpu_step.cpp (the C++17/OpenMP fp32 restatement of the mn-att-ques-im-hist + disc
training step).  TEST INFRASTRUCTURE / CPU BASELINE ONLY -- never imported by visdial_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, '_build', 'libvd_cpu_step.so')
SPEC_ENCODER, SPEC_DECODER = 'mn-att-ques-im-hist', 'disc'


class Dims(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ('B', 'R', 'Tq', 'Th', 'To', 'O', 'V', 'E', 'H', 'S2', 'C', 'K')]


_lib = None


def build():
    subprocess.check_call(['make', '-C', HERE, '-s'])


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        lib = C.CDLL(LIB)
        lib.vdcpu_train_step.restype = C.c_double
        lib.vdcpu_train_step.argtypes = [C.POINTER(Dims)] + [C.c_void_p] * 13 + [C.c_void_p, C.c_int, C.c_float,
                                                                                  C.c_void_p, C.c_void_p, C.c_int]
        lib.vdcpu_num_params.restype = C.c_int64
        lib.vdcpu_num_params.argtypes = [C.POINTER(Dims)]
        lib.vdcpu_num_threads.restype = C.c_int
        lib.vdcpu_set_num_threads.restype = None
        lib.vdcpu_set_num_threads.argtypes = [C.c_int]
        lib.vdcpu_gemm_kernel.restype = C.c_char_p
        lib.vdcpu_gemm.restype = None
        lib.vdcpu_gemm.argtypes = [C.c_int] * 3 + [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                                   C.c_void_p, C.c_int64, C.c_int]
        _lib = lib
        q = cpu_quota()
        if q:       # never run more OpenMP threads than twice the container's CPU quota (they would only be throttled)
            lib.vdcpu_set_num_threads(max(1, min(int(lib.vdcpu_num_threads()), int(round(2 * q)))))
    return _lib


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited.  The GPU
    boxes expose 256 logical CPUs but run the container under a 16-CPU quota: 128 OpenMP threads are throttled to a
    third of the rate 32 threads reach (scripts/cpu_probe.py: 570 vs 1 650 GFLOP/s on the 20 000 x 512 x 2 048 GEMM)."""
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        return None if q == 'max' else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def num_threads():
    return int(load().vdcpu_num_threads())


def fit_threads_to_quota():
    """OpenMP threads = 2 x the CPU quota (the best of the 8..256 sweep), never more than the default.  Returns the
    thread count in use."""
    lib = load()
    q = cpu_quota()
    if q:
        want = max(1, min(num_threads(), int(round(2 * q))))
        lib.vdcpu_set_num_threads(want)
    return num_threads()


def gemm_kernel():
    return load().vdcpu_gemm_kernel().decode()


def dims_of(p, batch):
    B, R, Tq = batch['ques_fwd'].shape
    N, O, To = batch['options'].shape
    return Dims(B=B, R=R, Tq=Tq, Th=batch['hist'].shape[2], To=To, O=O, V=p['vocabSize'], E=p['embedSize'],
                H=p['rnnHiddenSize'], S2=p['imgSpatialSize'] ** 2, C=p['imgFeatureSize'],
                K=p.get('commonEmbeddingSize', 512))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class CpuStep(object):
    """Holds the flat fp32 parameter / gradient / Adam vectors (layout = visdial_oracle.param_spec order)."""

    def __init__(self, p, spec, params):
        assert int(p.get('numAttentionLayers', 1) or 1) == 1, "cpu_step.cpp restates the default single attention hop"
        self.p, self.spec = p, spec
        self.W = np.ascontiguousarray(np.concatenate([np.asarray(params[e[0]], np.float32).reshape(-1) for e in spec]))
        self.G = np.zeros_like(self.W)
        self.m = np.zeros_like(self.W)
        self.v = np.zeros_like(self.W)
        self.t = 0

    def named(self, vec):
        out, o = {}, 0
        for e in self.spec:
            k = int(np.prod(e[1]))
            out[e[0]] = vec[o:o + k].reshape(e[1])
            o += k
        return out

    def step(self, batch, drop=None, update=False, lr=1e-3, want_scores=False):
        """drop: dict of keep-masks keyed like the numpy oracle (q_emb, h_emb, hatt, img_tr, iqc, u) or None.
        Returns (loss, scores or None); self.G holds the gradients (clamped if update)."""
        lib = load()
        d = dims_of(self.p, batch)
        assert lib.vdcpu_num_params(C.byref(d)) == self.W.size, "parameter layout mismatch"
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        ques = i32(batch['ques_fwd'].reshape(-1, d.Tq))
        hist = i32(batch['hist'].reshape(-1, d.Th))
        opts = i32(batch['options'])
        ans = i32(batch['answer_ind'].reshape(-1))
        img = np.ascontiguousarray(batch['img_feat'], dtype=np.float32)
        mk = lambda k: None if drop is None else np.ascontiguousarray(np.asarray(drop[k]) != 0, dtype=np.uint8)
        masks = [mk(k) for k in ('q_emb', 'h_emb', 'hatt', 'img_tr', 'iqc', 'u')]
        scores = np.zeros((d.B * d.R, d.O), np.float32) if want_scores else None
        if update:
            self.t += 1
        loss = lib.vdcpu_train_step(C.byref(d), _ptr(self.W), _ptr(self.G), _ptr(ques), _ptr(hist), _ptr(img), _ptr(opts),
                                    _ptr(ans), *[_ptr(m) for m in masks], _ptr(scores), int(update), float(lr),
                                    _ptr(self.m), _ptr(self.v), int(self.t))
        return float(loss), scores


def gemm(A, B, transa=False, transb=False):
    """C = op(A) @ op(B) through the library's kernel (unit tests)."""
    lib = load()
    A = np.ascontiguousarray(A, np.float32)
    B = np.ascontiguousarray(B, np.float32)
    M, K = (A.shape[1], A.shape[0]) if transa else A.shape
    N = B.shape[0] if transb else B.shape[1]
    Cm = np.zeros((M, N), np.float32)
    rsa, csa = (1, A.shape[1]) if transa else (A.shape[1], 1)
    rsb, csb = (1, B.shape[1]) if transb else (B.shape[1], 1)
    lib.vdcpu_gemm(M, N, K, _ptr(A), rsa, csa, _ptr(B), rsb, csb, _ptr(Cm), N, 0)
    return Cm
