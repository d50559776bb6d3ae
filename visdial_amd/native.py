"""`NativeModel` -- the training / retrieval step of any encoder x decoder plug-in pair driven entirely through the
MODEL-LEVEL C ABI (include/visdial_hip.h, csrc/runtime.hip): the same handful of calls a LuaJIT `model.lua` proxy
makes.  Python only converts numpy batches to host pointers; streams, the skewed LSTM wavefront,
the length sort, workspaces and launch order live in the library.  Same method names as visdial_amd.model.Model /
the reference's Model (model.lua): trainIteration, forwardBackward, update, retrieveBatch."""
import ctypes as C

import numpy as np

from . import _lib
from .split_eval import SplitEval

call = _lib.call


class _DevArray(object):
    """a raw device pointer as something torch.as_tensor understands (plumbing for torch.distributed only)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {'shape': (n,), 'typestr': '<f4', 'data': (ptr, False), 'version': 2}


class _Optims(object):
    """`model.optims` of the reference (model.lua:58-63): only learningRate is host-visible; the Adam moments live in
    the library"""

    def __init__(self, model):
        self._m = model

    def _lr(self, value=None):
        v = C.c_double(0.0 if value is None else float(value))
        call("vd_model_learning_rate", self._m.h, C.byref(v), 0 if value is None else 1)
        return float(v.value)

    def __getitem__(self, k):
        if k != 'learningRate':
            raise KeyError(k)
        return self._lr()

    def __setitem__(self, k, v):
        if k != 'learningRate':
            raise KeyError(k)
        self._lr(v)

    def keys(self):
        return ['learningRate']


class NativeModel(SplitEval):
    def __init__(self, params, init_seed=1234, dist_group=None, library_comm=False):
        """library_comm=True: the gradient all-reduce is the library's own (vd_comm_init has been called in this process,
        visdial_amd.parallel.init_library_comm): RCCL behind the ABI, no Python in the reduce path.  dist_group: the
        older host-side path over torch.distributed (kept as the gloo test vehicle and for the operator-level host)."""
        p = params
        self.dist_group = dist_group
        self.library_comm = bool(library_comm)
        self._dW = None
        mp = _lib.ModelParams(
            vocabSize=p['vocabSize'], embedSize=p['embedSize'], rnnHiddenSize=p['rnnHiddenSize'],
            imgFeatureSize=p.get('imgFeatureSize', 0), imgSpatialSize=p.get('imgSpatialSize', 1),
            commonEmbeddingSize=p.get('commonEmbeddingSize', 512), numAttentionLayers=int(p.get('numAttentionLayers', 1) or 1),
            maxQuesCount=p['maxQuesCount'], numOptions=p.get('numOptions', 100),
            learningRate=p.get('learningRate', 1e-3), lrDecayRate=p.get('lrDecayRate', 0.9997592083),
            minLRate=p.get('minLRate', 5e-5), seed=int(p.get('seed', 1234)) + 7919 * int(p.get('rank', 0)),
            lstmBf16={'fp32': 0, 'bf16': 1, 'split9': 9, 'split6': 6, 'split3': 3}[p.get('lstmPrecision', 'split9')], useStreams=int(p.get('useStreams', 1)),
            numLayers=int(p.get('numLayers', 2)), imgEmbedSize=int(p.get('imgEmbedSize', 300)),
            dropout=float(p.get('dropout', 0.5)))
        self.params = p
        h = C.c_void_p()
        call("vd_model_create", C.byref(mp), p['encoder'].encode(), p['decoder'].encode(), C.byref(h))
        self.h = h
        lib = _lib.load()
        self.tensors = []
        name = C.create_string_buffer(64)
        off, r, c = C.c_int64(), C.c_int64(), C.c_int64()
        for i in range(lib.vd_model_num_tensors(h)):
            call("vd_model_tensor_info", h, i, name, C.byref(off), C.byref(r), C.byref(c))
            self.tensors.append((name.value.decode(), int(off.value), int(r.value), int(c.value)))
        call("vd_model_init_params", h, int(init_seed))
        self._keep = None
        self._next_batch, self._next_src = None, None
        self.runningLoss = 0.0
        self._W = None
        self.optims = _Optims(self)

    def close(self):
        if self.h:
            _lib.load().vd_model_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ parameters
    def _shape(self, name, r, c):
        return (c,) if name.endswith('.b') else (r, c)

    def set_parameters_dict(self, d):
        for name, _, r, c in self.tensors:
            a = np.ascontiguousarray(d[name], dtype=np.float32)
            assert a.size == r * c, name
            call("vd_model_set_tensor", self.h, name.encode(), a.ctypes.data, a.size)

    def _get(self, which):
        out = {}
        for name, _, r, c in self.tensors:
            a = np.empty(self._shape(name, r, c), np.float32)
            call("vd_model_get_tensor", self.h, name.encode(), which, a.ctypes.data, a.size)
            out[name] = a
        return out

    def get_parameters_dict(self):
        return self._get(0)

    def get_gradients_dict(self):
        return self._get(1)

    @property
    def wrapperW(self):
        """the flat parameter vector (device tensor aliasing the library's buffer; same aligned layout as
        visdial_amd.model.Model.wrapperW, so `.pt` checkpoints are interchangeable between the two hosts)"""
        if self._W is None:
            import torch
            ptrs = [C.c_void_p() for _ in range(4)]
            call("vd_model_flat_pointers", self.h, *[C.byref(x) for x in ptrs])
            n = int(_lib.load().vd_model_flat_size(self.h))
            self._W = torch.as_tensor(_DevArray(ptrs[0].value, n), device='cuda')
        self.synchronize()
        return self._W

    def _entries(self):
        kind = lambda n: 'embed' if n == 'embed' else ('lstm' if n.split('.')[0].rstrip('0123456789') in
                                                        ('hist', 'ques', 'opt', 'dec', 'dialog') else 'lin') + \
            ('_w' if n.endswith('.W') else '_b')
        return [(n, self._shape(n, r, c), kind(n)) for n, _, r, c in self.tensors]

    def load_flat_parameters(self, modelW):
        """`model.wrapperW:copy(savedModel.modelW)` for a flat vector in the REFERENCE's getParameters() layout
        (see visdial_amd.model.Model.load_flat_parameters)"""
        from . import t7
        self.set_parameters_dict(t7.flat_to_named(np.asarray(modelW), self._entries(), self.params['encoder']))

    def flat_parameters(self):
        from . import t7
        return t7.named_to_flat(self.get_parameters_dict(), self._entries(), self.params['encoder'])

    def training(self, on=True):
        call("vd_model_set_training", self.h, int(on))

    def set_dropout_masks(self, masks):
        call("vd_model_set_dropout_mask", self.h, None, None, 0)
        for k, v in (masks or {}).items():
            a = np.ascontiguousarray(np.asarray(v) != 0, dtype=np.uint8).reshape(-1)
            call("vd_model_set_dropout_mask", self.h, k.encode(), a.ctypes.data, a.size)

    # ------------------------------------------------------------------ step
    def upload(self, batch):
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        keep = []                                   # host buffers stay alive until the call returns

        def ptr(a):
            keep.append(a)
            return a.ctypes.data
        q = i32(batch['ques_fwd'].reshape(-1, batch['ques_fwd'].shape[2]))
        hb = _lib.Batch(B=batch['ques_fwd'].shape[0], Tq=q.shape[1], ques_fwd=ptr(q))
        if 'hist' in batch:
            h = i32(batch['hist'].reshape(-1, batch['hist'].shape[2]))
            hb.Th, hb.hist = h.shape[1], ptr(h)
        if 'img_feat' in batch:
            hb.img_feat = ptr(np.ascontiguousarray(batch['img_feat'], dtype=np.float32))
        if 'options' in batch:
            o = i32(batch['options'])
            hb.To, hb.options = o.shape[-1], ptr(o)
        if 'answer_ind' in batch:
            hb.answer_ind = ptr(i32(batch['answer_ind'].reshape(-1)))
        if 'answer_in' in batch:
            a = i32(batch['answer_in'])
            hb.Ta, hb.answer_in, hb.answer_out = a.shape[-1], ptr(a), ptr(i32(batch['answer_out']))
        if 'option_in' in batch:
            a = i32(batch['option_in'])
            hb.To, hb.option_in, hb.option_out = a.shape[-1], ptr(a), ptr(i32(batch['option_out']))
        call("vd_model_upload_batch", self.h, C.byref(hb))      # host buffers are consumed before it returns
        self._N = batch['ques_fwd'].shape[0] * batch['ques_fwd'].shape[1]
        # non-pad target tokens of THIS batch: the gen loss EMA divides by it (model.lua:76-85)
        self._numTokens = float((np.asarray(batch['answer_out']) > 0).sum()) if 'answer_out' in batch else 0.0
        # whatever was prefetched for trainIteration has just been replaced (evaluate / retrieve / predict / generate
        # between two training steps): the next trainIteration must fetch its own batch
        self._keep = None

    def forwardBackward(self, batch=None, onlyForward=False, deferLoss=False):
        if batch is not None:
            self.upload(batch)
        call("vd_model_forward_backward", self.h, int(onlyForward))
        return self.loss if deferLoss else self.loss()

    def loss(self):
        v = C.c_float()
        call("vd_model_loss", self.h, C.byref(v))
        return float(v.value)

    def _dp_active(self):
        import os
        if self.dist_group is None:
            return False
        import torch.distributed as dist
        return dist.get_world_size(self.dist_group) > 1 or os.environ.get('VD_FORCE_ALLREDUCE') == '1'

    def update(self, gscale=1.0):
        """[all-reduce of the flat gradient over the group] -> clamp -> adam -> lr decay.  The collective is the
        host's: the library hands out the device pointer of wrapperdW, its main stream, the encoder's flat range and a
        "encoder gradients are final" wait.  Two buckets: the encoder's tensors start reducing on a
        communication stream as soon as the encoder backward has ended -- underneath the option-LSTM backward --
        the shared embedding and the decoder's tensors follow behind the step on the library's main stream."""
        if self.library_comm:
            import os
            from .parallel import library_comm_world
            world = library_comm_world()
            assert world > 0, "library_comm=True but no communicator: call visdial_amd.parallel.init_library_comm first"
            if world > 1 or os.environ.get('VD_FORCE_ALLREDUCE') == '1':
                call("vd_model_allreduce_grads", self.h)        # two buckets, library-owned communication stream
                gscale = 1.0 / world
        elif self._dp_active():
            import torch
            from .parallel import reduce_gradients
            if self._dW is None:
                ptrs = [C.c_void_p() for _ in range(4)]
                call("vd_model_flat_pointers", self.h, *[C.byref(x) for x in ptrs])
                n = int(_lib.load().vd_model_flat_size(self.h))
                self._dW = torch.as_tensor(_DevArray(ptrs[1].value, n), device='cuda')
                self._stream = torch.cuda.ExternalStream(_lib.load().vd_model_stream(self.h))
                self._comm = torch.cuda.Stream()
                lo, hi = C.c_int64(), C.c_int64()
                call("vd_model_encoder_range", self.h, C.byref(lo), C.byref(hi))
                self._enc_range = (int(lo.value), int(hi.value))
            lo, hi = self._enc_range
            work = None
            if hi > lo:
                call("vd_model_wait_encoder_grads", self.h, C.c_void_p(self._comm.cuda_stream))
                with torch.cuda.stream(self._comm):
                    _, work = reduce_gradients(self._dW[lo:hi], self.dist_group, async_op=True)
            with torch.cuda.stream(self._stream):          # ordered behind the step on the library's main stream
                for sl in (self._dW[:lo], self._dW[hi:]):
                    gscale, _ = reduce_gradients(sl, self.dist_group)
                if work is not None:
                    work.wait()                            # the main stream waits for bucket 1
                    self._stream.wait_stream(self._comm)
        call("vd_model_update", self.h, float(gscale))

    def trainIteration(self, dataloader):
        """model.lua:66-106, software-pipelined like Model.trainIteration: enqueue the step, upload the next batch
        (copy stream, second slot) while the device executes, then wait for this step's loss."""
        if self._keep is None or self._keep is not dataloader:
            # nothing usable in the library's slot: first call, another dataloader, or evaluate / retrieve / predict
            # replaced the prefetched batch.  The batch drawn for this step is still on the host: upload it again
            # (the dataloader's sample stream is not advanced twice).
            if self._next_batch is not None and self._next_src is dataloader:
                self.upload(self._next_batch)
            else:
                self.upload(dataloader.getTrainBatch(self.params))
        numTokens = self._numTokens                             # of the batch this step trains on
        call("vd_model_forward_backward", self.h, 0)
        self.update()
        self._next_batch, self._next_src = dataloader.getTrainBatch(self.params), dataloader
        self.upload(self._next_batch)                           # prefetch: the second slot, on the copy stream
        self._keep = dataloader                                 # (upload() clears it: set AFTER the prefetch)
        curLoss = self.loss()
        # model.lua:73-93: gen feeds curLoss / numTokens into the EMA (the criterion sums over tokens), disc curLoss
        cur = curLoss / max(numTokens, 1.0) if self.params['decoder'] == 'gen' else curLoss
        self.runningLoss = 0.95 * self.runningLoss + 0.05 * cur if self.runningLoss > 0 else cur
        return curLoss

    def scores(self, N, O):
        a = np.empty((N, O), np.float32)
        call("vd_model_scores", self.h, a.ctypes.data, a.size)
        return a

    def retrieveBatch(self, batch, useGt=None):
        """model.lua:344-430: ground-truth ranks [N] (useGt; default params['useGt']) or all ranks [N x O]"""
        if useGt is None:
            useGt = bool(self.params.get('useGt', True))
        self.upload(batch)
        call("vd_model_retrieve", self.h)
        N, O = self._N, int(self.params.get('numOptions', 100))
        out = np.empty(N if useGt else (N, O), np.int32)
        call("vd_model_ranks", self.h, int(useGt), out.ctypes.data)
        return out

    # Model:evaluate / retrieve / predict (model.lua:109-246): visdial_amd/split_eval.py, shared with the Python host
    def _set_training(self, on):
        self.training(on)

    # Model:generateAnswers (host loop in split_eval.py); the four device steps through the model-level ABI
    def _gen_encode(self, batch):
        self.upload({k: v for k, v in batch.items() if k in ('ques_fwd', 'hist', 'img_feat')})
        call("vd_model_encode", self.h)

    def _gen_begin(self, rounds):
        r = np.ascontiguousarray(rounds, dtype=np.int32)
        call("vd_model_decode_begin", self.h, r.ctypes.data, r.size)
        self._gen_n = r.size

    def _gen_step(self, tokens):
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        assert t.size == self._gen_n
        out = np.empty((t.size, int(self.params['vocabSize'])), np.float32)
        call("vd_model_decode_step", self.h, t.ctypes.data, out.ctypes.data)
        return out

    def _gen_select(self, src, n_keep):
        s_ = np.ascontiguousarray(src, dtype=np.int32)
        call("vd_model_decode_select", self.h, s_.ctypes.data, int(n_keep))

    def option_rows(self):
        """(rows the option LSTM executes, N * O candidates) of the current batch: the upload de-duplicates candidates"""
        a, b = C.c_int64(), C.c_int64()
        call("vd_model_option_rows", self.h, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def family_ms(self):
        a = (C.c_float * 3)()
        call("vd_model_family_ms", self.h, a)
        return [float(x) for x in a]

    def synchronize(self):
        call("vd_model_synchronize", self.h)
