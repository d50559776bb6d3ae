"""`Model` -- the step driver, counterpart of the reference's model.lua (class Model, :8-615).

Same public surface: Model(params), :trainIteration(dataloader), :forwardBackward(batch,
onlyForward, encOutOnly), :retrieve / :predict / :retrieveBatch, fields wrapperW / wrapperdW /
optims.  Encoder and decoder are resolved by plug-in file name (model.lua:19-26).  Every tensor
op is a launch into libvisdial_hip.so; this file only prepares inputs and orders the launches.
"""
import math

import numpy as np
import torch

import os

from . import decoders, encoders, ops, utils
from .nn import DropoutState, SeqSort, StreamPool, Workspace
from .params import FlatParams, ParamSpec, init_host
from .split_eval import SplitEval


class Wrapper(object):
    """Stands in for `nn.Sequential():add(enc):add(dec)` (model.lua:42): mode switches,
    zeroGradParameters, getParameters."""

    def __init__(self, model):
        self._m = model

    def training(self):
        self._m.drop.training = True

    def evaluate(self):
        self._m.drop.training = False

    def zeroGradParameters(self):
        ops.zero(self._m.fp.dW)

    def getParameters(self):
        return self._m.fp.W, self._m.fp.dW

    def get(self, i):
        return self._m.encoder if i == 1 else self._m.decoder


class Model(SplitEval):
    def __init__(self, params, dist_group=None):
        self.params = params
        if not torch.cuda.is_available():
            raise RuntimeError("visdial_amd.Model needs an MI355X (HIP device); there is no CPU path")
        gpuid = int(params.get('gpuid', 0))
        self.device = torch.device('cuda', gpuid)
        torch.cuda.set_device(self.device)
        self.dist_group = dist_group           # torch.distributed process group (RCCL) or None
        self.world = 1
        if dist_group is not None:
            import torch.distributed as dist
            self.world = dist.get_world_size(dist_group)

        # build the model - encoder, decoder (model.lua:19-26)
        self.encFile = encoders.load(params['encoder'])
        self.decFile = decoders.load(params['decoder'])
        spec = ParamSpec()
        spec.embed('embed', params['vocabSize'] + 1, params['embedSize'])
        self.encFile.declare(params, spec)
        self.decFile.declare(params, spec)
        self.fp = FlatParams(spec, self.device)
        self.fp.load_host(init_host(spec, params['rnnHiddenSize'], int(params.get('seed', 1234))))
        self.ws = Workspace(self.device)
        self.drop = DropoutState(self.ws, seed=int(params.get('seed', 1234)) + 7919 * int(params.get('rank', 0)))
        self.streams = StreamPool(self.device, enabled=True)
        self.encoder = self.encFile.model(params, self.fp, self.ws, self.drop, self.streams)
        self.decoder = self.decFile.model(params, self.encoder, self.fp, self.ws, self.drop)
        self.decoder.streams = self.streams
        # decoder hooks (model.lua:28-29)
        self.forwardConnect = self.decFile.forwardConnect
        self.backwardConnect = self.decFile.backwardConnect
        self.decoderConnect = getattr(self.decFile, 'decoderConnect', None)
        self.wrapper = Wrapper(self)
        self.wrapperW, self.wrapperdW = self.wrapper.getParameters()     # model.lua:55
        # criterion (model.lua:32-38): disc = CrossEntropyCriterion; gen = SequencerCriterion(MaskZeroCriterion(ClassNLL)),
        # fused into the generative decoder's vocabulary head (decoders/gen.py)
        from .criterion import CrossEntropyCriterion
        self.criterion = CrossEntropyCriterion(self.ws) if params['decoder'] == 'disc' else None
        self.wrapper.training()                                          # model.lua:57
        # optimiser state (model.lua:60-61; optim_updates.lua:68-77)
        self.optims = {'learningRate': float(params['learningRate']), 't': 0}
        self.adam_m = torch.zeros_like(self.wrapperW)
        self.adam_v = torch.zeros_like(self.wrapperW)
        self.runningLoss = 0.0
        self._mask_cache = {}
        # flat-vector slice holding the encoder's own tensors (everything between the shared embedding and
        # the decoder tensors): the first gradient bucket of the data-parallel all-reduce
        names = [e[0] for e in spec.entries]
        dec_first = None
        tmp = ParamSpec()
        self.decFile.declare(params, tmp)
        dec_first = tmp.entries[0][0]
        self._enc_slice = (self.fp.offsets[names[1]][0], self.fp.offsets[dec_first][0])
        self._enc_bucket_work = None
        # input pipeline: the next batch is fetched and uploaded on its own stream while the device still executes
        # the current step (Model.trainIteration); the loss leaves the device through a pinned buffer
        self._copy_stream = torch.cuda.Stream(device=self.device)
        self._next = None
        self._keepalive = None

    # ------------------------------------------------------------------ helpers
    def _dev(self, a, dtype):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(self.device, non_blocking=False)

    def _causal_mask(self, B):
        """model.lua:280-294: mask[i][j] = 0 iff j <= i, tiled over the batch ([N x R] bytes)."""
        R = int(self.params['maxQuesCount'])
        m = self._mask_cache.get(B)
        if m is None:
            base = (np.arange(R)[None, :] > np.arange(R)[:, None]).astype(np.uint8)
            m = self._dev(np.tile(base, (B, 1)), np.uint8)
            self._mask_cache[B] = m
        return m

    def prepare_inputs(self, batch):
        """Input re-layout of model.lua:255-294 (+ the :cuda() copies of dataloader.lua:410-475):
        time-major token matrices, per-image feature map, causal mask."""
        p = self.params
        inputs = []
        q = batch['ques_fwd']
        B = q.shape[0]
        sort_rows = (p['encoder'].startswith('mn') or
                                                                      p['encoder'].startswith('lf-att'))

        def tokens(a):
            th = np.ascontiguousarray(a.reshape(-1, a.shape[2]).T, dtype=np.int32)     # [T x N] time-major
            t = self._dev(th, np.int32)
            if sort_rows:
                t.vd_sort = SeqSort(th, self.device)    # host-side length sort: the recurrences skip pad rows
            return t
        inputs.append(tokens(q))                                                     # [Tq x N]
        if p.get('useIm'):
            f = batch['img_feat']
            if 'att' in p['encoder']:
                f = f.reshape(-1, f.shape[-1])                                        # [B*S2 x C]
            inputs.append(self._dev(f, np.float32))
        if p.get('useHistory'):
            inputs.append(tokens(batch['hist']))                                     # [Th x N]
        if 'mn' in p['encoder']:
            inputs.append(self._causal_mask(B))
        dec_in = {}
        if 'answer_ind' in batch:
            dec_in['gt'] = self._dev(np.asarray(batch['answer_ind']).reshape(-1) - 1, np.int32)       # 0-based
        if p['decoder'] == 'disc':
            o = batch['options']
            rows = np.ascontiguousarray(o.reshape(-1, o.shape[2]), dtype=np.int32)   # [N*O x To]
            uid = None
            if rows.shape[0] > 1:
                # encode every DISTINCT candidate once (decoders/disc.lua:4-15: the encoding depends on the tokens only);
                # same rule as the native runtime (csrc/runtime.hip: vd_model_upload_batch)
                # (rows compared as opaque byte strings: np.unique(axis=0) sorts lexicographically over To columns and took 31 ms on
                #  the 32 000 x 20 headline batch, longer than the device step; the 1-D void view takes a quarter of that)
                key = rows.view(np.dtype((np.void, rows.dtype.itemsize * rows.shape[1]))).reshape(-1)
                _, first, inv = np.unique(key, return_index=True, return_inverse=True)
                if first.shape[0] <= 0.95 * rows.shape[0]:
                    uid, total, rows = self._dev(inv.reshape(-1), np.int32), rows.shape[0], rows[first]
            dec_in['options'] = self._dev(rows.T, np.int32)                          # [To x rows]
            if uid is not None:
                dec_in['options'].vd_uid, dec_in['options'].vd_total = uid, total
        else:
            for k in ('answer_in', 'answer_out'):
                if k in batch:
                    a = batch[k]
                    dec_in[k] = self._dev(a.reshape(-1, a.shape[2]).T, np.int32)
            for k in ('option_in', 'option_out'):                                     # retrieval (model.lua:393-399)
                if k in batch:
                    a = batch[k]
                    dec_in[k] = self._dev(a.reshape(-1, a.shape[3]).T, np.int32)       # [T x N*O], row = n*O + o
        return inputs, dec_in

    # ------------------------------------------------------------------ training
    def _fetch(self, dataloader):
        """getTrainBatch (model.lua:71) + the input re-layout / upload of model.lua:255-294, on the copy stream."""
        batch = dataloader.getTrainBatch(self.params)
        with torch.cuda.stream(self._copy_stream):
            prepared = self.prepare_inputs(batch)
            ready = torch.cuda.Event()
            ready.record()
        return batch, prepared, ready

    def trainIteration(self, dataloader):
        """model.lua:66-106.  Same order of effects as the reference (zero grads, batch, forward/backward, loss
        EMA, clamp, adam, lr decay); the host side is software-pipelined: the whole step is enqueued first, then
        the NEXT batch is fetched and uploaded on the copy stream while the device executes, and only then does
        the host wait for this step's loss."""
        self.wrapper.zeroGradParameters()
        if self._next is None or self._next[3] is not dataloader:
            self._next = self._fetch(dataloader) + (dataloader,)
        batch, prepared, ready, _ = self._next
        self._next = None
        torch.cuda.current_stream().wait_event(ready)
        pending = self.forwardBackward(batch, prepared=prepared, deferLoss=True)
        self.update()
        self._next = self._fetch(dataloader) + (dataloader,)       # software pipeline: prefetch the next batch
        curLoss = pending()
        # The loss leaves the device right after the criterion, i.e. BEFORE this step's backward has finished reading
        # the uploaded inputs: keep them referenced until the NEXT step's loss has arrived (stream order then
        # guarantees this step is complete), so the allocator cannot hand their memory to a new upload early.
        self._keepalive = (batch, prepared)
        if self.params['decoder'] == 'gen':
            numTokens = float((batch['answer_out'] > 0).sum())
            cur = curLoss / numTokens
        else:
            cur = curLoss
        self.runningLoss = 0.95 * self.runningLoss + 0.05 * cur if self.runningLoss > 0 else cur
        return curLoss

    def update(self):
        """[all-reduce] -> clamp(-5, 5) -> adam -> lr decay (model.lua:96-105)."""
        gscale = 1.0
        if self._dp_active():
            from .parallel import reduce_gradients
            if self._enc_bucket_work is not None:
                # bucket 1 (encoder tensors) was launched when the encoder backward finished and has been
                # running under the option-LSTM backward; only embedding + decoder slices remain exposed
                lo, hi = self._enc_slice
                for sl in (self.wrapperdW[:lo], self.wrapperdW[hi:]):
                    gscale, _ = reduce_gradients(sl, self.dist_group)
                self._enc_bucket_work.wait()
                self._enc_bucket_work = None
            else:
                gscale, _ = reduce_gradients(self.wrapperdW, self.dist_group)   # RCCL sum over xGMI
        o = self.optims
        o['t'] += 1
        t = o['t']
        step = o['learningRate'] * math.sqrt(1.0 - 0.999 ** t) / (1.0 - 0.9 ** t)
        ops.clamp_adam(self.wrapperW, self.wrapperdW, self.adam_m, self.adam_v, step, gscale=gscale, clip=5.0)
        if o['learningRate'] > self.params.get('minLRate', 5e-5):
            o['learningRate'] *= self.params.get('lrDecayRate', 0.9997592083)
        self.drop.next_step()

    def _dp_active(self):
        return self.dist_group is not None and (self.world > 1 or os.environ.get('VD_FORCE_ALLREDUCE') == '1')

    def forwardBackward(self, batch, onlyForward=False, encOutOnly=False, prepared=None, deferLoss=False):
        """model.lua:249-342.  Returns curLoss (python float); with deferLoss a callable that waits for the
        device and returns it (everything is enqueued when forwardBackward returns)."""
        inputs, dec_in = prepared if prepared is not None else self.prepare_inputs(batch)
        # LookupTableMaskZero zeroes the pad row on every forward
        ops.zero(self.fp.w['embed'][0])
        if self.params['decoder'] == 'disc' and not encOutOnly:
            pending = self._forwardBackward_disc(inputs, dec_in, onlyForward)
            return pending if deferLoss else pending()
        encOut = self.encoder.forward(inputs)
        self.forwardConnect(self.encoder, self.decoder, encOut, inputs[0].shape[0])
        if encOutOnly:
            return encOut
        loss = self.decoder.forward_backward_gen(self, inputs, dec_in, encOut, onlyForward)
        return (lambda: loss) if deferLoss else loss

    def _forwardBackward_disc(self, inputs, dec_in, onlyForward):
        """disc branch of model.lua:326-338, call for call:
            decOut  = decoder:forward({options, encOut})
            curLoss = criterion:forward(decOut, answerInd)
            gradCriterionOut = criterion:backward(decOut, answerInd)
            t = decoder:backward({options, encOut}, gradCriterionOut);  encoder:backward(inputs, t[2])
        The encoder (latency-bound chains) runs on a side stream concurrently with the option LSTM (throughput-bound)
        in both directions.  Returns a callable that waits for the device and yields curLoss."""
        st = self.streams
        # Enqueue order matters: the option LSTM is a handful of big launches, the encoder ~200 small
        # ones -- the big chain goes to the main stream first so the GPU is busy while the host is
        # still enqueueing the encoder on the side streams.
        enc_out_buf = self.encoder.output_buffer(inputs)
        d_in = (dec_in['options'], enc_out_buf)
        start = torch.cuda.Event()
        start.record()
        decOut = self.decoder.forward(d_in)                                       # model.lua:329
        with st.fork('enc', after=start):
            encOut = self.encoder.forward(inputs)                                 # model.lua:297
        assert encOut.data_ptr() == enc_out_buf.data_ptr()
        self.forwardConnect(self.encoder, self.decoder, encOut, inputs[0].shape[0])
        st.join('enc')
        crit = self.criterion.forward(decOut, dec_in['gt'], needGrad=not onlyForward)   # model.lua:330
        pending = crit.loss_handle()
        if not onlyForward:
            gradCriterionOut = self.criterion.backward(decOut, dec_in['gt'])      # model.lua:334
            self._ce_event = crit.done
            t = self.decoder.backward(d_in, gradCriterionOut)                     # model.lua:335 (main stream)
            with st.fork('enc', after=self._ce_done()):
                self.encoder.backward(inputs, t[1])                               # model.lua:337 (t[2] in Lua)
                if self._dp_active() and st.enabled:
                    # gradient bucket 1: the encoder's own tensors are final here -> start their RCCL
                    # all-reduce now, overlapped with the option-LSTM backward on the main stream
                    from .parallel import reduce_gradients
                    lo, hi = self._enc_slice
                    _, self._enc_bucket_work = reduce_gradients(self.wrapperdW[lo:hi], self.dist_group, async_op=True)
            st.join('enc')
            self.decoder.backward_embed()
        return pending

    def _ce_done(self):
        return self._ce_event

    # ------------------------------------------------------------------ retrieval (model.lua:142-246,344-430)
    def retrieveBatch(self, batch):
        inputs, dec_in = self.prepare_inputs(batch)
        ops.zero(self.fp.w['embed'][0])
        encOut = self.encoder.forward(inputs)
        if self.params['decoder'] == 'gen':
            scores = self.decoder.retrieve_lhood(self, dec_in['option_in'], dec_in['option_out'], encOut,
                                                 inputs[0].shape[0])
            self.scores = scores
        else:
            scores = self.decoder.forward((dec_in['options'], encOut)).materialize()
            self.scores = scores
        gt = dec_in.get('gt') if self.params.get('useGt') else None
        return utils.computeRanks(scores, gt, self.ws)

    # Model:evaluate / retrieve / predict (model.lua:109-246): visdial_amd/split_eval.py, shared with the native host
    def _set_training(self, on):
        self.wrapper.training() if on else self.wrapper.evaluate()

    # ------------------------------------------------------------------ generation (model.lua:432-613)
    # host loop: split_eval.SplitEval.generateAnswers; the four device steps over the operator-level entry points:
    def _gen_encode(self, batch):
        inputs, _ = self.prepare_inputs(batch)
        ops.zero(self.fp.w['embed'][0])
        encOut = self.encoder.forward(inputs)                                   # forwardBackward(batch, true, true)
        self._gen = dict(encOut=encOut, encLayers=getattr(self.encoder, 'rnnLayers', None), Tq=batch['ques_fwd'].shape[2])

    def _gen_begin(self, rounds):
        """hiddenBeams of model.lua:478-503: hypothesis i starts from the encoder state of QA round rounds[i]"""
        g, H, L = self._gen, self.params['rnnHiddenSize'], len(self.decoder.rnnLayers)
        idx = torch.tensor(np.asarray(rounds, np.int32), device=self.device)
        encOut, encLayers, Tq = g['encOut'], g['encLayers'], g['Tq']
        hid = []
        for level in range(L):
            if encLayers is not None:
                h = encOut if level == len(encLayers) - 1 else encLayers[level].output[Tq - 1]
                c = encLayers[level].cell[Tq - 1]
                hid.append((ops.embed_gather(h, idx, torch.empty(len(idx), H, device=self.device)),
                            ops.embed_gather(c, idx, torch.empty(len(idx), H, device=self.device))))
            else:
                z = torch.zeros(len(idx), H, device=self.device)
                h = ops.embed_gather(encOut, idx, torch.empty(len(idx), H, device=self.device)) if level == L - 1 else z
                hid.append((h, z.clone()))
        g['hidden'] = hid

    def _gen_step(self, tokens):
        g = self._gen
        tok = torch.tensor(np.asarray(tokens, np.int32)[None, :], device=self.device)
        logp, g['stepped'] = self.decoder.step_logprobs(tok, g['hidden'])
        return logp

    def _gen_select(self, src, n_keep):
        g, H = self._gen, self.params['rnnHiddenSize']
        idx = torch.tensor(np.asarray(src, np.int32)[:n_keep], device=self.device)
        hidden = []
        for (h_old, c_old), (h_new, c_new) in zip(g['hidden'], g['stepped']):
            h, c = h_old.clone(), c_old.clone()                                   # slots >= n_keep keep their old state
            h[:n_keep] = ops.embed_gather(h_new, idx, torch.empty(n_keep, H, device=self.device))
            c[:n_keep] = ops.embed_gather(c_new, idx, torch.empty(n_keep, H, device=self.device))
            hidden.append((h, c))
        g['hidden'] = hidden

    # ------------------------------------------------------------------ test / checkpoint helpers
    def get_parameters_dict(self):
        return self.fp.to_host('w')

    def get_gradients_dict(self):
        return self.fp.to_host('g')

    def set_parameters_dict(self, d):
        self.fp.load_host(d)

    def load_flat_parameters(self, modelW):
        """`model.wrapperW:copy(savedModel.modelW)` (train.lua:79, evaluate.lua:91) for a flat vector in the
        reference's getParameters() layout (no alignment padding, Torch7 module order) -- e.g. `modelW` of a .t7
        checkpoint."""
        from . import t7
        self.set_parameters_dict(t7.flat_to_named(np.asarray(modelW), self.fp.spec.entries, self.params['encoder']))

    def flat_parameters(self):
        """the parameters as the reference's flat `modelW` (Torch7 module order for this encoder)"""
        from . import t7
        return t7.named_to_flat(self.get_parameters_dict(), self.fp.spec.entries, self.params['encoder'])

    def set_dropout_masks(self, masks):
        """Pin nn.Dropout noise (dict name -> uint8 numpy array) for parity runs; None = generator."""
        if masks is None:
            self.drop.external = None
        else:
            self.drop.external = {k: self._dev(v.reshape(-1), np.uint8) for k, v in masks.items()}
