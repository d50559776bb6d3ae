-- model.lua -- drop-in for the reference's model.lua (class Model, model.lua:8-615) on the MI355X-native library.
--
-- Same surface the reference's train.lua / evaluate.lua use: Model(params), model:trainIteration(dataloader),
-- model:forwardBackward(batch, onlyForward), model:retrieveBatch(batch), model:retrieve / model:predict(dataloader,
-- dtype), model.optims.learningRate, model:getFlatParameters() / model:setFlatParameters(t) (wrapperW:copy / :float()).
-- The encoder / decoder are still resolved by plug-in FILE (model.lua:19-26): encoders/<name>.lua returns a table
-- with model(params), decoders/<name>.lua a table with model(params, enc), forwardConnect, backwardConnect -- here
-- those tables describe the native graph instead of building nn modules, and the whole step (stream fork/join,
-- skewed LSTM wavefront, length sort, workspaces) runs behind the model-level C ABI (include/visdial_hip.h,
-- csrc/runtime.hip).  Host code stays Lua; no cutorch / cunn / rnn is needed, torch only for the dataloader's
-- CPU tensors.
-- EXECUTED by the tests: tests/luavm (a Lua 5.1 evaluator with a LuaJIT-style ffi and a Torch7 tensor stub) runs this file against the real
-- library on the GPU -- loss, every gradient tensor and the post-Adam parameters against the library's model-level path, the fp64 oracle and the
-- golden fixtures (tests/test_lua_host_gpu.py) -- and against a bounds-checking dry library on the CPU (tests/test_luavm_cpu.py).  The Python host visdial_amd/native.py makes the same calls.
local ffi = require 'ffi'
local vd = dofile('visdial_ffi.lua')
local C = vd.C
local utils = dofile('utils.lua')          -- the reference's own utils.lua (processRanks, writeJSON, ...)

local Model = torch.class('Model')

local function as_int(t)   return t:int():contiguous() end      -- Long/Float/Cuda tensor -> IntTensor (token ids)
local function as_float(t) return t:float():contiguous() end

function Model:__init(params)
    print('Setting up model (MI355X-native)..')
    self.params = params
    -- build the model - encoder, decoder (model.lua:19-26): the plug-in files name the native graph
    local encFile = string.format('encoders/%s.lua', params.encoder)
    local decFile = string.format('decoders/%s.lua', params.decoder)
    print('Encoder: ', params.encoder); print('Decoder: ', params.decoder)
    self.encoder = dofile(encFile).model(params)
    local decoderNet = dofile(decFile)
    self.decoder = decoderNet.model(params, self.encoder)
    self.forwardConnect = decoderNet.forwardConnect
    self.backwardConnect = decoderNet.backwardConnect

    local p = ffi.new('vd_model_params')
    p.vocabSize = params.vocabSize;           p.embedSize = params.embedSize
    p.rnnHiddenSize = params.rnnHiddenSize;   p.imgFeatureSize = params.imgFeatureSize or 0
    p.imgSpatialSize = params.imgSpatialSize or 14
    p.commonEmbeddingSize = params.commonEmbeddingSize or 512
    p.numAttentionLayers = params.numAttentionLayers or 1
    p.maxQuesCount = params.maxQuesCount;     p.numOptions = params.numOptions or 100
    p.learningRate = params.learningRate;     p.lrDecayRate = params.lrDecayRate
    p.minLRate = params.minLRate;             p.seed = 1234
    -- recurrence arithmetic (-lstmPrecision, new relative to the reference): split9 (default: option recurrence on the exact 3 x bf16 split,
    -- fp32-grade) | fp32 (v_mfma_f32) | bf16 (option
    -- recurrence on the compact bf16 state + bf16 operands in the encoder's recurrent products and dense weight gradients: BASELINE configs[4])
    p.lstmBf16 = ({fp32 = 0, bf16 = 1, split9 = 9, split6 = 6, split3 = 3})[params.lstmPrecision or 'split9']
    p.useStreams = 1
    p.numLayers = params.numLayers or 2;      p.imgEmbedSize = params.imgEmbedSize or 300
    p.dropout = params.dropout or 0.5
    -- device: VD_DEVICE wins (lets the UNCHANGED train.lua run with `-gpuid -1`, i.e. without its
    -- `require 'cutorch'` branch, train.lua:15-20, while the library still computes on a GPU), else -gpuid, else 0
    local dev = tonumber(os.getenv('VD_DEVICE') or '') or ((params.gpuid and params.gpuid >= 0) and params.gpuid or 0)
    vd.call('vd_set_device', dev)
    local h = ffi.new('vd_model*[1]')
    vd.call('vd_model_create', p, self.encoder.native, self.decoder.native, h)
    self.h = ffi.gc(h[0], C.vd_model_destroy)
    vd.call('vd_model_init_params', self.h, 1234)                      -- library-default init (weight-init.lua is a no-op)
    -- optimiser state lives in the library; the learning rate is mirrored for train.lua's log line / checkpoints
    self.optims = {learningRate = params.learningRate}
    self.havePrefetched = false
    self.numTokens = 0
    self.numOptions = p.numOptions                                     -- what the library ranks over (100 in the reference: model.lua:148,199)
end

-- `model.wrapperW` (train.lua:79,100,120; evaluate.lua:91; generate.lua:83) is a real host FloatTensor: checked out
-- of the library on access (Model.__index at the end of this file) and written back before the next device call, so
-- `model.wrapperW:copy(savedModel.modelW)`, `torch.save({modelW = model.wrapperW, ...})` and `model.wrapperW:float()`
-- all work on the reference's scripts as they are.
function Model:commitW()
    local w = rawget(self, 'checkedOutW')
    if w then self:setFlatParameters(w); rawset(self, 'checkedOutW', nil) end
end

-- Data parallelism (new relative to the reference, which is single-GPU: train.lua:19): one `th` process per GPU.  Rank 0
-- calls Model.commUniqueId() and hands the 128-byte string to its peers by any channel the launcher has (a file, a
-- socket, an environment variable written by a wrapper script); every rank then calls model:initComm(rank, world, id).
-- From then on trainIteration sums the gradient over RCCL inside the library (two buckets, the encoder's under the
-- option-LSTM backward) and applies the 1/world average before clamp + adam -- the update of the concatenated batch.
function Model.commUniqueId()
    local id = ffi.new('char[128]')
    vd.call('vd_comm_unique_id', id)
    return ffi.string(id, 128)
end

-- no collective, no device: can this process load RCCL?  Exchange the answer BEFORE commUniqueId / initComm, so that a rank
-- without a usable librccl cannot strand its peers inside the collective ncclCommInitRank.  -> ok, NCCL_VERSION_CODE
function Model.commAvailable()
    local v = ffi.new('int[1]')
    local rc = C.vd_comm_available(v)
    return rc == 0, tonumber(v[0])
end

-- the last gradient all-reduce of this process: floats in the early (encoder) bucket and in the late one, whether the early one
-- was issued under the decoder's backward, calls since initComm -- what a first multi-GPU run prints to be diagnosable
function Model.commStats()
    local b1, b2, ov, n = ffi.new('int64_t[1]'), ffi.new('int64_t[1]'), ffi.new('int[1]'), ffi.new('int64_t[1]')
    vd.call('vd_comm_stats', b1, b2, ov, n)
    return {bucket1 = tonumber(b1[0]), bucket2 = tonumber(b2[0]), overlapped = ov[0] ~= 0, calls = tonumber(n[0])}
end

function Model:initComm(rank, world, id)
    assert(#id == 128, 'initComm: the rendezvous token is 128 bytes')
    vd.call('vd_comm_init', rank, world, ffi.cast('const void*', id))
    self.world = world
end

-- batch tables of dataloader.lua:324-339,378-475 -> vd_batch (host pointers; consumed before the call returns)
function Model:upload(batch)
    local keep = {}                                                    -- converted tensors stay alive until the call returns
    local function ints(t)   local x = as_int(t);   keep[#keep + 1] = x; return x end
    local function floats(t) local x = as_float(t); keep[#keep + 1] = x; return x end
    local b = ffi.new('vd_batch')
    local ques = ints(batch['ques_fwd'])
    b.B = ques:size(1); b.Tq = ques:size(3); b.ques_fwd = ques:data()
    if batch['hist'] then local x = ints(batch['hist']); b.Th = x:size(3); b.hist = x:data() end
    if batch['img_feat'] then b.img_feat = floats(batch['img_feat']):data() end
    if batch['options'] then local x = ints(batch['options']); b.To = x:size(x:dim()); b.options = x:data() end
    if batch['answer_ind'] then b.answer_ind = ints(batch['answer_ind']):data() end
    if batch['answer_in'] then                                         -- decoder gen, training (dataloader.lua:330-334)
        local x = ints(batch['answer_in']); b.Ta = x:size(x:dim()); b.answer_in = x:data()
        b.answer_out = ints(batch['answer_out']):data()
    end
    if batch['option_in'] then                                         -- decoder gen, retrieval (dataloader.lua:437-462)
        local x = ints(batch['option_in']); b.To = x:size(x:dim()); b.option_in = x:data()
        b.option_out = ints(batch['option_out']):data()
    end
    vd.call('vd_model_upload_batch', self.h, b)
    -- non-pad target tokens of THIS batch: the gen loss EMA divides by it (model.lua:76-85)
    self.numTokens = batch['answer_out'] and batch['answer_out']:gt(0):sum() or 0
end

function Model:loss()
    local v = ffi.new('float[1]')
    vd.call('vd_model_loss', self.h, v)
    return tonumber(v[0])
end

-- model.lua:66-106, software-pipelined: enqueue the step, upload the NEXT batch while the device runs, read the loss
function Model:trainIteration(dataloader)
    self:commitW()
    if not self.havePrefetched then
        -- first call, or evaluate / retrieve / predict replaced the prefetched batch in the library's slot: the batch
        -- drawn for this step is still on the host (the dataloader's sample stream is not advanced twice)
        self:upload(self.nextBatch or dataloader:getTrainBatch(self.params)); self.havePrefetched = true
    end
    local numTokens = self.numTokens                                   -- of the batch this step trains on
    vd.call('vd_model_forward_backward', self.h, 0)
    local lr = ffi.new('double[1]', self.optims.learningRate)
    vd.call('vd_model_learning_rate', self.h, lr, 1)
    local world = self.world or 1
    if world > 1 then vd.call('vd_model_allreduce_grads', self.h) end  -- RCCL sum over the ranks (enqueue only)
    vd.call('vd_model_update', self.h, 1.0 / world)                    -- [average] clamp(-5,5) + adam + lr decay (model.lua:96-105)
    vd.call('vd_model_learning_rate', self.h, lr, 0)
    self.optims.learningRate = tonumber(lr[0])
    self.nextBatch = dataloader:getTrainBatch(self.params)
    self:upload(self.nextBatch)                                        -- prefetch into the second slot (copy stream)
    local curLoss = self:loss()
    -- model.lua:73-93: the EMA lives in the GLOBAL `runningLoss` that train.lua:89 initialises and train.lua:113 prints;
    -- gen feeds curLoss / numTokens (the criterion sums over tokens), disc curLoss
    local cur = curLoss
    if self.params.decoder == 'gen' then cur = curLoss / math.max(numTokens, 1) end
    if (runningLoss or 0) > 0 then runningLoss = 0.95 * runningLoss + 0.05 * cur
    else runningLoss = cur end
    return curLoss
end

-- model.lua:249-342 (both decoder branches; forwardConnect / backwardConnect run inside the library)
function Model:forwardBackward(batch, onlyForward)
    self:commitW()
    self:upload(batch); self.havePrefetched = false
    vd.call('vd_model_forward_backward', self.h, onlyForward and 1 or 0)
    return self:loss()
end

-- model.lua:344-430 + utils.computeRanks (utils.lua:106-128)
function Model:retrieveBatch(batch)
    self:commitW()
    self:upload(batch); self.havePrefetched = false
    vd.call('vd_model_retrieve', self.h)                               -- disc: option scores; gen: candidate log-likelihoods
    local N = batch['ques_fwd']:size(1) * batch['ques_fwd']:size(2)
    local O = self.numOptions
    local useGt = self.params.useGt and 1 or 0
    local out = torch.IntTensor(useGt == 1 and N or N * O)
    vd.call('vd_model_ranks', self.h, useGt, out:data())
    if useGt == 1 then return out:double():view(-1, self.params.maxQuesCount) end
    return out:double():view(N, O)
end

-- Model:evaluate (model.lua:109-139): validation loss / perplexity over a split (train.lua:105-107)
function Model:evaluate(dataloader, dtype)
    self:setMode(false)
    local total = dataloader.numThreads[dtype]
    local curLoss, count, first = 0, 0, 1
    while first <= total do
        local batch, nxt = dataloader:getTestBatch(first, self.params, dtype)
        if batch['answer_out'] then                                   -- model.lua:124-127, both decoders (dataloader.lua:398-421)
            count = count + batch['answer_out']:gt(0):sum()            -- non-pad target tokens
            curLoss = curLoss + self:forwardBackward(batch, true)      -- gen: summed token NLL; disc: the batch's mean cross-entropy
        else                                                           -- synthetic disc batches without answers: mean over rounds
            local rounds = batch['answer_ind']:nElement()
            count = count + rounds
            curLoss = curLoss + self:forwardBackward(batch, true) * rounds
        end
        first = nxt
    end
    curLoss = curLoss / math.max(count, 1)
    print(string.format('\n%s\tLoss: %f\t Perplexity: %f\n', dtype, curLoss, math.exp(curLoss)))
    self:setMode(true)
    return curLoss
end

function Model:setMode(training)
    vd.call('vd_model_set_training', self.h, training and 1 or 0)
end

-- rank every dialog of a split batch by batch; perRound = number of rank values kept per round (1 or numOptions)
function Model:rankSplit(dataloader, dtype, useGt)
    self:setMode(false)
    self.params.useGt = useGt
    self.params.numOptions = self.numOptions                           -- model.lua:148,199 write the constant 100 here
    local O, R = self.numOptions, self.params.maxQuesCount
    local total = dataloader.numThreads[dtype]
    local ranks = useGt and torch.Tensor(total, R) or torch.Tensor(total, R, O)
    ranks:fill(O + 1)                                                  -- rounds never ranked keep rank 101
    local first = 1
    while first <= total do
        local batch, nxt = dataloader:getTestBatch(first, self.params, dtype)
        local got = self:retrieveBatch(batch)
        if useGt then ranks:narrow(1, first, nxt - first):copy(got)
        else ranks:narrow(1, first, nxt - first):copy(got:view(nxt - first, R, O)) end
        first = nxt
    end
    self:setMode(true)
    return ranks
end

-- {image_id, round_id, ranks} records: real image ids, only the rounds a dialog has; lastOnly = test split of
-- predict (one record per dialog, its last round) -- the EvalAI format
local function records(dataloader, dtype, ranks, lastOnly)
    local ids, rounds = dataloader['unique_img_' .. dtype], dataloader[dtype .. '_num_rounds']
    local tab, out = torch.totable(ranks:double()), {}
    for i = 1, #ids do
        local from = lastOnly and rounds[i] or 1
        for j = from, rounds[i] do
            out[#out + 1] = {image_id = ids[i], round_id = j, ranks = tab[i][j]}
        end
    end
    return out
end

-- Model:retrieve (model.lua:142-189): ground-truth ranks + R@k / MRR
function Model:retrieve(dataloader, dtype)
    local ranks = self:rankSplit(dataloader, dtype, true)
    print(string.format('\n%s - Retrieval:', dtype))
    utils.processRanks(ranks)
    return records(dataloader, dtype, ranks, false)
end

-- Model:predict (model.lua:192-246): all 100 ranks per round; the test split keeps the last round only
function Model:predict(dataloader, dtype)
    return records(dataloader, dtype, self:rankSplit(dataloader, dtype, false), dtype == 'test')
end

-- wrapperW:float() / wrapperW:copy(savedModel.modelW) (train.lua:79,99-102,120-121; evaluate.lua:91): the flat
-- vector in THIS library's layout (embed | encoder tensors | decoder tensors, no padding), tensor by tensor
-- Model:generateAnswers (model.lua:432-613): beam search (default) or temperature sampling with the generative decoder,
-- one dialog at a time.  Candidate bookkeeping is host control flow as in the reference; the device side is four calls:
-- vd_model_encode (encoder forward), vd_model_decode_begin (hiddenBeams), vd_model_decode_step (one decoder step for all
-- live hypotheses -> log-probabilities on the host), vd_model_decode_select (beam back-pointers).
function Model:generateAnswers(dataloader, dtype, params)
    if self.params.decoder == 'disc' then error('Sampling/beam search only for generative model') end
    params = params or {}
    local sampleWords = params.sampleWords == 1
    local temperature = params.temperature or 1.0
    local beamSize, beamLen = params.beamSize or 5, params.beamLen or 20
    local startToken, endToken = dataloader.word2ind['<START>'], dataloader.word2ind['<END>']
    local numThreads = params.maxThreads or dataloader.numThreads[dtype]
    local V = self.params.vocabSize
    local answerTable = {}
    self:commitW()
    self:setMode(false)
    for convId = 1, numThreads do
        local batch = dataloader:getIndexData(torch.LongTensor{convId}, self.params, dtype)
        local R = batch['ques_fwd']:size(2)
        self:upload({ques_fwd = batch['ques_fwd'], hist = batch['hist'], img_feat = batch['img_feat']})
        self.havePrefetched = false
        vd.call('vd_model_encode', self.h)                             -- forwardBackward(batch, true, true)
        local threadAnswers = {}
        local function words(ids) return utils.idToWords(ids, dataloader.ind2word) end
        if not sampleWords then
            local n = beamSize
            local rounds, toks, src = ffi.new('int32_t[?]', n), ffi.new('int32_t[?]', n), ffi.new('int32_t[?]', n)
            local logp = ffi.new('float[?]', n * V)
            for iter = 1, R do
                for i = 0, n - 1 do rounds[i] = iter - 1 end
                vd.call('vd_model_decode_begin', self.h, rounds, n)    -- hiddenBeams (model.lua:478-503)
                local beams, scores, finish = {}, {}, {}
                for i = 1, n do beams[i] = {startToken}; scores[i] = 0 end
                for step = 2, beamLen do
                    local explore = (step == 2) and 1 or n             -- all beams are <START> at first
                    for i = 1, n do toks[i - 1] = beams[i][step - 1] or 0 end
                    vd.call('vd_model_decode_step', self.h, toks, logp)
                    local cands = {}
                    for w = 1, explore do
                        local row = {}
                        for c = 1, V do row[c] = {c, logp[(w - 1) * V + c - 1]} end
                        table.sort(row, function(a, b) if a[2] ~= b[2] then return a[2] > b[2] end return a[1] < b[1] end)
                        for k = 1, math.min(n, V) do
                            local cid, lp = row[k][1], row[k][2]
                            local cb = {}
                            for t = 1, step - 1 do cb[t] = beams[w][t] end
                            cb[step] = cid
                            if cid == endToken then finish[#finish + 1] = {beam = cb, score = scores[w] + lp, order = #finish}
                            else cands[#cands + 1] = {beam = cb, score = scores[w] + lp, src = w - 1, order = #cands} end
                        end
                    end
                    table.sort(cands, function(a, b) if a.score ~= b.score then return a.score > b.score end return a.order < b.order end)
                    local keep = math.min(n, #cands)
                    if keep > 0 then
                        for i = 1, keep do src[i - 1] = cands[i].src end
                        vd.call('vd_model_decode_select', self.h, src, keep)   -- untouched slots keep their old state
                    end
                    for i = 1, keep do beams[i] = cands[i].beam; scores[i] = cands[i].score end
                end
                table.sort(finish, function(a, b) if a.score ~= b.score then return a.score > b.score end return a.order < b.order end)
                local best = (#finish > 0) and finish[1].beam or beams[1]
                threadAnswers[#threadAnswers + 1] = {question = words(batch['ques_fwd'][{1, iter}]), answer = words(torch.LongTensor(best))}
            end
        else
            local n = R
            local rounds, toks, src = ffi.new('int32_t[?]', n), ffi.new('int32_t[?]', n), ffi.new('int32_t[?]', n)
            local logp = ffi.new('float[?]', n * V)
            for i = 0, n - 1 do rounds[i] = i; src[i] = i; toks[i] = startToken end
            vd.call('vd_model_decode_begin', self.h, rounds, n)
            local answer = {}
            for i = 1, n do answer[i] = {startToken} end
            for _ = 1, beamLen do
                vd.call('vd_model_decode_step', self.h, toks, logp)
                vd.call('vd_model_decode_select', self.h, src, n)
                for i = 1, n do
                    local pr = torch.FloatTensor(V)
                    for c = 1, V do pr[c] = math.exp(logp[(i - 1) * V + c - 1] / temperature) end
                    local nxt = torch.multinomial(pr:div(pr:sum()), 1)[1]
                    answer[i][#answer[i] + 1] = nxt; toks[i - 1] = nxt
                end
            end
            for iter = 1, R do
                threadAnswers[#threadAnswers + 1] = {question = words(batch['ques_fwd'][{1, iter}]), answer = words(torch.LongTensor(answer[iter]))}
            end
        end
        local ids = dataloader['unique_img_' .. dtype]
        answerTable[#answerTable + 1] = {image_id = ids and ids[convId] or convId, dialog = threadAnswers}
    end
    self:setMode(true)
    return answerTable
end

-- The library's tensors in the order of the REFERENCE's `wrapper:getParameters()` flat vector, so that `modelW` of a checkpoint
-- written by the reference's train.lua loads tensor for tensor (and one written through this host loads in the reference and in
-- the Python hosts).  The table is the same one as visdial_amd/t7.py:_STEMS -- obtained by EXECUTING the reference's encoder
-- files and reading where getParameters() put every tensor (tests/golden/reference_param_order.json; tests/test_luavm_cpu.py
-- keeps the two tables equal).  'x*' = x1 .. xN (LSTM layers); for the four nngraph encoders the order is nngraph's forward-node
-- order (DERIVED, not verified on a Torch7-written file): mn-att reaches the hop-L img_common first, the (ques_common, att)
-- pairs follow in hop order.
local STEMS = {
    ['lf-ques'] = {'embed', 'ques*', 'fuse'},
    ['lf-ques-im'] = {'embed', 'ques*', 'fuse'},
    ['lf-ques-hist'] = {'embed', 'ques*', 'hist*', 'fuse'},
    ['lf-ques-im-hist'] = {'embed', 'ques*', 'hist*', 'fuse'},
    ['lf-att-ques-im-hist'] = {'img_proj', 'img_common', 'embed', 'ques*', 'hist*', 'qh', 'ques_common', 'att', 'out'},
    ['hre-ques-hist'] = {'embed', 'ques*', 'hist*', 'dialog'},
    ['hre-ques-im-hist'] = {'embed', 'img_embed', 'hist*', 'ques*', 'dialog'},
    ['hrea-ques-im-hist'] = {'embed', 'img_embed', 'hist*', 'ques*', 'att_q', 'att_h', 'dialog'},
    ['mn-ques-hist'] = {'embed', 'ques*', 'hist*', 'mn1', 'mn2'},
    ['mn-ques-im-hist'] = {'embed', 'ques*', 'qi', 'hist*', 'mn1', 'mn2'},
    ['mn-att-ques-im-hist'] = {'img_proj', 'img_common#rev', 'embed', 'ques*', 'hist*', 'mn1', 'mn2', 'ques_common+att#', 'out'},
}

-- Escape hatch: `params.paramOrder` = a Lua table of tensor names in flat order (e.g. read off `th lua/dump_param_order.lua <ckpt>`
-- under a real Torch7) replaces the table above; 'declaration' keeps the library's own declaration order.
function Model:tensors()
    local declared, name = {}, ffi.new('char[64]')
    local off, rows, cols = ffi.new('int64_t[1]'), ffi.new('int64_t[1]'), ffi.new('int64_t[1]')
    local byStem, stems = {}, {}
    for i = 0, tonumber(C.vd_model_num_tensors(self.h)) - 1 do
        vd.call('vd_model_tensor_info', self.h, i, name, off, rows, cols)
        local t = {name = ffi.string(name), numel = tonumber(rows[0] * cols[0])}
        local stem = string.match(t.name, '^(.-)%.[Wb]$') or t.name
        if not byStem[stem] then byStem[stem] = {}; stems[#stems + 1] = stem end
        table.insert(byStem[stem], t)
        declared[#declared + 1] = t
    end
    local function layered(pre)                 -- pre1 .. preN that the model declares, by N
        local out = {}
        for _, s in ipairs(stems) do
            local n = string.match(s, '^' .. pre .. '(%d+)$')
            if n then out[#out + 1] = {tonumber(n), s} end
        end
        table.sort(out, function(a, b) return a[1] < b[1] end)
        for i, e in ipairs(out) do out[i] = e[2] end
        return out
    end
    local function hops(pre)                    -- hop 1 has no suffix, hop i > 1 is <name><i>
        local out = layered(pre)
        table.insert(out, 1, pre)
        return out
    end
    local po = self.params.paramOrder
    if po == 'declaration' then return declared end
    if type(po) == 'table' then
        local byName, out = {}, {}
        for _, t in ipairs(declared) do byName[t.name] = t end
        for _, n in ipairs(po) do
            out[#out + 1] = assert(byName[n], 'Model:tensors: paramOrder names ' .. tostring(n) .. ' twice or the library declares no such tensor')
            byName[n] = nil
        end
        assert(#out == #declared, 'Model:tensors: paramOrder must name every tensor exactly once')
        return out
    end
    local want = {}
    for _, t in ipairs(assert(STEMS[self.params.encoder], 'Model:tensors: unknown encoder ' .. tostring(self.params.encoder))) do
        if string.sub(t, -1) == '*' then
            for _, s in ipairs(layered(string.sub(t, 1, -2))) do want[#want + 1] = s end
        elseif t == 'img_common#rev' then
            local h = hops('img_common')
            for i = #h, 1, -1 do want[#want + 1] = h[i] end
        elseif t == 'ques_common+att#' then
            local q, a = hops('ques_common'), hops('att')
            for i = 1, #q do want[#want + 1] = q[i]; want[#want + 1] = a[i] end
        else
            want[#want + 1] = t
        end
    end
    local out, placed = {}, {}
    for _, s in ipairs(want) do
        for _, t in ipairs(assert(byStem[s], 'Model:tensors: the library declares no tensor ' .. s)) do out[#out + 1] = t end
        placed[s] = true
    end
    for _, s in ipairs(stems) do                -- the decoder, in declaration order (decoders/disc.lua, decoders/gen.lua)
        if not placed[s] then for _, t in ipairs(byStem[s]) do out[#out + 1] = t end end
    end
    assert(#out == #declared)
    return out
end

-- pin the noise of the Dropout nodes for the NEXT steps (parity runs against a CPU restatement): masks = {site = ByteTensor keep-mask},
-- sites as in vd_model_set_dropout_mask (q_emb, h_emb, hatt, img_tr, iqc, u, fuse, img); nil clears every pin
function Model:setDropoutMasks(masks)
    vd.call('vd_model_set_dropout_mask', self.h, nil, nil, 0)
    for site, keep in pairs(masks or {}) do
        local k = keep:byte():contiguous()
        vd.call('vd_model_set_dropout_mask', self.h, site, k:data(), k:nElement())
    end
end

function Model:getFlatParameters()
    local ts, total = self:tensors(), 0
    for _, t in ipairs(ts) do total = total + t.numel end
    local flat, o = torch.FloatTensor(total), 0
    for _, t in ipairs(ts) do
        vd.call('vd_model_get_tensor', self.h, t.name, 0, flat:narrow(1, o + 1, t.numel):data(), t.numel)
        o = o + t.numel
    end
    return flat
end

function Model:setFlatParameters(flat)
    flat = flat:float():contiguous()
    local o = 0
    for _, t in ipairs(self:tensors()) do
        vd.call('vd_model_set_tensor', self.h, t.name, flat:narrow(1, o + 1, t.numel):data(), t.numel)
        o = o + t.numel
    end
end

-- `model.wrapperW`: every access checks the flat vector out of the library into a host FloatTensor (a fresh D2H
-- snapshot, 57 MB for the headline pair) and remembers it; commitW() -- first thing in every method that touches the
-- device -- writes it back, so both directions of the reference's uses work on the unchanged scripts:
--   model.wrapperW:copy(savedModel.modelW)                    train.lua:79, evaluate.lua:91, generate.lua:83
--   torch.save(path, {modelW = model.wrapperW, ...})          train.lua:100   (a plain FloatTensor is serialised)
--   model.wrapperW:float()                                    train.lua:120
-- (a read-only access costs one redundant H2D of identical values before the next step.)
local methods = Model.__index
Model.__index = function(self, key)
    if key == 'wrapperW' then
        local w = rawget(self, 'checkedOutW')
        if not w then w = self:getFlatParameters(); rawset(self, 'checkedOutW', w) end
        return w
    end
    if type(methods) == 'function' then return methods(self, key) end
    return methods[key]
end

return Model
