-- model_ops.lua -- class ModelOps: the reference's Model (model.lua:8-106, 249-342) over plug-in files that are composed IN LUA from
-- module objects (lua/vdnn.lua, operator-level C ABI) -- lua/encoders/lf-ques.lua, lua/encoders/mn-att-ques-im-hist.lua (the
-- flagship) + lua/decoders/disc.lua, and lua/encoders/lf-ques.lua + lua/decoders/gen.lua = BASELINE.json configs[0].  Same control flow
-- as the reference, call for call: encoder:forward(inputs) -> forwardConnect -> decoder:forward({options, encOut}) ->
-- criterion:forward / :backward -> decoder:backward -> encoder:backward(inputs, t[2]) (model.lua:297-337), wrapperW / wrapperdW from
-- getParameters() (model.lua:55), clamp(-5, 5) + adam + learning-rate decay (model.lua:96-105).  lua/model.lua is the other host:
-- the whole step behind the model-level ABI, any of the 11 x 2 pairs.
-- EXECUTED by the tests: tests/luavm (a Lua 5.1 evaluator with a LuaJIT-style ffi and a Torch7 tensor stub) runs this file against the real
-- library on the GPU -- loss, every gradient tensor and the post-Adam parameters against the library's model-level path, the fp64 oracle and the
-- golden fixtures (tests/test_lua_host_gpu.py) -- and against a bounds-checking dry library on the CPU (tests/test_luavm_cpu.py).
local ffi = require 'ffi'
local vdnn = dofile('vdnn.lua')
local vd = vdnn.vd

local ModelOps = torch.class('ModelOps')

function ModelOps:__init(params)
    self.params = params
    local encoder = dofile(string.format('encoders/%s.lua', params.encoder))
    local decoder = dofile(string.format('decoders/%s.lua', params.decoder))
    self.encoder = encoder.model(params)
    self.decoder = decoder.model(params, self.encoder)
    assert(self.encoder.build and self.decoder.build, 'this plug-in pair has no operator-level implementation in Lua: use lua/model.lua')
    self.forwardConnect, self.backwardConnect = decoder.forwardConnect, decoder.backwardConnect
    vd.call('vd_set_device', tonumber(os.getenv('VD_DEVICE') or '') or ((params.gpuid or -1) >= 0 and params.gpuid or 0))
    -- wrapper:getParameters(): embed | encoder tensors | decoder tensors
    local spec = {{'embed', (params.vocabSize + 1) * params.embedSize}}
    self.encoder:declare(spec); self.decoder:declare(spec)
    self.fp = vdnn.FlatParams(spec)
    self.wrapperW, self.wrapperdW = self.fp.W, self.fp.dW
    local wordEmbed = vdnn.LookupTableMaskZero(self.fp, 'embed', params.vocabSize, params.embedSize)
    self.encoder:build(vdnn, self.fp, wordEmbed); self.decoder:build(vdnn, self.fp, wordEmbed)
    self.wordEmbed = wordEmbed
    self.optims = {learningRate = params.learningRate, t = 0}
end

-- [B x R x T] (or [N x O x T]) token tensor -> device int32 [T x rows] time-major (model.lua:255-257: view(-1, T):t())
local function timeMajor(t)
    local T = t:size(t:dim())
    local tm = t:int():view(-1, T):t():contiguous()
    return {tok = vdnn.devInts(tm), T = T, N = tm:size(2)}
end

-- wrapper:training() / wrapper:evaluate() (model.lua:57,111,144): the Dropout nodes of the Lua-composed modules
function ModelOps:training() vdnn.training = true end
function ModelOps:evaluate() vdnn.training = false end

function ModelOps:forwardBackward(batch, onlyForward)
    local p = self.params
    vdnn.releaseStep()                                                                  -- the previous step's activations (vdnn.lua: step-scoped arena)
    -- the reference's input table, in its order (model.lua:252-294): ques [, img] [, hist] [, mask]
    local inputs = {timeMajor(batch['ques_fwd'])}
    if p.useIm == true then
        -- ONE feature map per image on the device: the reference's repeatTensor over the 10 rounds (model.lua:262-270) is an index
        -- computation inside the attention kernels' loaders
        local f = batch['img_feat']:float():contiguous()
        local d = vdnn.devFloats(f:nElement())
        vd.call('vd_memcpy_h2d', d, f:data(), f:nElement() * 4, nil)
        table.insert(inputs, {data = d, B = f:size(1)})
    end
    if p.useHistory == true then table.insert(inputs, timeMajor(batch['hist'])) end
    if string.match(p.encoder, 'mn') then
        -- model.lua:280-294: round i attends to facts j <= i; 1 = hidden; repeated over the dialogs of the batch
        local R = p.maxQuesCount
        local mask = torch.ones(R, R):byte()
        for i = 1, R do for j = 1, R do if j <= i then mask[i][j] = 0 end end end
        local maskRepeat = torch.repeatTensor(mask, batch['hist']:size(1), 1):contiguous()
        table.insert(inputs, vdnn.devBytesFrom(maskRepeat))
    end
    local N, H = inputs[1].N, p.rnnHiddenSize
    self.wordEmbed:zeroPad()
    local encOut = self.encoder:forward(inputs)                                        -- model.lua:297
    self.forwardConnect(self.encoder, self.decoder, encOut, inputs[1].T)                -- model.lua:300
    if p.decoder == 'gen' then
        -- model.lua:306-324: decoder:forward(answerIn) -> criterion (LogSoftMax + masked NLL, SUMMED over tokens, and its gradient in place
        -- of the logits: one kernel) -> decoder:backward -> backwardConnect -> encoder:backward(inputs, gradDecOut)
        local answerIn, answerOut = timeMajor(batch['answer_in']), timeMajor(batch['answer_out'])
        local rows = answerIn.T * answerIn.N
        local decOut = self.decoder:forward(answerIn)                                   -- model.lua:313
        local lossTok = vdnn.devFloats(rows)
        vd.call('vd_logsoftmax_nll', decOut, self.decoder.Vp, rows, p.vocabSize, answerIn.tok, answerOut.tok, lossTok, onlyForward and 0 or 1, nil)
        if not onlyForward then
            self.decoder:backward(answerIn, decOut)                                     -- model.lua:319 (decOut now holds gradCriterionOut)
            local gradDecOut = self.backwardConnect(self.encoder, self.decoder)         -- model.lua:322
            self.encoder:backward(inputs, gradDecOut)                                   -- model.lua:323
        end
        local hostTok = torch.FloatTensor(rows)
        vd.call('vd_stream_synchronize', nil)
        vd.call('vd_memcpy_d2h', hostTok:data(), lossTok, rows * 4, nil)
        return hostTok:sum(), batch['answer_out']:gt(0):sum()                           -- curLoss (the sum), numTokens (model.lua:76-85)
    end
    local options = timeMajor(batch['options'])
    local O = options.N / N
    -- 0-based targets; :clone() first -- :int() of an IntTensor is the SAME tensor and :add is in place: the caller's batch must not change
    local gt = vdnn.devInts(batch['answer_ind']:int():contiguous():view(-1):clone():add(-1))
    local decOut = self.decoder:forward({options, encOut})                              -- model.lua:329
    -- criterion:forward(decOut, answerInd) + :backward (model.lua:330,334): nn.MM + CrossEntropyCriterion and both gradients, one kernel
    local scores, lossRows = vdnn.devFloats(N * O), vdnn.devFloats(N)
    local dOptH, dEnc = vdnn.devFloats(options.N * H), vdnn.devFloats(N * H)
    if onlyForward then dOptH, dEnc = nil, nil end
    vd.call('vd_score_ce', decOut, encOut, gt, scores, lossRows, dOptH, dEnc, N, O, H, 1.0 / N, nil)
    if not onlyForward then
        local t = self.decoder:backward({options, encOut}, {dOptH, dEnc})              -- model.lua:335
        self.encoder:backward(inputs, t[2])                                            -- model.lua:337
    end
    local host = torch.FloatTensor(N)
    vd.call('vd_stream_synchronize', nil)
    vd.call('vd_memcpy_d2h', host:data(), lossRows, N * 4, nil)
    return host:mean(), scores, N, O
end

-- Model:retrieveBatch (model.lua:344-430, disc branch): the forward pass, then utils.computeRanks (utils.lua:106-128) on the option scores;
-- returns the ground-truth ranks [B x R] (params.useGt) or all ranks [N x O], as DoubleTensors like the reference
function ModelOps:retrieveBatch(batch)
    assert(self.params.decoder == 'disc', 'ModelOps:retrieveBatch: candidate log-likelihood retrieval of the gen decoder runs through lua/model.lua')
    local _, scores, N, O = self:forwardBackward(batch, true)
    local ranksDev = ffi.cast('int32_t*', vdnn.devBytes(N * O * 4))
    vd.call('vd_ranks', scores, ranksDev, N, O, nil)
    local ranks = torch.IntTensor(N, O)
    vd.call('vd_stream_synchronize', nil)
    vd.call('vd_memcpy_d2h', ranks:data(), ranksDev, N * O * 4, nil)
    if not self.params.useGt then return ranks:double() end
    local gtPos = batch['answer_ind']:view(-1)
    local out = torch.DoubleTensor(N)
    for n = 1, N do out[n] = ranks[n][gtPos[n]] end
    return out:view(-1, self.params.maxQuesCount)
end

function ModelOps:trainIteration(dataloader)
    self.fp:zeroGrad()                                                                  -- model.lua:68
    local batch = dataloader:getTrainBatch(self.params)
    local curLoss, numTokens = self:forwardBackward(batch)
    local cur = curLoss
    if self.params.decoder == 'gen' then cur = curLoss / numTokens end                  -- model.lua:76-85: the EMA is fed the per-token loss
    if (runningLoss or 0) > 0 then runningLoss = 0.95 * runningLoss + 0.05 * cur else runningLoss = cur end   -- model.lua:88-92
    -- wrapperdW:clamp(-5, 5); adam(wrapperW, wrapperdW, optims)  (model.lua:96-99; optim_updates.lua:62-91)
    local o = self.optims
    o.t = o.t + 1
    local step = o.learningRate * math.sqrt(1 - 0.999 ^ o.t) / (1 - 0.9 ^ o.t)
    vd.call('vd_clamp_adam', self.fp.W, self.fp.dW, self.fp.m, self.fp.v, self.fp.numel, 1.0, 5.0, 0.9, 0.999, 1e-8, step, nil)
    if o.learningRate > self.params.minLRate then o.learningRate = o.learningRate * self.params.lrDecayRate end   -- model.lua:102-105
    return curLoss
end

return ModelOps
