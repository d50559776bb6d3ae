-- encoders/hre-ques-im-hist.lua -- the reference's plug-in file contract (model.lua:19-25) with BOTH surfaces (see lua/encoders/lf-ques.lua):
--   * enc.native = 'hre-ques-im-hist': the name lua/model.lua hands to vd_model_create (model-level C ABI);
--   * enc:declare / :build / :forward(inputs) / :backward(inputs, gradOutput) composed IN LUA from module objects over the operator-level
--     C ABI (lua/vdnn.lua): the counterpart of encoders/hre-ques-im-hist.lua:5-97 of the reference -- history LSTM stack; question LSTM stack
--     over JoinTable{word embedding, MaskTime(Linear(image))}; dialog-level SeqLSTM(2H, H) over the R rounds of every dialog on
--     JoinTable{question state, history state}, between the two row permutations of nn.View / nn.Transpose.  With lua/decoders/disc.lua
--     this is BASELINE.json configs[2].
-- EXECUTED by the tests: tests/luavm (a Lua 5.1 evaluator with a LuaJIT-style ffi and a Torch7 tensor stub) runs this file against the real
-- library on the GPU -- loss, every gradient tensor and the post-Adam parameters against the library's model-level path, the fp64 oracle and the
-- golden fixtures (tests/test_lua_host_gpu.py) -- and against a bounds-checking dry library on the CPU (tests/test_luavm_cpu.py).
local encoderNet = {}

function encoderNet.model(params)
    local enc = {native = 'hre-ques-im-hist', params = params}
    enc.wordEmbed = {shared = 'embed'}           -- one table for question / history / option / answer tokens (model-level path)

    -- parameter tensors in getParameters() order: {name, numel}
    function enc:declare(spec)
        local E, H, F, DI = params.embedSize, params.rnnHiddenSize, params.imgFeatureSize, params.imgEmbedSize
        for layer = 1, params.numLayers do                                      -- hre:28-35
            local D = (layer == 1) and E or H
            table.insert(spec, {'hist' .. layer .. '.W', (D + H) * 4 * H}); table.insert(spec, {'hist' .. layer .. '.b', 4 * H})
        end
        table.insert(spec, {'img_embed.W', DI * F}); table.insert(spec, {'img_embed.b', DI})       -- hre:46
        for layer = 1, params.numLayers do                                      -- hre:66-75
            local D = (layer == 1) and (E + DI) or H
            table.insert(spec, {'ques' .. layer .. '.W', (D + H) * 4 * H}); table.insert(spec, {'ques' .. layer .. '.b', 4 * H})
        end
        table.insert(spec, {'dialog.W', (2 * H + H) * 4 * H}); table.insert(spec, {'dialog.b', 4 * H})   -- hre:92
    end

    function enc:build(vdnn, fp, wordEmbed)
        local E, H, F, DI = params.embedSize, params.rnnHiddenSize, params.imgFeatureSize, params.imgEmbedSize
        self.vdnn, self.wordEmbed, self.rnnLayers, self.histLayers = vdnn, wordEmbed, {}, {}
        for layer = 1, params.numLayers do
            self.histLayers[layer] = vdnn.SeqLSTM(fp, 'hist' .. layer, (layer == 1) and E or H, H)
            self.rnnLayers[layer] = vdnn.SeqLSTM(fp, 'ques' .. layer, (layer == 1) and (E + DI) or H, H)
        end
        self.img_embed = vdnn.Linear(fp, 'img_embed', F, DI)                    -- plain nn.Linear
        self.dialog = vdnn.SeqLSTM(fp, 'dialog', 2 * H, H)
    end

    -- row permutations of nn.View(-1, R, 2H) + nn.Transpose({1, 2}) (hre:88-93), as device index vectors (built once per batch size)
    function enc:indices(N)
        if self.idxN == N then return self.rep, self.toRb, self.toN end
        local R = params.maxQuesCount
        local B = N / R
        local rep, toRb, toN = torch.IntTensor(N), torch.IntTensor(N), torch.IntTensor(N)
        for i = 1, N do
            local n = i - 1
            rep[i] = math.floor(n / R)                          -- image row of QA round n = b * R + r
            toRb[i] = (n % B) * R + math.floor(n / B)           -- round-major row r * B + b  <-  dialog-major row b * R + r
            toN[i] = (n % R) * B + math.floor(n / R)            -- and back
        end
        self.idxN = N
        self.vdnn.persistent(function()                                                  -- cached across steps
            self.rep, self.toRb, self.toN = self.vdnn.devInts(rep), self.vdnn.devInts(toRb), self.vdnn.devInts(toN)
        end)
        return self.rep, self.toRb, self.toN
    end

    -- inputs = {ques, img, hist} in the order of the reference's input table (model.lua:252-279); img = {data = device float [B x F], B}
    function enc:forward(inputs)
        local vd, vdnn = self.vdnn.vd, self.vdnn
        local ques, img, hist = inputs[1], inputs[2], inputs[3]
        local E, H, F, DI, R = params.embedSize, params.rnnHiddenSize, params.imgFeatureSize, params.imgEmbedSize, params.maxQuesCount
        local N, Tq, Th = ques.N, ques.T, hist.T
        local B, L, DQ = N / R, #self.rnnLayers, E + DI
        local rep, toRb, toN = self:indices(N)
        local x = self.wordEmbed:forward(hist.tok, Th * N)
        for layer = 1, L do x = self.histLayers[layer]:forward(x, Th, N, hist.tok) end
        local hh = x + (Th - 1) * N * H
        -- question branch: the image embedding is repeated over the time steps of its round, zero at pad steps (MaskTime), and joined to the
        -- word embedding column-wise
        local qx = self.wordEmbed:forward(ques.tok, Tq * N)
        local imgRep, xi = vdnn.devFloats(N * F), vdnn.devFloats(Tq * N * DI)
        vd.call('vd_embed_gather', img.data, rep, nil, imgRep, N, F, 1.0, nil)
        local imgE = self.img_embed:forward(imgRep, N)                                            -- hre:43-48
        vd.call('vd_mask_time_forward', imgE, ques.tok, xi, Tq, N, DI, nil)                       -- hre:50-53
        local qcat = vdnn.devFloats(Tq * N * DQ)
        vd.call('vd_copy_2d', qcat, DQ, qx, E, Tq * N, E, nil)                                    -- nn.JoinTable(2, 2)
        vd.call('vd_copy_2d', qcat + E, DQ, xi, DI, Tq * N, DI, nil)
        x = qcat
        for layer = 1, L do x = self.rnnLayers[layer]:forward(x, Tq, N, ques.tok) end
        local hq = x + (Tq - 1) * N * H
        -- dialog-level recurrence over the rounds (hre:84-95): rows to round-major, JoinTable{question, history}, SeqLSTM(2H, H), rows back
        local fRb, sRb, dcat = vdnn.devFloats(N * H), vdnn.devFloats(N * H), vdnn.devFloats(N * 2 * H)
        vd.call('vd_embed_gather', hq, toRb, nil, fRb, N, H, 1.0, nil)
        vd.call('vd_embed_gather', hh, toRb, nil, sRb, N, H, 1.0, nil)
        vd.call('vd_copy_2d', dcat, 2 * H, fRb, H, N, H, nil)
        vd.call('vd_copy_2d', dcat + H, 2 * H, sRb, H, N, H, nil)
        self.dialog:forward(dcat, R, B, nil)
        self.N = N
        self.output = vdnn.devFloats(N * H)
        vd.call('vd_embed_gather', self.dialog.output, toN, nil, self.output, N, H, 1.0, nil)
        return self.output
    end

    function enc:backward(inputs, gradOutput)
        local vd, vdnn = self.vdnn.vd, self.vdnn
        local ques, hist = inputs[1], inputs[3]
        local E, H, DI = params.embedSize, params.rnnHiddenSize, params.imgEmbedSize
        local N, Tq, L, DQ = self.N, ques.T, #self.rnnLayers, E + DI
        local _, toRb, toN = self:indices(N)
        local gRb = vdnn.devFloats(N * H)
        vd.call('vd_embed_gather', gradOutput, toRb, nil, gRb, N, H, 1.0, nil)
        local ddcat = self.dialog:backward(gRb, nil, true)                                        -- [R*B x 2H]: the gradient arrives at every round
        local dfRb, dsRb, dq, dh = vdnn.devFloats(N * H), vdnn.devFloats(N * H), vdnn.devFloats(N * H), vdnn.devFloats(N * H)
        vd.call('vd_copy_2d', dfRb, H, ddcat, 2 * H, N, H, nil)
        vd.call('vd_copy_2d', dsRb, H, ddcat + H, 2 * H, N, H, nil)
        vd.call('vd_embed_gather', dfRb, toN, nil, dq, N, H, 1.0, nil)
        vd.call('vd_embed_gather', dsRb, toN, nil, dh, N, H, 1.0, nil)
        local dSeq = self.histLayers[L]:backward(nil, dh, true)
        for layer = L - 1, 1, -1 do dSeq = self.histLayers[layer]:backward(dSeq, nil, true) end
        self.wordEmbed:backward(hist.tok, hist.T * N, dSeq)
        dSeq = self.rnnLayers[L]:backward(nil, dq, true)
        for layer = L - 1, 1, -1 do dSeq = self.rnnLayers[layer]:backward(dSeq, nil, true) end
        local dqx, dxi, dimgE = vdnn.devFloats(Tq * N * E), vdnn.devFloats(Tq * N * DI), vdnn.devFloats(N * DI)
        vd.call('vd_copy_2d', dqx, E, dSeq, DQ, Tq * N, E, nil)                                   -- JoinTable backward
        vd.call('vd_copy_2d', dxi, DI, dSeq + E, DQ, Tq * N, DI, nil)
        self.wordEmbed:backward(ques.tok, Tq * N, dqx)
        vd.call('vd_mask_time_backward', dxi, ques.tok, dimgE, Tq, N, DI, nil)
        self.img_embed:backward(dimgE, false)
    end

    return enc
end

return encoderNet
