-- encoders/hre-ques-hist.lua -- the reference's plug-in file contract (model.lua:19-25) with BOTH surfaces (see lua/encoders/lf-ques.lua):
--   * enc.native = 'hre-ques-hist': the name lua/model.lua hands to vd_model_create (model-level C ABI);
--   * enc:declare / :build / :forward(inputs) / :backward(inputs, gradOutput) composed IN LUA from module objects over the operator-level
--     C ABI (lua/vdnn.lua): the counterpart of encoders/hre-ques-hist.lua of the reference -- history LSTM stack; question LSTM stack;
--     dialog-level SeqLSTM(2H, H) over the R rounds of every dialog on JoinTable{question state, history state}, between the two row
--     permutations of nn.View / nn.Transpose (hre-ques-im-hist.lua without the image branch).
-- EXECUTED by the tests: tests/luavm (a Lua 5.1 evaluator with a LuaJIT-style ffi and a Torch7 tensor stub) runs this file against the real
-- library on the GPU -- loss, every gradient tensor and the post-Adam parameters against the library's model-level path, the fp64 oracle and the
-- golden fixtures (tests/test_lua_host_gpu.py) -- and against a bounds-checking dry library on the CPU (tests/test_luavm_cpu.py).
local encoderNet = {}

function encoderNet.model(params)
    local enc = {native = 'hre-ques-hist', params = params}
    enc.wordEmbed = {shared = 'embed'}           -- one table for question / history / option / answer tokens (model-level path)

    -- parameter tensors in getParameters() order: {name, numel}
    function enc:declare(spec)
        local E, H = params.embedSize, params.rnnHiddenSize
        for layer = 1, params.numLayers do
            local D = (layer == 1) and E or H
            table.insert(spec, {'hist' .. layer .. '.W', (D + H) * 4 * H}); table.insert(spec, {'hist' .. layer .. '.b', 4 * H})
        end
        for layer = 1, params.numLayers do
            local D = (layer == 1) and E or H
            table.insert(spec, {'ques' .. layer .. '.W', (D + H) * 4 * H}); table.insert(spec, {'ques' .. layer .. '.b', 4 * H})
        end
        table.insert(spec, {'dialog.W', (2 * H + H) * 4 * H}); table.insert(spec, {'dialog.b', 4 * H})
    end

    function enc:build(vdnn, fp, wordEmbed)
        local E, H = params.embedSize, params.rnnHiddenSize
        self.vdnn, self.wordEmbed, self.rnnLayers, self.histLayers = vdnn, wordEmbed, {}, {}
        for layer = 1, params.numLayers do
            self.histLayers[layer] = vdnn.SeqLSTM(fp, 'hist' .. layer, (layer == 1) and E or H, H)
            self.rnnLayers[layer] = vdnn.SeqLSTM(fp, 'ques' .. layer, (layer == 1) and E or H, H)
        end
        self.dialog = vdnn.SeqLSTM(fp, 'dialog', 2 * H, H)
    end

    -- row permutations of nn.View(-1, R, 2H) + nn.Transpose({1, 2}) (hre:88-93), as device index vectors (built once per batch size)
    function enc:indices(N)
        if self.idxN == N then return self.toRb, self.toN end
        local R = params.maxQuesCount
        local B = N / R
        local toRb, toN = torch.IntTensor(N), torch.IntTensor(N)
        for i = 1, N do
            local n = i - 1
            toRb[i] = (n % B) * R + math.floor(n / B)           -- round-major row r * B + b  <-  dialog-major row b * R + r
            toN[i] = (n % R) * B + math.floor(n / R)            -- and back
        end
        self.idxN = N
        self.vdnn.persistent(function() self.toRb, self.toN = self.vdnn.devInts(toRb), self.vdnn.devInts(toN) end)   -- cached across steps
        return self.toRb, self.toN
    end

    -- inputs = {ques, hist} in the order of the reference's input table (model.lua:252-279)
    function enc:forward(inputs)
        local vd, vdnn = self.vdnn.vd, self.vdnn
        local ques, hist = inputs[1], inputs[2]
        local H, R = params.rnnHiddenSize, params.maxQuesCount
        local N, Tq, Th = ques.N, ques.T, hist.T
        local B, L = N / R, #self.rnnLayers
        local toRb, toN = self:indices(N)
        local x = self.wordEmbed:forward(hist.tok, Th * N)
        for layer = 1, L do x = self.histLayers[layer]:forward(x, Th, N, hist.tok) end
        local hh = x + (Th - 1) * N * H
        x = self.wordEmbed:forward(ques.tok, Tq * N)
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
        local ques, hist = inputs[1], inputs[2]
        local H = params.rnnHiddenSize
        local N, Tq, L = self.N, ques.T, #self.rnnLayers
        local toRb, toN = self:indices(N)
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
        self.wordEmbed:backward(ques.tok, Tq * N, dSeq)
    end

    return enc
end

return encoderNet
