-- encoders/lf-ques-hist.lua -- the reference's plug-in file contract (model.lua:19-25) with BOTH surfaces (see lua/encoders/lf-ques.lua):
--   * enc.native = 'lf-ques-hist': the name lua/model.lua hands to vd_model_create (model-level C ABI);
--   * enc:declare / :build / :forward(inputs) / :backward(inputs, gradOutput) composed IN LUA from module objects over the operator-level
--     C ABI (lua/vdnn.lua): the counterpart of encoders/lf-ques-hist.lua of the reference -- question LSTM stack and history LSTM stack (on
--     the concatenated dialog) -> Select(1,-1), JoinTable{question state, history state} -> Dropout -> Linear -> Tanh.  enc.rnnLayers =
--     the QUESTION layers (what decoders/gen.lua connects to, gen.lua:31-35).
-- EXECUTED by the tests: tests/luavm (a Lua 5.1 evaluator with a LuaJIT-style ffi and a Torch7 tensor stub) runs this file against the real
-- library on the GPU -- loss, every gradient tensor and the post-Adam parameters against the library's model-level path, the fp64 oracle and the
-- golden fixtures (tests/test_lua_host_gpu.py) -- and against a bounds-checking dry library on the CPU (tests/test_luavm_cpu.py).
local encoderNet = {}

function encoderNet.model(params)
    local enc = {native = 'lf-ques-hist', params = params}
    enc.wordEmbed = {shared = 'embed'}           -- one table for question / history / option / answer tokens (model-level path)

    -- parameter tensors in getParameters() order: {name, numel}
    function enc:declare(spec)
        local E, H = params.embedSize, params.rnnHiddenSize
        for _, name in ipairs({'ques', 'hist'}) do
            for layer = 1, params.numLayers do
                local D = (layer == 1) and E or H
                table.insert(spec, {name .. layer .. '.W', (D + H) * 4 * H})
                table.insert(spec, {name .. layer .. '.b', 4 * H})
            end
        end
        table.insert(spec, {'fuse.W', H * (H + H)}); table.insert(spec, {'fuse.b', H})
    end

    function enc:build(vdnn, fp, wordEmbed)
        local E, H = params.embedSize, params.rnnHiddenSize
        self.vdnn, self.wordEmbed, self.rnnLayers, self.histLayers = vdnn, wordEmbed, {}, {}
        for layer = 1, params.numLayers do
            self.rnnLayers[layer] = vdnn.SeqLSTM(fp, 'ques' .. layer, (layer == 1) and E or H, H)
            self.histLayers[layer] = vdnn.SeqLSTM(fp, 'hist' .. layer, (layer == 1) and E or H, H)
        end
        self.fuse = vdnn.LinearTanh(fp, 'fuse', H + H, H)
        self.drop = vdnn.Dropout(params.dropout or 0.5)                      -- nn.Dropout(dropout) in front of the Linear (lf-ques-im-hist.lua:55-57)
    end

    -- inputs = {ques, hist} in the order of the reference's input table (model.lua:252-279): ques / hist = {tok = device int32
    -- [T x N] time-major, T, N}.  Dropout: the vdnn.Dropout module, as in lua/encoders/lf-ques.lua.
    function enc:forward(inputs)
        local vd, vdnn = self.vdnn.vd, self.vdnn
        local ques, hist = inputs[1], inputs[2]
        local H = params.rnnHiddenSize
        local N, Tq, Th = ques.N, ques.T, hist.T
        local L = #self.rnnLayers
        local Dcat = H + H
        local x = self.wordEmbed:forward(ques.tok, Tq * N)
        for layer = 1, L do x = self.rnnLayers[layer]:forward(x, Tq, N, ques.tok) end
        local qLast = x + (Tq - 1) * N * H                                  -- nn.Select(1, -1)
        x = self.wordEmbed:forward(hist.tok, Th * N)
        for layer = 1, L do x = self.histLayers[layer]:forward(x, Th, N, hist.tok) end
        local hLast = x + (Th - 1) * N * H
        local cat = vdnn.devFloats(N * Dcat)
        vd.call('vd_copy_2d', cat, Dcat, qLast, H, N, H, nil)               -- nn.JoinTable(1, 1)
        vd.call('vd_copy_2d', cat + H, Dcat, hLast, H, N, H, nil)
        self.N = N
        self.m_f = ((params.dropout or 0.5) > 0) and self.drop:mask(N * Dcat, 'fuse') or nil    -- nil = identity (evaluate(), or dropout = 0)
        self.output = self.fuse:forward(self.drop:apply(cat, self.m_f, N * Dcat), N)
        return self.output
    end

    function enc:backward(inputs, gradOutput)
        local vd, vdnn = self.vdnn.vd, self.vdnn
        local ques, hist = inputs[1], inputs[2]
        local H = params.rnnHiddenSize
        local N, L, Dcat = self.N, #self.rnnLayers, H + H
        local dCat = self.drop:apply(self.fuse:backward(gradOutput), self.m_f, N * Dcat)
        local dq, dhl = vdnn.devFloats(N * H), vdnn.devFloats(N * H)        -- JoinTable backward: question and history slices
        vd.call('vd_copy_2d', dq, H, dCat, Dcat, N, H, nil)
        vd.call('vd_copy_2d', dhl, H, dCat + H, Dcat, N, H, nil)
        local dSeq = self.histLayers[L]:backward(nil, dhl, true)
        for layer = L - 1, 1, -1 do dSeq = self.histLayers[layer]:backward(dSeq, nil, true) end
        self.wordEmbed:backward(hist.tok, hist.T * N, dSeq)
        dSeq = self.rnnLayers[L]:backward(nil, dq, true)
        for layer = L - 1, 1, -1 do dSeq = self.rnnLayers[layer]:backward(dSeq, nil, true) end
        self.wordEmbed:backward(ques.tok, ques.T * N, dSeq)
    end

    return enc
end

return encoderNet
