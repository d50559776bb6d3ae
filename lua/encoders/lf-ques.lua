-- encoders/lf-ques.lua -- the reference's plug-in file contract (model.lua:19-25: the file is `dofile`d and returns a table with
-- model(params)) with BOTH surfaces:
--   * enc.native = 'lf-ques': the name lua/model.lua hands to vd_model_create (model-level C ABI: the whole step in the library);
--   * enc:build(fp) / enc:forward(inputs) / enc:backward(inputs, gradOutput): the encoder composed IN LUA from module objects over
--     the operator-level C ABI (lua/vdnn.lua) -- the counterpart of encoders/lf-ques.lua:6-33 of the reference (wordEmbed ->
--     numLayers x SeqLSTM:maskZero() -> Select(1,-1) -> Dropout -> Linear -> Tanh), driven by lua/model_ops.lua.  A user who wants a
--     new encoder writes a file like this one.
-- EXECUTED by the tests: tests/luavm (a Lua 5.1 evaluator with a LuaJIT-style ffi and a Torch7 tensor stub) runs this file against the real
-- library on the GPU -- loss, every gradient tensor and the post-Adam parameters against the library's model-level path, the fp64 oracle and the
-- golden fixtures (tests/test_lua_host_gpu.py) -- and against a bounds-checking dry library on the CPU (tests/test_luavm_cpu.py).
local encoderNet = {}

function encoderNet.model(params)
    local enc = {native = 'lf-ques', params = params}
    enc.wordEmbed = {shared = 'embed'}           -- one table for question / history / option / answer tokens (model-level path)

    -- parameter tensors in getParameters() order: {name, numel}
    function enc:declare(spec)
        local E, H = params.embedSize, params.rnnHiddenSize
        for layer = 1, params.numLayers do
            local D = (layer == 1) and E or H
            table.insert(spec, {'ques' .. layer .. '.W', (D + H) * 4 * H})
            table.insert(spec, {'ques' .. layer .. '.b', 4 * H})
        end
        table.insert(spec, {'fuse.W', H * H}); table.insert(spec, {'fuse.b', H})
    end

    -- module objects over the flat parameter vectors (vdnn.FlatParams); wordEmbed is created by the Model and shared with the decoder
    function enc:build(vdnn, fp, wordEmbed)
        local E, H = params.embedSize, params.rnnHiddenSize
        self.vdnn, self.wordEmbed, self.rnnLayers = vdnn, wordEmbed, {}
        for layer = 1, params.numLayers do
            self.rnnLayers[layer] = vdnn.SeqLSTM(fp, 'ques' .. layer, (layer == 1) and E or H, H)
        end
        self.fuse = vdnn.LinearTanh(fp, 'fuse', H, H)
        self.drop = vdnn.Dropout(params.dropout or 0.5)                      -- lf-ques.lua:29-31: nn.Dropout(dropout) in front of the Linear
    end

    -- inputs = {ques}: device int32 [Tq x N] time-major (+ .T, .N); returns encOut [N x H].  Dropout: the vdnn.Dropout module (identity
    -- after ModelOps:evaluate(); the C twin runs in evaluate mode, the training-mode plumbing is the flagship twin's)
    function enc:forward(inputs)
        local ques = inputs[1]
        local T, N, H = ques.T, ques.N, params.rnnHiddenSize
        local x = self.wordEmbed:forward(ques.tok, T * N)
        for layer = 1, #self.rnnLayers do x = self.rnnLayers[layer]:forward(x, T, N, ques.tok) end
        local last = x + (T - 1) * N * H                                  -- nn.Select(1, -1)
        self.N = N
        self.m_f = ((params.dropout or 0.5) > 0) and self.drop:mask(N * H, 'fuse') or nil    -- nil = identity (evaluate(), or dropout = 0)
        self.output = self.fuse:forward(self.drop:apply(last, self.m_f, N * H), N)
        return self.output
    end

    function enc:backward(inputs, gradOutput)
        local ques = inputs[1]
        local dLast = self.drop:apply(self.fuse:backward(gradOutput), self.m_f, self.N * params.rnnHiddenSize)
        local L = #self.rnnLayers
        local dSeq = self.rnnLayers[L]:backward(nil, dLast, true)        -- the gradient arrives at the last step only
        for layer = L - 1, 1, -1 do dSeq = self.rnnLayers[layer]:backward(dSeq, nil, true) end
        self.wordEmbed:backward(ques.tok, ques.T * ques.N, dSeq)
    end

    return enc
end

return encoderNet
