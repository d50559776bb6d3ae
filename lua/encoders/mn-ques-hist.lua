-- encoders/mn-ques-hist.lua -- the reference's plug-in file contract (model.lua:19-25) with BOTH surfaces (see lua/encoders/lf-ques.lua):
--   * enc.native = 'mn-ques-hist': the name lua/model.lua hands to vd_model_create (model-level C ABI);
--   * enc:declare / :build / :forward(inputs) / :backward(inputs, gradOutput) composed IN LUA from module objects over the operator-level
--     C ABI (lua/vdnn.lua), node for node of the reference's nngraph: shared LookupTableMaskZero -> Dropout -> 2 x SeqLSTM:maskZero() per text branch ->
--     Select(1,-1); memory network over the dialog's facts (mn-ques-hist.lua:43-58).
-- A sibling of lua/encoders/mn-att-ques-im-hist.lua (same blocks).
-- EXECUTED by the tests: tests/luavm (a Lua 5.1 evaluator with a LuaJIT-style ffi and a Torch7 tensor stub) runs this file against the real
-- library on the GPU -- loss, every gradient tensor and the post-Adam parameters against the library's model-level path, the fp64 oracle and the
-- golden fixtures (tests/test_lua_host_gpu.py) -- and against a bounds-checking dry library on the CPU (tests/test_luavm_cpu.py).
local encoderNet = {}

function encoderNet.model(params)
    local enc = {native = 'mn-ques-hist', params = params}
    enc.wordEmbed = {shared = 'embed'}           -- one table for question / history / option / answer tokens (model-level path)

    -- parameter tensors in getParameters() order: {name, numel}
    function enc:declare(spec)
        local E, H = params.embedSize, params.rnnHiddenSize
        for _, name in ipairs({'hist', 'ques'}) do                            -- two layers per branch are hard-coded in the nngraph encoders
            table.insert(spec, {name .. '1.W', (E + H) * 4 * H}); table.insert(spec, {name .. '1.b', 4 * H})
            table.insert(spec, {name .. '2.W', (H + H) * 4 * H}); table.insert(spec, {name .. '2.b', 4 * H})
        end
        table.insert(spec, {'mn1.W', H * H}); table.insert(spec, {'mn1.b', H})
        table.insert(spec, {'mn2.W', H * H}); table.insert(spec, {'mn2.b', H})
    end

    function enc:build(vdnn, fp, wordEmbed)
        local E, H = params.embedSize, params.rnnHiddenSize
        self.vdnn, self.fp, self.wordEmbed = vdnn, fp, wordEmbed
        self.hist1, self.hist2 = vdnn.SeqLSTM(fp, 'hist1', E, H), vdnn.SeqLSTM(fp, 'hist2', H, H)
        self.ques1, self.ques2 = vdnn.SeqLSTM(fp, 'ques1', E, H), vdnn.SeqLSTM(fp, 'ques2', H, H)
        -- no enc.rnnLayers: the reference's nngraph encoders do not expose their LSTMs, so decoders/gen.lua:39-41,57-59 connects the
        -- top decoder layer to encOut only
        self.drop = vdnn.Dropout(0.5)             -- the nngraph encoders hard-code Dropout(0.5)
        self.mn1, self.mn2 = vdnn.LinearTanh(fp, 'mn1', H, H), vdnn.LinearTanh(fp, 'mn2', H, H)
    end

    -- inputs = {ques, hist, mask} in the order of the reference's input table (model.lua:255-294 with useIm = false); mask = device uint8
    -- [N x R], 1 = hidden
    function enc:forward(inputs)
        local vd, vdnn, drop = self.vdnn.vd, self.vdnn, self.drop
        local ques, hist, mask = inputs[1], inputs[2], inputs[3]
        local E, H, R = params.embedSize, params.rnnHiddenSize, params.maxQuesCount
        local N, Tq, Th = ques.N, ques.T, hist.T
        local B = N / R
        local S5 = drop.scale
        -- text branches: embedding + Dropout fused in the gather; maskZero via the token matrix
        self.m_h, self.m_q = drop:mask(Th * N * E, 'h_emb'), drop:mask(Tq * N * E, 'q_emb')
        local hx = self.wordEmbed:forward(hist.tok, Th * N, self.m_h, S5)
        local qx = self.wordEmbed:forward(ques.tok, Tq * N, self.m_q, S5)
        self.hist1:forward(hx, Th, N, hist.tok); self.hist2:forward(self.hist1.output, Th, N, hist.tok)
        self.ques1:forward(qx, Tq, N, ques.tok); self.ques2:forward(self.ques1.output, Tq, N, ques.tok)
        local h3 = self.hist2.output + (Th - 1) * N * H                               -- nn.Select(1, -1)
        local q3 = self.ques2.output + (Tq - 1) * N * H
        self.h3, self.q3, self.N, self.B = h3, q3, N, B
        local query = q3
        -- memory network over the dialog's facts: nn.MM -> MaskSoftMax -> nn.MM -> Tanh(Linear(Dropout)) -> Tanh(Linear(hAttTr + query))
        self.query = query
        self.prob = vdnn.devFloats(N * R)
        local hatt = vdnn.devFloats(N * H)
        vd.call('vd_mn_attention_forward', query, h3, mask, self.prob, hatt, B, R, H, nil)
        self.m_hatt = drop:mask(N * H, 'hatt')
        local hattTr = self.mn1:forward(drop:apply(hatt, self.m_hatt, N * H), N)
        local s2 = vdnn.devFloats(N * H)
        vd.call('vd_axpby', hattTr, query, s2, N * H, 1.0, 1.0, nil)                                   -- nn.CAddTable
        local u = self.mn2:forward(s2, N)
        self.output = u
        return self.output
    end

    function enc:backward(inputs, gradOutput)
        local vd, vdnn, drop = self.vdnn.vd, self.vdnn, self.drop
        local ques, hist = inputs[1], inputs[2]
        local H, R = params.rnnHiddenSize, params.maxQuesCount
        local N, B, S5 = self.N, self.B, drop.scale
        local du = gradOutput
        -- memory block
        local ds2 = self.mn2:backward(du)
        local dhatt = drop:apply(self.mn1:backward(ds2), self.m_hatt, N * H)
        local dq_att, dh3, dquery = vdnn.devFloats(N * H), vdnn.devFloats(N * H), vdnn.devFloats(N * H)
        vd.call('vd_mn_attention_backward', self.query, self.h3, self.prob, dhatt, dq_att, dh3, B, R, H, nil)
        vd.call('vd_axpby', dq_att, ds2, dquery, N * H, 1.0, 1.0, nil)
        local dq3 = dquery
        -- text branches: the gradient arrives at the last step of the top layers only
        local dh1_seq = self.hist2:backward(nil, dh3, true)
        local dhx = self.hist1:backward(dh1_seq, nil, true)
        local dq1_seq = self.ques2:backward(nil, dq3, true)
        local dqx = self.ques1:backward(dq1_seq, nil, true)
        self.wordEmbed:backward(hist.tok, hist.T * N, dhx, self.m_h, S5)
        self.wordEmbed:backward(ques.tok, ques.T * N, dqx, self.m_q, S5)
    end

    return enc
end

return encoderNet
