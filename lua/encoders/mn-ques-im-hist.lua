-- encoders/mn-ques-im-hist.lua -- the reference's plug-in file contract (model.lua:19-25) with BOTH surfaces (see lua/encoders/lf-ques.lua):
--   * enc.native = 'mn-ques-im-hist': the name lua/model.lua hands to vd_model_create (model-level C ABI);
--   * enc:declare / :build / :forward(inputs) / :backward(inputs, gradOutput) composed IN LUA from module objects over the operator-level
--     C ABI (lua/vdnn.lua), node for node of the reference's nngraph: text branches; query = Tanh(Linear(JoinTable{question
--     state, image feature})) (mn-ques-im-hist.lua:47-48); memory network over the dialog's facts (:50-65).
-- A sibling of lua/encoders/mn-att-ques-im-hist.lua (same blocks).
-- EXECUTED by the tests: tests/luavm (a Lua 5.1 evaluator with a LuaJIT-style ffi and a Torch7 tensor stub) runs this file against the real
-- library on the GPU -- loss, every gradient tensor and the post-Adam parameters against the library's model-level path, the fp64 oracle and the
-- golden fixtures (tests/test_lua_host_gpu.py) -- and against a bounds-checking dry library on the CPU (tests/test_luavm_cpu.py).
local encoderNet = {}

function encoderNet.model(params)
    local enc = {native = 'mn-ques-im-hist', params = params}
    enc.wordEmbed = {shared = 'embed'}           -- one table for question / history / option / answer tokens (model-level path)

    -- parameter tensors in getParameters() order: {name, numel}
    function enc:declare(spec)
        local E, H, F = params.embedSize, params.rnnHiddenSize, params.imgFeatureSize
        for _, name in ipairs({'hist', 'ques'}) do                            -- two layers per branch are hard-coded in the nngraph encoders
            table.insert(spec, {name .. '1.W', (E + H) * 4 * H}); table.insert(spec, {name .. '1.b', 4 * H})
            table.insert(spec, {name .. '2.W', (H + H) * 4 * H}); table.insert(spec, {name .. '2.b', 4 * H})
        end
        table.insert(spec, {'qi.W', H * (H + F)}); table.insert(spec, {'qi.b', H})
        table.insert(spec, {'mn1.W', H * H}); table.insert(spec, {'mn1.b', H})
        table.insert(spec, {'mn2.W', H * H}); table.insert(spec, {'mn2.b', H})
    end

    function enc:build(vdnn, fp, wordEmbed)
        local E, H, F = params.embedSize, params.rnnHiddenSize, params.imgFeatureSize
        self.vdnn, self.fp, self.wordEmbed = vdnn, fp, wordEmbed
        self.hist1, self.hist2 = vdnn.SeqLSTM(fp, 'hist1', E, H), vdnn.SeqLSTM(fp, 'hist2', H, H)
        self.ques1, self.ques2 = vdnn.SeqLSTM(fp, 'ques1', E, H), vdnn.SeqLSTM(fp, 'ques2', H, H)
        -- no enc.rnnLayers: the reference's nngraph encoders do not expose their LSTMs, so decoders/gen.lua:39-41,57-59 connects the
        -- top decoder layer to encOut only
        self.drop = vdnn.Dropout(0.5)             -- the nngraph encoders hard-code Dropout(0.5)
        self.qi = vdnn.LinearTanh(fp, 'qi', H + F, H)
        self.mn1, self.mn2 = vdnn.LinearTanh(fp, 'mn1', H, H), vdnn.LinearTanh(fp, 'mn2', H, H)
    end

    -- inputs = {ques, img, hist, mask} in the order of the reference's input table (model.lua:255-294); img = {data = device float [B x F], B}:
    -- one feature row per DIALOG (the repeatTensor over its rounds, model.lua:266-270, is a row gather here); mask = device uint8 [N x R]
    function enc:forward(inputs)
        local vd, vdnn, drop = self.vdnn.vd, self.vdnn, self.drop
        local ques, img, hist, mask = inputs[1], inputs[2], inputs[3], inputs[4]
        local E, H, F, R = params.embedSize, params.rnnHiddenSize, params.imgFeatureSize, params.maxQuesCount
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
        -- the image joins the question before the memory (mn-ques-im-hist.lua:47-48)
        local rep = torch.IntTensor(N)
        for n = 1, N do rep[n] = math.floor((n - 1) / R) end
        local imgRep, cat = vdnn.devFloats(N * F), vdnn.devFloats(N * (H + F))
        vd.call('vd_embed_gather', img.data, vdnn.devInts(rep), nil, imgRep, N, F, 1.0, nil)
        vd.call('vd_copy_2d', cat, H + F, q3, H, N, H, nil)
        vd.call('vd_copy_2d', cat + H, H + F, imgRep, F, N, F, nil)
        local query = self.qi:forward(cat, N)
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
        local ques, hist = inputs[1], inputs[3]
        local H, F, R = params.rnnHiddenSize, params.imgFeatureSize, params.maxQuesCount
        local N, B, S5 = self.N, self.B, drop.scale
        local du = gradOutput
        -- memory block
        local ds2 = self.mn2:backward(du)
        local dhatt = drop:apply(self.mn1:backward(ds2), self.m_hatt, N * H)
        local dq_att, dh3, dquery = vdnn.devFloats(N * H), vdnn.devFloats(N * H), vdnn.devFloats(N * H)
        vd.call('vd_mn_attention_backward', self.query, self.h3, self.prob, dhatt, dq_att, dh3, B, R, H, nil)
        vd.call('vd_axpby', dq_att, ds2, dquery, N * H, 1.0, 1.0, nil)
        local dcat = self.qi:backward(dquery)                                -- the question slice of JoinTable{question, image}
        local dq3 = vdnn.devFloats(N * H)
        vd.call('vd_copy_2d', dq3, H, dcat, H + F, N, H, nil)
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
