-- encoders/mn-att-ques-im-hist.lua -- the reference's plug-in file contract (model.lua:19-25: the file is `dofile`d and returns a table
-- with model(params)) with BOTH surfaces (see lua/encoders/lf-ques.lua):
--   * enc.native = 'mn-att-ques-im-hist': the name lua/model.lua hands to vd_model_create (model-level C ABI: the whole step in the
--     library -- the path bench.py measures);
--   * enc:declare / enc:build / enc:forward(inputs) / enc:backward(inputs, gradOutput): the encoder composed IN LUA, node for node of
--     the reference's nngraph (encoders/mn-att-ques-im-hist.lua:21-106), from module objects over the operator-level C ABI
--     (lua/vdnn.lua), driven by lua/model_ops.lua.  A user who wants to change the flagship encoder edits THIS file.
-- EXECUTED by the tests: tests/luavm (a Lua 5.1 evaluator with a LuaJIT-style ffi and a Torch7 tensor stub) runs this file against the real
-- library on the GPU -- loss, every gradient tensor and the post-Adam parameters against the library's model-level path, the fp64 oracle and the
-- golden fixtures (tests/test_lua_host_gpu.py) -- and against a bounds-checking dry library on the CPU (tests/test_luavm_cpu.py).
local encoderNet = {}

function encoderNet.model(params)
    local enc = {native = 'mn-att-ques-im-hist', params = params}
    enc.wordEmbed = {shared = 'embed'}           -- one table for question / history / option / answer tokens (model-level path)

    -- parameter tensors in getParameters() order: {name, numel}
    function enc:declare(spec)
        local E, H, C, K = params.embedSize, params.rnnHiddenSize, params.imgFeatureSize, params.commonEmbeddingSize
        for _, name in ipairs({'hist', 'ques'}) do                            -- mn-att:27-41 (two layers are hard-coded there)
            table.insert(spec, {name .. '1.W', (E + H) * 4 * H}); table.insert(spec, {name .. '1.b', 4 * H})
            table.insert(spec, {name .. '2.W', (H + H) * 4 * H}); table.insert(spec, {name .. '2.b', 4 * H})
        end
        table.insert(spec, {'mn1.W', H * H}); table.insert(spec, {'mn1.b', H})               -- mn-att:64
        table.insert(spec, {'mn2.W', H * H}); table.insert(spec, {'mn2.b', H})               -- mn-att:65
        table.insert(spec, {'img_proj.W', H * C}); table.insert(spec, {'img_proj.b', H})     -- mn-att:77
        table.insert(spec, {'img_common.W', K * H}); table.insert(spec, {'img_common.b', K}) -- mn-att:84
        table.insert(spec, {'ques_common.W', K * H}); table.insert(spec, {'ques_common.b', K})   -- mn-att:88
        table.insert(spec, {'att.W', K}); table.insert(spec, {'att.b', 1})                   -- mn-att:93
        table.insert(spec, {'out.W', H * H}); table.insert(spec, {'out.b', H})               -- mn-att:106
    end

    -- module objects over the flat parameter vectors (vdnn.FlatParams); wordEmbed is created by the Model and shared with the decoder
    function enc:build(vdnn, fp, wordEmbed)
        local E, H, C, K = params.embedSize, params.rnnHiddenSize, params.imgFeatureSize, params.commonEmbeddingSize
        self.vdnn, self.fp, self.wordEmbed = vdnn, fp, wordEmbed
        self.hist1, self.hist2 = vdnn.SeqLSTM(fp, 'hist1', E, H), vdnn.SeqLSTM(fp, 'hist2', H, H)
        self.ques1, self.ques2 = vdnn.SeqLSTM(fp, 'ques1', E, H), vdnn.SeqLSTM(fp, 'ques2', H, H)
        -- no enc.rnnLayers: the reference's nngraph encoders do not expose their LSTMs, so decoders/gen.lua:39-41,57-59 connects the
        -- top decoder layer to encOut only
        self.mn1, self.mn2 = vdnn.LinearTanh(fp, 'mn1', H, H), vdnn.LinearTanh(fp, 'mn2', H, H)
        self.img_proj = vdnn.LinearTanh(fp, 'img_proj', C, H)
        self.ques_common = vdnn.Linear(fp, 'ques_common', H, K)
        self.out = vdnn.LinearTanh(fp, 'out', H, H)
        self.drop = vdnn.Dropout(0.5)             -- the nngraph encoders hard-code Dropout(0.5) (mn-att:24,25,64,74,92,106)
    end

    -- inputs = {ques, img, hist, mask} in the order of the reference's input table (model.lua:255-294):
    --   ques / hist = {tok = device int32 [T x N] time-major, T, N}; img = {data = device float [B*S2 x C], B}: ONE map per image (the
    --   10x repeatTensor of model.lua:262-265 is folded into the attention kernels' loaders); mask = device uint8 [N x R], 1 = hidden.
    -- Returns encOut [N x H].
    function enc:forward(inputs)
        local vd, vdnn, fp, drop = self.vdnn.vd, self.vdnn, self.fp, self.drop
        local ques, img, hist, mask = inputs[1], inputs[2], inputs[3], inputs[4]
        local E, H, K, R = params.embedSize, params.rnnHiddenSize, params.commonEmbeddingSize, params.maxQuesCount
        local S2 = params.imgSpatialSize * params.imgSpatialSize
        local N, Tq, Th = ques.N, ques.T, hist.T
        local B = N / R
        local S5 = drop.scale
        -- text branches (mn-att:21-45): embedding + Dropout fused in the gather; maskZero via the token matrix
        self.m_h, self.m_q = drop:mask(Th * N * E, 'h_emb'), drop:mask(Tq * N * E, 'q_emb')
        local hx = self.wordEmbed:forward(hist.tok, Th * N, self.m_h, S5)
        local qx = self.wordEmbed:forward(ques.tok, Tq * N, self.m_q, S5)
        self.hist1:forward(hx, Th, N, hist.tok); self.hist2:forward(self.hist1.output, Th, N, hist.tok)
        self.ques1:forward(qx, Tq, N, ques.tok); self.ques2:forward(self.ques1.output, Tq, N, ques.tok)
        local h3 = self.hist2.output + (Th - 1) * N * H                               -- nn.Select(1, -1)
        local q3 = self.ques2.output + (Tq - 1) * N * H
        self.h3, self.q3, self.N, self.B = h3, q3, N, B
        -- memory network over the dialog's facts (mn-att:48-65)
        self.prob = vdnn.devFloats(N * R)
        local hatt = vdnn.devFloats(N * H)
        vd.call('vd_mn_attention_forward', q3, h3, mask, self.prob, hatt, B, R, H, nil)
        self.m_hatt = drop:mask(N * H, 'hatt')
        local hattTr = self.mn1:forward(drop:apply(hatt, self.m_hatt, N * H), N)
        local s2 = vdnn.devFloats(N * H)
        vd.call('vd_axpby', hattTr, q3, s2, N * H, 1.0, 1.0, nil)                                      -- nn.CAddTable
        local qh2 = self.mn2:forward(s2, N)
        -- stacked attention over the S x S regions, one hop (mn-att:68-104): per-IMAGE projection, per-round Dropout masks in the loaders
        self.pre = self.img_proj:forward(img.data, B * S2)                            -- Tanh(Linear(img)), pre-Dropout
        self.m1, self.m2 = drop:mask(N * S2 * H, 'img_tr'), drop:mask(N * S2 * K, 'iqc')
        self.sc = self.m1 ~= nil and S5 or 1.0
        local qc = self.ques_common:forward(qh2, N)                                    -- mn-att:88
        local Wc, _ = fp:view('img_common.W'); local bc, _ = fp:view('img_common.b')
        local wa, _ = fp:view('att.W'); local ba, _ = fp:view('att.b')
        self.iqc, self.patt = vdnn.devFloats(N * S2 * K), vdnn.devFloats(N * S2)
        local u1 = vdnn.devFloats(N * H)
        vd.call('vd_img_common_forward', self.pre, self.m1, Wc, bc, qc, self.m2, self.iqc, N, R, S2, H, K, self.sc, nil)     -- mn-att:83-92
        vd.call('vd_img_att_forward', self.iqc, wa, ba, self.pre, self.m1, qh2, self.patt, u1, N, R, S2, H, K, self.sc, nil)  -- mn-att:93-102
        self.m_u = drop:mask(N * H, 'u')
        self.output = self.out:forward(drop:apply(u1, self.m_u, N * H), N)              -- mn-att:106
        return self.output
    end

    function enc:backward(inputs, gradOutput)
        local vd, vdnn, fp, drop = self.vdnn.vd, self.vdnn, self.fp, self.drop
        local ques, hist = inputs[1], inputs[3]
        local E, H, K, R = params.embedSize, params.rnnHiddenSize, params.commonEmbeddingSize, params.maxQuesCount
        local S2 = params.imgSpatialSize * params.imgSpatialSize
        local N, B, sc, S5 = self.N, self.B, self.sc, drop.scale
        local Wc, dWc = fp:view('img_common.W'); local _, dbc = fp:view('img_common.b')
        local wa, dwa = fp:view('att.W'); local _, dba = fp:view('att.b')
        local du = drop:apply(self.out:backward(gradOutput), self.m_u, N * H)           -- d att of the hop + its residual
        local dpre, dqc, work = vdnn.devFloats(B * S2 * H), vdnn.devFloats(N * K), vdnn.devFloats(N * S2)
        vd.call('vd_img_att_backward', self.iqc, wa, self.pre, self.m1, self.m2, self.patt, du, dwa, dba, dqc, work, N, R, S2, H, K, sc, nil)
        local dz = self.iqc                                                             -- iqc now holds dz
        vd.call('vd_colsum_acc', dz, K, N * S2, K, dbc, nil)
        vd.call('vd_img_common_wgrad', dz, self.pre, self.m1, dWc, N, R, S2, H, K, sc, nil)
        vd.call('vd_img_tr_backward', dz, Wc, self.patt, du, self.m1, dpre, N, R, S2, H, K, sc, nil)        -- += into dpre
        local du_q = self.ques_common:backward(dqc)
        local dqh2 = vdnn.devFloats(N * H)
        vd.call('vd_axpby', du_q, du, dqh2, N * H, 1.0, 1.0, nil)                        -- residual CAddTable (mn-att:102)
        self.img_proj:backward(dpre, false)                                             -- tanh' + dW, db of mn-att:77
        -- memory block
        local ds2 = self.mn2:backward(dqh2)
        local dhatt = drop:apply(self.mn1:backward(ds2), self.m_hatt, N * H)
        local dq_att, dh3, dq3 = vdnn.devFloats(N * H), vdnn.devFloats(N * H), vdnn.devFloats(N * H)
        vd.call('vd_mn_attention_backward', self.q3, self.h3, self.prob, dhatt, dq_att, dh3, B, R, H, nil)
        vd.call('vd_axpby', dq_att, ds2, dq3, N * H, 1.0, 1.0, nil)
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
