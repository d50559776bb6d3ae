-- decoders/disc.lua -- the reference's plug-in file contract (model.lua:22-29): returns a table with model(params, enc),
-- forwardConnect(enc, dec, encOut, seqLen), backwardConnect(enc, dec) -- with BOTH surfaces (see lua/encoders/lf-ques.lua):
-- dec.native = 'disc' for the model-level path, and dec:build / dec:forward({options, encOut}) / dec:backward(...) composed in Lua
-- from the operator-level ABI: the counterpart of decoders/disc.lua:3-32 of the reference.  The 100 weight-shared clones under
-- nn.Concat(2) ARE one batch of N*100 option sequences; embed -> x*Wx + b is a gather from the table Emb*Wx + b (exact: no dropout
-- on option embeddings, disc.lua:12-14); nn.MM + nn.Squeeze are folded into the criterion kernel (lua/model_ops.lua).
local decoderNet = {}

function decoderNet.model(params, enc)
    local dec = {native = 'disc', params = params, wordEmbed = enc.wordEmbed}      -- shares the encoder's embedding (disc.lua:12)

    function dec:declare(spec)
        local E, H = params.embedSize, params.rnnHiddenSize
        table.insert(spec, {'opt.W', (E + H) * 4 * H}); table.insert(spec, {'opt.b', 4 * H})
    end

    function dec:build(vdnn, fp, wordEmbed)
        self.vdnn, self.wordEmbed = vdnn, wordEmbed
        self.optionLSTM = vdnn.SeqLSTM(fp, 'opt', params.embedSize, params.rnnHiddenSize)
    end

    -- input = {options, encOut}: options = device int32 [To x N*O] time-major (+ .T, .N = N*O); returns the last option states
    -- [N*O x H] (the scores are formed by the criterion kernel together with the loss)
    function dec:forward(input)
        local vd, vdnn = self.vdnn.vd, self.vdnn
        local opts, l = input[1], self.optionLSTM
        local V, E, H, T, NO = params.vocabSize, params.embedSize, params.rnnHiddenSize, input[1].T, input[1].N
        self.table_ = vdnn.devFloats((V + 1) * 4 * H)
        vd.call('vd_gemm_nn', self.wordEmbed.weight, E, l.W, 4 * H, l.b, self.table_, 4 * H, V + 1, 4 * H, E, 0, nil)
        l.T, l.N = T, NO
        l.gates = vdnn.devFloats(T * NO * 4 * H); l.output = vdnn.devFloats(T * NO * H); l.cell = vdnn.devFloats(T * NO * H)
        vd.call('vd_lstm_forward', self.table_, 0, 4 * H, opts.tok, nil, l:Wh(), nil, nil, l.gates, l.output, l.cell, T, NO, H, 0, nil)
        self.output = l.output + (T - 1) * NO * H
        return self.output
    end

    -- gradOutput = {d optH [N*O x H], d encOut [N x H]} from the criterion; returns {nil, gradEncOut} like the reference's table of
    -- input gradients (model.lua:335-337 uses t[2])
    function dec:backward(input, gradOutput)
        local vd, vdnn = self.vdnn.vd, self.vdnn
        local opts, l = input[1], self.optionLSTM
        local V, E, H, T, NO = params.vocabSize, params.embedSize, params.rnnHiddenSize, input[1].T, input[1].N
        local dc = vdnn.devFloats(NO * H)
        vd.call('vd_lstm_backward', l:Wh(), l.gates, l.cell, nil, nil, gradOutput[1], nil, dc, nil, nil, nil, T, NO, H, 0, nil)
        if T > 1 then
            vd.call('vd_gemm_tn_acc', l.output, H, l.gates + NO * 4 * H, 4 * H, l.dW + E * 4 * H, 4 * H, H, 4 * H, (T - 1) * NO, 0, nil)
        end
        -- gradient of the gathered table: counting sort of the tokens + segmented row sum of da, then its three consumers
        local ffi = require 'ffi'
        local offs, work, perm = vdnn.devBytes((V + 2) * 4), vdnn.devBytes(2 * (V + 1) * 4), vdnn.devBytes(T * NO * 4)
        local dtab = vdnn.devFloats((V + 1) * 4 * H)
        vd.call('vd_token_sort', opts.tok, T * NO, V + 1, ffi.cast('int32_t*', offs), ffi.cast('int32_t*', work), ffi.cast('int32_t*', perm), nil)
        vd.call('vd_segment_rowsum_acc', l.gates, 4 * H, opts.tok, ffi.cast('const int32_t*', perm), T * NO, 4 * H, dtab, 4 * H, nil)
        vd.call('vd_colsum_acc', dtab, 4 * H, V + 1, 4 * H, l.db, nil)
        vd.call('vd_gemm_tn_acc', self.wordEmbed.weight, E, dtab, 4 * H, l.dW, 4 * H, E, 4 * H, V + 1, 0, nil)
        vd.call('vd_gemm_nt', dtab, 4 * H, l.W, 4 * H, nil, self.wordEmbed.gradWeight, E, V + 1, E, 4 * H, vd.C.VD_ACT_NONE, 2, nil)   -- dEmb += dTable * Wx^T
        return {nil, gradOutput[2]}
    end

    return dec
end

-- the state hand-off between encoder and decoder: no-ops for disc (disc.lua:35,38)
function decoderNet.forwardConnect(enc, dec, encOut, seqLen) end
function decoderNet.backwardConnect(enc, dec) end

return decoderNet
