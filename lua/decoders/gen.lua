-- decoders/gen.lua -- the reference's plug-in file contract (model.lua:22-29): returns a table with model(params, enc),
-- forwardConnect(enc, dec, encOut, seqLen), backwardConnect(enc, dec) and decoderConnect(dec) -- with BOTH surfaces (see
-- lua/encoders/lf-ques.lua): dec.native = 'gen' for the model-level path (the whole step in the library), and dec:declare / :build /
-- :forward(answerIn) / :backward(answerIn, gradOutput) composed in Lua from module objects over the operator-level ABI (lua/vdnn.lua):
-- the counterpart of decoders/gen.lua:3-68 of the reference --
--   shared wordEmbed -> numLayers x SeqLSTM:maskZero() -> Linear(H, V) [-> LogSoftMax, fused into the criterion kernel: lua/model_ops.lua]
-- and the three connect functions with the reference's own field names (userPrevOutput / userPrevCell, userNextGradCell /
-- gradPrevOutput, userGradPrevOutput / userGradPrevCell on enc.rnnLayers[i] / dec.rnnLayers[i]).  With lua/encoders/lf-ques.lua this
-- is BASELINE.json configs[0], the reference's CPU-runnable pair.
-- EXECUTED by the tests: tests/luavm (a Lua 5.1 evaluator with a LuaJIT-style ffi and a Torch7 tensor stub) runs this file against the real
-- library on the GPU -- loss, every gradient tensor and the post-Adam parameters against the library's model-level path, the fp64 oracle and the
-- golden fixtures (tests/test_lua_host_gpu.py) -- and against a bounds-checking dry library on the CPU (tests/test_luavm_cpu.py).
local decoderNet = {}

function decoderNet.model(params, enc)
    local dec = {native = 'gen', params = params, wordEmbed = enc.wordEmbed}      -- shares the encoder's embedding (gen.lua:10)

    function dec:declare(spec)
        local E, H, V = params.embedSize, params.rnnHiddenSize, params.vocabSize
        for layer = 1, params.numLayers do                                          -- gen.lua:17-22
            local D = (layer == 1) and E or H
            table.insert(spec, {'dec' .. layer .. '.W', (D + H) * 4 * H})
            table.insert(spec, {'dec' .. layer .. '.b', 4 * H})
        end
        table.insert(spec, {'vocab.W', V * H}); table.insert(spec, {'vocab.b', V})  -- gen.lua:23
    end

    function dec:build(vdnn, fp, wordEmbed)
        local E, H = params.embedSize, params.rnnHiddenSize
        self.vdnn, self.fp, self.wordEmbed, self.rnnLayers = vdnn, fp, wordEmbed, {}
        for layer = 1, params.numLayers do
            self.rnnLayers[layer] = vdnn.SeqLSTM(fp, 'dec' .. layer, (layer == 1) and E or H, H)
        end
    end

    -- answerIn = {tok = device int32 [Ta x N] time-major (<START> + tokens, 0 = pad), T, N}; returns the logits [Ta*N x Vp]
    -- (Vp = V rounded up to 4; the criterion kernel applies LogSoftMax and the masked NLL in place)
    function dec:forward(answerIn)
        local vd, vdnn = self.vdnn.vd, self.vdnn
        local H, V, T, N = params.rnnHiddenSize, params.vocabSize, answerIn.T, answerIn.N
        local rows, Vp = T * N, math.floor((V + 3) / 4) * 4
        local x = self.wordEmbed:forward(answerIn.tok, rows)
        for layer = 1, #self.rnnLayers do x = self.rnnLayers[layer]:forward(x, T, N, answerIn.tok) end
        self.rows, self.Vp = rows, Vp
        local Wv, _ = self.fp:view('vocab.W'); local bv, _ = self.fp:view('vocab.b')
        self.output = vdnn.devFloats(rows * Vp)
        vd.call('vd_gemm_nt', x, H, Wv, H, bv, self.output, Vp, rows, V, H, vd.C.VD_ACT_NONE, 0, nil)
        return self.output
    end

    -- gradOutput = d loss / d logits [Ta*N x Vp] (written in place of the logits by the criterion kernel)
    function dec:backward(answerIn, gradOutput)
        local vd, vdnn = self.vdnn.vd, self.vdnn
        local H, V, E, T, N = params.rnnHiddenSize, params.vocabSize, params.embedSize, answerIn.T, answerIn.N
        local rows, Vp = self.rows, self.Vp
        local L = #self.rnnLayers
        local Wv, dWv = self.fp:view('vocab.W'); local _, dbv = self.fp:view('vocab.b')
        local dech = self.rnnLayers[L].output
        vd.call('vd_gemm_tn_acc', gradOutput, Vp, dech, H, dWv, H, V, H, rows, 0, nil)
        vd.call('vd_colsum_acc', gradOutput, Vp, rows, V, dbv, nil)
        local dh = vdnn.devFloats(rows * H)
        vd.call('vd_gemm_nn', gradOutput, Vp, Wv, H, nil, dh, H, rows, H, V, 0, nil)
        local dSeq = dh
        for layer = L, 1, -1 do dSeq = self.rnnLayers[layer]:backward(dSeq, nil, true) end
        self.wordEmbed:backward(answerIn.tok, rows, dSeq)
    end

    return dec
end

-- gen.lua:30-42: decoder layer i starts from the encoder layer's final (h, c); the top layer's h from the encoder output.  An encoder
-- without enc.rnnLayers (the four nngraph encoders) hands over encOut alone: the top decoder layer starts from (encOut, c = 0).
-- (Lua-composed objects only; on the model-level path the hand-off happens inside the library's step and these are no-ops.)
function decoderNet.forwardConnect(enc, dec, encOut, seqLen)
    if dec.rnnLayers == nil then return end
    local H = dec.params.rnnHiddenSize
    if enc.rnnLayers ~= nil then
        local n = #enc.rnnLayers
        for ii = 1, n do
            local l = enc.rnnLayers[ii]
            dec.rnnLayers[ii].userPrevOutput = l.output + (seqLen - 1) * l.N * H       -- enc.rnnLayers[ii].output[seqLen]
            dec.rnnLayers[ii].userPrevCell = l.cell + (seqLen - 1) * l.N * H           -- enc.rnnLayers[ii].cell[seqLen]
        end
        dec.rnnLayers[n].userPrevOutput = encOut
    else
        dec.rnnLayers[#dec.rnnLayers].userPrevOutput = encOut
    end
end

-- gen.lua:45-60: cell / hidden gradients back into the encoder layers; returns d loss / d encOut
function decoderNet.backwardConnect(enc, dec)
    if dec.rnnLayers == nil then return nil end
    if enc.rnnLayers ~= nil then
        local n = #dec.rnnLayers
        for ii = 1, n do
            enc.rnnLayers[ii].userNextGradCell = dec.rnnLayers[ii].userGradPrevCell
            if ii ~= n then enc.rnnLayers[ii].gradPrevOutput = dec.rnnLayers[ii].userGradPrevOutput end
        end
        return dec.rnnLayers[#enc.rnnLayers].userGradPrevOutput
    end
    return dec.rnnLayers[#dec.rnnLayers].userGradPrevOutput
end

-- gen.lua:63-68 (sampling: chain the decoder to itself, one step at a time)
function decoderNet.decoderConnect(dec)
    if dec.rnnLayers == nil then return end
    for ii = 1, #dec.rnnLayers do
        dec.rnnLayers[ii].userPrevCell = dec.rnnLayers[ii].cell
        dec.rnnLayers[ii].userPrevOutput = dec.rnnLayers[ii].output
    end
end

return decoderNet
