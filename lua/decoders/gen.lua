-- decoders/gen.lua -- plug-in file contract of the reference (model.lua:22-29): returns a table with model(params, enc),
-- forwardConnect(enc, dec, encOut, seqLen), backwardConnect(enc, dec) and decoderConnect(dec).
local decoderNet = {}

function decoderNet.model(params, enc)
    return {native = 'gen', params = params, wordEmbed = enc.wordEmbed}      -- shares the encoder's embedding (gen.lua:10)
end

-- the state hand-off between encoder and decoder happens inside the library's step (gen.lua:30-60)
function decoderNet.forwardConnect(enc, dec, encOut, seqLen) end
function decoderNet.backwardConnect(enc, dec) end
function decoderNet.decoderConnect(dec) end

return decoderNet
