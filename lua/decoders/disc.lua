-- decoders/disc.lua -- plug-in file contract of the reference (model.lua:22-29): returns a table with model(params, enc),
-- forwardConnect(enc, dec, encOut, seqLen), backwardConnect(enc, dec).
local decoderNet = {}

function decoderNet.model(params, enc)
    return {native = 'disc', params = params, wordEmbed = enc.wordEmbed}      -- shares the encoder's embedding (disc.lua:12)
end

-- the state hand-off between encoder and decoder happens inside the library's step (disc.lua:35,38: no-ops)
function decoderNet.forwardConnect(enc, dec, encOut, seqLen) end
function decoderNet.backwardConnect(enc, dec) end

return decoderNet
