-- encoders/mn-att-ques-im-hist.lua -- plug-in file contract of the reference (model.lua:19-25: the file is `dofile`d and must return a
-- table with model(params)).  Instead of building nn / nngraph modules it names the native graph; the object keeps
-- the fields decoders read: .wordEmbed (disc.lua:12, gen.lua:10) is the shared embedding, owned by the library.
local encoderNet = {}

function encoderNet.model(params)
    local enc = {native = 'mn-att-ques-im-hist', params = params}
    enc.wordEmbed = {shared = 'embed'}           -- one table for question / history / option / answer tokens
    return enc
end

return encoderNet
