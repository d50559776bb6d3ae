-- encoders/mn-ques-im-hist.lua -- plug-in file contract of the reference (model.lua:19-25: the file is `dofile`d and must return a
-- table with model(params)).  Instead of building nn / nngraph modules it names the native graph; the object keeps
-- the fields decoders read: .wordEmbed (disc.lua:12, gen.lua:10) is the shared embedding, owned by the library.
local encoderNet = {}

function encoderNet.model(params)
    local enc = {native = 'mn-ques-im-hist', params = params}
    enc.wordEmbed = {shared = 'embed'}           -- one table for question / history / option / answer tokens
    -- the model-level runtime (csrc/runtime.hip) covers mn-att-ques-im-hist + disc so far; this encoder runs through the
    -- operator-level entry points (host: visdial_amd/encoders/mn_ques_im_hist.py) -- vd_model_create reports it
    return enc
end

return encoderNet
