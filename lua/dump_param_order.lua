-- Escape hatch for the checkpoint layout (visdial_amd/t7.py, lua/model.lua:Model:tensors()): run under a REAL Torch7, from the
-- reference checkout, on any checkpoint train.lua wrote:      th /path/to/lua/dump_param_order.lua checkpoint.t7 > order.json
-- Prints where wrapper:getParameters() (model.lua:41-54) put the weight / bias of every parameterised module, in flat order
-- (0-based offsets; shared storages once).  `-paramOrder order.json` on train.py / evaluate.py / generate.py then checks this repo's
-- DERIVED order for the nngraph encoders against it (and follows the file where the two disagree).
require 'nn'; require 'nngraph'; require 'rnn'
torch.setdefaulttensortype('torch.FloatTensor')
local saved = torch.load(assert(arg[1], 'usage: th dump_param_order.lua checkpoint.t7'))
local params = saved.modelParams
params.gpuid = -1
require 'model'
local model = Model(params)
local rows, seen = {}, {}
for k, m in ipairs(model.wrapper:listModules()) do
    for _, field in ipairs({'weight', 'bias'}) do
        local t = m[field]
        if t and torch.isTensor(t) and t:nElement() > 0 and not seen[t:storageOffset()] then
            seen[t:storageOffset()] = true
            rows[#rows + 1] = {t:storageOffset() - 1, string.format(
                '{"module": %d, "type": "%s", "field": "%s", "offset": %d, "numel": %d, "rows": %d}',
                k, torch.type(m), field, t:storageOffset() - 1, t:nElement(), t:size(1))}
        end
    end
end
table.sort(rows, function(a, b) return a[1] < b[1] end)
for i, r in ipairs(rows) do rows[i] = r[2] end
print('{"encoder": "' .. params.encoder .. '", "decoder": "' .. params.decoder .. '", "total": ' .. model.wrapperW:nElement() ..
      ', "tensors": [\n' .. table.concat(rows, ',\n') .. '\n]}')
