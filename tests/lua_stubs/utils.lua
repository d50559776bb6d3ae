-- utils.lua -- TEST STAND-IN for the reference's utils.lua, which lua/model.lua `dofile`s when it is overlaid on a reference checkout
-- (processRanks for Model:retrieve, idToWords for Model:generateAnswers).  Written for the luavm tests; not a copy: only the two
-- functions the Lua host calls, with the reference's contracts (utils.lua:71-83 idToWords, :131-160 processRanks).
local utils = {}

-- ranks: DoubleTensor [numDialogs x numRounds] of 1-based ground-truth ranks -> prints and returns R@1/5/10, mean rank, MRR
function utils.processRanks(ranks)
    local flat = ranks:double():view(-1)
    local n = flat:nElement()
    local r1, r5, r10, mean, mrr = 0, 0, 0, 0, 0
    for i = 1, n do
        local r = flat[i]
        if r <= 1 then r1 = r1 + 1 end
        if r <= 5 then r5 = r5 + 1 end
        if r <= 10 then r10 = r10 + 1 end
        mean = mean + r; mrr = mrr + 1 / r
    end
    local out = {r1 = r1 / n, r5 = r5 / n, r10 = r10 / n, meanR = mean / n, mrr = mrr / n}
    print(string.format('\tR@1: %.4f\tR@5: %.4f\tR@10: %.4f\tmeanR: %.4f\tMRR: %.4f', out.r1, out.r5, out.r10, out.meanR, out.mrr))
    utils.lastRanks = out
    return out
end

-- ids: LongTensor of word ids (0 = pad) -> the sentence
function utils.idToWords(ids, ind2word)
    local words = {}
    for i = 1, ids:nElement() do
        local id = ids[i]
        if id > 0 then words[#words + 1] = ind2word[id] or ind2word[tostring(id)] or '<UNK>' end
    end
    return table.concat(words, ' ')
end

return utils
