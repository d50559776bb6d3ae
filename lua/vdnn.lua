-- vdnn.lua -- module objects over the OPERATOR-LEVEL C ABI (include/visdial_hip.h): what `nn` / `rnn` are to the reference's
-- plug-in files (encoders/*.lua, decoders/*.lua build nn.LookupTableMaskZero, nn.SeqLSTM, nn.Linear, ... and model.lua calls
-- :forward / :backward on them), for a host whose arithmetic lives in libvisdial_hip.so.  Every method is a handful of ABI
-- calls; there is no arithmetic in Lua.  A plug-in file written with these modules is lua/encoders/lf-ques.lua (with
-- lua/decoders/disc.lua and lua/model_ops.lua = the operator-level Model).
--
-- EXECUTED by the tests: tests/luavm (a Lua 5.1 evaluator with a LuaJIT-style ffi and a Torch7 tensor stub) runs this file against the real
-- library on the GPU -- loss, every gradient tensor and the post-Adam parameters against the library's model-level path, the fp64 oracle and the
-- golden fixtures (tests/test_lua_host_gpu.py) -- and against a bounds-checking dry library on the CPU (tests/test_luavm_cpu.py).
local ffi = require 'ffi'
local vd = dofile('visdial_ffi.lua')

local M = {}

-- ---- device memory (the host has no CUDA tensor type: vd_malloc, vd_memcpy_*) -------------------------------------------------
-- Activations are STEP-SCOPED: every buffer a module allocates during forward / backward is recorded in the current arena and freed by
-- M.releaseStep(), which the Model calls at the top of each forwardBackward (the reference gets the same lifetime from Torch's
-- allocator reusing module.output / gradInput across iterations).  Parameters, gradients and Adam moments are allocated under
-- M.persistent(fn) and live as long as the model.
local arena, keepAlive = {}, false

local function dmalloc(bytes)
    local p = ffi.new('void*[1]')
    vd.call('vd_malloc', p, bytes)
    if not keepAlive then arena[#arena + 1] = p[0] end
    return p[0]
end

function M.persistent(fn)
    local before = keepAlive
    keepAlive = true
    local r = fn()
    keepAlive = before
    return r
end

function M.releaseStep()
    vd.call('vd_stream_synchronize', nil)                 -- nothing in flight may still read them
    for i = 1, #arena do vd.call('vd_free', arena[i]) end
    arena = {}
end

function M.devFloats(n)                                   -- torch.CudaTensor(n):zero()
    local bytes = math.max(n, 4) * 4
    local p = dmalloc(bytes)
    vd.call('vd_memset', p, 0, bytes, nil)
    return ffi.cast('float*', p)
end

function M.devInts(intTensor)                             -- IntTensor (host, contiguous) -> device int32
    local n = intTensor:nElement()
    local p = dmalloc(n * 4)
    vd.call('vd_memcpy_h2d', p, intTensor:data(), n * 4, nil)
    return ffi.cast('int32_t*', p)
end

function M.devBytes(n)                                    -- raw device bytes (byte masks, sort scratch)
    return dmalloc(math.max(n, 4))
end

function M.devBytesFrom(byteTensor)                       -- ByteTensor (host, contiguous) -> device uint8
    local n = byteTensor:nElement()
    local p = M.devBytes(n)
    vd.call('vd_memcpy_h2d', p, byteTensor:data(), n, nil)
    return ffi.cast('uint8_t*', p)
end

-- ---- nn.Dropout(p): wrapper:training() / :evaluate() flip M.training (model.lua:57,111,144) -------------------------------------
-- :mask(n) draws this forward's noise (nil = identity: evaluate); :apply(x, mask, n) is the forward AND the backward of the node.
-- The fused kernels (embedding gather / scatter, image attention) take the mask pointer + the 1/(1-p) scale instead.
M.training = true
local Dropout = {}
Dropout.__index = Dropout
local dropSeed = 1234

function M.Dropout(p) return setmetatable({p = p, scale = 1.0 / (1.0 - p)}, Dropout) end

-- site: the name of the Dropout node (q_emb / h_emb / hatt / img_tr / iqc / u of the nngraph encoders, fuse of lf-*, img of hrea --
-- the same names vd_model_set_dropout_mask takes).  M.pinMasks{site = ByteTensor keep-mask, ...} makes the NEXT forward passes use
-- the given noise instead of drawing it (parity runs against a CPU restatement); M.pinMasks(nil) returns to drawing.
local pinned = nil
function M.pinMasks(t) pinned = t end

function Dropout:mask(n, site)
    if not M.training then return nil end
    if pinned ~= nil and site ~= nil and pinned[site] ~= nil then
        local keep = pinned[site]:byte():contiguous()
        assert(keep:nElement() == n, string.format('pinned Dropout mask %s has %d elements, the node draws %d', site, keep:nElement(), n))
        return M.devBytesFrom(keep)
    end
    local m = ffi.cast('uint8_t*', M.devBytes(n))
    vd.call('vd_dropout_mask', m, n, dropSeed, self.p, nil)
    dropSeed = dropSeed + 1
    return m
end

function Dropout:apply(x, mask, n)
    if mask == nil then return x end
    local y = M.devFloats(n)
    vd.call('vd_dropout_apply', x, mask, y, n, self.scale, nil)
    return y
end

local function align4(n) return math.floor((n + 3) / 4) * 4 end

-- ---- wrapper:getParameters() (model.lua:55): flat W / dW (+ Adam m, v), every tensor 16-byte aligned ---------------------------
-- spec = { {name, numel}, ... } in module order; returns an object with .W .dW .m .v (float*), .numel, :view(name) -> W, dW
local FlatParams = {}
FlatParams.__index = FlatParams

function M.FlatParams(spec)
    local self = setmetatable({off = {}, size = {}, order = {}}, FlatParams)
    local o = 0
    for _, e in ipairs(spec) do
        self.off[e[1]] = o; self.size[e[1]] = e[2]; table.insert(self.order, e[1])
        o = o + align4(e[2])
    end
    self.numel = o
    M.persistent(function()
        self.W = M.devFloats(o); self.dW = M.devFloats(o); self.m = M.devFloats(o); self.v = M.devFloats(o)
    end)
    return self
end

function FlatParams:view(name) return self.W + self.off[name], self.dW + self.off[name] end

function FlatParams:zeroGrad() vd.call('vd_memset', self.dW, 0, self.numel * 4, nil) end       -- wrapper:zeroGradParameters()

-- wrapperW:copy(flat FloatTensor in getParameters() order, tensors back to back) / wrapperW:float()
function FlatParams:copyFrom(flat)
    flat = flat:float():contiguous()
    local src = 0
    for _, name in ipairs(self.order) do
        vd.call('vd_memcpy_h2d', self.W + self.off[name], flat:data() + src, self.size[name] * 4, nil)
        src = src + self.size[name]
    end
end

function FlatParams:toFloat(which)
    local total = 0
    for _, name in ipairs(self.order) do total = total + self.size[name] end
    local flat, dst = torch.FloatTensor(total), 0
    vd.call('vd_stream_synchronize', nil)
    for _, name in ipairs(self.order) do
        vd.call('vd_memcpy_d2h', flat:data() + dst, (which == 'dW' and self.dW or self.W) + self.off[name], self.size[name] * 4, nil)
        dst = dst + self.size[name]
    end
    return flat
end

-- ---- nn.LookupTableMaskZero(V, E) (encoders/lf-ques.lua:12): table [(V+1) x E], row 0 = pad ------------------------------------
local Lookup = {}
Lookup.__index = Lookup

function M.LookupTableMaskZero(fp, name, V, E)
    local W, dW = fp:view(name)
    return setmetatable({weight = W, gradWeight = dW, V = V, E = E}, Lookup)
end

function Lookup:zeroPad() vd.call('vd_memset', self.weight, 0, self.E * 4, nil) end         -- the pad row is re-zeroed on every forward

-- tok: device int32 [rows]; returns [rows x E].  mask / scale: the nn.Dropout that follows the table in the nngraph encoders
-- (mn-att:24-25), fused into the gather (nil = none)
function Lookup:forward(tok, rows, mask, scale)
    local out = M.devFloats(rows * self.E)
    vd.call('vd_embed_gather', self.weight, tok, mask, out, rows, self.E, mask ~= nil and scale or 1.0, nil)
    return out
end

function Lookup:backward(tok, rows, dx, mask, scale)
    vd.call('vd_embed_scatter_acc', self.gradWeight, tok, mask, dx, rows, self.E, mask ~= nil and scale or 1.0, nil)
end

-- ---- nn.SeqLSTM(D, H):maskZero() (encoders/lf-ques.lua:18-24, decoders/disc.lua:4): W = [Wx ; Wh] [(D+H) x 4H], gates i,f,o,g --
local SeqLSTM = {}
SeqLSTM.__index = SeqLSTM

function M.SeqLSTM(fp, name, D, H)
    local W, dW = fp:view(name .. '.W')
    local b, db = fp:view(name .. '.b')
    return setmetatable({D = D, H = H, W = W, b = b, dW = dW, db = db}, SeqLSTM)
end

function SeqLSTM:Wh() return self.W + self.D * 4 * self.H end

-- The state hand-off fields of Element-Research's SeqLSTM that decoders/gen.lua:30-60 reads and writes (all device float* [N x H], nil = none):
--   .userPrevOutput / .userPrevCell        initial h / c of the NEXT forward (consumed by it)
--   .gradPrevOutput / .userNextGradCell    extra gradient into the last step's h / c of the NEXT backward (consumed by it)
--   .userGradPrevOutput / .userGradPrevCell  out: gradients w.r.t. the initial h / c of the last forward
-- x: [T*N x D] rows (time-major); tokMask: device int32 [T x N] or nil (maskZero); .output = h [T x N x H], .cell = c
function SeqLSTM:forward(x, T, N, tokMask)
    local H = self.H
    self.x, self.T, self.N = x, T, N
    self.gates = M.devFloats(T * N * 4 * H); self.output = M.devFloats(T * N * H); self.cell = M.devFloats(T * N * H)
    -- hoisted input projection x*Wx + b straight into the gates buffer, then the recurrence in place
    vd.call('vd_gemm_nn', x, self.D, self.W, 4 * H, self.b, self.gates, 4 * H, T * N, 4 * H, self.D, 0, nil)
    self.h0, self.c0 = self.userPrevOutput, self.userPrevCell          -- consumed once, like the reference's module
    self.userPrevOutput, self.userPrevCell = nil, nil
    if self.h0 ~= nil and self.c0 == nil then self.c0 = M.devFloats(N * H) end     -- userPrevOutput alone (gen.lua:40): the cell starts at 0
    assert((self.h0 == nil) == (self.c0 == nil), 'SeqLSTM: userPrevCell needs userPrevOutput')
    vd.call('vd_lstm_forward', self.gates, N * 4 * H, 4 * H, nil, tokMask, self:Wh(), self.h0, self.c0, self.gates, self.output, self.cell,
            T, N, H, 0, nil)
    return self.output
end

-- dhSeq [T x N x H] or nil, dhLast [N x H] or nil; accumulates gradWeight / gradBias; returns dx [T*N x D] (or nil)
function SeqLSTM:backward(dhSeq, dhLast, needDx)
    local H, T, N = self.H, self.T, self.N
    local dc = M.devFloats(N * H)
    if self.gradPrevOutput ~= nil then                                   -- gen.lua:49-51: the decoder's gradient w.r.t. this layer's final h
        if dhLast == nil then dhLast = self.gradPrevOutput
        else
            local t = M.devFloats(N * H)
            vd.call('vd_axpby', dhLast, self.gradPrevOutput, t, N * H, 1.0, 1.0, nil)
            dhLast = t
        end
    end
    local dcLast = self.userNextGradCell
    self.gradPrevOutput, self.userNextGradCell = nil, nil
    local dh0 = nil
    if self.h0 ~= nil then dh0 = M.devFloats(N * H) end
    vd.call('vd_lstm_backward', self:Wh(), self.gates, self.cell, self.c0, dhSeq, dhLast, dcLast, dc, dh0, nil, nil, T, N, H, 0, nil)
    self.userGradPrevOutput = dh0
    self.userGradPrevCell = (self.h0 ~= nil) and dc or nil
    local dWh = self.dW + self.D * 4 * H                                   -- da now lives in self.gates
    if T > 1 then vd.call('vd_gemm_tn_acc', self.output, H, self.gates + N * 4 * H, 4 * H, dWh, 4 * H, H, 4 * H, (T - 1) * N, 0, nil) end
    if self.h0 ~= nil then vd.call('vd_gemm_tn_acc', self.h0, H, self.gates, 4 * H, dWh, 4 * H, H, 4 * H, N, 0, nil) end   -- step 0 multiplied the initial state
    vd.call('vd_colsum_acc', self.gates, 4 * H, T * N, 4 * H, self.db, nil)
    vd.call('vd_gemm_tn_acc', self.x, self.D, self.gates, 4 * H, self.dW, 4 * H, self.D, 4 * H, T * N, 0, nil)
    if not needDx then return nil end
    local dx = M.devFloats(T * N * self.D)
    vd.call('vd_gemm_nt', self.gates, 4 * H, self.W, 4 * H, nil, dx, self.D, T * N, self.D, 4 * H, vd.C.VD_ACT_NONE, 0, nil)   -- da * Wx^T
    return dx
end

-- ---- nn.Linear(nIn, nOut) [+ nn.Tanh] (encoders/lf-ques.lua:29-31; mn-att:64-65,77,88,106) -------------------------------------
local LinearTanh = {}
LinearTanh.__index = LinearTanh

function M.LinearTanh(fp, name, nIn, nOut, noTanh)          -- noTanh = true: plain nn.Linear (mn-att:88 ques_common)
    local W, dW = fp:view(name .. '.W')
    local b, db = fp:view(name .. '.b')
    return setmetatable({nIn = nIn, nOut = nOut, W = W, b = b, dW = dW, db = db, noTanh = noTanh or false}, LinearTanh)
end

function M.Linear(fp, name, nIn, nOut) return M.LinearTanh(fp, name, nIn, nOut, true) end

function LinearTanh:forward(x, rows)
    self.x, self.rows = x, rows
    self.output = M.devFloats(rows * self.nOut)
    vd.call('vd_gemm_nt', x, self.nIn, self.W, self.nIn, self.b, self.output, self.nOut, rows, self.nOut, self.nIn,
            self.noTanh and vd.C.VD_ACT_NONE or vd.C.VD_ACT_TANH, 0, nil)
    return self.output
end

-- returns dx [rows x nIn] (nil when needDx == false); accumulates gradWeight / gradBias
function LinearTanh:backward(dy, needDx)
    local rows = self.rows
    local dpre = dy
    if not self.noTanh then
        dpre = M.devFloats(rows * self.nOut)
        vd.call('vd_tanh_backward', dy, self.output, dpre, rows * self.nOut, nil)
    end
    vd.call('vd_gemm_tn_acc', dpre, self.nOut, self.x, self.nIn, self.dW, self.nIn, self.nOut, self.nIn, rows, 0, nil)
    vd.call('vd_colsum_acc', dpre, self.nOut, rows, self.nOut, self.db, nil)
    if needDx == false then return nil end
    local dx = M.devFloats(rows * self.nIn)
    vd.call('vd_gemm_nn', dpre, self.nOut, self.W, self.nIn, nil, dx, self.nIn, rows, self.nIn, self.nOut, 0, nil)
    return dx
end

M.vd = vd
return M
