"""Command-line options: same flag names and defaults as the reference's opts.lua:5-40, plus the
derived capability flags it infers from the encoder NAME (opts.lua:54-67)."""
import argparse
import time

# (flag, default, help) -- opts.lua:5-40
OPTIONS = [
    ('inputImg', 'data/data_img.h5', 'HDF5 file with image features'),
    ('inputQues', 'data/visdial_data.h5', 'HDF5 file with preprocessed questions'),
    ('inputJson', 'data/visdial_params.json', 'JSON file with info and vocab'),
    ('savePath', 'checkpoints/', 'Path to save checkpoints'),
    ('saveIter', 2, 'Save model checkpoint after every saveIter epochs'),
    ('encoder', 'lf-ques-hist', 'Name of the encoder to use'),
    ('decoder', 'gen', 'Name of the decoder to use (gen/disc)'),
    ('imgNorm', 1, 'normalize the image feature. 1=yes, 0=no'),
    ('imgEmbedSize', 300, 'Size of the multimodal embedding'),
    ('imgFeatureSize', 4096, 'Channel size of the image feature'),
    ('imgSpatialSize', 14, 'Spatial size of image features (for attention-based encoders)'),
    ('embedSize', 300, 'Size of input word embeddings'),
    ('rnnHiddenSize', 512, 'Size of the LSTM state'),
    ('maxHistoryLen', 60, 'Maximum history to consider when using concatenated QA pairs'),
    ('numLayers', 2, 'Number of layers in LSTM'),
    ('commonEmbeddingSize', 512, 'Common embedding size in MN-ATT-QIH'),
    ('numAttentionLayers', 1, 'No. of attention hops in MN-ATT-QIH'),
    ('loadPath', '', 'Checkpoint path to load from'),
    ('batchSize', 40, 'Batch size (number of threads)'),
    ('learningRate', 1e-3, 'Learning rate'),
    ('weightInit', 'xavier', 'Weight initialization strategy (accepted and ignored, as in the reference)'),
    ('dropout', 0.5, 'Dropout'),
    ('numEpochs', 100, 'Epochs'),
    ('LRateDecay', 10, 'unused (as in the reference)'),
    ('lrDecayRate', 0.9997592083, 'Decay for learning rate'),
    ('minLRate', 5e-5, 'Minimum learning rate'),
    ('gpuid', 0, 'GPU id to use'),
    ('backend', 'cudnn', 'accepted for CLI compatibility; ignored'),
    # not a reference flag: the arithmetic of the option recurrence.  DEFAULT = split9 (fp32-grade: every fp32 operand as the exact sum of three
    # bf16 values, all nine products, fp32 accumulate -- what bench.py's headline measures; it applies at throughput shapes, >= 2 048 option rows,
    # smaller recurrences run the fp32 MFMA either way)
    ('lstmPrecision', 'split9', "arithmetic of the option-LSTM recurrence GEMMs: 'split9' (default: exact 3-way bf16 split of both operands, 9 bf16 MFMAs per fp32 one, fp32-grade) | 'fp32' (v_mfma_f32_32x32x2_f32) | 'bf16' (BASELINE configs[4]: bf16 operands, fp32 accumulation; through the native host also the encoder's recurrent products and dense weight gradients) | 'split6' / 'split3' (fewer products: data only)"),
    ('saveFormat', 't7', "checkpoint files: 't7' = model_epoch_%d.t7 / model_final.t7 in the Torch7 binary format "
                         "(train.lua:99-102,120-121) | 'pt' = torch.save of the same three fields"),
    ('host', 'python', "which host drives the library: 'python' (operator-level C ABI, visdial_amd/model.py) | 'native' "
                       "(model-level C ABI, the calls lua/model.lua makes; visdial_amd/native.py)"),
]


def derive(opt):
    """opts.lua:54-67: capabilities come from substrings of the encoder file name."""
    enc = opt['encoder']
    opt['useHistory'] = 'hist' in enc
    opt['useIm'] = 'im' in enc
    opt['concatHistory'] = 'lf' in enc
    if 'att' in enc:
        if opt.get('inputImg') == 'data/data_img.h5':
            opt['inputImg'] = 'data/data_img_pool5.h5'
        opt['imgNorm'] = 0
    return opt


def default_params(**overrides):
    opt = {k: v for k, v, _ in OPTIONS}
    # sizes the reference copies from the dataloader (train.lua:55-59)
    opt.update(vocabSize=11322, maxQuesCount=10, maxQuesLen=20, maxAnsLen=20, numOptions=100)
    opt.update(overrides)
    return derive(opt)


def parse(argv=None):
    ap = argparse.ArgumentParser(description='Train the Visual Dialog model')
    for k, v, h in OPTIONS:
        ap.add_argument('-' + k, '--' + k, type=type(v), default=v, help=h)
    # synthetic-data knobs (no HDF5 offline)
    ap.add_argument('--vocabSize', type=int, default=11322)
    ap.add_argument('--numTrainThreads', type=int, default=2000)
    ap.add_argument('--maxIters', type=int, default=0, help='stop after this many iterations (0 = numEpochs)')
    ap.add_argument('-paramOrder', '--paramOrder', default='',
                    help="flat-vector layout of a .t7 given to -loadPath: '' (this repo's table of the reference's getParameters() order), "
                         "'declaration', or a JSON file: {\"order\": [names]} or the output of lua/dump_param_order.lua under Torch7")
    ap.add_argument('-synthetic', '--synthetic', type=int, default=0,
                    help='1 = allow resuming (-loadPath) on synthetic batches when no dataset files exist')
    opt = vars(ap.parse_args(argv))
    opt.update(maxQuesCount=10, maxQuesLen=20, maxAnsLen=20, numOptions=100)
    if opt['savePath'] == 'checkpoints/':
        t = time.localtime()
        opt['savePath'] = 'checkpoints/model-%d-%d-%d-%d:%d:%d-%s-%s/' % (
            t.tm_mon, t.tm_mday, t.tm_year, t.tm_hour, t.tm_min, t.tm_sec, opt['encoder'], opt['decoder'])
    return derive(opt)
