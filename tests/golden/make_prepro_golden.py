"""RUN the reference's own data/prepro.py (the real file, from /root/reference, `__main__` and all) on a small VisDial-v1.0-format
dialog set and keep what it writes -- `visdial_data.h5` + `visdial_params.json`, the input contract of rows a3 / f1 -- as
fixtures.  This is the one piece of the reference that can execute in this container unmodified (Python + h5py), so the fixture is the
reference's OUTPUT, not a restatement of it:

    /opt/conda/bin/python3.9 tests/golden/make_prepro_golden.py          (build container only: needs h5py, nltk, tqdm, /root/reference)
    python tests/golden/make_reference_dataloader_golden.py prepro       (then: the reference's dataloader.lua executed on the same file)

What runs: prepro.py:35-181 `tokenize_data` (word counts, round padding, the test split's missing answers), the vocabulary build
(:227-240, threshold + UNK), `encode_vocab` (:82-102), `create_data_mats` (:105-181: truncation at max_*_len, 1-based options and
ans_index, test-split options only where present, num_rounds), the uint32 HDF5 writer (:263-277), the image-id -> path table (:283-296).
Shims, all in THIS script, the reference file is untouched:
  * `numpy.int` (removed in numpy 1.24; prepro.py:108,113,128,129,172 use it) = int
  * `nltk.tokenize.word_tokenize` needs the punkt model, which the image does not hold (no network): replaced by a regex that splits
    words and punctuation.  Tokenisation is not part of the contract under test (the dialogs below are written pre-tokenised, blanks
    between tokens; prepro.py:51 appends '?' to a question, which the regex splits off as punkt would).
The image feature file is NOT prepro.py's product (prepro_img_*.lua, Torch7): `data_img.h5` is written here with h5py in the layout
dataloader.lua:59-70 reads (`images_<split>` float32 [n x F]).
Outputs (tests/golden/prepro/): the three input JSONs, visdial_data.h5, visdial_params.json, data_img.h5, and expected.npz = every dataset
read back with h5py (the independent reader h5lite is compared with)."""
import json
import os
import re
import runpy
import shutil
import sys
import tempfile

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, 'prepro')
REF_PREPRO = '/root/reference/data/prepro.py'

WORDS = ('is the a man woman dog cat wearing hat red blue green what color yes no it on in table street two three people any are there '
         'sunny day night can you see tree car big small looks like maybe not sure very bright holding umbrella frisbee').split()
RARE = 'zebra giraffe kayak saxophone pergola'.split()          # below the count threshold -> UNK


def sentence(rng, lo, hi, rare=0.04):
    n = int(rng.randint(lo, hi + 1))
    return ' '.join(str(rng.choice(RARE)) if rng.rand() < rare else str(rng.choice(WORDS)) for _ in range(n))


def make_split(rng, split, n_dialogs, n_q, n_a, first_image):
    questions = [sentence(rng, 2, 11) for _ in range(n_q)]       # up to 11 words + '?': longer than -max_ques_len 8 (truncation)
    answers = [sentence(rng, 1, 8) for _ in range(n_a)]          # longer than -max_ans_len 6
    answers[3] = ''                                              # an empty answer / option (length 0)
    dialogs = []
    for i in range(n_dialogs):
        rounds = 10 if split != 'test' else int(rng.randint(1, 11))
        dlg = []
        for j in range(rounds):
            r = {'question': int(rng.randint(n_q))}
            last = j == rounds - 1
            if split != 'test' or not last:
                r['answer'] = int(rng.randint(n_a))
            if split != 'test' or last:                          # v1.0 test: options only for the round to be answered, no gt_index
                r['answer_options'] = [int(x) for x in rng.permutation(n_a)[:100]]
            if split != 'test':
                r['gt_index'] = int(rng.randint(100))
                r['answer_options'][r['gt_index']] = r['answer']
            dlg.append(r)
        dialogs.append({'image_id': first_image + i, 'caption': sentence(rng, 4, 20), 'dialog': dlg})   # longer than -max_cap_len 16 (which dataloader.lua:236 needs >= max_ques_len + max_ans_len)
    return {'version': '1.0', 'split': split, 'data': {'dialogs': dialogs, 'questions': questions, 'answers': answers}}


def main():
    rng = np.random.RandomState(20)
    work = tempfile.mkdtemp()
    splits = {'train': make_split(rng, 'train', 5, 30, 130, 1001), 'val': make_split(rng, 'val', 4, 25, 120, 2001),
              'test': make_split(rng, 'test', 3, 20, 110, 3001)}
    for s, d in splits.items():
        json.dump(d, open(os.path.join(work, 'visdial_1.0_%s.json' % s), 'w'))
    # image files: prepro.py:287-288 globs <image_root>/*/*.jpg and keys them by int(path[-12:-4])
    folder = {'train': 'train2014', 'val': 'VisualDialog_val2018', 'test': 'VisualDialog_test2018'}
    for s, d in splits.items():
        os.makedirs(os.path.join(work, 'images', folder[s]))
        for dlg in d['data']['dialogs']:
            pre = 'COCO_train2014_' if s == 'train' else folder[s] + '_'
            open(os.path.join(work, 'images', folder[s], '%s%012d.jpg' % (pre, dlg['image_id'])), 'w').close()

    # ---- shims (see the docstring), then the reference file itself as __main__
    if not hasattr(np, 'int'):
        np.int = int
    import nltk.tokenize
    nltk.tokenize.word_tokenize = lambda s: re.findall(r"\w+|[^\w\s]", s)
    argv = ['prepro.py', '-input_json_train', os.path.join(work, 'visdial_1.0_train.json'),
            '-input_json_val', os.path.join(work, 'visdial_1.0_val.json'), '-input_json_test', os.path.join(work, 'visdial_1.0_test.json'),
            '-image_root', os.path.join(work, 'images'), '-output_json', os.path.join(work, 'visdial_params.json'),
            '-output_h5', os.path.join(work, 'visdial_data.h5'), '-max_ques_len', '8', '-max_ans_len', '6', '-max_cap_len', '16',
            '-word_count_threshold', '3']
    old = sys.argv
    sys.argv = argv
    try:
        runpy.run_path(REF_PREPRO, run_name='__main__')
    finally:
        sys.argv = old

    # ---- the image features file (not prepro.py's; layout of dataloader.lua:59-70)
    F = 16
    with h5py.File(os.path.join(work, 'data_img.h5'), 'w') as f:
        for s, d in splits.items():
            f.create_dataset('images_' + s, data=rng.randn(len(d['data']['dialogs']), F).astype(np.float32))

    # ---- install the fixtures + an h5py read-back of every dataset
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    os.makedirs(OUT)
    for name in ['visdial_1.0_train.json', 'visdial_1.0_val.json', 'visdial_1.0_test.json', 'visdial_params.json', 'visdial_data.h5', 'data_img.h5']:
        shutil.copy(os.path.join(work, name), os.path.join(OUT, name))
    rec = {}
    for name, pre in (('visdial_data.h5', 'data.'), ('data_img.h5', 'img.')):
        with h5py.File(os.path.join(OUT, name), 'r') as f:
            for k in f.keys():
                rec[pre + k] = f[k][()]
    np.savez_compressed(os.path.join(OUT, 'expected.npz'), **rec)
    info = json.load(open(os.path.join(OUT, 'visdial_params.json')))
    print('vocabulary %d words (UNK = %d), datasets: %s' % (len(info['word2ind']), info['word2ind']['UNK'],
                                                           {k: (v.shape, str(v.dtype)) for k, v in sorted(rec.items())}))
    shutil.rmtree(work)


if __name__ == '__main__':
    main()
