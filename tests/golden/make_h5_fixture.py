"""Writes tests/golden/tiny_visdial.h5 with h5py (run with an interpreter that has it, e.g.
/opt/conda/bin/python3.9) plus the same arrays as tiny_visdial_expected.npz.  Dataset names, dtypes and
layouts follow what data/prepro.py:263-297 writes (uint32 token / length / index matrices) and what the image
scripts write (float32 features); sizes are tiny.  The fixture pins visdial_amd/h5lite.py (ctypes over libhdf5)
against a file produced by an independent HDF5 writer."""
import os

import h5py
import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
rng = np.random.RandomState(7)
arrays = {
    'ques_train': rng.randint(0, 50, size=(3, 10, 20)).astype(np.uint32),
    'ques_length_train': rng.randint(1, 21, size=(3, 10)).astype(np.uint32),
    'ans_train': rng.randint(0, 50, size=(3, 10, 20)).astype(np.uint32),
    'ans_length_train': rng.randint(1, 21, size=(3, 10)).astype(np.uint32),
    'ans_index_train': rng.randint(0, 100, size=(3, 10)).astype(np.uint32),
    'opt_train': rng.randint(1, 301, size=(3, 10, 100)).astype(np.uint32),
    'opt_list_train': rng.randint(0, 50, size=(300, 20)).astype(np.uint32),
    'opt_length_train': rng.randint(1, 21, size=(300,)).astype(np.uint32),
    'cap_train': rng.randint(0, 50, size=(3, 40)).astype(np.uint32),
    'cap_length_train': rng.randint(1, 41, size=(3,)).astype(np.uint32),
    'img_pos_train': np.arange(3).astype(np.int64),
    'images_train': rng.randn(3, 8, 2, 2).astype(np.float32),
    'scalar_like': np.array([3.5], dtype=np.float64),
    'small_i16': np.array([-3, 7, 9], dtype=np.int16),
}
with h5py.File(os.path.join(here, 'tiny_visdial.h5'), 'w') as f:
    for k, v in arrays.items():
        if k == 'images_train':
            f.create_dataset(k, data=v, chunks=(1, 8, 2, 2), compression='gzip')    # chunked + deflate, as h5py users often write
        else:
            f.create_dataset(k, data=v)
np.savez(os.path.join(here, 'tiny_visdial_expected.npz'), **arrays)
print({k: (v.shape, str(v.dtype)) for k, v in arrays.items()})
