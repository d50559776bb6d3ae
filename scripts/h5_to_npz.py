"""Convert the reference's HDF5 inputs (data/visdial_data.h5, data/data_img*.h5 -- written by data/prepro.py
and data/prepro_img_*.lua) into .npz twins with identical dataset names, for interpreters without h5py
(visdial_amd.dataloader._open_arrays falls back to <name>.npz).  Run with any python that has h5py."""
import sys

import h5py
import numpy as np

for path in sys.argv[1:]:
    with h5py.File(path, 'r') as f:
        arrays = {k: np.asarray(f[k]) for k in f.keys()}
    out = path[:-3] + '.npz' if path.endswith('.h5') else path + '.npz'
    np.savez(out, **arrays)
    print(path, '->', out, {k: v.shape for k, v in arrays.items()})
