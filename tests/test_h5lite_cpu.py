"""visdial_amd.h5lite (ctypes over libhdf5) against an HDF5 file written by h5py (tests/golden/make_h5_fixture.py)."""
import os

import numpy as np
import pytest

from visdial_amd import h5lite
from visdial_amd.dataloader import _open_arrays

HERE = os.path.dirname(os.path.abspath(__file__))
H5 = os.path.join(HERE, 'golden', 'tiny_visdial.h5')
NPZ = os.path.join(HERE, 'golden', 'tiny_visdial_expected.npz')

needs_lib = pytest.mark.skipif(not h5lite.available(), reason="no libhdf5 on this machine (the .npz twin path is used)")


@needs_lib
def test_reads_every_dataset_bit_exact():
    exp = np.load(NPZ)
    with h5lite.File(H5) as f:
        assert sorted(f.keys()) == sorted(exp.files)
        for k in exp.files:
            assert k in f
            got = f[k]
            assert got.dtype == exp[k].dtype and got.shape == exp[k].shape, k
            np.testing.assert_array_equal(got, exp[k])          # integers and floats alike: bit-exact
        assert 'nope' not in f
        with pytest.raises(KeyError):
            f['nope']


@needs_lib
def test_dataloader_opens_h5_without_h5py():
    f = _open_arrays(H5)
    np.testing.assert_array_equal(np.asarray(f['ques_train']), np.load(NPZ)['ques_train'])


def test_missing_file_is_loud():
    if not h5lite.available():
        pytest.skip("no libhdf5")
    with pytest.raises(OSError):
        h5lite.File(os.path.join(HERE, 'golden', 'does_not_exist.h5'))
