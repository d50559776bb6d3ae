"""Minimal read-only HDF5 access over the libhdf5 C library (ctypes), for interpreters without h5py.

The reference's inputs are HDF5 files written by data/prepro.py (visdial_data.h5: uint32 token / length /
index matrices) and data/prepro_img_*.lua (data_img*.h5: float32 feature arrays), read in the reference by
torch-hdf5 (dataloader.lua:36-140).  Only what that path needs is bound: open a file, list the root group,
read a whole numeric dataset into a numpy array.

    f = h5lite.File('data/visdial_data.h5')
    'ques_train' in f ; f.keys() ; f['ques_train']  -> numpy array (native dtype of the dataset)

The library is looked up in $VD_HDF5_LIB, the dynamic loader's path, and the usual conda / distro locations;
`available()` says whether one was found (visdial_amd.dataloader falls back to .npz twins otherwise).
"""
import ctypes as C
import ctypes.util
import os

import numpy as np

_CANDIDATES = [
    os.environ.get('VD_HDF5_LIB'),
    ctypes.util.find_library('hdf5'),
    ctypes.util.find_library('hdf5_serial'),
    '/opt/conda/lib/libhdf5.so',
    '/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so',
    '/usr/lib/x86_64-linux-gnu/libhdf5_serial.so',
    '/usr/lib64/libhdf5.so',
    '/usr/local/lib/libhdf5.so',
]
_lib = None
_hid = C.c_int64          # hid_t is 64-bit since HDF5 1.10


def _load():
    global _lib, _hid
    if _lib is not None:
        return _lib
    last = None
    for cand in _CANDIDATES:
        if not cand:
            continue
        try:
            lib = C.CDLL(cand)
        except OSError as e:
            last = e
            continue
        lib.H5open()
        maj, mnr, rel = C.c_uint(), C.c_uint(), C.c_uint()
        lib.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel))
        if (maj.value, mnr.value) < (1, 10):
            _hid = C.c_int               # hid_t was a 32-bit int before 1.10
        hid = _hid
        lib.H5Fopen.restype, lib.H5Fopen.argtypes = hid, [C.c_char_p, C.c_uint, hid]
        lib.H5Fclose.argtypes = [hid]
        lib.H5Dopen2.restype, lib.H5Dopen2.argtypes = hid, [hid, C.c_char_p, hid]
        lib.H5Dclose.argtypes = [hid]
        lib.H5Dget_space.restype, lib.H5Dget_space.argtypes = hid, [hid]
        lib.H5Dget_type.restype, lib.H5Dget_type.argtypes = hid, [hid]
        lib.H5Sget_simple_extent_ndims.argtypes = [hid]
        lib.H5Sget_simple_extent_dims.argtypes = [hid, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        lib.H5Sclose.argtypes = [hid]
        lib.H5Tget_class.argtypes = [hid]
        lib.H5Tget_size.restype, lib.H5Tget_size.argtypes = C.c_size_t, [hid]
        lib.H5Tget_sign.argtypes = [hid]
        lib.H5Tclose.argtypes = [hid]
        lib.H5Dread.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]
        lib.H5Lexists.argtypes = [hid, C.c_char_p, hid]
        lib.H5Literate.argtypes = [hid, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p]
        _lib = lib
        return lib
    raise OSError("libhdf5 not found (set VD_HDF5_LIB); last error: %s" % last)


def available():
    try:
        _load()
        return True
    except OSError:
        return False


def _native(lib, name):
    return _hid.in_dll(lib, name).value


class File(object):
    """Read-only view of the root group of one HDF5 file."""

    def __init__(self, path, mode='r'):
        assert mode == 'r', "h5lite is read-only"
        self._lib = _load()
        self._fid = self._lib.H5Fopen(os.fsencode(path), 0, 0)      # H5F_ACC_RDONLY, H5P_DEFAULT
        if self._fid < 0:
            raise OSError("cannot open %s as HDF5" % path)
        self.path = path

    def close(self):
        if getattr(self, '_fid', -1) >= 0:
            self._lib.H5Fclose(self._fid)
            self._fid = -1

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
        return False

    def __contains__(self, name):
        return self._lib.H5Lexists(self._fid, name.encode(), 0) > 0

    def keys(self):
        names = []
        cb_t = C.CFUNCTYPE(C.c_int, _hid, C.c_char_p, C.c_void_p, C.c_void_p)
        cb = cb_t(lambda g, name, info, data: names.append(name.decode()) or 0)
        idx = C.c_uint64(0)
        rc = self._lib.H5Literate(self._fid, 0, 0, C.byref(idx), C.cast(cb, C.c_void_p), None)   # by name, increasing
        if rc < 0:
            raise OSError("H5Literate failed on %s" % self.path)
        return names

    def __getitem__(self, name):
        lib = self._lib
        d = lib.H5Dopen2(self._fid, name.encode(), 0)
        if d < 0:
            raise KeyError("%s has no dataset '%s'" % (self.path, name))
        try:
            sp = lib.H5Dget_space(d)
            nd = lib.H5Sget_simple_extent_ndims(sp)
            dims = (C.c_uint64 * max(nd, 1))()
            if nd > 0:
                lib.H5Sget_simple_extent_dims(sp, dims, None)
            lib.H5Sclose(sp)
            shape = tuple(int(dims[i]) for i in range(nd))
            t = lib.H5Dget_type(d)
            cls, size, sign = lib.H5Tget_class(t), lib.H5Tget_size(t), lib.H5Tget_sign(t)
            lib.H5Tclose(t)
            if cls == 0:      # H5T_INTEGER
                key = ('U' if sign == 0 else '') + 'INT%d' % (8 * size)
                dtype = np.dtype(('u' if sign == 0 else 'i') + str(size))
            elif cls == 1:    # H5T_FLOAT
                key = {4: 'FLOAT', 8: 'DOUBLE'}[size]
                dtype = np.dtype('f' + str(size))
            else:
                raise TypeError("dataset '%s': only integer / float datasets are supported (class %d)" % (name, cls))
            out = np.empty(shape, dtype=dtype)
            rc = lib.H5Dread(d, _native(lib, 'H5T_NATIVE_%s_g' % key), 0, 0, 0, out.ctypes.data)   # H5S_ALL, H5P_DEFAULT
            if rc < 0:
                raise OSError("H5Dread failed for '%s'" % name)
            return out
        finally:
            lib.H5Dclose(d)
