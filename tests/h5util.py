"""Test helper: write a GraphINVENT-style preprocessed ``.h5`` file (three contiguous, uncompressed int8 datasets
``nodes`` / ``edges`` / ``APDs`` — what DataProcesser.py:157-161, 273-289 creates through h5py) with libhdf5 through
ctypes, so HDF tests do not depend on the reference checkout (absent on the GPU box) or on h5py (absent everywhere)."""
import ctypes

import numpy as np

from graphinvent_amd.loader import _load_libhdf5

H5F_ACC_TRUNC = 2


def have_libhdf5() -> bool:
    try:
        _load_libhdf5()
        return True
    except RuntimeError:
        return False


def write_h5(path: str, nodes: np.ndarray, edges: np.ndarray, apds: np.ndarray) -> None:
    lib, int8, lock = _load_libhdf5()
    i64 = ctypes.c_int64
    lib.H5Fcreate.restype = i64
    lib.H5Fcreate.argtypes = [ctypes.c_char_p, ctypes.c_uint, i64, i64]
    lib.H5Dcreate2.restype = i64
    lib.H5Dcreate2.argtypes = [i64, ctypes.c_char_p, i64, i64, i64, i64, i64]
    lib.H5Dwrite.argtypes = [i64] * 5 + [ctypes.c_void_p]
    with lock:
        f = lib.H5Fcreate(path.encode(), H5F_ACC_TRUNC, 0, 0)
        if f < 0:
            raise OSError(f"cannot create {path}")
        try:
            for name, arr in ((b"nodes", nodes), (b"edges", edges), (b"APDs", apds)):
                arr = np.ascontiguousarray(arr, dtype=np.int8)
                dims = (ctypes.c_uint64 * arr.ndim)(*arr.shape)
                sp = lib.H5Screate_simple(arr.ndim, dims, None)
                d = lib.H5Dcreate2(f, name, int8, sp, 0, 0, 0)
                ok = d >= 0 and lib.H5Dwrite(d, int8, 0, 0, 0, arr.ctypes.data) >= 0
                if d >= 0:
                    lib.H5Dclose(d)
                lib.H5Sclose(sp)
                if not ok:
                    raise OSError(f"writing {name!r} to {path} failed")
        finally:
            lib.H5Fclose(f)
