"""Read-only access to HDF5 files through the HDF5 C library (libhdf5.so) and ctypes -- what the sequence loader uses for
the reference's `.h5` files when `h5py` is not installed for this interpreter (this image ships libhdf5 1.10 under
/opt/conda/lib but no h5py for /usr/bin/python3).  Only what dataloader/h5.py needs of the reference's file layout
(dataloader/h5.py:24-42,68,127-131): numeric datasets of any rank (contiguous or chunked, any filter the library was built
with), partial reads of 1-D datasets, scalar numeric attributes of the file and of datasets, group listings in name order
(the order h5py's `visititems` reports).

    f = File(path); f.dataset("events/ts")[a:b]; f.attr("t0"); f.names("images"); f.dataset("images/image000000001").attr("timestamp")
"""

import ctypes
import ctypes.util
import os

import numpy as np

hid_t = ctypes.c_int64
hsize_t = ctypes.c_uint64
herr_t = ctypes.c_int

_CANDIDATES = (os.environ.get("EVF_LIBHDF5"), ctypes.util.find_library("hdf5"), "/opt/conda/lib/libhdf5.so",
               "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so", "libhdf5.so")
_lib = None

# H5T_class_t / H5T_sign_t / H5T_direction_t, H5F_ACC_RDONLY, H5P_DEFAULT, H5S_ALL, H5S_SELECT_SET, H5_INDEX_NAME, H5_ITER_INC
_INTEGER, _FLOAT, _ENUM = 0, 1, 8
_SGN_NONE = 0
_DIR_ASCEND = 1


class Hdf5Error(OSError):
    pass


class _H5G_info(ctypes.Structure):
    _fields_ = [("storage_type", ctypes.c_int), ("nlinks", hsize_t), ("max_corder", ctypes.c_int64), ("mounted", ctypes.c_int)]


def available():
    try:
        load()
        return True
    except Hdf5Error:
        return False


def load():
    """dlopen the HDF5 library once; raises Hdf5Error when none of the candidate paths loads."""
    global _lib
    if _lib is not None:
        return _lib
    err = None
    for cand in _CANDIDATES:
        if not cand:
            continue
        try:
            lib = ctypes.CDLL(cand)
            break
        except OSError as e:
            err = e
    else:
        raise Hdf5Error(f"no HDF5 library found (tried EVF_LIBHDF5, the linker path, /opt/conda/lib): {err}")
    sig = {
        "H5open": ([], herr_t), "H5Fopen": ([ctypes.c_char_p, ctypes.c_uint, hid_t], hid_t), "H5Fclose": ([hid_t], herr_t),
        "H5Dopen2": ([hid_t, ctypes.c_char_p, hid_t], hid_t), "H5Dclose": ([hid_t], herr_t), "H5Dget_space": ([hid_t], hid_t),
        "H5Dget_type": ([hid_t], hid_t), "H5Dread": ([hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p], herr_t),
        "H5Sclose": ([hid_t], herr_t), "H5Sget_simple_extent_ndims": ([hid_t], ctypes.c_int),
        "H5Sget_simple_extent_dims": ([hid_t, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t)], ctypes.c_int),
        "H5Screate_simple": ([ctypes.c_int, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t)], hid_t),
        "H5Sselect_hyperslab": ([hid_t, ctypes.c_int, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t),
                                 ctypes.POINTER(hsize_t)], herr_t),
        "H5Tclose": ([hid_t], herr_t), "H5Tget_class": ([hid_t], ctypes.c_int), "H5Tget_size": ([hid_t], ctypes.c_size_t),
        "H5Tget_sign": ([hid_t], ctypes.c_int), "H5Tget_native_type": ([hid_t, ctypes.c_int], hid_t),
        "H5Tget_super": ([hid_t], hid_t),
        "H5Aopen": ([hid_t, ctypes.c_char_p, hid_t], hid_t),
        "H5Aopen_by_name": ([hid_t, ctypes.c_char_p, ctypes.c_char_p, hid_t, hid_t], hid_t), "H5Aclose": ([hid_t], herr_t), "H5Aget_type": ([hid_t], hid_t),
        "H5Aread": ([hid_t, hid_t, ctypes.c_void_p], herr_t), "H5Aexists": ([hid_t, ctypes.c_char_p], ctypes.c_int),
        "H5Gopen2": ([hid_t, ctypes.c_char_p, hid_t], hid_t), "H5Gclose": ([hid_t], herr_t),
        "H5Gget_info": ([hid_t, ctypes.POINTER(_H5G_info)], herr_t),
        "H5Lget_name_by_idx": ([hid_t, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, hsize_t, ctypes.c_char_p, ctypes.c_size_t, hid_t],
                               ctypes.c_ssize_t),
        "H5Lexists": ([hid_t, ctypes.c_char_p, hid_t], ctypes.c_int),
        "H5Eset_auto2": ([hid_t, ctypes.c_void_p, ctypes.c_void_p], herr_t),
    }
    for name, (args, res) in sig.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = args, res
    if lib.H5open() < 0:
        raise Hdf5Error("H5open failed")
    lib.H5Eset_auto2(0, None, None)  # failures surface as Python exceptions, not as stack dumps on stderr
    _lib = lib
    return lib


def _np_dtype(lib, tid):
    """numpy dtype of an HDF5 integer / float type (native byte order: reads go through H5Tget_native_type)."""
    cls, size = lib.H5Tget_class(tid), lib.H5Tget_size(tid)
    if cls == _FLOAT:
        return np.dtype(f"f{size}")
    if cls == _INTEGER:
        return np.dtype(("u" if lib.H5Tget_sign(tid) == _SGN_NONE else "i") + str(size))
    if cls == _ENUM:  # h5py stores numpy booleans (the event polarities `ps`) as an enumeration over int8: read the base integers
        base = lib.H5Tget_super(tid)
        dt = _np_dtype(lib, base)
        lib.H5Tclose(base)
        return dt
    raise Hdf5Error(f"unsupported HDF5 type class {cls} (the sequence files hold integers and floats)")


def _read_attr(lib, obj, name):
    if lib.H5Aexists(obj, name.encode()) <= 0:
        raise KeyError(name)
    a = lib.H5Aopen(obj, name.encode(), 0)
    if a < 0:
        raise Hdf5Error(f"cannot open attribute {name!r}")
    try:
        ft = lib.H5Aget_type(a)
        mt = lib.H5Tget_native_type(ft, _DIR_ASCEND)
        buf = np.empty((), dtype=_np_dtype(lib, mt))
        rc = lib.H5Aread(a, mt, buf.ctypes.data)
        lib.H5Tclose(mt)
        lib.H5Tclose(ft)
        if rc < 0:
            raise Hdf5Error(f"cannot read attribute {name!r}")
        return buf[()]
    finally:
        lib.H5Aclose(a)


class Dataset:
    """A numeric dataset: len(), shape, dtype, ds[i], ds[a:b] (first axis), ds[:] / np.asarray(ds), ds.attr(name)."""

    def __init__(self, lib, did, name):
        self._lib, self._id, self.name = lib, did, name
        sp = lib.H5Dget_space(did)
        nd = lib.H5Sget_simple_extent_ndims(sp)
        dims = (hsize_t * max(nd, 1))()
        lib.H5Sget_simple_extent_dims(sp, dims, None)
        lib.H5Sclose(sp)
        self.shape = tuple(int(dims[i]) for i in range(nd))
        ft = lib.H5Dget_type(did)
        self._mt = lib.H5Tget_native_type(ft, _DIR_ASCEND)
        lib.H5Tclose(ft)
        self.dtype = _np_dtype(lib, self._mt)

    def __len__(self):
        return self.shape[0] if self.shape else 1

    def attr(self, name):
        return _read_attr(self._lib, self._id, name)

    def _read(self, start, count):
        lib = self._lib
        out = np.empty((count,) + self.shape[1:], dtype=self.dtype)
        if out.size == 0:
            return out
        nd = len(self.shape)
        fs = lib.H5Dget_space(self._id)
        st = (hsize_t * nd)(start, *([0] * (nd - 1)))
        cn = (hsize_t * nd)(count, *self.shape[1:])
        if lib.H5Sselect_hyperslab(fs, 0, st, None, cn, None) < 0:
            lib.H5Sclose(fs)
            raise Hdf5Error(f"cannot select {self.name}[{start}:{start + count}]")
        ms = lib.H5Screate_simple(nd, cn, None)
        rc = lib.H5Dread(self._id, self._mt, ms, fs, 0, out.ctypes.data)
        lib.H5Sclose(ms)
        lib.H5Sclose(fs)
        if rc < 0:
            raise Hdf5Error(f"cannot read {self.name}[{start}:{start + count}]")
        return out

    def __getitem__(self, key):
        if not self.shape:  # scalar dataset
            out = np.empty((), dtype=self.dtype)
            if self._lib.H5Dread(self._id, self._mt, 0, 0, 0, out.ctypes.data) < 0:
                raise Hdf5Error(f"cannot read {self.name}")
            return out[()]
        n = self.shape[0]
        if isinstance(key, slice):
            a, b, step = key.indices(n)
            if step != 1:
                return self._read(0, n)[key]
            return self._read(a, max(b - a, 0))
        if key is Ellipsis or key == ():
            return self._read(0, n)
        i = int(key)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(key)
        return self._read(i, 1)[0]

    def __array__(self, dtype=None, copy=None):
        a = self._read(0, self.shape[0]) if self.shape else np.asarray(self[()])
        return a.astype(dtype) if dtype is not None else a

    def close(self):
        if self._id >= 0:
            self._lib.H5Tclose(self._mt)
            self._lib.H5Dclose(self._id)
            self._id = -1


class File:
    def __init__(self, path):
        self._lib = load()
        self._id = self._lib.H5Fopen(os.fsencode(path), 0, 0)
        if self._id < 0:
            raise Hdf5Error(f"cannot open {path} as HDF5")
        self._open = {}

    def attr(self, name):
        return _read_attr(self._lib, self._id, name)

    def exists(self, path):
        cur = ""
        for part in path.strip("/").split("/"):  # every intermediate link must exist before H5Lexists may be asked about the next
            cur = cur + "/" + part
            if self._lib.H5Lexists(self._id, cur.encode(), 0) <= 0:
                return False
        return True

    def dataset(self, path, keep=True):
        """keep=True: the handle stays open until File.close() (the few `events/*` arrays a loader slices all the time);
        keep=False: a fresh handle the CALLER closes -- frame / flow-map datasets, of which a real sequence has tens of thousands."""
        if keep and path in self._open:
            return self._open[path]
        did = self._lib.H5Dopen2(self._id, path.encode(), 0)
        if did < 0:
            raise KeyError(path)
        d = Dataset(self._lib, did, path)
        if keep:
            self._open[path] = d
        return d

    def read(self, path):
        """The whole dataset as an array; nothing stays open."""
        d = self.dataset(path, keep=False)
        try:
            return np.asarray(d)
        finally:
            d.close()

    def dataset_attr(self, path, name):
        """An attribute of a dataset without keeping (or, where the library allows, without opening) the dataset."""
        lib = self._lib
        if hasattr(lib, "H5Aopen_by_name"):
            a = lib.H5Aopen_by_name(self._id, path.encode(), name.encode(), 0, 0)
            if a < 0:
                raise KeyError(f"{path}@{name}")
            try:
                ft = lib.H5Aget_type(a)
                mt = lib.H5Tget_native_type(ft, _DIR_ASCEND)
                buf = np.empty((), dtype=_np_dtype(lib, mt))
                rc = lib.H5Aread(a, mt, buf.ctypes.data)
                lib.H5Tclose(mt)
                lib.H5Tclose(ft)
                if rc < 0:
                    raise Hdf5Error(f"cannot read attribute {name!r} of {path}")
                return buf[()]
            finally:
                lib.H5Aclose(a)
        d = self.dataset(path, keep=False)
        try:
            return d.attr(name)
        finally:
            d.close()

    def names(self, group):
        """Link names of a group in increasing name order (the order of h5py's visit / visititems)."""
        lib = self._lib
        gid = lib.H5Gopen2(self._id, group.encode(), 0)
        if gid < 0:
            raise KeyError(group)
        try:
            info = _H5G_info()
            if lib.H5Gget_info(gid, ctypes.byref(info)) < 0:
                raise Hdf5Error(f"cannot query group {group!r}")
            out = []
            for i in range(int(info.nlinks)):
                n = lib.H5Lget_name_by_idx(gid, b".", 0, 0, i, None, 0, 0)
                if n < 0:
                    raise Hdf5Error(f"cannot list link {i} of group {group!r}")
                buf = ctypes.create_string_buffer(n + 1)
                if lib.H5Lget_name_by_idx(gid, b".", 0, 0, i, buf, n + 1, 0) < 0:
                    raise Hdf5Error(f"cannot list link {i} of group {group!r}")
                out.append(buf.value.decode())
            return out
        finally:
            lib.H5Gclose(gid)

    def close(self):
        if self._id >= 0:
            for d in self._open.values():
                d.close()
            self._open = {}
            self._lib.H5Fclose(self._id)
            self._id = -1
