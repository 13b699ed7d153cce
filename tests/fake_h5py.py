"""A dict-backed stand-in for an open ``h5py.File`` (this image has no h5py): dataset paths map to
objects with ``.shape`` / ``.dtype`` and numpy-style ``[...]``, groups to nested mappings, and the
file is a context manager -- the part of the h5py surface ``tools/h5_to_store.convert`` and the
reference's ``calibration.py:304-423`` use. Test infrastructure only."""
import numpy as np


class Dataset(object):
    def __init__(self, array):
        self._a = np.asarray(array)
        self.shape, self.dtype, self.ndim = self._a.shape, self._a.dtype, self._a.ndim
        self.reads = 0          # how many times the data was sliced (the converter goes step by step)

    def __getitem__(self, key):
        self.reads += 1
        return np.array(self._a[key])        # a copy, as h5py hands out

    def __len__(self):
        return self.shape[0]


class File(object):
    """``File({'MERRA2/Tmin': array, ...})``; ``f['MERRA2']['Tmin']`` and ``f['MERRA2/Tmin']`` both work."""

    def __init__(self, datasets, mode='r'):
        self._d = {k.strip('/'): (v if isinstance(v, Dataset) else Dataset(v)) for k, v in datasets.items()}

    def __getitem__(self, path):
        path = path.strip('/')
        if path in self._d:
            return self._d[path]
        sub = {k[len(path) + 1:]: v for k, v in self._d.items() if k.startswith(path + '/')}
        if not sub:
            raise KeyError("Unable to open object (object '%s' doesn't exist)" % path)
        return File(sub)

    def __contains__(self, path):
        try:
            self[path]
            return True
        except KeyError:
            return False

    def keys(self):
        return sorted({k.split('/')[0] for k in self._d})

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
