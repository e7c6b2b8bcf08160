"""
Checkpoint container of ``VB.save`` / ``VB.load`` (reference: vmp.py:237-356,
stochastic.py:305-355, expfamily.py:507-535).

The reference writes HDF5 through h5py with the layout ``nodes/<name>/{u0.., phi0.., f, g,
observed}``, ``L``, ``cputime``, ``iter``, ``converged``, ``boundterms/<name>``.  The same
logical layout is kept here.  When h5py is importable the file is HDF5 with exactly those
groups (node groups of the generic engine are then readable by the reference); otherwise --
h5py is not part of this image -- the same keys go into a NumPy ``.npz`` archive with
``/``-joined names.  Fused blocks additionally keep their device state vector under
``plans/<i>/...`` because their node parameters live there in packed form.
"""
import numpy as np

try:                                   # pragma: no cover - not installed in this image
    import h5py
except Exception:                      # noqa: BLE001
    h5py = None


class Writer:
    def __init__(self, filename):
        self.filename = filename
        self.items = {}

    def put(self, path, value):
        self.items[path] = np.asarray(value)

    def close(self):
        if h5py is not None:
            with h5py.File(self.filename, 'w') as f:
                for k, v in self.items.items():
                    f[k] = v
        else:
            with open(self.filename, 'wb') as fh:      # keep the caller's file name as is
                np.savez(fh, **self.items)


class Reader:
    def __init__(self, filename):
        self._h5 = None
        if h5py is not None and h5py.is_hdf5(filename):
            self._h5 = h5py.File(filename, 'r')
        else:
            self._npz = np.load(filename, allow_pickle=False)

    def has(self, path):
        return (path in self._h5) if self._h5 is not None else (path in self._npz.files)

    def keys(self):
        """All leaf paths."""
        if self._h5 is None:
            return list(self._npz.files)
        out = []
        self._h5.visititems(lambda name, obj: out.append(name) if hasattr(obj, 'shape') else None)
        return out

    def get(self, path):
        if not self.has(path):
            raise KeyError(path)
        return self._h5[path][...] if self._h5 is not None else self._npz[path]

    def close(self):
        if self._h5 is not None:
            self._h5.close()
        else:
            self._npz.close()
