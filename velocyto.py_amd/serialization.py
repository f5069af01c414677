"""``velocyto/serialization.py``: the checkpoint container of a VelocytoLoom (serialization.py:9-115), on top of the ctypes-bound
libhdf5 of ``loom_io`` (no h5py in the image).

Format, as the reference writes it: one HDF5 dataset per attribute - numeric ndarrays under their own name (2-d ones chunked and
gzip-compressed when asked), every other object pickled (`pickle_protocol`, default 2), zlib-compressed (`noarray_compression`)
and stored as a uint8 dataset called ``"&" + name`` (`_obj2uint` / `_uint2obj`).  ``dump_hdf5`` walks the attributes of ANY object;
an object that keeps part of its state elsewhere - this package's VelocytoLoom: device matrices, compact neighbour-list results -
says what to write through ``_export_state(exclude)`` and rebuilds what it needs through ``_import_state()`` after ``load_hdf5``
has set the attributes.  Device matrices go out as the reference's (genes, cells) float64 arrays, so files round-trip between the
two implementations."""
from __future__ import annotations

import pickle
import zlib
from typing import Any, Dict, Iterable, Optional

import numpy as np

__all__ = ["dump_hdf5", "load_hdf5"]


def _obj2uint(obj: object, compression: int = 9, protocol: int = 2) -> np.ndarray:
    """serialization.py:9-26: a python object as a uint8 array (pickle, then zlib)."""
    return np.frombuffer(zlib.compress(pickle.dumps(obj, protocol=protocol), compression), dtype=np.uint8)


def _uint2obj(uint: np.ndarray) -> object:
    """serialization.py:29-41."""
    return pickle.loads(zlib.decompress(np.asarray(uint, dtype=np.uint8).tobytes()))


def _is_plain_array(val: Any) -> bool:
    """What goes into a dataset of its own: numeric / boolean ndarrays (the reference tests `type(v) is np.ndarray` and lets h5py
    refuse object arrays; strings and object arrays are pickled here like any other object)."""
    return isinstance(val, np.ndarray) and val.dtype.kind in "fiub"


def dump_hdf5(obj: Any, filename: str, data_compression: int = 7, chunks=(2048, 2048), noarray_compression: int = 9, pickle_protocol: int = 2,
              exclude_attributes: Optional[Iterable[str]] = None) -> None:
    """serialization.py:44-97: every attribute of `obj` into one HDF5 file.  2-d datasets chunked + gzip (`data_compression`,
    `chunks`; 0 = stored as they are), everything that is not a numeric array pickled with `pickle_protocol` and zlib level
    `noarray_compression` under "&name".  `exclude_attributes` (an extension) leaves attributes out."""
    from .loom_io import hdf5_dump
    exclude = set(exclude_attributes or ())
    state: Dict[str, Any] = obj._export_state(exclude) if hasattr(obj, "_export_state") else \
        {k: v for k, v in vars(obj).items() if k not in exclude}
    out: Dict[str, np.ndarray] = {}
    for name, val in state.items():
        if _is_plain_array(val):
            out[name] = val
        else:
            out["&" + name] = _obj2uint(val, compression=int(noarray_compression), protocol=int(pickle_protocol))
    hdf5_dump(filename, out, compression=int(data_compression), chunks=tuple(chunks))


def load_hdf5(filename: str, obj_class: type = None, dtype=None):
    """serialization.py:100-115: an instance of `obj_class` - ANY class, made with `obj_class.__new__(obj_class)` as the reference does
    (no `__init__` runs, so a subclass with another signature loads too); default: this package's VelocytoLoom - with every dataset of
    the file set as an attribute, "&name" datasets decoded back into the objects they held.  An object that knows how to rebuild
    device-side state (`_import_state`, VelocytoLoom) is asked to.  `dtype` (an extension) is the device storage type of a VelocytoLoom."""
    from .loom_io import hdf5_load
    if obj_class is None:
        from .analysis import VelocytoLoom
        obj_class = VelocytoLoom
    if not isinstance(obj_class, type):
        raise TypeError("obj_class must be a class")
    obj = obj_class.__new__(obj_class)
    if dtype is not None and hasattr(obj, "_dtype"):
        from . import ops
        object.__setattr__(obj, "_dtype", ops.resolve_dtype(dtype))
    for name, arr in hdf5_load(filename).items():
        if name.startswith("&"):
            setattr(obj, name[1:], _uint2obj(arr))
        else:
            setattr(obj, name, arr)
    if hasattr(obj, "_import_state"):
        obj._import_state()
    return obj
