"""``velocyto/serialization.py`` call surface: ``dump_hdf5`` / ``load_hdf5`` (serialization.py:44-115) on top of the
ctypes-bound libhdf5 of ``loom_io`` (no h5py in the image).  Device matrices are written as the reference's
(genes, cells) float64 datasets, so files round-trip between the two implementations."""
from __future__ import annotations

import pickle
import zlib
from typing import Any

import numpy as np

__all__ = ["dump_hdf5", "load_hdf5"]


def _obj2uint(obj: object, compression: int = 9, protocol: int = 2) -> np.ndarray:
    """serialization.py:9-26: a python object as a uint8 array (pickle, then zlib)."""
    return np.frombuffer(zlib.compress(pickle.dumps(obj, protocol=protocol), compression), dtype=np.uint8)


def _uint2obj(uint: np.ndarray) -> object:
    """serialization.py:29-41."""
    return pickle.loads(zlib.decompress(np.asarray(uint, dtype=np.uint8).tobytes()))


def dump_hdf5(obj: Any, filename: str, data_compression: int = 7, chunks=(2048, 2048), noarray_compression: int = 9, pickle_protocol: int = 2,
              exclude_attributes=None) -> None:
    """serialization.py:44-97 for a VelocytoLoom of this package: 2-d datasets chunked + gzip (`data_compression`, `chunks`),
    everything that is not an array pickled with `pickle_protocol` and zlib level `noarray_compression`
    (`exclude_attributes` is an extension)."""
    obj.to_hdf5(filename, exclude=set(exclude_attributes or ()), data_compression=data_compression, chunks=chunks,
                noarray_compression=noarray_compression, pickle_protocol=pickle_protocol)


def load_hdf5(filename: str, obj_class: type = None, dtype=None):
    """serialization.py:100-115: `obj_class` (VelocytoLoom or a subclass) is the class that is instantiated."""
    from .analysis import load_velocyto_hdf5
    return load_velocyto_hdf5(filename, dtype=dtype, obj_class=obj_class)
