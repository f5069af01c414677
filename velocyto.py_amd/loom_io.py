"""Minimal .loom ingest (analysis.py:56-64 reads layers spliced/unspliced/ambiguous + row/col attrs).

The loom format is HDF5 (``/layers/{spliced,unspliced,ambiguous}``, ``/row_attrs/*``, ``/col_attrs/*``;
writer: commands/_run.py:283-297).  The image ships neither loompy nor h5py, so this module uses
whichever is importable and otherwise fails loudly; a libhdf5-backed reader streaming straight into
pinned host buffers is the first "next" row."""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np


def read_loom(path: str) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray], Dict[str, np.ndarray]]:
    try:
        import h5py
    except ImportError:
        h5py = None
    if h5py is not None:
        with h5py.File(path, "r") as f:
            layers = {k: f["layers"][k][:, :] for k in ("spliced", "unspliced", "ambiguous") if k in f["layers"]}
            ca = {k: f["col_attrs"][k][...] for k in f["col_attrs"]}
            ra = {k: f["row_attrs"][k][...] for k in f["row_attrs"]}
        return layers, ca, ra
    try:
        import loompy
    except ImportError:
        raise ImportError("reading .loom files needs h5py or loompy (neither is installed); "
                          "use VelocytoLoom.from_arrays(S, U, A, ca, ra) with in-memory layers") from None
    ds = loompy.connect(path)
    try:
        layers = {k: ds.layer[k][:, :] for k in ("spliced", "unspliced", "ambiguous")}
        ca, ra = dict(ds.col_attrs.items()), dict(ds.row_attrs.items())
    finally:
        ds.close()
    return layers, ca, ra
