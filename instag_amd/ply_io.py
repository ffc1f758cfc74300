"""PLY point-cloud files without the ``plyfile`` package: the layout ``scene/gaussian_model.py`` writes and reads
(save_ply :443-463, load_ply :497-527: one ``vertex`` element, float32 properties x y z nx ny nz f_dc_* f_rest_* opacity
scale_* rot_*, binary little-endian as ``PlyData([el]).write`` emits on x86) and the readers' side of the usual variants
(ASCII, big-endian, other scalar property types)."""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np

_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
          "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
          "double": "f8", "float64": "f8"}


def write_vertex_ply(path, columns: "OrderedDict[str, np.ndarray]"):
    """columns: name -> [N] array; everything is stored as float32 in insertion order."""
    names = list(columns)
    n = len(next(iter(columns.values()))) if names else 0
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    rec = np.empty(n, dtype=[(k, "<f4") for k in names])
    for k in names:
        rec[k] = np.asarray(columns[k], dtype=np.float32).reshape(n)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {n}"]
    header += [f"property float {k}" for k in names]
    header += ["end_header", ""]
    with open(path, "wb") as fh:
        fh.write("\n".join(header).encode("ascii"))
        fh.write(rec.tobytes())


def read_vertex_ply(path) -> "OrderedDict[str, np.ndarray]":
    """-> name -> [N] array of the first element (scalar properties only)."""
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements = None, []
        while True:
            line = fh.readline()
            if not line:
                raise ValueError(f"{path}: header without end_header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append((tok[1], int(tok[2]), []))
            elif tok[0] == "property":
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not supported")
                elements[-1][2].append((tok[2], _TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if not elements:
            raise ValueError(f"{path}: no element")
        name, n, props = elements[0]
        if fmt == "ascii":
            rows = np.loadtxt(fh, max_rows=n, ndmin=2) if n else np.zeros((0, len(props)))
            return OrderedDict((k, rows[:, i].astype(t)) for i, (k, t) in enumerate(props))
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(k, order + t) for k, t in props])
        data = np.frombuffer(fh.read(dt.itemsize * n), dtype=dt, count=n)
        return OrderedDict((k, np.ascontiguousarray(data[k])) for k, _ in props)
