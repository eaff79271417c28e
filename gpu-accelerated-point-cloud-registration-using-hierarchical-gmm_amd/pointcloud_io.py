"""Point-cloud file I/O and voxel down-sampling without Open3D.

The reference's drivers read their inputs through Open3D (``o3.read_point_cloud`` +
``o3.voxel_down_sample``: run_gmm_static.py:27-28, hgmm_gpu.py:821-832), which this image lacks
and which is not part of the hot path; these are small NumPy readers for the formats the
reference ships: ASCII/binary PLY (data/bun*.ply, dragon.ply) and PCD ``ascii`` / ``binary`` /
``binary_compressed`` (waymo*.pcd, bunny.pcd).  Host-side, float64 output like Open3D.
"""
from __future__ import annotations

import numpy as np

_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4",
              "float": "f4", "double": "f8", "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2",
              "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8"}


def read_ply(path) -> np.ndarray:
    """Vertex positions [N,3] of a PLY file (ascii, binary_little_endian, binary_big_endian).
    Only the ``vertex`` element is read; it must come first (true for the Stanford scans)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s: not a PLY file" % path)
        fmt, n_vertex, props, in_vertex, first_element = None, None, [], False, None
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: unterminated PLY header" % path)
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if first_element is None:
                    first_element = tok[1]
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n_vertex = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list property inside the vertex element is not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if n_vertex is None or first_element != "vertex":
            raise ValueError("%s: vertex element missing or not first" % path)
        names = [p[0] for p in props]
        if not all(a in names for a in "xyz"):
            raise ValueError("%s: vertex element has no x/y/z" % path)
        cols = [names.index(a) for a in "xyz"]
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=n_vertex, dtype=np.float64, ndmin=2)
            return np.ascontiguousarray(data[:, cols])
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, order + t) for n, t in props])
        rec = np.frombuffer(f.read(dt.itemsize * n_vertex), dtype=dt, count=n_vertex)
        return np.stack([rec[a].astype(np.float64) for a in "xyz"], axis=1)


def _lzf_decompress(src: bytes, out_len: int) -> bytes:
    """LZF (liblzf) decompression, the codec of PCD ``binary_compressed``."""
    out = bytearray(out_len)
    ip, op, n = 0, 0, len(src)
    while ip < n:
        ctrl = src[ip]
        ip += 1
        if ctrl < 32:                                   # literal run of ctrl + 1 bytes
            run = ctrl + 1
            if ip + run > n or op + run > out_len:
                raise ValueError("corrupt LZF stream (literal run past the end)")
            out[op:op + run] = src[ip:ip + run]
            ip += run
            op += run
        else:                                           # back reference
            length = ctrl >> 5
            if length == 7:
                if ip >= n:
                    raise ValueError("corrupt LZF stream (truncated)")
                length += src[ip]
                ip += 1
            if ip >= n:
                raise ValueError("corrupt LZF stream (truncated)")
            ref = op - ((ctrl & 0x1f) << 8) - src[ip] - 1
            ip += 1
            length += 2
            if ref < 0 or op + length > out_len:
                raise ValueError("corrupt LZF stream (bad back reference)")
            dist = op - ref
            if dist >= length:                          # source and destination do not overlap
                out[op:op + length] = out[ref:ref + length]
            else:                                       # overlapping run = the last `dist` bytes repeated
                pat = bytes(out[ref:op])
                reps = -(-length // dist)
                out[op:op + length] = (pat * reps)[:length]
            op += length
    if op != out_len:
        raise ValueError("LZF stream decoded to %d bytes, expected %d" % (op, out_len))
    return bytes(out)


def read_pcd(path, drop_nan=True) -> np.ndarray:
    """Positions [N,3] of a PCD v0.7 file (DATA ascii | binary | binary_compressed)."""
    with open(path, "rb") as f:
        hdr = {}
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: unterminated PCD header" % path)
            s = line.decode("ascii", "replace").strip()
            if not s or s.startswith("#"):
                continue
            key, _, val = s.partition(" ")
            hdr[key.upper()] = val.split()
            if key.upper() == "DATA":
                break
        fields = hdr["FIELDS"]
        sizes = [int(v) for v in hdr["SIZE"]]
        types = hdr["TYPE"]
        counts = [int(v) for v in hdr.get("COUNT", ["1"] * len(fields))]
        npts = int(hdr["POINTS"][0]) if "POINTS" in hdr else int(hdr["WIDTH"][0]) * int(hdr["HEIGHT"][0])
        kind = {"F": "f", "I": "i", "U": "u"}
        dts = [np.dtype("<%s%d" % (kind[t], s)) for t, s in zip(types, sizes)]
        mode = hdr["DATA"][0]
        if any(c != 1 for c in counts):
            raise ValueError("PCD fields with COUNT != 1 are not supported")
        ix = [fields.index(a) for a in "xyz"]
        if mode == "ascii":
            data = np.loadtxt(f, max_rows=npts, dtype=np.float64, ndmin=2)
            pts = data[:, ix]
        elif mode == "binary":
            rec = np.dtype([(n, d) for n, d in zip(fields, dts)])
            raw = np.frombuffer(f.read(rec.itemsize * npts), dtype=rec, count=npts)
            pts = np.stack([raw[a].astype(np.float64) for a in "xyz"], axis=1)
        elif mode == "binary_compressed":
            comp, uncomp = np.frombuffer(f.read(8), dtype="<u4")
            buf = _lzf_decompress(f.read(int(comp)), int(uncomp))
            cols, off = {}, 0
            for name, d in zip(fields, dts):            # stored field-by-field (structure of arrays)
                cols[name] = np.frombuffer(buf, dtype=d, count=npts, offset=off).astype(np.float64)
                off += d.itemsize * npts
            pts = np.stack([cols[a] for a in "xyz"], axis=1)
        else:
            raise ValueError("unknown PCD DATA mode %r" % mode)
    if drop_nan:
        pts = pts[np.isfinite(pts).all(axis=1)]
    return np.ascontiguousarray(pts)


def read_point_cloud(path) -> np.ndarray:
    p = str(path).lower()
    if p.endswith(".ply"):
        return read_ply(path)
    if p.endswith(".pcd"):
        return read_pcd(path)
    if p.endswith(".npy"):
        return np.asarray(np.load(path), dtype=np.float64)
    raise ValueError("unsupported point-cloud format: %s" % path)


def voxel_down_sample(points, voxel_size: float) -> np.ndarray:
    """One point per occupied voxel = the mean of the points inside it (what Open3D's
    ``voxel_down_sample`` computes: grid origin at ``min_bound - voxel_size / 2``).  Output order
    is by voxel index (Open3D's order is hash-map dependent; the set of points is what matters)."""
    P = np.asarray(points, dtype=np.float64)
    if voxel_size <= 0:
        raise ValueError("voxel_size must be positive")
    origin = P.min(axis=0) - 0.5 * voxel_size
    idx = np.floor((P - origin) / voxel_size).astype(np.int64)
    _, inv, cnt = np.unique(idx, axis=0, return_inverse=True, return_counts=True)
    inv = inv.reshape(-1)
    out = np.zeros((len(cnt), 3))
    np.add.at(out, inv, P)
    return out / cnt[:, None]
