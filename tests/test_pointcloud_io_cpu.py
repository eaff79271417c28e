"""Point-cloud readers and voxel down-sampling.  CPU only; files are synthesised
here in every encoding the reference's data uses (ASCII/binary PLY, PCD ascii/binary/
binary_compressed)."""
import struct

import os

import numpy as np
import pytest


def _pts(n=57, seed=0):
    return np.random.RandomState(seed).rand(n, 3).astype(np.float32)


def test_ply_ascii_with_trailing_elements(tmp_path, bunny):
    from hgmm_amd import pointcloud_io as io
    P = bunny[:500]
    f = tmp_path / "a.ply"
    with open(f, "w") as fh:
        fh.write("ply\nformat ascii 1.0\nobj_info is_mesh 0\nelement vertex %d\nproperty float x\n"
                 "property float y\nproperty float z\nelement range_grid 3\n"
                 "property list uchar int vertex_indices\nend_header\n" % len(P))
        for p in P:
            fh.write("%.7g %.7g %.7g \n" % tuple(p))
        fh.write("1 0\n0\n1 2\n")                      # the range grid the Stanford scans carry
    got = io.read_point_cloud(str(f))
    assert got.dtype == np.float64 and got.shape == (500, 3)
    np.testing.assert_allclose(got, P, rtol=1e-6)


@pytest.mark.parametrize("endian", ["little", "big"])
def test_ply_binary(tmp_path, endian):
    from hgmm_amd import pointcloud_io as io
    P = _pts()
    f = tmp_path / "b.ply"
    o = "<" if endian == "little" else ">"
    with open(f, "wb") as fh:
        fh.write(("ply\nformat binary_%s_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
                  "property float z\nproperty uchar intensity\nend_header\n" % (endian, len(P))).encode())
        for p in P:
            fh.write(struct.pack(o + "fffB", p[0], p[1], p[2], 7))
    np.testing.assert_array_equal(io.read_ply(str(f)), P.astype(np.float64))


def _pcd_header(n, mode):
    return ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\n"
            "COUNT 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n" % (n, n, mode)).encode()


def _lzf_literal_only(raw):
    out = bytearray()
    for i in range(0, len(raw), 32):
        chunk = raw[i:i + 32]
        out.append(len(chunk) - 1)
        out += chunk
    return bytes(out)


def test_pcd_all_encodings(tmp_path):
    from hgmm_amd import pointcloud_io as io
    P = _pts(101, 3)
    P[5] = np.nan                                       # organised clouds carry NaN holes
    fa, fb, fc = tmp_path / "a.pcd", tmp_path / "b.pcd", tmp_path / "c.pcd"
    with open(fa, "wb") as fh:
        fh.write(_pcd_header(len(P), "ascii"))
        for p in P:
            fh.write(("%.9g %.9g %.9g\n" % tuple(p)).encode())
    with open(fb, "wb") as fh:
        fh.write(_pcd_header(len(P), "binary"))
        fh.write(P.tobytes())
    raw = np.ascontiguousarray(P.T).tobytes()           # compressed payload is field-major
    comp = _lzf_literal_only(raw)
    with open(fc, "wb") as fh:
        fh.write(_pcd_header(len(P), "binary_compressed"))
        fh.write(struct.pack("<II", len(comp), len(raw)))
        fh.write(comp)
    want = P[np.isfinite(P).all(1)].astype(np.float64)
    for f in (fa, fb, fc):
        got = io.read_pcd(str(f))
        assert got.shape == (100, 3)
        np.testing.assert_allclose(got, want, rtol=1e-7)
    assert io.read_pcd(str(fb), drop_nan=False).shape == (101, 3)


def test_lzf_back_references():
    from hgmm_amd.pointcloud_io import _lzf_decompress
    # literal "abc", then a 3-byte back reference at distance 3, then an overlapping run
    stream = bytes([2]) + b"abc" + bytes([(1 << 5) | 0, 2]) + bytes([0]) + b"x" + bytes([(7 << 5) | 0, 3, 0])
    assert _lzf_decompress(stream, 6 + 1 + 12) == b"abcabc" + b"x" + b"x" * 12
    with pytest.raises(ValueError):
        _lzf_decompress(bytes([(1 << 5) | 0, 9]), 3)
    # truncated / overlong streams raise ValueError instead of silently shrinking the output
    for bad, n in ((bytes([5]) + b"ab", 6),                       # literal run longer than the stream
                   (bytes([2]) + b"abc" + bytes([(7 << 5) | 0]), 20),     # long back reference cut before its length
                   (bytes([2]) + b"abc" + bytes([(1 << 5) | 0]), 6),      # back reference cut before its offset
                   (bytes([2]) + b"abc" + bytes([(1 << 5) | 0, 2]), 4),   # back reference past the output size
                   (bytes([2]) + b"abc", 7)):                              # decodes to fewer bytes than announced
        with pytest.raises(ValueError):
            _lzf_decompress(bad, n)
    # overlapping back reference with a pattern longer than one byte: "ab" repeated
    stream = bytes([1]) + b"ab" + bytes([(5 << 5) | 0, 1])
    assert _lzf_decompress(stream, 2 + 7) == b"ab" + b"abababa"


def test_voxel_down_sample_properties(bunny):
    from hgmm_amd.pointcloud_io import voxel_down_sample
    P = bunny.astype(np.float64)
    for v in (0.002, 0.005, 0.02):
        Q = voxel_down_sample(P, v)
        assert 0 < len(Q) <= len(P)
        # every output point is the centroid of the input points of one voxel
        origin = P.min(0) - 0.5 * v
        vid = np.floor((P - origin) / v).astype(np.int64)
        qid = np.floor((Q - origin) / v).astype(np.int64)
        assert len(np.unique(qid, axis=0)) == len(Q) == len(np.unique(vid, axis=0))
        for k in range(0, len(Q), max(1, len(Q) // 25)):
            members = (vid == qid[k]).all(axis=1)
            np.testing.assert_allclose(Q[k], P[members].mean(0), rtol=1e-12, atol=1e-15)
    one = voxel_down_sample(P, 10.0)
    np.testing.assert_allclose(one, P.mean(0)[None], rtol=1e-12)


REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout only exists in the build container")
def test_readers_on_the_reference_s_own_files(bunny):
    """the package's readers on the files the reference ships, against fixtures that were written
    by an independent parse (tools/gen_golden.py: read_ply_vertices / gen_waymo_frames) -- ASCII PLY with a range
    grid after the vertex block, binary PCD, and the organised binary_compressed (LZF) PCD with NaN holes."""
    from conftest import GOLDEN
    from hgmm_amd.pointcloud_io import read_point_cloud, read_pcd
    P = read_point_cloud(os.path.join(REF, "data/bun000.ply"))
    assert P.dtype == np.float64 and np.array_equal(P.astype(np.float32), bunny)
    P45 = read_point_cloud(os.path.join(REF, "data/bun045.ply"))
    assert np.array_equal(P45.astype(np.float32), np.load(os.path.join(GOLDEN, "bun045_xyz.npy")))
    # in-test independent parse of the PLY vertex block (header says how many lines to take)
    lines = open(os.path.join(REF, "data/bun045.ply")).read().split("\n")
    n_v = int([l for l in lines if l.startswith("element vertex")][0].split()[-1])
    body = lines[lines.index("end_header") + 1:][:n_v]
    raw = np.array([[float(v) for v in l.split()[:3]] for l in body])
    assert np.array_equal(P45, raw)
    # the three copies of bunny.pcd (512 x 400 organised grid, LZF-compressed, 164544 NaN holes) hold exactly the
    # vertices of bun000.ply, in the same order
    for sub in ("gmmreg_gpu", "hgmm", "gmm_waymo/data"):
        path = os.path.join(REF, "src/python", sub, "bunny.pcd")
        Q = read_point_cloud(path)
        assert np.array_equal(Q.astype(np.float32), bunny)
        full = read_pcd(path, drop_nan=False)
        assert full.shape == (204800, 3) and int(np.isnan(full).any(axis=1).sum()) == 204800 - 40256
    frames = np.load(os.path.join(GOLDEN, "waymo_frames.npz"))
    for k in (1, 2, 5, 10, 50):
        W = read_point_cloud(os.path.join(REF, "src/python/gmmreg_gpu/waymo%d.pcd" % k))
        assert W.dtype == np.float64 and np.array_equal(W.astype(np.float32), frames["waymo%d" % k])
    for k in (1, 2):                                     # the hgmm directory holds copies of two of them
        W = read_point_cloud(os.path.join(REF, "src/python/hgmm/waymo%d.pcd" % k))
        assert np.array_equal(W.astype(np.float32), frames["waymo%d" % k])
    D = read_point_cloud(os.path.join(REF, "src/python/gmm_waymo/data/dragon.ply"))
    assert D.shape == (41841, 3) and np.isfinite(D).all()


def test_voxel_down_sample_hand_computed():
    """Open3D's documented rule (PointCloud::VoxelDownSample): voxel index = floor((p - (min_bound - voxel / 2)) /
    voxel), output = the mean of the points of each occupied voxel.  Seven points, voxel 1.0, worked by hand:
    min_bound = (0, 0, 0) -> grid origin (-0.5, -0.5, -0.5); indices: (0,0,0) for the first three points,
    (1,0,0) for the next two, (2,3,0) and (0,0,1) for the last two."""
    from hgmm_amd.pointcloud_io import voxel_down_sample
    P = np.array([[0.0, 0.0, 0.0], [0.4, 0.2, 0.1], [0.2, 0.4, 0.2],        # voxel (0,0,0)
                  [0.6, 0.0, 0.0], [1.4, 0.4, 0.3],                          # voxel (1,0,0): 1.1 and 1.9 -> floor 1
                  [2.2, 2.6, 0.1],                                           # voxel (2,3,0): 2.7, 3.1
                  [0.1, 0.2, 0.5]])                                          # voxel (0,0,1): z + 0.5 = 1.0 -> floor 1
    want = np.array([[0.2, 0.2, 0.1], [1.0, 0.2, 0.15], [2.2, 2.6, 0.1], [0.1, 0.2, 0.5]])
    got = voxel_down_sample(P, 1.0)
    assert got.shape == (4, 3)
    order = lambda a: a[np.lexsort(a.T[::-1])]
    np.testing.assert_allclose(order(got), order(want), rtol=0, atol=1e-15)
    # a point exactly on a voxel face belongs to the upper voxel (floor), and a shifted cloud shifts the grid with it
    np.testing.assert_allclose(order(voxel_down_sample(P + 7.25, 1.0)), order(want + 7.25), rtol=0, atol=1e-12)
