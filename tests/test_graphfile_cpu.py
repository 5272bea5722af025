"""Graph files of the reference (NetCDF classic family, schema of Grid::Save,
Grid.cpp:1066-1097) read and written without PnetCDF.

The reference ships no graph file and PnetCDF is not installed, so the format
is pinned three ways: scipy.io.netcdf_file (an independent implementation of
CDF-1 / CDF-2) reads what this library writes and writes what this library
reads; a CDF-5 header assembled byte by byte from the format specification in
this file is read back; and CDF-5 files round-trip.
"""
import os
import struct
import subprocess

import numpy as np
import pytest
from scipy.io import netcdf_file

from admm_amd import graphfile as GF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _grid_edges(rows, cols):
    """Edge order of Grid::Generate (Grid.cpp:581-615), stated independently."""
    el = []
    for r in range(rows):
        for c in range(cols - 1):
            el.append((r * cols + c, r * cols + c + 1))
        if r < rows - 1:
            for c in range(cols):
                el.append((r * cols + c, (r + 1) * cols + c))
    return np.array(el, dtype=np.int32).reshape(-1, 2)


def _problem(rows, cols, k, seed=0):
    rng = np.random.default_rng(seed)
    el = _grid_edges(rows, cols)
    un = rng.standard_normal((rows * cols, k)).astype(np.float32)
    pw = rng.standard_normal((el.shape[0], k * k)).astype(np.float32)
    return el, un, pw


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "admm_graphfile.h")).read()
    import re
    declared = set(re.findall(r"\b(admm_graphfile_\w+)\s*\(", hdr))
    assert declared == set(GF.EXPORTS)
    for name in GF.EXPORTS:
        getattr(GF.lib, name)


@pytest.mark.parametrize("version", [0, 1, 2])
def test_scipy_reads_what_we_write(tmp_path, version):
    rows, cols, k = 5, 4, 3
    el, un, pw = _problem(rows, cols, k)
    f = tmp_path / "g.nc"
    GF.write(f, rows * cols, el, un, pw, rows=rows, cols=cols, version=version)
    with netcdf_file(str(f), "r", mmap=False) as nc:
        assert nc.version_byte == (2 if version == 2 else 1)
        # dimensions, in Grid::Save's order of definition (Grid.cpp:1070-1075)
        assert list(nc.dimensions.items()) == [
            ("numVertices", rows * cols), ("numVerticePotentials", k), ("numEdges", el.shape[0]),
            ("numEdgePotentials", k * k), ("numEndPointsPerEdge", 2)]
        assert list(nc.variables) == ["edgeList", "edgePotentials", "verticePotentials"]  # Grid.cpp:1079-1085
        assert nc.variables["edgeList"].dimensions == ("numEdges", "numEndPointsPerEdge")
        assert nc.variables["edgePotentials"].dimensions == ("numEdges", "numEdgePotentials")
        assert nc.variables["verticePotentials"].dimensions == ("numVertices", "numVerticePotentials")
        assert np.array_equal(nc.variables["edgeList"][:], el)
        assert np.array_equal(nc.variables["edgePotentials"][:], pw)
        assert np.array_equal(nc.variables["verticePotentials"][:], un)
        assert nc.variables["edgeList"][:].dtype.kind == "i"
        # attributes (Grid.cpp:1087-1096); scipy drops the trailing NUL of the
        # 5-character graphType, whose length the CDF-5 header test checks
        assert nc.graphType == b"Grid"
        assert (nc.numRows, nc.numCols, nc.numEdges, nc.numVertices) == (rows, cols, el.shape[0], rows * cols)
        assert (nc.numVerticePotentials, nc.numEdgePotentials) == (k, k * k)


@pytest.mark.parametrize("version", [1, 2])
def test_we_read_what_scipy_writes(tmp_path, version):
    rows, cols, k = 4, 6, 2
    el, un, pw = _problem(rows, cols, k, seed=3)
    f = tmp_path / "s.nc"
    with netcdf_file(str(f), "w", version=version) as nc:
        nc.createDimension("numVertices", rows * cols)
        nc.createDimension("numVerticePotentials", k)
        nc.createDimension("numEdges", el.shape[0])
        nc.createDimension("numEdgePotentials", k * k)
        nc.createDimension("numEndPointsPerEdge", 2)
        nc.createVariable("edgeList", "i4", ("numEdges", "numEndPointsPerEdge"))[:] = el
        nc.createVariable("edgePotentials", "f4", ("numEdges", "numEdgePotentials"))[:] = pw
        nc.createVariable("verticePotentials", "f4", ("numVertices", "numVerticePotentials"))[:] = un
        nc.graphType = b"Grid\x00"
        for name, v in [("numRows", rows), ("numCols", cols), ("numEdges", el.shape[0]), ("numVertices", rows * cols),
                        ("numVerticePotentials", k), ("numEdgePotentials", k * k)]:
            setattr(nc, name, np.int32(v))
    info = GF.read_info(f)
    assert (info.version, info.rows, info.cols, info.num_vertices, info.num_edges, info.k, info.k2) == \
        (version, rows, cols, rows * cols, el.shape[0], k, k * k)
    assert np.array_equal(GF.read_edges(f), el)
    assert np.array_equal(GF.read_unaries(f, negate=False), un)
    assert np.array_equal(GF.read_pairwise(f, negate=False), pw)
    # the loader's sign flip (Grid.cpp:547-560)
    assert np.array_equal(GF.read_unaries(f), -un)
    assert np.array_equal(GF.read_pairwise(f), -pw)


# ---- CDF-5 header assembled from the format specification ----

def _name5(s):
    b = s.encode()
    return struct.pack(">q", len(b)) + b + b"\0" * (-len(b) % 4)


def _cdf5_file(dims, atts, variables, data):
    """dims: [(name, len)], atts: [(name, int)], variables: [(name, type, [dimids], nbytes)];
    data: bytes per variable.  Counts, sizes, dimension ids and offsets are 8 bytes in CDF-5."""
    def header(begins):
        h = b"CDF\x05" + struct.pack(">q", 0)
        h += struct.pack(">iq", 0x0A, len(dims))
        for n, l in dims:
            h += _name5(n) + struct.pack(">q", l)
        h += struct.pack(">iq", 0x0C, len(atts))
        for n, v in atts:
            h += _name5(n) + struct.pack(">iq", 4, 1) + struct.pack(">i", v)  # NC_INT, one element
        h += struct.pack(">iq", 0x0B, len(variables))
        for (n, t, ids, nbytes), b in zip(variables, begins):
            h += _name5(n) + struct.pack(">q", len(ids)) + b"".join(struct.pack(">q", i) for i in ids)
            h += struct.pack(">iq", 0, 0)  # no variable attributes: ABSENT
            h += struct.pack(">iqq", t, nbytes, b)
        return h
    hlen = len(header([0] * len(variables)))
    begins, at = [], (hlen + 3) // 4 * 4
    for (_, _, _, nbytes) in variables:
        begins.append(at)
        at += nbytes
    blob = header(begins)
    blob += b"\0" * (begins[0] - len(blob)) if variables else b""
    return blob + b"".join(data)


def test_reads_handmade_cdf5(tmp_path):
    rows, cols, k = 3, 3, 2
    el, un, pw = _problem(rows, cols, k, seed=5)
    V, E = rows * cols, el.shape[0]
    dims = [("numVertices", V), ("numVerticePotentials", k), ("numEdges", E), ("numEdgePotentials", k * k), ("numEndPointsPerEdge", 2)]
    atts = [("numRows", rows), ("numCols", cols), ("numEdges", E), ("numVertices", V), ("numVerticePotentials", k), ("numEdgePotentials", k * k)]
    variables = [("edgeList", 4, [2, 4], E * 2 * 4), ("edgePotentials", 5, [2, 3], E * k * k * 4), ("verticePotentials", 5, [0, 1], V * k * 4)]
    data = [el.astype(">i4").tobytes(), pw.astype(">f4").tobytes(), un.astype(">f4").tobytes()]
    f = tmp_path / "h5.nc"
    f.write_bytes(_cdf5_file(dims, atts, variables, data))
    info = GF.read_info(f)
    assert (info.version, info.rows, info.cols, info.num_vertices, info.num_edges, info.k, info.k2) == (5, rows, cols, V, E, k, k * k)
    assert np.array_equal(GF.read_edges(f), el)
    assert np.array_equal(GF.read_pairwise(f, negate=False), pw)
    assert np.array_equal(GF.read_unaries(f, negate=False), un)


def test_cdf5_written_header_matches_the_specification(tmp_path):
    """What the writer puts before the data of a CDF-5 file is, byte for byte,
    the header the specification prescribes (with graphType as a text attribute)."""
    rows, cols, k = 2, 3, 2
    el, un, pw = _problem(rows, cols, k, seed=7)
    V, E = rows * cols, el.shape[0]
    f = tmp_path / "w5.nc"
    GF.write(f, V, el, un, pw, rows=rows, cols=cols, version=5)
    blob = f.read_bytes()
    assert blob[:4] == b"CDF\x05"
    info = GF.read_info(f)
    assert info.version == 5
    # walk the header with an independent parser
    pos = 4
    def q():
        nonlocal pos
        v = struct.unpack_from(">q", blob, pos)[0]; pos += 8; return v
    def i():
        nonlocal pos
        v = struct.unpack_from(">i", blob, pos)[0]; pos += 4; return v
    def name():
        nonlocal pos
        n = q(); s = blob[pos:pos + n]; pos += (n + 3) // 4 * 4; return s.decode()
    assert q() == 0  # numrecs
    assert i() == 0x0A and q() == 5
    dims = [(name(), q()) for _ in range(5)]
    assert dims == [("numVertices", V), ("numVerticePotentials", k), ("numEdges", E), ("numEdgePotentials", k * k), ("numEndPointsPerEdge", 2)]
    assert i() == 0x0C and q() == 7
    assert name() == "graphType" and i() == 2 and q() == 5
    assert blob[pos:pos + 8] == b"Grid\0\0\0\0"; pos += 8
    for nm, val in [("numRows", rows), ("numCols", cols), ("numEdges", E), ("numVertices", V), ("numVerticePotentials", k), ("numEdgePotentials", k * k)]:
        assert name() == nm and i() == 4 and q() == 1 and i() == val
    assert i() == 0x0B and q() == 3
    begins = {}
    for nm, typ, ids, nbytes in [("edgeList", 4, [2, 4], E * 8), ("edgePotentials", 5, [2, 3], E * k * k * 4), ("verticePotentials", 5, [0, 1], V * k * 4)]:
        assert name() == nm and q() == 2 and [q(), q()] == ids
        assert i() == 0 and q() == 0  # ABSENT attribute list
        assert i() == typ and q() == nbytes
        begins[nm] = q()
    assert begins["edgeList"] >= pos and begins["edgeList"] % 4 == 0
    assert begins["edgePotentials"] == begins["edgeList"] + E * 8
    assert begins["verticePotentials"] == begins["edgePotentials"] + E * k * k * 4
    assert len(blob) == begins["verticePotentials"] + V * k * 4
    assert np.array_equal(np.frombuffer(blob, ">i4", E * 2, begins["edgeList"]).reshape(E, 2), el)
    assert np.array_equal(np.frombuffer(blob, ">f4", V * k, begins["verticePotentials"]).reshape(V, k), un)


@pytest.mark.parametrize("version", [1, 2, 5])
def test_round_trip_and_hyperslabs(tmp_path, version):
    rows, cols, k = 7, 5, 4
    el, un, pw = _problem(rows, cols, k, seed=11)
    f = tmp_path / "rt.nc"
    GF.write(f, rows * cols, el, un, pw, rows=rows, cols=cols, version=version)
    assert GF.read_info(f).version == version
    assert np.array_equal(GF.read_edges(f), el)
    # slices as Grid::PartitionByEdge cuts them (Grid.cpp:446-454): E over 3 ranks
    E = el.shape[0]
    base, extra = divmod(E, 3)
    off = 0
    for r in range(3):
        n = base + (1 if r < extra else 0)
        assert np.array_equal(GF.read_edges(f, off, n), el[off:off + n])
        assert np.array_equal(GF.read_pairwise(f, off, n, negate=False), pw[off:off + n])
        off += n
    assert np.array_equal(GF.read_unaries(f, 3, 10, negate=False), un[3:13])
    assert GF.read_edges(f, E, 0).shape == (0, 2)  # a rank with no edges (Grid.cpp:481-484)


def test_shared_table_detection(tmp_path):
    rows, cols, k = 4, 4, 3
    el, un, _ = _problem(rows, cols, k)
    W = np.arange(k * k, dtype=np.float32).reshape(k, k)
    f = tmp_path / "shared.nc"
    GF.write(f, rows * cols, el, un, W, rows=rows, cols=cols)  # one table, replicated by the writer
    assert np.array_equal(GF.read_pairwise(f, negate=False), np.tile(W.reshape(1, -1), (el.shape[0], 1)))
    table, shared = GF.shared_table(f)
    assert shared and np.array_equal(table, -W)
    pw = np.tile(W.reshape(1, -1), (el.shape[0], 1))
    pw[-1, 2] += 1.0
    g = tmp_path / "not_shared.nc"
    GF.write(g, rows * cols, el, un, pw, rows=rows, cols=cols)
    assert GF.shared_table(g)[1] is False


def test_sizes_fall_back_to_dimensions(tmp_path):
    """A file without the size attributes still states its sizes as dimensions."""
    f = tmp_path / "noatt.nc"
    with netcdf_file(str(f), "w", version=1) as nc:
        nc.createDimension("numVertices", 4)
        nc.createDimension("numVerticePotentials", 2)
        nc.createDimension("numEdges", 3)
        nc.createDimension("numEdgePotentials", 4)
        nc.createDimension("numEndPointsPerEdge", 2)
        nc.createVariable("edgeList", "i4", ("numEdges", "numEndPointsPerEdge"))[:] = np.array([[0, 1], [1, 2], [2, 3]], dtype=np.int32)
        nc.createVariable("edgePotentials", "f4", ("numEdges", "numEdgePotentials"))[:] = np.zeros((3, 4), dtype=np.float32)
        nc.createVariable("verticePotentials", "f4", ("numVertices", "numVerticePotentials"))[:] = np.ones((4, 2), dtype=np.float32)
    info = GF.read_info(f)
    assert (info.rows, info.cols, info.num_vertices, info.num_edges, info.k, info.k2) == (-1, -1, 4, 3, 2, 4)


def test_converter(tmp_path):
    """Converter.cpp: a manifest of four raw files -> graph file, values unchanged."""
    V, E, k = 6, 7, 3
    rng = np.random.default_rng(2)
    el = rng.integers(0, V, size=(E, 2)).astype(np.int32)
    un = rng.standard_normal((V, k)).astype(np.float32)
    pw = rng.standard_normal((E, k * k)).astype(np.float32)
    (tmp_path / "info.txt").write_text("%d\n%d\n%d\n" % (V, E, k))  # Grid.cpp:1009-1013
    el.tofile(tmp_path / "edges.bin")
    un.tofile(tmp_path / "node.bin")
    pw.tofile(tmp_path / "edge.bin")
    manifest = tmp_path / "manifest.txt"
    manifest.write_text("\n".join(str(tmp_path / n) for n in ["info.txt", "edges.bin", "node.bin", "edge.bin"]) + "\n")
    out = tmp_path / "conv.nc"
    exe = os.path.join(ROOT, "admm_amd", "converter")
    r = subprocess.run([exe, str(manifest), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    info = GF.read_info(out)
    assert (info.rows, info.cols, info.num_vertices, info.num_edges, info.k, info.k2) == (-1, -1, V, E, k, k * k)
    assert np.array_equal(GF.read_edges(out), el)
    assert np.array_equal(GF.read_unaries(out, negate=False), un)
    assert np.array_equal(GF.read_pairwise(out, negate=False), pw)
    # usage message and exit status without arguments (Converter.cpp:12-15)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode != 0 and "<infile> <outfile>" in r.stderr
    # a raw file that is too short is an error, not zeros
    un[:2].tofile(tmp_path / "node.bin")
    with pytest.raises(GF.GraphFileError):
        GF.convert(manifest, tmp_path / "bad.nc")


def test_errors(tmp_path):
    with pytest.raises(GF.GraphFileError):
        GF.read_info(tmp_path / "missing.nc")
    junk = tmp_path / "junk.nc"
    junk.write_bytes(b"not a netcdf file at all")
    with pytest.raises(GF.GraphFileError) as e:
        GF.read_info(junk)
    assert e.value.code == -3
    hdf = tmp_path / "hdf.nc"
    hdf.write_bytes(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    with pytest.raises(GF.GraphFileError) as e:
        GF.read_info(hdf)
    assert "HDF5" in str(e.value)
    el, un, pw = _problem(3, 3, 2)
    f = tmp_path / "ok.nc"
    GF.write(f, 9, el, un, pw, rows=3, cols=3)
    with pytest.raises(GF.GraphFileError):
        GF.read_edges(f, 10, 5)  # past the end
    blob = f.read_bytes()
    cut = tmp_path / "cut.nc"
    cut.write_bytes(blob[:60])  # inside the header
    with pytest.raises(GF.GraphFileError):
        GF.read_info(cut)
    cut.write_bytes(blob[:-8])  # inside the last variable
    with pytest.raises(GF.GraphFileError):
        GF.read_unaries(cut)


def test_variable_shape_must_match_the_stated_sizes(tmp_path):
    """A caller sizes its buffers from the sizes the file states; a variable whose own
    dimensions disagree (here: three end points per edge) is refused, not read."""
    f = tmp_path / "odd.nc"
    with netcdf_file(str(f), "w", version=1) as nc:
        nc.createDimension("numVertices", 4)
        nc.createDimension("numVerticePotentials", 2)
        nc.createDimension("numEdges", 3)
        nc.createDimension("numEdgePotentials", 4)
        nc.createDimension("numEndPointsPerEdge", 3)
        nc.createVariable("edgeList", "i4", ("numEdges", "numEndPointsPerEdge"))[:] = np.zeros((3, 3), dtype=np.int32)
        nc.createVariable("edgePotentials", "f4", ("numEdges", "numEdgePotentials"))[:] = np.zeros((3, 4), dtype=np.float32)
        nc.createVariable("verticePotentials", "f4", ("numVertices", "numVerticePotentials"))[:] = np.ones((4, 2), dtype=np.float32)
        nc.numEdges = np.int32(5)  # the attribute disagrees with the dimension
    with pytest.raises(GF.GraphFileError) as e:
        GF.read_edges(f, 0, 3)
    assert e.value.code == -3
    with pytest.raises(GF.GraphFileError):
        GF.read_pairwise(f, 0, 3)
    assert np.array_equal(GF.read_unaries(f, negate=False), np.ones((4, 2), dtype=np.float32))


def test_reader_survives_truncation_and_corruption(tmp_path):
    """Every prefix of a valid file and a few hundred corrupted headers: an error or a
    result, never a fault."""
    el, un, pw = _problem(3, 4, 3, seed=21)
    good = tmp_path / "good.nc"
    bad = tmp_path / "bad.nc"
    rng = np.random.default_rng(5)
    for version in (1, 5):
        GF.write(good, 12, el, un, pw, rows=3, cols=4, version=version)
        blob = good.read_bytes()
        cases = [blob[:n] for n in range(0, min(len(blob), 520), 3)]
        for _ in range(300):
            b = bytearray(blob)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, min(len(b), 500)))] = int(rng.integers(0, 256))
            cases.append(bytes(b))
        for c in cases:
            bad.write_bytes(c)
            try:
                info = GF.read_info(bad)
                if not (0 <= info.num_edges <= 1000 and 0 <= info.num_vertices <= 1000 and 0 < info.k <= 64 and 0 < info.k2 <= 4096):
                    continue
                GF.read_edges(bad)
                GF.read_unaries(bad)
                GF.read_pairwise(bad)
                GF.shared_table(bad)
            except GF.GraphFileError:
                pass


def test_grid_convert_then_save_keeps_the_raw_values(tmp_path):
    """Converter.cpp's sequence through the host mirror: Convert(manifest), Save(file)."""
    import admm_amd as A
    V, E, k = 5, 6, 2
    rng = np.random.default_rng(3)
    el = rng.integers(0, V, size=(E, 2)).astype(np.int32)
    un = rng.standard_normal((V, k)).astype(np.float32)
    pw = rng.standard_normal((E, k * k)).astype(np.float32)
    (tmp_path / "info.txt").write_text("%d %d %d\n" % (V, E, k))
    el.tofile(tmp_path / "edges.bin"); un.tofile(tmp_path / "node.bin"); pw.tofile(tmp_path / "edge.bin")
    manifest = tmp_path / "m.txt"
    manifest.write_text("\n".join(str(tmp_path / n) for n in ["info.txt", "edges.bin", "node.bin", "edge.bin"]) + "\n")
    g = A.Grid()
    g.Convert(manifest)
    out = tmp_path / "out.nc"
    g.Save(out)
    assert np.array_equal(GF.read_edges(out), el)
    assert np.array_equal(GF.read_unaries(out, negate=False), un)
    assert np.array_equal(GF.read_pairwise(out, negate=False), pw)
    info = GF.read_info(out)
    assert (info.rows, info.cols, info.num_vertices, info.num_edges) == (-1, -1, V, E)


def test_a_strip_that_loaded_from_a_file_can_save(tmp_path):
    """A strip of several keeps no host copy of the unaries after Load(fname) (it reads its share
    when it uploads it); Save reads them back from the file, as every rank of the reference can
    save what it loaded (Grid.cpp:1049-1131)."""
    import admm_amd as A
    R, C, k = 7, 9, 3
    rng = np.random.default_rng(11)
    un = rng.standard_normal((R * C, k)).astype(np.float32)
    W = rng.standard_normal((k, k)).astype(np.float32)
    src = tmp_path / "grid.nc"
    GF.write(src, R * C, A.grid_edges(R, C), un, W, rows=R, cols=C)
    g = A.Grid(rank=1, nprocs=3)
    g.SetPartitioner(A.ROW)
    g.SetGlobalInfo(src)
    g.Load(src)
    assert g._unary is None
    out = tmp_path / "again.nc"
    g.Save(out)
    assert np.array_equal(GF.read_unaries(out, negate=False), un)
    assert np.array_equal(GF.read_edges(out), GF.read_edges(src))
    table, shared = GF.shared_table(out, negate=False)
    assert shared and np.array_equal(np.asarray(table).reshape(k, k), W)
    empty = A.Grid()
    empty.SetPartitioner(A.ROW)
    with pytest.raises(A.AdmmError, match="before Save"):
        empty.Save(tmp_path / "nothing.nc")

