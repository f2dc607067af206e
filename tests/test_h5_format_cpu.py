"""HDF5 format parity, pinned in BOTH directions against the reference's own writers and readers.

oracle/_ref/h5_ref calls mara::write / mara::read, write_schedule / read_schedule, write_config / read_config, write_tree /
read_tree from the reference's headers where they lie (app_serialize.hpp:69-162,195-347, app_serialize_tree.hpp:73-177) in the
sequences of subprog_sedov.cpp:329-346,486-495 and subprog_cloud.cpp:590-609,758-767; mara3_amd/host/h5_tool does the same
through the hosts' own checkpoint layer (h5_checkpoint.hpp). For the same content:

    reference-written file == mara_hip-written file          h5dump of both, types, dataspaces and every value
    the reference's readers return the bits mara_hip wrote   (its readers accept these files)
    mara_hip's readers return the bits the reference wrote   (restart=<reference-written file>)

Out of reach (and said so in DESIGN.md): the compound specialisations of subprog_binary_io.cpp - that translation unit needs the
generated app_compile_opts.hpp. The orbital-element compounds are pinned here through the reference's own struct types and its
h5::Datatype::compound machinery with the member lists restated by name."""
import math
import os
import struct
import subprocess

import numpy as np
import pytest
from conftest import ROOT

H5DUMP = "/opt/conda/bin/h5dump"
REF = os.path.join(ROOT, "oracle", "_ref", "h5_ref")
TOOL = os.path.join(ROOT, "mara3_amd", "host", "h5_tool")
REF_ENV = dict(os.environ, LD_LIBRARY_PATH="/usr/lib/x86_64-linux-gnu")      # the system libstdc++ in front of conda's (oracle/Makefile)


def build():
    if os.path.isdir("/root/reference/src") and not os.path.exists(REF):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/h5_ref"])
    if not os.path.exists(TOOL):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "mara3_amd", "host"), "h5_tool"])
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/h5_ref is built where the reference tree is present")
    if not os.path.exists(H5DUMP):
        pytest.skip("needs the HDF5 tools of the image")


def ref(*args):
    subprocess.run([REF] + list(args), check=True, env=REF_ENV, capture_output=True, text=True)


def tool(*args):
    p = subprocess.run([TOOL] + list(args), capture_output=True, text=True)
    if p.returncode == 77:
        pytest.skip("libhdf5 cannot be loaded here")
    assert p.returncode == 0, p.stderr


def dump(path):
    """h5dump of the whole file with round-trip float formatting, without the line that names the file"""
    out = subprocess.run([H5DUMP, "-m", "%.17g", path], check=True, capture_output=True, text=True).stdout
    return out.split("\n", 1)[1]


def make_spec(kind, seed):
    rng = np.random.default_rng(seed)
    special = [0.0, -0.0, 5e-324, -1.7976931348623157e308, math.pi, 1.0 / 3.0]
    if kind == "sedov":
        n0, n1, rank = 11, 1, 1
        arrays = [("vertices", 10.0 ** np.linspace(-0.5, 2.0, n0 + 1))]
    else:
        n0, n1, rank = 6, 5, 2
        arrays = [("radial_vertices", 10.0 ** np.linspace(0.0, 1.0, n0 + 1)), ("polar_vertices", np.linspace(0.0, math.pi, n1 + 1))]
    u = rng.standard_normal(n0 * n1 * 5) * 10.0 ** rng.integers(-30, 30, n0 * n1 * 5)
    u[:len(special)] = special
    tasks = [("write_checkpoint", 3, 0.75), ("write_diagnostics", 0, 0.0), ("write_time_series", 1207, 1.0e-3 * math.e)]
    config = [("outdir", "s", "data/run 7"), ("restart", "s", ""), ("nr", "i", 256), ("rk_order", "i", -2), ("tfinal", "d", 1.5),
              ("cfl_number", "d", 0.4), ("plm_theta", "d", 1.2000000000000002), ("a_long_key_name_for_an_item", "s", "x")]
    return dict(kind=kind, time=0.1 + seed, iteration=(1234567, 1) if kind == "sedov" else (7, 2), arrays=arrays, rank=rank, n0=n0, n1=n1,
                conserved=u, tasks=tasks, config=config)


def write_spec(spec, path):
    hx = lambda x: float(x).hex()
    with open(path, "w") as f:
        f.write("kind %s\ntime %s\niteration %d %d\n" % (spec["kind"], hx(spec["time"]), spec["iteration"][0], spec["iteration"][1]))
        for name, v in spec["arrays"]:
            f.write("array %s %d %s\n" % (name, len(v), " ".join(hx(x) for x in v)))
        f.write("conserved %d %d %d %s\n" % (spec["rank"], spec["n0"], spec["n1"], " ".join(hx(x) for x in spec["conserved"])))
        for name, num, last in spec["tasks"]:
            f.write("task %s %d %s\n" % (name, num, hx(last)))
        for key, t, v in spec["config"]:
            f.write("config %s %s %s\n" % (key, t, ("%d %s" % (len(v), v)) if t == "s" else (hx(v) if t == "d" else str(v))))


def bits(x):
    return struct.pack("d", float(x))


def read_spec(path):
    """a spec file as comparable data: every double by its bits"""
    out = dict(arrays=[], tasks=[], config=[])
    for line in open(path).read().split("\n"):
        if not line:
            continue
        w = line.split(" ")
        if w[0] == "kind":
            out["kind"] = w[1]
        elif w[0] == "time":
            out["time"] = bits(float.fromhex(w[1]))
        elif w[0] == "iteration":
            out["iteration"] = (int(w[1]), int(w[2]))
        elif w[0] == "array":
            out["arrays"].append((w[1], int(w[2]), [bits(float.fromhex(x)) for x in w[3:]]))
        elif w[0] == "conserved":
            out["shape"] = tuple(int(x) for x in w[1:4])
            out["conserved"] = [bits(float.fromhex(x)) for x in w[4:]]
        elif w[0] == "task":
            out["tasks"].append((w[1], int(w[2]), bits(float.fromhex(w[3]))))
        elif w[0] == "config":
            if w[2] == "s":
                n = int(w[3])
                value = line.split(" ", 4)[4] if n else ""
                assert len(value) == n
            else:
                value = int(w[3]) if w[2] == "i" else bits(float.fromhex(w[3]))
            out["config"].append((w[1], w[2], value))
    out["tasks"].sort()
    out["config"].sort()
    return out


@pytest.mark.parametrize("kind", ["sedov", "cloud"])
def test_checkpoints_written_by_the_reference_and_by_mara_hip_are_the_same_file_and_read_back_the_same(tmp_path, kind):
    build()
    d = str(tmp_path)
    spec = os.path.join(d, "spec.txt")
    write_spec(make_spec(kind, 3), spec)
    want = read_spec(spec)
    ref("write", spec, os.path.join(d, "ref.h5"))
    tool("write", spec, os.path.join(d, "ours.h5"))
    # one layout, one set of types, one set of values
    a, b = dump(os.path.join(d, "ref.h5")), dump(os.path.join(d, "ours.h5"))
    assert a == b
    assert "H5T_ARRAY { [5] H5T_IEEE_F64LE }" in a and "H5T_ARRAY { [2] H5T_STD_I32LE }" in a and "STRSIZE 10" in a
    # each side's reader on each side's file: all four give back the bits that went in
    for reader, name in ((ref, "ref.h5"), (ref, "ours.h5"), (tool, "ref.h5"), (tool, "ours.h5")):
        out = os.path.join(d, "back.txt")
        reader("read", kind, os.path.join(d, name), out)
        got = read_spec(out)
        if reader is ref:
            # upstream quirk, kept visible: an empty string is stored with size 1 (core_hdf5.hpp:474-477) and the reference's reader hands
            # back that one NUL character; the hosts' reader strips trailing NULs (an empty `restart=` stays empty after a restart)
            assert ("restart", "s", "\x00") in got["config"]
            got["config"] = sorted((k, t, "" if (t == "s" and v == "\x00") else v) for k, t, v in got["config"])
        assert got == want, (reader.__name__, name)


def test_tree_datasets_and_orbital_element_compounds_are_the_same_files_both_ways(tmp_path):
    build()
    d = str(tmp_path)
    p = lambda n: os.path.join(d, n)
    ref("tree_write", p("tree_ref.h5"))
    tool("tree_write", p("tree_ours.h5"))
    a = dump(p("tree_ref.h5"))
    assert a == dump(p("tree_ours.h5"))
    for name in ('"1:0-0"', '"2:2-3"', '"3:7-6"', '"4:14-15"', '"4:15-15"'):       # format_tree_index: zero-padded from level 4 on
        assert "DATASET " + name in a
    assert a.count("DATASET ") == 13 and "H5T_ARRAY { [3] H5T_IEEE_F64LE }" in a and "( 4, 4 )" in a
    ref("tree_read", p("tree_ours.h5"), p("t1.txt"))
    ref("tree_read", p("tree_ref.h5"), p("t2.txt"))
    tool("tree_read", p("tree_ref.h5"), p("t3.txt"))
    tool("tree_read", p("tree_ours.h5"), p("t4.txt"))
    trees = [sorted(open(p("t%d.txt" % k)).read().split("\n")) for k in (1, 2, 3, 4)]
    assert trees[0] == trees[1] == trees[2] == trees[3] and len(trees[0]) == 14
    ref("elements_write", p("el_ref.h5"))
    tool("elements_write", p("el_ours.h5"))
    e = dump(p("el_ref.h5"))
    assert e == dump(p("el_ours.h5"))
    assert 'H5T_IEEE_F64LE "eccentricity";\n            } "elements";' in e
    ref("elements_read", p("el_ours.h5"), p("e1.txt"))
    tool("elements_read", p("el_ref.h5"), p("e2.txt"))
    assert open(p("e1.txt")).read() == open(p("e2.txt")).read() and len(open(p("e1.txt")).read().split()) == 14
