"""Child process of tests/test_gpu_row_range.py: loads the library named by MARA_HIP_LIBRARY (a check build, csrc/row_check.hpp), runs
one kernel family over a list of configurations and prints one JSON line per configuration:
    {"family": ..., "config": ..., "n0": rows of the field (the largest member's for slabs / blocks), "cut": the sides are cuts of the fused 2-D step,
     "lo": smallest row index requested, "hi": largest, "status": status word}
usage: python tests/row_range_child.py <family>      family = euler2d | euler2d_fused | euler2d_fused_cuts | cloud | cloud_fused | cloud_fused_cuts | euler3d | binary"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mara3_amd
from mara3_amd import engine, setups
from mara3_amd import _lib as L

lib = mara3_amd.load_library()
FAMILY = {"euler2d": 0, "euler2d_fused": 1, "cloud": 2, "cloud_fused": 3, "euler3d_strict": 4, "euler3d_fast": 5, "binary_strict": 6, "binary_fast": 7}


def take(family):
    out = (C.c_int32 * 2)()
    L.check(lib.mh_debug_row_range(FAMILY[family], out, 1))
    return int(out[0]), int(out[1])


def report(family, config, n0, status, cut=False, read=None):
    lo, hi = take(read or family)
    print(json.dumps({"family": family, "config": config, "n0": n0, "cut": cut, "lo": lo, "hi": hi, "status": status}), flush=True)


def euler2d(fused):
    fam = "euler2d_fused" if fused else "euler2d"
    cases = [((67, 200), 7, None, "outflow"), ((250, 300), 0, None, "periodic"), ((129, 113), 16, None, "outflow"), ((40, 500), 2, None, "periodic"),
             ((64, 56), 0, None, "outflow"), ((600, 130), 0, (64, 4), "outflow"), ((600, 130), 32, (96, 8), "periodic"), ((9, 70), 3, None, "outflow")]
    for shape, chunk, tail, bc in cases:
        for arith in (("fast",) if fused else ("strict", "fast")):
            for riemann in ("hlle", "hllc"):
                if fused and tail is not None:
                    continue
                take(fam)
                u0 = setups.wave_ic(shape, 1.4, seed=3)
                s = engine.EulerCartSolver(shape, (1.0 / shape[0], 1.0 / shape[1]), 1.4, 1.5, riemann, 2, bc, arith=arith, chunk_rows=chunk, tail=tail,
                                           fuse=True if fused else False)
                s.upload(u0)
                s.step(2e-4, 3)
                st = s.status()
                s.close()
                report(fam, "%dx%d chunk %d tail %s %s %s %s" % (shape + (chunk, tail, bc, arith, riemann)), shape[0], st)
    if fused:
        # the planar kernel (a field without third momentum: the blast) marches the same rows
        for shape, chunk, bc in (((130, 250), 0, "outflow"), ((67, 200), 7, "periodic")):
            take(fam)
            u0 = setups.blast_ic(shape, 1.4, radius=0.3)
            s = engine.EulerCartSolver(shape, (1.0 / shape[0], 1.0 / shape[1]), 1.4, 1.5, "hllc", 2, bc, arith="fast", chunk_rows=chunk, fuse=True)
            s.upload(u0)
            assert s.is_planar()
            s.step(2e-4, 3)
            st = s.status()
            s.close()
            report(fam, "%dx%d chunk %d %s planar" % (shape + (chunk, bc)), shape[0], st)
    if not fused:
        # PCM and RK1 use the same row loop with fewer rows ahead
        for theta, rk in ((-1.0, 1), (1.5, 1), (-1.0, 2)):
            take(fam)
            shape = (50, 90)
            s = engine.EulerCartSolver(shape, (1.0 / 50, 1.0 / 90), 1.4, theta, "hlle", rk, "outflow", arith="strict", chunk_rows=6)
            s.upload(setups.wave_ic(shape, 1.4, seed=4))
            s.step(2e-4, 2)
            st = s.status()
            s.close()
            report(fam, "50x90 chunk 6 theta %g rk %d" % (theta, rk), 50, st)


def euler2d_fused_cuts():
    """the fused step across the cuts of a slab decomposition: MH_BC_EXTERNAL sides, two row segments per edge launch (slab.hip)"""
    from mara3_amd.slab import NativeSlabGroup
    os.environ["MH_SLAB_FUSED_CUTS"] = "1"
    for shape, world, bc, chunk in (((48, 200), 3, "outflow", 0), ((64, 130), 4, "periodic", 0), ((60, 70), 2, "outflow", 5), ((36, 300), 3, "periodic", 2)):
        take("euler2d_fused")
        take("euler2d")
        g = NativeSlabGroup(shape, (1.0 / shape[0], 1.0 / shape[1]), 1.4, 1.5, "hllc", 2, bc, world=world, arith="fast", chunk_rows=chunk)
        g.upload(setups.wave_ic(shape, 1.4, seed=5))
        g.step(2e-4, 3)
        g.synchronize()
        st = g.status()[0]
        rows = max(b - a for a, b in g.rows)
        g.close()
        report("euler2d_fused_cuts", "%dx%d in %d slabs %s chunk %d" % (shape + (world, bc, chunk)), rows, st, cut=True, read="euler2d_fused")


def cloud_state(nr, nq):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_cloud_fused import smooth_cloud_state
    return smooth_cloud_state(engine, nr, nq, seed=nr + nq)


def cloud(fused):
    fam = "cloud_fused" if fused else "cloud"
    for nr, nq, chunk, tail in ((130, 250, 0, None), (64, 117, 9, None), (33, 57, 2, None), (200, 64, 3, None), (600, 70, 0, (64, 4)), (5, 40, 4, None)):
        for arith in (("fast",) if fused else ("strict", "fast")):
            if fused and tail is not None:
                continue
            take(fam)
            rv, qv, u0, inflow, dt = cloud_state(nr, nq)
            s = engine.CloudSolver(rv, qv, 2, 1.2, 0.0, arith=arith, chunk_rows=chunk, tail=tail, fuse=True if fused else False)
            s.upload(u0)
            for n in range(3):
                s.set_inflow(inflow[n])
                s.step(dt, 1)
            st = s.status()
            s.close()
            report(fam, "%dx%d chunk %d tail %s %s" % (nr, nq, chunk, tail, arith), nr, st)
    if not fused:
        # radial slabs: MH_BC_EXTERNAL sides with the stored two ghost rows
        from mara3_amd.slab import NativeSlabGroup
        for world in (2, 3):
            take(fam)
            rv, qv, u0, inflow, dt = cloud_state(70, 90)
            g = NativeSlabGroup(r_vertices=rv, q_vertices=qv, rk_order=2, plm_theta=1.2, world=world, arith="fast", temperature_floor=0.0)
            g.upload(u0)
            for n in range(3):
                g.set_inflow(inflow[n])
                g.step(dt, 1)
            g.synchronize()
            st = g.status()[0]
            rows = max(b - a for a, b in g.rows)
            g.close()
            report(fam, "70x90 in %d radial slabs" % world, rows, st)


def cloud_fused_cuts():
    """round 5: the one-launch `cloud` step across radial cuts - MH_BC_EXTERNAL radial sides with FOUR stored rows of the neighbour, two row
    segments per edge launch, the nozzle rows on the slab that owns row 0 only (cloud_fused.hip, slab.hip)"""
    from mara3_amd.slab import NativeSlabGroup
    os.environ["MH_SLAB_FUSED_CUTS"] = "1"
    for nr, nq, world, chunk in ((70, 90, 2, 0), (97, 250, 3, 0), (61, 64, 4, 5), (50, 300, 3, 2), (48, 40, 4, 0)):
        take("cloud_fused")
        take("cloud")
        rv, qv, u0, inflow, dt = cloud_state(nr, nq)
        g = NativeSlabGroup(r_vertices=rv, q_vertices=qv, rk_order=2, plm_theta=1.2, world=world, arith="fast", temperature_floor=0.0, chunk_rows=chunk)
        assert g.launches_per_step() == [1] * world
        g.upload(u0)
        for n in range(3):
            g.set_inflow(inflow[n])
            g.step(dt, 1)
        g.synchronize()
        st = g.status()[0]
        rows = max(b - a for a, b in g.rows)
        g.close()
        report("cloud_fused_cuts", "%dx%d in %d radial slabs chunk %d" % (nr, nq, world, chunk), rows, st, cut=True, read="cloud_fused")


def euler3d():
    from mara3_amd.block import NativeBlockGroup
    for arith in ("strict", "fast"):
        fam = "euler3d_" + arith
        for shape, chunk, bc in (((24, 20, 70), 0, "outflow"), ((33, 9, 61), 5, "periodic"), ((10, 17, 130), 3, "outflow")):
            for riemann in ("hlle", "hllc"):
                take(fam)
                s = engine.EulerCartSolver(shape, tuple(1.0 / n for n in shape), 1.4, 1.5, riemann, 2, bc, arith=arith, chunk_rows=chunk)
                s.upload(setups.wave_ic(shape, 1.4, seed=6))
                s.step(2e-4, 2)
                st = s.status()
                s.close()
                report(fam, "%dx%dx%d chunk %d %s %s" % (shape + (chunk, bc, riemann)), shape[0], st)
        # the boxes of the block stepper: shell + interior of every block of a (2,2,2) and a (1,2,3) decomposition
        for shape, world in (((40, 36, 130), 8), ((30, 40, 150), 6)):
            take(fam)
            g = NativeBlockGroup(shape, tuple(1.0 / max(shape) for _ in shape), 1.4, 1.5, "hlle", 2, "outflow", world=world, arith=arith)
            g.upload(setups.wave_ic(shape, 1.4, seed=7))
            g.step(2e-4, 2)
            g.synchronize()
            st = g.status()[0]
            rows = max(m.count[0] for m in g.members)
            g.close()
            report(fam, "%dx%dx%d in %d blocks" % (shape + (world,)), rows, st)


def binary():
    """the uniform-depth mesh as one periodic grid (its own ghost rows: the periodic images) and as bands of block rows (ghost rows from the
    neighbours, edges first or not): chunks from 2 rows to the whole band, both forms of the conserved state, RK1 / RK2"""
    from mara3_amd import binary as B
    for arith in ("strict", "fast"):
        fam = "binary_" + arith
        for overrides, chunk in ((dict(depth=2, block_size=16), 0), (dict(depth=2, block_size=16, fixed_dt=1, rk_order=1), 5), (dict(depth=3, block_size=8), 2),
                                 (dict(depth=2, block_size=16, conserve_linear_p=0), 7), (dict(depth=2, block_size=32, mass_ratio=0.5, eccentricity=0.3), 64)):
            take(fam)
            cfg = B.config(**overrides)
            s = B.BinarySolver(cfg, arith=arith, chunk_rows=chunk)
            st = s.next(3)
            n = B.grid_size(cfg)
            s.close()
            report(fam, "%s chunk %d" % (sorted(overrides.items()), chunk), n, st)
        for world, edge in ((2, None), (3, None), (4, -1)):
            take(fam)
            cfg = B.config(depth=2, block_size=16)
            g = B.BinaryBandGroup(cfg, world=world, arith=arith, edge_rows=edge)
            st = g.next(3)
            rows = max(b - a for a, b in g.rows)
            g.close()
            report(fam, "64x64 in %d bands, edge_rows %s" % (world, edge), rows, st)


if __name__ == "__main__":
    {"binary": binary, "euler2d": lambda: euler2d(False), "euler2d_fused": lambda: euler2d(True), "euler2d_fused_cuts": euler2d_fused_cuts,
     "cloud": lambda: cloud(False), "cloud_fused": lambda: cloud(True), "cloud_fused_cuts": cloud_fused_cuts, "euler3d": euler3d}[sys.argv[1]]()
