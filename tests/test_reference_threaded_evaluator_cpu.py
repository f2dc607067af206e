"""The reference's own threaded evaluator in the reference-composed drivers (oracle/_ref, built from /root/reference where it lies):
`mara::evaluate_on<N>()` (src/app_parallel.hpp:72-103) in the places where upstream pipes `| evaluate` (src/subprog_cloud.cpp:525-533, :582)
must leave every bit of the result where nd::to_shared() puts it - that is upstream's own claim ("a drop-in replacement for nd::to_shared()"),
and what lets bench.py / bench_configs.py quote this path, on the host cores, as `cpu_reference` (kind "reference", cores > 1).
Runs wherever the prebuilt drivers are (this container and the GPU box); skipped where they are not."""
import os
import subprocess
import numpy as np
import pytest
from conftest import ROOT, golden

EULER = os.path.join(ROOT, "oracle", "_ref", "euler_cart_ref")
CLOUD = os.path.join(ROOT, "oracle", "_ref", "cloud_ref")


def hx(x):
    return float(x).hex()


@pytest.mark.skipif(not os.path.exists(EULER), reason="oracle/_ref/euler_cart_ref not built (needs /root/reference)")
@pytest.mark.parametrize("case,rank", [("euler2d_blast64_plm15_rk2", 2), ("euler3d_blast24_plm15_rk2", 3)])
def test_euler_step_through_the_threaded_evaluator_reproduces_the_golden_steps(tmp_path, case, rank):
    """the golden fixture was written by the materialising composition through nd::to_shared(); the lazy composition through
    evaluate_on<1 | 2 | 8>() lands on the same bits"""
    g = golden(case)
    nsteps = int(max(g["nsteps"]))
    u0, un = g["u0"], g["u_%d" % nsteps]
    shape = u0.shape[:-1]
    n = list(shape) + [1] * (3 - rank)
    dl = list(g["dl"]) + [1.0] * (3 - rank)
    fin, fout = str(tmp_path / "in"), str(tmp_path / "out")
    np.ascontiguousarray(u0).tofile(fin)
    for threads in ("-1", "1", "2", "8"):
        args = [EULER, str(rank), str(n[0]), str(n[1]), str(n[2]), hx(g["gamma"]), hx(g["theta"]), str(int(g["rk"])), "0", hx(g["dt"]),
                hx(dl[0]), hx(dl[1]), hx(dl[2]), str(nsteps), fin, fout, threads]
        subprocess.check_call(args)
        got = np.fromfile(fout).reshape(un.shape)
        assert np.array_equal(got.view(np.uint64), np.ascontiguousarray(un).view(np.uint64)), (case, threads, np.abs(got - un).max())


@pytest.mark.skipif(not os.path.exists(CLOUD), reason="oracle/_ref/cloud_ref not built (needs /root/reference)")
def test_cloud_step_through_the_threaded_evaluator_is_bit_identical(tmp_path):
    outs = {}
    for threads in ("", "1", "8", "12"):
        prefix = str(tmp_path / ("c" + threads))
        subprocess.check_call([CLOUD, "48", "1", "2", "2", "1.2", "3", prefix] + ([threads] if threads else []))
        outs[threads] = {f: open(os.path.join(tmp_path, f), "rb").read() for f in sorted(os.listdir(tmp_path)) if f.startswith("c" + threads + ".")}
    names = sorted(n.split(".", 1)[1] for n in outs[""])
    assert names
    for threads in ("1", "8", "12"):
        for n in names:
            assert outs[threads]["c%s.%s" % (threads, n)] == outs[""]["c." + n], (threads, n)
