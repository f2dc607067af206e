"""Test-only stand-in for the HIP stage: computes one stage of a slab (device layout [n0+4][5][n1], ghost
rows valid) with the plain-C oracle, so that the slab/exchange logic of mara3_amd.slab can run on CPU."""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


class OracleStage:
    def __init__(self, dl, gamma, theta, riemann, periodic, stepper_ref):
        import mara_oracle
        self.o = mara_oracle
        self.dl, self.gamma, self.theta = dl, gamma, theta
        self.kind = mara_oracle.RIEMANN_HLLC if riemann == "hllc" else mara_oracle.RIEMANN_HLLE
        self.bc = mara_oracle.BC_PERIODIC if periodic else mara_oracle.BC_OUTFLOW
        self.stepper_ref = stepper_ref      # list holding the stepper (for its desc: which ghost rows are physical)

    def __call__(self, u_in, u_base, u_out, dt, weight, row_ranges):
        st = self.stepper_ref[0]
        n0 = st.n0
        ext = u_in.numpy().transpose(0, 2, 1).copy()                   # AoS [n0+4][n1][5] incl. ghost rows
        adv = self.o.euler_cart_advance(ext, self.dl, dt, self.gamma, self.theta, self.kind, self.bc)
        new = adv[2:n0 + 2]                                            # rows whose 5-point stencil is inside `ext`
        if weight != 1.0:
            base = u_base.numpy()[2:n0 + 2].transpose(0, 2, 1)
            new = base * (1.0 - weight) + new * weight
        out = u_out.numpy()
        for a, b in row_ranges:
            out[2 + a:2 + b] = new[a:b].transpose(0, 2, 1)
        st.fill_ghosts_physical_only(u_out)


class OracleCloudStage:
    """`cloud` stage of one radial slab with the oracle: the slab plus its two ghost rows per cut side is advanced as a small
    domain of its own (global vertices of those rows); rows within two of a cut are discarded - they are exactly the ghost
    rows - so the oracle's physical boundary treatment at the artificial ends never reaches a kept row."""

    def __init__(self, rv, qv, theta, tfloor, stepper_ref):
        import mara_oracle
        self.o = mara_oracle
        self.rv, self.qv, self.theta, self.tfloor = rv, qv, theta, tfloor
        self.stepper_ref = stepper_ref

    def __call__(self, u_in, u_base, u_out, dt, weight, row_ranges):
        st = self.stepper_ref[0]
        n0 = st.n0
        lo = 2 if st.desc.bc_lo0 == 2 else 0          # MH_BC_EXTERNAL
        hi = 2 if st.desc.bc_hi0 == 2 else 0
        ext = u_in.numpy()[2 - lo:n0 + 2 + hi].transpose(0, 2, 1).copy()
        rv = self.rv[st.row0 - lo:st.row1 + hi + 1]
        inflow = st.inflow.numpy().T.copy()[None]
        adv, status = self.o.cloud_run(ext, rv, self.qv, inflow, dt, 1, rk=1, theta=self.theta, tfloor=self.tfloor)
        assert status == 0
        new = adv[lo:lo + n0]
        if weight != 1.0:
            base = u_base.numpy()[2:n0 + 2].transpose(0, 2, 1)
            new = base * (1.0 - weight) + new * weight
        out = u_out.numpy()
        for a, b in row_ranges:
            out[2 + a:2 + b] = new[a:b].transpose(0, 2, 1)
