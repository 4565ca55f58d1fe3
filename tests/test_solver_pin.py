"""What anchors `midpoint` / `rk4` without the dependency (torchdiffeq==0.1 is absent offline; reference call site
vihds/ode.py:79-81, solver list tests/test_ode_solvers.py:43-92).  CPU tests, float64, on the oracle and on the constants the
kernels are compiled with:

  (a) observed order of convergence under grid refinement on the dr_constant right-hand side (models/dr_constant.py:77-112)
      with the reference-generated fixture's parameters: 4 (rk4), 2 (midpoint), 2 (modeuler / modeulerwhile), 1 (euler);
  (b) rk4 / midpoint ON THE FIXTURE'S OWN GRID (the plate reader's, non-uniform) against the tight-tolerance adaptive solution
      (dopri5, rtol 1e-10): they differ by the scheme's truncation error, estimated by Richardson from the same grid halved;
  (c) the order conditions of a 4-stage 4th-order method (and the 2nd-order ones) evaluated on Rk<SOLVER>::a, b, c as
      csrc/vihds_dr_scan.hpp holds them -- a host harness (tests/micro/tableau_dump.hip) prints the header's own constexpr
      functions -- and one step of the oracle's `_rk4_38_step` / `_midpoint_step` / modified Euler against a generic explicit
      Runge-Kutta step driven by those printed numbers: the oracle and the kernels hold ONE tableau.

What this does NOT anchor: that torchdiffeq 0.1's `rk4` is this 3/8-rule member of the 4th-order family rather than the
classic one (both pass every check here), and its choice grid == t with outputs at the grid points.  Those two facts are
recalled from the dependency's published source (rk_common.rk4_alt_step_func, FixedGridODESolver.integrate) and stay
"parity unpinned" in the oracle's header."""
import json
import math
import os
import shutil
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from fixture_util import Fixture  # noqa: E402
from oracle import vihds_oracle as O  # noqa: E402


def _problem(dtype=torch.float64):
    fx = Fixture("dr_constant_icml_tiny_modeuler")
    th = {k: v.to(dtype) for k, v in fx.theta_dict().items()}
    rhs, x0 = O.make_dr_constant(th, fx.t("inputs").to(dtype))
    return fx, rhs, x0.to(dtype)


def _uniform(t0, t1, n):
    return torch.linspace(float(t0), float(t1), n + 1, dtype=torch.float64)


@pytest.mark.parametrize("solver,order", [("rk4", 4), ("midpoint", 2), ("modeuler", 2), ("modeulerwhile", 2), ("euler", 1)])
def test_observed_order_of_convergence(solver, order):
    fx, rhs, x0 = _problem()
    t = fx.t("times").double()
    t0, t1 = t[0], t[-1]
    with torch.no_grad():
        exact = O.SOLVERS["rk4"](rhs, x0, _uniform(t0, t1, 85 * 64))[-1]
        errs = []
        for n in (85, 170, 340, 680):
            y = O.SOLVERS[solver](rhs, x0, _uniform(t0, t1, n))[-1]
            errs.append(float(((y - exact).abs().amax((0, 1)) / exact.abs().amax((0, 1))).max()))
    rates = [math.log2(errs[k] / errs[k + 1]) for k in range(3)]
    # the asymptotic regime: the two finest refinements within 0.35 of the nominal order
    assert all(abs(r - order) < 0.35 for r in rates[1:]), (solver, errs, rates)
    assert errs[0] < {4: 1e-4, 2: 3e-2, 1: 3e-1}[order], (solver, errs)


@pytest.mark.parametrize("solver,order", [("rk4", 4), ("midpoint", 2)])
def test_fixed_grid_scheme_against_tight_adaptive_solution_on_the_plate_grid(solver, order):
    fx, rhs, x0 = _problem()
    t = fx.t("times").double()
    assert float((t[1:] - t[:-1]).std()) > 0  # (the plate reader's grid: not uniform)
    half = torch.stack([t[:-1], 0.5 * (t[:-1] + t[1:])], 1).reshape(-1)
    half = torch.cat([half, t[-1:]])
    with torch.no_grad():
        y = O.SOLVERS[solver](rhs, x0, t)                  # outputs at the grid points
        y2 = O.SOLVERS[solver](rhs, x0, half)[::2]         # the same grid with every step halved
        ref, n_acc, _ = O.odeint_adaptive("dopri5", rhs, x0, t, rtol=1e-10, atol=1e-12)
    scale = ref.abs().amax((0, 1, 2))                      # per species
    err = ((y - ref).abs().amax((0, 1, 2)) / scale).max()
    err2 = ((y2 - ref).abs().amax((0, 1, 2)) / scale).max()
    rich = ((y - y2).abs().amax((0, 1, 2)) / scale).max() / (1.0 - 2.0 ** -order)   # Richardson estimate of y's error
    assert n_acc > 85
    assert float(err) < {4: 2e-3, 2: 1e-1}[order], float(err)   # (rk4: 7.9e-4 in cfp, the stiffest sampled parameters)
    assert 0.5 < float(err / rich) < 2.0, (float(err), float(rich))    # the distance to the truth IS the truncation error
    assert float(err2) < float(err) * 2.0 ** -(order - 0.5)              # ... and halving the steps shrinks it by 2^order


def _header_tableaux():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available: the header's tableau cannot be printed")
    exe = os.path.join(ROOT, "tests", "micro", "bin", "tableau_dump")
    src = os.path.join(ROOT, "tests", "micro", "tableau_dump.hip")
    hdr = os.path.join(ROOT, "vi-hds_amd", "csrc", "vihds_dr_scan.hpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O1", "-w", "-o", exe, src])
    return json.loads(subprocess.check_output([exe]).decode())


def test_order_conditions_on_the_header_constants():
    tabs = _header_tableaux()
    for name, tab in tabs.items():
        a, b, c = tab["a"], tab["b"], tab["c"]
        ns = tab["ns"]
        tol = 1e-6  # (the header holds floats)
        for i in range(ns):
            assert abs(sum(a[i]) - c[i]) < tol, (name, "row sums")
            assert all(a[i][j] == 0 for j in range(i, ns)), (name, "explicit")
        assert abs(sum(b) - 1) < tol, name
        if name == "euler":
            continue
        assert abs(sum(b[i] * c[i] for i in range(ns)) - 0.5) < tol, name          # order 2
        if name != "rk4":
            continue
        r = range(ns)
        assert abs(sum(b[i] * c[i] ** 2 for i in r) - 1 / 3) < tol                   # order 3
        assert abs(sum(b[i] * a[i][j] * c[j] for i in r for j in r) - 1 / 6) < tol
        assert abs(sum(b[i] * c[i] ** 3 for i in r) - 1 / 4) < tol                   # order 4
        assert abs(sum(b[i] * c[i] * a[i][j] * c[j] for i in r for j in r) - 1 / 8) < tol
        assert abs(sum(b[i] * a[i][j] * c[j] ** 2 for i in r for j in r) - 1 / 12) < tol
        assert abs(sum(b[i] * a[i][j] * a[j][k] * c[k] for i in r for j in r for k in r) - 1 / 24) < tol
        # the 3/8 rule, not the classic tableau
        assert [round(8 * v) for v in b] == [1, 3, 3, 1] and abs(c[1] - 1 / 3) < tol and abs(c[2] - 2 / 3) < tol
    assert tabs["modeuler"]["fixed_h"] and not tabs["modeulerwhile"]["fixed_h"]        # solvers.py:12 vs :21


@pytest.mark.parametrize("solver", ["rk4", "midpoint", "modeulerwhile", "euler"])
def test_oracle_steps_are_the_header_tableau(solver):
    """One step of the oracle's hand-written step function == a generic explicit RK step with the header's numbers."""
    tab = _header_tableaux()[solver]
    fx, rhs, x0 = _problem()
    t = fx.t("times").double()
    t0, dt = t[3], t[4] - t[3]
    with torch.no_grad():
        y = O.SOLVERS["rk4"](rhs, x0, t[:4])[-1]  # a state away from the initial one
        got = O.SOLVERS[solver](rhs, y, torch.stack([t0, t0 + dt]))[-1]
        ks = []
        for s in range(tab["ns"]):
            Y = y.clone()
            for q in range(s):
                if tab["a"][s][q] != 0:
                    Y = Y + dt * round(tab["a"][s][q] * 24) / 24 * ks[q]   # (the header's floats are n/24 exactly rounded)
            ks.append(rhs(t0 + round(tab["c"][s] * 24) / 24 * dt, Y))
        want = y + dt * sum(round(tab["b"][s] * 24) / 24 * ks[s] for s in range(tab["ns"]))
    assert float((got - want).abs().max() / want.abs().max()) < 1e-13
