#!/usr/bin/env python3
"""Extracts the golden vectors of the hot path from the reference's committed logs.

Run in the build container (where /root/reference is mounted):

    python tests/golden/make_golden.py

Writes ``tests/golden/reference_logs.json`` (numbers only: per run, per cycle, the values the
reference prints around LaplaceProblem::solve, src/step-50.cc:946-952, 1009-1014, 1086,
1409-1418, 1460, 1532-1540, plus the ``Solve`` timer rows) and copies the small LAMMPS input
files the reference's tests use (data, not source).  Nothing here is executed from the
reference; the logs are read as text.
"""
import json
import os
import re
import shutil
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

LOGS = {
    "tests/gaussian-charges.mpirun=1": "tests/gaussian-charges.mpirun=1.output",
    "tests/gaussian-charges.mpirun=3": "tests/gaussian-charges.mpirun=3.output",
    "tests/gaussian-charges.mpirun=7": "tests/gaussian-charges.mpirun=7.output",
    "tests/test_with_optimal_parameters.mpirun=1": "tests/test_with_optimal_parameters.mpirun=1.output",
    "tests/step-16.mpirun=1": "tests/step-16.mpirun=1.output",
    "tests_2D/step-16.mpirun=1": "tests_2D/step-16.mpirun=1.output",
    "tests_2D/gaussian-charges.mpirun=1": "tests_2D/gaussian-charges.mpirun=1.output",
    "tests_3D/step-16.mpirun=1": "tests_3D/step-16.mpirun=1.output",
    "tests_3D/gaussian-charges.mpirun=1": "tests_3D/gaussian-charges.mpirun=1.output",
    "tests_rhs_rc_variation/rc_variation.mpirun=1": "tests_rhs_rc_variation/rc_variation.mpirun=1.output",
    "cluster/SSOR_run": "Cluster runs output and postprocessing/SSOR_run.o876223",
    "cluster/SSOR_64k_atoms": "Cluster runs output and postprocessing/SSOR_64k_atoms.o876224",
    "cluster/without_opti": "Cluster runs output and postprocessing/without_opti.o875054",
}

ATOM_FILES = ["tests/atom_n1_2.data", "tests/atom_2.data", "atom/atom_n1_8.data", "atom/atom_n3_216.data",
              # expected output of the reference's current regression test, compared line by line with ./step50_mi355x
              "tests/gaussian-charges.mpirun=1.output",
              # |rhs norm with cutoff lists - without| per cutoff (the setup of tests_rhs_rc_variation: 2 atoms, 16^3 cells)
              "Plotting/RHS_Norm_value_comparison_L2.dat", "Plotting/RHS_Norm_value_comparison_Linf.dat",
              # same for the l2 norm of the per-DoF integrated charge density (rc_variation.cc: charge_density_test)
              "Plotting/Total_charge_density_AbsErr_L2.dat"]

NUM = r"([-+0-9.eE]+|nan|inf)"
FIELDS = [
    ("active_cells", r"Number of active cells:\s+(\d+)", int),
    ("rhs_l1", r"L1 rhs norm " + NUM, float),
    ("rhs_l2", r"L2 rhs norm " + NUM, float),
    ("rhs_linf", r"LInfinity rhs norm " + NUM, float),
    ("matrix_l1", r"L1 Matrix norm " + NUM, float),
    ("matrix_linf", r"LInfinity Matrix norm " + NUM, float),
    ("matrix_frobenius", r"Frobenius Matrix norm " + NUM, float),
    ("starting_value", r"Starting value " + NUM, float),
    ("cg_iterations", r"CG converged in (\d+) iterations", int),
    ("convergence_value", r"Convergence value " + NUM, float),
    ("sol_l1", r"L1 solution norm " + NUM, float),
    ("sol_l2", r"L2 solution norm " + NUM, float),
    ("sol_linf", r"LInfinity solution norm " + NUM, float),
    ("refine_threshold", r"Threshold value for refinement:\s+" + NUM, float),
    ("energy_analytical", r"Total analytical electrostatic energy :\s+" + NUM, float),
    ("energy_short", r"Short-ranged energy contribution :\s+" + NUM, float),
    ("energy_fe_long", r"FE solution long-ranged energy contribution :\s+" + NUM, float),
    ("energy_self", r"Self energy contribution :\s+" + NUM, float),
    ("energy_total_split", r"Total electrostatic energy with split in short- and long-ranged :\s+" + NUM, float),
    ("energy_abs_error", r"Absolute Error between both energies :\s+" + NUM, float),
    ("energy_norm_error", r"Error in FE solution in energy norm:\s+" + NUM, float),
]


def parse(text):
    runs = []
    run = None
    cyc = None
    for line in text.splitlines():
        if line.startswith("Problem type is:"):
            run = {"problem": line.split(":", 1)[1].strip(), "cycles": [], "timers": {}}
            runs.append(run)
            cyc = None
            continue
        if run is None:
            continue
        m = re.match(r"Number of atoms:\s+(\d+)", line)
        if m:
            run["n_atoms"] = int(m.group(1))
        m = re.match(r"Running with \w+ on (\d+) MPI", line)
        if m:
            run["mpi_ranks"] = int(m.group(1))
        if "Rhs assembly optimization ENABLED" in line:
            run["rhs_optimization"] = True
        if "Without rhs assembly optimization" in line:
            run["rhs_optimization"] = False
        m = re.match(r"Cycle (\d+):", line)
        if m:
            cyc = {"cycle": int(m.group(1))}
            run["cycles"].append(cyc)
            continue
        m = re.search(r"Number of degrees of freedom:\s+(\d+) \(by level: ([0-9, ]+)\)", line)
        if m and cyc is not None:
            cyc["dofs"] = int(m.group(1))
            cyc["dofs_by_level"] = [int(s) for s in m.group(2).split(",")]
            continue
        m = re.match(r"\|\s*(.+?)\s*\|\s+(\d+)\s+\|\s+" + NUM + r"s\s+\|", line)
        if m:
            run["timers"][m.group(1)] = {"calls": int(m.group(2)), "wall_s": float(m.group(3))}
            continue
        m = re.search(r"Total Elapsed wall time for solution:\s+" + NUM, line)
        if m:
            run["total_wall_s"] = float(m.group(1))
        m = re.search(r"L2 rhs norm " + NUM, line)
        if cyc is None:
            continue
        for key, pat, conv in FIELDS:
            m = re.search(pat, line)
            if m:
                try:
                    cyc[key] = conv(m.group(1))
                except ValueError:
                    pass
                break
    return runs


def main():
    out = {}
    for key, rel in LOGS.items():
        with open(os.path.join(REF, rel), errors="replace") as fh:
            out[key] = {"source": rel, "runs": parse(fh.read())}
    with open(os.path.join(HERE, "reference_logs.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    for rel in ATOM_FILES:
        shutil.copyfile(os.path.join(REF, rel), os.path.join(HERE, os.path.basename(rel)))
        os.chmod(os.path.join(HERE, os.path.basename(rel)), 0o644)
    n = sum(len(r["cycles"]) for v in out.values() for r in v["runs"])
    print("wrote", len(out), "logs,", n, "cycles")


if __name__ == "__main__":
    main()
