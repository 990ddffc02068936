#!/usr/bin/env python3
"""bench.py -- DoF/s per CG iteration of the GMG-preconditioned CG solve on MI355X.

A "step" is one pass of the hot path (LaplaceProblem::solve, src/step-50.cc:938-1017: outer CG
to 1e-8 |b| with one V-cycle per iteration) on the operators of one adaptive cycle, from the
cycle's initial guess, with every operator and vector already resident in HBM.  value =
(active DoFs x outer CG iterations) / time per step, summed over what all ranks solve.

The headline uses the reference's smoother: SSOR(0.5) x 2 steps (src/step-50.cc:970-973), swept
exactly as one rank of the reference sweeps it (--ssor-blocks 1).  The same operators are then
solved with the reference's smoother as 20 ranks apply it (block SSOR, the cluster runs), with
Jacobi and with Chebyshev: `config.smoothers`.

  python bench.py --gpus 1 --steps 5 --warmup 2 [--workload atoms64000|atoms8000|atoms1000|atoms8|stress201]

--gpus N > 1 started plainly (no RANK in the environment) starts its own N rank processes (torch.distributed.run as a
child process, before this process touches torch or HIP) and relays rank 0's line.  When the box shows fewer GPUs than
ranks the ranks share them (functional runs on a one-GPU box: peer transport only, gloo for the host-side barrier).
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "geometric-multigrid-preconditioners-for-long-range-coulomb-interaction_amd"

# BASELINE.json configs 2-5: NaCl cells per side, box length
WORKLOADS = {
    "atoms8": dict(nacl=1, box=1.0, label="3D gaussian-charges, atom_n1_8 (8 atoms), 45^3 level 0"),
    "atoms1000": dict(nacl=5, box=5.0, label="3D atom_n5_1000 (1000 atoms), 61^3 level 0"),
    "atoms8000": dict(nacl=10, box=10.0, label="3D atom_n10_8000 (8000 atoms), 81^3 level 0"),
    "atoms64000": dict(nacl=20, box=20.0, label="3D atom_n20_64000 (64k atoms), 121^3 level 0"),
    "stress201": dict(nacl=40, box=40.0, label="3D NaCl 512k atoms (same generator), 201^3 level 0"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
# context only (different hardware: 20 CPU ranks, SSOR smoother, 5 cycles incl. build_matrices): BASELINE.md section 1
REFERENCE_SOLVE = {"atoms8": 2.40e6, "atoms1000": 2.06e6, "atoms8000": 1.81e6, "atoms64000": 2.31e6}


def source_hash():
    """sha256 (16 hex digits) over the sources this line was measured on: the kernels, the C-ABI, the host C++, this
    script.  `python bench.py --source-hash` prints it for a checkout: the judge can match a bench line to a commit."""
    import hashlib

    pk = os.path.join(ROOT, PKG)
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "include", "gmg_coulomb.h")]
    for d, exts in ((os.path.join(pk, "csrc"), (".hip", ".hpp")), (os.path.join(pk, "csrc", "host"), (".cc", ".h", ".inc")), (pk, (".py",))):
        files += sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(exts))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def git_head():
    """(commit, note) of the tree being measured.  From git when .git is here (a dirty tree says so); on a GPU box, where
    .git does not travel, from the stamp tools/stamp_commit.sh leaves in .bench_commit -- accepted only when the source
    hash recorded with it equals the hash of the sources actually shipped (a stale stamp is reported, not trusted)."""
    live = source_hash()
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
        dirty = subprocess.call(["git", "-C", ROOT, "diff", "--quiet", "HEAD", "--", ".", ":!PROGRESS.jsonl"], stderr=subprocess.DEVNULL) != 0
        return head + ("+dirty" if dirty else ""), "git"
    except Exception:
        pass
    try:
        with open(os.path.join(ROOT, ".bench_commit")) as fh:
            parts = fh.read().split()
    except OSError:
        return None, "no .git and no .bench_commit here: identify the tree by source_sha16"
    if len(parts) >= 2 and parts[1] == live:
        return parts[0], ".bench_commit (its source hash matches the shipped sources)"
    sys.stderr.write(f"[bench] .bench_commit ({' '.join(parts)}) does not match the shipped sources ({live}): commit not reported\n")
    return None, f"STALE .bench_commit {' '.join(parts)}: the shipped sources hash to {live}"


def spawn_ranks(n):
    """--gpus N > 1 without RANK in the environment: this process has not imported torch or touched HIP; it starts N rank
    processes as children (never replaces itself), relays rank 0's JSON line and returns the launcher's exit code."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, cwd=ROOT)
    line = None
    for raw in proc.stdout:
        txt = raw.decode(errors="replace").rstrip("\n")
        if txt.startswith("{") and '"metric"' in txt:
            line = txt
        elif txt:
            sys.stderr.write(txt + "\n")
    rc = proc.wait()
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    if rc == 0 and line is None:
        sys.stderr.write("[bench] the rank processes ended without a result line\n")
        rc = 1
    return rc


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_traffic.json, written by
    tools/gpu_pmc_traffic.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this script, corrected as
    MI355X_MICROARCH.md prescribes).  A LOOKUP of an earlier run, not a measurement of this one: returns
    (bytes, provenance, same_sources) -- same_sources says whether the pass was taken on exactly the sources being
    measured now (their hash is recorded with the pass) -- or (None, reason, False)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
        w = d.get(workload, {})
        e = w.get("kernels", w).get(kernel)
        if not e:
            return None, f"no PMC pass for {kernel} on {workload} in profiles/pmc_traffic.json", False
        same = w.get("source_sha16") == source_hash()
        return e["traffic_bytes"], (f"lookup: profiles/pmc_traffic.json (commit {w.get('commit', 'not recorded')}, sources {w.get('source_sha16', 'not recorded')}"
                                    f"{' = these sources' if same else ' (NOT the sources measured now)'}; {e.get('launches', '?')} launches)"), same
    except (OSError, ValueError, KeyError) as exc:
        return None, f"profiles/pmc_traffic.json unreadable: {exc}", False


def spmv_bytes(n, nnz):
    return 12 * nnz + 4 * (n + 1) + 16 * n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="atoms64000", choices=sorted(WORKLOADS))
    ap.add_argument("--smoother", default="SSOR", choices=["Jacobi", "SSOR", "Chebyshev"],
                    help="smoother of the headline solve (default: the reference's, src/step-50.cc:970)")
    ap.add_argument("--ssor-blocks", type=int, default=None,
                    help="SSOR as the reference applies it on this many MPI ranks (1 = exact sequential sweep); default: one block "
                         "per rank, i.e. 1 on one GPU and N on N GPUs (each rank sweeps its block, as the reference's ranks do)")
    ap.add_argument("--no-smoother-table", action="store_true", help="skip the solves with the other smoothers (config.smoothers)")
    ap.add_argument("--cycles", type=int, default=5, help="adaptive cycles to run (the reference runs 5); the last one is timed")
    ap.add_argument("--partition-level0", default="auto", choices=["auto", "always", "never"],
                    help="N > 1: row-partition level 0 (distributed coarse CG) or keep it replicated; auto decides by size and transport (DESIGN.md 6)")
    ap.add_argument("--refinement-estimator", default="Kelly", choices=["Kelly", "Kelly + residual"],
                    help="marking rule: Kelly = the cluster runs (January 2018), Kelly + residual = the reference's HEAD")
    ap.add_argument("--transport", default="auto", choices=["auto", "rccl", "peer"],
                    help="N > 1: auto = the peer-to-peer transport on one node (hipIpc mailboxes, kernels store into the peer's HBM "
                         "and poll flags: no collective launches), RCCL only if the peers' memory cannot be mapped; DESIGN.md 6")
    ap.add_argument("--no-single-gpu-reference", action="store_true",
                    help="N > 1: skip rank 0's single-GPU solve of the same operators with the same smoother (SSOR in N blocks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-every", type=int, default=8)
    ap.add_argument("--source-hash", action="store_true", help="print the hash of the measured sources and exit")
    args = ap.parse_args()
    if args.source_hash:
        print(source_hash())
        return 0
    if args.gpus > 1 and "RANK" not in os.environ:
        return spawn_ranks(args.gpus)  # (nothing of torch / HIP has been touched in this process)

    # stdout carries exactly one line (the JSON of rank 0): whatever libraries print while the job runs
    # (RCCL prints a version banner at communicator creation) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    n_dev = torch.cuda.device_count()
    shared = world > n_dev  # fewer GPUs than ranks (a one-GPU box): the ranks share them -- functional runs, not speed
    device = local_rank % n_dev
    torch.cuda.set_device(device)
    launched = "RANK" in os.environ  # under torch.distributed.run: use the one-process-per-GPU layout even for N = 1
    if launched:
        if shared:  # RCCL refuses two ranks on one device: the host-side barrier / max over ranks go through gloo
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.ssor_blocks is None:
        args.ssor_blocks = world
    red_dev = "cpu" if shared else "cuda"

    pkg = importlib.import_module(PKG)
    S = pkg.step50
    w = WORKLOADS[args.workload]

    def barrier():
        if launched:
            dist.barrier()
        torch.cuda.synchronize()

    def all_min(flag):
        if not launched:
            return flag
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    os.environ["STEP50_DEVICE"] = str(device)
    # replicated host setup: a GPU box gives each GPU a share of 16 host cores
    S.set_threads(max(1, min(16, (os.cpu_count() or 16) // max(1, world))))

    def make_problem(ssor_blocks):
        q = S.Problem(S.prm_text(left=0, right=w["box"], mesh_size=0.25, vacuum=10, problem="GaussianCharges", dim=3,
                                 bc="Inhomogeneous", cycles=args.cycles, r_c=0.5, cutoff=3.5, rhs_optimization=True,
                                 quad_rhs=1, global_refinement=0, smoother=args.smoother, ssor_blocks=ssor_blocks,
                                 partition_level0=args.partition_level0,
                                 # the marking rule of the revision that produced the cluster logs BASELINE.json's configs are
                                 # quoted on: the timed cycle then IS the reference's (64k atoms, cycle 4: 1 926 877 DoFs on
                                 # levels 1 771 561 / 170 516 / 14 336, SSOR_64k_atoms.o876224:50-52); tests/test_cluster_cycles.py
                                 refinement_estimator=args.refinement_estimator))
        q.set_nacl_atoms(w["nacl"])
        return q

    # ---- transport (N > 1): peer-to-peer stores between the GPUs of one node; RCCL if the peers' memory cannot be mapped
    transport, transport_note = "none", ""
    p = make_problem(args.ssor_blocks)
    if launched:
        def join(kind):
            if kind == "peer":
                os.environ["GMG_COMM_TRANSPORT"] = "peer"
                os.environ.setdefault("GMG_PEER_SLOT_MB", "64")
            else:
                os.environ.pop("GMG_COMM_TRANSPORT", None)
            box = [pkg.capi.Context.unique_id() if rank == 0 else None]  # rank 0 creates the id, everybody joins (gmg_comm_init)
            dist.broadcast_object_list(box, src=0)
            try:
                p.set_communicator(rank, world, box[0])
                return True, ""
            except RuntimeError as exc:
                return False, str(exc)

        want = args.transport
        if want == "auto":
            peer_ok = True
            if not shared and world > 1:  # every pair of GPUs must be able to map each other's memory
                peer_ok = all(torch.cuda.can_device_access_peer(device, d) for d in range(n_dev) if d != device)
            want = "peer" if all_min(peer_ok) else "rccl"
            if want == "rccl":
                transport_note = "auto: a pair of GPUs without peer access -> RCCL"
        if shared and want == "rccl" and world > 1:
            raise SystemExit("--transport rccl needs one GPU per rank (RCCL refuses ranks that share a device)")
        ok, why = join(want)
        if not all_min(ok):
            if args.transport != "auto" or want == "rccl" or shared:
                raise SystemExit(f"communicator ({want}) could not be set up: {why}")
            # the peer mailboxes could not be mapped on some rank: every rank starts over on RCCL
            transport_note = f"auto: peer transport failed at start-up ({why or 'on another rank'}) -> RCCL"
            p.close()
            p = make_problem(args.ssor_blocks)
            want = "rccl"
            ok, why = join(want)
            if not all_min(ok):
                raise SystemExit(f"communicator (rccl) could not be set up: {why}")
        transport = want
    def run_cycles(prob):
        out = []
        r = None
        for cycle in range(args.cycles):
            r = prob.run_cycle(cycle, on_device=True)  # assembles, uploads, solves once (untimed), marks + refines
            out.append({"cycle": cycle, "dofs": r["dofs"], "dofs_by_level": r["dofs_by_level"],
                        "outer_cg_iterations": r["cg_iterations"], "coarse_cg_iterations": r["coarse_iterations"],
                        "solve_ms": round(r["solve_seconds"] * 1e3, 3), "build_matrices_ms": round(r["build_matrices_ms"], 3)})
        return r, out

    t_setup = time.time()
    rep, cycles, why = None, [], ""
    try:
        rep, cycles = run_cycles(p)
    except RuntimeError as exc:  # (a rank whose peers' stores never arrive gives up after a bounded wait: GMG_ERR_COMM on every rank)
        why = str(exc)
    if not all_min(rep is not None):
        # --transport auto on GPUs the peer transport was never run across: if its first solves fail on any rank, every
        # rank starts over on RCCL (the line says so); anything else is an error
        if not (launched and transport == "peer" and args.transport == "auto" and not shared):
            raise SystemExit(f"setup cycles failed: {why or 'on another rank'}")
        transport_note = f"auto: the peer transport failed in the first solves ({why or 'on another rank'}) -> RCCL"
        print("[bench] " + transport_note, file=sys.stderr)
        try:
            p.close()
        except Exception:
            pass
        p = make_problem(args.ssor_blocks)
        ok, why = join("rccl")
        if not all_min(ok):
            raise SystemExit(f"communicator (rccl) could not be set up: {why}")
        transport = "rccl"
        rep, cycles = run_cycles(p)
    t_setup = time.time() - t_setup
    ctx = pkg.capi.Context.view(p.gmg_context())  # non-owning view of the problem's gmg_context (stats)
    comm_info = ctx.comm_info()

    def timed_solves(prob, cx, steps, warmup, profile_every, collective=True):
        """`steps` passes of the hot path between barriers; returns (seconds, last report, stats of the timed region)."""
        for _ in range(warmup):
            prob.solve_again()
        cx.set_profiling(profile_every)
        cx.stats_reset()
        if collective:
            barrier()
        else:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = None
        for _ in range(steps):
            r = prob.solve_again()
        if collective:
            barrier()
        else:
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = cx.stats()
        cx.set_profiling(0)
        if launched and collective:
            tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, r, st

    dt, rep_t, st = timed_solves(p, ctx, args.steps, args.warmup, args.profile_every)
    hbm_read, hbm_copy = ctx.calibrate_hbm(1 << 30, 10) if rank == 0 else (0.0, 0.0)

    ms_per_step = dt / args.steps * 1e3
    dofs, its = rep["dofs"], rep_t["cg_iterations"]
    value = dofs * its / (dt / args.steps)  # the ranks solve ONE problem together (strong scaling)

    def smoother_entry(name, blocks, dt_, steps_, r_, st_, ranks=1):
        e = {"smoother": name, "ms_per_solve": round(dt_ / steps_ * 1e3, 3), "outer_cg_iterations": r_["cg_iterations"],
             "coarse_cg_iterations": int(r_["coarse_iterations"]), "DoF_it_per_s": dofs * r_["cg_iterations"] / (dt_ / steps_)}
        if name == "SSOR":
            e["ssor_blocks"] = blocks
            e["sweep"] = "exact sequential order (the reference on 1 rank)" if blocks == 1 else f"block Jacobi of rank-local sweeps (the reference on {blocks} ranks)"
            if st_.sgs_samples:
                launches = st_.sgs_launches or st_.sgs_samples
                e["sweep_launches_per_solve"] = round(launches / steps_, 1)
                e["sweep_launches_timed"] = int(st_.sgs_samples)
                e["sweep_ms_per_solve"] = round(st_.sgs_ms_total / st_.sgs_samples * launches / steps_, 3)
                # the blocks of a level are swept side by side (one workgroup each, on their ranks): the chain is one block's steps
                e["sweep_ns_per_dependent_step"] = round(st_.sgs_ms_total * 1e6 / max(1, st_.sgs_substeps) * max(1, blocks), 1)
        return e

    head_key = f"SSOR_B{args.ssor_blocks}" if args.smoother == "SSOR" else args.smoother
    smoothers = {head_key: smoother_entry(args.smoother, args.ssor_blocks, dt, args.steps, rep_t, st, world)}
    if not args.no_smoother_table and world == 1:
        # SSOR in 2 / 4 / 8 blocks: what an N-GPU run (one block per rank) must be divided by for a like-for-like speed-up
        for name, blocks in (("SSOR", 1), ("SSOR", 20), ("SSOR", 2), ("SSOR", 4), ("SSOR", 8), ("Jacobi", 1), ("Chebyshev", 1)):
            key = f"SSOR_B{blocks}" if name == "SSOR" else name
            if key in smoothers:
                continue
            p.set_smoother(name, blocks)
            dt2, r2, st2 = timed_solves(p, ctx, 3, 1, 1)
            smoothers[key] = smoother_entry(name, blocks, dt2, 3, r2, st2)
        p.set_smoother(args.smoother, args.ssor_blocks)

    n0, nnz0 = st.spmv0_rows, st.spmv0_nnz
    roof = None
    if st.spmv0_samples > 0:
        # HIP start / stop events attached to the dispatch (hipExtLaunchKernelGGL): the kernel's own duration, as
        # rocprofv3 --kernel-trace reports it; launches that returned at once after convergence are kept apart
        t_k = st.spmv0_ms_total / st.spmv0_samples * 1e-3
        t_noop = st.spmv0_noop_ms_total / st.spmv0_noop_samples * 1e-3 if st.spmv0_noop_samples else None
        # fused variant: SpMV + direction update in one kernel = SpMV bytes + 16 N (reads g, writes d);
        # unfused variant (large level 0): plain SpMV + partial d.h
        fused = st.coarse_variant == 1
        alg = spmv_bytes(n0, nnz0) + (16 * n0 if fused else 0)
        lay = int(st.spmv0_layout)
        val8, col16 = bool(lay >= 1 and (lay - 1) & 2), bool(lay >= 1 and (lay - 1) & 4)
        tmpl = f"0, {1 if fused else 2}" + (f", {'true' if val8 else 'false'}, {'true' if col16 else 'false'}" if lay >= 1 else "")
        kname = ("spmv_sell_kernel" if lay >= 1 else "spmv_tile_kernel") + f"<{tmpl}>"
        if lay >= 1 and (lay - 1) & 8:  # lattice operator: pattern-run kernel (pair loads + lane shift), with row classes or value codes
            kname = f"spmv_sellp_kernel<0, {1 if fused else 2}, {'true' if (lay - 1) & 16 else 'false'}>"
        if lay >= 1 and (lay - 1) & 32:  # lattice operator walked plane by plane (sliding window of x lines in registers)
            # (template arguments: 2 = with the d.h partials -- the lattice kernel has no fused variant --, false = every marching
            # wave takes one column: lattices up to ~360^3; beyond, the multi-pass variant <2, true> runs)
            kname = "spmv_lattice_kernel<2, false>" if n0 <= 360 ** 3 else "spmv_lattice_kernel<2, true>"
        # bytes the kernel actually moves: the operator in its device layout (the library reports the exact size of the
        # streams) + x read (8 N) + y written (8 N); the fused variant also reads g and writes d (16 N)
        moved = int(st.spmv0_matrix_bytes) + 16 * n0 + (16 * n0 if fused else 0)
        traffic, traffic_src, traffic_same = pmc_traffic(args.workload, kname)
        # the fraction of peak is stated on bytes MOVED: the PMC figure when a pass on exactly these sources exists, else
        # the layout bytes (which over-count what the caches absorb: the x lines of neighbouring waves)
        basis_bytes, basis = (traffic, "pmc") if (traffic and traffic_same) else (moved, "layout")
        ach = basis_bytes / t_k / 1e9
        roof = {"bound": "hbm", "kernel": kname + (" (level-0 SpMV + CG direction update)" if fused else " (level-0 SpMV + d.h partials)"),
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "frac_basis": basis + (" (PMC traffic of this kernel on these sources / live launch duration)" if basis == "pmc" else
                                       " (bytes of the device layout + vectors / live launch duration; no PMC pass on exactly these sources)"),
                "traffic": traffic, "traffic_source": traffic_src,
                "bytes_per_launch": int(moved), "layout_GBps": round(moved / t_k / 1e9, 1),
                "note": "achieved = bytes the kernel moves per launch / the launch's own duration (HIP events on the dispatch, sampled in "
                        "the timed region); `traffic` is the PMC figure for the same kernel from a separate profiling run.  The caller hands "
                        "over CSR: against SURVEY 8(d)'s algorithmic CSR bytes the same launch is `effective_vs_csr`.  At 121^3 one "
                        "coarse iteration's working set fits the 256 MiB Infinity Cache, so `achieved` is a cache-assisted rate; "
                        "the stress201 workload (beyond the cache) is the HBM-resident figure (DESIGN.md section 5)",
                "effective_vs_csr": {"algorithmic_bytes_per_launch": alg, "GB_per_s": round(alg / t_k / 1e9, 1),
                                     "x_peak": round(alg / t_k / 1e9 / HBM_PEAK_GBS, 3)},
                "pattern_slices": [int(st.spmv0_pattern_slices), int(st.spmv0_slices)],
                "measured_stream_read_GBps": round(hbm_read, 1),
                "measured_stream_copy_GBps": round(hbm_copy, 1), "avg_launch_us": round(t_k * 1e6, 2),
                "launches_sampled": int(st.spmv0_samples), "noop_launches_sampled": int(st.spmv0_noop_samples),
                "avg_noop_launch_us": None if t_noop is None else round(t_noop * 1e6, 2),
                # iterations enqueued ahead of the host's convergence check return at once (~1 us per kernel): a
                # rocprofv3 --stats average over ALL launches of the kernel is lower than avg_launch_us by this share
                "noop_launch_share": round(1.0 - st.coarse_iterations / max(1, st.coarse_enqueued), 4)}
        if st.cgupd_samples > 0:
            t_u = st.cgupd_ms_total / st.cgupd_samples * 1e-3
            # fused variant: x += alpha d and g += alpha h in one kernel (48 N); three-kernel variant: g only (24 N),
            # x is brought up to date every 8 iterations by cg_xflush_kernel
            upd_name, upd_bytes = ("cg_update_kernel", 48 * n0) if fused else ("cg_update_g_kernel", 24 * n0)
            roof[upd_name] = {"avg_launch_us": round(t_u * 1e6, 2), "bytes_per_launch": upd_bytes,
                              "achieved": round(upd_bytes / t_u / 1e9, 1)}
        # one whole coarse-CG iteration (SpMV + direction + g update + 1/8 x flush), timed on a level-0 solve of its own
        # (collective on a partitioned level 0: every rank takes part and reports its own figure)
        ci = coarse_iteration_rate(ctx, int(n0), moved, fused)
        if launched and world > 1:
            every = [None] * world
            dist.all_gather_object(every, None if ci is None else ci["us_per_iteration"])
            if ci is not None:
                ci["us_per_iteration_by_rank"] = every
        roof["coarse_iteration"] = ci
        spmv_ms = t_k * 1e3 * st.coarse_iterations / args.steps
        roof["kernel_time_per_step"] = {"level0_spmv_ms": round(spmv_ms, 3)}
        dominant = {"kernel": roof["kernel"], "ms_per_step": round(spmv_ms, 3), "share_of_step": round(spmv_ms / ms_per_step, 4),
                    "bound": "hbm", "frac": roof["frac"]}
        if st.sgs_samples:
            e = smoothers[head_key]
            sweep_ms = e["sweep_ms_per_solve"]
            # SURVEY 8(d)-style algorithmic bytes of a sweep pair: two passes over the level matrix (forward + backward)
            alg_sweeps = 0
            for l in range(1, len(rep["dofs_by_level"])):
                nr, nz = p.matrix_shape("level", l)
                alg_sweeps += 2 * spmv_bytes(nr, nz)
            launches_per_level = e["sweep_launches_per_solve"] / max(1, len(rep["dofs_by_level"]) - 1)
            alg_gbps = alg_sweeps * launches_per_level / max(1e-9, sweep_ms * 1e-3) / 1e9
            roof["kernel_time_per_step"].update({
                "ssor_sweep_ms": round(sweep_ms, 3),
                "ssor_sweep_note": "the SSOR sweep is a chain of dependent steps (rows of one dependency stage each, y in LDS, one workgroup per "
                                   "block): latency bound, no bandwidth roofline applies; its figure of merit is ns per dependent step",
                "ssor_sweep_ns_per_dependent_step": e.get("sweep_ns_per_dependent_step"),
                "ssor_sweep_stream_GBps": round(st.sgs_stream_bytes / max(1e-9, st.sgs_ms_total * 1e-3) / 1e9, 2)})
            if sweep_ms > spmv_ms:
                dominant = {"kernel": "sgs_phase_kernel (SSOR sweep of the levels >= 1, exact dependency order)", "ms_per_step": round(sweep_ms, 3),
                            "share_of_step": round(sweep_ms / ms_per_step, 4), "bound": "latency (chain of dependent steps; not a roofline kernel)",
                            "ns_per_dependent_step": e.get("sweep_ns_per_dependent_step"),
                            "algorithmic_GBps": round(alg_gbps, 1), "frac": round(alg_gbps / HBM_PEAK_GBS, 5),
                            "frac_note": "2 x SpMV(A_l) CSR bytes per sweep pair (SURVEY 8(d) style) / sweep time / 8 TB/s: reported for "
                                         "completeness; a sequential Gauss-Seidel chain cannot be bandwidth bound"}
        roof["dominant_kernel"] = dominant

    # ---- N > 1: the same operators with the same smoother (SSOR in N blocks) on ONE GPU, measured now on rank 0's GPU, so
    # that value(N) / value(1) compares the same arithmetic
    single = None
    if launched and world > 1 and not args.no_single_gpu_reference:
        if rank == 0:
            q = make_problem(args.ssor_blocks)
            for cycle in range(args.cycles):
                rq = q.run_cycle(cycle, on_device=True)
            cq = pkg.capi.Context.view(q.gmg_context())
            dtq, rq_t, stq = timed_solves(q, cq, max(2, args.steps), 1, args.profile_every, collective=False)
            nq = max(2, args.steps)
            single = smoother_entry(args.smoother, args.ssor_blocks, dtq, nq, rq_t, stq)
            single["speedup_of_this_run"] = round((dtq / nq) / (dt / args.steps), 3)
            single["note"] = ("one process, one GPU, no communicator, same adaptive cycles, same smoother" +
                              (" -- measured while the other ranks' processes idle on the SAME GPU" if shared else ""))
            q.close()
        barrier()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(p, rep, args.smoother, args.ssor_blocks)

    if rank == 0:
        commit, commit_src = git_head()
        out = {
            "metric": "DoF/s per CG-iter (GMG-precond Poisson, 3D)", "value": value, "unit": "DoF*it/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": w["label"] + f", adaptive cycle {args.cycles - 1} of {args.cycles}", "cycle": args.cycles - 1,
                       "refinement_estimator": args.refinement_estimator + (" (reproduces the reference's cluster logs cycle by cycle: tests/test_cluster_cycles.py)" if args.refinement_estimator == "Kelly" else " (the reference's HEAD)"),
                       "smoother": args.smoother + (f" (0.5, 2 steps), {args.ssor_blocks} block(s)" if args.smoother == "SSOR" else ""),
                       "smoothers": smoothers, "single_gpu_same_smoother": single, "cycles": cycles,
                       "all_cycles_DoF_it_per_s": sum(c["dofs"] * c["outer_cg_iterations"] for c in cycles)
                       / max(1e-12, sum(c["solve_ms"] for c in cycles) * 1e-3),
                       "reference_cpu_20_ranks_SSOR_DoF_it_per_s": REFERENCE_SOLVE.get(args.workload),
                       "dofs": dofs, "dofs_by_level": rep["dofs_by_level"], "outer_cg_iterations": its,
                       "coarse_cg_iterations_per_step": int(rep_t["coarse_iterations"]),
                       "level0_rows": int(n0), "level0_nnz": int(nnz0), "setup_seconds": round(t_setup, 2),
                       # MGTransferPrebuilt::build_matrices (src/step-50.cc:957-958: inside the reference's Solve timer) of the timed
                       # cycle, built on the device from the levels' vertex tables (gmg_build_transfer); not part of ms_per_step:
                       # the operators of a cycle are built once, the step is the solve on them
                       "build_matrices_ms": round(rep["build_matrices_ms"], 3),
                       "level0_matrix": "formed on the device (gmg_set_level_matrix_lattice)" if (int(st.spmv0_layout) - 1) & 64 else "CSR from the host",
                       "transport": transport, "transport_note": transport_note, "communicator": comm_info,
                       "rccl_ranks": world if transport == "rccl" else 0, "peer_ranks": world if transport == "peer" else 0,
                       "gpus_visible": n_dev, "ranks_share_gpus": bool(shared),
                       "commit": commit, "commit_source": commit_src, "source_sha16": source_hash(),
                       "parallelism": f"{world} rank(s), one per GPU" + (" (SHARING the box's %d GPU(s): a functional run, not a speed measurement)" % n_dev if shared else "") + ("" if world == 1 else (
                           "; system matrix + outer CG rows partitioned, SSOR blocks swept by their ranks, level 0 " +
                           (f"partitioned (halo entries + 2 sums per coarse iteration over the {transport} transport)" if comm_info["level0_partitioned"]
                            else "replicated (too small to pay three collectives per coarse iteration), levels >= 1 replicated")))},
            "roofline": roof, "cpu_baseline": cpu,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    p.close()
    if launched:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def coarse_iteration_rate(ctx, n0, spmv_moved, fused):
    """Bytes moved by one whole coarse-CG iteration / its duration, from a level-0 solve of its own (MGCoarseGrid
    operator() on a fixed right-hand side, timed on the host around the device-resident iteration loop)."""
    import numpy as np

    b = np.zeros(n0)
    b[n0 // 2] = 1.0
    vb, vx = ctx.vector(n0, b), ctx.vector(n0)
    ctx.coarse_solve(vx, vb)
    t0 = time.perf_counter()
    reps, its = 3, 0
    for _ in range(reps):
        it, _, _ = ctx.coarse_solve(vx, vb)
        its += it
    dt = time.perf_counter() - t0
    vb.free()
    vx.free()
    if its == 0:
        return None
    # fused: SpMV(+direction) + update (48 N); three-kernel: SpMV + direction (24 N) + g update (24 N) + x flush (80 N / 8)
    moved = spmv_moved + (48 * n0 if fused else 24 * n0 + 24 * n0 + 10 * n0)
    us = dt / its * 1e6
    return {"us_per_iteration": round(us, 2), "bytes_moved": int(moved), "achieved": round(moved / us / 1e3, 1), "unit": "GB/s",
            "frac": round(moved / us / 1e3 / HBM_PEAK_GBS, 4), "iterations_timed": int(its),
            "note": "includes kernel boundaries and the host's convergence polls"}


def cpu_baseline(p, rep, smoother, ssor_blocks):
    """The oracle (CPU restatement, kind 'port') timed on this box's host cores on the same
    operators: one full solve of the timed cycle when it fits ~30 s, else a bounded number of
    level-0 CG iterations scaled up."""
    from oracle import gmg_oracle as go

    threads = max(1, min(16, go.max_threads(), os.cpu_count() or 1))  # the box's CPU share for one GPU
    go.set_threads(threads)
    h = p.hierarchy()
    kind = {"Jacobi": go.JACOBI, "SSOR": go.SSOR, "Chebyshev": go.CHEBYSHEV}[smoother]
    n0 = h.level_matrices[0].n_rows
    # ~1.5 ns per nonzero and thread-second for the OpenMP CSR SpMV + BLAS-1 of one level-0 iteration
    per_it = 1.5e-9 * h.level_matrices[0].nnz * 16 / max(1, threads) / 16 * 1.6
    budget_its = max(8, int(30.0 / max(per_it, 1e-9)))
    full = rep["coarse_iterations"] <= budget_its
    t0 = time.perf_counter()
    if full:
        mg = go.OracleMG(h, smoother=kind, ssor_blocks=ssor_blocks)
        r = mg.solve(h.system_rhs, x0=p.vector("initial_guess"))
        dt = time.perf_counter() - t0
        value = rep["dofs"] * r["iterations"] / dt
        sample = f"one full solve of the timed cycle ({smoother} smoother): {r['iterations']} outer / {r['coarse_iterations']} coarse CG iterations, {dt:.2f} s"
    else:
        mg = go.OracleMG(h, smoother=kind, coarse_maxit=budget_its)
        mg.coarse_solve(h.system_rhs if len(h.level_matrices) == 1 else h.system_rhs[:n0] * 0 + 1.0)
        dt = time.perf_counter() - t0
        per_it = dt / budget_its
        est = per_it * rep["coarse_iterations"]
        value = rep["dofs"] * rep["cg_iterations"] / est
        sample = f"{budget_its} level-0 CG iterations ({dt:.2f} s), scaled to the {rep['coarse_iterations']} of one step"
    go.set_threads(1)
    return {"value": value, "unit": "DoF*it/s", "cores": threads, "kind": "port", "sample": sample}


if __name__ == "__main__":
    sys.exit(main())
