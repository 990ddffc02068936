"""Runs an NaCl workload (cells per side) on 2 and 3 ranks sharing one GPU (shared-memory transport) and compares every
cycle with the single-GPU layout: python tools/multi_rank_check.py 5 [ranks:always|auto ...]  (at most 6 GPU processes per box)"""
import os, sys, json, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
os.environ["GMG_COMM_TRANSPORT"] = "peer"; os.environ["GMG_PEER_SLOT_MB"] = "64"
from gpu_util import capi
HERE = os.path.join(ROOT, "tests")
def run(n_ranks, nacl, part):
    uid = capi().Context.unique_id()
    name = uid[len(b"GMGPEER:"):].split(b"\0")[0].decode()
    d = tempfile.mkdtemp()
    outs = [os.path.join(d, f"r{r}.json") for r in range(max(1, n_ranks))]
    ps = [subprocess.Popen([sys.executable, os.path.join(HERE, "two_rank_worker.py"), str(r), str(n_ranks), uid.hex(), os.path.join(HERE, "golden"), outs[r], str(nacl), part]) for r in range(max(1, n_ranks))]
    for p in ps: assert p.wait(timeout=400) == 0
    try: os.unlink("/dev/shm" + name)
    except OSError: pass
    return [json.load(open(o)) for o in outs]
nacl = int(sys.argv[1])
one = run(0, nacl, "auto")[0]
configs = [(int(a.split(":")[0]), a.split(":")[1]) for a in sys.argv[2:]] or [(2, "always"), (3, "always"), (2, "auto")]
for n, part in configs:
    reps = run(n, nacl, part)
    for rk, rep in enumerate(reps):
        for c, (r, g) in enumerate(zip(rep, one)):
            ok = r["cg_iterations"] == g["cg_iterations"] and abs(r["sol_l2"] - g["sol_l2"]) <= 1e-9 * g["sol_l2"] and r["dofs_by_level"] == g["dofs_by_level"]
            print(f"ranks {n} {part} rank {rk} cycle {c}: its {r['cg_iterations']} ({g['cg_iterations']}) coarse {r['coarse_iterations']} ({g['coarse_iterations']}) sol_l2 rel diff {abs(r['sol_l2']-g['sol_l2'])/g['sol_l2']:.2e} solve {r['solve_seconds']*1e3:.1f} ms {'OK' if ok else 'MISMATCH'}")
