#!/usr/bin/env python3
"""The dynamic half of the release gate (round 6; the static half is scripts/check_exec_restore.py, run by the Makefile on every object): the kernels the BENCH runs — and the
instantiation that was wrong in round 5's builds — taken out of the objects the build has just produced, executed instruction by instruction in the gfx950 interpreter of
tests/isaemu over an 8 x 8 (12 x 12) crop of a fixture, against the oracle's film.  No GPU; ~40 s on five cores.  __graft_entry__.build() runs it (HPT_BUILD_SKIP_ISAEMU=1 skips it);
`python scripts/isaemu_gate.py` prints the table and exits non-zero on a film that differs.  Every OTHER shipped kernel: scripts/isaemu_all_kernels.py (20 minutes)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def kernels():
    from tests.isaemu import run as R
    from tests.wavemu import emu as w
    K = R.kernel_symbol
    return [("killeroo (configuration 5)", "basic", K(False, False, 1, 4, 0, True, False, True), w.K_BASIC_STEAL, "cfg1", 8),
            ("killeroo (configuration 7: configuration 5 in its second compilation)", "basic_v", K(False, False, 1, 4, 1, True, False, True), w.K_BASIC_STEAL, "cfg1", 8),
            ("bunny: the headline (configuration 5, out-of-line kd-tree walk)", "measured", K(False, False, 3, 4, 0, True, False, True), w.K_MEASURED_STEAL, "b8", 8),
            ("soup (configuration 6)", "basic", K(False, False, 1, 3, 0, True, False, True), w.K_BASIC_STEAL, "env", 8),
            ("anim (instanced, configuration 6)", "basic_i", K(False, True, 1, 3, 0, True, False, True), w.K_STEAL, "anim", 8),
            ("metal.pbrt (lean set, configuration 6)", "lean", K(False, False, 61, 3, 0, True, False, True), w.K_LEAN_STEAL, "metal", 8),
            ("plain lock step (configuration 3: trees too deep for the stealing rows)", "basic", K(False, False, 1, 4, 0, True, False, False), w.K_LOCKSTEP, "envmap", 12),
            ("the instantiation round 5's builds had wrong (instanced extension set, configuration 6)", "ext_i", K(False, True, 31, 3, 0, True, False, True), w.K_STEAL, "aquad", 8),
            ("the instantiation round 6's greedy build had wrong (instanced extension set, top-level walk, configuration 5)", "ext_i", K(False, True, 31, 4, 0, True, False, True, False, True), w.K_STEAL_TOP, "oinst", 8)]


def run_one(i):
    import numpy as np
    from tests.isaemu import run as R
    from tests.util import hash_rd, load_case
    from oracle import orc   # the checker
    import importlib
    film = importlib.import_module("pbrt-v2_amd.film")
    label, unit, sym, kid, case, n = kernels()[i]
    t0 = time.time()
    s = load_case(case)
    rd = hash_rd(s, seed=3)
    rd.x_start += (rd.x_count - n) // 2; rd.y_start += (rd.y_count - n) // 2; rd.x_count = rd.y_count = n
    fo, so = orc.OracleScene(s).render(s.camera, rd)
    f, info = R.BinaryRender(s, R.code_object_for(unit, sym), sym, kid).render(s.camera, rd)
    a, b = film.xyzw_to_rgb(f), film.xyzw_to_rgb(fo)
    rmse = float(film.rmse(a, b))
    off = int((np.abs(a - b).max(axis=-1) > 1e-2).sum())
    ok = info["samples"] == int(so[0]) and info["bad"] == 0 and bool(np.array_equal(f[..., 3], fo[..., 3])) and rmse < 1e-5 and off == 0
    return (label, unit, rmse, off, info["samples"], int(so[0]), info["bad"], time.time() - t0, ok)


def main(raise_on_failure=False):
    """one child process per kernel (a crash or a hang of one is ITS failure, with a time limit — not a pool that waits for ever), all at once"""
    import json
    import subprocess
    n = len(kernels())
    from tests.isaemu import run as R
    for _, unit, sym, _, _, _ in kernels():      # (unbundle every code object once, here: the children only read them)
        R.code_object_for(unit, sym)
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--one", str(i)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT) for i in range(n)]
    rows, t0 = [], time.time()
    for i, pr in enumerate(procs):
        try:
            out, err = pr.communicate(timeout=max(1.0, 600.0 - (time.time() - t0)))
            rows.append(tuple(json.loads(out.strip().splitlines()[-1])) if pr.returncode == 0 else (kernels()[i][0], kernels()[i][1], float("nan"), -1, 0, 0, 0, time.time() - t0, False, err[-300:]))
        except subprocess.TimeoutExpired:
            pr.kill()
            rows.append((kernels()[i][0], kernels()[i][1], float("nan"), -1, 0, 0, 0, time.time() - t0, False, "timed out"))
    bad = [r for r in rows if not r[8]]
    print("# kernel, unit, RGB rmse against the oracle, pixels off by > 1e-2, camera samples completed / in the job, bad samples, seconds")
    for r in rows:
        label, unit, rmse, off, ns, nj, nb, dt, ok = r[:9]
        print("%-90s %-10s rmse %.2e, %d px off, samples %d/%d, bad %d, %.0f s%s" % (label, unit, rmse, off, ns, nj, nb, dt, "" if ok else "   <-- DIFFERS / FAILED %s" % (r[9] if len(r) > 9 else "")))
    if bad and raise_on_failure:
        raise RuntimeError("shipped kernel binaries render a film that differs from the oracle's in the interpreter (scripts/isaemu_gate.py): %s" % [r[0] for r in bad])
    return 1 if bad else 0


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--one":
        import json
        print(json.dumps(run_one(int(sys.argv[2]))))
        sys.exit(0)
    sys.exit(main())
