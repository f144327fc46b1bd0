import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np
import diffsol_amd as H
for n in (30, 48, 60, 64):
    for nb in (256, 4096):
        c = H.HipContext(nbatch=nb)
        rng = np.random.default_rng(n)
        a = rng.standard_normal((nb, n, n)) + n * np.eye(n)
        A = H.HipMat.from_array(a, c)
        lu = H.HipLU(c, n)
        lu.set_structure(True)
        b = H.HipVec.from_vec(rng.standard_normal((nb, n)), c)
        lu.factor(A); lu.solve_in_place(b); c.synchronize() if hasattr(c, "synchronize") else None
        best_f = best_s = 1e9
        for rep in range(5):
            t0 = time.perf_counter(); lu.factor(A); x = b.clone_as_vec() if rep == 99 else None; _ = lu.n_singular(); t1 = time.perf_counter()
            best_f = min(best_f, t1 - t0)
            t0 = time.perf_counter(); lu.solve_in_place(b); _ = lu.n_singular(); t1 = time.perf_counter()
            best_s = min(best_s, t1 - t0)
        print(f"n={n} nb={nb}: factor {best_f*1e6:.0f} us, solve {best_s*1e6:.0f} us (wall, incl. launch + one readback)", flush=True)
