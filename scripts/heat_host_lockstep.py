import sys, time, json, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import diffsol_amd as H
nb = int(sys.argv[1]); method = sys.argv[2]
hm = {"bdf": H.METHOD_BDF, "tr_bdf2": H.METHOD_TR_BDF2}[method]
D = np.random.default_rng(12345).uniform(0.5, 2.0, nb)
s = H.Solver("heat1d", D[:, None], nbatch=nb, model_size=512, rtol=1e-6, atol=[1e-6], method=hm)
t0 = time.perf_counter(); y, _ = s.solve_to_points([0.5]); wall = time.perf_counter() - t0
print(json.dumps(dict(leg="host_driven_lockstep", n=512, nbatch=nb, method=method, wall_s=wall, steps=s.stats()["number_of_steps"])))
