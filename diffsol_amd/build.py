"""Build the native libraries of diffsol_amd in-tree (diffsol_amd/lib/*.so) for gfx950.

    python -m diffsol_amd.build            # device library (hipcc) + host library (g++)

hipcc cross-compiles gfx950 without a GPU, so this runs in the CPU-only build container; the resulting .so files travel
with the repository snapshot to the GPU box.
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose=False, jobs=8):
    out = None if verbose else subprocess.DEVNULL
    subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), f"-j{jobs}"], check=True, stdout=out)
    subprocess.run(["make", "-C", os.path.join(_HERE, "host")], check=True, stdout=out)
    return [os.path.join(_HERE, "lib", "libdiffsol_hip.so"), os.path.join(_HERE, "lib", "libdiffsol_hip_host.so")]


MANIFEST_DIR = os.path.join(_HERE, "jit_manifest")


def replay_manifests(jobs=None, verbose=False, only=None, budget_s=420.0, preload_torch=False):
    """Compile every request of the committed manifests (diffsol_amd/jit_manifest/*.rec — written on a GPU box under DSH_JIT_RECORD by bench.py and the `-m gpu`
    tests, scripts/record_jit_manifest.sh) into the in-tree cache diffsol_amd/_jit_cache/ with hiprtc: no GPU needed, one process per core, requests dealt round-robin.
    A fresh box then loads every code object it asks for instead of compiling at first use.  Returns (requests, compiled).
    BEST EFFORT: this only pre-warms a cache.  A worker that fails (no torch to preload, another hiprtc, a manifest whose format no longer matches) is
    reported on stderr and its requests are left for first-use compilation; nothing here fails the build."""
    import glob
    import gzip
    import shutil
    import tempfile
    tmpdir = tempfile.mkdtemp(prefix="dsh_manifest_")
    files = []
    for gz in sorted(glob.glob(os.path.join(MANIFEST_DIR, "*.rec.gz"))):  # committed gzip-compressed (the translation units are text: 21 MB -> 1 MB)
        if only and os.path.basename(gz).split(".")[0] not in only:
            continue
        out = os.path.join(tmpdir, os.path.basename(gz)[:-3])
        with gzip.open(gz, "rb") as fi, open(out, "wb") as fo:
            shutil.copyfileobj(fi, fo)
        files.append(out)
    if not files:
        return 0, 0
    jobs = jobs or max(1, min(len(os.sched_getaffinity(0)), 16))
    # worker k compiles requests k, k + jobs, ... of every manifest, one dsh_jit_replay call per request, and starts no new one after the deadline (what is left is
    # compiled at first use, as before)
    # preload_torch: PyTorch-ROCm bundles its own HIP runtime / hiprtc (another ROCm release than /opt/rocm); a process that imports torch first resolves the library's
    # HIP symbols there, so its cache key (toolchain versions are part of it) and its code generator differ from a plain process's.  bench.py imports torch: its
    # manifest is replayed in both environments.
    code = (("try:\n    import torch\nexcept Exception as e:\n    import sys; print('jit replay: torch not importable, replaying without it:', e, file=sys.stderr)\n" if preload_torch else "") +
            "import ctypes as C, sys, time\n"
            "from diffsol_amd import _ffi\n"
            "dev = _ffi.load_device_lib()\n"
            "k, jobs, deadline = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])\n"
            "tot = [0, 0, 0, 0]\n"
            "for path in sys.argv[4:]:\n"
            "    r, c = C.c_int64(0), C.c_int64(0)\n"
            "    if dev.dsh_jit_replay(path.encode(), (1 << 30) - 1, 1 << 30, C.byref(r), C.byref(c)) != 0:\n"
            "        print('jit replay: unreadable manifest', path, dev.dsh_last_error(), file=sys.stderr)\n"
            "        tot[3] += 1\n"
            "        continue\n"
            "    n = r.value\n"
            "    tot[0] += n\n"
            "    for i in range(k, n, jobs):\n"
            "        if time.time() > deadline:\n"
            "            tot[2] += 1\n"
            "            continue\n"
            "        if dev.dsh_jit_replay(path.encode(), i, n, C.byref(r), C.byref(c)) != 0:\n"
            "            print('jit replay: request', i, 'of', path, 'failed:', dev.dsh_last_error(), file=sys.stderr)\n"
            "            tot[3] += 1\n"
            "            continue\n"
            "        tot[1] += c.value\n"
            "print(tot[0], tot[1], tot[2], tot[3])\n")
    env = dict(os.environ, PYTHONPATH=os.path.dirname(_HERE) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("DSH_JIT_RECORD", None)
    import time
    deadline = time.time() + budget_s
    procs = [subprocess.Popen([sys.executable, "-c", code, str(k), str(jobs), repr(deadline)] + files, env=env, stdout=subprocess.PIPE, text=True) for k in range(jobs)]
    requests = compiled = skipped = failed = 0
    for k, pr in enumerate(procs):
        out, _ = pr.communicate()
        try:
            if pr.returncode != 0:
                raise ValueError(f"exit code {pr.returncode}")
            r, c, sk, fl = out.split()[-4:]
            requests = max(requests, int(r))
            compiled += int(c)
            skipped += int(sk)
            failed += int(fl)
        except ValueError as e:  # a cache pre-warm: log and go on, the requests of this worker compile at first use
            failed += 1
            print(f"jit replay: worker {k} failed ({e}); its requests are left for first-use compilation", file=sys.stderr)
    shutil.rmtree(tmpdir, ignore_errors=True)
    if verbose or failed:
        print(f"jit manifests: {requests} requests, {compiled} compiled now, {skipped} left for first use (time budget {budget_s:g} s), {failed} failed", file=sys.stderr if failed else sys.stdout)
    return requests, compiled


if __name__ == "__main__":
    for p in build(verbose="-v" in sys.argv):
        print(p)
