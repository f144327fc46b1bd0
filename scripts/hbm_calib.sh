#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (scripts/ubench/hbm_calib.hip): one rocprofv3 --pmc pass per counter.
#   gpurun --timeout 600 -- 'bash scripts/hbm_calib.sh'   ->  gpurun_out/hbm_calib.txt
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
BIN=/tmp/hbm_calib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $BIN scripts/ubench/hbm_calib.hip || exit 1
OUT=gpurun_out/hbm_calib.txt
$BIN 4 > $OUT 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  W=/tmp/calib_$c; rm -rf $W
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $W -o pmc -- $BIN 4 > $W.log 2>&1 < /dev/null
done
python - >> $OUT <<'PY'
import glob, sqlite3, collections
known = {"read8": (1, 0), "write8": (0, 1), "copy8": (1, 1), "rmw8": (1, 1), "read16": (2, 0), "write16": (0, 2), "write4": (0, 0.5)}
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for db in glob.glob(f"/tmp/calib_{c}/**/*.db", recursive=True):
        con = sqlite3.connect(db)
        acc = collections.defaultdict(list)
        for name, cname, val in con.execute("select kernel_name,counter_name,value from counters_collection"):
            acc[(name.split("(")[0].split()[-1], cname)].append(val)
        for (k, cname), v in acc.items():
            res[k][cname] = sum(v) / len(v)
rows = 4 * 1073741824 // (8 * 262144)
unit = rows * 262144 * 8
print(f"\ncounter KB x 1024 / known bytes (unit pass = {unit} bytes)")
print(f"{'kernel':10s} {'known read':>12s} {'FETCH_SIZE':>12s} {'ratio':>7s} {'known write':>12s} {'WRITE_SIZE':>12s} {'ratio':>7s}")
for k, (r, w) in known.items():
    f = res.get(k, {}).get("FETCH_SIZE", float('nan')) * 1024
    ws = res.get(k, {}).get("WRITE_SIZE", float('nan')) * 1024
    print(f"{k:10s} {r*unit:12.4g} {f:12.4g} {f/(r*unit) if r else float('nan'):7.3f} {w*unit:12.4g} {ws:12.4g} {ws/(w*unit) if w else float('nan'):7.3f}")
PY
cat $OUT
