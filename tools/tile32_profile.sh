#!/bin/bash
# tools/tile32_profile.sh <tag> -- plan MFMA32 under rocprofv3: --kernel-trace --stats of the sweep kernels at 4096 problems x 128 knot
# points for a list of shapes (the roofline fraction = SURVEY 8(d) bytes / average duration / 8 TB/s is appended per kernel), then
# FETCH_SIZE and WRITE_SIZE in passes of their own for two of them (MI355X_MICROARCH.md: bytes = 2 * 1024 * FETCH_SIZE + 1024 * WRITE_SIZE)
TAG=${1:-r06c}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_${TAG}_tile32; rm -rf $OUT; mkdir -p $OUT
SUM=gpurun_out/${TAG}_tile32_rocprofv3.txt; : > $SUM
for shp in 13,4 14,7 16,4 20,4 24,8 28,4; do
  echo "# ==== (n, m) = ($shp): rocprofv3 --kernel-trace --stats -- python tools/tile32_check.py --time-only $shp   (MI355X, 4096 problems, N = 128, fp64)" >> $SUM
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/s_$shp -o t -- python tools/tile32_check.py --time-only $shp > $OUT/s_$shp.log 2>&1
  grep "N=" $OUT/s_$shp.log >> $SUM
  python tools/rocpd_summary.py $(find $OUT/s_$shp -name "*.db") | grep "tile32\|avg_us" | cut -c1-170 >> $SUM
  python - "$shp" $(find $OUT/s_$shp -name "*.db") >> $SUM <<'PY'
import sys, sqlite3
n, m = (int(v) for v in sys.argv[1].split(","))
db = sqlite3.connect(sys.argv[2])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select s.kernel_name, count(*), avg(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name").fetchall()
el = {"backward": 3 * n * n + 3 * n * m + m * m + 3 * n + 2 * m, "forward": 2 * n * n + 2 * n * m + 4 * n + 2 * m}
for name, cnt, avg in rows:
    for key in el:
        if "tile32_" + key in name:
            by = el[key] * 8.0 * 128 * 4096
            print("   roofline %-8s: %d elements x 8 B x 128 x 4096 = %.3f GB / %.1f us (avg of %d launches) = %.2f TB/s = %.2f of 8 TB/s" % (key, el[key], by / 1e9, avg / 1e3, cnt, by / avg / 1e3, by / avg / 1e3 / 8.0))
PY
done
for shp in 13,4 28,4; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    echo "# ---- ($shp): --pmc $ctr" >> $SUM
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/p_${ctr}_$shp -o t -- python tools/tile32_check.py --time-only $shp > $OUT/p_${ctr}_$shp.log 2>&1
    python tools/rocpd_summary.py $(find $OUT/p_${ctr}_$shp -name "*.db") | grep "counter\|tile32" | cut -c1-200 >> $SUM
  done
done
find $OUT -name "*.db" -delete
cat $SUM
