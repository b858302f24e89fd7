#!/bin/bash
# On the GPU box: HBM traffic of fx_reblock_kernel against its algorithmic bytes (FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes, as the
# MI355X guide prescribes; FETCH_SIZE x 2 on gfx950).  Usage: tools/reblock_traffic.sh [channels window block blocks]   default 16384 4096 2047 16
set -u
export TMPDIR=/tmp
ARGS=${*:-16384 4096 2047 16}
ROOTDIR=$(pwd)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/prof_rbt_$c
  (cd /tmp && rocprofv3 --pmc $c --output-format csv -d $ROOTDIR/gpurun_out/prof_rbt_$c -- python3 $ROOTDIR/tools/device_blocks.py $ARGS > $ROOTDIR/gpurun_out/prof_rbt_$c.log 2>&1)
done
python3 - <<'PY'
import csv, glob
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    n = 0; s = 0.0
    for f in glob.glob("gpurun_out/prof_rbt_%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(f)):
            if "fx_reblock_kernel" in row["Kernel_Name"] and row["Counter_Name"] == c:
                s += float(row["Counter_Value"]); n += 1
    tot[c] = (s, n)
print(open("gpurun_out/prof_rbt_FETCH_SIZE.log").read().strip().splitlines()[-1])
f, nf = tot["FETCH_SIZE"]; w, nw = tot["WRITE_SIZE"]
print("fx_reblock_kernel over %d / %d dispatches (three passes of the stream): FETCH_SIZE %.4g KB (x 2 on gfx950 = %.4g B read), WRITE_SIZE %.4g KB = %.4g B written"
      % (nf, nw, f, 2048.0 * f, w, 1024.0 * w))
PY
