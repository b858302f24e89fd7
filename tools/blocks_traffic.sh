#!/bin/bash
# On the GPU box: HBM traffic of a stream of device blocks through fx_push_samples, every kernel of the library counted (FETCH_SIZE and WRITE_SIZE in
# separate rocprofv3 --pmc passes, as the MI355X guide prescribes; FETCH_SIZE x 2 on gfx950), per call, beside the sample bytes a call delivers:
#   (a) 480-sample blocks, the one-frame kernels reading the blocks themselves (round 6)
#   (b) the same with every call re-blocked first (test hook 16: the round-5 path)
#   (c) 512-sample blocks = whole hops, analysed in place (what the analysis alone moves)
# Usage: tools/blocks_traffic.sh [channels window [block blocks]]        default 8192 1024 480 64; (c) uses the whole hops nearest the block
#        (e.g. 8192 1024 4097 16: calls of eight hops -- the batch kernel's block-fed form against re-blocking first, against 4096-sample blocks in place)
set -u
export TMPDIR=/tmp
C=${1:-8192}; N=${2:-1024}; B=${3:-480}; NB=${4:-64}
H=$((N / 2)); if [ $B -lt $H ]; then W=$H; else W=$((B / H * H)); fi
ROOTDIR=$(pwd)
mkdir -p gpurun_out/blocks_traffic
for cfg in "$B $NB" "$B $NB reblock" "$W $NB"; do
  tag=$(echo $cfg | tr ' ' '_')
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/blocks_traffic/prof_${tag}_$c
    (cd /tmp && rocprofv3 --pmc $c --output-format csv -d $ROOTDIR/gpurun_out/blocks_traffic/prof_${tag}_$c -- python3 $ROOTDIR/tools/device_blocks.py $C $N $cfg > $ROOTDIR/gpurun_out/blocks_traffic/${tag}_$c.log 2>&1)
  done
done
python3 - $C $N $B $NB $W <<'PY'
import csv, glob, sys
C, N, B, NB, W = (int(v) for v in sys.argv[1:6])
for tag, what, n in (("%d_%d" % (B, NB), "%d-sample blocks, read by the analysis kernels themselves" % B, B), ("%d_%d_reblock" % (B, NB), "%d-sample blocks, every call re-blocked first" % B, B),
                     ("%d_%d" % (W, NB), "%d-sample blocks = whole hops in place" % W, W)):
    per = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("gpurun_out/blocks_traffic/prof_%s_%s/**/*counter_collection.csv" % (tag, c), recursive=True):
            for row in csv.DictReader(open(f)):
                if "fxk::" in row["Kernel_Name"] and row["Counter_Name"] == c:
                    k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                    d = per.setdefault(k, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0})
                    d[c] += float(row["Counter_Value"]); d["n"] += c == "FETCH_SIZE"
    calls = 3 * NB                      # device_blocks.py: three passes of NB blocks
    sample_bytes = C * n * 4
    tot_r = sum(2048.0 * d["FETCH_SIZE"] for d in per.values()) / calls
    tot_w = sum(1024.0 * d["WRITE_SIZE"] for d in per.values()) / calls
    print("%s (%d channels x %d-pt): per call %.3g B read + %.3g B written = %.3g B; the block itself is %.3g B: %.2f x" % (what, C, N, tot_r, tot_w, tot_r + tot_w, sample_bytes, (tot_r + tot_w) / sample_bytes))
    for k, d in sorted(per.items()):
        print("    %-70s %4d launches  %.3g B read  %.3g B written per launch" % (k[:70], d["n"], 2048.0 * d["FETCH_SIZE"] / max(d["n"], 1), 1024.0 * d["WRITE_SIZE"] / max(d["n"], 1)))
PY
rm -rf gpurun_out/blocks_traffic/prof_*
