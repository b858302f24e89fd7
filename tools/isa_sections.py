"""Static per-section instruction counts of fx_frame_kernel<N> from the FXMARK comments in the ISA."""
import re, subprocess, sys, os, tempfile
N = sys.argv[1] if len(sys.argv) > 1 else "1024"
PAIR = len(sys.argv) > 2 and sys.argv[2] == "pair"      # the pair kernel (2048 / 4096) instead of fx_frame_kernel
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = tempfile.mkdtemp()
sys.path.insert(0, os.path.join(root, "feature-extractor_amd"))
import build as fxbuild
subprocess.run(["hipcc"] + fxbuild.flags_for_window(int(N)) + ["-x", "hip", "-c", os.path.join(root, "feature-extractor_amd/csrc/fx_kernels.hip"), "-o", os.path.join(d, "fx.o"), "-save-temps"],
               cwd=d, stderr=subprocess.DEVNULL)
src = open(os.path.join(d, "fx_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
start = src.index(("_ZN3fxk14fx_pair_kernelILi%sEEEvNS_11FrameParamsE:" if PAIR else "_ZN3fxk15fx_frame_kernelILi%sELb1ELb1EEEvNS_11FrameParamsE:") % N)
body = src[start:src.index("s_endpgm", start)]
sec = "pre"; counts = {}; order = []
DUMP = sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "dump" else None    # print one section's ISA instead of the table
for line in body.splitlines():
    l = line.strip()
    m = re.match(r"; FXMARK (\w+)", l)
    if m:
        sec = m.group(1); continue
    if DUMP == sec and l and not l.startswith(";"): print(l.split(";")[0].rstrip())
    if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"): continue
    op = l.split()[0]
    if sec not in counts:
        counts[sec] = dict(v=0, pk=0, f64=0, s=0, ds=0, vmem=0); order.append(sec)
    c = counts[sec]
    if op.startswith("v_"):
        c["v"] += 1
        if op.startswith("v_pk_"): c["pk"] += 1
        if "f64" in op: c["f64"] += 1
    elif op.startswith("s_"): c["s"] += 1
    elif op.startswith("ds_"): c["ds"] += 1
    elif op.startswith(("global_", "buffer_", "scratch_", "flat_")): c["vmem"] += 1
if DUMP: sys.exit(0)
tot = dict(v=0, s=0, ds=0)
print("%-12s %6s %6s %6s %6s %6s %6s" % ("section", "valu", "pk", "f64", "salu", "lds", "vmem"))
for k in order:
    c = counts[k]
    print("%-12s %6d %6d %6d %6d %6d %6d" % (k, c["v"], c["pk"], c["f64"], c["s"], c["ds"], c["vmem"]))
    for t in tot: tot[t] += c[t]
print(tot)
