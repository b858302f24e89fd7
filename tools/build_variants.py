"""Here (no GPU needed): build experiment variants of libfx_hip.so into feature-extractor_amd/lib/variants/<name>.so, so that the
GPU box spends its minutes measuring, not compiling.  Only the kernel unit the flags concern is recompiled (small = windows
<= 1024, large = 2048 / 4096, hop); the rest is linked from the shipped objects.
Usage: python tools/build_variants.py small|large|hop|offline name=flags [name=flags ...]        (flags: quoted, space separated)
On the box a variant is selected by path: FX_LIBRARY_OVERRIDE=feature-extractor_amd/lib/variants/<name>.so (feature-extractor_amd/capi.py); the shipped
library is never overwritten (tools/ab_variants.sh, ab_blocks.sh, section_costs.sh, fma_experiment.sh do that)."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "feature-extractor_amd"))
import build as fxbuild
fxbuild.build()
unit = {"small": 0, "large": 1, "hop": 2, "offline": 3}[sys.argv[1]]
src, objname, extra = fxbuild.UNITS[unit]
vdir = os.path.join(fxbuild.LIB_DIR, "variants")
os.makedirs(vdir, exist_ok=True)
def one(arg):
    name, _, flags = arg.partition("=")
    obj = os.path.join(vdir, name + ".o")
    fxbuild._run([fxbuild._hipcc()] + fxbuild.HIPCC_FLAGS + extra + flags.split() + ["-x", "hip", "-c", os.path.join(fxbuild.CSRC, src), "-o", obj])
    objs = [obj if o == objname else os.path.join(fxbuild.LIB_DIR, o) for _, o, _ in fxbuild.UNITS]
    subprocess.check_call([fxbuild._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(vdir, name + ".so")] + objs + ["-ldl", "-lpthread"])
    os.remove(obj)
    return name
with ThreadPoolExecutor(max_workers=4) as pool:
    for n in pool.map(one, sys.argv[2:]): print("built", n, flush=True)
