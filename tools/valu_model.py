"""Writes profiles/valu_model.json: per window size, the frame kernel's VALU instructions per frame (measured:
rocprofv3 SQ_INSTS_VALU of a bench.py run / frames per launch) and their mean issue cost (static opcode mix of the
kernel's ISA priced with tools/ubench/valu_rates.hip's per-class wave-instruction costs at 4 waves per SIMD).
bench.py turns that into roofline.valu_issue_frac = time the kernel's VALU instructions need / kernel time.

    python tools/valu_model.py <window>:<summary.json>:<frames per launch> ...
e.g. python tools/valu_model.py 1024:gpurun_out/prof_r02/summary.json:524288 2048:gpurun_out/prof_r02_2048/summary.json:131072
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# ns per wave-instruction per SIMD (tools/ubench, MI355X, 4 waves resident)
FAST_NS, SLOW_NS = 1.1, 1.85
FAST = re.compile(r"^v_(mul|add|sub|subrev|fma|fmac|mac|mad)_f32(_e32|_e64)?$|^v_(add|sub|subrev)_(u32|i32|co_u32)(_e32|_e64)?$|^v_(and|or|xor|mov)_b32(_e32|_e64)?$|^v_add3_u32$|^v_lshl_add_u32$")


def static_mix(window):
    d = tempfile.mkdtemp()
    sys.path.insert(0, os.path.join(ROOT, "feature-extractor_amd"))
    import importlib
    build = importlib.import_module("build")
    subprocess.run([build._hipcc()] + build.flags_for_window(window) + ["-x", "hip", "-c", os.path.join(build.CSRC, "fx_kernels.hip"), "-o", os.path.join(d, "fx.o"), "-save-temps"],
                   cwd=d, stderr=subprocess.DEVNULL, check=True)
    src = open(os.path.join(d, "fx_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    start = src.index("_ZN3fxk15fx_frame_kernelILi%dELb1ELb1EEEvNS_11FrameParamsE:" % window)
    body = src[start:src.index("s_endpgm", start)]
    fast = slow = 0
    for line in body.splitlines():
        t = line.strip()
        if not t.startswith("v_"):
            continue
        op = t.split()[0]
        if "_dpp" in op or " quad_perm" in t or " row_" in t or " wave_shr" in t:
            slow += 1
        elif FAST.match(op):
            fast += 1
        else:
            slow += 1
    return fast, slow


def main():
    out_path = os.path.join(ROOT, "profiles", "valu_model.json")
    try:
        model = json.load(open(out_path))
    except Exception:
        model = {}
    for arg in sys.argv[1:]:
        window, path, frames = arg.split(":")
        summ = json.load(open(path))
        valu = summ["pmc_per_launch"]["SQ_INSTS_VALU"] / float(frames)
        fast, slow = static_mix(int(window))
        mean_ns = (fast * FAST_NS + slow * SLOW_NS) / (fast + slow)
        model[window] = {"valu_per_frame": round(valu, 1), "mean_issue_ns": round(mean_ns, 3),
                         "static_mix": {"plain_fp32_int": fast, "packed_fp64_dpp_cmp_cvt": slow},
                         "source": "SQ_INSTS_VALU per launch from %s / %s frames; opcode mix from the kernel's ISA priced at %.2f / %.2f ns per "
                                   "wave-instruction per SIMD (tools/ubench/valu_rates.hip)" % (os.path.relpath(path, ROOT), frames, FAST_NS, SLOW_NS)}
        print(window, model[window])
    json.dump(model, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
