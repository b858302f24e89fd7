"""Registers, scratch and LDS of every kernel in the library AS BUILT: read from the code objects inside feature-extractor_amd/lib/libfx_hip.so
(the .hip_fatbin section holds one clang offload bundle per translation unit; each bundle's gfx950 code object carries the kernels'
metadata notes), not from a separate compilation.

    python tools/kernel_resources.py                      # the table
    python tools/kernel_resources.py --write profiles/r05_resources.txt

tests/test_docs_cpu.py holds DESIGN.md section 3.8's scratch / VGPR statements to this dump.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def demangle(names):
    try:
        exe = os.path.join(LLVM, "llvm-cxxfilt")
        out = subprocess.run([exe if os.path.exists(exe) else "c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def kernels_of(library=None):
    """[{name, vgpr, agpr, sgpr, scratch, lds, max_threads}] for every kernel in the library's fat binary"""
    library = library or os.path.join(ROOT, "feature-extractor_amd", "lib", "libfx_hip.so")
    out = []
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, library], check=True, capture_output=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)] + [len(blob)]
        for i in range(len(starts) - 1):
            piece = os.path.join(d, "bundle%d" % i)
            open(piece, "wb").write(blob[starts[i]:starts[i + 1]])
            co = piece + ".co"
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + piece,
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            cur = None
            for line in notes.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
                if not m:
                    continue
                key, val = m.group(1), m.group(2).strip().strip("'\"")
                if key == "agpr_count":                 # first key of a kernel's map (keys are sorted)
                    cur = {"agpr": int(val)}
                    out.append(cur)
                elif cur is not None:
                    if key == "name":
                        cur["name"] = val
                    elif key in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size"):
                        cur[{"vgpr_count": "vgpr", "sgpr_count": "sgpr", "private_segment_fixed_size": "scratch",
                             "group_segment_fixed_size": "lds", "max_flat_workgroup_size": "max_threads"}[key]] = int(val)
    out = [k for k in out if "name" in k and "vgpr" in k]
    pretty = demangle([k["name"] for k in out])
    for k in out:
        k["pretty"] = re.sub(r"\(.*\)$", "", pretty[k["name"]]).replace("void ", "")
    return sorted(out, key=lambda k: k["pretty"])


def table(kernels):
    lines = ["kernel resources of feature-extractor_amd/lib/libfx_hip.so as built (tools/kernel_resources.py; gfx950 code objects' metadata)",
             "static LDS only: the frame / pair / hop kernels take their LDS dynamically (FrameLds<N>::bytes, fx_capi.cpp picks the shape)", "",
             "%-72s %5s %5s %5s %8s %8s %8s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "LDS", "threads")]
    for k in kernels:
        lines.append("%-72s %5d %5d %5d %7dB %7dB %8d" % (k["pretty"][:72], k["vgpr"], k.get("agpr", 0), k.get("sgpr", 0), k.get("scratch", 0),
                                                          k.get("lds", 0), k.get("max_threads", 0)))
    spills = [k for k in kernels if k.get("scratch", 0)]
    lines += ["", "kernels with scratch: %d" % len(spills)] + ["  %s: %d B per lane" % (k["pretty"], k["scratch"]) for k in spills]
    return "\n".join(lines) + "\n"


# the kernels DESIGN.md section 3.8 tabulates (the analysis kernels; the offline analyser's and the byte mover are in the full dump)
DESIGN_PREFIXES = ("fxk::fx_frame_kernel<", "fxk::fx_frame_tail_kernel<", "fxk::fx_pair_kernel<", "fxk::fx_hop_kernel<", "fxk::fx_hop_pair_kernel<",
                   "fxk::fx_finalise_kernel", "fxk::fx_epilogue_kernel", "fxk::fx_history_kernel", "fxk::fx_tail_fused_kernel")


def markdown(kernels):
    """the rows of DESIGN.md's "registers and scratch as built" table"""
    rows = ["| kernel | VGPRs | scratch (B / lane) |", "|---|---|---|"]
    for k in kernels:
        if k["pretty"].startswith(DESIGN_PREFIXES):
            rows.append("| `%s` | %d | %d |" % (k["pretty"], k["vgpr"], k.get("scratch", 0)))
    return "\n".join(rows) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", default=None)
    ap.add_argument("--markdown", action="store_true", help="print the table DESIGN.md section 3.8 carries (tests/test_docs_cpu.py compares the two)")
    ap.add_argument("--library", default=None)
    args = ap.parse_args()
    if args.markdown:
        sys.stdout.write(markdown(kernels_of(args.library)))
        return
    text = table(kernels_of(args.library))
    if args.write:
        open(args.write, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
