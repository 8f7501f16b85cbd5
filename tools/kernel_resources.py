"""Per-kernel register / scratch / LDS use of the built library, read from the code objects inside lyssandra_amd/build/*.o
(llvm-objcopy the .hip_fatbin section, clang-offload-bundler --unbundle, llvm-readelf --notes): no recompilation.
usage: python tools/kernel_resources.py [substring ...]     (kernels whose demangled-ish name contains every substring)
Round 3 lost performance twice to scratch nobody had asked for (a 4-element vector indexed at run time, a loop over
mutable phase state): tests/test_abi.py::test_headline_kernels_have_no_scratch keeps the product kernels at zero."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(objdir=None):
    """[{name, vgpr, spill, scratch, lds, sgpr, object}] for every kernel of every object file."""
    objdir = objdir or os.path.join(ROOT, "lyssandra_amd", "build")
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for o in sorted(glob.glob(os.path.join(objdir, "*.o"))):
            fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
            for f in (fat, co):
                if os.path.exists(f):
                    os.remove(f)
            subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", o, fat], check=True)
            if not os.path.exists(fat) or os.path.getsize(fat) == 0:
                continue  # no device code in this object
            subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--input=" + fat, "--output=" + co, "--unbundle"], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], check=True, stdout=subprocess.PIPE,
                                   text=True).stdout
            for blk in re.findall(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", notes, re.S):
                g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, blk).group(1))
                out.append({"name": re.search(r"\.name:\s+(\S+)", blk).group(1), "vgpr": g("vgpr_count"),
                            "spill": g("vgpr_spill_count"), "scratch": g("private_segment_fixed_size"),
                            "lds": g("group_segment_fixed_size"), "sgpr": g("sgpr_count"), "object": os.path.basename(o)})
    return out


if __name__ == "__main__":
    want = sys.argv[1:]
    rows = [k for k in kernels() if all(w in k["name"] for w in want)]
    for k in sorted(rows, key=lambda r: (-r["scratch"], r["name"])):
        print("%-12s vgpr %3d spill %3d scratch %4d lds %6d  %s" % (k["object"], k["vgpr"], k["spill"], k["scratch"],
                                                                    k["lds"], k["name"][:110]))
    print("%d kernels, %d with scratch" % (len(rows), sum(1 for k in rows if k["scratch"] > 0)))
