"""Instruction mix per kernel of the built library (disassembly of the code objects inside lyssandra_amd/build/*.o, no
recompilation): VALU / SALU / v_cndmask / index-mode regions / packed FMAs.  Round 5 found two things this shows at a glance:
a run-time index into a vector of <= 8 floats becomes a chain of selects (v_cndmask share), and every s_set_gpr_idx_on / off
pair costs a dependent chain ~50 cycles.   usage: python tools/kernel_mix.py [substring ...]"""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def mixes(objdir=None):
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
                continue
            subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--input=" + fat, "--output=" + co, "--unbundle"], check=True, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL)
            dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, stdout=subprocess.PIPE,
                                 text=True).stdout
            cur = None
            for line in dis.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = {"name": m.group(1), "object": os.path.basename(o), "valu": 0, "salu": 0, "cnd": 0, "idx": 0, "pk": 0,
                           "lds": 0, "vmem": 0, "nop": 0}
                    out.append(cur)
                    continue
                t = line.strip().split()
                if cur is None or not t:
                    continue
                op = t[0]
                if op.startswith("v_"):
                    cur["valu"] += 1
                elif op.startswith("s_"):
                    cur["salu"] += 1
                elif op.startswith("ds_"):
                    cur["lds"] += 1
                elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
                    cur["vmem"] += 1
                if "cndmask" in op:
                    cur["cnd"] += 1
                if op == "s_set_gpr_idx_on":
                    cur["idx"] += 1
                if op.startswith("v_pk_"):
                    cur["pk"] += 1
                if op == "s_nop":
                    cur["nop"] += 1
    return out


if __name__ == "__main__":
    want = sys.argv[1:]
    rows = [k for k in mixes() if all(w in k["name"] for w in want) and k["valu"] > 0]
    for k in sorted(rows, key=lambda r: -r["cnd"] / max(1, r["valu"])):
        print("%-14s valu %5d salu %5d cndmask %5d (%4.1f %%) idx-regions %4d pk %5d lds %4d vmem %4d  %s" % (
            k["object"], k["valu"], k["salu"], k["cnd"], 100.0 * k["cnd"] / k["valu"], k["idx"], k["pk"], k["lds"], k["vmem"],
            k["name"][:90]))
