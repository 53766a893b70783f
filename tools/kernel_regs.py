"""Register / LDS / scratch use of every kernel of one translation unit, from hipcc's own remarks.

    python tools/kernel_regs.py altro_amd/csrc/ilqr_launch_mfma16.hip [substring ...]

Compiles the unit for gfx950 with -Rpass-analysis=kernel-resource-usage (no GPU needed) and prints one line per kernel
whose demangled name contains any of the substrings (all kernels when none is given).
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return out[:len(names)]
    except OSError:
        return names


def main():
    src = sys.argv[1]
    subs = sys.argv[2:]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "altro_amd", "csrc"),
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    names = demangle([b.split()[0] for b in blocks])
    print("%-100s %5s %5s %6s %8s %4s %7s" % ("kernel", "VGPR", "AGPR", "spill", "scratch", "occ", "LDS"))
    for b, dn in zip(blocks, names):
        if subs and not any(s in dn for s in subs):
            continue

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return m.group(1) if m else "?"
        short = re.sub(r"\(altro_hip::.*$", "", dn).replace("void altro_hip::", "")
        print("%-100s %5s %5s %6s %8s %4s %7s" % (short[:100], g("VGPRs"), g("AGPRs"), g("VGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"),
                                                   g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
    if "error:" in txt:
        print(txt[-3000:])


if __name__ == "__main__":
    main()
