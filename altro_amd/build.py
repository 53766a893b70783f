"""Builds altro_amd/lib/libaltro_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m altro_amd.build [--force]
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "lib", "libaltro_hip.so")
CSRC = os.path.join(HERE, "csrc")


def sources():
    out = []
    for d, _, files in os.walk(CSRC):
        out += [os.path.join(d, f) for f in files if f.endswith((".hip", ".h", ".hpp", ".cpp", ".inc"))]
    for d, _, files in os.walk(os.path.join(ROOT, "include")):
        out += [os.path.join(d, f) for f in files if f.endswith((".h", ".hpp"))]
    return out


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in sources())


def build(force=False, verbose=False):
    """One object per translation unit, compiled in parallel (the kernel instantiations of the two element types
    and the TVLQR plans are separate units), then one link."""
    if not (force or stale()):
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objdir = os.path.join(HERE, "lib", "obj")
    os.makedirs(objdir, exist_ok=True)
    units = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    host = os.path.join(CSRC, "host")
    units += [os.path.join(host, f) for f in sorted(os.listdir(host)) if f.endswith(".cpp")]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
             "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]

    def embedded(src):
        """Files a unit pulls in with .incbin (ALTRO_EMBED in capi_rtc.hip: the sources handed to hiprtc).  The assembler
        reads them, not the preprocessor, so -MMD never lists them: they are dependencies all the same."""
        try:
            text = open(src).read()
        except OSError:
            return []
        return [os.path.join(CSRC, m) for m in re.findall(r'^ALTRO_EMBED\(\s*\w+\s*,\s*"([^"]+)"\s*\)', text, re.M)]

    def fresh(src, obj, dep):
        """An object is reused when neither its source, nor any header its last compile read (-MMD), nor any file it
        embeds with .incbin is newer."""
        if force or not (os.path.exists(obj) and os.path.exists(dep)):
            return False
        t = os.path.getmtime(obj)
        try:
            words = open(dep).read().replace("\\\n", " ").split()
        except OSError:
            return False
        files = [w for w in words[1:] if not w.endswith(":")] + embedded(src)
        return all(os.path.exists(f) and os.path.getmtime(f) <= t for f in files) and bool(files)

    def compile_one(src):
        # object names follow the path below csrc/ (host/x.cpp and x.hip must not collide)
        obj = os.path.join(objdir, os.path.splitext(os.path.relpath(src, CSRC))[0].replace(os.sep, "__") + ".o")
        dep = obj[:-2] + ".d"
        if fresh(src, obj, dep):
            return obj
        cmd = [hipcc] + flags + ["-MMD", "-MF", dep, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, units))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-Wl,-soname,libaltro_hip.so"] + objs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
