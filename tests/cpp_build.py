"""Builds the C++ test programs under tests/cpp/ against libaltro_hip.so (g++, no GPU needed)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
LIBDIR = os.path.join(ROOT, "altro_amd", "lib")


def build(name, extra_sources=(), extra_link=(), defines=(), out_name=None):
    src = os.path.join(CPP, name + ".cpp")
    out = os.path.join(CPP, (out_name or name) + ".bin")
    deps = [src, os.path.join(LIBDIR, "libaltro_hip.so")] + [os.path.join(ROOT, s) for s in extra_sources]
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps if os.path.exists(d)):
        return out
    cmd = ["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include")] + ["-D" + d for d in defines] + [src] + \
          [os.path.join(ROOT, s) for s in extra_sources] + \
          ["-L" + LIBDIR, "-laltro_hip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,$ORIGIN/../../altro_amd/lib"] + list(extra_link) + ["-o", out]
    subprocess.check_call(cmd)
    return out


def run(name, extra_sources=(), timeout=300, extra_link=(), args=(), defines=(), out_name=None):
    exe = build(name, extra_sources, extra_link, defines, out_name)
    p = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout)
    return p.returncode, p.stdout, p.stderr
