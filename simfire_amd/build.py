"""Build the HIP extension in-tree:  python -m simfire_amd.build"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["simfire_hip.hip"]
OUT = os.path.join(CSRC, "libsimfire_hip.so")
# builds of the same sources that only tests load (simfire_amd/_lib.py: VARIANTS)
VARIANT_FLAGS = {"exp": ["-DSF_EXPERIMENTAL"], "sow": ["-DSF_STORE_ORDER_WAIT"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value"]


def needs_build(out=OUT):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps.append(os.path.join(os.path.dirname(_HERE), "include", "simfire_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def _cmd(out, extra=()):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return [hipcc] + FLAGS + list(extra) + ["-o", out] + [os.path.join(CSRC, s) for s in SOURCES]


def variant_out(v):
    from . import _lib
    return os.path.join(CSRC, _lib.VARIANTS[v])


def build(force=False, verbose=False, variants=()):
    """Product library + the named test-only variants (the compiles run side by side)."""
    jobs = []
    if force or needs_build(OUT):
        jobs.append(_cmd(OUT))
    for v in variants:
        if force or needs_build(variant_out(v)):
            jobs.append(_cmd(variant_out(v), VARIANT_FLAGS[v]))
    procs = []
    for cmd in jobs:
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, variants=tuple(VARIANT_FLAGS) if "--all" in sys.argv else ()))
