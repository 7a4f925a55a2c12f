"""Build the HIP extension in-tree:  python -m simfire_amd.build"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# four translation units, compiled side by side (the first: host code + every kernel every handle launches; the others: k_run
# instantiations only some handles launch - teams / the closed loop, several bitmap words per thread, two-word teams)
SOURCES = ["simfire_hip.hip", "simfire_hip_run2.hip", "simfire_hip_run3.hip", "simfire_hip_run4.hip"]
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


def _compile_cmds(out, extra=()):
    """One `hipcc -c` per translation unit + the link."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tag = os.path.splitext(os.path.basename(out))[0]
    objs = [os.path.join(CSRC, f".{tag}.{os.path.splitext(src)[0]}.o") for src in SOURCES]
    compile_flags = [f for f in FLAGS if f != "-shared"]
    compiles = [[hipcc] + compile_flags + list(extra) + ["-c", "-o", obj, os.path.join(CSRC, src)] for src, obj in zip(SOURCES, objs)]
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
    return compiles, link, objs


def variant_out(v):
    from . import _lib
    return os.path.join(CSRC, _lib.VARIANTS[v])


def build(force=False, verbose=False, variants=()):
    """Product library + the named test-only variants (all compiles run side by side, then the links)."""
    targets = []
    if force or needs_build(OUT):
        targets.append((OUT, ()))
    for v in variants:
        if force or needs_build(variant_out(v)):
            targets.append((variant_out(v), VARIANT_FLAGS[v]))
    plans = [_compile_cmds(out, extra) for out, extra in targets]
    procs = []
    for compiles, _, _ in plans:
        for cmd in compiles:
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    for _, link, objs in plans:
        if verbose:
            print(" ".join(link))
        subprocess.check_call(link)
        for o in objs:
            os.remove(o)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, variants=tuple(VARIANT_FLAGS) if "--all" in sys.argv else ()))
