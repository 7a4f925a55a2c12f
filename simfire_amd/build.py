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
VARIANT_FLAGS = {"sow": ["-DSF_STORE_ORDER_WAIT"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value"]


def source_hash(extra=()):
    """sha256 over every source the library is built from, the flags of THIS target (base + the variant's extra ones) and the
    compiler's version (what a `.srchash` file beside a built library records)."""
    import hashlib
    h = hashlib.sha256()
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    deps.append(os.path.join(os.path.dirname(_HERE), "include", "simfire_hip.h"))
    lab = os.path.join(os.path.dirname(_HERE), "include", "simfire_hip_lab.h")
    if os.path.exists(lab):
        deps.append(lab)
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(" ".join(list(FLAGS) + list(extra)).encode())
    h.update(hipcc_version().encode())
    return h.hexdigest()


def needs_build(out=OUT, extra=()):
    """True unless `out` exists and was built from exactly these sources with exactly these flags by this compiler (a hash stored
    beside it, not modification times)."""
    if os.environ.get("SF_FORCE_BUILD") == "1" or not os.path.exists(out):
        return True
    try:
        with open(out + ".srchash") as f:
            return f.read().strip() != source_hash(extra)
    except OSError:
        return True


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


def hipcc_version():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    try:
        out = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
        return next((ln.strip() for ln in out.splitlines() if "HIP version" in ln), out.splitlines()[0].strip() if out else "?")
    except OSError:
        return "hipcc not found"


def build(force=False, verbose=False, variants=(), jobs=None):
    """Product library + the named test-only variants: at most `jobs` compiles side by side (default: the CPUs), then the links.
    Every compile is waited for even if one fails, and the object files are removed whatever happens."""
    targets = []
    if force or needs_build(OUT):
        targets.append((OUT, ()))
    for v in variants:
        if force or needs_build(variant_out(v), VARIANT_FLAGS[v]):
            targets.append((variant_out(v), VARIANT_FLAGS[v]))
    plans = [_compile_cmds(out, extra) for out, extra in targets]
    jobs = jobs or max(1, min(os.cpu_count() or 1, 8))
    pending = [cmd for compiles, _, _ in plans for cmd in compiles]
    running, failed = [], None
    try:
        while pending or running:
            while pending and len(running) < jobs and failed is None:
                cmd = pending.pop(0)
                if verbose:
                    print(" ".join(cmd))
                running.append((cmd, subprocess.Popen(cmd)))
            if not running:
                break
            cmd, p = running.pop(0)
            if p.wait() != 0 and failed is None:
                failed = (p.returncode, cmd)
        if failed is not None:
            raise subprocess.CalledProcessError(*failed)
        for (out, extra), (_, link, _) in zip(targets, plans):
            if verbose:
                print(" ".join(link))
            if os.path.exists(out + ".srchash"):
                os.remove(out + ".srchash")                     # (a failed link must not leave the old library looking current)
            subprocess.check_call(link)
            with open(out + ".srchash", "w") as f:
                f.write(source_hash(extra) + "\n")
    finally:
        for _, p in running:
            p.kill()
            p.wait()
        for _, _, objs in plans:
            for o in objs:
                if os.path.exists(o):
                    os.remove(o)
        # hipcc's temporaries of an interrupted compile of THIS build's objects (another process may be building beside this one:
        # parallel test workers, a variant build - its temporaries are not ours to delete)
        mine = tuple(os.path.basename(o) for _, _, objs in plans for o in objs)
        for f in os.listdir(CSRC):
            if f.endswith(".tmp") and f.startswith(mine) and mine:
                os.remove(os.path.join(CSRC, f))
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, variants=tuple(VARIANT_FLAGS) if "--all" in sys.argv else ()))
