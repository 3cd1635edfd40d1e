"""Build libevflow_hip.so (gfx950) in-tree with hipcc.

    python -m event_flow_amd.build [--force]

The shared object lands next to this file so it travels with the source tree
to the GPU box (it is git-ignored, not gpurun-ignored)."""

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libevflow_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",      # bit-exact warp arithmetic (no implicit FMA)
    "-munsafe-fp-atomics",    # hardware global_atomic_add_f32
    "-Wall",
    "-Wno-unused-function",
]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "evflow.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every csrc/*.hip into one shared library.  Returns the path."""
    if not force and not _stale():
        return LIB
    objs = []
    procs = []
    for src in sources():
        obj = src[:-4] + ".o"
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def build_variant(name, defines, verbose=False):
    """A/B build: libevflow_<name>.so with extra -D flags for some sources ({"evf_x.hip": ["-DFOO=1"]}), every other object
    taken from the main build.  Loaded through EVF_LIB=<path> (measurements only; never the product path)."""
    build(verbose=verbose)
    vdir = os.path.join(CSRC, "_var_" + name)
    os.makedirs(vdir, exist_ok=True)
    objs, procs = [], []
    for src in sources():
        base = os.path.basename(src)
        if base in defines:
            obj = os.path.join(vdir, base[:-4] + ".o")
            procs.append((src, subprocess.Popen([HIPCC] + FLAGS + list(defines[base]) + ["-c", src, "-o", obj])))
        else:
            obj = src[:-4] + ".o"
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    out = os.path.join(HERE, f"libevflow_{name}.so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
