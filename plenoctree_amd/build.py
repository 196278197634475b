"""Builds the in-tree gfx950 shared library (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libplenoctree_hip.so")
SOURCES = ["pxo_api.hip", "mlp_kernels.hip", "mlp_x3_kernels.hip", "mlp_x6_kernels.hip", "wgrad_kernels.hip", "wgrad_x6_kernels.hip", "render_kernels.hip", "optim_kernels.hip",
           "octree_kernels.hip"]
# per-source flags: the octree marchers are compiled without fused contraction (see the file header)
SOURCE_FLAGS = {"octree_kernels.hip": ["-ffp-contract=off"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


STAMP = os.path.join(HERE, "libplenoctree_hip.stamp")   # travels with the .so (git-ignored, not gpurun-ignored)


def _source_hash():
    """sha256 over every file the library is compiled from, plus the flags."""
    import hashlib
    h = hashlib.sha256(repr((FLAGS, sorted(SOURCE_FLAGS.items()))).encode())
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    deps.append(os.path.join(HERE, "..", "include", "plenoctree_hip.h"))
    deps.append(os.path.join(HERE, "..", "include", "plenoctree_octree.h"))
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale():
    """The library is current iff the stamp written next to the objects matches the sources' hash."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != _source_hash()


def build(force=False, verbose=True, extra_flags=(), suffix=""):
    """hipcc --offload-arch=gfx950 -> plenoctree_amd/libplenoctree_hip<suffix>.so
    (`suffix`/`extra_flags` build A/B variants of the kernels, selected at run time with PXO_LIB)."""
    global LIB
    if suffix:
        lib_main, LIB = LIB, LIB.replace(".so", suffix + ".so")
        try:
            return _build(True, verbose, extra_flags, "build" + suffix)
        finally:
            LIB = lib_main
    if not force and not _stale():
        return LIB
    return _build(force, verbose, extra_flags, "build")


def _build(force, verbose, extra_flags, objdir):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, objdir), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, *FLAGS, *SOURCE_FLAGS.get(src, []), *extra_flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out and verbose:
            sys.stdout.write(out.decode(errors="replace"))
        if p.returncode != 0:
            failed = True
            print(f"hipcc failed on {src}", file=sys.stderr)
    if failed:
        raise RuntimeError("hipcc build failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    if objdir == "build":
        with open(STAMP, "w") as f:
            f.write(_source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
