"""link_amd/build.py -- compile the HIP sources into link_amd/lib/liblink_amd.so (gfx950 only).

One plain `hipcc --offload-arch=gfx950 -shared -fPIC` invocation per translation unit + one link;
no torch involved (the library's ABI is include/link_amd.h: pointers, sizes, a stream).  hipcc
cross-compiles without a GPU, so this runs in the build container; the .so travels to the GPU box
with the snapshot (it is git-ignored, not gpurun-ignored).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
SO = os.path.join(LIBDIR, "liblink_amd.so")
SOURCES = ["ops.hip", "index.hip", "aggregate.hip", "elk.hip", "conv.hip", "conv_pairs.hip", "bn.hip", "dense.hip", "dense_fused.hip", "dense_fused_f16.hip", "dense_fused_bf16.hip",
           "dense_tiles.hip", "dense_tiles_f16.hip", "dense_tiles_bf16.hip", "elk_tiles.hip", "elk_tiles_f16.hip", "elk_tiles_bf16.hip",
           "elk_lean.hip", "elk_lean_f16.hip", "elk_lean_bf16.hip", "block.hip", "dense_batch.hip", "dense_batch_f16.hip", "dense_batch_bf16.hip"]
# --offload-compress: the gfx950 code objects are stored compressed in the fat binary (25.4 -> 6.1 MB; the HIP runtime inflates them when
# the library is loaded: +0.2 s on the first call of a process, measured on the GPU box)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--offload-compress", "-I" + os.path.join(ROOT, "include"),
         "-I" + CSRC, "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("LINK_AMD_CXXFLAGS", "").split()      # A/B experiments (-DNAME=value); part of the build stamp


def _hipcc():
    for p in ("/opt/rocm/bin/hipcc", "hipcc"):
        if p == "hipcc" or os.path.exists(p):
            return p


def _stamp():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)) + ["../../include/link_amd.h"]:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode()); h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _obj_key(src, dep_file):
    """Hash of one translation unit's inputs: the files its last compile read (hipcc -MD) + the flags."""
    deps = [os.path.join(CSRC, src)]
    if os.path.exists(dep_file):
        txt = open(dep_file).read().replace("\\\n", " ")
        deps = [d for d in txt.split(":", 1)[1].split() if d.startswith(ROOT)] or deps
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for d in sorted(set(deps)):
        if not os.path.exists(d):
            return None
        with open(d, "rb") as f:
            h.update(d.encode()); h.update(f.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    stamp_file = os.path.join(LIBDIR, "stamp")
    stamp = _stamp()
    if not force and os.path.exists(SO) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return SO
    cc = _hipcc()

    def one(src):
        # per-object rebuild: an object is kept while the files its last compile read (its .d list) and the flags are unchanged
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        dep, key_file = obj + ".d", obj + ".key"
        key = _obj_key(src, dep)
        if not force and key and os.path.exists(obj) and os.path.exists(key_file) and open(key_file).read() == key:
            return obj
        cmd = [cc] + FLAGS + ["-MD", "-MF", dep, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        key = _obj_key(src, dep)
        if key:
            with open(key_file, "w") as f:
                f.write(key)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(one, SOURCES))
    subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs)
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
