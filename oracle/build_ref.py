"""oracle/build_ref.py -- compile the reference's own CPU ops into oracle/_ref/ (TEST INFRA ONLY).

Recipe (runs only where /root/reference exists, i.e. in the build container; the GPU box only uses
the prebuilt oracle/_ref/ref_backend.so that travels with the snapshot):

  g++ (via torch.utils.cpp_extension.load) on the reference sources WHERE THEY LIE:
      backend/hash/hash_cpu.cpp  backend/others/count_cpu.cpp
      backend/voxelize/voxelize_cpu.cpp  backend/devoxelize/devoxelize_cpu.cpp
  + oracle/ref_bind.cpp (our binding) + oracle/link_oracle.c (for the query restatement),
  flags -O3 -fopenmp as the reference's setup.py:30-33 uses.  Output: oracle/_ref/ref_backend.so.

Not built: backend/others/query_cpu.cpp + backend/hashmap/* (need Google sparsehash, absent in this
image -> unbuildable here; see ref_bind.cpp) and every *.cu (CUDA; no nvcc, and not to be hipified).
No reference source is copied into the repo; oracle/_ref/ is git-ignored.
"""
import os
import shutil
import sys

REF = "/root/reference/segmentation/torchsparse-u/torchsparse/backend"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
SO = os.path.join(OUT, "ref_backend.so")


def available() -> bool:
    return os.path.isdir(REF)


def build(force: bool = False) -> str:
    if not available():
        if os.path.exists(SO):
            return SO
        raise RuntimeError("reference tree not present and no prebuilt oracle/_ref/ref_backend.so")
    srcs = [os.path.join(REF, p) for p in ("hash/hash_cpu.cpp", "others/count_cpu.cpp",
                                           "voxelize/voxelize_cpu.cpp", "devoxelize/devoxelize_cpu.cpp")]
    mine = [os.path.join(HERE, "ref_bind.cpp"), os.path.join(HERE, "link_oracle.c")]
    newest = max(os.path.getmtime(p) for p in srcs + mine)
    if not force and os.path.exists(SO) and os.path.getmtime(SO) >= newest:
        return SO
    os.makedirs(os.path.join(OUT, "build"), exist_ok=True)
    from torch.utils.cpp_extension import load
    load(name="ref_backend", sources=srcs + mine, extra_cflags=["-O3", "-fopenmp"],
         extra_ldflags=["-lgomp"], extra_include_paths=[REF],
         build_directory=os.path.join(OUT, "build"), verbose=False)
    shutil.copy(os.path.join(OUT, "build", "ref_backend.so"), SO)
    return SO


def load_module():
    """Import oracle/_ref/ref_backend.so as a Python module (needs torch imported first)."""
    import importlib.util
    import torch  # noqa: F401  (libtorch symbols)
    spec = importlib.util.spec_from_file_location("ref_backend", build())
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def import_reference_python():
    """Import the reference's Python layer (torchsparse package + segmentation/core) with
    `torchsparse.backend` bound to oracle/_ref.  Build-container only."""
    import types
    mod = load_module()
    ts_root = "/root/reference/segmentation/torchsparse-u"
    seg_root = "/root/reference/segmentation"
    for p in (seg_root, ts_root):
        if p not in sys.path:
            sys.path.insert(0, p)
    sys.modules["torchsparse.backend"] = mod
    import torchsparse
    torchsparse.backend = mod
    return torchsparse, mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
