#!/usr/bin/env python3
"""tools/kernel_regs.py -- registers / LDS / scratch of the gfx950 kernels inside an object file or the built library.

    python tools/kernel_regs.py [file.o | liblink_amd.so] [substring ...]

Reads the code object's metadata notes (llvm-readelf): vgpr, agpr, sgpr, static LDS, scratch bytes, workgroup size.
What the persistent batch kernels' co-residency argument is checked against, and what
tests/test_cpu_abi.py::test_batch_kernels_resource_shape asserts.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.split("\n")[:len(names)]
    except Exception:
        return names


def kernel_table(path):
    """[(demangled name, vgpr, agpr, sgpr, lds_static, scratch, max_wg)] of every gfx950 kernel in `path`."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", path],
                              stderr=subprocess.DEVNULL)
        blob = open(fat, "rb").read()
        # a shared library concatenates the fat binaries of its objects: split at the bundler magic
        # (plain bundles start with the bundler's magic, compressed ones -- hipcc --offload-compress -- with "CCOB"; the bundler inflates them itself)
        starts = sorted(m.start() for m in re.finditer(rb"__CLANG_OFFLOAD_BUNDLE__|CCOB", blob))
        rows = []
        for i, s in enumerate(starts):
            part = os.path.join(td, f"part{i}.bin")
            with open(part, "wb") as f:
                f.write(blob[s:(starts[i + 1] if i + 1 < len(starts) else len(blob))])
            co = os.path.join(td, f"part{i}.co")
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True)
            if r.returncode != 0 or not os.path.exists(co):
                continue
            txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            for e in re.split(r"\n\s+- \.agpr_count", txt)[1:]:
                e = ".agpr_count" + e
                g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, e) or [None, "?"])[1]
                rows.append([g("name"), g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("group_segment_fixed_size"),
                             g("private_segment_fixed_size"), g("max_flat_workgroup_size")])
        names = _demangle([r[0] for r in rows])
        return [(n, *r[1:]) for n, r in zip(names, rows)]


def main():
    args = sys.argv[1:]
    path = os.path.join(ROOT, "link_amd", "lib", "liblink_amd.so")
    if args and os.path.exists(args[0]):
        path, args = args[0], args[1:]
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'lds':>7} {'scratch':>7} {'wg':>5}  kernel")
    for name, v, a, s, lds, scr, wg in kernel_table(path):
        if args and not any(k in name for k in args):
            continue
        print(f"{v:>5} {a:>5} {s:>5} {lds:>7} {scr:>7} {wg:>5}  {name[:150]}")


if __name__ == "__main__":
    main()
