"""tools/mkvariant.py -- build link_amd/lib/variants/lib_<NAME>.so: the library with extra -D flags on SOME translation units
(the others are taken from the last full build's objects), for A/B runs of prebuilt variants on one GPU box (tools/ab_bench.sh).
usage: python tools/mkvariant.py NAME "-DDC_K2Q_CW=4 -DDC_K2Q_PMAP=0" [dense_fused.hip dense_fused_f16.hip ...]
Default translation units: the three dense_fused*.hip."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from link_amd import build as B

name, flags = sys.argv[1], sys.argv[2].split()
tus = sys.argv[3:] or ["dense_fused.hip", "dense_fused_f16.hip", "dense_fused_bf16.hip"]
B.build()                                   # the default objects are current
vdir = os.path.join(B.LIBDIR, "variants")
odir = os.path.join(vdir, "obj_" + name)
os.makedirs(odir, exist_ok=True)
cc = B._hipcc()


def one(src):
    obj = os.path.join(odir, src.replace(".hip", ".o"))
    subprocess.check_call([cc] + B.FLAGS + flags + ["-c", os.path.join(B.CSRC, src), "-o", obj])
    return obj


with ThreadPoolExecutor(max_workers=len(tus)) as ex:
    new = dict(zip(tus, ex.map(one, tus)))
objs = [new.get(s, os.path.join(B.OBJDIR, s.replace(".hip", ".o"))) for s in B.SOURCES]
so = os.path.join(vdir, f"lib_{name}.so")
subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
print(so)
