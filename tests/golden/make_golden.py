"""tests/golden/make_golden.py -- generate the committed golden fixtures (build container only).

Imports the REFERENCE itself -- its Python layer from /root/reference and its C++ CPU ops compiled
from where they lie by oracle/build_ref.py -- runs it on seeded inputs and stores inputs + outputs
as small .npz files in this directory.  Nothing of the reference (source, bytecode, .so) is written
here: fixtures are data only.  Run:  python tests/golden/make_golden.py

Provenance caveats recorded in every file's `meta` field:
  * `hash_query_cpu` is the oracle's restatement (sparsehash absent -> query_cpu.cpp unbuildable,
    see oracle/ref_bind.cpp); everything else on the r=2 path is reference code.
  * r=3: the reference CPU devoxelize hard-wires 8 neighbours (devoxelize_cpu.cpp:19-24), so the one
    call spdevoxelize is replaced by a torch restatement of the CUDA kernel's semantics
    (devoxelize_cuda.cu:21-33) -- SURVEY.md section 8c.  Round 6: every r=3 fixture ALSO holds the
    same forward with spdevoxelize = the reference's compiled CPU op on four 8-wide slices of the
    padded map (`out_refcpu`, `core_refcpu`; _CompiledDevox below): the r=3 forward is pinned on
    reference output, the restatement agrees with it to 2e-7.
  * gradient arrays: regenerating them reproduces the committed ones to 2e-6 relative, not bit for
    bit (thread-order rounding inside torch's CPU autograd kernels); the committed arrays were kept
    when the r=3 companions were added.
  * batch>0 neighbour hashes: the CPU kernel_hash twin has the data[3] defect (hash_cpu.cpp:29), so
    multi-batch neighbour maps are produced with the batch column patched per frame.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import build_ref  # noqa: E402

torchsparse, backend = build_ref.import_reference_python()
import torchsparse.nn.functional as TSF  # noqa: E402
from torchsparse import SparseTensor  # noqa: E402
from torchsparse.nn.utils import get_kernel_offsets  # noqa: E402
import core.models.utils as ref_utils  # noqa: E402
from core.models.semantic_kitti.linkunet import ELKBlock as RefELKBlockUNet  # noqa: E402
from core.models.semantic_kitti.linkencoder import ELKBlock as RefELKBlockEnc  # noqa: E402

META_COMMON = {
    "generator": "tests/golden/make_golden.py",
    "reference": "MCG-NJU/LinK @ 2024_08_07, imported from /root/reference",
    "hash_query_cpu": "oracle restatement (sparsehash absent; oracle/ref_bind.cpp)",
    "torch": torch.__version__,
}


def save(name, meta, **arrays):
    m = dict(META_COMMON)
    m.update(meta)
    np.savez_compressed(os.path.join(HERE, name), meta=np.array(json.dumps(m)), **arrays)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrays.items()})


def rand_coords(n, grid, seed, batches=1, lo=0):
    g = torch.Generator().manual_seed(seed)
    per = []
    for b in range(batches):
        lin = torch.randperm(grid ** 3, generator=g)[:n]
        x, y, z = lin % grid, (lin // grid) % grid, lin // (grid * grid)
        per.append(torch.stack([x + lo, y + lo, z + lo, torch.full_like(x, b)], 1))
    return torch.cat(per, 0).int()


# ------------------------------------------------------------------ devoxelize r=3 restatement
class _PatchedDevox:
    """Route spdevoxelize through a torch restatement of devoxelize_cuda.cu:21-33 (for r != 2)."""

    def __enter__(self):
        self.orig = TSF.spdevoxelize
        self.orig_u = ref_utils.F.spdevoxelize

        def spdevoxelize(feats, coords, weights, r=2):
            q = coords.long()
            return (feats[q.clamp(min=0)] * weights[..., None]).sum(1)

        ref_utils.F.spdevoxelize = spdevoxelize
        return self

    def __exit__(self, *a):
        ref_utils.F.spdevoxelize = self.orig_u


class _CompiledDevox:
    """r != 2 through the reference's COMPILED CPU op (round 6: the reference-OUTPUT pin of the r = 3 forward).
    devoxelize_forward_cpu hard-wires K = 8 neighbours per row (devoxelize_cpu.cpp:19-24), so the [N, r^3] map is padded to
    a multiple of 8 columns with (-1, weight 0) -- an absent neighbour, which the op skips (devoxelize_cpu.cpp:23) -- the op is
    called on each 8-wide slice, and the partial outputs are added.  Same products as the CUDA kernel's single 27-term loop
    (devoxelize_cuda.cu:21-33), summed in four groups instead of one chain: agreement to rounding, not bit for bit."""

    def __enter__(self):
        self.orig_u = ref_utils.F.spdevoxelize

        def spdevoxelize(feats, coords, weights, r=2):
            n, k = coords.shape
            pad = (-k) % 8
            idx = torch.cat([coords.int(), torch.full((n, pad), -1, dtype=torch.int32)], 1)
            w = torch.cat([weights.float(), torch.zeros(n, pad)], 1)
            out = None
            for j in range(0, k + pad, 8):
                part = backend.devoxelize_forward_cpu(feats.contiguous().float(), idx[:, j:j + 8].contiguous(),
                                                      w[:, j:j + 8].contiguous())
                out = part if out is None else out + part
            return out

        ref_utils.F.spdevoxelize = spdevoxelize
        return self

    def __exit__(self, *a):
        ref_utils.F.spdevoxelize = self.orig_u


def ref_aggregate(feats, coords, s, r, compiled=False):
    st = SparseTensor(feats.clone(), coords.clone(), 1)
    aux, idx, counts = ref_utils.voxel_to_aux(st, s)
    aux_f = aux.F.clone()
    if r == 2:
        out = ref_utils.aux_to_voxel(aux, st, idx, counts, r)
    else:
        with (_CompiledDevox() if compiled else _PatchedDevox()):
            out = ref_utils.aux_to_voxel(aux, st, idx, counts, r)
    return aux_f, aux.C, idx, counts, out.F


def main():
    torch.manual_seed(0)

    # ---------------------------------------------------------------- G-hash
    kat = torch.tensor([[0, 0, 0, 0], [1, 2, 3, 0], [-1, 5, 7, 1], [255, 255, 255, 0], [1, 1, 1, 0],
                        [1, 1, 1, 1]], dtype=torch.int)
    g = torch.Generator().manual_seed(11)
    rnd = torch.randint(-2000, 2000, (4096, 4), generator=g).int()
    rnd[:, 3] = torch.randint(0, 4, (4096,), generator=g).int()
    off3 = get_kernel_offsets(3, 1, 1)
    off2 = get_kernel_offsets(2, 1, 1)
    single = rnd.clone()
    single[:, 3] = 0   # single-batch: CPU kernel_hash defect (hash_cpu.cpp:29) is invisible
    save("g_hash.npz", {"what": "sphash / sphash(offsets) / sphashquery / spcount known answers "
                                "from the reference CPU ops (hash_cpu.cpp, count_cpu.cpp)"},
         kat_coords=kat.numpy(), kat_hash=TSF.sphash(kat).numpy(),
         rnd_coords=rnd.numpy(), rnd_hash=TSF.sphash(rnd).numpy(),
         single_coords=single.numpy(),
         khash_r3=TSF.sphash(single, off3).numpy(), khash_r2=TSF.sphash(single, off2).numpy(),
         # batch>0 with the reference CPU defect reproduced verbatim (row 0's batch used everywhere)
         khash_r3_cpu_defect=TSF.sphash(rnd, off3).numpy(),
         query_q=np.array([5, 9, 11, 0], np.int64), query_ref=np.array([5, 7, 5, 9], np.int64),
         query_out=TSF.sphashquery(torch.tensor([5, 9, 11, 0]), torch.tensor([5, 7, 5, 9])).numpy(),
         count_idx=np.array([0, 2, 2, -1, 5, 2, 0], np.int32),
         count_out=TSF.spcount(torch.tensor([0, 2, 2, -1, 5, 2, 0], dtype=torch.int), 7).numpy(),
         unique_in=np.array([[1, 0, 0, 0], [0, 5, 0, 0], [0, 0, -1, 0], [0, 0, -1, 0]], np.int32),
         unique_out=torch.unique(torch.tensor([[1, 0, 0, 0], [0, 5, 0, 0], [0, 0, -1, 0], [0, 0, -1, 0]],
                                              dtype=torch.int), dim=0).numpy())

    # ---------------------------------------------------------------- G-koff
    save("g_koff.npz", {"what": "get_kernel_offsets(size,1,1) from torchsparse/nn/utils/kernel.py"},
         r2=off2.numpy(), r3=off3.numpy(), r4=get_kernel_offsets(4, 1, 1).numpy(),
         r5=get_kernel_offsets(5, 1, 1).numpy())

    # ---------------------------------------------------------------- G-index / G-agg
    for tag, n, grid, lo, batches, seed in (("a", 2000, 32, 0, 1, 3), ("neg", 1500, 24, -12, 1, 4),
                                            ("b2", 900, 20, 0, 2, 5)):
        coords = rand_coords(n, grid, seed, batches, lo)
        for s in (3, 7):
            for r in (2, 3):
                for W in ((8, 16) if tag == "a" else (8,)):
                    gen = torch.Generator().manual_seed(100 + W)
                    feats = torch.randn(coords.shape[0], W, generator=gen)
                    if batches > 1:
                        # per-frame run (each frame alone has batch column = const, so the CPU
                        # kernel_hash defect cannot bite), then merge in torch.unique order
                        outs = torch.empty_like(feats)
                        outs_c = torch.empty_like(feats)
                        for b in range(batches):
                            sel = coords[:, 3] == b
                            outs[sel] = ref_aggregate(feats[sel], coords[sel], s, r)[4]
                            if r != 2:
                                outs_c[sel] = ref_aggregate(feats[sel], coords[sel], s, r, compiled=True)[4]
                        st = SparseTensor(feats.clone(), coords.clone(), 1)
                        aux, idx, counts = ref_utils.voxel_to_aux(st, s)
                        aux_f, aux_c, out = aux.F, aux.C, outs
                        nbr = None
                    else:
                        aux_f, aux_c, idx, counts, out = ref_aggregate(feats, coords, s, r)
                        outs_c = ref_aggregate(feats, coords, s, r, compiled=True)[4] if r != 2 else None
                        offs = get_kernel_offsets(r, 1, 1)
                        nbr = TSF.sphashquery(TSF.sphash(aux_c, offs), TSF.sphash(aux_c)).t().contiguous()
                    arrays = dict(coords=coords.numpy(), feats=feats.numpy(), small_c=aux_c.numpy(),
                                  idx_query=idx.numpy(), counts=counts.numpy(), aux_f=aux_f.numpy(),
                                  out=out.numpy())
                    if nbr is not None:
                        arrays["nbr"] = nbr.numpy().astype(np.int32)
                    if r != 2:
                        # the SAME aux_to_voxel with spdevoxelize = the reference's compiled CPU op on 8-wide slices
                        arrays["out_refcpu"] = outs_c.numpy()
                    save(f"g_agg_{tag}_s{s}_r{r}_w{W}.npz",
                         {"what": "voxel_to_aux + aux_to_voxel from segmentation/core/models/utils.py:44-84",
                          "s": s, "r": r, "W": W,
                          "devoxelize": "reference CPU op" if r == 2 else
                          "out: torch restatement of devoxelize_cuda.cu:21-33 (CPU op hard-wires K=8); out_refcpu -- r=3 forward: "
                          "reference compiled ops (devoxelize_forward_cpu on four 8-wide slices of the map padded to 32 columns, summed)",
                          "multi_batch": batches > 1}, **arrays)

    # ---------------------------------------------------------------- G-block (+ G-grad)
    C = 8
    coords = rand_coords(1200, 24, 7)
    gen = torch.Generator().manual_seed(8)
    feats = torch.randn(coords.shape[0], C, generator=gen)
    for variant, cls in (("unet", RefELKBlockUNet), ("encoder", RefELKBlockEnc)):
        for baseop, groups, s, r in (("cos", 2, 3, 2), ("sin", 2, 3, 2), ("cos_x", 1, 3, 2),
                                     ("cos", 2, 7, 3), ("cos_x", 1, 2, 3)):
            torch.manual_seed(2)
            blk = cls(C, C, groups=groups, baseop=baseop).eval()
            with torch.no_grad():   # make LayerNorm affine / alpha non-trivial
                for p_name, p in blk.named_parameters():
                    if "norm" in p_name or "pre_mix.1" in p_name or p_name == "alpha":
                        p.add_(0.25 * torch.randn_like(p))
            tstride = 2 if variant == "encoder" else 1
            f_in = feats.clone().requires_grad_(True)
            st = SparseTensor(f_in, coords.clone(), tstride)

            # R_core capture: hook the output of self.norm; local_mix captured from its module
            cap = {}
            h1 = blk.norm.register_forward_hook(lambda m, i, o: cap.__setitem__("core", o))
            h2 = blk.local_mix.register_forward_hook(lambda m, i, o: cap.__setitem__("local", o.F))
            with torch.no_grad():
                st = SparseTensor(feats.clone(), coords.clone(), tstride)
                if r == 2:
                    out_st = blk(st, s, r)          # all-reference forward
                else:
                    with _PatchedDevox():
                        out_st = blk(st, s, r)
            assert out_st is st   # in-place contract (linkunet.py:158,183-185)
            core_fwd, local_fwd, out_fwd = cap["core"].clone(), cap["local"].clone(), out_st.F.clone()
            core_c = out_c = None
            if r != 2:            # round 6: the r = 3 forward again with spdevoxelize = the reference's compiled CPU op
                with torch.no_grad():
                    st_c = SparseTensor(feats.clone(), coords.clone(), tstride)
                    with _CompiledDevox():
                        out_c = blk(st_c, s, r).F.clone()
                    core_c = cap["core"].clone()
            # gradient pass: ALWAYS through the torch-restated devoxelize, because the reference CPU
            # devoxelize_backward_cpu is defective (devoxelize_cpu.cpp:43-55, SURVEY.md 8c defect 2)
            f_in = feats.clone().requires_grad_(True)
            st = SparseTensor(f_in, coords.clone(), tstride)
            with _PatchedDevox():
                blk(st, s, r)
            h1.remove(); h2.remove()
            core = cap["core"]
            assert torch.allclose(core, core_fwd, rtol=1e-5, atol=1e-6)
            gen2 = torch.Generator().manual_seed(9)
            gout = torch.randn(core.shape, generator=gen2)
            params = [p for _, p in blk.named_parameters() if "local_mix" not in _ and "norm_local" not in _]
            names = [n_ for n_, _ in blk.named_parameters() if "local_mix" not in n_ and "norm_local" not in n_]
            grads = torch.autograd.grad(core, [f_in] + params, gout, allow_unused=True)
            arrays = dict(coords=coords.numpy(), feats=feats.numpy(), core=core_fwd.numpy(),
                          local=local_fwd.numpy(), out=out_fwd.numpy(),
                          grad_out=gout.numpy(), grad_feats=grads[0].numpy())
            for n_, g_ in zip(names, grads[1:]):
                if g_ is not None:
                    arrays["grad__" + n_] = g_.numpy()
            if core_c is not None:
                arrays["core_refcpu"], arrays["out_refcpu"] = core_c.numpy(), out_c.numpy()
            for k, v in blk.state_dict().items():
                arrays["sd__" + k] = v.numpy()
            save(f"g_block_{variant}_{baseop}_s{s}_r{r}.npz",
                 {"what": f"ELKBlock.forward ({variant}) segmentation/core/models/semantic_kitti/"
                          f"link{'unet' if variant == 'unet' else 'encoder'}.py:124-185; core = output of "
                          "self.norm (R_core), local = local_mix(st).F, out = final st.F; grads of "
                          "sum(core*grad_out) by torch.autograd through the reference modules with "
                          "spdevoxelize restated in torch (reference CPU devoxelize backward is defective); "
                          "spvoxelize backward is the reference CPU op"
                          + ("; core_refcpu / out_refcpu -- r=3 forward: reference compiled ops (devoxelize_forward_cpu on four 8-wide "
                             "slices of the padded map, summed)" if r != 2 else ""),
                  "baseop": baseop, "groups": groups, "s": s, "r": r, "C": C,
                  "tensor_stride": tstride, "variant": variant}, **arrays)

    # ---------------------------------------------------------------- G-size (config checkpoints)
    import hashlib
    sizes = {}
    for n, cch, s in ((10_000, 16, 7), (100_000, 64, 7), (100_000, 64, 3)):
        g0 = torch.Generator().manual_seed(0)
        G = 256
        lin = torch.randperm(G ** 3, generator=g0)[:n]
        coords = torch.stack([lin % G, (lin // G) % G, lin // (G * G), torch.zeros_like(lin)], 1).int()
        st = SparseTensor(torch.zeros(n, 1), coords, 1)
        aux, idx, counts = ref_utils.voxel_to_aux(st, s)
        offs = get_kernel_offsets(3, 1, 1)
        nbr = TSF.sphashquery(TSF.sphash(aux.C, offs), TSF.sphash(aux.C)).t().contiguous()
        sizes[f"N{n}_s{s}"] = {
            "M": int(aux.C.shape[0]),
            "r3_miss_rate": float((nbr == -1).float().mean()),
            "sha256_idx": hashlib.sha256(idx.numpy().astype(np.int64).tobytes()).hexdigest(),
            "sha256_counts": hashlib.sha256(counts.numpy().astype(np.int32).tobytes()).hexdigest(),
            "sha256_small_c": hashlib.sha256(aux.C.numpy().astype(np.int32).tobytes()).hexdigest(),
            "sha256_nbr_r3": hashlib.sha256(nbr.numpy().astype(np.int32).tobytes()).hexdigest(),
        }
        print(n, s, sizes[f"N{n}_s{s}"]["M"], sizes[f"N{n}_s{s}"]["r3_miss_rate"])
    with open(os.path.join(HERE, "g_size.json"), "w") as f:
        json.dump({"meta": META_COMMON, "generator_spec": "SURVEY.md section 8d S-uniform: "
                   "randperm(256^3, seed 0)[:N]; x=lin%G, y=(lin//G)%G, z=lin//G^2, b=0", "sizes": sizes},
                  f, indent=1)


if __name__ == "__main__":
    main()
