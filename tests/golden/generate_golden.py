"""Generate the golden vectors under tests/golden/ from the REFERENCE's own Python.

Run ONLY in the build container (needs /root/reference, which never travels to the GPU box):

    python tests/golden/generate_golden.py

The reference package cannot be imported as-is (its __init__ loads a Poplar .so and both _impl
modules `import poptorch`), so - as SURVEY.md 8(c) records - a three-line `poptorch` stand-in is
registered first and the two _impl modules are loaded by path.  Nothing from the reference is copied:
this script only *calls* it and stores inputs/outputs as .npz data.

Vectors:
  G1  config-1: seed 1123581321 (t_fa.py:59), qkv=randn(3,1,128,64) fp32, (nq,nk)=(4,4) -> out
  G2  same qkv cast to fp16 / bf16 -> out of the reference CPU path in that dtype
  G3  C++-test-shaped G=4,L=6,D=2 (t_cpp:245-251), all 7 chunk pairs -> out (identical by construction)
  G4  SDPA: q,k,v=frozen_randn(seed+4,(3,2,3,256,64)) -> sampled rows + checksums of passthrough out and fake-custom_op IPU-branch out + the
      (nq,nk) the heuristic chose, incl. the heuristic table of 8(c)
  G5  larger shapes from the frozen-stream generator -> sampled rows + fp64 checksums
  G6  error-message strings of the 4 ValueErrors and 5 NotImplementedErrors
  G7  head dims other than 64 / 128 (the op has no head-dim restriction, FA.cpp:175-195): G=2, L=512, D in {32, 96, 192, 256}, frozen-stream inputs
      scaled by 0.5, fp32 / bf16 / f16 -> sampled rows + fp64 checksums          (`python tests/golden/generate_golden.py g7` writes this file only)
"""

import importlib.util
import json
import pathlib
import sys
import types

import numpy as np
import torch

REF = pathlib.Path("/root/reference/flash_attention_ipu")
HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))

from oracle.dense_attention import frozen_randn  # noqa: E402

SEED = 1123581321


def load_reference():
    poptorch = types.ModuleType("poptorch")
    poptorch._on_ipu = False
    poptorch.isRunningOnIpu = lambda: poptorch._on_ipu
    sys.modules["poptorch"] = poptorch
    pkg = types.ModuleType("flash_attention_ipu")
    pkg.__path__ = [str(REF)]
    sys.modules["flash_attention_ipu"] = pkg
    sub = types.ModuleType("flash_attention_ipu._impl")
    sub.__path__ = [str(REF / "_impl")]
    sys.modules["flash_attention_ipu._impl"] = sub
    mods = {}
    for name in ("utils", "flash_attention_qkv_packed", "scaled_dot_product_attention"):
        full = f"flash_attention_ipu._impl.{name}"
        spec = importlib.util.spec_from_file_location(full, REF / "_impl" / f"{name}.py")
        m = importlib.util.module_from_spec(spec)
        sys.modules[full] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return poptorch, mods


G7_DIMS = (32, 96, 192, 256)
G7_SHAPE = (2, 512)      # G, L


def generate_g7(fa):
    g7 = {}
    G, L = G7_SHAPE
    rows = sorted({0, 1, 31, 32, 63, 64, 255, 256, L - 1})
    for D in G7_DIMS:
        x = frozen_randn(SEED + 7 + D, (3, G, L, D)) * 0.5
        for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16), ("f16", torch.float16)):
            o = fa(x.to(dt), 2, 2).double()
            g7[f"D{D}_{dt_name}_rows"] = o[:, rows, :].numpy()
            g7[f"D{D}_{dt_name}_sum"] = np.array(o.sum().item())
            g7[f"D{D}_{dt_name}_abssum"] = np.array(o.abs().sum().item())
    g7["rowidx"] = np.array(rows)
    g7["dims"] = np.array(G7_DIMS)
    g7["shape"] = np.array(G7_SHAPE)
    np.savez_compressed(HERE / "g7_head_dims.npz", **g7)


def main():
    orig_sdpa = torch.nn.functional.scaled_dot_product_attention
    poptorch, mods = load_reference()
    fa = mods["flash_attention_qkv_packed"].flash_attention_qkv_packed
    if sys.argv[1:] == ["g7"]:
        generate_g7(fa)
        torch.nn.functional.scaled_dot_product_attention = orig_sdpa
        print("wrote g7_head_dims.npz")
        return
    patched = torch.nn.functional.scaled_dot_product_attention
    assert patched.__wrapped__ is orig_sdpa

    # ---- G1 / G2
    torch.manual_seed(SEED)
    qkv = torch.randn(3, 1, 128, 64)
    g1 = fa(qkv, 4, 4)
    g2h = fa(qkv.half(), 4, 4)
    g2b = fa(qkv.bfloat16(), 4, 4)
    np.savez_compressed(HERE / "g1_g2_config1.npz", qkv=qkv.numpy(), out_f32=g1.numpy(),
                        out_f16=g2h.float().numpy(), out_bf16=g2b.float().numpy())

    # ---- G3
    torch.manual_seed(SEED)
    qkv3 = torch.randn(3, 4, 6, 2)
    pairs = [(1, 1), (1, 2), (2, 1), (2, 2), (2, 3), (3, 2), (3, 3)]
    outs = np.stack([fa(qkv3, a, b).numpy() for a, b in pairs])
    np.savez_compressed(HERE / "g3_tiny_chunks.npz", qkv=qkv3.numpy(), pairs=np.array(pairs), outs=outs)

    # ---- G4: SDPA passthrough + emulated IPU branch
    q, k, v = frozen_randn(SEED + 4, (3, 2, 3, 256, 64))  # inputs regenerated, not stored
    out_pass = torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True)
    calls = []

    def fake_custom_op(name, domain_version, domain, inputs, example_outputs, attributes):
        calls.append(dict(name=name, domain=domain, domain_version=domain_version,
                          shape=list(inputs[0].shape), attributes=dict(attributes)))
        poptorch._on_ipu = False
        try:
            return (fa(inputs[0], attributes["num_chunks_q"], attributes["num_chunks_kv"]),)
        finally:
            poptorch._on_ipu = True

    poptorch.custom_op = fake_custom_op
    poptorch._on_ipu = True
    out_ipu = torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True)
    table = {}
    for (L, D) in [(128, 64), (256, 64), (1024, 64), (2048, 64), (4096, 64), (8192, 64),
                   (32768, 128), (65536, 64), (16384, 128), (64, 64), (32, 64)]:
        calls.clear()
        x = torch.zeros(1, L, D)
        # only the heuristic matters: intercept before compute
        poptorch.custom_op = lambda **kw: (calls.append(kw["attributes"]) or (kw["inputs"][0][0],))
        torch.nn.functional.scaled_dot_product_attention(x, x, x, is_causal=True)
        table[f"{L},{D}"] = [int(calls[0]["num_chunks_q"]), int(calls[0]["num_chunks_kv"])]
    poptorch._on_ipu = False
    rows4 = [0, 1, 2, 31, 32, 63, 64, 127, 128, 255]
    np.savez_compressed(HERE / "g4_sdpa.npz", rowidx=np.array(rows4),
                        out_passthrough_rows=out_pass[..., rows4, :].numpy(),
                        out_ipu_branch_rows=out_ipu[..., rows4, :].numpy(),
                        out_passthrough_sum=np.array(out_pass.double().sum().item()),
                        out_ipu_branch_sum=np.array(out_ipu.double().sum().item()),
                        out_ipu_branch_abssum=np.array(out_ipu.double().abs().sum().item()))

    # ---- G5: larger shapes, frozen-stream inputs, sampled rows + checksums
    g5 = {}
    for tag, (G, L, D) in {"N4_L1024_D128": (4, 1024, 128), "G2_L4096_D64": (2, 4096, 64)}.items():
        x = frozen_randn(SEED, (3, G, L, D))
        rows = sorted({0, 1, 63, 64, 511, 512, L - 1})
        for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16), ("f16", torch.float16)):
            o = fa(x.to(dt), 1, 1).double()
            g5[f"{tag}_{dt_name}_rows"] = o[:, rows, :].numpy()
            g5[f"{tag}_{dt_name}_sum"] = np.array(o.sum().item())
            g5[f"{tag}_{dt_name}_abssum"] = np.array(o.abs().sum().item())
        g5[f"{tag}_rowidx"] = np.array(rows)
        g5[f"{tag}_shape"] = np.array([G, L, D])
    np.savez_compressed(HERE / "g5_large_sampled.npz", **g5)

    # ---- G7: head dims other than 64 / 128
    generate_g7(fa)

    # ---- G6: error strings
    errs = {"ValueError": [], "NotImplementedError": []}
    for bad, a, b in [(torch.zeros(3, 4, 4), 1, 1), (torch.zeros(2, 1, 4, 4), 1, 1),
                      (torch.zeros(3, 1, 6, 4), 4, 1), (torch.zeros(3, 1, 6, 4), 1, 4)]:
        try:
            fa(bad, a, b)
        except ValueError as e:
            errs["ValueError"].append(str(e))
    x = torch.zeros(1, 8, 4)
    for kwargs in [dict(attn_mask=torch.ones(1, dtype=torch.bool), is_causal=True), dict(dropout_p=0.1, is_causal=True),
                   dict(is_causal=False)]:
        try:
            torch.nn.functional.scaled_dot_product_attention(x, x, x, **kwargs)
        except NotImplementedError as e:
            errs["NotImplementedError"].append(str(e))
    for qq, kk, vv in [(x, torch.zeros(1, 4, 4), x), (x, x, torch.zeros(1, 8, 2))]:
        try:
            torch.nn.functional.scaled_dot_product_attention(qq, kk, vv, is_causal=True)
        except NotImplementedError as e:
            errs["NotImplementedError"].append(str(e))
    meta = dict(seed=SEED, torch=torch.__version__, heuristic=table, sdpa_ipu_call=dict(
        name="FlashAttentionQKVPacked", domain="ai.graphcore", domain_version=1, input_shape=[3, 6, 256, 64],
        attributes=table["256,64"]), errors=errs)
    (HERE / "g6_meta.json").write_text(json.dumps(meta, indent=1))
    torch.nn.functional.scaled_dot_product_attention = orig_sdpa
    print(json.dumps(meta, indent=1))
    print("max|ipu-branch - passthrough| =", (out_ipu - out_pass).abs().max().item())


if __name__ == "__main__":
    main()
