"""Developer aid (test infrastructure): differential fuzz of the forward kernels.  Random shapes, dtypes, scales,
layouts, mask modes (causal / none) and key-value head counts (grouped-query); every MFMA kernel variant (dma / pipe
x 1 / 2 query blocks per wave, 1..8 waves) against the register-staged baseline kernel and, for short sequences,
against the fp64 oracle.  Variants are forced through fa_debug_set_option (_native.debug_options).

    python tests/tools/fuzz_kernels.py [cases] [seed]
"""
import ctypes, os, pathlib, random, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch
from flash_attention_amd._impl import _native
from oracle import dense_attention_lse

lib = _native.load_native_library()


def run(q, k, v, heads, scale, knobs, causal=True):
    B, H, L, D = q.shape
    out = torch.empty(B, H, L, D, device="cuda", dtype=q.dtype)
    lse = torch.empty(B * H, L, device="cuda", dtype=torch.float32)
    def st(t):
        return (ctypes.c_int64 * 3)(t.stride(0), t.stride(1), t.stride(2))
    with _native.debug_options(**knobs):
        rc = lib.fa_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), B * H, L, D, H, k.shape[1],
                        st(q), st(k), st(v), st(out), 1, 1, _native.FA_DTYPE[q.dtype], ctypes.c_float(scale), int(causal),
                        torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.fa_last_error().decode()
    torch.cuda.synchronize()
    return out, lse


def fuzz(cases, seed, log=print):
  rng = random.Random(seed)
  bad = 0
  only = os.environ.get("FUZZ_ONLY")
  for case in range(cases):
      D = rng.choice([64, 64, 128])
      L = rng.choice([rng.randint(1, 300), rng.randint(1, 1500), rng.randint(1500, 5000)])
      B, H = rng.randint(1, 3), rng.randint(1, 5)
      dt = rng.choice([torch.bfloat16, torch.float16])
      scale = rng.choice([1.0, D ** -0.5])
      amp = rng.choice([0.3, 1.0]) * float(os.environ.get("FUZZ_AMP", "1"))   # FUZZ_AMP=5: logits far past the skip-max range
      causal = rng.random() < 0.7
      Hkv = rng.choice([h for h in range(1, H + 1) if H % h == 0]) if rng.random() < 0.3 else H
      torch.manual_seed(case)
      if rng.random() < 0.5 and Hkv == H:      # head-inner projection layout [B, L, 3, H, D] -> strided views
          x = (torch.randn(B, L, 3, H, D, device="cuda") * amp).to(dt)
          q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
      else:
          q = (torch.randn(B, H, L, D, device="cuda") * amp).to(dt)
          k, v = ((torch.randn(B, Hkv, L, D, device="cuda") * amp).to(dt) for _ in range(2))
      variants = [{}]
      for kern in ("dma", "pipe"):
          for qb in (1, 2):
              variants.append({"kernel": kern, "qb": qb, "waves": rng.choice([1, 2, 4, 4, 8]), "sched": rng.choice([1, 2, 8])})
      if causal:        # q-tile pairs per workgroup, and the placement-balanced block order (applies to grids below one round of slots)
          variants.append({"kernel": "dma", "waves": 4, "pair": 1, "sched": rng.choice([1, 2, 8])})
          variants.append({"kernel": "dma", "waves": 4, "pair": 0, "balance": rng.choice([0, 1])})
      if D == 128:      # the pipelined kernel on 16x16x32 MFMAs: 4- and 8-wave workgroups at head_dim 128,
          variants.append({"kernel": "pipe16", "waves": rng.choice([4, 8]), "sched": rng.choice([1, 2, 8])})
      else:             # 32 or 64 query rows per wave at head_dim 64
          variants.append({"kernel": "pipe16", "waves": 4, "qb": rng.choice([1, 2]), "sched": rng.choice([1, 2, 8])})
      variants.append({"split": 2 if D == 128 else rng.choice([2, 4]), "sched": rng.choice([1, 2, 8])})   # kv split groups (dma kernel)
      variants.append({"kernel": "kh", "sched": rng.choice([1, 2, 8])})      # 64-row workgroups, waves = (row block, key half) ...
      variants.append({"kernel": "kh", "split": 2 if D == 128 else rng.choice([2, 4]), "balance": rng.choice([0, 1])})   # ... with kv split groups
      if only is not None and case != int(only):
          continue
      # the baseline of the cheap cross-check: the plain 64-key-tile LDS-DMA kernel with 4 waves x one block (the oracle arbitrates every
      # disagreement either way)
      base, base_lse = run(q, k, v, H, scale, {"kernel": "dma", "waves": 4, "qb": 1, "sched": 1}, causal)
      ulp = 2.0 ** -6 if dt == torch.bfloat16 else 2.0 ** -9
      ref = None
      if L <= 700:
          ref, ref_lse = dense_attention_lse(q.cpu(), k.cpu(), v.cpu(), scale=scale, causal=causal)
          ref = ref.reshape(B * H, L, D)
      for env in variants:
          out, lse = run(q, k, v, H, scale, env, causal)
          d = (out.float() - base.float()).abs()
          # output error scales with |V| (P carries 8 / 11 bits); with FUZZ_AMP > 1 the softmax is nearly one-hot and
          # 16-bit P underflow shows as a few more ulps in every kernel: twice the slack there
          vs = max(1.0, amp) * (2.0 if amp > 1.0 else 1.0)
          # (the MFMA kernels normalise by the sum of the 16-bit-rounded P - the weights the PV product uses -, the
          # register-staged baseline by the exact sum, and P is rounded relative to different reference maxima: two
          # correct kernels sit up to 2^-8 (bf16) of |V| apart per dominant key, each within the oracle tolerance)
          tol = vs * 1e-2 * (1 if dt == torch.bfloat16 else 0.1) + ulp * (2.0 if amp > 1.0 else 1.0) * base.float().abs()
          finite = bool(torch.isfinite(out).all())
          ok = finite and bool((d <= tol).all()) and bool(((lse - base_lse).abs() <= (6e-3 if dt == torch.bfloat16 else 2e-3) + 1e-4 * base_lse.abs()).all())   # (bf16: |dLSE| <= 2^-8, see above)
          if not ok and ref is None and finite:
              # two correct kernels may sit further apart than this cheap cross-check allows (near-one-hot rows: P is rounded
              # relative to different reference maxima): then the oracle decides, whatever the sequence length
              ref, ref_lse = dense_attention_lse(q.cpu(), k.cpu(), v.cpu(), scale=scale, causal=causal)
              ref = ref.reshape(B * H, L, D)
              eb = (base.reshape(B * H, L, D).cpu().double() - ref).abs()
              log(f"case {case}: a variant differs from the baseline kernel by {d.max().item():.3e}; the baseline sits {eb.max().item():.3e} from the oracle")
          if ref is not None and finite:      # the oracle is the authority wherever it was computed
              e = (out.reshape(B * H, L, D).cpu().double() - ref).abs()
              el = (lse.cpu().double() - ref_lse.reshape(B * H, L)).abs()
              ok = bool((e <= max(1.0, amp) * 1e-2 + 1e-2 * ref.abs()).all()) and bool((el <= (6e-3 if dt == torch.bfloat16 else 2e-3) + 1e-4 * ref_lse.reshape(B * H, L).abs()).all())
          if not ok:
              bad += 1
              excess = (d - tol).max().item()
              dl = (lse - base_lse).abs()
              log(f"MISMATCH case {case}: B={B} H={H} Hkv={Hkv} L={L} D={D} {dt} scale={scale:.3g} amp={amp} causal={causal} knobs={env} max|d|={d.max().item():.3e} "
                  f"max(d - tol)={excess:.3e} max|dLSE|={dl.max().item():.3e} (base LSE there {base_lse.flatten()[dl.argmax()].item():.3f}) finite={bool(torch.isfinite(out).all())}")
      if case % 20 == 0:
          log(f"case {case}: B={B} H={H} L={L} D={D} {dt} ok so far, bad={bad}")
  # head dims above 128 (native at 192 / 256, the multiples of 8 between zero-extended): the 64-key-tile kernel with 4 and with 8 waves against each
  # other and, for short sequences, the fp64 oracle
  for case in range(max(2, cases // 3)):
      D = rng.choice([136, 160, 192, 192, 200, 224, 256, 256])
      L = rng.choice([rng.randint(1, 300), rng.randint(1, 1500), rng.randint(1500, 3000)])
      B, H = rng.randint(1, 2), rng.randint(1, 4)
      dt = rng.choice([torch.bfloat16, torch.float16])
      scale = rng.choice([0.5, D ** -0.5])
      amp = rng.choice([0.3, 1.0]) * (0.25 if scale == 0.5 else 1.0)
      causal = rng.random() < 0.7
      Hkv = rng.choice([h for h in range(1, H + 1) if H % h == 0]) if rng.random() < 0.3 else H
      torch.manual_seed(20_000 + case)
      if rng.random() < 0.5 and Hkv == H:
          x = (torch.randn(B, L, 3, H, D, device="cuda") * amp).to(dt)
          q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
      else:
          q = (torch.randn(B, H, L, D, device="cuda") * amp).to(dt)
          k, v = ((torch.randn(B, Hkv, L, D, device="cuda") * amp).to(dt) for _ in range(2))
      base, base_lse = run(q, k, v, H, scale, {"waves": 4, "sched": 1}, causal)
      ref = None
      if L <= 700:
          ref, ref_lse = dense_attention_lse(q.cpu(), k.cpu(), v.cpu(), scale=scale, causal=causal)
      for env in ({}, {"waves": 8, "sched": rng.choice([1, 2, 8])}, {"waves": 4, "sched": rng.choice([2, 8])}):
          out, lse = run(q, k, v, H, scale, env, causal)
          ok = bool(torch.isfinite(out).all()) and bool(((out.float() - base.float()).abs() <= 1e-2 * (1 if dt == torch.bfloat16 else 0.1) + 2.0 ** -6 * base.float().abs()).all()) \
              and bool(((lse - base_lse).abs() <= 6e-3 + 1e-4 * base_lse.abs()).all())
          if ref is not None:
              e = (out.cpu().double() - ref).abs()
              el = (lse.cpu().double() - ref_lse.reshape(B * H, L)).abs()
              ok = ok and bool((e <= 1e-2 + 1e-2 * ref.abs()).all()) and bool((el <= (6e-3 if dt == torch.bfloat16 else 2e-3) + 1e-4 * ref_lse.reshape(B * H, L).abs()).all())
          if not ok:
              bad += 1
              log(f"MISMATCH wide case {case}: B={B} H={H} Hkv={Hkv} L={L} D={D} {dt} scale={scale:.3g} causal={causal} knobs={env} max|d|={(out.float() - base.float()).abs().max().item():.3e}")
      if case % 20 == 0:
          log(f"wide case {case}: B={B} H={H} L={L} D={D} {dt} ok so far, bad={bad}")
  # fp32 inputs: the fp32 MFMA kernel against the generic kernel (FA_AMD_F32=generic) and the oracle
  for case in range(max(1, cases // 5)):
      D = rng.choice([64, 128])
      L = rng.choice([rng.randint(1, 300), rng.randint(1, 1200)])
      B, H = rng.randint(1, 2), rng.randint(1, 4)
      scale = rng.choice([1.0, D ** -0.5])
      torch.manual_seed(10_000 + case)
      if rng.random() < 0.5:
          x = torch.randn(B, L, 3, H, D, device="cuda")
          q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
      else:
          q, k, v = (torch.randn(B, H, L, D, device="cuda") for _ in range(3))
      causal = rng.random() < 0.7
      base, base_lse = run(q, k, v, H, scale, {"f32_generic": 1}, causal)
      out, lse = run(q, k, v, H, scale, {}, causal)
      ok = bool(torch.isfinite(out).all()) and (out - base).abs().max().item() <= 1e-4 and (lse - base_lse).abs().max().item() <= 1e-4
      if L <= 500:
          ref, _ = dense_attention_lse(q.cpu(), k.cpu(), v.cpu(), scale=scale, causal=causal)
          ok = ok and (out.cpu().double() - ref).abs().max().item() <= 1e-4
      if not ok:
          bad += 1
          log(f"MISMATCH fp32 case {case}: B={B} H={H} L={L} D={D} scale={scale:.3g} max|d|={(out - base).abs().max().item():.3e}")
  # backward: random shapes / dtypes / mask modes / scales through autograd of the override's training path against
  # autograd through the fp64 oracle (MFMA kernels for 16-bit and fp32 at head_dim 64 / 128, generic kernels otherwise)
  from flash_attention_amd._impl import ops
  for case in range(max(2, cases // 6)):
      D = rng.choice([64, 64, 128, 128, 24, 80, 160, 192, 256])
      L = rng.choice([rng.randint(1, 130), rng.randint(1, 420)])
      G = rng.randint(1, 6)
      dt = rng.choice([torch.bfloat16, torch.float16, torch.float32])
      scale = rng.choice([1.0, D ** -0.5])
      causal = rng.random() < 0.7
      torch.manual_seed(20_000 + case)
      x = (torch.randn(3, G, L, D) * 0.6).to(dt)
      g = torch.randn(G, L, D).to(dt)
      xr = x.double().requires_grad_(True)
      ref, _ = dense_attention_lse(xr[0][None], xr[1][None], xr[2][None], scale=scale, causal=causal)
      ref[0].backward(g.double())
      xc = x.cuda().requires_grad_(True)
      out = ops.attention(xc[0], xc[1], xc[2], scale, 1, 1, causal)
      out.backward(g.cuda())
      torch.cuda.synchronize()
      tol = {torch.float16: 4e-3, torch.bfloat16: 1.5e-2, torch.float32: 1e-5}[dt]
      gr, want = xc.grad.cpu().double(), xr.grad
      floor = tol * max(1e-3, want.abs().max().item())
      ok = bool(torch.isfinite(gr).all()) and bool(((gr - want).abs() <= floor + 2e-2 * want.abs()).all())
      if not ok:
          bad += 1
          log(f"MISMATCH backward case {case}: G={G} L={L} D={D} {dt} scale={scale:.3g} causal={causal} max|d|={(gr - want).abs().max().item():.3e} (ref max {want.abs().max().item():.2f})")
  # backward with grouped-query K/V heads (dK / dV of a shared head summed inside the kernels) against oracle autograd
  saved_gqa_threshold = ops.GQA_NATIVE_MIN_WORKGROUPS
  ops.GQA_NATIVE_MIN_WORKGROUPS = 0          # (small shapes: force the kernels' own head mapping)   restored below: later tests of the same process must see the product's route selection
  try:
    for case in range(max(2, cases // 8)):
        D = rng.choice([64, 128, rng.choice([8, 24, 40, 96])])
        L = rng.choice([rng.randint(1, 130), rng.randint(1, 420)])
        B, Hkv, rep = rng.randint(1, 2), rng.randint(1, 3), rng.choice([2, 3, 4])
        H = Hkv * rep
        dt = rng.choice([torch.bfloat16, torch.float16, torch.float32])
        scale = rng.choice([1.0, D ** -0.5])
        causal = rng.random() < 0.7
        torch.manual_seed(30_000 + case)
        q0 = (torch.randn(B, H, L, D) * 0.6).to(dt)
        k0, v0 = ((torch.randn(B, Hkv, L, D) * 0.6).to(dt) for _ in range(2))
        g = torch.randn(B, H, L, D).to(dt)
        qr, kr, vr = (t.double().requires_grad_(True) for t in (q0, k0, v0))
        ref, _ = dense_attention_lse(qr, kr, vr, scale=scale, causal=causal)
        ref.backward(g.double())
        qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q0, k0, v0))
        ops.attention_gqa(qc, kc, vc, scale, 1, 1, causal).backward(g.cuda())
        torch.cuda.synchronize()
        tol = {torch.float16: 4e-3, torch.bfloat16: 1.5e-2, torch.float32: 1e-5}[dt]
        for name, got, want in (("dq", qc.grad, qr.grad), ("dk", kc.grad, kr.grad), ("dv", vc.grad, vr.grad)):
            got = got.cpu().double()
            # (absolute floor: a gradient that is exactly zero - L = 1: P = 1, dS = dP - Dsum = 0 - comes out as the rounding noise of the two
            #  equal sums it is the difference of, so the floor scales with them and with the operand dS multiplies)
            cancel = (g.double() * ref.detach()).sum(-1).abs().max().item() * max(q0.abs().max().item(), k0.abs().max().item()) * scale
            floor = tol * max(1e-1, want.abs().max().item(), cancel)
            if not (bool(torch.isfinite(got).all()) and bool(((got - want).abs() <= floor + 2e-2 * want.abs()).all())):
                bad += 1
                log(f"MISMATCH grouped-query backward case {case} {name}: B={B} H={H}/{Hkv} L={L} D={D} {dt} scale={scale:.3g} causal={causal} max|d|={(got - want).abs().max().item():.3e}")
  finally:
      ops.GQA_NATIVE_MIN_WORKGROUPS = saved_gqa_threshold
  # attention masks / biases and dropout (fa_fwd_ex / fa_bwd_ex) against the dense fp64 formula with the numpy restatement of the generator
  from flash_attention_amd._impl import ops
  from oracle import dense_attention_extra
  for case in range(max(2, cases // 4)):
      D = rng.choice([64, 128])
      L = rng.choice([rng.randint(1, 140), rng.randint(1, 420)])
      B, Hkv, rep = rng.randint(1, 2), rng.randint(1, 2), rng.choice([1, 1, 2, 3])
      H = Hkv * rep
      dt = rng.choice([torch.bfloat16, torch.float16])
      scale = rng.choice([D ** -0.5, 0.5 * D ** -0.5])
      causal = rng.random() < 0.4
      p = rng.choice([0.0, 0.0, 0.15, 0.5])
      kind = rng.choice(["none", "bool_full", "bool_pad", "bias_full", "bias_head", "bias_row"])
      gen = torch.Generator().manual_seed(40_000 + case)
      if kind == "none":
          mask = None
      elif kind == "bool_full":
          mask = (torch.rand(B, H, L, L, generator=gen) < 0.6) | torch.eye(L, dtype=torch.bool)
      elif kind == "bool_pad":
          mask = (torch.arange(L)[None, :] < torch.randint(1, L + 1, (B,), generator=gen)[:, None]).view(B, 1, 1, L)
      elif kind == "bias_full":
          mask = torch.randn(B, H, L, L, generator=gen).to(dt)
      elif kind == "bias_head":
          mask = torch.randn(1, H, L, L, generator=gen).to(dt)
      else:
          mask = torch.randn(L, generator=gen).to(dt)
      if mask is None and p == 0.0:
          p = 0.25
      seed_, off_ = rng.getrandbits(64), rng.getrandbits(40)
      torch.manual_seed(50_000 + case)
      q0 = (torch.randn(B, H, L, D) * 0.6).to(dt)
      k0, v0 = ((torch.randn(B, Hkv, L, D) * 0.6).to(dt) for _ in range(2))
      g = torch.randn(B, H, L, D).to(dt)
      qr, kr, vr = (t.double().requires_grad_(True) for t in (q0, k0, v0))
      ref, _ = dense_attention_extra(qr, kr, vr, scale, causal, mask, p, seed_, off_)
      ref.backward(g.double())
      qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q0, k0, v0))
      mc = None
      if mask is not None:
          mc = mask.cuda().expand(B, H, L, L)
          mc = mc.view(torch.uint8) if mc.dtype == torch.bool else mc
          mc = mc if mc.stride(-1) == 1 or L == 1 else mc.contiguous()
      out = ops.attention_extra(qc, kc, vc, mc, scale, causal, p, seed_, off_)
      out.backward(g.cuda())
      torch.cuda.synchronize()
      what = f"extra case {case}: B={B} H={H}/{Hkv} L={L} D={D} {dt} mask={kind} p={p} causal={causal}"
      o = out.detach().cpu().double()
      if not (bool(torch.isfinite(o).all()) and bool(((o - ref.detach()).abs() <= 1e-2 + 1e-2 * ref.detach().abs()).all())):
          bad += 1
          log(f"MISMATCH forward {what}: max|d|={(o - ref.detach()).abs().max().item():.3e}")
      tol = {torch.float16: 4e-3, torch.bfloat16: 1.5e-2}[dt]
      for name, got, want in (("dq", qc.grad, qr.grad), ("dk", kc.grad, kr.grad), ("dv", vc.grad, vr.grad)):
          got = got.cpu().double()
          # (a gradient that is exactly 0 - L = 1: dS = P (dP - Dsum) = 0 - comes out as the rounding of Dsum = rowsum(dO o O) over the 16-bit O:
          #  ~1e-3 relative to |dO| |O| ~ 1, hence the absolute floor)
          floor = tol * max(1e-2, want.abs().max().item())
          if not (bool(torch.isfinite(got).all()) and bool(((got - want).abs() <= floor + 2e-2 * want.abs()).all())):
              bad += 1
              log(f"MISMATCH {name} {what}: max|d|={(got - want).abs().max().item():.3e} (ref max {want.abs().max().item():.2f})")
  return bad, len(variants)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    bad, nv = fuzz(n, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"{n} cases x {nv} variants: {bad} mismatches")
    sys.exit(1 if bad else 0)
