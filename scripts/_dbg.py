import sys, pathlib, json
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd"), str(ROOT / "demo")]
import torch
import bench
dev = torch.device("cuda")
r = bench.secondary_gpt2_forward(dev)
print("alone:", r["value"], r["attention_ms_per_step"], r["attention_tflops"])
r0 = bench.secondary_attention("cfg4_attn", dev)
print("cfg4_attn:", r0["value"])
r = bench.secondary_gpt2_forward(dev)
print("after cfg4_attn:", r["value"], r["attention_ms_per_step"], r["attention_tflops"])
r0 = bench.secondary_attention("cfg2", dev, steps=10, warmup=3, backward=True)
print("bwd:", r0["value"])
r = bench.secondary_gpt2_forward(dev)
print("after bwd:", r["value"], r["attention_ms_per_step"], r["attention_tflops"])
