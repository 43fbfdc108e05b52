"""Timing of the fp16 VAE encode + decode (the non-DDIM part of a sampling pass) per batch size; output checksum for A/B of library variants."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
dev = torch.device("cuda:0")
vae, den, sch = bench.build_models(dev, "fp16", "fp16")
for B in [int(x) for x in (sys.argv[1:] or ["32", "256", "512"])]:
    g = torch.Generator(device="cpu").manual_seed(1234)
    motion = torch.randn(B, 196, 132, generator=g).to(dev)
    z = torch.randn(1, B, 256, generator=g).to(dev)
    lengths = [196] * B
    for _ in range(3):
        d = vae.encode_dist(motion, lengths); f = vae.decode(z, lengths)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); d = vae.encode_dist(motion, lengths); f = vae.decode(z, lengths); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(json.dumps({"B": B, "lib": os.path.basename(os.environ.get("SEEME_HIP_LIB", "")), "encode_decode_ms": round(float(np.median(ts)), 4),
                      "sum_dist": float(d.double().sum()), "sum_feats": float(f.double().sum())}), flush=True)
