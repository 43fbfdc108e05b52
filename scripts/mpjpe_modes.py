"""Decomposition of the MPJPE difference between the 16-bit throughput modes and the fp32 parity path
(north_star: "MPJPE within 1e-3 mm of the reference"): denoiser weight image {fp32, fp16, bf16} x VAE encode operands
{fp32, fp16} x VAE decode operands {fp32, fp16}, same inputs, same initial latents, same condition noise.
config_mld_egobody (interactee-only), B=32, T=196, nfeats 75, synthetic SMPL; the fp32 combination is the path
tests/test_gpu_parity.py::test_mld_sample_vs_oracle_mpjpe pins to the oracle (7.7e-6 mm).  Prints one JSON line per mode."""
import json, os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seeme_amd.config import parse_config
from seeme_amd.mld import MLD, SyntheticEgoDataModule, EgoMetrics
from seeme_amd.smpl import SMPL
from seeme_amd.weights_recipe import load_recipe_

dev = torch.device("cuda", 0)
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, T = int(os.environ.get("B", 32)), int(os.environ.get("T", 196))
cfg = parse_config(os.path.join(repo, "configs", "config_mld_egobody.yaml"))
dm = SyntheticEgoDataModule(nfeats=75, T=T, device=dev)
model = MLD(cfg, dm, smpl_model=SMPL.synthetic(1234))
load_recipe_(model.vae), load_recipe_(model.denoiser)
model = model.to(dev).eval()
batch = dm.batch(B, idx=1)
g = torch.Generator().manual_seed(5)
lat = torch.randn(B, 1, 256, generator=g).to(dev)
eps = torch.randn(1, B, 256, generator=g).to(dev)
feats_ref, transl, beta = batch[0].float(), batch[1].float(), batch[2].float()
lengths = [T] * B
f_int = model._wearer_features(feats_ref, transl, 1)
f_gt = model.renorm(model._wearer_features(feats_ref, transl, 0))
j_gt = model._feats_to_joints(f_gt, beta[:, 0])


def run(den, enc, dec, reps=0):
    model.denoiser.weight_dtype = den
    model.vae.precision = enc
    z_c, _ = model._sample_latent(f_int, lengths, eps)
    z = model._diffusion_reverse(z_c.permute(1, 0, 2).contiguous(), lengths, latents=lat)
    model.vae.precision = dec
    feats = model.renorm(model.vae.decode(z, lengths))
    j = model._feats_to_joints(feats, beta[:, 0])
    m = EgoMetrics.per_sequence(j, j_gt, lengths)["MPJPE"].double().mean().item()
    ms = None
    if reps:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            model.vae.precision = enc
            z_c, _ = model._sample_latent(f_int, lengths, eps)
            z = model._diffusion_reverse(z_c.permute(1, 0, 2).contiguous(), lengths, latents=lat)
            model.vae.precision = dec
            model.vae.decode(z, lengths)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
    return z, feats, j, m, ms


with torch.no_grad():
    z0, f0, j0, m0, ms0 = run("fp32", "fp32", "fp32", reps=10)
    print(json.dumps({"mode": "fp32/fp32/fp32", "mpjpe_mm": m0, "ms_per_pass": round(ms0, 3)}))
    for den, enc, dec in itertools.product(("fp32", "fp16", "bf16"), ("fp32", "fp16"), ("fp32", "fp16")):
        if (den, enc, dec) == ("fp32", "fp32", "fp32"):
            continue
        run(den, enc, dec)                      # warm (weight images)
        z, f, j, m, ms = run(den, enc, dec, reps=10)
        print(json.dumps({"mode": f"den {den} / enc {enc} / dec {dec}", "mpjpe_mm": m, "mpjpe_delta_mm": abs(m - m0),
                          "joint_to_joint_mm": float((j - j0).norm(dim=-1).mean() * 1000),
                          "latent_rel_err": float((z - z0).abs().max() / z0.abs().max()),
                          "feats_rel_err": float((f - f0).abs().max() / f0.abs().max()), "ms_per_pass": round(ms, 3)}), flush=True)
