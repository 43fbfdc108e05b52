"""GPU: one fp16-mode VAE encode / decode and one 5-step DDIM sample on fixed inputs, saved to argv[1].  Run under different
SEEME_DEN_XCDS / SEEME_LAYER_ROWS settings by tests/test_gpu_flows.py::test_launch_mapping_knobs_do_not_change_results:
the knobs move workgroups between XCDs / change the rows per workgroup and must not change a single bit."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seeme_amd.mld_vae import MldVae
from seeme_amd.mld_denoiser import MldDenoiser
from seeme_amd.schedulers import DDIMScheduler
from seeme_amd.weights_recipe import load_recipe_
dev = torch.device("cuda:0")
abl = types.SimpleNamespace(MLP_DIST=False, PE_TYPE="mld", SKIP_CONNECT=True, VAE_TYPE="actor", DIFF_PE_TYPE="mld", MD_TRANS=True)
g = torch.Generator().manual_seed(7)
B, T = int(sys.argv[2]), 70
vae = load_recipe_(MldVae(abl, nfeats=75, latent_dim=[1, 256], arch="encoder_decoder")).to(dev).eval()
vae.precision = "fp16"
x = torch.randn(B, T, 75, generator=g).to(dev)
lengths = [T - (i % 5) * 3 for i in range(B)]
lengths[0] = T
out = {}
with torch.no_grad():
    out["dist"] = vae.encode_dist(x, lengths).cpu()
    out["feats"] = vae.decode(torch.randn(1, B, 256, generator=g).to(dev), lengths).cpu()
    den = load_recipe_(MldDenoiser(abl, nfeats=75, condition=["text", "interactee"], latent_dim=[1, 256], ff_size=128, num_layers=5, num_heads=1,
                                   weight_dtype="fp16")).to(dev).eval()
    sch = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                        set_alpha_to_one=False, steps_offset=1)
    sch.set_timesteps(5)
    lat = torch.randn(B, 1, 256, generator=g).to(dev)
    cond = torch.randn(B, 1, 256, generator=g).to(dev)
    out["latent"] = den.sample_loop(lat, cond, sch).cpu()
torch.save(out, sys.argv[1])
