"""GPU debug: save fp16-path VAE encode / decode outputs (SEEME_VAE_FUSED selects the sequence), or compare two saves."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        d = (a[k] - b[k]).abs()
        print(k, "max abs diff", float(d.max()), "ref max", float(b[k].abs().max()), "where", [int(i) for i in torch.nonzero(d == d.max())[0]])
    sys.exit(0)
from seeme_amd.mld_vae import MldVae
from seeme_amd.weights_recipe import load_recipe_
dev = torch.device("cuda:0")
import types
abl = types.SimpleNamespace(MLP_DIST=False, PE_TYPE="mld", SKIP_CONNECT=True, VAE_TYPE="actor", DIFF_PE_TYPE="mld", MD_TRANS=True)
vae = load_recipe_(MldVae(abl, nfeats=75, latent_dim=[1, 256], arch="encoder_decoder")).to(dev).eval()
vae.precision = "fp16"
g = torch.Generator().manual_seed(1)
B, T = 32, 196
x = torch.randn(B, T, 75, generator=g).to(dev)
lengths = [T] * B
lengths[3] = 150
with torch.no_grad():
    dist = vae.encode_dist(x, lengths)
    z = torch.randn(1, B, 256, generator=g).to(dev)
    feats = vae.decode(z, lengths)
    vae.precision = "fp32"
    d32, f32 = vae.encode_dist(x, lengths), vae.decode(z, lengths)
print("vs fp32 path: dist max abs err", float((dist - d32).abs().max()), "feats", float((feats - f32).abs().max()),
      " mean abs err dist", float((dist - d32).abs().mean()), "feats", float((feats - f32).abs().mean()))
torch.save({"dist": dist.cpu(), "feats": feats.cpu()}, sys.argv[1])
