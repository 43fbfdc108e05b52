"""Parity of the HIP path (through the C-ABI) against the reference-generated golden fixtures and the
numpy oracle.  Needs a real MI355X: run with `pytest -m gpu`."""
import ctypes as C
import os
import types

import numpy as np
import pytest
import torch

from conftest import REPO, elem_err, load_golden, rel_err
from oracle import mld_oracle as O
from seeme_amd.weights_recipe import load_recipe_, recipe_state_dict
from seeme_amd import shapes

pytestmark = pytest.mark.gpu

# fp32 MFMA / fp32 VALU path vs fp32 reference: BASELINE.json asks for 1e-4 relative
TOL_F32 = 1e-4
# bf16-rounded weights, fp32 activations/accumulation (reported, not the parity gate)
TOL_BF16W = 3e-2


def ablation():
    return types.SimpleNamespace(MLP_DIST=False, PE_TYPE="mld", SKIP_CONNECT=True, VAE_TYPE="actor",
                                 DIFF_PE_TYPE="mld", MD_TRANS=True)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    return torch.device("cuda:0")


def make_vae(F, dev):
    from seeme_amd.mld_vae import MldVae
    return load_recipe_(MldVae(ablation(), nfeats=F, latent_dim=[1, 256], arch="encoder_decoder")).to(dev).eval()


def make_den(dev, cond=("text", "interactee"), weight_dtype="fp32"):
    from seeme_amd.mld_denoiser import MldDenoiser
    return load_recipe_(MldDenoiser(ablation(), nfeats=75, condition=list(cond), latent_dim=[1, 256], ff_size=128,
                                    num_layers=5, num_heads=1, weight_dtype=weight_dtype)).to(dev).eval()


# ----------------------------------------------------------------------------- generic linear
@pytest.mark.parametrize("M,N,K,act,pre_act,ln,res,concat,pre_ln", [
    (32, 256, 256, 0, 0, False, False, False, False),
    (70, 768, 256, 0, 0, False, False, False, False),
    (45, 256, 132, 0, 0, False, True, False, False),
    (33, 256, 512, 0, 0, True, True, True, False),
    (64, 128, 256, 2, 0, False, False, False, False),
    (19, 132, 256, 0, 0, False, False, False, True),
    (50, 5120, 256, 0, 3, False, False, False, False),
    (100, 512, 3, 1, 0, False, False, False, False),
    (37, 1024, 256, 1, 1, False, False, False, False),
])
def test_linear(dev, M, N, K, act, pre_act, ln, res, concat, pre_ln):
    from seeme_amd import _lib as L
    rng = np.random.default_rng(M * 1000 + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32)
    lw, lb = (1 + 0.1 * rng.standard_normal(N)).astype(np.float32), (0.1 * rng.standard_normal(N)).astype(np.float32)
    pw, pb = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32), (0.1 * rng.standard_normal(K)).astype(np.float32)
    Kp = (K + 15) // 16 * 16
    Wp = np.zeros((N, Kp), np.float32)
    Wp[:, :K] = W
    acts = {0: lambda x: x, 1: O.relu, 2: O.gelu, 3: O.silu}
    a = A
    if pre_ln:
        a = O.layer_norm(a, pw, pb)
    ref = acts[act](acts[pre_act](a) @ W.T + b)
    if res:
        ref = ref + R
    if ln:
        ref = O.layer_norm(ref, lw, lb)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    tA, tW, tb, tR, tlw, tlb, tpw, tpb = map(t, (A, Wp, b, R, lw, lb, pw, pb))
    Y = torch.full((M, N), float("nan"), device=dev)
    args = L.LinearArgs()
    if concat:
        K1 = K // 2
        tA1, tA2 = tA[:, :K1].contiguous(), tA[:, K1:].contiguous()
        args.A, args.lda, args.A2, args.lda2, args.K1 = tA1.data_ptr(), K1, tA2.data_ptr(), K - K1, K1
    else:
        args.A, args.lda, args.A2, args.lda2, args.K1 = tA.data_ptr(), K, 0, 0, K
    args.W, args.ldw, args.bias = tW.data_ptr(), Kp, tb.data_ptr()
    args.res, args.ldr = (tR.data_ptr(), N) if res else (0, 0)
    args.ln_w, args.ln_b = (tlw.data_ptr(), tlb.data_ptr()) if ln else (0, 0)
    args.pre_ln_w, args.pre_ln_b = (tpw.data_ptr(), tpb.data_ptr()) if pre_ln else (0, 0)
    args.Y, args.ldy, args.M, args.N, args.K = Y.data_ptr(), N, M, N, K
    args.pre_act, args.act, args.eps = pre_act, act, 1e-5
    L.check(L.lib().seeme_linear(C.byref(args), L.current_stream()), "seeme_linear")
    torch.cuda.synchronize()
    assert rel_err(Y.cpu().numpy(), ref) < 2e-5


# ----------------------------------------------------------------------------- VAE
@pytest.mark.parametrize("name,F", [("vae_F132_T24.npz", 132), ("vae_F75_T60.npz", 75), ("vae_F132_T196.npz", 132)])
def test_vae_golden(dev, name, F):
    g = load_golden(name)
    vae = make_vae(F, dev)
    lengths = g["lengths"].tolist()
    x = torch.from_numpy(g["features"]).to(dev)
    _, dist = vae.encode(x, None, lengths)
    torch.cuda.synchronize()
    assert rel_err(dist.loc.cpu().numpy(), g["mu"]) < TOL_F32
    assert rel_err(dist.scale.cpu().numpy(), g["std"]) < TOL_F32
    assert elem_err(dist.loc.cpu().numpy(), g["mu"]) < TOL_F32 and elem_err(dist.scale.cpu().numpy(), g["std"]) < TOL_F32   # element by element
    dec = vae.decode(torch.from_numpy(g["mu"]).to(dev), lengths)
    torch.cuda.synchronize()
    assert dec.shape == g["decoded"].shape
    assert rel_err(dec.cpu().numpy(), g["decoded"]) < TOL_F32 and elem_err(dec.cpu().numpy(), g["decoded"]) < TOL_F32


def test_vae_vs_oracle_bench_shape(dev):
    """BASELINE config 1: B=4, T=196, F=132 random-pose sequences, ragged lengths."""
    rng = np.random.Generator(np.random.PCG64(1234))
    x = rng.standard_normal((4, 196, 132)).astype(np.float32)
    lengths = [196, 1, 77, 196]
    P = recipe_state_dict(shapes.vae_shapes(132))
    mu, std = O.vae_encode(P, x, lengths)
    vae = make_vae(132, dev)
    _, dist = vae.encode(torch.from_numpy(x).to(dev), None, lengths)
    assert rel_err(dist.loc.cpu().numpy(), mu) < TOL_F32
    dec = vae.decode(torch.from_numpy(mu).to(dev), lengths)
    assert rel_err(dec.cpu().numpy(), O.vae_decode(P, mu, lengths)) < TOL_F32


def test_vae_batch_independence(dev):
    """Size-independent property at the bench batch: every sequence is encoded independently of its
    neighbours (no cross-batch op on the path)."""
    vae = make_vae(132, dev)
    x = torch.randn(32, 196, 132, device=dev)
    lengths = [196] * 32
    full = vae.encode_dist(x, lengths)
    part = vae.encode_dist(x[5:7].contiguous(), lengths[5:7])
    assert torch.equal(full[:, 5:7], part)
    d_full = vae.decode(full[0:1], lengths)
    d_part = vae.decode(full[0:1, 5:7].contiguous(), lengths[5:7])
    assert torch.equal(d_full[5:7], d_part)


# ----------------------------------------------------------------------------- denoiser
@pytest.mark.parametrize("N", [1, 2])
def test_denoiser_golden(dev, N):
    g = load_golden(f"denoiser_N{N}.npz")
    den = make_den(dev)
    s, c = torch.from_numpy(g["sample"]).to(dev), torch.from_numpy(g["cond"]).to(dev)
    for t in (981, 501, 1):
        y = den(sample=s, timestep=torch.tensor(t), encoder_hidden_states=c)[0]
        assert rel_err(y.cpu().numpy(), g[f"out_t{t}"]) < TOL_F32 and elem_err(y.cpu().numpy(), g[f"out_t{t}"]) < TOL_F32
    y = den(sample=s, timestep=torch.from_numpy(g["tvec"]).to(dev), encoder_hidden_states=c)[0]
    assert rel_err(y.cpu().numpy(), g["out_tvec"]) < TOL_F32 and elem_err(y.cpu().numpy(), g["out_tvec"]) < TOL_F32


@pytest.mark.parametrize("wd,tol", [("bf16", TOL_BF16W), ("fp16", 4e-3)])
def test_denoiser_16bit_weights(dev, wd, tol):
    """16-bit weight images (activations / accumulation stay fp32): error is MEASURED and reported; the 1e-4 gate
    applies to the fp32 image only."""
    g = load_golden("denoiser_N1.npz")
    den = make_den(dev, weight_dtype=wd)
    s, c = torch.from_numpy(g["sample"]).to(dev), torch.from_numpy(g["cond"]).to(dev)
    y = den(sample=s, timestep=torch.tensor(501), encoder_hidden_states=c)[0]
    e = rel_err(y.cpu().numpy(), g["out_t501"])
    gl = load_golden("ddim50_N1_B3.npz")
    sch = _sched()
    sch.set_timesteps(50)
    z = den.sample_loop(torch.from_numpy(gl["latents"]).to(dev), torch.from_numpy(gl["cond_bf"]).to(dev), sch)
    e50 = rel_err(z.cpu().numpy(), gl["out"])
    print(f"{wd}-weight denoiser rel err: one forward {e:.3e}, 50-step DDIM latent {e50:.3e}")
    assert e < tol


def _sched(kind="ddim"):
    from seeme_amd.schedulers import DDIMScheduler, DDPMScheduler
    kw = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
              clip_sample=False)
    if kind == "ddim":
        return DDIMScheduler(set_alpha_to_one=False, steps_offset=1, **kw)
    return DDPMScheduler(variance_type="fixed_small", **kw)


def test_ddim_loop_golden(dev):
    g = load_golden("ddim50_N1_B3.npz")
    den = make_den(dev)
    sch = _sched()
    sch.set_timesteps(int(g["steps"]))
    out = den.sample_loop(torch.from_numpy(g["latents"]).to(dev), torch.from_numpy(g["cond_bf"]).to(dev), sch)
    assert out.shape == g["out"].shape
    assert rel_err(out.cpu().numpy(), g["out"]) < 5e-4   # 50 chained steps
    assert elem_err(out.cpu().numpy(), g["out"]) < 5e-4


def test_ddim_loop_cfg_golden(dev):
    g = load_golden("ddim10_N2_B2_cfg.npz")
    den = make_den(dev, cond=("text", "scene", "interactee"))
    sch = _sched()
    sch.set_timesteps(int(g["steps"]))
    out = den.sample_loop(torch.from_numpy(g["latents"]).to(dev), torch.from_numpy(g["cond_bf"]).to(dev), sch,
                          guidance_scale=float(g["guidance_scale"]))
    assert rel_err(out.cpu().numpy(), g["out"]) < 5e-4


def test_loop_equals_stepwise(dev):
    """The fused loop equals stepping the module + scheduler one step at a time (both on the HIP path)."""
    den = make_den(dev)
    sch = _sched()
    sch.set_timesteps(50)
    B = 32
    lat = torch.randn(B, 1, 256, device=dev)
    cond = torch.randn(B, 1, 256, device=dev)
    fused = den.sample_loop(lat, cond, sch)
    x = lat.clone()
    for t in sch.timesteps:
        eps = den(sample=x, timestep=t, encoder_hidden_states=cond.permute(1, 0, 2))[0]
        x = sch.step(eps, t, x, eta=0.0).prev_sample
    assert rel_err(fused.cpu().numpy(), x.permute(1, 0, 2).cpu().numpy()) < 1e-4


def test_ddpm_loop_vs_oracle(dev):
    den = make_den(dev)
    sch = _sched("ddpm")
    sch.set_timesteps(1000)
    B, steps = 2, 1000
    rng = np.random.default_rng(3)
    lat = rng.standard_normal((B, 1, 256)).astype(np.float32)
    cond = rng.standard_normal((B, 1, 256)).astype(np.float32)
    noise = rng.standard_normal((steps, B, 1, 256)).astype(np.float32)
    # oracle only for the first 20 steps (CPU time); compare the same prefix by truncating the schedule
    k = 20
    sch.timesteps = sch.timesteps[:k]
    out = den.sample_loop(torch.from_numpy(lat).to(dev), torch.from_numpy(cond).to(dev), sch,
                          step_noise=torch.from_numpy(noise[:k]).to(dev))
    P = recipe_state_dict(shapes.denoiser_shapes())
    acp = O.alphas_cumprod(O.make_betas())
    x = lat.copy()
    cs = np.transpose(cond, (1, 0, 2))
    for i, t in enumerate(range(999, 999 - k, -1)):
        eps = O.denoiser_forward(P, x, t, cs)
        x = O.ddpm_step(acp, eps, t, x, noise[i])
    assert rel_err(out.cpu().numpy(), np.transpose(x, (1, 0, 2))) < 2e-4


# ----------------------------------------------------------------------------- SMPL LBS
@pytest.mark.parametrize("rotmat", [False, True])
def test_smpl_lbs(dev, rotmat):
    from seeme_amd.smpl import SMPL
    model = O.make_synthetic_smpl(1234)
    smpl = SMPL.synthetic(1234).to(dev)
    rng = np.random.default_rng(5)
    M = 7
    betas = (rng.standard_normal((M, 10)) * 0.5).astype(np.float32)
    go = (rng.standard_normal((M, 3)) * 0.8).astype(np.float32)
    bp = (rng.standard_normal((M, 69)) * 0.4).astype(np.float32)
    bp[0] = 0
    go[0] = 0
    tr = rng.standard_normal((M, 3)).astype(np.float32)
    t = lambda x: torch.from_numpy(x).to(dev)
    if rotmat:
        Rg = O.rodrigues(go).reshape(M, 1, 3, 3)
        Rb = O.rodrigues(bp.reshape(-1, 3)).reshape(M, 23, 3, 3)
        jo, vo = O.smpl_lbs(model, betas, Rg, Rb, tr, pose2rot=False)
        out = smpl(betas=t(betas), body_pose=t(Rb), global_orient=t(Rg), transl=t(tr), pose2rot=False)
    else:
        jo, vo = O.smpl_lbs(model, betas, go, bp, tr)
        out = smpl(betas=t(betas), body_pose=t(bp), global_orient=t(go), transl=t(tr))
    assert out.joints.shape == (M, 45, 3) and out.vertices.shape == (M, 6890, 3)
    assert rel_err(out.joints.cpu().numpy(), jo) < TOL_F32
    assert rel_err(out.vertices.cpu().numpy(), vo) < TOL_F32
    # joints-only fast path gives the same joints without forming the mesh
    out2 = smpl(betas=t(betas), body_pose=t(bp if not rotmat else Rb), global_orient=t(go if not rotmat else Rg),
                transl=t(tr), pose2rot=not rotmat, return_verts=False)
    assert out2.vertices is None
    assert torch.equal(out2.joints, out.joints)
    # index bookkeeping is exact (BASELINE.json: joint/vertex indices bit-exact)
    assert smpl.parents.tolist() == O.SMPL_PARENTS.tolist()
    assert smpl.vertex_joint_selector.extra_joints_idxs.tolist() == O.SMPL_EXTRA_VERTEX_IDS.tolist()


# ----------------------------------------------------------------------------- geometry / PointNet
def test_geometry_golden(dev):
    from seeme_amd import geometry as G
    g = load_golden("misc.npz")
    aa, r6 = torch.from_numpy(g["aa"]).to(dev), torch.from_numpy(g["rot6d"]).to(dev)
    assert rel_err(G.aa_to_quat(aa).cpu().numpy(), g["aa_to_quat"]) < 1e-5
    assert rel_err(G.aa_to_rotmat(aa).cpu().numpy(), g["aa_to_rotmat"]) < 1e-5
    assert rel_err(G.rot6d_to_rotmat(r6).cpu().numpy(), g["rot6d_to_rotmat"]) < 1e-5
    assert rel_err(G.rot6d_to_rotmat(r6, "diffusion").cpu().numpy(), O.rot6d_to_rotmat(g["rot6d"], "diffusion")) < 1e-5
    q = torch.randn(33, 4, device=dev)
    assert rel_err(G.quat_to_rotmat(q).cpu().numpy(), O.quat_to_rotmat(q.cpu().numpy())) < 1e-5
    x = torch.randn(3, 7, 75, device=dev)
    mean, std = torch.randn(1, 90), torch.rand(1, 90) + 0.5
    assert rel_err(G.renorm(x, mean, std).cpu().numpy(), O.renorm(x.cpu().numpy(), mean.numpy(), std.numpy())) < 1e-6


def test_pointnet_golden(dev):
    from seeme_amd.respointnet import ResnetPointnet
    g = load_golden("misc.npz")
    pn = load_recipe_(ResnetPointnet(512, 256)).to(dev).eval()
    out = pn(torch.from_numpy(g["points"]).to(dev))
    assert rel_err(out.cpu().numpy(), g["pointnet"]) < TOL_F32
    # ragged point count (not a multiple of the 32-row tile) against the oracle
    pts = np.random.default_rng(2).uniform(-3, 3, (3, 777, 3)).astype(np.float32)
    P = recipe_state_dict(shapes.pointnet_shapes())
    assert rel_err(pn(torch.from_numpy(pts).to(dev)).cpu().numpy(), O.pointnet_forward(P, pts)) < TOL_F32


# ----------------------------------------------------------------------------- MLD orchestration
def _mld(dev, cfg_name="config_mld_egobody.yaml", T=24, **over):
    import os
    from conftest import REPO
    from seeme_amd.config import parse_config
    from seeme_amd.mld import MLD, SyntheticEgoDataModule
    from seeme_amd.smpl import SMPL
    cfg = parse_config(os.path.join(REPO, "configs", cfg_name))
    for k, v in over.items():
        cfg.model[k] = v
    dm = SyntheticEgoDataModule(nfeats=cfg.model.nfeats, T=T, n_points=512, device=dev)
    model = MLD(cfg, dm, smpl_model=SMPL.synthetic(1234))
    load_recipe_(model.vae), load_recipe_(model.denoiser)
    if hasattr(model, "proscene"):
        load_recipe_(model.proscene.scene_enc)
    return model.to(dev).eval(), dm, cfg


def test_mld_sample_vs_oracle_mpjpe(dev):
    """ego_eval (condition -> 50-step DDIM -> decode -> renorm -> SMPL joints) against the oracle chain;
    BASELINE.json gate: MPJPE within 1e-3 mm of the reference path on identical inputs."""
    model, dm, cfg = _mld(dev, T=24)
    B = 3
    batch = dm.batch(B, idx=1)
    gen = torch.Generator().manual_seed(5)
    lat = torch.randn(B, 1, 256, generator=gen).to(dev)
    eps_c = torch.randn(1, B, 256, generator=gen).to(dev)     # the condition is a SAMPLE of the posterior (mld.py:1280)
    rs = model.ego_eval(batch, latents=lat, cond_noise=eps_c)
    # oracle
    Pv, Pd = recipe_state_dict(shapes.vae_shapes(75)), recipe_state_dict(shapes.denoiser_shapes())
    motion, transl, beta = (t.cpu().numpy() for t in batch[:3])
    lengths = [24] * B
    f_int = np.concatenate([motion[:, :, 1], transl[:, 1]], -1)
    mu, sd = O.vae_encode(Pv, f_int, lengths)
    z_cond = mu + eps_c.cpu().numpy() * sd
    z = O.diffusion_reverse(Pd, np.transpose(z_cond, (1, 0, 2)), lat.cpu().numpy(), 50)
    feats = O.renorm(O.vae_decode(Pv, z, lengths), dm.mean.cpu().numpy(), dm.std.cpu().numpy())
    smpl = O.make_synthetic_smpl(1234)
    j, _ = O.smpl_lbs(smpl, beta[:, 0].reshape(-1, 10), feats[..., :3].reshape(-1, 3), feats[..., 3:72].reshape(-1, 69),
                      feats[..., -3:].reshape(-1, 3), return_verts=False)
    j = j.reshape(B, 24, 45, 3)[:, :, :24]
    got = rs["joints_rst"].cpu().numpy()
    assert rel_err(rs["m_rst"].cpu().numpy(), feats) < 5e-4
    mpjpe_between = float(np.linalg.norm(got - j, axis=-1).mean() * 1000.0)
    print("MPJPE(HIP path, oracle path) =", mpjpe_between, "mm")
    assert mpjpe_between < 1e-2         # joint-to-joint distance between the two paths' outputs (measured 2e-3 mm)
    # metric plumbing: MPJPE of prediction vs ground truth equals the oracle's definition after alignment
    from seeme_amd.mld import EgoMetrics
    m = EgoMetrics.per_sequence(rs["joints_rst"], rs["joints_ref"], rs["lengths"], rs["orientation_quat_rst"], rs["orientation_quat_ref"])
    assert all(torch.isfinite(v).all() for v in m.values())
    want = O.ego_metrics(rs["joints_rst"].double().cpu().numpy(), rs["joints_ref"].double().cpu().numpy(),
                         rs["orientation_quat_rst"].double().cpu().numpy(), rs["orientation_quat_ref"].double().cpu().numpy(),
                         rs["lengths"], "val")
    em = EgoMetrics()
    em.update(rs["joints_rst"], rs["joints_ref"], rs["lengths"], rs["orientation_quat_rst"], rs["orientation_quat_ref"], split="val")
    got = em.compute()
    assert abs(got["MPJPE"] - want["MPJPE"]) < 1e-3 * max(1.0, want["MPJPE"]) and got["count_seq"] == want["count_seq"]
    # the gate itself: the MPJPE metric (prediction vs ground truth, mm) of the HIP path and of the oracle path, which
    # received the same inputs, differ by less than 1e-3 mm
    ref_path = O.ego_metrics(j.astype(np.float64), rs["joints_ref"].double().cpu().numpy(),
                             rs["orientation_quat_rst"].double().cpu().numpy(), rs["orientation_quat_ref"].double().cpu().numpy(),
                             rs["lengths"], "val")
    print("MPJPE vs ground truth: HIP path", want["MPJPE"], "mm, oracle path", ref_path["MPJPE"], "mm, difference",
          abs(want["MPJPE"] - ref_path["MPJPE"]), "mm")
    assert abs(want["MPJPE"] - ref_path["MPJPE"]) < 1e-3
    # the throughput mode of bench.py (fp16 weight image, fp16 VAE operands) on the same inputs: measured and reported,
    # not held to the fp32 gate
    model.denoiser.weight_dtype = "fp16"
    model.vae.precision = "fp16"
    rs16 = model.ego_eval(batch, latents=lat, cond_noise=eps_c)
    m16 = O.ego_metrics(rs16["joints_rst"].double().cpu().numpy(), rs16["joints_ref"].double().cpu().numpy(),
                        rs16["orientation_quat_rst"].double().cpu().numpy(), rs16["orientation_quat_ref"].double().cpu().numpy(),
                        rs16["lengths"], "val")
    between16 = float(np.linalg.norm(rs16["joints_rst"].cpu().numpy() - j, axis=-1).mean() * 1000.0)
    print("fp16 mode: MPJPE vs ground truth", m16["MPJPE"], "mm (oracle path", ref_path["MPJPE"], "mm), difference",
          abs(m16["MPJPE"] - ref_path["MPJPE"]), "mm; joint-to-joint distance to the oracle path", between16, "mm")
    assert between16 < 5.0


def test_throughput_mode_meets_mpjpe_gate_at_bench_size(dev):
    """north_star: "MPJPE within 1e-3 mm of the reference" for the configuration bench.py times (fp16 denoiser weight image,
    fp16 VAE MFMA operands) at the bench size B=32, T=196.  MPJPE is a mean over the evaluated sequences (compute.py:488-580):
    over six batches (192 sequences) the throughput mode's MPJPE stays within 1e-3 mm of the fp32 path's, and for the first
    batch the fp32 path's MPJPE is itself within 1e-3 mm of the CPU oracle's on the same inputs (measured: 1e-5 mm).  A SINGLE
    batch shifts by 5e-5 ... 1.6e-3 mm: that is rounding noise of the 16-bit operands -- the one- and three-kernel-per-layer
    schedules of the same fp16 VAE, both 4.3e-4 mean abs error from the fp32 path, move one batch's MPJPE by up to 1.3e-3 mm
    against each other -- so single batches are bounded at 3e-3 only.  The bf16 weight image does NOT meet the gate
    (2e-3 ... 5e-3 mm per batch): reported."""
    from oracle import mld_oracle_torch as OT
    from seeme_amd.mld import EgoMetrics
    model, dm, cfg = _mld(dev, T=196)
    B = 32
    deltas, deltas_bf16, seqs = [], [], {"fp32": [], "fp16": [], "bf16": []}
    for it in range(6):
        batch = dm.batch(B, idx=20 + it)
        gen = torch.Generator().manual_seed(100 + it)
        lat, eps = torch.randn(B, 1, 256, generator=gen).to(dev), torch.randn(1, B, 256, generator=gen).to(dev)
        m = {}
        for tag, wd, vp in (("fp32", "fp32", "fp32"), ("fp16", "fp16", "fp16"), ("bf16", "bf16", "fp16")):
            model.denoiser.weight_dtype, model.vae.precision = wd, vp
            rs = model.ego_eval(batch, latents=lat, cond_noise=eps)
            ps = EgoMetrics.per_sequence(rs["joints_rst"], rs["joints_ref"], rs["lengths"])["MPJPE"].double()
            seqs[tag].append(ps)
            m[tag] = ps.mean().item()
            if tag == "fp32" and it == 0:
                # the oracle chain on the same inputs (PyTorch-CPU restatement, pinned by the reference fixtures)
                Pv, Pd = OT.to_torch(recipe_state_dict(shapes.vae_shapes(75))), OT.to_torch(recipe_state_dict(shapes.denoiser_shapes()))
                motion, transl, beta = (t.cpu() for t in batch[:3])
                f_int = torch.cat([motion[:, :, 1], transl[:, 1]], -1)
                mu, sd = OT.vae_encode(Pv, f_int, [196] * B)
                z = OT.diffusion_reverse(Pd, (mu + eps.cpu() * sd).permute(1, 0, 2), lat.cpu(), 50)
                feats = OT.vae_decode(Pv, z, [196] * B).numpy() * dm.std.cpu().numpy()[0, :75] + dm.mean.cpu().numpy()[0, :75]
                smpl = O.make_synthetic_smpl(1234)
                j, _ = O.smpl_lbs(smpl, beta[:, 0].reshape(-1, 10).numpy(), feats[..., :3].reshape(-1, 3), feats[..., 3:72].reshape(-1, 69),
                                  feats[..., -3:].reshape(-1, 3), return_verts=False)
                j = torch.from_numpy(j.reshape(B, 196, 45, 3)[:, :, :24]).to(dev)
                mo = EgoMetrics.per_sequence(j, rs["joints_ref"], rs["lengths"])["MPJPE"].double().mean().item()
                print(f"MPJPE fp32 HIP path {m['fp32']:.6f} mm, oracle path {mo:.6f} mm, difference {abs(m['fp32'] - mo):.2e} mm")
                assert abs(m["fp32"] - mo) < 1e-3
        deltas.append(abs(m["fp16"] - m["fp32"]))
        deltas_bf16.append(abs(m["bf16"] - m["fp32"]))
    tot = {k: torch.cat(v).mean().item() for k, v in seqs.items()}
    print("MPJPE delta of the fp16 throughput mode vs the fp32 path, mm: per batch", deltas, " bf16 weight image:", deltas_bf16,
          " over the 192 sequences: fp16", abs(tot["fp16"] - tot["fp32"]), "bf16", abs(tot["bf16"] - tot["fp32"]))
    assert abs(tot["fp16"] - tot["fp32"]) < 1e-3 and max(deltas) < 3e-3, (tot, deltas)


def test_autograd_twin_matches_hip(dev):
    from seeme_amd.denoiser_autograd import denoiser_forward_torch
    den = make_den(dev, cond=("text", "scene", "interactee"))
    g = load_golden("denoiser_N2.npz")
    s, c = torch.from_numpy(g["sample"]).to(dev), torch.from_numpy(g["cond"]).to(dev)
    t = torch.from_numpy(g["tvec"]).to(dev)
    y_hip = den(sample=s, timestep=t, encoder_hidden_states=c)[0]
    y_tw = denoiser_forward_torch(den, s, t, c)
    assert rel_err(y_tw.detach().cpu().numpy(), g["out_tvec"]) < TOL_F32
    assert rel_err(y_hip.cpu().numpy(), y_tw.detach().cpu().numpy()) < TOL_F32


def test_training_step_scene_interactee(dev):
    """BASELINE config 3 shape (scene + interactee, N = 2) at small size: loss is finite, gradients reach every
    trainable parameter the reference trains, and a few AdamW steps reduce the loss on a fixed batch."""
    model, dm, cfg = _mld(dev, "config_mld_scene.yaml", T=16)
    model.train()
    batch = dm.batch(4, idx=3, with_scene=True)
    g = torch.Generator().manual_seed(11)
    noise = torch.randn(4, 1, 256, generator=g).to(dev)
    ts = torch.randint(0, 1000, (4,), generator=g).to(dev)
    losses = []
    for _ in range(8):
        torch.manual_seed(5)                                 # the same rsample noise and dropout masks every step: a fixed objective
        rs = model.train_diffusion_forward(batch, noise=noise, timesteps=ts)
        # frozen stochastic parts (vae rsample) make z differ per call: fix by reusing the first target
        loss = model.losses["train"].update(rs)
        model.optimizer_step(loss)
        losses.append(float(loss))
    assert all(np.isfinite(losses))
    trainable = {n for n, p in model.named_parameters() if p.requires_grad}
    assert any(n.startswith("denoiser.") for n in trainable) and any(n.startswith("output_scene.") for n in trainable)
    assert not any(n.startswith("vae.") or n.startswith("proscene.") or n.startswith("smpl_model.") for n in trainable)
    no_grad = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
    assert no_grad == ["denoiser.mem_pos.pe"]          # never used by trans_enc
    assert losses[-1] < losses[0]


# ----------------------------------------------------------------------------- edge cases
@pytest.mark.parametrize("H", [2, 4])
def test_denoiser_multihead_vs_oracle(dev, H):
    """num_heads is a config knob of the reference (denoiser.yaml:7); shipped configs use 1."""
    from seeme_amd.mld_denoiser import MldDenoiser
    den = load_recipe_(MldDenoiser(ablation(), condition=["text", "scene", "interactee"], latent_dim=[1, 256], ff_size=128,
                                   num_layers=5, num_heads=H)).to(dev).eval()
    rng = np.random.default_rng(H)
    s = rng.standard_normal((5, 1, 256)).astype(np.float32)
    c = rng.standard_normal((3, 5, 256)).astype(np.float32)       # N = 3 condition tokens
    t = np.array([3, 999, 250, 0, 600])
    P = recipe_state_dict(shapes.denoiser_shapes())
    ref = O.denoiser_forward(P, s, t, c, nhead=H)
    y = den(sample=torch.from_numpy(s).to(dev), timestep=torch.from_numpy(t).to(dev),
            encoder_hidden_states=torch.from_numpy(c).to(dev))[0]
    assert rel_err(y.cpu().numpy(), ref) < TOL_F32


@pytest.mark.parametrize("wd,tol", [("fp32", 5e-4), ("fp16", 6e-3), ("bf16", TOL_BF16W)])
@pytest.mark.parametrize("H", [1, 2])
@pytest.mark.parametrize("N", [1, 3])
@pytest.mark.parametrize("cfg", [False, True])
def test_sampling_kernel_variants(dev, wd, tol, H, N, cfg):
    """Every compiled variant of the persistent kernel (weight type x CFG pair x one/many condition tokens x
    folded/unfolded out_proj) against the oracle loop, 6 DDIM steps, B = 3."""
    from seeme_amd.mld_denoiser import MldDenoiser
    den = load_recipe_(MldDenoiser(ablation(), condition=["text", "scene", "interactee"], latent_dim=[1, 256], ff_size=128,
                                   num_layers=5, num_heads=H, weight_dtype=wd)).to(dev).eval()
    rng = np.random.default_rng(100 * H + 10 * N + int(cfg))
    B, steps, gs = 3, 6, (2.5 if cfg else 1.0)
    lat = rng.standard_normal((B, 1, 256)).astype(np.float32)
    cond = rng.standard_normal(((2 * B if cfg else B), N, 256)).astype(np.float32)
    P = recipe_state_dict(shapes.denoiser_shapes())
    ref = O.diffusion_reverse(P, cond, lat, steps, guidance_scale=gs, nhead=H)
    sch = _sched()
    sch.set_timesteps(steps)
    out = den.sample_loop(torch.from_numpy(lat).to(dev), torch.from_numpy(cond).to(dev), sch, guidance_scale=gs)
    err = rel_err(out.cpu().numpy(), ref)
    assert err < tol, f"{wd} H={H} N={N} cfg={cfg}: rel err {err:.3e}"


@pytest.mark.parametrize("wd", ["fp32", "fp16"])
def test_large_batch_pairs_match_single_chains(dev, wd):
    """Batches above 256 run two independent samples per workgroup (one weight stream); the result must equal the
    one-sample-per-workgroup path (itself pinned to the oracle above).  B odd: the last workgroup has one sample.
    (k_den_sample on both sides: the cluster kernels have their own tests.)"""
    den = make_den(dev, weight_dtype=wd)
    den.cluster = 0
    g = torch.Generator(device="cpu").manual_seed(7)
    B, steps = 2 * 256 + 3, 4
    lat = torch.randn(B, 1, 256, generator=g).to(dev)
    cond = torch.randn(B, 1, 256, generator=g).to(dev)
    noise = torch.randn(steps, B, 256, generator=g).to(dev)
    sch = _sched("ddpm")
    sch.set_timesteps(steps)
    full = den.sample_loop(lat, cond, sch, step_noise=noise)                       # pairs
    parts = [den.sample_loop(lat[i:i + 200], cond[i:i + 200], sch, step_noise=noise[:, i:i + 200].contiguous())
             for i in range(0, B, 200)]                                            # single chains
    ref = torch.cat(parts, dim=1)
    assert full.shape == ref.shape == (1, B, 256)
    tol = 1e-5 if wd == "fp32" else 1e-4     # same arithmetic per sample; fp32 path differs only by accumulation order
    assert rel_err(full.cpu().numpy(), ref.cpu().numpy()) < tol


def test_vae_extreme_shapes(dev):
    """B = 1, the longest sequence the learned PE allows (T = 498 -> 500 tokens), and a length-1 sequence in a batch."""
    P = recipe_state_dict(shapes.vae_shapes(75))
    vae = make_vae(75, dev)
    rng = np.random.default_rng(9)
    for B, T, lengths in ((1, 498, [498]), (2, 40, [40, 1]), (1, 1, [1])):
        x = rng.standard_normal((B, T, 75)).astype(np.float32)
        mu, std = O.vae_encode(P, x, lengths)
        _, dist = vae.encode(torch.from_numpy(x).to(dev), None, lengths)
        assert rel_err(dist.loc.cpu().numpy(), mu) < TOL_F32
        dec = vae.decode(torch.from_numpy(mu).to(dev), lengths)
        assert rel_err(dec.cpu().numpy(), O.vae_decode(P, mu, lengths)) < TOL_F32
    with pytest.raises(ValueError):
        vae.encode(torch.zeros(2, 8, 75, device=dev), None, [8, 9, 3])     # wrong number of lengths
    with pytest.raises(Exception):
        vae.encode(torch.zeros(1, 600, 75, device=dev), None, [600])       # beyond the 500 learned positions


def test_full_size_bench_shape_properties(dev):
    """BASELINE full size (B=32, T=196, 50 steps): size-independent properties instead of a CPU oracle run --
    determinism, per-sample independence of the fused loop, and loop == 50 single steps (checked above at B=32)."""
    den = make_den(dev)
    sch = _sched()
    sch.set_timesteps(50)
    lat = torch.randn(32, 1, 256, device=dev)
    cond = torch.randn(32, 1, 256, device=dev)
    a = den.sample_loop(lat, cond, sch)
    b = den.sample_loop(lat, cond, sch)
    assert torch.equal(a, b)
    sub = den.sample_loop(lat[7:9].contiguous(), cond[7:9].contiguous(), sch)
    assert torch.equal(a[:, 7:9], sub)
    assert torch.isfinite(a).all()


def test_pointnet_bf16_vs_fp32(dev):
    """Fused bf16-MFMA PointNet against the fp32 path / oracle (bf16 rounding of weights and activations:
    tolerance 3e-2 of the output range, reported)."""
    from seeme_amd.respointnet import ResnetPointnet
    pn = load_recipe_(ResnetPointnet(512, 256)).to(dev).eval()
    pts = torch.from_numpy(np.random.default_rng(4).uniform(-3, 3, (3, 1000, 3)).astype(np.float32)).to(dev)   # ragged: 1000 = 7*128 + 104
    ref = pn(pts)
    pn.precision = "bf16"
    got = pn(pts)
    e = rel_err(got.cpu().numpy(), ref.cpu().numpy())
    print("bf16 PointNet rel err vs fp32 path:", e)
    assert e < 3e-2
    P = recipe_state_dict(shapes.pointnet_shapes())
    assert rel_err(got.cpu().numpy(), O.pointnet_forward(P, pts.cpu().numpy())) < 3e-2


def test_fused_adamw_matches_torch(dev):
    """seeme_adamw_step (one launch over all tensors) against torch.optim.AdamW on identical parameters and gradients:
    three steps incl. an LR change, odd sizes / unaligned views, state_dict round trip.  Same formula, different
    association of the fp32 operations: tolerance 2e-6 relative."""
    from seeme_amd.optim import FusedAdamWStep
    g = torch.Generator(device="cpu").manual_seed(5)
    shapes_ = [(256, 256), (768,), (3, 5, 7), (1,), (20001,), (1024, 256)]
    mk = lambda: [torch.nn.Parameter(torch.randn(*s, generator=torch.Generator().manual_seed(i)).to(dev)) for i, s in enumerate(shapes_)]
    pa, pb = mk(), mk()
    oa = torch.optim.AdamW(pa, lr=1e-3)
    ob = torch.optim.AdamW(pb, lr=1e-3)
    fb = FusedAdamWStep(ob)
    for it in range(3):
        for x, y in zip(pa, pb):
            gr = torch.randn(x.shape, generator=g).to(dev)
            x.grad, y.grad = gr.clone(), gr.clone()
        if it == 2:
            for o in (oa, ob):
                o.param_groups[0]["lr"] = 3e-4
        oa.step()
        fb.step()
        for x, y in zip(pa, pb):
            assert rel_err(y.detach().cpu().numpy(), x.detach().cpu().numpy()) < 2e-6
    sa, sb = oa.state_dict(), ob.state_dict()
    assert float(sb["state"][0]["step"]) == float(sa["state"][0]["step"]) == 3.0
    for k in sa["state"]:
        assert rel_err(sb["state"][k]["exp_avg_sq"].cpu().numpy(), sa["state"][k]["exp_avg_sq"].cpu().numpy()) < 2e-6
    # the checkpointed state continues in a fresh optimiser (what cli.py does on resume)
    pc = mk()
    for x, y in zip(pc, pb):
        x.data.copy_(y.data)
    oc = torch.optim.AdamW(pc, lr=3e-4)
    import copy
    oc.load_state_dict(copy.deepcopy(sb))            # (load_state_dict alone would alias ob's live moment tensors)
    fc = FusedAdamWStep(oc)
    for x, y, z in zip(pa, pb, pc):
        gr = torch.randn(x.shape, generator=g).to(dev)
        x.grad, y.grad, z.grad = gr.clone(), gr.clone(), gr.clone()
    oa.step(); fb.step(); fc.step()
    for x, y, z in zip(pa, pb, pc):
        assert rel_err(y.detach().cpu().numpy(), x.detach().cpu().numpy()) < 2e-6
        assert torch.equal(y.detach(), z.detach())


def test_pointnet_bf16_tile_walk(dev):
    """The persistent block kernels give every workgroup a contiguous range of 64-point tiles that may span scene
    boundaries, with the column max carried in registers between tiles.  Per-point arithmetic and the max are
    independent of that partition, so a scene encoded alone (one tile per workgroup) and inside a large batch (several
    tiles per workgroup, scene changes inside a range, partial last tiles) must agree BIT-exactly; the fp32 path bounds
    the values themselves."""
    from seeme_amd.respointnet import ResnetPointnet
    pn = load_recipe_(ResnetPointnet(512, 256)).to(dev).eval()
    rng = np.random.default_rng(11)
    for B, P, probe in ((40, 1000, (0, 17, 39)), (6, 20000, (0, 3, 5)), (700, 50, (0, 333, 699)), (2, 64, (1,))):
        pts = torch.from_numpy(rng.uniform(-3, 3, (B, P, 3)).astype(np.float32)).to(dev)
        pn.precision = "bf16"
        got = pn(pts)
        assert torch.isfinite(got).all()
        for i in probe:
            alone = pn(pts[i:i + 1].contiguous())
            assert torch.equal(alone[0], got[i]), (B, P, i, float((alone[0] - got[i]).abs().max()))
        pn.precision = "fp32"
        ref = pn(pts[list(probe)].contiguous())
        assert rel_err(got[list(probe)].cpu().numpy(), ref.cpu().numpy()) < 3e-2


def test_vae_fp16_mfma_mode(dev):
    """Throughput mode of the VAE (fp16 MFMA operands, fp32 accumulation and residual stream): measured against the
    reference fixtures; bound 2e-2 of the output range (reported, not the 1e-4 parity gate)."""
    from seeme_amd.mld_vae import MldVae
    for name, F in (("vae_F132_T24.npz", 132), ("vae_F75_T60.npz", 75), ("vae_F132_T196.npz", 132)):
        g = load_golden(name)
        vae = load_recipe_(MldVae(ablation(), nfeats=F, latent_dim=[1, 256], arch="encoder_decoder", precision="fp16")).to(dev).eval()
        lengths = g["lengths"].tolist()
        _, dist = vae.encode(torch.from_numpy(g["features"]).to(dev), None, lengths)
        e_mu = rel_err(dist.loc.cpu().numpy(), g["mu"])
        dec = vae.decode(torch.from_numpy(g["mu"]).to(dev), lengths)
        e_dec = rel_err(dec.cpu().numpy(), g["decoded"])
        print(f"fp16-MFMA VAE {name}: mu rel err {e_mu:.3e}, decode rel err {e_dec:.3e}")
        assert e_mu < 2e-2 and e_dec < 2e-2


def test_vae_long_ragged_sequences_fp16_vs_fp32_vs_oracle(dev):
    """Sequences longer than one 256-key chunk (S = T + 2 = 302 -> two score chunks, 8 score columns per lane in the
    register softmax) with ragged lengths: fp32 path against the oracle (1e-4), fp16 path against the fp32 path."""
    from seeme_amd.mld_vae import MldVae
    F_, T = 75, 300
    rng = np.random.default_rng(21)
    x = rng.standard_normal((3, T, F_)).astype(np.float32)
    lengths = [300, 257, 64]
    P = recipe_state_dict(shapes.vae_shapes(F_))
    vae = load_recipe_(MldVae(ablation(), nfeats=F_, latent_dim=[1, 256], arch="encoder_decoder")).to(dev).eval()
    _, d32 = vae.encode(torch.from_numpy(x).to(dev), None, lengths)
    mu_o, std_o = O.vae_encode(P, x, lengths)
    assert rel_err(d32.loc.cpu().numpy(), mu_o) < TOL_F32 and rel_err(d32.scale.cpu().numpy(), std_o) < TOL_F32
    dec32 = vae.decode(d32.loc, lengths)
    dec_o = O.vae_decode(P, mu_o, lengths)
    for b, Lb in enumerate(lengths):                 # frames past a sequence's length are unspecified (mld_vae.py:253)
        assert rel_err(dec32[b, :Lb].cpu().numpy(), dec_o[b, :Lb]) < TOL_F32
    vae.precision = "fp16"
    _, d16 = vae.encode(torch.from_numpy(x).to(dev), None, lengths)
    dec16 = vae.decode(d32.loc, lengths)
    assert rel_err(d16.loc.cpu().numpy(), d32.loc.cpu().numpy()) < 2e-2
    for b, Lb in enumerate(lengths):
        assert rel_err(dec16[b, :Lb].cpu().numpy(), dec32[b, :Lb].cpu().numpy()) < 2e-2


def test_vae_fp16_large_batch_matches_small_batch(dev):
    """From 256 row tiles on (B >= 167 at T=196) the fp16 QKV projection takes q, k and v from one staged operand tile
    instead of three workgroups per tile: the same sequences must come out BIT-identical in a large and a small batch
    (per-row arithmetic does not depend on the partition), and close to the fp32 path."""
    from seeme_amd.mld_vae import MldVae
    F_, T, B = 75, 196, 176
    rng = np.random.default_rng(33)
    x = torch.from_numpy(rng.standard_normal((B, T, F_)).astype(np.float32)).to(dev)
    lengths = [T] * B
    vae = load_recipe_(MldVae(ablation(), nfeats=F_, latent_dim=[1, 256], arch="encoder_decoder", precision="fp16")).to(dev).eval()
    big = vae.encode_dist(x, lengths)                                    # [2,B,256]
    idx = [0, 57, 175]
    small = vae.encode_dist(x[idx].contiguous(), [T] * len(idx))
    assert torch.equal(big[:, idx], small)
    vae.precision = "fp32"
    ref = vae.encode_dist(x[idx].contiguous(), [T] * len(idx))
    assert rel_err(small.cpu().numpy(), ref.cpu().numpy()) < 2e-2


# ----------------------------------------------------------------------------- train.py / test.py equivalents
@pytest.mark.gpu
def test_cli_train_resume_and_test(dev, tmp_path):
    """Two epochs of stage-2 training on synthetic batches, Lightning-layout checkpoints, resume from the newest one,
    strict reload for testing, metrics file (train.py:26-53,114-123,155-167; test.py:111-152)."""
    from seeme_amd import cli
    cfgp = os.path.join(REPO, "configs", "config_mld_egobody.yaml")
    common = ["--cfg", cfgp, "--batch_size", "4", "--nodebug", "--folder", str(tmp_path), "--frames", "24",
              "--iters_per_epoch", "2"]
    r1 = cli.train_main(common + ["--epochs", "2"])
    ck = sorted(os.listdir(r1["checkpoints"]))
    assert ck == ["epoch=0.ckpt", "epoch=1.ckpt"] and r1["step"] == 4 and np.isfinite(r1["total"])
    sd = cli.read_checkpoint(os.path.join(r1["checkpoints"], "epoch=1.ckpt"))["state_dict"]
    assert any(k.startswith("vae.encoder.") for k in sd) and any(k.startswith("denoiser.encoder.") for k in sd)
    # resume: continues after the newest checkpoint
    import yaml
    rcfg = tmp_path / "resume.yaml"
    base = yaml.safe_load(open(cfgp))
    base.setdefault("TRAIN", {})["RESUME"] = r1["folder"]
    rcfg.write_text(yaml.safe_dump(base))
    for fn in ("base.yaml",):
        (tmp_path / fn).write_text(open(os.path.join(REPO, "configs", fn)).read())
    os.makedirs(tmp_path / "modules", exist_ok=True)
    for fn in os.listdir(os.path.join(REPO, "configs", "modules")):
        (tmp_path / "modules" / fn).write_text(open(os.path.join(REPO, "configs", "modules", fn)).read())
    r2 = cli.train_main(["--cfg", str(rcfg)] + common[2:] + ["--epochs", "3"])
    assert r2["epoch"] == 2 and r2["step"] == 6 and os.path.exists(os.path.join(r2["checkpoints"], "epoch=2.ckpt"))
    # test: strict load, metrics json
    out = cli.test_main(["--cfg", cfgp, "--batch_size", "4", "--folder", str(tmp_path), "--frames", "24", "--test_batches", "2",
                         "--checkpoint", os.path.join(r2["checkpoints"], "epoch=2.ckpt")])
    assert np.isfinite(out["Metrics/MPJPE/mean"]) and os.path.exists(out["file"])


def test_ego_eval_variants(dev):
    """The config-toggled variants of ego_eval on the HIP path: POSE_ESTIMATION_TASK (interactee ground truth at the
    end of the batch -> joints_interactee_gt and the mpjpe_interactee metric, mld.py:1119-1131,1843-1866), SEE_FUTURE
    (half-length decode, :1357-1358) and classifier-free guidance with a scene token (:1144-1158)."""
    from seeme_amd.config import parse_config
    from seeme_amd.mld import MLD, SyntheticEgoDataModule
    from seeme_amd.smpl import SMPL
    cfg = parse_config(os.path.join(REPO, "configs", "config_mld_scene.yaml"))
    cfg.TEST.POSE_ESTIMATION_TASK = True
    cfg.model.guidance_scale = 2.0
    dm = SyntheticEgoDataModule(nfeats=75, T=24, n_points=512, device=dev)
    model = MLD(cfg, dm, smpl_model=SMPL.synthetic(1234)).to(dev).eval()
    batch = dm.batch(3, idx=1, with_scene=True, pose_estimation=True, lengths=[24, 17, 24])
    rs = model.ego_eval(batch)
    assert rs["joints_interactee_gt"].shape == rs["joints_interactee"].shape == (3, 24, 24, 3)
    model.EgoMetric.reset()
    model.test_step(batch)
    got = model.EgoMetric.compute()
    want = float(np.mean([np.linalg.norm((rs["joints_interactee"][b, :L] - rs["joints_interactee"][b, :L, [0]]
                                          - rs["joints_interactee_gt"][b, :L] + rs["joints_interactee_gt"][b, :L, [0]]).cpu().numpy(),
                                         axis=-1).mean() * 1000 for b, L in enumerate([24, 17, 24])]))
    assert want > 0 and abs(got["mpjpe_interactee"] - want) < 1e-3 * want
    # SEE_FUTURE: the decoded motion covers half of each length
    cfg2 = parse_config(os.path.join(REPO, "configs", "config_mld_egobody.yaml"))
    cfg2.TEST.SEE_FUTURE = True
    m2 = MLD(cfg2, dm, smpl_model=SMPL.synthetic(1234)).to(dev).eval()
    rs2 = m2.ego_eval(dm.batch(2, idx=2))
    assert rs2["m_rst"].shape[1] == 12 and rs2["lengths"] == [12, 12] and torch.isfinite(rs2["joints_rst"]).all()


@pytest.mark.parametrize("N", [1, 2])
def test_hip_backward_matches_autograd(dev, N):
    """Hand-written backward of the denoiser chain (k_den_bwd + host batch reductions) vs PyTorch autograd over the
    differentiable twin: loss, every parameter gradient and the gradient of the condition tokens."""
    from seeme_amd.denoiser_autograd import denoiser_forward_torch
    from seeme_amd.denoiser_train import denoiser_forward_hip_train
    den = make_den(dev, cond=("text", "scene", "interactee"))
    den.train()
    g = torch.Generator(device="cpu").manual_seed(11 + N)
    B = 5
    sample = torch.randn(B, 1, 256, generator=g).to(dev)
    cond = torch.randn(N, B, 256, generator=g).to(dev).requires_grad_(True)
    t = torch.tensor([3, 999, 250, 0, 600], device=dev)
    target = torch.randn(B, 1, 256, generator=g).to(dev)

    def run(fn):
        for p in den.parameters():
            p.grad = None
        cond.grad = None
        out = fn(den, sample, t, cond)
        loss = torch.nn.functional.mse_loss(out, target)
        loss.backward()
        return out.detach(), float(loss.detach()), {k: (p.grad.clone() if p.grad is not None else None) for k, p in den.named_parameters()}, cond.grad.clone()

    # training mode: the HIP path draws the keep-masks of the MD layers' dropout sites (mdiff_transformer.py:137-165,241-254;
    # cross_attention.py:264-273); the twin gets those very masks, so the comparison covers dropout forward and backward
    o_hip, l_hip, g_hip, c_hip = run(denoiser_forward_hip_train)
    masks = den._train_pack.last_masks
    assert masks is not None and masks.shape == (B, 10960) and abs(float(masks.float().mean()) - 0.9) < 0.02
    o_ref, l_ref, g_ref, c_ref = run(lambda d, s_, t_, c_: denoiser_forward_torch(d, s_, t_, c_, masks=masks.clone()))
    den.eval()                                               # eval mode: no masks are drawn, the arithmetic is the inference one
    o_ev = denoiser_forward_hip_train(den, sample, t, cond)
    assert den._train_pack.last_masks is None
    assert rel_err(o_ev.detach().cpu().numpy(), denoiser_forward_torch(den, sample, t, cond).detach().cpu().numpy()) < TOL_F32
    assert rel_err(o_ev.detach().cpu().numpy(), o_hip.cpu().numpy()) > 1e-2     # ... and dropout did change the training forward
    den.train()
    assert rel_err(o_hip.cpu().numpy(), o_ref.cpu().numpy()) < TOL_F32 and abs(l_hip - l_ref) < 1e-5 * max(1.0, abs(l_ref))
    assert rel_err(c_hip.cpu().numpy(), c_ref.cpu().numpy()) < 2e-4
    worst = ("", 0.0)
    gmax = max(float(gr.abs().max()) for gr in g_ref.values() if gr is not None)
    for k, gr in g_ref.items():
        if gr is None:
            assert g_hip[k] is None or float(g_hip[k].abs().max()) == 0.0, k
            continue
        assert g_hip[k] is not None, f"no gradient for {k}"
        if float(gr.abs().max()) < 1e-6 * gmax:
            # numerically zero in the reference too (one condition token: the query path has no influence, its exact
            # gradient is 0 and autograd returns rounding noise) -- only require the same order of nothing
            assert float(g_hip[k].abs().max()) < 1e-5 * gmax, k
            continue
        e = rel_err(g_hip[k].cpu().numpy(), gr.cpu().numpy())
        if e > 1e-3:
            print(f"   grad mismatch {k}: rel err {e:.3e}")
        if e > worst[1]:
            worst = (k, e)
    print(f"HIP backward vs autograd (N={N}): worst parameter-gradient rel err {worst[1]:.3e} at {worst[0]}")
    assert worst[1] < 5e-4, worst
    # gradient accumulation: a second backward without zero_grad adds to the first (the .grad views are still in use);
    # in eval mode, so that both passes are the same function
    den.eval()
    _, _, g_one, _ = run(denoiser_forward_hip_train)
    out = denoiser_forward_hip_train(den, sample, t, cond)
    torch.nn.functional.mse_loss(out, target).backward()
    k = "encoder.middle_block.sa_block.linear1.weight"
    acc = dict(den.named_parameters())[k].grad
    assert rel_err(acc.cpu().numpy(), 2 * g_one[k].cpu().numpy()) < 1e-5
    kin = "encoder.middle_block.sa_block.self_attn.in_proj_weight"          # shared by the chain and the tables
    assert rel_err(dict(den.named_parameters())[kin].grad.cpu().numpy(), 2 * g_one[kin].cpu().numpy()) < 1e-5


def test_vae_autograd_twin_matches_hip_and_stage1_trains(dev):
    """Stage-1 twins (seeme_amd/vae_autograd.py): same numbers as the HIP VAE / SMPL forward, and a stage-1 training step
    (recons + joints + KL losses, mld/models/losses/mld.py:113-156) produces finite gradients and a decreasing loss."""
    from seeme_amd.vae_autograd import vae_encode_torch, vae_decode_torch, smpl_joints_torch
    from seeme_amd.config import parse_config
    from seeme_amd.mld import MLD, SyntheticEgoDataModule
    from seeme_amd.smpl import SMPL
    vae = make_vae(75, dev)
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(3, 24, 75, generator=g).to(dev)
    lengths = [24, 17, 24]
    with torch.no_grad():
        mu_t, std_t = vae_encode_torch(vae, x, lengths)
        dist = vae.encode_dist(x, lengths)
        assert rel_err(mu_t.cpu().numpy(), dist[0:1].cpu().numpy()) < TOL_F32
        z = mu_t
        assert rel_err(vae_decode_torch(vae, z, lengths).cpu().numpy(), vae.decode(z, lengths).cpu().numpy()) < TOL_F32
        smpl = SMPL.synthetic(1234).to(dev)
        betas, pose, tr = torch.randn(7, 10, generator=g).to(dev) * 0.5, torch.randn(7, 72, generator=g).to(dev) * 0.4, torch.randn(7, 3, generator=g).to(dev)
        j_hip = smpl(betas=betas, body_pose=pose[:, 3:], global_orient=pose[:, :3], transl=tr, return_verts=False).joints[:, :24]
        assert rel_err(smpl_joints_torch(smpl, betas, pose, tr).cpu().numpy(), j_hip.cpu().numpy()) < TOL_F32
    cfg = parse_config(os.path.join(REPO, "configs", "config_vae_egobody.yaml"))
    cfg.TRAIN.OPTIM.LR = 1e-4
    torch.manual_seed(1234)
    dm = SyntheticEgoDataModule(nfeats=75, T=24, device=dev)
    model = MLD(cfg, dm, smpl_model=SMPL.synthetic(1234)).to(dev).train()
    batch = dm.batch(4, idx=0)
    losses = []
    for _ in range(60):
        torch.manual_seed(99)                  # the same rsample noise and dropout masks every step: a fixed objective
        loss = model.training_step(batch)
        model.optimizer_step(loss)
        losses.append(float(loss.detach()))
    # measured on MI355X: 0.7171 -> 0.6953, the pelvis-translation term (0.42 of it; values and weights of every term are
    # pinned against the oracle in tests/test_gpu_flows.py) falling slowest
    assert all(np.isfinite(losses)) and losses[-1] < 0.985 * losses[0] and np.mean(losses[-10:]) < np.mean(losses[:10]), losses
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in model.vae.named_parameters() if "query_pos" not in n)


def test_smpl_joints_backward_matches_autograd(dev):
    """seeme_smpl_joints_backward (hand-written: world rotations, leaf-to-root accumulation, d Rodrigues / d axis-angle) against
    autograd through the PyTorch twin of smplx's lbs (vae_autograd.smpl_joints_torch): forward joints and the gradients w.r.t.
    pose and translation for random poses, a zero pose (the 1e-8 regularised angle) and a padded (GIMO-style) pose."""
    from seeme_amd.smpl import SMPL, smpl_joints_hip
    from seeme_amd.vae_autograd import smpl_joints_torch
    smpl = SMPL.synthetic(1234).to(dev)
    g = torch.Generator().manual_seed(8)
    M = 37
    betas = (torch.randn(M, 10, generator=g) * 0.5).to(dev)
    pose = (torch.randn(M, 72, generator=g) * 0.5).to(dev)
    pose[3] = 0.0
    pose[5, 66:] = 0.0
    tr = torch.randn(M, 3, generator=g).to(dev)
    wgt = torch.randn(M, 24, 3, generator=g).to(dev)
    res = []
    for fn in (smpl_joints_hip, smpl_joints_torch):
        p, t = pose.clone().requires_grad_(True), tr.clone().requires_grad_(True)
        j = fn(smpl, betas, p, t)
        (j * wgt).sum().backward()
        res.append((j.detach(), p.grad.clone(), t.grad.clone()))
    assert rel_err(res[0][0].cpu().numpy(), res[1][0].cpu().numpy()) < TOL_F32
    assert rel_err(res[0][1].cpu().numpy(), res[1][1].cpu().numpy()) < 1e-4
    assert rel_err(res[0][2].cpu().numpy(), res[1][2].cpu().numpy()) < 1e-5
    p = pose.clone().requires_grad_(True)                   # without a translation
    smpl_joints_hip(smpl, betas, p, None).mul(wgt).sum().backward()
    assert rel_err(p.grad.cpu().numpy(), res[1][1].cpu().numpy()) < 1e-4


# ----------------------------------------------------------------------------- one sample split over C CUs (k_den_cluster)
def _with_cluster(den, Cc, place=0, flags=0):
    den.cluster, den.cluster_placement, den.cluster_flags = Cc, place, flags
    return den


@pytest.mark.parametrize("Cc", [8, 4, 2])
def test_cluster_sampler_vs_reference_fixtures(dev, Cc):
    """k_den_cluster (csrc/den_cluster.inc.hip) with fp32 weights against the fixtures generated from the reference modules:
    one forward (scalar and per-sample timesteps, mld_denoiser.py:151-244) and the 50-step DDIM loop (mld.py:467-497)."""
    g = load_golden("denoiser_N1.npz")
    den = _with_cluster(make_den(dev), Cc)
    s, c = torch.from_numpy(g["sample"]).to(dev), torch.from_numpy(g["cond"]).to(dev)
    for t in (981, 501, 1):
        y = den(sample=s, timestep=torch.tensor(t), encoder_hidden_states=c)[0]
        assert rel_err(y.cpu().numpy(), g[f"out_t{t}"]) < TOL_F32 and elem_err(y.cpu().numpy(), g[f"out_t{t}"]) < TOL_F32
    y = den(sample=s, timestep=torch.from_numpy(g["tvec"]).to(dev), encoder_hidden_states=c)[0]
    assert rel_err(y.cpu().numpy(), g["out_tvec"]) < TOL_F32 and elem_err(y.cpu().numpy(), g["out_tvec"]) < TOL_F32
    gl = load_golden("ddim50_N1_B3.npz")
    sch = _sched()
    sch.set_timesteps(int(gl["steps"]))
    out = den.sample_loop(torch.from_numpy(gl["latents"]).to(dev), torch.from_numpy(gl["cond_bf"]).to(dev), sch)
    assert rel_err(out.cpu().numpy(), gl["out"]) < 5e-4
    assert den.cluster_status()[0] == 0                       # no cluster gave up waiting for a peer


@pytest.mark.parametrize("wd,tol", [("fp32", 1e-5), ("fp16", 5e-4), ("bf16", 5e-3)])
def test_cluster_sampler_equals_one_cu_kernel(dev, wd, tol):
    """Every cluster size, placement (one XCD / C XCDs) and granule-store flavour (L2-local / write-through) computes the same
    latents as the one-CU-per-sample kernel on the same weight image dtype (contraction order differs: fp32 rounding), twice
    bit-identically (the replicated epilogues of a cluster must agree bit for bit, and the sum over the publishers has a fixed
    order); ragged batches (clusters beyond B exit), DDPM with injected step noise, per-sample timesteps."""
    den = make_den(dev, weight_dtype=wd)
    sch = _sched()
    sch.set_timesteps(50)
    torch.manual_seed(5)
    lat, cond = torch.randn(32, 1, 256, device=dev), torch.randn(32, 1, 256, device=dev)
    base = _with_cluster(den, 0).sample_loop(lat, cond, sch)
    for Cc in (8, 4, 2):
        for place in (0, 1):
            for flags in (0, 1):
                _with_cluster(den, Cc, place, flags)
                z, z2 = den.sample_loop(lat, cond, sch), den.sample_loop(lat, cond, sch)
                code, local = den.cluster_status()
                assert code == 0
                assert torch.equal(z, z2), (Cc, place, flags)
                assert rel_err(z.cpu().numpy(), base.cpu().numpy()) < tol, (Cc, place, flags)
                if flags == 1 or place == 1:
                    assert local == 0                        # write-through requested, or the cluster spans XCDs
    # ragged batch
    z5 = _with_cluster(den, 8).sample_loop(lat[:5].contiguous(), cond[:5].contiguous(), sch)
    assert rel_err(z5.cpu().numpy(), base[:, :5].cpu().numpy()) < tol
    # DDPM ancestral steps with injected noise
    schp = _sched("ddpm")
    schp.set_timesteps(1000)
    schp.timesteps = schp.timesteps[:40]
    noise = torch.randn(40, 32, 256, device=dev)
    bp = _with_cluster(den, 0).sample_loop(lat, cond, schp, step_noise=noise)
    zp = _with_cluster(den, 8).sample_loop(lat, cond, schp, step_noise=noise)
    assert rel_err(zp.cpu().numpy(), bp.cpu().numpy()) < tol
    # one forward with per-sample timesteps
    t = torch.randint(0, 1000, (32,), device=dev)
    y0 = _with_cluster(den, 0)(sample=lat, timestep=t, encoder_hidden_states=cond.permute(1, 0, 2))[0]
    y8 = _with_cluster(den, 8)(sample=lat, timestep=t, encoder_hidden_states=cond.permute(1, 0, 2))[0]
    assert rel_err(y8.cpu().numpy(), y0.cpu().numpy()) < tol


def test_cluster_sampler_under_uneven_load(dev):
    """The in-launch exchanges (data-tagged granules, cdna_hip_programming.md Guideline 16 R2) under UNEVEN load: a second stream
    keeps part of the chip busy with HBM-bound copies and small GEMMs while clusters sample; results must be bit-identical to
    the quiet run for both store flavours, and no cluster may give up."""
    den = make_den(dev, weight_dtype="fp16")
    sch = _sched()
    sch.set_timesteps(50)
    torch.manual_seed(11)
    lat, cond = torch.randn(24, 1, 256, device=dev), torch.randn(24, 1, 256, device=dev)
    side = torch.cuda.Stream()
    big = torch.randn(64 * 1024 * 1024, device=dev)
    a = torch.randn(1024, 1024, device=dev)
    for flags in (0, 1):
        _with_cluster(den, 8, 0, flags)
        quiet = den.sample_loop(lat, cond, sch)
        torch.cuda.synchronize()
        for rep in range(3):
            with torch.cuda.stream(side):
                for _ in range(6):
                    big.mul_(1.0001)
                    a = (a @ a).clamp_(-1, 1)
            z = den.sample_loop(lat, cond, sch)
            torch.cuda.synchronize()
            assert den.cluster_status()[0] == 0
            assert torch.equal(z, quiet), (flags, rep)


def test_auto_cluster_policy(dev):
    """cluster='auto': 8 / 4 / 2 CUs per sample while B x C fits one workgroup per CU, the one-CU kernel beyond, for CFG pairs,
    more than two condition tokens and several heads."""
    den = make_den(dev)
    assert [den._cluster_size(B, 1, False) for B in (1, 32, 33, 64, 65, 128, 129, 256)] == [8, 8, 4, 4, 2, 2, 0, 0]
    assert den._cluster_size(32, 2, False) == 8 and den._cluster_size(32, 3, False) == 0 and den._cluster_size(32, 1, True) == 0
    # a device with fewer CUs (partitioned / masked): every workgroup of a cluster launch must still be resident at once
    assert [den._cluster_size(32, 1, False, cus=c) for c in (256, 128, 64, 32)] == [8, 4, 2, 0]
    # the plan: (CUs per cluster, samples per cluster).  fp32 image: one sample per cluster as above
    assert [den._cluster_plan(B, 1, False, False) for B in (32, 64, 65, 128, 129)] == [(8, 1), (4, 1), (2, 1), (2, 1), (0, 1)]
    # fp16 image, above 64 samples: 64 clusters of 4 CUs with ceil(B / 64) samples each, up to B = 512
    d16 = make_den(dev, weight_dtype="fp16")
    assert [d16._cluster_plan(B, 1, False, False) for B in (32, 33, 64, 65, 128, 129, 256, 257, 512, 513)] == \
        [(8, 1), (4, 1), (4, 1), (4, 2), (4, 2), (4, 3), (4, 4), (4, 5), (4, 8), (0, 1)]
    # ... with two condition tokens up to six samples per cluster; not for CFG pairs, per-sample timesteps, or when switched off
    assert [d16._cluster_plan(B, 2, False, False) for B in (64, 128, 384, 385)] == [(4, 1), (4, 2), (4, 6), (0, 1)]
    assert d16._cluster_plan(128, 3, False, False) == (0, 1) and d16._cluster_plan(128, 1, True, False) == (0, 1)
    assert d16._cluster_plan(128, 1, False, True) == (2, 1)
    d16.cluster_ms = False
    assert d16._cluster_plan(128, 1, False, False) == (2, 1)
    # the bf16 image: four A rows per sample, up to four samples per cluster (B <= 256)
    db = make_den(dev, weight_dtype="bf16")
    assert [db._cluster_plan(B, 1, False, False) for B in (64, 65, 256, 257)] == [(4, 1), (4, 2), (4, 4), (0, 1)]


@pytest.mark.parametrize("B,sched", [(50, "ddim8"), (100, "ddim"), (130, "ddpm"), (512, "ddim")])
def test_cluster_ms_equals_one_sample_cluster(dev, B, sched):
    """k_den_cluster_ms (csrc/den_cluster_ms.inc.hip; batches above 64: a cluster of 4 CUs owns up to 8 samples, two MFMA A rows each)
    against k_den_cluster with the same C on the same samples, 256 / C at a time: the same weight image, the same exchanges, the same order
    of every addition -- bit-identical, for 8-CU clusters with two samples (B = 50), ragged last clusters (B = 100: 50 clusters of 2; B = 130: 3 per cluster, the last one 1),
    the full 8 samples per cluster (B = 512), DDIM and DDPM with injected step noise (mld.py:467-497); and within fp16 rounding of the
    one-CU-per-sample kernel, whose image keeps the skip linears unfolded."""
    den = make_den(dev, weight_dtype="fp16")
    steps = 12
    forced8 = sched == "ddim8"                                   # (8 CUs x 2 samples: built, not what the policy picks)
    if forced8:
        sched = "ddim"
        os.environ["SEEME_DEN_CLUSTER_MS_PLAN"] = "8,2"
    sch = _sched(sched)
    sch.set_timesteps(1000 if sched == "ddpm" else 50)
    sch.timesteps = sch.timesteps[:steps]
    torch.manual_seed(B)
    lat, cond = torch.randn(B, 1, 256, device=dev), torch.randn(B, 1, 256, device=dev)
    noise = torch.randn(steps, B, 256, device=dev) if sched == "ddpm" else None
    Cc, spc = den._cluster_plan(B, 1, False, False)
    assert (Cc, spc) == ((8, 2) if forced8 else (4, -(-B // 64)))
    z = den.sample_loop(lat, cond, sch, step_noise=noise)
    z2 = den.sample_loop(lat, cond, sch, step_noise=noise)
    torch.cuda.synchronize()
    os.environ.pop("SEEME_DEN_CLUSTER_MS_PLAN", None)
    assert den.cluster_status()[0] == 0
    assert torch.equal(z, z2)
    den.cluster_ms = False
    _with_cluster(den, Cc, 1)
    ch = 256 // Cc
    ref = torch.cat([den.sample_loop(lat[i:i + ch].contiguous(), cond[i:i + ch].contiguous(), sch,
                                     step_noise=None if noise is None else noise[:, i:i + ch].contiguous()) for i in range(0, B, ch)], 1)
    torch.cuda.synchronize()
    assert torch.equal(z, ref), float((z - ref).abs().max())
    _with_cluster(den, 0)
    one = den.sample_loop(lat, cond, sch, step_noise=noise)
    assert rel_err(z.cpu().numpy(), one.cpu().numpy()) < 1e-3


@pytest.mark.parametrize("B,sched", [(100, "ddim"), (330, "ddpm")])
def test_cluster_ms_two_condition_tokens(dev, B, sched):
    """k_den_cluster_ms with scene + interactee (N = 2, the shipped config_mld_egobody.yaml:114): the ca_block's query / proj_out stages and
    the third exchange per sample (mdiff_transformer.py:219-239), bit-identical to k_den_cluster<.., 4, Q> on the same samples 64 at a time
    and within fp16 rounding of the one-CU kernel."""
    den = make_den(dev, cond=("text", "scene", "interactee"), weight_dtype="fp16")
    steps = 10
    sch = _sched(sched)
    sch.set_timesteps(1000 if sched == "ddpm" else 50)
    sch.timesteps = sch.timesteps[:steps]
    torch.manual_seed(B)
    lat, cond = torch.randn(B, 1, 256, device=dev), torch.randn(B, 2, 256, device=dev)
    noise = torch.randn(steps, B, 256, device=dev) if sched == "ddpm" else None
    assert den._cluster_plan(B, 2, False, False) == (4, -(-B // 64))
    z = den.sample_loop(lat, cond, sch, step_noise=noise)
    torch.cuda.synchronize()
    assert den.cluster_status()[0] == 0
    den.cluster_ms = False
    _with_cluster(den, 4, 1)
    ref = torch.cat([den.sample_loop(lat[i:i + 64].contiguous(), cond[i:i + 64].contiguous(), sch,
                                     step_noise=None if noise is None else noise[:, i:i + 64].contiguous()) for i in range(0, B, 64)], 1)
    torch.cuda.synchronize()
    assert torch.equal(z, ref), float((z - ref).abs().max())
    _with_cluster(den, 0)
    one = den.sample_loop(lat, cond, sch, step_noise=noise)
    assert rel_err(z.cpu().numpy(), one.cpu().numpy()) < 1e-3


@pytest.mark.parametrize("ntok,wd", [(1, "fp16"), (2, "fp16"), (1, "bf16"), (2, "bf16")])
def test_cluster_ms_random_shapes(dev, ntok, wd):
    """A short soak of k_den_cluster_ms (scripts/cluster_ms_soak.py runs thousands): random batch sizes 65 .. 512 (384 with two condition
    tokens, 256 with the bf16 image: four A rows per sample), 3-6 steps, DDIM or DDPM with injected noise, two denoiser objects with their own buffers alternating -- every result
    bit-identical to k_den_cluster with four CUs per sample on the same samples, and no cluster gives up."""
    conds = ("text", "scene", "interactee") if ntok == 2 else ("text", "interactee")
    dens = [make_den(dev, cond=conds, weight_dtype=wd), make_den(dev, cond=conds, weight_dtype=wd)]
    ref = _with_cluster(make_den(dev, cond=conds, weight_dtype=wd), 4, 1)
    ref.cluster_ms = False
    rng = np.random.default_rng(77 + ntok)
    for it in range(24):
        B = int(rng.integers(65, 257 if wd == "bf16" else (513 if ntok == 1 else 385)))
        steps = int(rng.integers(3, 7))
        kind = "ddpm" if rng.random() < 0.4 else "ddim"
        sch = _sched(kind)
        sch.set_timesteps(1000 if kind == "ddpm" else 50)
        sch.timesteps = sch.timesteps[:steps]
        g = torch.Generator(device="cpu").manual_seed(500 + it)
        lat, cond = torch.randn(B, 1, 256, generator=g).to(dev), torch.randn(B, ntok, 256, generator=g).to(dev)
        noise = torch.randn(steps, B, 256, generator=g).to(dev) if kind == "ddpm" else None
        den = dens[it % 2]
        assert den._cluster_plan(B, ntok, False, False) == (4, -(-B // 64))
        z = den.sample_loop(lat, cond, sch, step_noise=noise)
        r = torch.cat([ref.sample_loop(lat[i:i + 64].contiguous(), cond[i:i + 64].contiguous(), sch,
                                       step_noise=None if noise is None else noise[:, i:i + 64].contiguous()) for i in range(0, B, 64)], 1)
        torch.cuda.synchronize()
        assert den.cluster_status()[0] == 0 and ref.cluster_status()[0] == 0
        assert torch.equal(z, r), (it, B, steps, kind, float((z - r).abs().max()))


def test_single_forward_stays_off_the_large_batch_kernel(dev):
    """MldDenoiser.forward (one step, no scheduler; mld_denoiser.py:151-244) hands the kernel one table row PER SAMPLE (scalar and vector
    timesteps alike), which k_den_cluster_ms does not take: at a batch size whose sampling loop runs on it, forward() stays on the
    one-sample kernels and equals their result on the same samples in smaller batches."""
    den = make_den(dev, weight_dtype="fp16")
    torch.manual_seed(21)
    B = 150
    x, c = torch.randn(B, 1, 256, device=dev), torch.randn(1, B, 256, device=dev)
    assert den._cluster_plan(B, 1, False, False) == (4, 3) and den._cluster_plan(B, 1, False, True) == (0, 1)
    tv = torch.randint(0, 1000, (B,), device=dev)
    for t in (torch.tensor(417), tv):
        den.cluster = "auto"
        y = den(sample=x, timestep=t, encoder_hidden_states=c)[0]
        den.cluster = 0
        ref = torch.cat([den(sample=x[i:i + 64].contiguous(), timestep=t if t.dim() == 0 else t[i:i + 64].contiguous(),
                             encoder_hidden_states=c[:, i:i + 64].contiguous())[0] for i in range(0, B, 64)], 0)
        assert torch.equal(y, ref), float((y - ref).abs().max())


@pytest.mark.parametrize("Cc", [8, 4, 2])
def test_cluster_sampler_two_condition_tokens(dev, Cc):
    """k_den_cluster with scene + interactee (N = 2: the reference's shipped config_mld_egobody.yaml:114): the ca_block keeps its query
    and proj_out stages (mdiff_transformer.py:219-239, 152-163) -- query column-split by dims, the softmax over head_dim and the per-token
    products combined across the cluster like a split softmax (third exchange per layer), proj_out replicated.  fp32 weights against the
    fixture generated from the reference module (denoiser_N2.npz); every weight dtype against the one-CU kernel, both placements, twice
    bit-identically."""
    g = load_golden("denoiser_N2.npz")
    s, c = torch.from_numpy(g["sample"]).to(dev), torch.from_numpy(g["cond"]).to(dev)
    sch = _sched()
    sch.set_timesteps(50)
    torch.manual_seed(3)
    lat, cond = torch.randn(32, 1, 256, device=dev), torch.randn(32, 2, 256, device=dev)
    for wd, tol in (("fp32", 1e-5), ("fp16", 8e-4), ("bf16", 8e-3)):
        den = make_den(dev, cond=("text", "scene", "interactee"), weight_dtype=wd)
        base = _with_cluster(den, 0).sample_loop(lat, cond, sch)
        _with_cluster(den, Cc)
        if wd == "fp32":
            for t in (981, 501, 1):
                y = den(sample=s, timestep=torch.tensor(t), encoder_hidden_states=c)[0]
                assert rel_err(y.cpu().numpy(), g[f"out_t{t}"]) < TOL_F32 and elem_err(y.cpu().numpy(), g[f"out_t{t}"]) < TOL_F32
            y = den(sample=s, timestep=torch.from_numpy(g["tvec"]).to(dev), encoder_hidden_states=c)[0]
            assert rel_err(y.cpu().numpy(), g["out_tvec"]) < TOL_F32
        for place in (0, 1):
            _with_cluster(den, Cc, place)
            z, z2 = den.sample_loop(lat, cond, sch), den.sample_loop(lat, cond, sch)
            assert den.cluster_status()[0] == 0 and torch.equal(z, z2)
            assert rel_err(z.cpu().numpy(), base.cpu().numpy()) < tol, (wd, place)
