"""Pin the numpy oracle against fixtures produced by the reference's own PyTorch modules
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from oracle import mld_oracle as O
from seeme_amd.weights_recipe import recipe_state_dict
from seeme_amd import shapes

TOL = 2e-5  # fp32 numpy vs fp32 torch CPU, same algorithm, different summation order


def vae_params(F):
    return recipe_state_dict(shapes.vae_shapes(F))


def den_params():
    return recipe_state_dict(shapes.denoiser_shapes())


@pytest.mark.parametrize("name,F", [("vae_F132_T24.npz", 132), ("vae_F75_T60.npz", 75), ("vae_F132_T196.npz", 132)])
def test_vae_encode_decode(name, F):
    g = load_golden(name)
    P = vae_params(F)
    mu, std = O.vae_encode(P, g["features"], g["lengths"].tolist())
    assert mu.shape == g["mu"].shape
    assert rel_err(mu, g["mu"]) < TOL
    assert rel_err(std, g["std"]) < TOL
    dec = O.vae_decode(P, g["mu"], g["lengths"].tolist())
    assert dec.shape == g["decoded"].shape
    assert rel_err(dec, g["decoded"]) < TOL


def test_vae_fp64_is_closer_than_tolerance():
    g = load_golden("vae_F132_T24.npz")
    P = O.cast_params(vae_params(132), np.float64)
    mu, _ = O.vae_encode(P, g["features"].astype(np.float64), g["lengths"].tolist())
    assert rel_err(mu, g["mu"]) < TOL


@pytest.mark.parametrize("N", [1, 2])
def test_denoiser_forward(N):
    g = load_golden(f"denoiser_N{N}.npz")
    P = den_params()
    for t in (981, 501, 1):
        y = O.denoiser_forward(P, g["sample"], t, g["cond"])
        assert rel_err(y, g[f"out_t{t}"]) < TOL
    y = O.denoiser_forward(P, g["sample"], g["tvec"], g["cond"])
    assert rel_err(y, g["out_tvec"]) < TOL


def test_ddim_loop_against_reference_denoiser():
    g = load_golden("ddim50_N1_B3.npz")
    out = O.diffusion_reverse(den_params(), g["cond_bf"], g["latents"], int(g["steps"]))
    assert out.shape == g["out"].shape
    assert rel_err(out, g["out"]) < 2e-4  # 50 chained steps


def test_ddim_loop_cfg():
    g = load_golden("ddim10_N2_B2_cfg.npz")
    out = O.diffusion_reverse(den_params(), g["cond_bf"], g["latents"], int(g["steps"]),
                              guidance_scale=float(g["guidance_scale"]))
    assert rel_err(out, g["out"]) < 2e-4


# ----------------------------------------------------------------------------- the torch-on-CPU edition of the oracle
def test_torch_oracle_matches_reference_fixtures():
    """oracle/mld_oracle_torch.py (the cpu_baseline of bench.py) against the same reference-generated fixtures."""
    import torch
    from oracle import mld_oracle_torch as OT
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    for name, F in (("vae_F132_T24.npz", 132), ("vae_F75_T60.npz", 75)):
        g = load_golden(name)
        P = OT.to_torch(vae_params(F))
        mu, std = OT.vae_encode(P, tt(g["features"]), g["lengths"].tolist())
        assert rel_err(mu.numpy(), g["mu"]) < TOL and rel_err(std.numpy(), g["std"]) < TOL
        assert rel_err(OT.vae_decode(P, tt(g["mu"]), g["lengths"].tolist()).numpy(), g["decoded"]) < TOL
    Pd = OT.to_torch(den_params())
    for N in (1, 2):
        g = load_golden(f"denoiser_N{N}.npz")
        for t in (981, 1):
            assert rel_err(OT.denoiser_forward(Pd, tt(g["sample"]), t, tt(g["cond"])).numpy(), g[f"out_t{t}"]) < TOL
        assert rel_err(OT.denoiser_forward(Pd, tt(g["sample"]), g["tvec"], tt(g["cond"])).numpy(), g["out_tvec"]) < TOL
    g = load_golden("ddim50_N1_B3.npz")
    out = OT.diffusion_reverse(Pd, tt(g["cond_bf"]), tt(g["latents"]), int(g["steps"]))
    assert out.shape == g["out"].shape and rel_err(out.numpy(), g["out"]) < 2e-4
    g = load_golden("ddim10_N2_B2_cfg.npz")
    out = OT.diffusion_reverse(Pd, tt(g["cond_bf"]), tt(g["latents"]), int(g["steps"]), guidance_scale=float(g["guidance_scale"]))
    assert rel_err(out.numpy(), g["out"]) < 2e-4


def test_misc():
    g = load_golden("misc.npz")
    # sin/cos of arguments up to 999: one fp32 ulp of the frequency is 6e-5 in the argument
    assert rel_err(O.timestep_features(g["t"]), g["timestep_features"]) < 1e-4
    assert rel_err(O.aa_to_quat(g["aa"]), g["aa_to_quat"]) < 1e-6
    assert rel_err(O.aa_to_rotmat(g["aa"]), g["aa_to_rotmat"]) < 1e-6
    assert rel_err(O.rot6d_to_rotmat(g["rot6d"]), g["rot6d_to_rotmat"]) < 1e-6
    P = recipe_state_dict(shapes.pointnet_shapes())
    assert rel_err(O.pointnet_forward(P, g["points"]), g["pointnet"]) < TOL


def test_quaternion_matrix_known_answers():
    # the only known-answer vectors in the reference tree (mld/utils/geometry2.py:10-18 docstring)
    R = O.quat_to_rotmat(np.array([[1.0, 0, 0, 0], [0, 1.0, 0, 0]]))
    assert np.allclose(R[0], np.eye(3))
    assert np.allclose(R[1], np.diag([1, -1, -1]))


def test_scheduler_properties():
    acp = O.alphas_cumprod(O.make_betas())
    ts = O.ddim_timesteps(50)
    assert ts[0] == 981 and ts[-1] == 1 and len(ts) == 50
    # x0-consistency: if eps is the true noise, one DDIM step lands on the t_prev marginal of the same x0
    rng = np.random.default_rng(0)
    x0 = rng.standard_normal((2, 1, 256)).astype(np.float32)
    eps = rng.standard_normal((2, 1, 256)).astype(np.float32)
    xt = O.ddpm_add_noise(acp, x0, eps, np.array([981, 981]))
    xp = O.ddim_step(acp, eps, 981, xt, 50)
    assert rel_err(xp, O.ddpm_add_noise(acp, x0, eps, np.array([961, 961]))) < 1e-5
    # DDPM posterior mean at t=0 returns x0 exactly when eps is true
    x1 = O.ddpm_add_noise(acp, x0, eps, np.array([0, 0]))
    assert rel_err(O.ddpm_step(acp, eps, 0, x1), x0) < 1e-4


def test_smpl_lbs_properties():
    m = O.make_synthetic_smpl(V=512)
    ids = O.SMPL_EXTRA_VERTEX_IDS % 512
    rng = np.random.default_rng(1)
    M = 3
    betas = rng.standard_normal((M, 10)).astype(np.float32) * 0.5
    zero = np.zeros((M, 69), np.float32)
    go = np.zeros((M, 3), np.float32)
    old = O.SMPL_EXTRA_VERTEX_IDS
    try:
        O.SMPL_EXTRA_VERTEX_IDS = ids
        j, v = O.smpl_lbs(m, betas, go, zero)
        # zero pose: vertices = shaped template, joints = regressor @ shaped
        vs = m["v_template"][None] + np.einsum("bl,mkl->bmk", betas, m["shapedirs"])
        assert rel_err(v, vs) < 1e-5
        assert rel_err(j[:, :24], np.einsum("bik,ji->bjk", vs, m["J_regressor"])) < 1e-5
        assert j.shape == (M, 45, 3)
        # global rotation about the pelvis + translation is a rigid motion of the zero-pose mesh
        go2 = rng.standard_normal((M, 3)).astype(np.float32)
        tr = rng.standard_normal((M, 3)).astype(np.float32)
        j2, v2 = O.smpl_lbs(m, betas, go2, zero, tr)
        R = O.rodrigues(go2)
        piv = j[:, :1]
        assert rel_err(v2, np.einsum("bij,bvj->bvi", R, v - piv) + piv + tr[:, None]) < 1e-4
        # joints-only path agrees with the full path
        j3, _ = O.smpl_lbs(m, betas, go2, rng.standard_normal((M, 69)).astype(np.float32) * 0.3, tr, return_verts=False)
        j4, _ = O.smpl_lbs(m, betas, go2, rng.standard_normal((M, 69)).astype(np.float32) * 0.3, tr, return_verts=True)
    finally:
        O.SMPL_EXTRA_VERTEX_IDS = old
    assert j3.shape == j4.shape
