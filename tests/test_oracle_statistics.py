"""Independent evidence for the oracle's float chain where no reference-produced vector exists
(DESIGN.md §3): the PUBLISHED property of the construction.  Sparse-convolution noise with the
squared-exponential splatting kernel, normalised by sqrt(sparseConvNoiseVariance) (GPF.cpp:741-760),
is a unit-variance field whose two-point covariance is the SE kernel:

    Var[(f(p) - mean(p)) / sigma] = 1,      Cov[...] (p, p + d) = exp(-|d|^2 / (2 l^2))

(l = "lengthScale" in world units; with "aniso" a the distance is measured in p/a).  A wrong
normalisation constant, a wrong l_conv = l*sqrt(2)/2 (GPF.cpp:654-679), a wrong kernel matrix
(GPF.cpp:774-802), or impulses drawn from the wrong cells all show up here.  Realisations are
independent across paths when single_realization = false (computeSeed, SCN.cpp:40-49), so ONE batch
of queries with distinct pixel words samples the ensemble.  Checked in the three sampling spaces:
world, isotropic-ray (the headline config's space) and 1D along the ray."""
import numpy as np
import pytest

N = 40000          # realisations per estimate: standard error of a correlation estimate <= 1/sqrt(N) = 0.005
TOL = 0.03         # 6 standard errors (the field is close to, not exactly, Gaussian: rho impulses per cell)


def _ensemble(pkg, orc, pts, direction):
    """value at every point of `pts` for N independent realisations: array (len(pts), N) of (f - mean)/sigma"""
    out = []
    for p in pts:
        q = np.zeros(N, dtype=pkg.QUERY)
        q["p"] = np.asarray(p, dtype=np.float32)
        q["dir"] = np.asarray(direction, dtype=np.float32)
        q["pixel"][:, 0] = np.arange(N) % 4096          # the path identity = the realisation (SCN.cpp:40-49)
        q["pixel"][:, 1] = np.arange(N) // 4096
        q["spp"] = 3
        q["segment"] = 0
        q["scene_seed"] = 0xBA5EBA11
        v, _ = orc.eval_value(q)
        out.append(v.astype(np.float64))
    return np.array(out)


def _params(pkg, space, rho, aniso=(1, 1, 1)):
    p = pkg.params_for_config("C0")
    p["single_realization"] = 0
    p["correlation_context"] = pkg.CTX.NONE
    p["impulse_density"] = rho
    p["aniso"] = aniso
    p["mean"]["type"] = pkg.MEAN_TYPE.HOMOGENEOUS      # mean = offset: subtracting it is exact
    p["mean"]["offset"] = 0.25
    if space == "iso_ray":
        p["isotropic_3d_sampling"] = 1
    elif space == "1d":
        p["isotropic_3d_sampling"] = 1
        p["sampling_1d"] = 1
    return p


@pytest.mark.parametrize("space,rho", [("world", 8), ("world", 32), ("iso_ray", 32), ("1d", 32)])
def test_unit_variance_and_se_covariance(pkg, ob, space, rho):
    params = _params(pkg, space, rho)
    orc = ob.Oracle(params, threads=8)
    sigma, l = float(params["sigma"]), float(params["length_scale"])
    d = np.array([0.0, 0.0, 1.0])                        # ray direction; 1D noise only varies along it
    base = np.array([0.31, -0.17, 0.23])
    seps = [0.0, 0.5 * l, 1.0 * l, 2.0 * l]
    pts = [base + s * d for s in seps]
    if space != "1d":                                     # 3D fields: also an off-axis separation
        pts.append(base + 1.0 * l * np.array([0.6, 0.8, 0.0]))
        seps.append(1.0 * l)
    vals = (_ensemble(pkg, orc, pts, d) - 0.25) / sigma
    assert abs(vals.mean()) < TOL
    var = vals.var(axis=1)
    assert np.all(np.abs(var - 1.0) < 2 * TOL), var       # variance estimates carry kurtosis: twice the tolerance
    for k in range(1, len(pts)):
        corr = float(np.mean(vals[0] * vals[k]) / np.sqrt(var[0] * var[k]))
        want = float(np.exp(-seps[k] ** 2 / (2 * l * l)))
        assert abs(corr - want) < TOL, (space, rho, seps[k] / l, corr, want)


def test_anisotropic_kernel_covariance(pkg, ob):
    """aniso = (a_x, a_y, a_z) stretches the length scale per axis (GPF.cpp:659-666): the covariance along axis k at
    separation s is exp(-s^2 / (2 (l a_k)^2)) — in world space and, through the whitening transform, in
    isotropic-ray space."""
    aniso = (1.0, 2.0, 0.5)
    for space in ("world", "iso_ray"):
        params = _params(pkg, space, 16, aniso)
        orc = ob.Oracle(params, threads=8)
        sigma, l = float(params["sigma"]), float(params["length_scale"])
        base = np.array([0.11, 0.42, -0.3])
        s = 0.8 * l
        pts = [base] + [base + s * np.eye(3)[k] for k in range(3)]
        vals = (_ensemble(pkg, orc, pts, (0.0, 0.6, 0.8)) - 0.25) / sigma
        var = vals.var(axis=1)
        assert np.all(np.abs(var - 1.0) < 2 * TOL), (space, var)
        for k in range(3):
            corr = float(np.mean(vals[0] * vals[k + 1]) / np.sqrt(var[0] * var[k + 1]))
            want = float(np.exp(-s ** 2 / (2 * (l * aniso[k]) ** 2)))
            assert abs(corr - want) < TOL, (space, k, corr, want)


def test_gradient_variance(pkg, ob):
    """The derivative of an SE field has variance sigma^2 / l^2 per axis: pins the gradient chain
    (splattingKernel3D's gradient part, GPF.cpp:804-817, and the l2w transform of SCN.cpp:305)."""
    for space in ("world", "iso_ray"):
        params = _params(pkg, space, 16)
        orc = ob.Oracle(params, threads=8)
        sigma, l = float(params["sigma"]), float(params["length_scale"])
        q = np.zeros(N, dtype=pkg.QUERY)
        q["p"] = (0.2, -0.4, 0.1)
        q["dir"] = (0.0, 0.6, 0.8)
        q["pixel"][:, 0] = np.arange(N) % 4096
        q["pixel"][:, 1] = np.arange(N) // 4096
        q["scene_seed"] = 0xBA5EBA11
        g = orc.eval_gradient(q).astype(np.float64)
        var = g.var(axis=0) / (sigma / l) ** 2
        assert np.all(np.abs(var - 1.0) < 3 * TOL), (space, var)
        c = np.corrcoef(g.T)
        assert abs(c[0, 1]) < TOL and abs(c[0, 2]) < TOL and abs(c[1, 2]) < TOL


def test_variance_field_scales_the_amplitude(pkg, ob):
    """proc_nonstationary "var": the field's standard deviation at p is var(p) * sigma (GPF.cpp:1235-1237, 1638-1641), var
    being the bottom_top ramp sqrt(exp(lerp(log((min+1)^2), log((max+1)^2), clamp((y - start) / (end - start))))) - 1
    (GPF.cpp:43-55)."""
    params = _params(pkg, "world", 16)
    params["nonstationary"] = 1
    params["multi_resolution_grid"] = 0
    params["ls_ramp_type"] = 0
    params["ls_min"], params["ls_max"], params["ls_start"], params["ls_end"] = 1.0, 1.0, -1.0, 1.0      # constant length scale
    params["var"]["enabled"] = 1
    params["var"]["type"] = 0
    params["var"]["min"], params["var"]["max"], params["var"]["start"], params["var"]["end"] = 0.5, 2.0, -1.0, 1.0
    orc = ob.Oracle(params, threads=8)
    sigma = float(params["sigma"])
    ys = [-1.5, -0.5, 0.0, 0.5, 1.5]
    pts = [np.array([0.2, y, -0.1]) for y in ys]
    vals = (_ensemble(pkg, orc, pts, (0.0, 0.0, 1.0)) - 0.25) / sigma
    for y, v in zip(ys, vals):
        u = min(max((y + 1.0) / 2.0, 0.0), 1.0)
        var = np.sqrt(np.exp(np.log(1.5 ** 2) * (1 - u) + np.log(3.0 ** 2) * u)) - 1.0
        assert abs(v.std() / var - 1.0) < TOL, (y, v.std(), var)


@pytest.mark.parametrize("kernel", ["matern_0.5", "matern_2.5", "gabor_aniso", "gabor_iso"])
def test_other_kernels_are_normalised(pkg, ob, kernel):
    """Matérn (v = 1/2, 5/2) and Gabor splatting kernels with their own sparseConvNoiseVariance3D (GPF.cpp:1029-1046, 1134-1138,
    1199-1203): the normalised field has unit variance.  gabor_iso is the exception THE REFERENCE makes: its constant has
    exp(-2 pi f / a^2) where the integral of the squared kernel gives exp(-2 pi f^2 / a^2) (GPF.cpp:1201), so the field's
    variance is off by the ratio of the two brackets — reproduced here as the reference computes it."""
    p = pkg.params_for_config("C0")
    p["single_realization"] = 0
    p["impulse_density"] = 16
    p["mean"]["type"] = pkg.MEAN_TYPE.HOMOGENEOUS
    p["mean"]["offset"] = 0.25
    want = 1.0
    if kernel.startswith("matern"):
        p["kernel_type"], p["matern_v"] = 1, float(kernel.split("_")[1])
    else:
        p["kernel_type"] = 2 if kernel == "gabor_aniso" else 3
        p["gabor_a_inv"], p["gabor_f_inv"], p["gabor_omega"] = 0.08, 0.06, (0.0, 1.0, 0.0)
        if kernel == "gabor_iso":
            a, f = 1 / 0.08, 1 / 0.06
            want = np.sqrt((1 - np.exp(-2 * np.pi * f * f / (a * a))) / (1 - np.exp(-2 * np.pi * f / (a * a))))
    orc = ob.Oracle(p, threads=8)
    vals = (_ensemble(pkg, orc, [np.array([0.3, -0.2, 0.1]), np.array([-0.6, 0.4, 0.2])], (0.0, 0.0, 1.0)) - 0.25) / float(p["sigma"])
    # v = 1/2: the splatting kernel is exp(-r/l)/r (GPF.cpp:1052), singular at the impulse: k^4 is not integrable in 3D, the field
    # has no fourth moment and the sample variance converges slowly — wider band for that one kernel
    tol = 0.08 if kernel == "matern_0.5" else TOL
    assert np.all(np.abs(vals.std(axis=1) / want - 1.0) < tol), (kernel, vals.std(axis=1), want)
    # the gradient is the derivative of the value: central difference on one realisation.  Not for v = 1/2: the reference's
    # gradient there is exp(-r/l) (1/r^3 - 1/(r^2 l)) (GPF.cpp:1069) where d/dr of exp(-r/l)/r gives a PLUS; restated as written.
    if kernel == "matern_0.5":
        return
    # The truncated kernels make the field piecewise smooth (a lattice cell entering the neighbourhood adds its impulses at once,
    # SCN.cpp:136-149), so the difference quotient is taken with a small step at several points and the median is compared.
    h, nb = 1e-4, 9
    q = np.zeros(3 * nb, dtype=pkg.QUERY)
    base = np.array([0.3, -0.2, 0.1]) + np.arange(nb)[:, None] * np.array([0.0137, 0.0071, -0.0093])
    q["p"][0::3] = base
    q["p"][1::3] = base + (h, 0, 0)
    q["p"][2::3] = base - (h, 0, 0)
    q["dir"] = (0, 0, 1); q["pixel"] = (5, 6); q["scene_seed"] = 0xBA5EBA11
    v, _ = orc.eval_value(q)
    g = orc.eval_gradient(q[0::3])[:, 0]
    fd = (v[1::3].astype(np.float64) - v[2::3]) / (2 * h)
    err = np.abs(fd - g) / np.maximum(1.0, np.abs(g))
    assert np.median(err) < 0.03, (kernel, fd, g)


@pytest.mark.parametrize("space,a", [("world", 0.0), ("world", 1.0), ("iso_ray", 0.5)])
def test_aniso_field_turns_the_kernel(pkg, ob, space, a):
    """proc_nonstationary "aniso" (GPF.cpp:1600-1602, 1678-1689): the field gives an angle a * pi/2 about z by which the in-plane
    anisotropy (1.5 along the first axis, 1/1.5 along the second) is turned.  The covariance of the normalised field is then
    exp(-d^T R^T diag(1/1.5^2, 1.5^2, 1) R d / (2 l^2)) with unit variance (the matrix has determinant 1)."""
    params = _params(pkg, space, 16)
    params["nonstationary"] = 1
    params["multi_resolution_grid"] = 0
    params["ls_ramp_type"] = 0
    params["ls_min"], params["ls_max"], params["ls_start"], params["ls_end"] = 1.0, 1.0, -1.0, 1.0      # constant length scale
    params["aniso_field"]["enabled"] = 1
    params["aniso_field"]["type"] = 0
    params["aniso_field"]["min"], params["aniso_field"]["max"] = a, a                                      # constant angle
    params["aniso_field"]["start"], params["aniso_field"]["end"] = -1.0, 1.0
    orc = ob.Oracle(params, threads=8)
    assert orc.derived()["kernel_radius_world"] == np.float32(np.float32(3.0) * np.float32(1.5) * np.float32(np.float32(0.05) * np.sqrt(np.float32(2)) / 2))
    sigma, l = float(params["sigma"]), float(params["length_scale"])
    base = np.array([0.31, -0.17, 0.23])
    ang = a * np.pi / 2
    e1 = np.array([np.cos(ang), np.sin(ang), 0.0])       # long axis (1.5 l)
    e2 = np.array([-np.sin(ang), np.cos(ang), 0.0])      # short axis (l / 1.5)
    h = 0.04
    pts = [base, base + h * e1, base + h * e2, base + np.array([0, 0, h])]
    # In isotropic-ray space the reference applies the matrix to offsets expressed in the RAY's tangent frame (SCN.cpp:296-305
    # hands p_iso_ray to the kernel, GPF.cpp:776-786 does not rotate aniso_inv): the anisotropy axes follow the ray.  For a ray
    # along +z the Duff frame is the world axes, which is where the closed form below holds.
    direction = (0.0, 0.0, 1.0) if space == "iso_ray" else (0.3, 0.2, 0.9)
    vals = (_ensemble(pkg, orc, pts, direction) - 0.25) / sigma
    # ... and there the kernel radius stays 3 (splattingKernelRadius returns _kernelScale for the identity space before the 1.5
    # of sparseConvNoiseMaxAnisotropyScale is used, GPF.cpp:696-699, 1245-1249) while the kernel's long axis has standard
    # deviation 1.5: it is cut at exp(-2), so the closed form only holds to a few percent — wider band.
    tol = 2 * TOL if space == "iso_ray" else TOL
    assert np.all(np.abs(vals.std(axis=1) - 1.0) < tol), vals.std(axis=1)
    for i, stretch in ((1, 1.5), (2, 1 / 1.5), (3, 1.0)):
        want = np.exp(-h * h / (2 * (l * stretch) ** 2))
        got = float(np.mean(vals[0] * vals[i]))
        assert abs(got - want) < tol, (space, a, i, got, want)
