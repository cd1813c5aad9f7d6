"""The certified guide field: its error bound must hold everywhere (self-check against the exact
lattice sum on ~10^6 points), and the guided march must give the same bytes as the oracle."""
import ctypes

import numpy as np
import pytest

from gpu_util import to_dev, scene_rays, shadow_rays_from

pytestmark = pytest.mark.gpu


def _cluster_points(rng, n_groups, half):
    """groups of 64 points within a fraction of a cell (the evaluator's coherent case), group centres
    uniform over the tabulated volume, including cell faces and the volume's border"""
    centres = rng.uniform(-half + 0.05, half - 0.15, (n_groups, 1, 3))
    centres[:8, 0, :] = np.round(centres[:8, 0, :])            # on lattice cell corners
    pts = centres + rng.uniform(0, 0.08, (n_groups, 64, 3))
    return pts.reshape(-1, 3).astype(np.float32)


@pytest.mark.parametrize("cfg,ppc,dense", [("C1", 32, False), ("C1", 32, True), ("C1", 8, False), ("C0", 16, False), ("C1", 64, False)])
def test_guide_bound_holds(pkg, cfg, ppc, dense):
    """Both bounds of the certificate against the exact lattice sum on ~10^6 points: |N| <= amax in EVERY block of the field (level 0:
    the sign of the mean decides), and |N - G| <= Err in the tabulated bricks (level 1).  The field is stored in bricks and only those
    near the zero level set of the mean are tabulated (ppc >= 32; `dense` = GPIS_GUIDE_DENSE: every brick), so the mean's sphere is
    shrunk to put the surface inside the 5-cell field of this test."""
    import os
    half = 5
    params = pkg.params_for_config(cfg)
    params["mean"]["radius"] = 0.32          # 3 of the field's 5 cells
    med = pkg.Medium(params)
    if dense:
        os.environ["GPIS_GUIDE_DENSE"] = "1"
    try:
        med.build_guide(half, ppc)
    finally:
        os.environ.pop("GPIS_GUIDE_DENSE", None)
    info = med.guide_info()
    rng = np.random.default_rng(7)
    pts = _cluster_points(rng, 16384, half)
    d = to_dev(pts)
    checked, bad, ratio, bound = med.guide_selfcheck(d.data_ptr(), len(pts))
    tab = med.guide_info()["selfcheck_points_tabulated"]
    print("guide %s ppc=%d%s: %d of %d bricks tabulated (%.2f of %.2f GB), checked %d (%d in tabulated bricks), violations %d, max |err|/bound %.3f, mean bound %.3f" % (
        cfg, ppc, " dense" if dense else "", info["bricks_allocated"], info["bricks_total"], info["bytes_samples"] / 1e9, info["bytes_dense"] / 1e9,
        checked, tab, bad, ratio, bound))
    assert checked > 0.95 * len(pts)
    assert bad == 0 and ratio < 1.0
    assert 0 < bound < 50 and tab > 0.05 * checked
    if dense or ppc < 32:
        assert info["bricks_allocated"] == info["bricks_total"] and tab == checked
    else:
        # the a-priori bound comes from the field at ppc / 4: tight at 16 points per cell (ppc = 64), loose at 8 (this 5-cell field
        # lies within its reach of the surface almost everywhere)
        assert info["bricks_usable"] < (0.8 if ppc == 64 else 1.0) * info["bricks_total"] and info["bricks_usable"] <= info["bricks_allocated"]


def _far_bundle(rng, template, dist, n=256):
    """n rays whose origins lie `dist` world units from the medium and whose segments [dist - 1.5, dist + 1.5] cross it:
    large ray parameters and origin coordinates in fp32 (the guide's index coordinates are anchored at near_t for these)"""
    far = template[:n].copy()
    d = np.column_stack([rng.uniform(-0.02, 0.02, n), rng.uniform(-0.02, 0.02, n), np.ones(n)])
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    far["pos"] = (dist * d).astype(np.float32)
    aim = -d + rng.uniform(-0.1 / dist, 0.1 / dist, (n, 3))
    far["dir"] = (aim / np.linalg.norm(aim, axis=1, keepdims=True)).astype(np.float32)
    far["near_t"] = dist - 1.5
    far["far_t"] = dist + 1.5
    return far


def _raycheck_bundles(pkg, ob, med, orc, cfg, label):
    scene = ob.default_scene_s(320, 180, 1)
    rays, us = scene_rays(ob, orc, scene, step=3)
    sh = shadow_rays_from(ob, scene, rays, us, orc.sample_distance(rays))
    rng = np.random.default_rng(11)
    # bundles whose near_t is 0, i.e. whose anchor is the distant origin itself: 15 world units (141 cells) away the
    # unanchored magnitudes are still inside the error budget and steps are certified; 500 units away nothing is certified
    near15 = _far_bundle(rng, rays, 15.0)
    near15["near_t"] = 0.0
    unanchored = _far_bundle(rng, rays, 500.0, n=64)
    unanchored["near_t"] = 0.0
    total = 0
    for name, batch, steps in (("camera", rays, 400), ("shadow", sh, 400), ("far 44", _far_bundle(rng, rays, 44.0), 400),
                               ("far 500", _far_bundle(rng, rays, 500.0), 400), ("far 2000", _far_bundle(rng, rays, 2000.0), 400),
                               ("origin 15, near_t 0", near15, 1700), ("origin 500, near_t 0", unanchored, 50300)):
        d = to_dev(batch)
        certified, bad = med.guide_raycheck(d.data_ptr(), len(batch), steps)
        print("%s %s %s: %d rays, %d certified steps, %d violations" % (cfg, label, name, len(batch), certified, bad))
        assert bad == 0, name
        if name.startswith("far") or name.startswith("origin 15"):
            assert certified > 50 * len(batch), name          # the far bundles ARE certified (their steps cross the field)
        if name.startswith("origin 500"):
            assert certified == 0, name
        total += certified
    assert total > 50 * len(rays)


@pytest.mark.parametrize("cfg,half,ppc", [("C1", 16, 32), ("C0", 16, 16), ("C1", 6, 8)])
def test_certified_signs_agree_with_exact_values(pkg, ob, cfg, half, ppc):
    """The certificate as the march uses it (index coordinates linear in t and anchored at near_t, fp32 mean, sigma/norm
    folded): every certified sign along real camera and shadow rays, and along bundles that start 44 / 500 / 2000 world
    units away, equals the exact value's sign."""
    params = pkg.params_for_config(cfg)
    med, orc = pkg.Medium(params), ob.Oracle(params, threads=16)
    med.build_guide(half, ppc)
    _raycheck_bundles(pkg, ob, med, orc, cfg, "%d:%d" % (half, ppc))


def test_headline_guide_configuration_16_64(pkg, ob):
    """The guide resolution bench.py runs the headline on (16:64: side 2048, 34 GB, the 24-bit row index and the 64-bit element
    index at their largest): bound self-check over the whole tabulated volume, certificate ray-check on every bundle, and the
    guided march against the oracle, field by field."""
    params = pkg.params_for_config("C1")
    med, orc = pkg.Medium(params), ob.Oracle(params, threads=16)
    med.build_guide(16, 64)
    rng = np.random.default_rng(17)
    pts = _cluster_points(rng, 16384, 16)
    d = to_dev(pts)
    checked, bad, ratio, bound = med.guide_selfcheck(d.data_ptr(), len(pts))
    info = med.guide_info()
    print("guide C1 16:64: %d of %d bricks tabulated = %.2f GB of samples + %.2f GB of bounds (dense: %.1f GB); checked %d (%d in tabulated bricks), "
          "violations %d, max |err|/bound %.3f, mean bound %.3f" % (info["bricks_allocated"], info["bricks_total"], info["bytes_samples"] / 1e9,
                                                                     info["bytes_bounds"] / 1e9, info["bytes_dense"] / 1e9, checked,
                                                                     info["selfcheck_points_tabulated"], bad, ratio, bound))
    assert checked > 0.95 * len(pts) and bad == 0 and ratio < 1.0
    assert info["selfcheck_points_tabulated"] > 0.02 * checked
    assert info["bytes_samples"] + info["bytes_bounds"] <= 10e9          # the whole guide: <= 10 GB (34.4 + 1.1 GB as a dense grid)
    _raycheck_bundles(pkg, ob, med, orc, "C1", "16:64")
    scene = ob.default_scene_s(480, 270, 2)
    rays, us = scene_rays(ob, orc, scene, step=6)
    want = orc.sample_distance(rays)
    got = med.sample_distance(rays)
    for f in got.dtype.names:
        assert np.array_equal(got[f], want[f], equal_nan=True), f
    sh = shadow_rays_from(ob, scene, rays, us, want)
    assert np.array_equal(med.transmittance(sh), orc.transmittance(sh))
    # far bundles through the march itself (anchored index coordinates): bit-identical to the oracle
    far = np.concatenate([_far_bundle(rng, rays, 500.0), _far_bundle(rng, rays, 2000.0)])
    got, want = med.sample_distance(far), orc.sample_distance(far)
    for f in got.dtype.names:
        assert np.array_equal(got[f], want[f], equal_nan=True), f
    assert np.array_equal(med.transmittance(far), orc.transmittance(far))


def test_guided_march_matches_oracle_and_saves_evaluations(pkg, ob):
    params = pkg.params_for_config("C1")
    med, orc = pkg.Medium(params), ob.Oracle(params, threads=16)
    scene = ob.default_scene_s(480, 270, 2)
    rays, us = scene_rays(ob, orc, scene, step=6)
    want = orc.sample_distance(rays)
    sh = shadow_rays_from(ob, scene, rays, us, want)
    vis_o = orc.transmittance(sh)
    med.reset_counters()
    base = med.sample_distance(rays)
    e_exact = med.counters()[0]
    med.build_guide(16, 32)
    med.reset_counters()
    got = med.sample_distance(rays)
    e_guided, n_guide = med.counters()[0], med.guide_steps()
    for f in got.dtype.names:
        assert np.array_equal(got[f], want[f], equal_nan=True), f
        assert np.array_equal(got[f], base[f], equal_nan=True), f
    assert np.array_equal(med.transmittance(sh), vis_o)
    print("exact evaluations: %d unguided, %d guided (+%d certified steps) for %d segments" % (e_exact, e_guided, n_guide, len(rays)))
    assert e_guided < 0.35 * e_exact
    # every evaluation of the exact march is either certified or performed; the only extra work is the
    # one re-evaluation of a certified previous step when a crossing is refined (at most one per segment)
    assert 0 <= e_guided + n_guide - e_exact <= len(rays)
    # the same rays in a random order through the entry for scattered batches (wavefront form of the march)
    perm = np.random.default_rng(3).permutation(len(rays))
    med.set_batch_order(True)
    shuffled = med.sample_distance(rays[perm])
    for f in shuffled.dtype.names:
        assert np.array_equal(shuffled[f], want[f][perm], equal_nan=True), f
    assert np.array_equal(med.transmittance(sh), vis_o)
    med.set_batch_order(False)
    assert med.L.lib.gpis_set_batch_order(med.h, 7) == -1
    med.drop_guide()
    again = med.sample_distance(rays)
    assert np.array_equal(again["t"], want["t"])


def test_guide_argument_errors(pkg):
    med = pkg.Medium(pkg.params_for_config("C1"))
    with pytest.raises(RuntimeError, match="gpis_build_guide"):
        med.build_guide(16, 12)          # points per cell must be 8/16/32/64
    p = pkg.params_for_config("C2")      # per-path realizations: no shared field exists
    med2 = pkg.Medium(p)
    with pytest.raises(RuntimeError, match="not covered"):
        med2.build_guide(8, 8)
