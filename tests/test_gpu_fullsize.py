"""Full-size C1 frame (1920x1080, 64 spp = BASELINE.json configs[1]) through size-independent
properties: determinism, invariance under row sharding and under the kernel path (guided vs exact),
a checksum of per-row checksums, and 64 probe pixels compared bit for bit with the oracle."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, SPP = 1920, 1080, 64


def _render(pkg, med, scene, parts=None):
    import torch
    rad = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    hits = torch.zeros(H * W, dtype=torch.int32, device="cuda")
    for y0, yc in (parts or [(0, H)]):
        part = scene.copy()
        part["y_begin"], part["y_count"] = y0, yc
        med.call("gpis_render_scene_s", part.ctypes.data_as(ctypes.c_void_p), rad.data_ptr(), hits.data_ptr(), None)
    torch.cuda.synchronize()
    return rad.cpu().numpy().reshape(H, W), hits.cpu().numpy().reshape(H, W)


def test_c1_full_frame_properties(pkg, ob):
    params = pkg.params_for_config("C1")
    med = pkg.Medium(params)
    scene = np.zeros((), dtype=pkg.SCENE_S)
    med.L.lib.gpis_default_scene_s(scene.ctypes.data, W, H, SPP)
    med.build_guide(16, 32)
    img, hits = _render(pkg, med, scene)
    # determinism + sharding invariance (16-pixel tile rows dealt to 3 "ranks", rendered in turn)
    rows = [(y, min(16, H - y)) for y in range(0, H, 16)]
    order = rows[0::3] + rows[1::3] + rows[2::3]
    img2, hits2 = _render(pkg, med, scene, order)
    assert np.array_equal(img, img2) and np.array_equal(hits, hits2)
    # the multi-GPU driver's form of the same split: ONE call per rank with shard_index / shard_count
    import torch
    rad3 = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    hits3 = torch.zeros(H * W, dtype=torch.int32, device="cuda")
    for r in range(3):
        part = scene.copy()
        part["shard_index"], part["shard_count"] = r, 3
        med.call("gpis_render_scene_s", part.ctypes.data_as(ctypes.c_void_p), rad3.data_ptr(), hits3.data_ptr(), None)
    torch.cuda.synchronize()
    assert np.array_equal(img, rad3.cpu().numpy().reshape(H, W)) and np.array_equal(hits, hits3.cpu().numpy().reshape(H, W))
    # checksum of checksums (sum of per-row sums in float64 is order independent here)
    assert float(img.astype(np.float64).sum(axis=1).sum()) == float(img2.astype(np.float64).sum(axis=1).sum())
    # plausible picture: the sphere covers the centre, radiance is bounded by spp * cos <= spp
    assert hits[H // 2, W // 2] == SPP and hits[0, 0] < SPP
    assert 0 < img.max() <= SPP and img.min() >= 0
    # the guide resolution the headline is measured with (bench.py --guide 16:64: side 2048, 34 GB): the SAME frame, bit for bit
    med.build_guide(16, 64)
    img64, hits64 = _render(pkg, med, scene)
    assert np.array_equal(img64, img) and np.array_equal(hits64, hits)
    # the exact (unguided) kernels give the same frame on a band of rows
    med.drop_guide()
    band = [(520, 32)]
    exact, _ = _render(pkg, med, scene, band)
    assert np.array_equal(exact[520:552], img[520:552])
    # probe pixels against the oracle (64 spp each)
    orc = ob.Oracle(params, threads=16)
    rng = np.random.default_rng(3)
    ys = rng.integers(0, H, 8)
    for y in ys:
        xs = rng.integers(0, W, 8)
        for x in xs:
            acc, nh = np.float32(0), 0
            for k in range(SPP):
                ok, ray, u = orc.scene_s_primary(scene, int(x), int(y), k)
                if not ok:
                    continue
                o = orc.sample_distance(ray[None])[0]
                nh += int(o["ok"] and not o["exited"])
                from gpu_util import shadow_rays_from
                sh = shadow_rays_from(ob, scene, ray[None], np.array([u], dtype=np.float32), o[None])
                if len(sh):
                    l = np.asarray(scene["light_dir"], dtype=np.float32)
                    l = l / np.float32(np.sqrt(np.float32((l * l).sum())))
                    a = o["aniso"]
                    n = (a / np.sqrt((a * a).sum())).astype(np.float32)
                    # same association as the estimator: s = n.x*l.x; s += n.y*l.y; s += n.z*l.z
                    c = np.float32(np.float32(np.float32(n[0] * l[0]) + np.float32(n[1] * l[1])) + np.float32(n[2] * l[2]))
                    vis = orc.transmittance(sh)[0]
                    acc = np.float32(acc + np.float32(np.float32(c * np.float32(1.0 if vis else 0.0)) * np.float32(1.0)))
            assert hits[y, x] == nh, (x, y)
            assert img[y, x] == acc, (x, y, img[y, x], acc)
